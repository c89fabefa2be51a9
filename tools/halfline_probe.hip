// tools/halfline_probe.hip -- does reading the 16-bit plane as 64-byte row pieces (32-column tiles) fetch more from memory than
// its bytes?  Three readers of the same 256 MiB array of 16-bit values [z][y][x] (512^3), all 8 bytes per lane, run under
// rocprofv3 --pmc FETCH_SIZE (and timed): (A) contiguous; (B) the column kernel's pattern -- a workgroup per (x-tile of 32
// columns, z), XCD-aware tile order, 8 threads x 4 columns per row; (C) the same with 64-column tiles (whole 128-byte lines).
// Diagnostics, not part of the library.   hipcc --offload-arch=gfx950 -O3 -o halfline_probe halfline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_contig(const v2u *__restrict__ in, uint32_t *__restrict__ out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v2u v = in[i]; acc += v[0] ^ v[1]; }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int COLS, bool NT>  // 32 or 64 columns per tile; NT: non-temporal loads (what the column kernels' fills use)
__global__ void __launch_bounds__(256) k_tiles(const uint16_t *__restrict__ in, uint32_t *__restrict__ out, int sx, int sy, int sz) {
  const uint32_t utx = (uint32_t)(sx / COLS);
  const uint32_t tt = blockIdx.x, x = tt & 7u, j = tt >> 3;
  const uint32_t jq = j / utx, jr = j - jq * utx;
  const uint32_t tile = (jq * 8u + x) * utx + jr;
  const uint32_t o = tile / utx, xt = tile - o * utx;
  if (o >= (uint32_t)sz) return;
  constexpr int TPR = COLS / 4;          // threads per row (4 columns = 8 bytes each)
  constexpr int RPS = 256 / TPR;         // rows per sweep
  const int t = threadIdx.x, r_in = t / TPR, cg = t % TPR;
  const uint16_t *src = in + (size_t)o * sx * sy + (size_t)xt * COLS + 4 * cg;
  uint32_t acc = 0;
  for (int i0 = 0; i0 < sy; i0 += RPS * 8) {
    v2u v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int row = i0 + RPS * k + r_in; v[k] = (v2u){0u, 0u}; if (row < sy) v[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2u *>(src + (size_t)row * sx)) : *reinterpret_cast<const v2u *>(src + (size_t)row * sx); }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k][0] ^ v[k][1];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <typename K, typename... A>
float timeit(K k, dim3 g, dim3 b, A... a) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
  const int n = 512; const size_t vox = (size_t)n * n * n;
  uint16_t *in; uint32_t *out;
  hipMalloc(&in, vox * 2); hipMalloc(&out, 256); hipMemset(in, 1, vox * 2);
  printf("A contiguous 8 B/lane                 %.4f ms\n", timeit(k_contig, dim3(2048), dim3(256), (const v2u *)in, out, vox / 4));
  printf("B 32-column tiles (64-byte pieces), nt loads     %.4f ms\n", timeit(k_tiles<32, true>, dim3((n / 32) * n), dim3(256), in, out, n, n, n));
  printf("B 32-column tiles (64-byte pieces), plain loads  %.4f ms\n", timeit(k_tiles<32, false>, dim3((n / 32) * n), dim3(256), in, out, n, n, n));
  printf("C 64-column tiles (128-byte pieces), nt loads    %.4f ms\n", timeit(k_tiles<64, true>, dim3((n / 64) * n), dim3(256), in, out, n, n, n));
  printf("C 64-column tiles (128-byte pieces), plain loads %.4f ms\n", timeit(k_tiles<64, false>, dim3((n / 64) * n), dim3(256), in, out, n, n, n));
  return 0;
}
