// tools/rowprobe.hip -- memory-pattern probe for pass 1 (diagnostics, not part of the library):
// same wave -> (z, 32-row band) mapping and buffer addressing as k_row_pass_wave, no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <int MODE, int ROWS_PER_ITER>
__global__ void __launch_bounds__(256) k_probe(const uint32_t *__restrict__ labels, float *__restrict__ out,
                                               int sx, int sy, int sz, int nby, int ngroups) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t sxy = (int64_t)sx * sy;
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int z = grp / nby, yb = grp - z * nby, y0 = yb * 32;
    const uint32_t *base = labels + ((int64_t)z * sy + y0) * sx;
    const rsrc_t rl = make_rsrc(base), rb = make_rsrc(z > 0 ? base - sxy : base), ro = make_rsrc(out + ((int64_t)z * sy + y0) * sx);
#pragma unroll 1
    for (int r = 0; r < 32; r += ROWS_PER_ITER) {
      uint32_t v[ROWS_PER_ITER][8];
#pragma unroll
      for (int k = 0; k < ROWS_PER_ITER; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint32_t soff = (uint32_t)((r + k) * sx) * 4u, xo = (uint32_t)(c * 64 + lane) * 4u;
          uint32_t a = __builtin_amdgcn_raw_buffer_load_b32(rl, xo, soff, 0);
          if (MODE >= 1) a += __builtin_amdgcn_raw_buffer_load_b32(rb, xo, soff, 0);
          if (MODE >= 2) a += __builtin_amdgcn_raw_buffer_load_b32(rl, xo > 0 ? xo - 4 : 0, soff, 0);
          v[k][c] = a;
        }
#pragma unroll
      for (int k = 0; k < ROWS_PER_ITER; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          __builtin_amdgcn_raw_buffer_store_b32(v[k][c], ro, (uint32_t)(c * 64 + lane) * 4u, (uint32_t)((r + k) * sx) * 4u, 0);
    }
  }
}
// plain grid-stride float4 copy for reference
__global__ void __launch_bounds__(256) k_copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
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
  uint32_t *lab; float *out;
  CHECK(hipMalloc(&lab, vox * 4)); CHECK(hipMalloc(&out, vox * 4));
  CHECK(hipMemset(lab, 1, vox * 4));
  const int nby = n / 32, ngroups = nby * n;
  for (int blocks : {2048, 1024, 4096}) {
    printf("blocks=%d\n", blocks);
    printf("  lab only        1 row/iter: %.3f ms\n", timeit(k_probe<0, 1>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
    printf("  lab+below       1 row/iter: %.3f ms\n", timeit(k_probe<1, 1>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
    printf("  lab+below+left  1 row/iter: %.3f ms\n", timeit(k_probe<2, 1>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
    printf("  lab+below       2 row/iter: %.3f ms\n", timeit(k_probe<1, 2>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
    printf("  lab+below       4 row/iter: %.3f ms\n", timeit(k_probe<1, 4>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
    printf("  lab only        4 row/iter: %.3f ms\n", timeit(k_probe<0, 4>, dim3(blocks), dim3(256), lab, out, n, n, n, nby, ngroups));
  }
  printf("float4 copy 512MB->512MB: %.3f ms\n", timeit(k_copy4, dim3(256 * 8), dim3(256), (const float4 *)lab, (float4 *)out, vox / 4));
  return 0;
}
