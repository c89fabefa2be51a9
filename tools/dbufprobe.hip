// tools/dbufprobe.hip -- persistent workgroups with DOUBLE-BUFFERED tile fills for the Y-pass pattern of 512^3: while a
// workgroup works on (pauses over) and writes back tile t, the global->LDS loads of tile t+1 are already in flight in
// the other LDS buffer.  Variants: 32-column tiles, 2 x 64 KiB per workgroup, ONE workgroup of 512 threads per CU;
// 16-column tiles, 2 x 32 KiB, TWO workgroups of 256 threads per CU.  Against the library's shape (one workgroup per
// 32-column tile, two per CU, no prefetch).  Memory pattern only (diagnostics, not part of the library).
// hipcc --offload-arch=gfx950 -O3 tools/dbufprobe.hip -o tools/dbufprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int TC, int THREADS>
__device__ __forceinline__ void issue_fill(v4f *tile, const float *base, int sx, int n) {
  constexpr int GPR = TC / 4, RPI = 64 / GPR, WAVES = THREADS / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i * RPI < n; i += WAVES) {
    const int r = i * RPI + lane / GPR;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)r * sx + (lane % GPR) * 4),
                                     (__attribute__((address_space(3))) void *)(tile + (size_t)i * 64), 16, 0, 2);
  }
}

// one workgroup per tile (the library's shape)
template <int TC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_plain(float *F, int sx, int n, int tiles_x, int delay) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f *tile = reinterpret_cast<v4f *>(smem);
  constexpr int GPR = TC / 4, RPP = THREADS / GPR;
  int b = blockIdx.x;
  { const int x = b & 7, j = b >> 3; b = ((j / tiles_x) * 8 + x) * tiles_x + (j % tiles_x); }
  float *base = F + (size_t)(b / tiles_x) * sx * n + (size_t)(b % tiles_x) * TC;
  issue_fill<TC, THREADS>(tile, base, sx, n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(10);
  __syncthreads();
  const int g = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
  for (int r = r0; r < n; r += RPP) {
    v4f v = tile[r * GPR + g];
    v.x += 1.0f;
    *reinterpret_cast<v4f *>(base + (size_t)r * sx + g * 4) = v;
  }
}

// persistent, double-buffered
template <int TC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_dbuf(float *F, int sx, int n, int tiles_x, int total, int delay) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int GPR = TC / 4, RPP = THREADS / GPR;
  v4f *buf[2] = {reinterpret_cast<v4f *>(smem), reinterpret_cast<v4f *>(smem) + (size_t)n * GPR};
  auto tile_base = [&](int t) {
    int b;
    { const int x = t & 7, j = t >> 3; b = ((j / tiles_x) * 8 + x) * tiles_x + (j % tiles_x); }
    return F + (size_t)(b / tiles_x) * sx * n + (size_t)(b % tiles_x) * TC;
  };
  const int g = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
  int t = blockIdx.x, cur = 0;
  if (t < total) issue_fill<TC, THREADS>(buf[0], tile_base(t), sx, n);
  for (; t < total; t += gridDim.x, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t has landed (and the previous write-back has left)
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < total) issue_fill<TC, THREADS>(buf[cur ^ 1], tile_base(tn), sx, n);  // next tile: in flight during the work below
    for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(10);
    __syncthreads();
    float *base = tile_base(t);
    for (int r = r0; r < n; r += RPP) {
      v4f v = buf[cur][r * GPR + g];
      v.x += 1.0f;
      *reinterpret_cast<v4f *>(base + (size_t)r * sx + g * 4) = v;
    }
  }
}

template <typename K, typename... A>
float timeit(K k, dim3 g, dim3 b, size_t lds, A... a) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, a...);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, a...);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main() {
  const int n = 512; const size_t vox = (size_t)n * n * n;
  float *F; (void)hipMalloc(&F, vox * 4); (void)hipMemset(F, 0, vox * 4);
  for (int rep = 0; rep < 2; ++rep)
    for (int delay : {0, 8, 16, 24}) {
      printf("delay %2d: plain 32-col (2 wg/CU) %.3f | dbuf 32-col 1 wg/CU x256 %.3f | dbuf 16-col 2 wg/CU x512 %.3f | dbuf 16-col x1024 (4/CU would need 40 KiB) %.3f ms\n", delay,
             timeit(k_plain<32, 512>, dim3(16 * n), dim3(512), (size_t)80 * 1024, F, n, n, 16, delay),
             timeit(k_dbuf<32, 512>, dim3(256), dim3(512), (size_t)130 * 1024, F, n, n, 16, 16 * n, delay),
             timeit(k_dbuf<16, 256>, dim3(512), dim3(256), (size_t)66 * 1024, F, n, n, 32, 32 * n, delay / 2),
             timeit(k_dbuf<16, 256>, dim3(1024), dim3(256), (size_t)66 * 1024, F, n, n, 32, 32 * n, delay / 2));
    }
  return 0;
}
