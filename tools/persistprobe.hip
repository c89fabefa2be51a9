// tools/persistprobe.hip -- does a PERSISTENT workgroup (a loop over tiles) move the Y-pass tiles of 512^3 faster than
// one workgroup per tile?  Memory pattern of the column kernel only: 32-column tiles filled by global->LDS DMA,
// a pause standing in for the compute phase, write-back; XCD-aware tile order.  (diagnostics, not part of the library)
// hipcc --offload-arch=gfx950 -O3 tools/persistprobe.hip -o tools/persistprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool PERSIST>
__global__ void __launch_bounds__(512) k_tile(float *F, int sx, int n, int tiles_x, int total, int delay) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f *tile = reinterpret_cast<v4f *>(smem);
  constexpr int GPR = 8, THREADS = 512, RPP = THREADS / GPR;
  const size_t ostride = (size_t)sx * n;
  const int g = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = blockIdx.x; t < total; t += PERSIST ? gridDim.x : total) {
    int b;
    { const int x = t & 7, j = t >> 3; b = ((j / tiles_x) * 8 + x) * tiles_x + (j % tiles_x); }
    const int xt = b % tiles_x, o = b / tiles_x;
    float *base = F + (size_t)o * ostride + (size_t)xt * 32;
    for (int i = wave; i * 8 < n; i += 8) {
      const int r = i * 8 + lane / GPR;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)r * sx + (lane % GPR) * 4),
                                       (__attribute__((address_space(3))) void *)(tile + (size_t)i * 64), 16, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(10);
    __syncthreads();
    for (int r = r0; r < n; r += RPP) {
      v4f v = tile[r * GPR + g];
      v.x += 1.0f;
      *reinterpret_cast<v4f *>(base + (size_t)r * sx + g * 4) = v;
    }
    __syncthreads();  // (the tile is free again once every thread has read its rows out)
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
  const int total = 16 * n;
  const size_t lds = 80 * 1024;  // as the library's kernel: two workgroups per CU
  for (int rep = 0; rep < 2; ++rep)
    for (int delay : {0, 8, 16, 24}) {
      printf("delay %2d: one workgroup per tile %.3f ms   persistent x512 %.3f ms   persistent x1024 %.3f ms\n", delay,
             timeit(k_tile<false>, dim3(total), dim3(512), lds, F, n, n, 16, total, delay),
             timeit(k_tile<true>, dim3(512), dim3(512), lds, F, n, n, 16, total, delay),
             timeit(k_tile<true>, dim3(1024), dim3(512), lds, F, n, n, 16, total, delay));
    }
  return 0;
}
