// tools/tileprobe.hip -- memory pattern of the column pass over a 1024-row axis (diagnostics):
// does a 16-column tile (64-byte row pieces, 68 KiB of LDS, two workgroups per CU) move data as fast
// as the 32-column tile (128-byte rows, 136 KiB, one workgroup per CU), and how much of a compute
// phase between load and store does each hide?   hipcc --offload-arch=gfx950 -O3 tileprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

// volume (sx, n, nouter): column tiles of TC columns x n rows, row stride sx floats, outer stride sx*n
template <int TC, int THREADS, int MAP, int DMA = 0>
__global__ void __launch_bounds__(THREADS) k_tile(float *F, size_t sx, int n, int tiles_x, int delay, size_t ostride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f *tile = reinterpret_cast<v4f *>(smem);
  constexpr int GPR = TC / 4;            // 16-byte granules per tile row
  constexpr int RPP = THREADS / GPR;     // rows per pass
  int b = blockIdx.x;
  if (MAP == 1) {  // the two halves of a 128-byte line back to back on ONE XCD (XCD = blockIdx % 8)
    const int x = b & 7, j = b >> 3;
    b = ((j >> 1) * 8 + x) * 2 + (j & 1);
  }
  if (MAP == 2) {  // XCD x walks the outer indices congruent to x (mod 8), all x-tiles of one back to back
    const int x = b & 7, j = b >> 3;
    b = ((j / tiles_x) * 8 + x) * tiles_x + (j % tiles_x);
  }
  const int xt = b % tiles_x, o = b / tiles_x;
  float *base = F + (size_t)o * ostride + (size_t)xt * TC;  // rows are sx floats apart
  const int g = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
  if (DMA) {
    // direct global -> LDS loads (what the library's kernel uses): lane l of a wave lands at base + 16*l
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int WAVES = THREADS / 64, RPI = 64 / GPR;  // rows per wave instruction
    for (int i = wave; i * RPI < n; i += WAVES) {
      const int r = i * RPI + lane / GPR;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)r * sx + (lane % GPR) * 4),
                                       (__attribute__((address_space(3))) void *)(tile + (size_t)i * 64), 16, 0, DMA == 2 ? 2 : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int r = r0; r < n; r += RPP) tile[r * GPR + g] = *reinterpret_cast<const v4f *>(base + (size_t)r * sx + g * 4);
  }
  __syncthreads();
  for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(10);
  __syncthreads();
  for (int r = r0; r < n; r += RPP) {
    v4f v = tile[r * GPR + g];
    v.x += 1.0f;
    *reinterpret_cast<v4f *>(base + (size_t)r * sx + g * 4) = v;
  }
}

template <typename K>
float timeit(K k, dim3 g, dim3 b, size_t lds, float *F, size_t sx, int n, int tiles_x, int delay, size_t ostride) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, F, sx, n, tiles_x, delay, ostride);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, F, sx, n, tiles_x, delay, ostride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main() {
  const size_t vox = (size_t)1024 * 1024 * 128;
  float *F; hipMalloc(&F, vox * 4); hipMemset(F, 0, vox * 4);
  // y pass of (1024, 1024, 128): 1024-row tiles, rows 4 KiB apart
  {
    const size_t rs = 1024, os = (size_t)1024 * 1024; const int n = 1024, nouter = 128, cols = 1024;
    printf("1024-row axis (y pass of 1024 x 1024 x 128), no compute\n");
    printf("  32-col: regs %.3f  dma %.3f  dma-nt %.3f ms\n",
      timeit(k_tile<32, 1024, 0, 0>, dim3(cols / 32 * nouter), dim3(1024), (size_t)n * 128 + 8192, F, rs, n, cols / 32, 0, os),
      timeit(k_tile<32, 1024, 0, 1>, dim3(cols / 32 * nouter), dim3(1024), (size_t)n * 128 + 8192, F, rs, n, cols / 32, 0, os),
      timeit(k_tile<32, 1024, 0, 2>, dim3(cols / 32 * nouter), dim3(1024), (size_t)n * 128 + 8192, F, rs, n, cols / 32, 0, os));
    printf("  16-col paired: regs %.3f  dma %.3f ms\n",
      timeit(k_tile<16, 512, 1, 0>, dim3(cols / 16 * nouter), dim3(512), (size_t)n * 64 + 4096, F, rs, n, cols / 16, 0, os),
      timeit(k_tile<16, 512, 1, 1>, dim3(cols / 16 * nouter), dim3(512), (size_t)n * 64 + 4096, F, rs, n, cols / 16, 0, os));
  }
  // y pass of 512^3: 512-row tiles (64 KiB, two workgroups per CU), 512 threads
  {
    const size_t rs = 512, os = (size_t)512 * 512; const int n = 512, nouter = 512, cols = 512;
    printf("512-row axis (y pass of 512^3), no compute\n");
    printf("  32-col: regs %.3f  dma %.3f  dma-nt %.3f ms\n",
      timeit(k_tile<32, 512, 0, 0>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rs, n, cols / 32, 0, os),
      timeit(k_tile<32, 512, 0, 1>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rs, n, cols / 32, 0, os),
      timeit(k_tile<32, 512, 0, 2>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rs, n, cols / 32, 0, os));
    // z pass of 512^3: rows 1 MiB apart
    const size_t rz = (size_t)512 * 512, oz = 512;
    printf("512-row axis (z pass of 512^3), no compute\n");
    printf("  32-col: regs %.3f  dma %.3f  dma-nt %.3f ms\n",
      timeit(k_tile<32, 512, 0, 0>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rz, n, cols / 32, 0, oz),
      timeit(k_tile<32, 512, 0, 1>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rz, n, cols / 32, 0, oz),
      timeit(k_tile<32, 512, 0, 2>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rz, n, cols / 32, 0, oz));
  }
  // 512-row axis with the XCD-aware order of the library: 32-column tiles (2 workgroups per CU) against
  // 16-column tiles (4 per CU), without and with a compute phase
  {
    const size_t rs = 512, os = (size_t)512 * 512; const int n = 512, nouter = 512, cols = 512;
    printf("512-row axis (y pass of 512^3), XCD-aware order\n");
    for (int delay : {0, 8, 16}) {
      printf("  delay %2d: 32-col %.3f ms   16-col %.3f ms\n", delay,
        timeit(k_tile<32, 512, 2, 1>, dim3(cols / 32 * nouter), dim3(512), (size_t)n * 128 + 4096, F, rs, n, cols / 32, delay, os),
        timeit(k_tile<16, 256, 2, 1>, dim3(cols / 16 * nouter), dim3(256), (size_t)n * 64 + 2048, F, rs, n, cols / 16, delay / 2, os));
    }
  }
  return 0;
}
