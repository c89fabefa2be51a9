// tools/u16probe.hip -- would a 16-bit X->Y intermediate (the distance INDEX instead of its squared fp32 value)
// pay?  Memory patterns only (diagnostics, not part of the library):
//   rows:  pass-1 pattern (labels + slice below + left neighbour in, one row per iteration) writing 4 B per voxel,
//          2 B per voxel, or 2 B per voxel packed as row pairs [z][y/2][x][2] (one dword store every second row);
//   tiles: Y-pass pattern of 512^3 (32-column tiles, XCD-aware order) filled by global->LDS DMA from fp32 (in place),
//          or through VGPRs from a u16 plane (8 B per lane) / from the row-pair plane (16 B per lane) with the
//          conversion index -> (w * k)^2, results written back as fp32.
// hipcc --offload-arch=gfx950 -O3 tools/u16probe.hip -o tools/u16probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

template <int OUT>
__global__ void __launch_bounds__(256) k_rows(const uint32_t *__restrict__ labels, void *__restrict__ out,
                                              int sx, int sy, int sz, int nby, int ngroups) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t sxy = (int64_t)sx * sy;
  const int xcd = blockIdx.x & 7, nyk = (nby - xcd + 7) >> 3;
  const int first = (blockIdx.x >> 3) * 4 + wave, step = (gridDim.x >> 3) * 4, count = nyk * sz;
  for (int i = first; i < count; i += step) {
    const int z = i / nyk, yb = xcd + 8 * (i - z * nyk), y0 = yb * 32;
    const uint32_t *base = labels + ((int64_t)z * sy + y0) * sx;
    const rsrc_t rl = make_rsrc(base), rb = make_rsrc(z > 0 ? base - sxy : base);
    const rsrc_t ro = make_rsrc((char *)out + ((int64_t)z * sy + y0) * sx * (OUT == 0 ? 4 : 2));
    uint32_t prev[8];
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
      uint32_t v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t soff = (uint32_t)(r * sx) * 4u, xo = (uint32_t)(c * 64 + lane) * 4u;
        uint32_t a = __builtin_amdgcn_raw_buffer_load_b32(rl, xo, soff, 0);
        a += __builtin_amdgcn_raw_buffer_load_b32(rb, xo, soff, 0);
        a += __builtin_amdgcn_raw_buffer_load_b32(rl, xo > 0 ? xo - 4 : 0, soff, 0);
        v[c] = a;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t x = (uint32_t)(c * 64 + lane);
        if (OUT == 0) __builtin_amdgcn_raw_buffer_store_b32(v[c], ro, x * 4u, (uint32_t)(r * sx) * 4u, 0);
        if (OUT == 1) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v[c], ro, x * 2u, (uint32_t)(r * sx) * 2u, 0);
        if (OUT == 2) {  // rows r-1, r of a pair leave together: [y/2][x][2]
          if (r & 1) __builtin_amdgcn_raw_buffer_store_b32((prev[c] & 0xFFFFu) | (v[c] << 16), ro, x * 4u, (uint32_t)((r >> 1) * sx) * 4u, 0);
          prev[c] = v[c];
        }
      }
    }
  }
}

__device__ __forceinline__ float conv(uint32_t k, float w) {
  float d = (float)k * w;
  float f = d * d;
  return k == 0xFFFFu ? INFINITY : f;
}

// SRC 0: fp32 in place by DMA; 1: u16 plane H[z][y][x]; 2: row-pair plane H[z][y/2][x][2]
template <int SRC>
__global__ void __launch_bounds__(512) k_tile(float *F, const uint16_t *__restrict__ H, int sx, int n, int tiles_x, int delay, float w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f *tile = reinterpret_cast<v4f *>(smem);
  constexpr int GPR = 8, THREADS = 512, RPP = THREADS / GPR;
  int b = blockIdx.x;
  { const int x = b & 7, j = b >> 3; b = ((j / tiles_x) * 8 + x) * tiles_x + (j % tiles_x); }
  const int xt = b % tiles_x, o = b / tiles_x;
  const size_t ostride = (size_t)sx * n;
  float *base = F + (size_t)o * ostride + (size_t)xt * 32;
  const int g = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
  if (SRC == 0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i * 8 < n; i += 8) {
      const int r = i * 8 + lane / GPR;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)r * sx + (lane % GPR) * 4),
                                       (__attribute__((address_space(3))) void *)(tile + (size_t)i * 64), 16, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (SRC == 1) {
    const uint16_t *hb = H + (size_t)o * ostride + (size_t)xt * 32;
    v2u q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(hb + (size_t)(r0 + k * RPP) * sx + g * 4));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v4f v;
      v.x = conv(q[k].x & 0xFFFFu, w); v.y = conv(q[k].x >> 16, w); v.z = conv(q[k].y & 0xFFFFu, w); v.w = conv(q[k].y >> 16, w);
      tile[(r0 + k * RPP) * GPR + g] = v;
    }
  } else {
    const uint16_t *hb = H + (size_t)o * ostride + (size_t)xt * 64;  // a pair row holds 2 * sx entries
    v4u q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(hb + (size_t)(r0 + k * RPP) * sx * 2 + g * 8));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pr = r0 + k * RPP;  // pair row: rows 2pr, 2pr+1
      v4f a, c;
      a.x = conv(q[k].x & 0xFFFFu, w); c.x = conv(q[k].x >> 16, w);
      a.y = conv(q[k].y & 0xFFFFu, w); c.y = conv(q[k].y >> 16, w);
      a.z = conv(q[k].z & 0xFFFFu, w); c.z = conv(q[k].z >> 16, w);
      a.w = conv(q[k].w & 0xFFFFu, w); c.w = conv(q[k].w >> 16, w);
      tile[(2 * pr) * GPR + g] = a;
      tile[(2 * pr + 1) * GPR + g] = c;
    }
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

template <typename K, typename... A>
float timeit(K k, dim3 g, dim3 b, size_t lds, A... a) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, a...);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, g, b, lds, 0, a...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main() {
  const int n = 512; const size_t vox = (size_t)n * n * n;
  uint32_t *lab; float *F; uint16_t *H;
  hipMalloc(&lab, vox * 4); hipMalloc(&F, vox * 4); hipMalloc(&H, vox * 2);
  hipMemset(lab, 1, vox * 4); hipMemset(F, 0, vox * 4); hipMemset(H, 1, vox * 2);
  const int nby = n / 32, ngroups = nby * n;
  for (int rep = 0; rep < 2; ++rep) {
    printf("rows (pass-1 pattern, 2048 workgroups): fp32 out %.3f ms   u16 out %.3f ms   u16 row pairs %.3f ms\n",
           timeit(k_rows<0>, dim3(2048), dim3(256), 0, (const uint32_t *)lab, (void *)F, n, n, n, nby, ngroups),
           timeit(k_rows<1>, dim3(2048), dim3(256), 0, (const uint32_t *)lab, (void *)H, n, n, n, nby, ngroups),
           timeit(k_rows<2>, dim3(2048), dim3(256), 0, (const uint32_t *)lab, (void *)H, n, n, n, nby, ngroups));
    for (int delay : {0, 8, 16})
      printf("tiles (Y pass of 512^3) delay %2d: fp32 dma %.3f ms   u16 %.3f ms   u16 row pairs %.3f ms\n", delay,
             timeit(k_tile<0>, dim3(16 * n), dim3(512), (size_t)n * 128 + 4096, F, (const uint16_t *)H, n, n, 16, delay, 6.0f),
             timeit(k_tile<1>, dim3(16 * n), dim3(512), (size_t)n * 128 + 4096, F, (const uint16_t *)H, n, n, 16, delay, 6.0f),
             timeit(k_tile<2>, dim3(16 * n), dim3(512), (size_t)n * 128 + 4096, F, (const uint16_t *)H, n, n, 16, delay, 6.0f));
  }
  return 0;
}
