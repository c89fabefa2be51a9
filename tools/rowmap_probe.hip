// tools/rowmap_probe.hip -- VERDICT r4 item 3: does another wave -> voxel MAPPING of pass X move its bytes faster?
// Bare access patterns at pass X's byte mix on a 512^3 uint32 volume (read 4 B/voxel of labels -- plus the re-reads of the left
// neighbour and of the slice below that the kernel makes -- write 2 B/voxel of indices), no arithmetic beyond keeping the
// loads alive.  Diagnostics, not part of the library.   hipcc --offload-arch=gfx950 -O3 -o rowmap_probe rowmap_probe.hip
//   A  the kernel's mapping: a wave = 32 consecutive rows of one slice, a row = 8 x (4 B per lane), 2-byte stores,
//      XCD-aware order (y-bands congruent to the XCD, z in order), loads of row r+1 in flight under row r
//   B  A with 16 B per lane (a row = 2 loads of 1 KiB), 8-byte stores
//   C  workgroup-contiguous: a workgroup of 256 lanes streams one contiguous block of 128 rows (256 KiB) with 16 B per
//      lane, 4 KiB per instruction of the workgroup, 8-byte stores; plain and XCD-aware block order
//   D  B with two rows in flight per wave
// `below`: 0 = labels only; 1 = + the slice below (an L2 hit when the order is right); 2 = + the left neighbour too (A only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void group_of(int i, int nby, int sz, bool by_xcd, int &z, int &yb) {
  if (by_xcd) {
    const int xcd = blockIdx.x & 7, nyk = (nby - xcd + 7) >> 3;
    z = i / nyk; yb = xcd + 8 * (i - z * nyk);
  } else { z = i / nby; yb = i - z * nby; }
}
// ---- A: the kernel's mapping -------------------------------------------------------------------
template <int BELOW>
__global__ void __launch_bounds__(256) k_A(const uint32_t *__restrict__ lab, uint16_t *__restrict__ out, int sx, int sy, int sz, int by_xcd) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nby = sy / 32; const int64_t sxy = (int64_t)sx * sy;
  const int xcd = blockIdx.x & 7, nyk = by_xcd ? (nby - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) * 4 + wave : (int)blockIdx.x * 4 + wave;
  const int step = by_xcd ? (int)(gridDim.x >> 3) * 4 : (int)gridDim.x * 4;
  const int count = by_xcd ? nyk * sz : nby * sz;
  for (int i = first; i < count; i += step) {
    int z, yb; group_of(i, nby, sz, by_xcd, z, yb);
    const uint32_t *base = lab + ((int64_t)z * sy + yb * 32) * sx;
    const rsrc_t rl = make_rsrc(base), rb = make_rsrc(z > 0 ? base - sxy : base), ro = make_rsrc(out + ((int64_t)z * sy + yb * 32) * sx);
    uint32_t v[8], nx[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t xo = (uint32_t)(c * 64 + lane) * 4u;
      v[c] = __builtin_amdgcn_raw_buffer_load_b32(rl, xo, 0, 0);
      if (BELOW >= 1) v[c] += __builtin_amdgcn_raw_buffer_load_b32(rb, xo, 0, 0);
      if (BELOW >= 2) v[c] += __builtin_amdgcn_raw_buffer_load_b32(rl, xo > 0 ? xo - 4 : 0, 0, 0);
    }
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
      const uint32_t soff = (uint32_t)((r + 1 < 32 ? r + 1 : r) * sx) * 4u;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t xo = (uint32_t)(c * 64 + lane) * 4u;
        nx[c] = __builtin_amdgcn_raw_buffer_load_b32(rl, xo, soff, 0);
        if (BELOW >= 1) nx[c] += __builtin_amdgcn_raw_buffer_load_b32(rb, xo, soff, 0);
        if (BELOW >= 2) nx[c] += __builtin_amdgcn_raw_buffer_load_b32(rl, xo > 0 ? xo - 4 : 0, soff, 0);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v[c], ro, (uint32_t)(c * 64 + lane) * 2u, (uint32_t)(r * sx) * 2u, 0);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = nx[c];
    }
  }
}
// ---- B / D: 16 B per lane, DEPTH rows in flight ---------------------------------------------------
template <int BELOW, int DEPTH>
__global__ void __launch_bounds__(256) k_B(const uint32_t *__restrict__ lab, uint16_t *__restrict__ out, int sx, int sy, int sz, int by_xcd) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nby = sy / 32; const int64_t sxy = (int64_t)sx * sy;
  const int xcd = blockIdx.x & 7, nyk = by_xcd ? (nby - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) * 4 + wave : (int)blockIdx.x * 4 + wave;
  const int step = by_xcd ? (int)(gridDim.x >> 3) * 4 : (int)gridDim.x * 4;
  const int count = by_xcd ? nyk * sz : nby * sz;
  for (int i = first; i < count; i += step) {
    int z, yb; group_of(i, nby, sz, by_xcd, z, yb);
    const uint32_t *base = lab + ((int64_t)z * sy + yb * 32) * sx;
    const rsrc_t rl = make_rsrc(base), rb = make_rsrc(z > 0 ? base - sxy : base), ro = make_rsrc(out + ((int64_t)z * sy + yb * 32) * sx);
    v4u v[DEPTH][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t xo = (uint32_t)(c * 256 + lane * 4) * 4u, so = (uint32_t)(d * sx) * 4u;
        v[d][c] = __builtin_amdgcn_raw_buffer_load_b128(rl, xo, so, 0);
        if (BELOW >= 1) v[d][c] += __builtin_amdgcn_raw_buffer_load_b128(rb, xo, so, 0);
      }
#pragma unroll 1
    for (int r = 0; r < 32; r += DEPTH) {
      v4u nx[DEPTH][2];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int rn = r + DEPTH + d < 32 ? r + DEPTH + d : 31;
        const uint32_t so = (uint32_t)(rn * sx) * 4u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint32_t xo = (uint32_t)(c * 256 + lane * 4) * 4u;
          nx[d][c] = __builtin_amdgcn_raw_buffer_load_b128(rl, xo, so, 0);
          if (BELOW >= 1) nx[d][c] += __builtin_amdgcn_raw_buffer_load_b128(rb, xo, so, 0);
        }
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const v2u p = {(v[d][c][0] & 0xFFFFu) | (v[d][c][1] << 16), (v[d][c][2] & 0xFFFFu) | (v[d][c][3] << 16)};
          __builtin_amdgcn_raw_buffer_store_b64(p, ro, (uint32_t)(c * 256 + lane * 4) * 2u, (uint32_t)((r + d) * sx) * 2u, 0);
        }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c) v[d][c] = nx[d][c];
    }
  }
}
// ---- C: workgroup-contiguous blocks of 128 rows ----------------------------------------------------
template <int BELOW>
__global__ void __launch_bounds__(256) k_C(const uint32_t *__restrict__ lab, uint16_t *__restrict__ out, int sx, int sy, int sz, int by_xcd) {
  const int t = threadIdx.x;
  const int nbk = sy / 128; const int64_t sxy = (int64_t)sx * sy;  // blocks of 128 rows per slice
  const int xcd = blockIdx.x & 7, nyk = by_xcd ? (nbk - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int count = by_xcd ? nyk * sz : nbk * sz;
  for (int i = first; i < count; i += step) {
    int z, kb;
    if (by_xcd) { z = i / nyk; kb = xcd + 8 * (i - z * nyk); } else { z = i / nbk; kb = i - z * nbk; }
    const uint32_t *base = lab + ((int64_t)z * sy + kb * 128) * sx;
    const rsrc_t rl = make_rsrc(base), rb = make_rsrc(z > 0 ? base - sxy : base), ro = make_rsrc(out + ((int64_t)z * sy + kb * 128) * sx);
    // 128 rows x 512 voxels = 65536 voxels = 64 iterations of 256 lanes x 4 voxels; 4 iterations in flight
    constexpr int U = 4;
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t xo = (uint32_t)(t * 4) * 4u, so = (uint32_t)(u * 1024) * 4u;
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rl, xo, so, 0);
      if (BELOW >= 1) v[u] += __builtin_amdgcn_raw_buffer_load_b128(rb, xo, so, 0);
    }
#pragma unroll 1
    for (int it = 0; it < 64; it += U) {
      v4u nx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int in = it + U + u < 64 ? it + U + u : 63;
        const uint32_t xo = (uint32_t)(t * 4) * 4u, so = (uint32_t)(in * 1024) * 4u;
        nx[u] = __builtin_amdgcn_raw_buffer_load_b128(rl, xo, so, 0);
        if (BELOW >= 1) nx[u] += __builtin_amdgcn_raw_buffer_load_b128(rb, xo, so, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const v2u p = {(v[u][0] & 0xFFFFu) | (v[u][1] << 16), (v[u][2] & 0xFFFFu) | (v[u][3] << 16)};
        __builtin_amdgcn_raw_buffer_store_b64(p, ro, (uint32_t)(t * 4) * 2u, (uint32_t)((it + u) * 1024) * 2u, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = nx[u];
    }
  }
}
template <typename K, typename... A>
float timeit(K k, dim3 g, dim3 b, A... a) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
  const int n = 512; const size_t vox = (size_t)n * n * n;
  uint32_t *lab; uint16_t *out;
  CHECK(hipMalloc(&lab, vox * 4)); CHECK(hipMalloc(&out, vox * 2));
  CHECK(hipMemset(lab, 1, vox * 4));
  printf("# 512^3: 512 MiB of labels read, 256 MiB of 16-bit values written; ms per pass (20 launches)\n");
  for (int blocks : {2048, 1024}) {
    for (int xcd : {1, 0}) {
      printf("blocks=%d order=%s\n", blocks, xcd ? "xcd-aware" : "plain");
      printf("  A  kernel mapping, 4 B/lane, labels only           %.4f\n", timeit(k_A<0>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  A  ... + slice below                               %.4f\n", timeit(k_A<1>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  A  ... + slice below + left neighbour (= pass X)   %.4f\n", timeit(k_A<2>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  B  16 B/lane, 8-byte stores, labels only           %.4f\n", timeit(k_B<0, 1>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  B  ... + slice below                               %.4f\n", timeit(k_B<1, 1>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  D  B with two rows in flight, labels only          %.4f\n", timeit(k_B<0, 2>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  D  ... + slice below                               %.4f\n", timeit(k_B<1, 2>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  C  workgroup-contiguous 256 KiB blocks, labels only %.4f\n", timeit(k_C<0>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
      printf("  C  ... + slice below                               %.4f\n", timeit(k_C<1>, dim3(blocks), dim3(256), lab, out, n, n, n, xcd));
    }
  }
  return 0;
}
