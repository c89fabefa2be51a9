// tools/zorder_probe.hip -- round 6: the memory side of pass Z (16-bit plane in, fp32 rows out, rows one z-slice apart) as a bare
// kernel, under several workgroup -> tile orders.  At 1024^3 the rows of a tile are exactly 4 MiB (output) / 2 MiB (plane) apart
// and pass Z takes 30-40 % more per voxel than at 1024 x 1008 x 1024 (profiles/r04_stride_probe.txt).  The library's order keeps the
// tiles in flight inside a 64 KiB window of the slice (16 outer rows x 32 x-tiles); does an order that spreads them over the slice
// take the effect away?        hipcc --offload-arch=gfx950 -O3 -o zorder_probe zorder_probe.hip && ./zorder_probe
//
//   order 0: plain                      tile = blockIdx
//   order 1: the library's              XCD x = b % 8 takes the outer rows = x (mod 8); all x-tiles of one outer row back to back
//   order 2: (XF, S)                    as 1, but the XCD's groups of 8 outer rows are visited S apart (group' = (q % S) * K + q / S,
//                                       K = ceil(G / S)), and only XF x-tiles of an outer row are walked before the next group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <int T>
__global__ void __launch_bounds__(T) k_z(const uint16_t *__restrict__ plane, float *__restrict__ out, int n, int64_t st, int sx,
                                         int tiles_x, int nouter, int order, int XF, int S, int what = 3, int64_t pst = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v2u *img = reinterpret_cast<v2u *>(smem);
  uint32_t tile;
  const uint32_t b = blockIdx.x, utx = (uint32_t)tiles_x;
  if (order == 0) {
    tile = b;
  } else if (order == 1) {
    const uint32_t x = b & 7u, j = b >> 3, jq = j / utx, jr = j - jq * utx;
    tile = (jq * 8u + x) * utx + jr;
  } else {
    const uint32_t G = ((uint32_t)nouter + 7u) >> 3, K = (G + (uint32_t)S - 1u) / (uint32_t)S, GP = K * (uint32_t)S;  // padded groups
    const uint32_t x = b & 7u, j = b >> 3;
    const uint32_t xi = j % (uint32_t)XF, r = j / (uint32_t)XF;
    const uint32_t q = r % GP, xb = r / GP;
    const uint32_t grp = (q % (uint32_t)S) * K + q / (uint32_t)S;
    if (grp >= G) return;
    tile = (grp * 8u + x) * utx + xb * (uint32_t)XF + xi;
  }
  if (tile >= utx * (uint32_t)nouter) return;
  const uint32_t o = tile / utx, xt = tile - o * utx;
  const int t = threadIdx.x, r_in = t >> 3, cg = t & 7;
  constexpr int RPS = T / 8;
  const uint16_t *src = plane + (int64_t)o * sx + xt * 32 + 4 * cg;
  float *dst = out + (int64_t)o * sx + xt * 32 + 4 * cg;
  v2u v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r_in + i * RPS;
    v[i] = (v2u){1u, 2u};
    if (row < n && (what & 1)) v[i] = *reinterpret_cast<const v2u *>(src + (int64_t)row * pst);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r_in + i * RPS;
    if (row < n) img[row * 8 + cg] = v[i];
  }
  __syncthreads();
  for (int row = r_in; row < n; row += RPS) {
    const v2u r = img[row * 8 + (cg ^ (row & 7))];  // (something the compiler cannot fold into the loads)
    const v4f f = {(float)(r[0] & 0xFFFFu), (float)(r[0] >> 16), (float)(r[1] & 0xFFFFu), (float)(r[1] >> 16)};
    if (what & 4) *reinterpret_cast<v4f *>(dst + (int64_t)row * st) = f;
    else if ((what & 2) || f[0] == 12345.0f) __builtin_nontemporal_store(f, reinterpret_cast<v4f *>(dst + (int64_t)row * st));
  }
}

__global__ void __launch_bounds__(256) k_copy(const v4f *__restrict__ a, v4f *__restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) b[i] = a[i];
}

static float run(int T, const uint16_t *plane, float *out, int n, int64_t st, int sx, int nouter, int order, int XF, int S, int reps, int what = 3, int64_t pad = 0) {
  const int tiles_x = sx / 32;
  uint32_t grid;
  if (order == 2) {
    const uint32_t G = (nouter + 7) / 8, K = (G + S - 1) / S;
    grid = 8u * (uint32_t)XF * (K * S) * (uint32_t)(tiles_x / XF);
  } else {
    grid = (uint32_t)tiles_x * ((nouter + 7) / 8 * 8);
  }
  const size_t lds = (size_t)n * 64 + 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (T == 512) hipLaunchKernelGGL((k_z<512>), dim3(grid), dim3(512), lds, 0, plane, out, n, st, sx, tiles_x, nouter, order, XF, S, what, st + pad);
    else hipLaunchKernelGGL((k_z<256>), dim3(grid), dim3(256), lds, 0, plane, out, n, st, sx, tiles_x, nouter, order, XF, S, what, st + pad);
  };
  launch(); launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_z<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_z<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct Shape { int sx, sy, sz; } shapes[] = {{1024, 1024, 1024}, {1024, 1008, 1024}, {512, 512, 512}, {1024, 1024, 128}, {1024, 512, 1024}, {1024, 1024, 512}, {2048, 512, 512},
                                        {1024, 2048, 512}, {768, 768, 768}, {512, 1024, 1024}, {2048, 2048, 256}, {640, 512, 512}, {1024, 768, 1024}};
  for (const Shape &s : shapes) {
    const int64_t vox = (int64_t)s.sx * s.sy * s.sz, st = (int64_t)s.sx * s.sy;
    uint16_t *plane; float *out;
    if (hipMalloc(&plane, vox * 2 + (int64_t)s.sz * (1 << 20)) != hipSuccess || hipMalloc(&out, vox * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(plane, 1, vox * 2 + (int64_t)s.sz * (1 << 20));
    const int T = s.sz > 512 ? 512 : 256, reps = s.sz * (int64_t)s.sx > 600000 ? 5 : 20;
    const double bytes = (double)vox * 6;
    printf("shape %d x %d x %d (T = %d), 6 B/voxel\n", s.sx, s.sy, s.sz, T);
    {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int64_t n16 = vox * 2 / 16;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const v4f *)out, (v4f *)out + n16, n16);
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const v4f *)out, (v4f *)out + n16, n16);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float cms; hipEventElapsedTime(&cms, e0, e1); cms /= 5;
      printf("  plain copy of %.2f GB           %.4f ms  %.2f TB/s\n", vox * 4 / 1e9, cms, vox * 4 / cms / 1e9);
    }
    for (int what = 1; what <= 5; ++what) {
      if (what == 4) continue;
      const float wms = run(T, plane, out, s.sz, st, s.sx, s.sy, 1, 0, 0, reps, what);
      printf("  order 1, %s   %.4f ms\n", what == 1 ? "loads only        " : what == 2 ? "nt stores only    " : what == 3 ? "loads + nt stores " : "loads + plain stores", wms);
    }
    float ms = run(T, plane, out, s.sz, st, s.sx, s.sy, 0, 0, 0, reps);
    printf("  order 0 plain                 %.4f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = run(T, plane, out, s.sz, st, s.sx, s.sy, 1, 0, 0, reps);
    printf("  order 1 library               %.4f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    const int64_t pads[] = {0, 4096, 8192, 8192 + 65536, 8192 + 2048};  // bytes per slice of the plane
    for (int64_t pad : pads) {
      const float l = run(T, plane, out, s.sz, st, s.sx, s.sy, 1, 0, 0, reps, 1, pad / 2);
      const float b = run(T, plane, out, s.sz, st, s.sx, s.sy, 1, 0, 0, reps, 3, pad / 2);
      printf("  order 1, plane slices padded by %7lld bytes: loads only %.4f ms, loads + nt stores %.4f ms  %.2f TB/s\n", (long long)pad, l, b, bytes / b / 1e9);
    }
    const int xfs[] = {0};
    const int ss[] = {1, 8, 64};
    for (int XF : xfs) if (XF > 0) {
      if (XF > s.sx / 32) continue;
      for (int S : ss) {
        if (S > (s.sy + 7) / 8) continue;
        ms = run(T, plane, out, s.sz, st, s.sx, s.sy, 2, XF, S, reps);
        printf("  order 2 XF = %2d S = %2d        %.4f ms  %.2f TB/s\n", XF, S, ms, bytes / ms / 1e9);
      }
    }
    hipFree(plane); hipFree(out);
  }
  return 0;
}
