// tools/yorder_probe.hip -- round 6: the memory side of pass Y (16-bit indices in, 16-bit plane out IN PLACE, rows sx elements
// apart inside one z-slice) as a bare kernel: the library's shape (32-column tiles = 64-byte row pieces, 256 threads, four
// workgroups per CU) against wider tiles (64 / 128 columns: whole 128-byte lines and more), 8 or 16 bytes per thread, and loads
// or stores alone.  What is the floor of the pattern, and is it the pattern or the kernel that keeps pass Y at 3.5 TB/s?
//     hipcc --offload-arch=gfx950 -O3 -o yorder_probe yorder_probe.hip && ./yorder_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// TC columns per tile, VB bytes per thread and row, T threads; XCD-aware order as the library's (edt_colq16.hip)
template <int T, int TC, int VB>
__global__ void __launch_bounds__(T) k_y(uint16_t *__restrict__ K, int n, int sx, int64_t pitch, int tiles_x, int nouter, int what,
                                         const uint32_t *__restrict__ rs = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename std::conditional<VB == 16, v4u, v2u>::type V;
  V *img = reinterpret_cast<V *>(smem);
  constexpr int TPR = TC * 2 / VB;  // threads per row
  constexpr int RPS = T / TPR;      // rows per sweep
  const uint32_t b = blockIdx.x, utx = (uint32_t)tiles_x;
  const uint32_t x = b & 7u, j = b >> 3, jq = j / utx, jr = j - jq * utx;
  const uint32_t tile = (jq * 8u + x) * utx + jr;
  if (tile >= utx * (uint32_t)nouter) return;
  const uint32_t o = tile / utx, xt = tile - o * utx;
  const int t = threadIdx.x, r_in = t / TPR, cg = t % TPR;
  uint16_t *base = K + (int64_t)o * pitch + xt * TC + cg * (VB / 2);
  if (what & 8) {
    // the run-start words of the tile ahead of the fill, as the library has them: [outer][band][x], a loop of load -> LDS store
    uint32_t *rsp = reinterpret_cast<uint32_t *>(smem + (size_t)n * TC * 2);
    const int nb = n >> 5;
    for (int u = t; u < nb * TC; u += T) {
      const int band = u / TC, col = u % TC;
      rsp[u] = rs[((int64_t)o * nb + band) * sx + xt * TC + col];
    }
  }
  uint32_t rsw[4] = {0u, 0u, 0u, 0u};
  if (what & 16) {
    // the same words into registers, stored to LDS only after the fill's loads are out
    const int nb = n >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = t + k * T;
      if (u < nb * TC) rsw[k] = rs[((int64_t)o * nb + u / TC) * sx + xt * TC + u % TC];
    }
  }
  constexpr int NL = 16;
  for (int r0 = 0; r0 < n; r0 += RPS * NL) {
    V v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = r0 + r_in + i * RPS;
      v[i] = V{};
      if (row < n && (what & 1)) v[i] = *reinterpret_cast<const V *>(base + (int64_t)row * sx);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = r0 + r_in + i * RPS;
      if (row < n) img[row * TPR + cg] = v[i];
    }
  }
  if (what & 16) {
    uint32_t *rsp = reinterpret_cast<uint32_t *>(smem + (size_t)n * TC * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (t + k * T < (n >> 5) * TC) rsp[t + k * T] = rsw[k];
  }
  __syncthreads();
  for (int row = r_in; row < n; row += RPS) {
    V r = img[row * TPR + (cg ^ (row & (TPR - 1) & 7))];
    r[0] += 1u;
    if ((what & 2) || r[0] == 0x12345u) *reinterpret_cast<V *>(base + (int64_t)row * sx) = r;
  }
}

template <int T, int TC, int VB>
static float run(uint16_t *K, int sx, int sy, int sz, int what, int reps) {
  const int tiles_x = sx / TC;
  const uint32_t grid = (uint32_t)tiles_x * ((sz + 7) / 8 * 8);
  const size_t lds = (size_t)sy * TC * 2 + (size_t)(sy / 32) * TC * 4 + 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_y<T, TC, VB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() { hipLaunchKernelGGL((k_y<T, TC, VB>), dim3(grid), dim3(T), lds, 0, K, sy, sx, (int64_t)sx * sy, tiles_x, sz, what); };
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

// the neighbours of pass Y in a step, as bare streams: pass X (4 B labels in, 2 B indices out), pass Z's bytes (2 B in, 4 B out)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_xlike(const v4u *__restrict__ lab, v2u *__restrict__ K, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const v4u a = __builtin_nontemporal_load(lab + i);
    K[i] = (v2u){(a[0] & 0xFFFFu) | (a[1] << 16), (a[2] & 0xFFFFu) | (a[3] << 16)};
  }
}
__global__ void __launch_bounds__(256) k_zlike(const v2u *__restrict__ K, v4f *__restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const v2u a = K[i];
    __builtin_nontemporal_store((v4f){(float)(a[0] & 0xFFFFu), (float)(a[0] >> 16), (float)(a[1] & 0xFFFFu), (float)(a[1] >> 16)}, out + i);
  }
}

// pass Y's pattern INSIDE a step: X-like stream, Y, Z-like stream, repeated; the time of Y alone (events around it)
template <int T, int TC, int VB>
static void in_step(uint16_t *K, uint32_t *lab, float *out, int sx, int sy, int sz, const char *name, int what = 3) {
  const int64_t n4 = (int64_t)sx * sy * sz / 4;
  const int tiles_x = sx / TC;
  const uint32_t grid = (uint32_t)tiles_x * ((sz + 7) / 8 * 8);
  const size_t lds = (size_t)sy * TC * 2 + (size_t)(sy / 32) * TC * 4 + 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_y<T, TC, VB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
  float tx = 0, ty = 0, tz = 0;
  const int reps = 20;
  for (int i = 0; i < reps + 3; ++i) {
    hipEventRecord(e[0]);
    hipLaunchKernelGGL(k_xlike, dim3(8192), dim3(256), 0, 0, (const v4u *)lab, (v2u *)K, n4);
    hipEventRecord(e[1]);
    hipLaunchKernelGGL((k_y<T, TC, VB>), dim3(grid), dim3(T), lds, 0, K, sy, sx, (int64_t)sx * sy, tiles_x, sz, what, (const uint32_t *)out);
    hipEventRecord(e[2]);
    hipLaunchKernelGGL(k_zlike, dim3(8192), dim3(256), 0, 0, (const v2u *)K, (v4f *)out, n4);
    hipEventRecord(e[3]); hipEventSynchronize(e[3]);
    if (i >= 3) { float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]); tx += a; ty += b; tz += c; }
  }
  printf("  in a step (X-like stream, Y, Z-like stream): %-44s X %.4f  Y %.4f  Z %.4f ms\n", name, tx / reps, ty / reps, tz / reps);
}

int main() {
  struct Shape { int sx, sy, sz; } shapes[] = {{512, 512, 512}, {1024, 1024, 128}};
  for (const Shape &s : shapes) {
    const int64_t vox = (int64_t)s.sx * s.sy * s.sz;
    uint16_t *K;
    if (hipMalloc(&K, vox * 2) != hipSuccess) return 1;
    hipMemset(K, 1, vox * 2);
    const double bytes = (double)vox * 4;
    printf("shape %d x %d x %d: 2 B in + 2 B out per voxel, in place\n", s.sx, s.sy, s.sz);
#define GO(T, TC, VB, name)                                                                                              \
    {                                                                                                                   \
      const float l = run<T, TC, VB>(K, s.sx, s.sy, s.sz, 1, 20), st = run<T, TC, VB>(K, s.sx, s.sy, s.sz, 2, 20),       \
                  bo = run<T, TC, VB>(K, s.sx, s.sy, s.sz, 3, 20);                                                       \
      printf("  %-58s loads %.4f  stores %.4f  both %.4f ms  %.2f TB/s\n", name, l, st, bo, bytes / bo / 1e9);          \
    }
    if (s.sy <= 512) {
      GO(256, 32, 8, "32 columns, 256 threads, 8 B/thread (the library)");
      GO(512, 32, 8, "32 columns, 512 threads, 8 B/thread");
      GO(256, 64, 16, "64 columns, 256 threads, 16 B/thread");
      GO(512, 64, 8, "64 columns, 512 threads, 8 B/thread");
      GO(512, 64, 16, "64 columns, 512 threads, 16 B/thread");
      GO(1024, 64, 8, "64 columns, 1024 threads, 8 B/thread");
      GO(512, 128, 16, "128 columns, 512 threads, 16 B/thread");
      GO(1024, 128, 16, "128 columns, 1024 threads, 16 B/thread");
      GO(256, 16, 8, "16 columns, 256 threads, 8 B/thread");
    } else {
      GO(512, 32, 8, "32 columns, 512 threads, 8 B/thread (the library)");
      GO(256, 32, 8, "32 columns, 256 threads, 8 B/thread");
      GO(1024, 32, 8, "32 columns, 1024 threads, 8 B/thread");
      GO(512, 64, 16, "64 columns, 512 threads, 16 B/thread");
      GO(1024, 64, 8, "64 columns, 1024 threads, 8 B/thread");
      GO(1024, 64, 16, "64 columns, 1024 threads, 16 B/thread");
      GO(256, 16, 8, "16 columns, 256 threads, 8 B/thread");
    }
    uint32_t *lab; float *out;
    if (hipMalloc(&lab, vox * 4) != hipSuccess || hipMalloc(&out, vox * 4) != hipSuccess) return 1;
    hipMemset(lab, 1, vox * 4);
    if (s.sy <= 512) {
      in_step<256, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "32 columns, 256 threads (the library)");
      in_step<256, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "the library + run-start words AHEAD of the fill", 3 | 8);
      in_step<256, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "the library + run-start words held in registers", 3 | 16);
      in_step<512, 64, 16>(K, lab, out, s.sx, s.sy, s.sz, "64 columns, 512 threads, 16 B/thread");
      in_step<256, 16, 8>(K, lab, out, s.sx, s.sy, s.sz, "16 columns, 256 threads");
    } else {
      in_step<512, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "32 columns, 512 threads (the library)");
      in_step<512, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "the library + run-start words AHEAD of the fill", 3 | 8);
      in_step<512, 32, 8>(K, lab, out, s.sx, s.sy, s.sz, "the library + run-start words held in registers", 3 | 16);
      in_step<1024, 64, 16>(K, lab, out, s.sx, s.sy, s.sz, "64 columns, 1024 threads, 16 B/thread");
    }
    hipFree(lab); hipFree(out);
    hipFree(K);
  }
  return 0;
}
