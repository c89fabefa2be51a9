// tools/copyprobe.hip -- what does a streaming read+write reach on this box? (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256) k_copy(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) v[u] = NT_LD ? __builtin_nontemporal_load(in + j) : in[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (j < n) { if (NT_ST) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u]; }
    }
  }
}
// each block owns a contiguous chunk (like torch's elementwise kernels)
template <int U>
__global__ void __launch_bounds__(256) k_copy_blocked(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U;
  v4f v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t j = base + u * 256 + threadIdx.x; if (j < n) v[u] = in[j]; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t j = base + u * 256 + threadIdx.x; if (j < n) out[j] = v[u]; }
}
template <typename K, typename... A>
float timeit(K k, dim3 g, dim3 b, A... a) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
  const size_t vox = (size_t)512 * 512 * 512, n4 = vox / 4;
  float *a, *b; hipMalloc(&a, vox * 4); hipMalloc(&b, vox * 4); hipMemset(a, 1, vox * 4);
  auto rep = [&](const char *name, float ms) { printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, 2.0 * vox * 4 / ms / 1e9); };
  for (int blocks : {2048, 8192, 32768}) {
    printf("grid-stride, blocks=%d\n", blocks);
    rep("  x1", timeit(k_copy<1, false, false>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
    rep("  x4", timeit(k_copy<4, false, false>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
    rep("  x4 nt-load", timeit(k_copy<4, true, false>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
    rep("  x4 nt-store", timeit(k_copy<4, false, true>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
    rep("  x4 nt both", timeit(k_copy<4, true, true>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
    rep("  x8", timeit(k_copy<8, false, false>, dim3(blocks), dim3(256), (const v4f *)a, (v4f *)b, n4));
  }
  rep("blocked x4 (32768 blocks)", timeit(k_copy_blocked<4>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), (const v4f *)a, (v4f *)b, n4));
  rep("blocked x8", timeit(k_copy_blocked<8>, dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), (const v4f *)a, (v4f *)b, n4));
  rep("blocked x2", timeit(k_copy_blocked<2>, dim3((unsigned)((n4 + 511) / 512)), dim3(256), (const v4f *)a, (v4f *)b, n4));
  // in place (read and write the same buffer, like the column passes)
  rep("in place blocked x4", timeit(k_copy_blocked<4>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), (const v4f *)a, (v4f *)a, n4));
  return 0;
}
