// tools/copysize.hip -- what does a plain streaming copy reach at the byte counts of ONE pass of a 512^3 step?  (A pass moves
// 0.5-0.8 GB; the 5.4-5.8 TB/s of copyprobe.hip are for 1.07 GB per launch: how much of the difference is the kernel's ramp-up
// and drain?)   hipcc --offload-arch=gfx950 -O3 -o copysize tools/copysize.hip && ./copysize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) k_copy_blocked(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U;
  v4f v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t j = base + u * 256 + threadIdx.x; if (j < n) v[u] = in[j]; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t j = base + u * 256 + threadIdx.x; if (j < n) out[j] = v[u]; }
}
// read `rd` 16-byte words per write word (pass X: 2 : 1, pass Z: 1 : 2 -> swap the roles)
template <int U, int RD>
__global__ void __launch_bounds__(256) k_mix(const v4f *__restrict__ in, v4f *__restrict__ out, size_t nout) {
  const size_t base = (size_t)blockIdx.x * 256 * U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t j = base + u * 256 + threadIdx.x;
    if (j < nout) {
      v4f acc = in[j * RD];
#pragma unroll
      for (int r = 1; r < RD; ++r) acc += in[j * RD + r];
      out[j] = acc;
    }
  }
}
int main() {
  const size_t maxb = (size_t)1 << 30;
  float *a, *b; hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMemset(a, 1, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {64, 128, 256, 512}) {  // MiB read = MiB written
    const size_t n4 = mb * 1024 * 1024 / 16;
    const unsigned blocks = (unsigned)((n4 + 1023) / 1024);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_copy_blocked<4>, dim3(blocks), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n4);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_copy_blocked<4>, dim3(blocks), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("copy %4zu MiB -> %4zu MiB: %.4f ms  %.2f TB/s\n", mb, mb, ms, 2.0 * mb * 1048576 / ms / 1e9);
  }
  {  // pass X's mix: 512 MiB read, 256 MiB written
    const size_t nout = (size_t)256 * 1024 * 1024 / 16;
    const unsigned blocks = (unsigned)((nout + 1023) / 1024);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mix<4, 2>), dim3(blocks), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, nout);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_mix<4, 2>), dim3(blocks), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, nout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("read 512 MiB, write 256 MiB (pass X's bytes): %.4f ms  %.2f TB/s\n", ms, 768.0 * 1048576 / ms / 1e9);
  }
  return 0;
}
