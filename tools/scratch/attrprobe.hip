#include <hip/hip_runtime.h>
#include <cstdio>
template <int V> __global__ void __launch_bounds__(256, 3) k3(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x] + V; }
template <int V> __global__ void __launch_bounds__(256, 2) k2(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x] + V; }
template <int V> __global__ void __launch_bounds__(256) k1(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x] + V; }
template <int V> __global__ void __launch_bounds__(256, 4) k4(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x] + V; }
int main() {
  float *p; hipMalloc(&p, 1024);
  for (int kb : {64, 80, 100, 128, 160}) {
    printf("%d KB: k1 %s | ", kb, hipGetErrorString(hipFuncSetAttribute((const void*)&k1<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024)));
    printf("k2 %s | ", hipGetErrorString(hipFuncSetAttribute((const void*)&k2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024)));
    printf("k3 %s | ", hipGetErrorString(hipFuncSetAttribute((const void*)&k3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024)));
    printf("k4 %s\n", hipGetErrorString(hipFuncSetAttribute((const void*)&k4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024)));
  }
  hipLaunchKernelGGL(k3<0>, dim3(1), dim3(256), 43000, 0, p); printf("launch k3 43000: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  hipLaunchKernelGGL(k3<0>, dim3(1), dim3(256), 70000, 0, p); printf("launch k3 70000: %s %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
