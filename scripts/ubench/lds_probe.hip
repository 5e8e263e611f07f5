#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) { extern __shared__ unsigned s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = s[(threadIdx.x + 1) % 64]; }
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int a = 0; hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
  printf("sharedMemPerBlock %zu attr %d maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, a, p.maxSharedMemoryPerMultiProcessor);
  unsigned *d; hipMalloc(&d, 1024);
  for (size_t lds : {48u*1024, 64u*1024, 96u*1024, 128u*1024, 160u*1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, d);
    hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    printf("lds %zu: launch %s sync %s\n", lds, hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 != hipSuccess) {
      hipError_t e3 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, d);
      e1 = hipGetLastError(); e2 = hipDeviceSynchronize();
      printf("   with attribute (%s): launch %s sync %s\n", hipGetErrorString(e3), hipGetErrorString(e1), hipGetErrorString(e2));
    }
  }
  return 0;
}
