#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = s[(threadIdx.x * 7) % n]; }
int main() {
  int* d; hipMalloc(&d, 4096);
  for (int kb : {48, 64, 96, 128, 160}) {
    hipError_t a = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), kb * 1024, 0, d, 256);
    hipError_t e = hipGetLastError(); hipError_t s = hipDeviceSynchronize();
    printf("%d KB: attr %s launch %s sync %s\n", kb, hipGetErrorName(a), hipGetErrorName(e), hipGetErrorName(s));
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
}
