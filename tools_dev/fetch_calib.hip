// dev: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for 4/8/16-byte-per-lane streams
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> __global__ void copy_kernel(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// named per width so that the profile rows can be told apart
__global__ void calib_copy_4B(const float* a, float* b, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
__global__ void calib_copy_8B(const f2* a, f2* b, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
__global__ void calib_copy_16B(const f4* a, f4* b, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB in, 1 GiB out: far beyond the 256 MiB Infinity Cache
  char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_copy_4B, dim3(4096), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4);
    hipLaunchKernelGGL(calib_copy_8B, dim3(4096), dim3(256), 0, 0, (const f2*)a, (f2*)b, bytes / 8);
    hipLaunchKernelGGL(calib_copy_16B, dim3(4096), dim3(256), 0, 0, (const f4*)a, (f4*)b, bytes / 16);
  }
  CK(hipDeviceSynchronize());
  printf("calib done: each kernel reads and writes %zu bytes\n", bytes);
  return 0;
}
