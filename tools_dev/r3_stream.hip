// dev microbenchmark: streaming rates of this box (read-only, write-only, copy; 16 B per lane), the ceiling the
// kernels' L2-miss traffic is compared with (hipcc --offload-arch=gfx950 -O3 tools_dev/r3_stream.hip -o tools_dev/r3_stream)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n, float* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * 4) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (MODE != 1) v[u] = j < n ? (NT ? __builtin_nontemporal_load(src + j) : src[j]) : acc;
      else v[u] = (f4){(float)j, 1, 2, 3};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (MODE == 0) acc += v[u];
      else if (j < n) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; }
    }
  }
  if (MODE == 0 && acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}

template <int MODE, int NT>
void run(const char* name, const f4* src, f4* dst, size_t n, float* sink, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, n, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double bytes = (double)n * 16 * (MODE == 2 ? 2 : 1);
  printf("%-28s %5d blocks: %7.3f ms  %6.2f TB/s\n", name, blocks, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t n = (size_t)1 << 27;  // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
  f4 *a, *b; float* sink;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
  for (int blocks : {2048, 8192}) {
    run<0, 0>("read", a, b, n, sink, blocks);
    run<0, 1>("read nt", a, b, n, sink, blocks);
    run<1, 0>("write", a, b, n, sink, blocks);
    run<1, 1>("write nt", a, b, n, sink, blocks);
    run<2, 0>("copy (read + write bytes)", a, b, n, sink, blocks);
    run<2, 1>("copy nt", a, b, n, sink, blocks);
  }
  return 0;
}
