// dev microbenchmark: vector-memory (TA/L1) throughput of gather-shaped loads on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int T = 256;
constexpr int ITERS = 64;
typedef float float2_ __attribute__((ext_vector_type(2)));
typedef float float4_ __attribute__((ext_vector_type(4)));

// each block sweeps a private 64 KB window repeatedly (L2-resident after first touch, partly L1)
template <int MODE>
__global__ __launch_bounds__(T) void k(const float* __restrict__ src, float* out, int shift) {
  const float* base = src + (size_t)blockIdx.x * 16384;
  float acc = 0.f;
  int lane = threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    int row = (it * 5) & 31;
    if (MODE == 0) {  // 4 dword taps (2x2), lanes consecutive, arbitrary alignment
      const float* p = base + row * 512 + lane + shift;
      acc += p[0] + p[1] + p[512] + p[513];
    } else if (MODE == 1) {  // 2 dwordx2 taps (pairs), 4-byte aligned only
      const float* p = base + row * 512 + lane + shift;
      float2_ a, b;
      __builtin_memcpy(&a, p, 8);
      __builtin_memcpy(&b, p + 512, 8);
      acc += a.x + a.y + b.x + b.y;
    } else if (MODE == 2) {  // 4 aligned dwordx4 loads (streaming shape), same bytes/instr x4
      const float4_* p = reinterpret_cast<const float4_*>(base + row * 512) + lane;
      float4_ a = p[0], b = p[256], c = p[512 + 0], d = p[768];
      acc += a.x + b.y + c.z + d.w;
    } else if (MODE == 3) {  // 4 dword loads, fully coalesced aligned (no overlap between taps)
      const float* p = base + row * 512 + lane;
      acc += p[0] + p[256] + p[1024] + p[1280];
    }
  }
  out[blockIdx.x * T + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const float* src, int shift, int ninstr, int bytes_per_lane) {
  float* out; CK(hipMalloc(&out, 8192 * T * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * 16;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, src, out, shift);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, src, out, shift);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double waveinstr = (double)blocks * (T / 64) * ITERS * ninstr;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-40s shift %d: %7.3f ms  %6.1f cyc/wave-instr/CU  %7.2f TB/s L1->reg\n", name, shift, ms,
         cyc / (waveinstr / 256), waveinstr * 64 * bytes_per_lane / (ms * 1e-3) / 1e12);
  CK(hipFree(out));
}

int main() {
  float* src; size_t n = (size_t)256 * 16 * 16384 + 4096;
  CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 0, n * 4));
  for (int shift : {0, 1, 3}) {
    run<0>("4x dword 2x2 taps", src, shift, 4, 4);
    run<1>("2x dwordx2 pair taps", src, shift, 2, 8);
  }
  run<2>("4x dwordx4 aligned", src, 0, 4, 16);
  run<3>("4x dword coalesced", src, 0, 4, 4);
  return 0;
}
