// dev microbenchmark: LDS scatter-add primitives on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int T = 512;
constexpr int N = 4096;   // floats in LDS
constexpr int ITERS = 256;

template <int MODE>
__global__ __launch_bounds__(T) void k(float* out, int stride, int pred) {
  __shared__ float lds[N];
  for (int i = threadIdx.x; i < N; i += T) lds[i] = 0.f;
  __syncthreads();
  int a = (threadIdx.x * stride) % (N - 64);
  float v = 1.0f + threadIdx.x * 1e-3f;
  unsigned acc = 0;
  for (int it = 0; it < ITERS; ++it) {
    int addr = a + (it & 31);
    if (MODE == 0) {  // ds_add_f32 no return
      atomicAdd(&lds[addr], v);
    } else if (MODE == 1) {  // plain write
      lds[addr] = v + it;
    } else if (MODE == 2) {  // read-modify-write, no atomic
      lds[addr] = lds[addr] + v;
    } else if (MODE == 3) {  // returning int atomic add
      acc += atomicAdd(reinterpret_cast<int*>(&lds[addr]), 1);
    } else if (MODE == 4) {  // exchange
      acc += atomicExch(reinterpret_cast<int*>(&lds[addr]), it);
    } else if (MODE == 5) {  // predicated ds_add_f32 (half lanes)
      if ((threadIdx.x + it) & pred) atomicAdd(&lds[addr], v);
    } else if (MODE == 6) {  // int atomic add no return
      atomicAdd(reinterpret_cast<int*>(&lds[addr]), 3);
    } else if (MODE == 7) {  // __hip_atomic relaxed workgroup scope fp add
      __hip_atomic_fetch_add(&lds[addr], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < N; i += T) s += lds[i];
  out[blockIdx.x * T + threadIdx.x] = s + acc;
}

template <int MODE>
void run(const char* name, int stride, int pred = 1) {
  float* out; CK(hipMalloc(&out, 4096 * T * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * 8;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, out, stride, pred);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, out, stride, pred);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double waveinstr = (double)blocks * (T / 64) * ITERS;
  double cyc_per_cu = ms * 1e-3 * 2.4e9;       // cycles available per CU
  double instr_per_cu = waveinstr / 256;
  printf("%-34s stride %2d: %8.3f ms  %6.1f cycles/wave-instr/CU  %7.1f Glane/s\n", name, stride, ms,
         cyc_per_cu / instr_per_cu, waveinstr * 64 / (ms * 1e-3) / 1e9);
  CK(hipFree(out));
}

int main() {
  for (int stride : {1, 2}) {
    run<1>("ds_write_b32", stride);
    run<2>("read+add+write", stride);
    run<0>("atomicAdd float (ds_add_f32)", stride);
    run<7>("hip_atomic relaxed wg float", stride);
    run<6>("atomicAdd int noret", stride);
    run<3>("atomicAdd int rtn", stride);
    run<4>("atomicExch rtn", stride);
    run<5>("predicated float add (pred=1)", stride, 1);
  }
  return 0;
}
