// dev microbenchmark 2: LDS integer atomics with duplicate addresses inside one wave instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int T = 512;
constexpr int N = 8192;
constexpr int ITERS = 256;

// MODE 0: int add no return, address = base + f(lane) where f compresses (num/den) -> duplicates
// MODE 1: same with plain ds_write (reference)
// MODE 2: int add, 4 "channels" strided by 2816 like the splat image
template <int MODE>
__global__ __launch_bounds__(T) void k(int* out, int num, int den) {
  __shared__ int lds[4 * 2816];
  for (int i = threadIdx.x; i < 4 * 2816; i += T) lds[i] = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int a = wave * 300 + (lane * num) / den;
  for (int it = 0; it < ITERS; ++it) {
    int addr = a + (it & 31);
    if (MODE == 0) atomicAdd(&lds[addr], it);
    else if (MODE == 1) lds[addr] = it;
    else if (MODE == 2) {
      atomicAdd(&lds[addr], it);
      atomicAdd(&lds[addr + 2816], it);
      atomicAdd(&lds[addr + 2 * 2816], it);
      atomicAdd(&lds[addr + 3 * 2816], it);
    }
  }
  __syncthreads();
  int s = 0;
  for (int i = threadIdx.x; i < 4 * 2816; i += T) s += lds[i];
  out[blockIdx.x * T + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int num, int den, int per_iter) {
  int* out; CK(hipMalloc(&out, 4096 * T * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * 6;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, out, num, den);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 0, 0, out, num, den);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double waveinstr = (double)blocks * (T / 64) * ITERS * per_iter;
  printf("%-28s map lane*%d/%d: %8.3f ms  %7.1f cycles/wave-instr/CU\n", name, num, den, ms,
         ms * 1e-3 * 2.4e9 / (waveinstr / 256));
  CK(hipFree(out));
}

int main() {
  int maps[][2] = {{1, 1}, {15, 16}, {7, 8}, {3, 4}, {1, 2}, {1, 4}, {1, 64}, {2, 1}, {5, 4}};
  for (auto& m : maps) {
    run<0>("ds_add_u32", m[0], m[1], 1);
    run<1>("ds_write_b32", m[0], m[1], 1);
  }
  run<2>("ds_add_u32 x4 planes", 1, 1, 4);
  run<2>("ds_add_u32 x4 planes", 15, 16, 4);
  return 0;
}
