// dev microbenchmark: issue rate of scalar vs packed fp32 VALU instructions on gfx950, by waves per SIMD
// (hipcc --offload-arch=gfx950 -O3 tools_dev/r3_valu.hip -o tools_dev/r3_valu)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x[8];
  f2 y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = (f2){x[i], x[i] + 1.0f}; }
  const f2 a2 = {a, a}, b2 = {b, b};
  const unsigned long long smask = __ballot(a > 0.5f);
  unsigned long long sm2 = 0;
  float a3 = a;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
      else if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
      else if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
      else if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
      else if (MODE == 6) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(y[i]) : "v"(a2));
      else if (MODE == 7) asm volatile("v_cvt_rpi_i32_f32 %0, %0" : "+v"(x[i]));
      else if (MODE == 8) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      else if (MODE == 10) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
      else if (MODE == 11) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      else if (MODE == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(smask));
      else if (MODE == 13) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
      else if (MODE == 14) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(x[i]), "v"(a) : "vcc");
      else if (MODE == 15) asm volatile("v_cmp_lt_f32_e64 %1, %2, %0" : "+v"(x[i]), "=s"(sm2) : "v"(a));
      else if (MODE == 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 17) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 19) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      else if (MODE == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_mul_f32 %2, %2, %1" : "+v"(x[i]), "+v"(a3) : "v"(a), "v"(a));
      else if (MODE == 21) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 22) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      else if (MODE == 23) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + y[i][0] + y[i][1];
  s += (float)sm2 + a3;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 4}) {
    const int blocks = 256 * wps;  // 256-thread blocks: one wave per SIMD each
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double instr_per_simd = (double)wps * ITERS * 8;
    printf("%-18s %d waves/SIMD: %7.3f ms  %5.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, wps, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
  }
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  run<0>("v_fma_f32", out);
  run<1>("v_pk_fma_f32", out);
  run<2>("v_mul_f32", out);
  run<3>("v_pk_mul_f32", out);
  run<4>("v_pk_add_f32", out);
  run<5>("v_cndmask_b32", out);
  run<6>("v_pk_mov_b32", out);
  run<7>("v_cvt_rpi_i32_f32", out);
  run<8>("v_mul_i32_i24", out);
  run<9>("v_mad_u32_u24", out);
  run<10>("v_floor_f32", out);
  run<11>("v_med3_f32", out);
  run<12>("v_cndmask e64 sgpr", out);
  run<13>("v_cmp + v_cndmask", out);
  run<14>("v_cmp_lt_f32 vcc", out);
  run<15>("v_cmp_lt_f32 e64", out);
  run<16>("v_add_u32", out);
  run<17>("v_lshl_add_u32", out);
  run<18>("v_max_f32", out);
  run<19>("v_bfi_b32", out);
  run<20>("cndmask + v_mul", out);
  run<21>("v_mov_b32", out);
  run<22>("v_sub_f32", out);
  run<23>("v_cvt_i32_f32", out);
  return 0;
}
