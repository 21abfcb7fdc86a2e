// dev microbenchmarks of round 3 (hipcc --offload-arch=gfx950 -O3 tools_dev/r3_micro.hip -o tools_dev/r3_micro):
//  A  vector-memory gathers in the kernels' wave shape (4 pixel rows x 16 columns): 2x2 bilinear taps of four
//     channel planes as 4 dword loads per plane vs 2 unaligned dwordx2 loads per plane (L2-resident data);
//  B  LDS integer atomics: ds_add_u32 x 4 planes vs ds_add_u64 x 2 plane pairs (the splat's 16 adds per pixel);
//  C  ds_read_b128 taps of a float4-texel image in that wave shape for image pitches 20 / 24 / 28 / 32 texels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

static float time_ms(void (*launch)(void*), void* arg, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(arg); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch(arg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

// ---------------------------------------------------------------- A
constexpr int kRows = 64, kPitchA = 512, kPlane = kRows * kPitchA;  // one plane of a window: 128 KB
constexpr int kItersA = 64;
struct ArgA { const float* src; float* out; int sxq; };  // sxq: x advance per pixel column in 1/16 texel

template <int MODE>
__global__ __launch_bounds__(256) void ka(const float* __restrict__ src, float* out, int sxq) {
  // 16 windows of four planes (8 MB: L2 / MALL resident); a block = a 16 x 16-pixel tile
  const float* base = src + (size_t)(blockIdx.x & 15) * 4 * kPlane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = wave * 4 + (lane >> 4), col = lane & 15;
  float acc = 0.f;
  for (int it = 0; it < kItersA; ++it) {
    const int x0 = ((it * 37) & 255) + ((col * sxq) >> 4) + (blockIdx.x & 3);
    const int y0 = ((it * 11) & 31) + row;
    const float* p = base + y0 * kPitchA + x0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* q = p + c * kPlane;
      if (MODE == 0) {
        acc += q[0] + q[1] + q[kPitchA] + q[kPitchA + 1];
      } else {
        f2 a, b;
        __builtin_memcpy(&a, q, 8);
        __builtin_memcpy(&b, q + kPitchA, 8);
        acc += a.x + a.y + b.x + b.y;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> static void la(void* a) {
  ArgA* g = (ArgA*)a;
  hipLaunchKernelGGL(ka<MODE>, dim3(256 * 16), dim3(256), 0, 0, g->src, g->out, g->sxq);
}

// ---------------------------------------------------------------- B
constexpr int kItersB = 128;
template <int MODE>
__global__ __launch_bounds__(512, 2) void kb(int* out, int pitch) {
  __shared__ __attribute__((aligned(16))) int lds[4 * (32 * 80 + 64)];
  const int plane = 32 * pitch + 64;
  for (int i = threadIdx.x; i < 4 * (32 * 80 + 64); i += 512) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = (lane >> 4), col = lane & 15;
  for (int it = 0; it < kItersB; ++it) {
    const int y = ((it * 5 + wave) & 15) + row, x = ((it * 7) & 31) + col;
    const int a = y * pitch + x;
    const int v = it + lane;
    if (MODE == 0) {  // 16 x ds_add_u32: four corners x four planes
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        atomicAdd(&lds[c * plane + a], v);
        atomicAdd(&lds[c * plane + a + 1], v + 1);
        atomicAdd(&lds[c * plane + a + pitch], v + 2);
        atomicAdd(&lds[c * plane + a + pitch + 1], v + 3);
      }
    } else {  // 8 x ds_add_u64: four corners x two plane pairs (64-bit words, same texel indexing)
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
      const unsigned long long w = ((unsigned long long)(unsigned)v << 32) | (unsigned)(v + 7);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        atomicAdd(&l64[c * plane + a], w);
        atomicAdd(&l64[c * plane + a + 1], w + 1);
        atomicAdd(&l64[c * plane + a + pitch], w + 2);
        atomicAdd(&l64[c * plane + a + pitch + 1], w + 3);
      }
    }
  }
  __syncthreads();
  int s = 0;
  for (int i = threadIdx.x; i < 4 * (32 * 80 + 64); i += 512) s += lds[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
struct ArgB { int* out; int pitch; };
template <int MODE> static void lb(void* a) {
  ArgB* g = (ArgB*)a;
  hipLaunchKernelGGL(kb<MODE>, dim3(256 * 8), dim3(512), 0, 0, g->out, g->pitch);
}

// ---------------------------------------------------------------- C
constexpr int kItersC = 256;
__global__ __launch_bounds__(256) void kc(float* out, int pitch, int sxq, int syq, int one) {
  __shared__ __attribute__((aligned(16))) float img[4 * 32 * 40];
  for (int i = threadIdx.x; i < 4 * 32 * 40; i += 256) img[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = wave * 4 + (lane >> 4), col = lane & 15;
  f4 acc = {0, 0, 0, 0};
  for (int it = 0; it < kItersC; ++it) {
    const int x = ((col * sxq) >> 4) + ((it * one) & 3), y = ((row * syq) >> 4) + (((it * one) >> 2) & 3);
    const f4* r0 = reinterpret_cast<const f4*>(img) + y * pitch + x;
    const f4* r1 = r0 + pitch;
    acc += r0[0] + r0[1] + r1[0] + r1[1];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
struct ArgC { float* out; int pitch, sxq, syq; };
static void lc(void* a) {
  ArgC* g = (ArgC*)a;
  hipLaunchKernelGGL(kc, dim3(256 * 16), dim3(256), 0, 0, g->out, g->pitch, g->sxq, g->syq, 1);
}

int main() {
  float* src; size_t n = (size_t)16 * 4 * kPlane + 8192;
  CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 0, n * 4));
  float* out; CK(hipMalloc(&out, 256 * 16 * 512 * 4));
  printf("A: gathers, 4 planes, 16 x 16-pixel tiles, %d blocks (cycles at 2.4 GHz)\n", 256 * 16);
  for (int sxq : {16, 13, 20}) {
    ArgA a{src, out, sxq};
    const double pl = (double)256 * 16 * 4 * kItersA / 256;  // (wave, layer) items per CU
    float t0 = time_ms(la<0>, &a, 5), t1 = time_ms(la<1>, &a, 5);
    printf("  x step %4.2f: 16 x dword   %7.3f ms = %6.1f cyc per wave-layer per CU (%5.1f B/clk/CU of taps)\n", sxq / 16.0, t0,
           t0 * 1e-3 * 2.4e9 / pl, 64 * 64.0 / (t0 * 1e-3 * 2.4e9 / pl));
    printf("  x step %4.2f:  8 x dwordx2 %7.3f ms = %6.1f cyc per wave-layer per CU (%5.1f B/clk/CU of taps)\n", sxq / 16.0, t1,
           t1 * 1e-3 * 2.4e9 / pl, 64 * 64.0 / (t1 * 1e-3 * 2.4e9 / pl));
  }
  printf("B: LDS integer atomics, 4 rows x 16 columns per wave, 512 threads, 2048 blocks\n");
  for (int pitch : {64, 80}) {
    ArgB b{(int*)out, pitch};
    const double items = (double)2048 * 8 * kItersB / 256;  // (wave, pixel) items per CU
    float t0 = time_ms(lb<0>, &b, 5), t1 = time_ms(lb<1>, &b, 5);
    printf("  pitch %d: 16 x ds_add_u32 %7.3f ms = %6.1f cyc per item per CU;  8 x ds_add_u64 %7.3f ms = %6.1f\n", pitch, t0,
           t0 * 1e-3 * 2.4e9 / items, t1, t1 * 1e-3 * 2.4e9 / items);
  }
  printf("C: 4 x ds_read_b128 taps of a float4-texel image, 4 rows x 16 columns per wave\n");
  for (int sxq : {16, 13, 20})
    for (int syq : {16, 13, 20})
      for (int pitch : {20, 24, 28, 32}) {
        ArgC c{out, pitch, sxq, syq};
        const double items = (double)256 * 16 * 4 * kItersC / 256;
        float t = time_ms(lc, &c, 5);
        printf("  x step %4.2f y step %4.2f pitch %2d: %7.3f ms = %5.1f cyc per 4-tap item per CU\n", sxq / 16.0, syq / 16.0, pitch, t,
               t * 1e-3 * 2.4e9 / items);
      }
  return 0;
}
