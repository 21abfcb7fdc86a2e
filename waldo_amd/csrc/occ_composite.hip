// A6: occlusion product  out_j = alpha_j * prod_i (1 - alpha_i * occ[i,j])  over L layers per
// pixel -- the (1 - alpha * occ).prod(dim) * alpha pattern of models/nets/lvd.py:651-652, 686,
// 764-765, 809 (spec: LVD.reduce_comp, lvd.py:109-111).  The reference materialises an
// (L, L, h, w) tensor per map; here every thread keeps its L alphas in registers, occ is
// wave-uniform (scalar cache), and nothing of size L*L*h*w exists.
#include "waldo_common.hip.h"

namespace waldo {

template <int LP>
__global__ __launch_bounds__(kBlock) void occ_composite_fwd_kernel(
    const float* __restrict__ alpha, const float* __restrict__ occ, float* __restrict__ out,
    int L, int64_t HW, int tiles, int64_t occ_div) {
  const int64_t m = blockIdx.x / tiles;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float* ap = alpha + m * L * HW + p;
  const float* oc = occ + (m / occ_div) * L * L;
  float a[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) a[l] = (l < L) ? ap[(int64_t)l * HW] : 0.0f;
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    if (j < L) {
      float pr = 1.0f;
#pragma unroll
      for (int i = 0; i < LP; ++i)
        if (i < L) pr *= (1.0f - a[i] * oc[i * L + j]);
      out[(m * L + j) * HW + p] = a[j] * pr;
    }
  }
}

// Backward.  grad_alpha is per pixel; grad_occ[m_occ][i][j] is a sum over ALL pixels of all maps that
// share the matrix.  A workgroup walks `tiles_per_block` pixel tiles of one map: per tile and column
// j every wave reduces its 64 pixels (wave_transpose_reduce: lane bitrev(i) ends up with row i) and
// adds the result to ITS OWN row of an LDS table -- no atomics, no barrier; at the end the four rows
// are summed in a fixed order and leave the workgroup as ONE float atomic per matrix entry (the
// first version issued one per wave, tile and entry: thousands of waves on a few hundred addresses,
// 0.37 ms for 22 MB at the LVD recipe -- the pattern measured 14x below the streaming atomic rate).
template <int LP>
__global__ __launch_bounds__(kBlock) void occ_composite_bwd_kernel(
    const float* __restrict__ alpha, const float* __restrict__ occ,
    const float* __restrict__ grad_out, float* __restrict__ grad_alpha,
    float* __restrict__ grad_occ, int L, int64_t HW, int tiles, int tiles_per_block, int groups,
    int64_t occ_div) {
  const int64_t m = blockIdx.x / groups;
  const int t0 = (int)(blockIdx.x % groups) * tiles_per_block;
  const int t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const float* oc = occ + (m / occ_div) * L * L;
  __shared__ float acc[4][LP * LP];
  if (grad_occ != nullptr)
    for (int e = lane; e < LP * LP; e += kWave) acc[wave][e] = 0.0f;  // wave-private row
  for (int tile = t0; tile < t1; ++tile) {
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HW;
    const int64_t pc = live ? p : HW - 1;
    const float* ap = alpha + m * L * HW + pc;
    float a[LP], ga[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      a[l] = (l < L) ? ap[(int64_t)l * HW] : 0.0f;
      ga[l] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      if (j < L) {
        float tf[LP], ex[LP];
        float pre = 1.0f;
#pragma unroll
        for (int i = 0; i < LP; ++i) {
          tf[i] = (i < L) ? (1.0f - a[i] * oc[i * L + j]) : 1.0f;
          ex[i] = pre;
          pre *= tf[i];
        }
        float suf = 1.0f;
#pragma unroll
        for (int i = LP - 1; i >= 0; --i) {
          ex[i] *= suf;
          suf *= tf[i];
        }
        const float go = live ? grad_out[(m * L + j) * HW + pc] : 0.0f;
        ga[j] = fmaf(go, pre, ga[j]);
        const float gaj = go * a[j];
        float gocc[LP];
#pragma unroll
        for (int i = 0; i < LP; ++i) {
          if (i < L) {
            ga[i] = fmaf(-gaj * oc[i * L + j], ex[i], ga[i]);
            gocc[i] = -gaj * a[i] * ex[i];
          } else {
            gocc[i] = 0.0f;
          }
        }
        if (grad_occ != nullptr) {
          const float red = wave_transpose_reduce<LP>(gocc, lane);
          const int i = bitrev6(lane);
          if (i < L) acc[wave][i * LP + j] += red;  // one lane per entry: plain read-modify-write
        }
      }
    }
    if (live) {
#pragma unroll
      for (int l = 0; l < LP; ++l)
        if (l < L) grad_alpha[(m * L + l) * HW + p] = ga[l];
    }
  }
  if (grad_occ != nullptr) {
    __syncthreads();
    for (int e = threadIdx.x; e < LP * LP; e += kBlock) {
      const int i = e / LP, j = e % LP;
      if (i < L && j < L)
        atomicAdd(grad_occ + (m / occ_div) * L * L + i * L + j, (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]));
    }
  }
}

static int pad_l(int L) {
  if (L <= 4) return 4;
  if (L <= 8) return 8;
  if (L <= 12) return 12;
  if (L <= 17) return 17;
  if (L <= 24) return 24;
  return 32;
}

static int check_occ(const char* fn, int64_t M, int L, int64_t HW, int64_t occ_div) {
  if (M < 0 || L < 1 || L > 32 || HW < 1 || occ_div < 1) {
    set_error("%s: bad shape M=%lld L=%d HW=%lld occ_div=%lld (need 1<=L<=32)", fn, (long long)M,
              L, (long long)HW, (long long)occ_div);
    return WALDO_EINVAL;
  }
  if (M * ((HW + kBlock - 1) / kBlock) > 2147483647) {
    set_error("%s: problem too large for one launch", fn);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

#define WALDO_OCC_CASE(LPV, KERNEL, ...)                                                    \
  case LPV:                                                                                 \
    hipLaunchKernelGGL((KERNEL<LPV>), dim3((unsigned)(M * tiles)), dim3(kBlock), 0, st,     \
                       __VA_ARGS__);                                                        \
    break;

extern "C" int waldo_occ_composite_fwd(const float* alpha, const float* occ, float* out,
                                       int64_t M, int L, int64_t HW, int64_t occ_div,
                                       waldo_stream_t stream) {
  int rc = check_occ("waldo_occ_composite_fwd", M, L, HW, occ_div);
  if (rc) return rc;
  if (M == 0) return WALDO_OK;
  if (!alpha || !occ || !out) {
    set_error("waldo_occ_composite_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  switch (pad_l(L)) {
    WALDO_OCC_CASE(4, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
    WALDO_OCC_CASE(8, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
    WALDO_OCC_CASE(12, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
    WALDO_OCC_CASE(17, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
    WALDO_OCC_CASE(24, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
    WALDO_OCC_CASE(32, occ_composite_fwd_kernel, alpha, occ, out, L, HW, tiles, occ_div)
  }
  return launch_status("waldo_occ_composite_fwd");
}

extern "C" int waldo_occ_composite_bwd(const float* alpha, const float* occ,
                                       const float* grad_out, float* grad_alpha, float* grad_occ,
                                       int64_t M, int L, int64_t HW, int64_t occ_div,
                                       waldo_stream_t stream) {
  int rc = check_occ("waldo_occ_composite_bwd", M, L, HW, occ_div);
  if (rc) return rc;
  if (M == 0) return WALDO_OK;
  if (!alpha || !occ || !grad_out || !grad_alpha) {
    set_error("waldo_occ_composite_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  // several pixel tiles per workgroup (fewer atomics per matrix entry) while keeping >= ~1024
  // workgroups in flight
  int tpb = (int)min((int64_t)16, max((int64_t)1, (M * tiles) / 1024));
  if (grad_occ == nullptr) tpb = 1;
  const int groups = (tiles + tpb - 1) / tpb;
#define WALDO_OCC_BWD(LPV)                                                                        \
  case LPV:                                                                                       \
    hipLaunchKernelGGL((occ_composite_bwd_kernel<LPV>), dim3((unsigned)(M * groups)), dim3(kBlock), 0, st, \
                       alpha, occ, grad_out, grad_alpha, grad_occ, L, HW, tiles, tpb, groups, occ_div); \
    break;
  switch (pad_l(L)) {
    WALDO_OCC_BWD(4)
    WALDO_OCC_BWD(8)
    WALDO_OCC_BWD(12)
    WALDO_OCC_BWD(17)
    WALDO_OCC_BWD(24)
    WALDO_OCC_BWD(32)
  }
#undef WALDO_OCC_BWD
  return launch_status("waldo_occ_composite_bwd");
}
