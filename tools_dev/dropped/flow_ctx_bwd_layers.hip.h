// Backward of flow_ctx_warp (models/nets/lvd.py:784-818) for SMALL launches, with the per-layer work of a pixel spread
// over the grid instead of walked by one lane (round 5, third attempt at the LVD-recipe step's largest kernel).
//
// What the counters of the one-kernel form say (flow_ctx_bwd.hip:flow_ctx_warp_bwd_kernel<17> at the LVD recipe, B = 2:
// profiles/r05_lvd_step_counters.txt): 2560 wavefronts in the whole launch = 2.5 per SIMD, ALL resident at once, each
// a dependent chain of 16.7 k vector instructions with two gather round trips per layer in it; VALU issue 31 %.  The
// launch takes as long as ONE wavefront takes to walk its two tiles: there is nothing to hide a stall behind, and a
// faster composite (flow_ctx_bwd_rows.hip.h) or more loads in flight per lane do not change that.  The work is not
// small -- 10 units x 17 layers x 32768 pixels -- it is folded 17-fold into the lanes.
//
// Here the sampling and the scatter run one lane per (unit, LAYER, pixel), 87 k wavefronts of a few dozen
// instructions, and only the L x L composite keeps a lane per pixel:
//   fcw_bwd_sample     per (m, l, pixel): flow_l, taps, the ghosted sample a_l with its derivatives, and
//                      gv_l = d loss / d v_l from grad_flow and grad_alpha_ctx           -> ws: a, dx, dy, gv
//   fcw_bwd_composite  per (m, pixel): disocclusion arg max, composite_bwd (grad_occ as before), the composited
//                      v_l                                                   ws: gv <- d loss / d a_l, a <- v_l
//   fcw_bwd_scatter    per (m, l, pixel): d / d flow_l (stored), d / d a01 (bilinear splat, float atomics)
// Same formulas and the same order of operations per value as the one-kernel form (its phases, cut at the points
// where it recomputes instead of keeping); the atomics of the splat and of grad_occ land in another order.
// Traffic: four workspace planes written and read per (unit, layer) instead of a second pass over the flow planes and
// the gathers -- about the same bytes (0.44 GB per call at the LVD recipe), all of it short independent wavefronts.
#pragma once
// included by flow_ctx_bwd.hip behind composite_bwd

namespace waldo {

struct FcwUnit {
  int m, l, b;
  int64_t p;
  bool live;
};

// (unit m, layer l, 256-pixel strip) of this workgroup; the strip index runs fastest: the strips of one plane follow
// each other through the dispatcher
__device__ __forceinline__ FcwUnit fcw_unit(int L, int tiles, int TcTp, int64_t HWd) {
  FcwUnit u;
  const int ml = blockIdx.x / tiles;
  u.m = ml / L;
  u.l = ml - u.m * L;
  u.b = u.m / TcTp;
  u.p = (int64_t)(blockIdx.x - ml * tiles) * kBlock + threadIdx.x;
  u.live = u.p < HWd;
  return u;
}

__global__ __launch_bounds__(kBlock) void fcw_bwd_sample_kernel(
    const float* __restrict__ flow_lr, const float* __restrict__ isobj_lr, const float* __restrict__ a01,
    const int64_t* __restrict__ ctx_ts, const float* __restrict__ g_flow, const float* __restrict__ g_actx,
    float* __restrict__ ws_a, float* __restrict__ ws_dx, float* __restrict__ ws_dy, float* __restrict__ ws_gv, int Tw,
    int TcTp, int L, int H, int W, int scale, int tiles) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const FcwUnit u = fcw_unit(L, tiles, TcTp, HWd);
  if (!u.live) return;
  const int m = u.m, l = u.l;
  const int64_t p = u.p;
  const int ts = __builtin_amdgcn_readfirstlane((int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(Tw - 1)));
  const int y = (int)(p / Wd), x = (int)(p - (int64_t)y * Wd);
  const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
  float gx0, gy0;
  identity_grid(x, y, Wd, Hd, gx0, gy0);
  const float* fl = flow_lr + (((int64_t)m * L + l) * 2) * HW;
  const float fxl = up_sample(fl, ut), fyl = up_sample(fl + HW, ut);
  const float gfx = g_flow != nullptr ? g_flow[((int64_t)m * 2) * HWd + p] : 0.0f;
  const float gfy = g_flow != nullptr ? g_flow[((int64_t)m * 2 + 1) * HWd + p] : 0.0f;
  const int64_t at = ((int64_t)m * L + l) * HWd + p;
  const float gac = g_actx != nullptr ? g_actx[at] : 0.0f;
  const Taps t = make_taps(gx0 + fxl, gy0 + fyl, Hd, Wd);
  float ddx, ddy;
  const float v = tap_sample_d(a01 + (((int64_t)u.b * Tw + ts) * L + l) * HWd, t, ddx, ddy);
  float ghost = 1.0f;
  if (isobj_lr != nullptr && l >= 1)
    ghost = (up_sample(isobj_lr + ((int64_t)m * (L - 1) + (l - 1)) * HW, ut) > 0.9f) ? 1.0f : 0.0f;
  float gv = fmaf(gfx, fxl, gfy * fyl);
  if (g_actx != nullptr) gv = fmaf(2.0f, gac, gv);
  ws_a[at] = v * ghost;
  ws_dx[at] = ddx * ghost;
  ws_dy[at] = ddy * ghost;
  ws_gv[at] = gv;
}

#ifndef WALDO_FCL_COMP_WAVES
#define WALDO_FCL_COMP_WAVES 3
#endif
template <int LP>
__global__ __launch_bounds__(kBlock, WALDO_FCL_COMP_WAVES) void fcw_bwd_composite_kernel(
    float* __restrict__ ws_a, float* __restrict__ ws_gv, const int64_t* __restrict__ pred_ts,
    const float* __restrict__ occ, const float* __restrict__ g_dis, float* __restrict__ g_occ, int T, int Tc, int Tp,
    int L, int64_t HWd, int tiles, int tiles_per_block, int groups) {
  const int m = blockIdx.x / groups;  // (b, tc, tp)
  const int tp = m % Tp, b = m / (Tc * Tp);
  const int t0 = (blockIdx.x % groups) * tiles_per_block, t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __shared__ float acc_o[4][LP * LP];
  __shared__ __attribute__((aligned(16))) float occm[OccLds<LP>::kFloats];
  for (int e = lane; e < LP * LP; e += kWave) acc_o[wave][e] = 0.0f;
  const int tpred = __builtin_amdgcn_readfirstlane((int)min(max(pred_ts[tp], (int64_t)0), (int64_t)(T - 1)));
  occ_stage<LP>(occm, occ + ((int64_t)b * T + tpred) * L * L, L);
  __syncthreads();

  for (int tile = t0; tile < t1; ++tile) {
    int fresh = 0;  // (keeps the reads of the order inside the loop: flow_ctx_alpha_bwd_kernel)
    asm volatile("" : "+v"(fresh));
    const float* occm_t = occm + fresh;
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HWd;
    const int64_t pc = live ? p : HWd - 1;
    float a[LP], gv[LP], ga[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int64_t at = ((int64_t)m * L + min(l, L - 1)) * HWd + pc;
      const float al = ws_a[at], gl = ws_gv[at];
      a[l] = l < L ? al : 0.0f;
      gv[l] = (l < L && live) ? gl : 0.0f;
    }
    float dis = -INFINITY;
    int amax = 0;
#pragma unroll
    for (int l = 0; l < LP; ++l)
      if (l < L && a[l] > dis) {
        dis = a[l];
        amax = l;
      }
    composite_bwd<LP>(a, gv, occm_t, L, ga, g_occ != nullptr ? acc_o[wave] : nullptr, lane);
    const float gd = (g_dis != nullptr && live) ? g_dis[(int64_t)m * HWd + pc] : 0.0f;
    typedef float f32x2_w __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < LP; j += 4) {
      // v_j (the composited alphas) of four layers, as the forward kernel takes them (same bits)
      f32x2_w prd[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const f32x4_o o = occ_quad<LP, false>(occm_t, i, j);
        const f32x2_w ai = {a[i], a[i]}, one = {1.0f, 1.0f};
        prd[0] = prd[0] * __builtin_elementwise_fma(-ai, (f32x2_w){o[0], o[1]}, one);
        if (j + 2 < LP) prd[1] = prd[1] * __builtin_elementwise_fma(-ai, (f32x2_w){o[2], o[3]}, one);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int l = j + k < LP ? j + k : LP - 1;
        if (j + k < LP && l < L && live) {
          const int64_t at = ((int64_t)m * L + l) * HWd + p;
          ws_a[at] = a[l] * prd[k >> 1][k & 1];              // v_l
          ws_gv[at] = ga[l] + (l == amax ? gd : 0.0f);       // d loss / d a_l
        }
      }
    }
  }
  __syncthreads();
  if (g_occ != nullptr)
    for (int e = threadIdx.x; e < LP * LP; e += kBlock) {
      const int i = e / LP, j = e % LP;
      if (i < L && j < L)
        atomicAdd(g_occ + ((int64_t)b * T + tpred) * L * L + i * L + j,
                  (acc_o[0][e] + acc_o[1][e]) + (acc_o[2][e] + acc_o[3][e]));
    }
}

__global__ __launch_bounds__(kBlock) void fcw_bwd_scatter_kernel(
    const float* __restrict__ flow_lr, const float* __restrict__ isobj_lr, const int64_t* __restrict__ ctx_ts,
    const float* __restrict__ g_flow, const float* __restrict__ ws_v, const float* __restrict__ ws_dx,
    const float* __restrict__ ws_dy, const float* __restrict__ ws_gs, float* __restrict__ g_fup,
    float* __restrict__ g_a01, int Tw, int TcTp, int L, int H, int W, int scale, int tiles) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const FcwUnit u = fcw_unit(L, tiles, TcTp, HWd);
  if (!u.live) return;
  const int m = u.m, l = u.l;
  const int64_t p = u.p;
  const int64_t at = ((int64_t)m * L + l) * HWd + p;
  const float gs = ws_gs[at], vjl = ws_v[at], dxg = ws_dx[at], dyg = ws_dy[at];
  const float gfx = g_flow != nullptr ? g_flow[((int64_t)m * 2) * HWd + p] : 0.0f;
  const float gfy = g_flow != nullptr ? g_flow[((int64_t)m * 2 + 1) * HWd + p] : 0.0f;
  const float hw = 0.5f * (float)Wd, hh = 0.5f * (float)Hd;
  // d / d flow_l: through the flow composite and through the sample position
  g_fup[(((int64_t)m * L + l) * 2) * HWd + p] = fmaf(gs * dxg, hw, vjl * gfx);
  g_fup[(((int64_t)m * L + l) * 2 + 1) * HWd + p] = fmaf(gs * dyg, hh, vjl * gfy);
  if (g_a01 == nullptr) return;
  // d / d a01: bilinear splat of gs * ghost at the sample position (taken again from the flow)
  const int y = (int)(p / Wd), x = (int)(p - (int64_t)y * Wd);
  const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
  float ghost = 1.0f;
  if (isobj_lr != nullptr && l >= 1)
    ghost = (up_sample(isobj_lr + ((int64_t)m * (L - 1) + (l - 1)) * HW, ut) > 0.9f) ? 1.0f : 0.0f;
  const float gsg = gs * ghost;
  if (gsg == 0.0f) return;
  float gx0, gy0;
  identity_grid(x, y, Wd, Hd, gx0, gy0);
  const float* fl = flow_lr + (((int64_t)m * L + l) * 2) * HW;
  const float fxl = up_sample(fl, ut), fyl = up_sample(fl + HW, ut);
  const Taps t = make_taps(gx0 + fxl, gy0 + fyl, Hd, Wd);
  const int ts = __builtin_amdgcn_readfirstlane((int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(Tw - 1)));
  float* gp = g_a01 + (((int64_t)u.b * Tw + ts) * L + l) * HWd;
  // the lerp form's weights: (1-fx)(1-fy) v00 ... with the validity of each corner
  const float wx0 = 1.0f - t.fx, wy0 = 1.0f - t.fy;
  const float w00 = wx0 * wy0 * (t.vx0 * t.vy0), w01 = t.fx * wy0 * (t.vx1 * t.vy0);
  const float w10 = wx0 * t.fy * (t.vx0 * t.vy1), w11 = t.fx * t.fy * (t.vx1 * t.vy1);
  if (w00 != 0.0f) atomicAdd(gp + (t.o00 >> 2), gsg * w00);
  if (w01 != 0.0f) atomicAdd(gp + (t.o01 >> 2), gsg * w01);
  if (w10 != 0.0f) atomicAdd(gp + (t.o10 >> 2), gsg * w10);
  if (w11 != 0.0f) atomicAdd(gp + (t.o11 >> 2), gsg * w11);
}

// floats of workspace the layer-parallel form wants BEHIND the (M, L, 2, Hd, Wd) upsampling workspace of scale > 1
inline int64_t fcw_layers_floats(int64_t M, int L, int H, int W, int scale) {
  return 4 * M * L * (int64_t)H * scale * W * scale;
}
// small launches only: from here on the one-kernel form has wavefronts enough to hide its chains behind each
// other and moves fewer bytes
inline bool fcw_layers_pays(int64_t M, int tiles) { return M * tiles <= 8192; }

}  // namespace waldo
