// Backward of the fused full-resolution passes of Warper.grid_to_flow[_ctx] and
// Warper.input_to_output (csrc/flow_ctx.hip; models/nets/lvd.py:602-853) -- the reference's LIVE
// backward path in LVD training (models/synthesizer.py:841: ctx_mode "prev", include_self), where
// autograd differentiates ~25 elementwise / interpolate / grid_sample / softmax / prod launches per
// method.  One kernel per forward kernel, one thread per HD pixel, the forward values recomputed in
// registers:
//
//   flow_ctx_alpha_bwd   d / d (rough alphas, class distributions, occlusion matrix)
//   flow_ctx_warp_bwd    d / d (per-layer flow, composited alpha of the context frames, occlusion
//                        matrix); the gradient of the sampled context alpha is a bilinear splat:
//                        float atomics into a zero-filled buffer (as F.grid_sample's backward)
//   frame_warp_fuse_bwd  d / d (composited flow, context alphas); the frames are data
//
// Gradients w.r.t. the LOW-resolution inputs (alpha_lr, flow_lr) are the transpose of the xS
// bilinear upsampling: the pixel kernels write the gradient at the HD raster and a gather kernel
// sums, for every LR texel, the <= (2S)^2 HD pixels that interpolate from it (no atomics); at
// S == 1 the pixel kernels write the LR gradient directly.
// Sums over all pixels (grad_occ, grad_dist) follow the pattern of occ_composite_bwd: per tile a
// wave transpose-reduce into the wave's own row of an LDS table, one float atomic per workgroup and
// entry at the end.
#include "flow_ctx_common.hip.h"

namespace waldo {

constexpr int kAccTilesMax = 16;  // pixel tiles a workgroup walks before it flushes its LDS tables
#ifndef WALDO_FCB_ALPHA_WAVES
#define WALDO_FCB_ALPHA_WAVES 3
#endif
#ifndef WALDO_FCB_MIN_WGS
#define WALDO_FCB_MIN_WGS 512
#endif
// The two flow_ctx backward kernels are compiled for >= 3 waves per SIMD (168 VGPRs): at L = 17 they
// want 210 / 300+ registers and would run one or two waves per SIMD; with the cap they spill ~100 /
// ~1000 bytes per lane to scratch and the LVD-recipe step is 5 % faster (3.85 -> 3.67 ms).

__device__ __forceinline__ float sgnf(float x) { return (x > 0.0f ? 1.0f : 0.0f) - (x < 0.0f ? 1.0f : 0.0f); }

// d loss / d a_i and d loss / d occ[i][j] of  v_j = a_j prod_i (1 - a_i occ[i][j])  for one pixel;
// gv[j] = d loss / d v_j.  The occ gradients go, reduced over the wave, into the wave's LDS row.
// occm: the order in LDS (OccLds; column j is a row of the transposed copy).  A padding layer has a == 0:
// its factor is exactly 1 and its occ gradient exactly 0, no guard needed; its ga is never stored.
template <int LP>
__device__ __forceinline__ void composite_bwd(const float (&a)[LP], const float (&gv)[LP],
                                              const float* occm, int L, float (&ga)[LP],
                                              float* acc_row, int lane) {
#pragma unroll
  for (int l = 0; l < LP; ++l) ga[l] = 0.0f;
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    if (j < L) {  // wave-uniform
      float tf[LP], ex[LP];
      float pre = 1.0f;
#pragma unroll
      for (int i0 = 0; i0 < LP; i0 += 4) {
        const f32x4_o o = occ_quad<LP, true>(occm, j, i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k < LP ? i0 + k : LP - 1;
          if (i0 + k < LP) {
            tf[i] = 1.0f - a[i] * o[k];
            ex[i] = pre;
            pre *= tf[i];
          }
        }
      }
      float suf = 1.0f;
#pragma unroll
      for (int i = LP - 1; i >= 0; --i) {
        ex[i] *= suf;
        suf *= tf[i];
      }
      ga[j] = fmaf(gv[j], pre, ga[j]);
      const float gaj = gv[j] * a[j];
      float gocc[LP];
#pragma unroll
      for (int i0 = 0; i0 < LP; i0 += 4) {
        const f32x4_o o = occ_quad<LP, true>(occm, j, i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k < LP ? i0 + k : LP - 1;
          if (i0 + k < LP) {
            ga[i] = fmaf(-gaj * o[k], ex[i], ga[i]);
            gocc[i] = -gaj * a[i] * ex[i];
          }
        }
      }
      if (acc_row != nullptr) {
        const float red = wave_transpose_reduce<LP>(gocc, lane);
        const int i = bitrev6(lane);
        if (i < L) acc_row[i * LP + j] += red;  // one lane per entry
      }
    }
  }
}

// d loss / d a01 from the gradients of the two outputs of the forward, a01 and alpha_out = 2 a01 - 1: what the caller
// summed as `g_a01 + 2 * g_alpha_out` in two passes over (B*Tw, L, Hd, Wd) before the call (the same bits: the
// doubling is exact).  Either pointer may be null (uniform), not both.  Both loads are UNCONDITIONAL -- a missing
// gradient re-reads the other one's address and a select drops it: with a branch per case the L loads of a pixel
// stopped being in flight together (one round trip per layer: +45 us per call at the LVD recipe).
struct GradOfA01 {
  const float* pa;
  const float* pb;
  bool has_a, has_b;
  __device__ __forceinline__ GradOfA01(const float* g_a01, const float* g_aout)
      : pa(g_a01 != nullptr ? g_a01 : g_aout), pb(g_aout != nullptr ? g_aout : g_a01), has_a(g_a01 != nullptr),
        has_b(g_aout != nullptr) {}
  __device__ __forceinline__ float operator()(int64_t at) const {
    const float va = pa[at], vb = pb[at];
    return has_b ? (has_a ? va + 2.0f * vb : 2.0f * vb) : va;
  }
};

template <int LP, int NCP>
__global__ __launch_bounds__(kBlock, WALDO_FCB_ALPHA_WAVES) void flow_ctx_alpha_bwd_kernel(
    const float* __restrict__ alpha_lr, const float* __restrict__ input,
    const float* __restrict__ dist, const float* __restrict__ occ, const float* __restrict__ g_a01,
    const float* __restrict__ g_aout, float* __restrict__ g_up, float* __restrict__ g_dist,
    float* __restrict__ g_occ, int T, int Tw, int L, int Nl, int C, int chan_off, int H, int W, int scale,
    int tiles, int tiles_per_block, int groups) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const int n = blockIdx.x / groups;  // (b, t) with t < Tw
  const int b = n / Tw, t = n % Tw;
  const int t0 = (blockIdx.x % groups) * tiles_per_block, t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int No = L - 1;
  __shared__ __attribute__((aligned(16))) float sdist[(LP - 1) * kMaxCls];
  __shared__ __attribute__((aligned(16))) float occm[OccLds<LP>::kFloats];
  __shared__ float acc_o[4][LP * LP];
  __shared__ float acc_d[4][(LP - 1) * kMaxCls];
  const GradOfA01 grad_of_a01(g_a01, g_aout);
  const bool filt = dist != nullptr;
  if (filt) dist_stage<LP>(sdist, dist + (int64_t)b * No * Nl, L, Nl);
  occ_stage<LP>(occm, occ + ((int64_t)b * T + t) * L * L, L);
  for (int e = lane; e < LP * LP; e += kWave) acc_o[wave][e] = 0.0f;
  for (int e = lane; e < (LP - 1) * kMaxCls; e += kWave) acc_d[wave][e] = 0.0f;
  __syncthreads();

  for (int tile = t0; tile < t1; ++tile) {
    // the LDS tables do not change while the workgroup walks its tiles, and the compiler knows: it would hoist
    // all their reads out of this loop, into ~600 registers it does not have.  An offset it cannot see through:
    int fresh = 0;
    asm volatile("" : "+v"(fresh));
    const float* occm_t = occm + fresh;
    const float* sdist_t = sdist + fresh;
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HWd;
    const int64_t pc = live ? p : HWd - 1;
    const int y = (int)(pc / Wd), x = (int)(pc - (int64_t)y * Wd);
    const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
    // (the upsampled rough alpha is not kept across the composite: it is one load to have again)
    float f[LP], a[LP], gv[LP], ga[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      f[l] = 1.0f;
      gv[l] = (l < L && live) ? grad_of_a01(((int64_t)n * L + min(l, L - 1)) * HWd + pc) : 0.0f;
    }
    float pr[NCP];
    if (filt) {
      const float* lg = input + (((int64_t)b * T + t) * C + chan_off) * HWd + pc;
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < NCP; ++c) {
        pr[c] = (c < Nl) ? lg[(int64_t)min(c, Nl - 1) * HWd] : -INFINITY;
        m = fmaxf(m, pr[c]);
      }
      float den = 0.0f;
#pragma unroll
      for (int c = 0; c < NCP; ++c) {
        pr[c] = (c < Nl) ? expf(pr[c] - m) : 0.0f;
        den += pr[c];
      }
#pragma unroll
      for (int c = 0; c < NCP; ++c) pr[c] = pr[c] / den;
#pragma unroll
      for (int l = 1; l < LP; ++l) f[l] = 1.0f - dist_l1(sdist_t + (l - 1) * kMaxCls, pr, Nl) / 2.0f;
    }
#pragma unroll
    for (int l = 0; l < LP; ++l)
      a[l] = (l < L) ? up_sample(alpha_lr + ((int64_t)n * L + min(l, L - 1)) * HW, ut) * f[l] : 0.0f;
    composite_bwd<LP>(a, gv, occm_t, L, ga, g_occ != nullptr ? acc_o[wave] : nullptr, lane);
    if (live) {
#pragma unroll
      for (int l = 0; l < LP; ++l)
        if (l < L) g_up[((int64_t)n * L + l) * HWd + p] = ga[l] * f[l];
    }
    if (filt && g_dist != nullptr) {
      // d f_l / d dist[l][c] = -1/2 sign(dist[l][c] - pr_c); summed over the pixels of the wave
#pragma unroll
      for (int l = 1; l < LP; ++l) {
        if (l < L) {  // wave-uniform
          const float aup = up_sample(alpha_lr + ((int64_t)n * L + l) * HW, ut);
          const float gf = live ? -0.5f * ga[l] * aup : 0.0f;
          float vals[NCP];
#pragma unroll
          for (int c = 0; c < NCP; ++c) {
            // gf sign(d): gf with d's sign bit (v_bfi), or 0 where d == 0 -- four operations, not seven
            // (classes from Nl on: dist and pr are both 0 there)
            const float d = sdist_t[(l - 1) * kMaxCls + c] - pr[c];
            vals[c] = d == 0.0f ? 0.0f : gf * __builtin_copysignf(1.0f, d);
          }
          const float red = wave_transpose_reduce<NCP>(vals, lane);
          const int c = bitrev6(lane);
          if (c < Nl) acc_d[wave][(l - 1) * kMaxCls + c] += red;
        }
      }
    }
  }
  __syncthreads();
  if (g_occ != nullptr)
    for (int e = threadIdx.x; e < LP * LP; e += kBlock) {
      const int i = e / LP, j = e % LP;
      if (i < L && j < L)
        atomicAdd(g_occ + ((int64_t)b * T + t) * L * L + i * L + j,
                  (acc_o[0][e] + acc_o[1][e]) + (acc_o[2][e] + acc_o[3][e]));
    }
  if (filt && g_dist != nullptr)
    for (int e = threadIdx.x; e < (LP - 1) * kMaxCls; e += kBlock) {
      const int o = e / kMaxCls, c = e % kMaxCls;
      if (o < No && c < Nl)
        atomicAdd(g_dist + ((int64_t)b * No + o) * Nl + c, (acc_d[0][e] + acc_d[1][e]) + (acc_d[2][e] + acc_d[3][e]));
    }
}

#ifndef WALDO_FCB_WARP_WAVES
#define WALDO_FCB_WARP_WAVES 3
#endif
template <int LP>
__global__ __launch_bounds__(kBlock, WALDO_FCB_WARP_WAVES) void flow_ctx_warp_bwd_kernel(
    const float* __restrict__ flow_lr, const float* __restrict__ isobj_lr,
    const float* __restrict__ a01, const int64_t* __restrict__ ctx_ts,
    const int64_t* __restrict__ pred_ts, const float* __restrict__ occ,
    const float* __restrict__ g_flow, const float* __restrict__ g_actx,
    const float* __restrict__ g_dis, float* __restrict__ g_fup, float* __restrict__ g_a01,
    float* __restrict__ g_occ, int T, int Tw, int Tc, int Tp, int L, int H, int W, int scale,
    int tiles, int tiles_per_block, int groups) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const int m = blockIdx.x / groups;  // (b, tc, tp)
  const int tp = m % Tp, b = m / (Tc * Tp);
  const int t0 = (blockIdx.x % groups) * tiles_per_block, t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __shared__ float acc_o[4][LP * LP];
  __shared__ __attribute__((aligned(16))) float occm[OccLds<LP>::kFloats];
  for (int e = lane; e < LP * LP; e += kWave) acc_o[wave][e] = 0.0f;
  // (wave-uniform; through the vector path they land in VGPRs and so does every plane address)
  const int ts = __builtin_amdgcn_readfirstlane((int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(Tw - 1)));
  const int tpred = __builtin_amdgcn_readfirstlane((int)min(max(pred_ts[tp], (int64_t)0), (int64_t)(T - 1)));
  occ_stage<LP>(occm, occ + ((int64_t)b * T + tpred) * L * L, L);
  __syncthreads();
  const float hw = 0.5f * (float)Wd, hh = 0.5f * (float)Hd;

  for (int tile = t0; tile < t1; ++tile) {
    int fresh = 0;  // (see flow_ctx_alpha_bwd_kernel: keeps the reads of the order inside the loop)
    asm volatile("" : "+v"(fresh));
    const float* occm_t = occm + fresh;
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HWd;
    const int64_t pc = live ? p : HWd - 1;
    const int y = (int)(pc / Wd), x = (int)(pc - (int64_t)y * Wd);
    const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
    float gx0, gy0;
    identity_grid(x, y, Wd, Hd, gx0, gy0);
    const float gfx = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2) * HWd + pc] : 0.0f;
    const float gfy = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2 + 1) * HWd + pc] : 0.0f;
    // ---- forward values: a_l = sample of the context alpha at (pixel + flow_l) times the ghost mask, and the
    // disocclusion arg max.  Only a[] and gv[] outlive this loop: the flows, the taps and the derivatives of
    // the samples are computed AGAIN in the last loop, where they are used -- kept (five more arrays of L) the
    // kernel wanted 210+ registers at L = 17 and spent its time on ~600 spilled dwords per tile.
    float a[LP], gv[LP], ga[LP];
    float dis = -INFINITY;
    int amax = 0;
    const float* ap = a01 + (((int64_t)b * Tw + ts) * L) * HWd;
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int lc = min(l, L - 1);
      const float* fl = flow_lr + (((int64_t)m * L + lc) * 2) * HW;
      const float fxl = up_sample(fl, ut), fyl = up_sample(fl + HW, ut);
      const Taps t = make_taps(gx0 + fxl, gy0 + fyl, Hd, Wd);
      float v = tap_sample(ap + (int64_t)lc * HWd, t);
      float ghost = 1.0f;
      if (isobj_lr != nullptr && l >= 1)
        ghost = (up_sample(isobj_lr + ((int64_t)m * (L - 1) + max(lc - 1, 0)) * HW, ut) > 0.9f) ? 1.0f : 0.0f;
      const bool real = l < L;
      v *= ghost;
      a[l] = real ? v : 0.0f;
      if (real && v > dis) {
        dis = v;
        amax = l;
      }
      gv[l] = (real && live) ? fmaf(gfx, fxl, gfy * fyl) : 0.0f;
      if (real && live && g_actx != nullptr) gv[l] = fmaf(2.0f, g_actx[((int64_t)m * L + lc) * HWd + pc], gv[l]);
      if ((l & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    composite_bwd<LP>(a, gv, occm_t, L, ga, g_occ != nullptr ? acc_o[wave] : nullptr, lane);
    const float gd = (g_dis != nullptr && live) ? g_dis[(int64_t)m * HWd + pc] : 0.0f;
    typedef float f32x2_w __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < LP; j += 4) {
      // v_j (the composited alphas) of four layers, as the forward kernel takes them (same bits) ...
      f32x2_w prd[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const f32x4_o o = occ_quad<LP, false>(occm_t, i, j);
        const f32x2_w ai = {a[i], a[i]}, one = {1.0f, 1.0f};
        prd[0] = prd[0] * __builtin_elementwise_fma(-ai, (f32x2_w){o[0], o[1]}, one);
        if (j + 2 < LP) prd[1] = prd[1] * __builtin_elementwise_fma(-ai, (f32x2_w){o[2], o[3]}, one);
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four rows in flight, not all LP
      }
      // ... and their part of the backward
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int l = j + k < LP ? j + k : LP - 1;
        if (j + k < LP && l < L) {  // wave-uniform
          const float vjl = a[l] * prd[k >> 1][k & 1];
          const float* fl = flow_lr + (((int64_t)m * L + l) * 2) * HW;
          const float fxl = up_sample(fl, ut), fyl = up_sample(fl + HW, ut);
          float ghost = 1.0f;
          if (isobj_lr != nullptr && l >= 1)
            ghost = (up_sample(isobj_lr + ((int64_t)m * (L - 1) + (l - 1)) * HW, ut) > 0.9f) ? 1.0f : 0.0f;
          const Taps t = make_taps(gx0 + fxl, gy0 + fyl, Hd, Wd);
          float ddx, ddy;
          (void)tap_sample_d(ap + (int64_t)l * HWd, t, ddx, ddy);
          const float gs = ga[l] + (l == amax ? gd : 0.0f);  // d loss / d a_l
          // d / d flow_l: through the flow composite and through the sample position
          const float gfxl = fmaf(gs * (ddx * ghost), hw, vjl * gfx);
          const float gfyl = fmaf(gs * (ddy * ghost), hh, vjl * gfy);
          if (live) {
            g_fup[(((int64_t)m * L + l) * 2) * HWd + p] = gfxl;
            g_fup[(((int64_t)m * L + l) * 2 + 1) * HWd + p] = gfyl;
          }
          // d / d a01: bilinear splat of gs * ghost
          if (g_a01 != nullptr) {
            const float gsg = live ? gs * ghost : 0.0f;
            if (gsg != 0.0f) {
              float* gp = g_a01 + (((int64_t)b * Tw + ts) * L + l) * HWd;
              // the lerp form's weights: (1-fx)(1-fy) v00 ... with the validity of each corner
              const float wx0 = 1.0f - t.fx, wy0 = 1.0f - t.fy;
              const float w00 = wx0 * wy0 * (t.vx0 * t.vy0), w01 = t.fx * wy0 * (t.vx1 * t.vy0);
              const float w10 = wx0 * t.fy * (t.vx0 * t.vy1), w11 = t.fx * t.fy * (t.vx1 * t.vy1);
#ifndef WALDO_ABL_FCB_NOATOMIC  // timing-only ablation: without the scatter
              if (w00 != 0.0f) atomicAdd(gp + (t.o00 >> 2), gsg * w00);
              if (w01 != 0.0f) atomicAdd(gp + (t.o01 >> 2), gsg * w01);
              if (w10 != 0.0f) atomicAdd(gp + (t.o10 >> 2), gsg * w10);
              if (w11 != 0.0f) atomicAdd(gp + (t.o11 >> 2), gsg * w11);
#else
              if (w00 + w01 + w10 + w11 == 123.0f) gp[0] = gsg;
#endif
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
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

// transpose of F.interpolate(scale_factor=S, "bilinear", align_corners=False) on P planes:
// g_lr[y][x] = sum over the HD pixels (Y, X) that interpolate from (y, x) of wy wx g_hd[Y][X]
__global__ __launch_bounds__(kBlock) void upsample_bwd_kernel(const float* __restrict__ g_hd,
                                                              float* __restrict__ g_lr, int64_t P,
                                                              int H, int W, int scale) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= P * H * W) return;
  const int x = (int)(e % W), y = (int)((e / W) % H);
  const int64_t pl = e / ((int64_t)H * W);
  const float inv = 1.0f / (float)scale;
  const float* g = g_hd + pl * Hd * Wd;
  const int Y0 = max(0, (y - 1) * scale), Y1 = min(Hd - 1, (y + 2) * scale - 1);
  const int X0 = max(0, (x - 1) * scale), X1 = min(Wd - 1, (x + 2) * scale - 1);
  float acc = 0.0f;
  for (int Y = Y0; Y <= Y1; ++Y) {
    const UpTap ty = up_tap(Y, inv, H);
    const float wy = (ty.i0 == y ? ty.l0 : 0.0f) + (ty.i1 == y ? ty.l1 : 0.0f);
    if (wy == 0.0f) continue;
    float row = 0.0f;
    for (int X = X0; X <= X1; ++X) {
      const UpTap tx = up_tap(X, inv, W);
      const float wx = (tx.i0 == x ? tx.l0 : 0.0f) + (tx.i1 == x ? tx.l1 : 0.0f);
      row = fmaf(wx, g[(int64_t)Y * Wd + X], row);
    }
    acc = fmaf(wy, row, acc);
  }
  g_lr[e] = acc;
}

// backward of frame_warp_fuse_kernel: see its header for the forward.  One thread per HD pixel of
// one (b, tp); per context the tap offsets, fractions and corner validities stay in registers and
// every input channel is sampled once with its derivatives.
template <int TCP>
__global__ __launch_bounds__(kBlock) void frame_warp_fuse_bwd_kernel(
    const float* __restrict__ input, const float* __restrict__ flow, const float* __restrict__ alpha,
    const int64_t* __restrict__ ctx_ts, const float* __restrict__ g_out, const float* __restrict__ g_raw,
    float* __restrict__ g_flow, float* __restrict__ g_alpha, int T, int Tc, int Tp, int C, int L, int Hd,
    int Wd, int include_self, float eps, int tiles) {
  const int64_t HWd = (int64_t)Hd * Wd;
  const int n = blockIdx.x / tiles;  // (b, tp)
  const int b = n / Tp, tp = n % Tp;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HWd) return;
  const int y = (int)(p / Wd), x = (int)(p - (int64_t)y * Wd);
  float gx0, gy0;
  identity_grid(x, y, Wd, Hd, gx0, gy0);
  const int Tcx = Tc + (include_self ? 1 : 0);
  uint32_t o00[TCP], o01[TCP], o10[TCP], o11[TCP];
  float tfx[TCP], tfy[TCP], m00[TCP], m01[TCP], m10[TCP], m11[TCP], sc[TCP];
  const float* frame[TCP];
  float ssum = 0.0f;
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    const int tcc = min(tc, Tc - 1);
    const bool real = tc < Tc;
    const int64_t m = ((int64_t)b * Tc + tcc) * Tp + tp;
    const float* fl = flow + m * 2 * HWd + p;
    const Taps t = make_taps(gx0 + fl[0], gy0 + fl[HWd], Hd, Wd);
    o00[tc] = t.o00;
    o01[tc] = t.o01;
    o10[tc] = t.o10;
    o11[tc] = t.o11;
    tfx[tc] = t.fx;
    tfy[tc] = t.fy;
    m00[tc] = t.vx0 * t.vy0;
    m01[tc] = t.vx1 * t.vy0;
    m10[tc] = t.vx0 * t.vy1;
    m11[tc] = t.vx1 * t.vy1;
    const int ts = (int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(T - 1));
    frame[tc] = input + ((int64_t)b * T + ts) * C * HWd;
    const float* al = alpha + m * L * HWd + p;
    float s = 0.0f;
    for (int l = 0; l < L; ++l) s += (al[(int64_t)l * HWd] + 1.0f) / 2.0f;
    sc[tc] = s;
    ssum += real ? fabsf(s + eps) : 0.0f;
  }
  if (include_self) ssum += fabsf(1.0f + eps);
  const float den = fmaxf(ssum, 1e-12f);
  const bool normd = ssum > 1e-12f;  // else the denominator is the constant floor
  float wt[TCP], gw[TCP], gpx[TCP], gpy[TCP];
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    wt[tc] = (tc < Tc) ? (sc[tc] + eps) / den : 0.0f;
    gw[tc] = gpx[tc] = gpy[tc] = 0.0f;
  }
  float gwself = 0.0f;
  const float* self = input + ((int64_t)b * T + min(tp, T - 1)) * C * HWd + p;
  const float* go = g_out != nullptr ? g_out + ((int64_t)b * Tp + tp) * (C + 1) * HWd + p : nullptr;
  const float* gr = g_raw != nullptr ? g_raw + ((int64_t)b * Tcx * Tp + tp) * (C + L) * HWd + p : nullptr;
  for (int c = 0; c < C; ++c) {  // (NOT unrolled: unrolled by four the kernel went from 55 to 79 us at the LVD recipe)
    const float goc = go != nullptr ? go[(int64_t)c * HWd] : 0.0f;
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc) {
      if (tc < Tc) {
        const float* plane = frame[tc] + (int64_t)c * HWd;
        const float v00 = ldb(plane, o00[tc]) * m00[tc], v01 = ldb(plane, o01[tc]) * m01[tc];
        const float v10 = ldb(plane, o10[tc]) * m10[tc], v11 = ldb(plane, o11[tc]) * m11[tc];
        const float top = fmaf(tfx[tc], v01 - v00, v00);
        const float bot = fmaf(tfx[tc], v11 - v10, v10);
        const float ddx = fmaf(tfy[tc], (v11 - v10) - (v01 - v00), v01 - v00);
        const float ddy = bot - top;
        const float v = fmaf(tfy[tc], ddy, top);
        const float gwarp = fmaf(goc, wt[tc], gr != nullptr ? gr[((int64_t)tc * Tp * (C + L) + c) * HWd] : 0.0f);
        gw[tc] = fmaf(goc, v, gw[tc]);
        gpx[tc] = fmaf(gwarp, ddx, gpx[tc]);
        gpy[tc] = fmaf(gwarp, ddy, gpy[tc]);
      }
    }
    if (include_self) gwself = fmaf(goc, self[(int64_t)c * HWd], gwself);
  }
  const float gos = go != nullptr ? go[(int64_t)C * HWd] : 0.0f;  // the score channel
  float S = 0.0f;
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    if (tc < Tc) {
      gw[tc] = fmaf(gos, sc[tc] * 2.0f - 1.0f, gw[tc]);
      S = fmaf(gw[tc], sc[tc] + eps, S);
    }
  }
  if (include_self) {
    gwself += gos;
    S = fmaf(gwself, 1.0f + eps, S);
  }
  const float hw = 0.5f * (float)Wd, hh = 0.5f * (float)Hd;
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    if (tc < Tc) {
      const int64_t m = ((int64_t)b * Tc + tc) * Tp + tp;
      const float u = sc[tc] + eps;
      float gu = gw[tc] / den;
      if (normd) gu -= sgnf(u) * S / (den * den);
      const float gsc = fmaf(gos * 2.0f, wt[tc], gu);  // d loss / d score: weights and the score channel
      g_flow[m * 2 * HWd + p] = gpx[tc] * hw;
      g_flow[(m * 2 + 1) * HWd + p] = gpy[tc] * hh;
      float* ga = g_alpha + m * L * HWd + p;
      for (int l = 0; l < L; ++l)
        ga[(int64_t)l * HWd] = fmaf(0.5f, gsc, gr != nullptr ? gr[((int64_t)tc * Tp * (C + L) + C + l) * HWd] : 0.0f);
    }
  }
}

static int acc_tiles(int64_t units, int64_t tiles) {  // tiles per workgroup: keep >= ~512 workgroups (fewer: measured slower, the per-tile reductions dominate)
  return (int)min((int64_t)kAccTilesMax, max((int64_t)1, units * tiles / WALDO_FCB_MIN_WGS));
}

// the lane-layer kernels (tools_dev/dropped/flow_ctx_bwd_rows.hip.h: round 5's experiment) serve 9 .. 17 layers; measured
// no faster than the per-pixel ones at the LVD recipe (warp 0.210 against 0.217 ms, alpha 0.206 against 0.165): they
// exist in VARIANT builds only (tools_dev/build_variant.py NAME -DWALDO_VARIANT_FCB_ROWS), where they take every
// launch they can serve
#ifdef WALDO_VARIANT_FCB_ROWS
static bool rows_kernels(int L) { return L >= 9 && L <= 17; }
#endif

}  // namespace waldo

#ifdef WALDO_VARIANT_FCB_ROWS
#include "flow_ctx_bwd_rows.hip.h"
#endif

using namespace waldo;

static int check_bwd_shape(const char* fn, int64_t N, int L, int H, int W, int scale) {
  if (N < 0 || L < 1 || L > 32 || H < 1 || W < 1 || scale < 1 || scale > 64 || (int64_t)H * scale > 32767 ||
      (int64_t)W * scale > 32767 || N * (((int64_t)H * scale * W * scale + kBlock - 1) / kBlock) > 2147483647) {
    set_error("%s: bad shape N=%lld L=%d H=%d W=%d scale=%d", fn, (long long)N, L, H, W, scale);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

#define WALDO_FCB_CASE(LPV, KERNEL, ...)                                                         \
  case LPV:                                                                                      \
    hipLaunchKernelGGL((KERNEL<LPV>), dim3((unsigned)(N * groups)), dim3(kBlock), 0, st, __VA_ARGS__); \
    break;

extern "C" int waldo_flow_ctx_alpha_bwd(const float* alpha_lr, const float* input, const float* dist,
                                        const float* occ, const float* grad_a01, const float* grad_alpha_out,
                                        float* grad_alpha_lr, float* grad_dist, float* grad_occ, float* workspace,
                                        int B, int T, int Tw, int L, int Nl, int C, int chan_off, int H, int W,
                                        int scale, waldo_stream_t stream) {
  const int64_t N = (int64_t)B * Tw;
  if (int rc = check_bwd_shape("waldo_flow_ctx_alpha_bwd", N, L, H, W, scale)) return rc;
  if (B < 0 || T < 1 || Tw < 1 || Tw > T ||
      (dist != nullptr && (Nl < 1 || Nl > kMaxCls || chan_off < 0 || chan_off + Nl > C))) {
    set_error("waldo_flow_ctx_alpha_bwd: bad frame window Tw=%d of T=%d or class channels", Tw, T);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!alpha_lr || !occ || (!grad_a01 && !grad_alpha_out) || !grad_alpha_lr || (dist != nullptr && !input) || (scale > 1 && !workspace)) {
    set_error("waldo_flow_ctx_alpha_bwd: null pointer (one of grad_a01 / grad_alpha_out; scale > 1 needs the (B*Tw, L, Hd, Wd) workspace)");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (int)(((int64_t)H * scale * W * scale + kBlock - 1) / kBlock);
  const int tpb = acc_tiles(N, tiles), groups = (tiles + tpb - 1) / tpb;
  float* gup = scale > 1 ? workspace : grad_alpha_lr;
#ifdef WALDO_VARIANT_FCB_ROWS
  if (rows_kernels(L)) {
    if (dist == nullptr || Nl <= 20)
      hipLaunchKernelGGL(flow_ctx_alpha_bwd_rows_kernel<20>, dim3((unsigned)(N * groups)), dim3(kBlock), 0, st, alpha_lr,
                         input, dist, occ, grad_a01, grad_alpha_out, gup, grad_dist, grad_occ, T, Tw, L, Nl, C, chan_off, H, W,
                         scale, tiles, tpb, groups);
    else
      hipLaunchKernelGGL(flow_ctx_alpha_bwd_rows_kernel<32>, dim3((unsigned)(N * groups)), dim3(kBlock), 0, st, alpha_lr,
                         input, dist, occ, grad_a01, grad_alpha_out, gup, grad_dist, grad_occ, T, Tw, L, Nl, C, chan_off, H, W,
                         scale, tiles, tpb, groups);
  } else
#endif
  {
    // (the class probabilities of a pixel live in registers: compiled for up to kFewCls classes and for kMaxCls)
#define WALDO_FCAB_CASE(LPV)                                                                                             \
  case LPV:                                                                                                              \
    if (dist == nullptr || Nl <= kFewCls)                                                                                \
      hipLaunchKernelGGL((flow_ctx_alpha_bwd_kernel<LPV, kFewCls>), dim3((unsigned)(N * groups)), dim3(kBlock), 0, st,   \
                         alpha_lr, input, dist, occ, grad_a01, grad_alpha_out, gup, grad_dist, grad_occ, T, Tw, L, Nl, C, \
                         chan_off, H, W, scale, tiles, tpb, groups);                                                     \
    else                                                                                                                 \
      hipLaunchKernelGGL((flow_ctx_alpha_bwd_kernel<LPV, kMaxCls>), dim3((unsigned)(N * groups)), dim3(kBlock), 0, st,   \
                         alpha_lr, input, dist, occ, grad_a01, grad_alpha_out, gup, grad_dist, grad_occ, T, Tw, L, Nl, C, \
                         chan_off, H, W, scale, tiles, tpb, groups);                                                     \
    break;
    switch (flow_ctx_pad_l(L)) {
      WALDO_FCAB_CASE(4)
      WALDO_FCAB_CASE(8)
      WALDO_FCAB_CASE(12)
      WALDO_FCAB_CASE(17)
      WALDO_FCAB_CASE(24)
      WALDO_FCAB_CASE(32)
    }
#undef WALDO_FCAB_CASE
  }
  if (scale > 1) {
    const int64_t P = N * L;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((P * H * W + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       st, workspace, grad_alpha_lr, P, H, W, scale);
  }
  return launch_status("waldo_flow_ctx_alpha_bwd");
}

extern "C" int waldo_flow_ctx_warp_bwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                                       const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ,
                                       const float* grad_flow, const float* grad_alpha_ctx,
                                       const float* grad_disocc, float* grad_flow_lr, float* grad_a01,
                                       float* grad_occ, float* workspace, int B, int T, int Tw, int Tc,
                                       int Tp, int L, int H, int W, int scale, waldo_stream_t stream) {
  const int64_t N = (int64_t)B * Tc * Tp;
  if (int rc = check_bwd_shape("waldo_flow_ctx_warp_bwd", N, L, H, W, scale)) return rc;
  if (B < 0 || T < 1 || Tw < 1 || Tw > T || Tc < 0 || Tp < 0) {
    set_error("waldo_flow_ctx_warp_bwd: bad frame counts T=%d Tw=%d Tc=%d Tp=%d", T, Tw, Tc, Tp);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!flow_lr || !a01 || !ctx_ts || !pred_ts || !occ || !grad_flow_lr || (scale > 1 && !workspace)) {
    set_error("waldo_flow_ctx_warp_bwd: null pointer (scale > 1 needs the (M, L, 2, Hd, Wd) workspace)");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (int)(((int64_t)H * scale * W * scale + kBlock - 1) / kBlock);
  const int tpb = acc_tiles(N, tiles), groups = (tiles + tpb - 1) / tpb;
  float* gup = scale > 1 ? workspace : grad_flow_lr;
#ifdef WALDO_VARIANT_FCB_ROWS
  if (rows_kernels(L))
    hipLaunchKernelGGL(flow_ctx_warp_bwd_rows_kernel, dim3((unsigned)(N * groups)), dim3(kBlock), 0, st, flow_lr, isobj_lr,
                       a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw,
                       Tc, Tp, L, H, W, scale, tiles, tpb, groups);
  else
#endif
  switch (flow_ctx_pad_l(L)) {
    WALDO_FCB_CASE(4, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
    WALDO_FCB_CASE(8, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
    WALDO_FCB_CASE(12, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
    WALDO_FCB_CASE(17, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
    WALDO_FCB_CASE(24, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
    WALDO_FCB_CASE(32, flow_ctx_warp_bwd_kernel, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, grad_flow, grad_alpha_ctx, grad_disocc, gup, grad_a01, grad_occ, T, Tw, Tc, Tp, L, H, W, scale, tiles, tpb, groups)
  }
  if (scale > 1) {
    const int64_t P = N * L * 2;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((P * H * W + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       st, workspace, grad_flow_lr, P, H, W, scale);
  }
  return launch_status("waldo_flow_ctx_warp_bwd");
}

extern "C" int waldo_frame_warp_fuse_bwd(const float* input, const float* flow, const float* alpha,
                                         const int64_t* ctx_ts, const float* grad_out, const float* grad_raw,
                                         float* grad_flow, float* grad_alpha, int B, int T, int Tc, int Tp,
                                         int C, int L, int Hd, int Wd, int include_self, float eps,
                                         waldo_stream_t stream) {
  if (B < 0 || T < 1 || Tc < 1 || Tc + (include_self ? 1 : 0) > kFwMaxCtx || Tp < 1 || C < 1 || L < 1 ||
      Hd < 1 || Wd < 1 || Hd > 32767 || Wd > 32767 || (include_self && Tp != T)) {
    set_error("waldo_frame_warp_fuse_bwd: bad shape B=%d T=%d Tc=%d Tp=%d C=%d L=%d Hd=%d Wd=%d include_self=%d",
              B, T, Tc, Tp, C, L, Hd, Wd, include_self);
    return WALDO_EINVAL;
  }
  const int64_t tiles = ((int64_t)Hd * Wd + kBlock - 1) / kBlock;
  if ((int64_t)B * Tp * tiles > 2147483647) {
    set_error("waldo_frame_warp_fuse_bwd: problem too large for one launch");
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!input || !flow || !alpha || !ctx_ts || !grad_flow || !grad_alpha) {
    set_error("waldo_frame_warp_fuse_bwd: null pointer");
    return WALDO_EINVAL;
  }
  const dim3 grid((unsigned)((int64_t)B * Tp * tiles));
  // (a padding context repeats the last real one's taps and its L alpha loads: the count is compiled in for 1, 2, 4)
#define WALDO_FWFB_LAUNCH(TCPV)                                                                                         \
  hipLaunchKernelGGL(frame_warp_fuse_bwd_kernel<TCPV>, grid, dim3(kBlock), 0, (hipStream_t)stream, input, flow, alpha, \
                     ctx_ts, grad_out, grad_raw, grad_flow, grad_alpha, T, Tc, Tp, C, L, Hd, Wd, include_self, eps,      \
                     (int)tiles)
  if (Tc == 1) WALDO_FWFB_LAUNCH(1);
  else if (Tc == 2) WALDO_FWFB_LAUNCH(2);
  else if (Tc <= 4) WALDO_FWFB_LAUNCH(4);
  else WALDO_FWFB_LAUNCH(8);
#undef WALDO_FWFB_LAUNCH
  return launch_status("waldo_frame_warp_fuse_bwd");
}
