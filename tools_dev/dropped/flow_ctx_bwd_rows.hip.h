// Backward of the two full-resolution flow passes for 9 .. 17 layers (background + up to 16 objects: the Cityscapes
// recipe's L = 17), with the L x L part -- the occlusion composite's backward -- on LANE-LAYERS instead of one pixel
// per lane (round 5).
//
// What was wrong with one pixel per lane at L = 17 (flow_ctx_bwd.hip:composite_bwd, still used for the other layer
// counts): a lane carries a[17], gv[17], ga[17] and, per column j, tf[17], ex[17], gocc[17]; the kernels want 210 / 300+
// registers, are capped at 168 and spill 100 / 1000 bytes per lane; every column ends in a 17-value transpose-reduce
// over the wavefront for grad_occ; a wavefront's dependent chain is 17-21 k vector instructions long and the
// LVD-recipe step has only 2560 such wavefronts: 2.5 per SIMD, VALU issue 31 / 49 % (profiles/r04_lvd_step_counters.txt).
//
// Here a wavefront still owns 64 pixels, in three phases that talk through two wave-private LDS tables
// A[layer][pixel], G[layer][pixel] (pitch 66 floats: conflict-free under both lane mappings):
//   1. lane = pixel: the forward values a_l (sampled / upsampled alpha) and the incoming gradients gv_l of every
//      layer, coalesced loads as before                                              -> A, G
//   2. lane = (pixel q of 4, object i of 16), sixteen passes of four pixels: v_j = a_j prod_i (1 - a_i occ[i][j]).
//      Per column j a lane forms ITS factor, the product of the row's other fifteen comes from a four-level DPP
//      butterfly (quad_perm / row_half_mirror / row_mirror: plain VALU, no LDS crossbar) that also yields the full
//      product; the background layer -- one value per pixel -- is carried by every lane of the row.  d / d a_i of all
//      17 columns accumulates in ONE register, grad_occ[i][j] over all the pixels a lane ever sees in 17 registers:
//      no per-pixel reduction at all, one 4-row sum per lane at the end of the workgroup.     A <- v, G <- d / d a
//   3. lane = pixel: what the old kernels did with ga[] and v[] (sample derivatives, splat, stores), coalesced.
// 128 registers, no scratch inside phase 2, four waves per SIMD.  Products are formed in butterfly order instead of
// layer order: same value up to rounding (tests/test_gpu_warper.py: oracle autograd in fp32 and fp64, and the
// per-pixel kernels).
//
// MEASURED (LVD-recipe step, A/B on one box, profiles/r05_ab_lvd_rows_kernels.txt): flow_ctx_warp_bwd 0.210 ms against
// 0.217 per pixel, flow_ctx_alpha_bwd 0.206 against 0.165 -- NOT faster, so these kernels run only behind
// WALDO_DEBUG_FCB_ROWS.  Timing-only ablations of the warp kernel: without phase 2 0.172, without the splat's atomics
// 0.170, without both 0.113: the L x L part that this design attacks is a fifth of the kernel; what binds it are phases
// 1 and 3, ~9 k static instructions per tile of per-layer sampling code (1500 scalar multiply / adds of 64-bit plane
// addresses, 800 v_readlane / v_writelane of spilled scalars), the same in the per-pixel kernels.  More waves (one
// tile per workgroup), deeper load groups and three waves per SIMD all left the time where it was.
#pragma once
// included by flow_ctx_bwd.hip behind its helpers (sgnf, acc_tiles)

namespace waldo {

#ifndef WALDO_ROWS_GROUP
#define WALDO_ROWS_GROUP 4  // layers whose loads phases 1 and 3 keep in flight together (a power of two)
#endif
#ifndef WALDO_ROWS_WAVES
#define WALDO_ROWS_WAVES 4
#endif
constexpr int kRowsLP = 17;     // background + 16 objects
constexpr int kRowsPitch = 66;  // floats per layer row of a wave's table

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  // (every source lane of these patterns is valid: the `old` operand is never used -- zero with bound_ctrl lets the
  // compiler fold the move into the consuming multiply as a DPP operand)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lane J of every 16-lane row, to all lanes of that row (ds_swizzle, bit-mask mode: lane' = (lane & 0x10) | J within each
// half of the wavefront; LDS crossbar, no memory)
template <int J>
__device__ __forceinline__ float row_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x10 | (J << 5)));
}
template <int J>
__device__ __forceinline__ float row_bcast_or(float v, float first) {
  if constexpr (J < 0) return first;
  else return row_bcast<J>(v);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One pass of phase 2 for this lane's (pixel, object i1 + 1): a_me / gv_me the object's forward value and incoming
// gradient, a0 / gv0 the pixel's background layer (the same in the row's 16 lanes).  o_me[j] = occ[i1 + 1][j],
// o0[j] = occ[0][j].  Returns d loss / d a of the object (ga_me) and of the background (ga0), and the full products
// of the object's own column (pcol_me) and of column 0 (p0): v_j = a_j * product.  gocc_me[j] / gocc0[j] accumulate
// d loss / d occ[i1 + 1][j] and d loss / d occ[0][j] (the latter identically in the row's lanes).
// Reference: the backward of `alpha * (1 - alpha.unsqueeze(3) * occ).prod(dim=2)` (models/nets/lvd.py:651-652, 686,
// 764-765, 809); per column: flow_ctx_bwd.hip:composite_bwd.
template <int Jm1>
struct RowsColumn {
  __device__ __forceinline__ static void run(float a_me, float a0, float c_me, float c0, const float (&o_me)[kRowsLP],
                                             const float (&o0)[kRowsLP], int i1, int L, float& ga_me, float& ga0,
                                             float& pcol_me, float& p0, float (&gocc_me)[kRowsLP], float& g0col,
                                             float& g00) {
    constexpr int j = Jm1 + 1;
    if (j < L) {  // wave-uniform
      const float t = fmaf(-a_me, o_me[j], 1.0f);
      const float t0 = fmaf(-a0, o0[j], 1.0f);
      // e: product of the factors of the row's OTHER lanes; b: product of the whole block so far (pair, quad, half row, row)
      float b = t;
      float e = dpp_mov<0xB1>(b);   // quad_perm [1, 0, 3, 2]
      b = b * dpp_mov<0xB1>(b);
      e = e * dpp_mov<0x4E>(b);     // quad_perm [2, 3, 0, 1]
      b = b * dpp_mov<0x4E>(b);
      e = e * dpp_mov<0x141>(b);    // row_half_mirror
      b = b * dpp_mov<0x141>(b);
      e = e * dpp_mov<0x140>(b);    // row_mirror
      b = b * dpp_mov<0x140>(b);
      const float ex = e * t0;      // all layers but this lane's object
      const float prod = b * t0;    // the column's full product
      const float cj = row_bcast_or<Jm1>(c_me, c0);  // gv_j a_j
      const float w = cj * ex;
      ga_me = fmaf(-w, o_me[j], ga_me);
      gocc_me[j] = fmaf(-w, a_me, gocc_me[j]);
      const float w0 = cj * b;      // (background row: everything but the background's own factor)
      ga0 = fmaf(-w0, o0[j], ga0);
      if constexpr (j == 0) {
        p0 = prod;
        g00 = fmaf(-w0, a0, g00);
      } else {
        const bool mine = i1 == Jm1;  // the lane whose object IS layer j
        pcol_me = mine ? prod : pcol_me;
        g0col = fmaf(-w0, mine ? a0 : 0.0f, g0col);
      }
    }
    if constexpr ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (a few columns' broadcasts in flight, not all)
    if constexpr (j + 1 < kRowsLP)
      RowsColumn<Jm1 + 1>::run(a_me, a0, c_me, c0, o_me, o0, i1, L, ga_me, ga0, pcol_me, p0, gocc_me, g0col, g00);
  }
};

// gocc_me[j] accumulates d loss / d occ[i1 + 1][j]; the background row d loss / d occ[0][j] is computed by every lane
// of a pixel's row alike, so lane i1 keeps column i1 + 1 of it (g0col) and column 0 stays with all of them (g00).
__device__ __forceinline__ void rows_composite_bwd(float a_me, float gv_me, float a0, float gv0,
                                                   const float (&o_me)[kRowsLP], const float (&o0)[kRowsLP], int i1,
                                                   int L, float& ga_me, float& ga0, float& pcol_me, float& p0,
                                                   float (&gocc_me)[kRowsLP], float& g0col, float& g00) {
  ga_me = 0.0f;
  ga0 = 0.0f;
  pcol_me = 1.0f;
  p0 = 1.0f;
  RowsColumn<-1>::run(a_me, a0, gv_me * a_me, gv0 * a0, o_me, o0, i1, L, ga_me, ga0, pcol_me, p0, gocc_me, g0col, g00);
  // the column of a layer itself: d v_j / d a_j also has the bare product
  ga_me = fmaf(gv_me, pcol_me, ga_me);
  ga0 = fmaf(gv0, p0, ga0);
}

// phase 2 over the wave's 64 pixels: A = forward values in, composited values v out; G = incoming gradients in,
// d loss / d a out
__device__ __forceinline__ void rows_phase2(float (*A)[kRowsPitch], float (*G)[kRowsPitch], const float (&o_me)[kRowsLP],
                                            const float (&o0)[kRowsLP], int q, int i1, int L,
                                            float (&gocc_me)[kRowsLP], float& g0col, float& g00) {
#pragma unroll 1
  for (int s = 0; s < 16; ++s) {
    const int px = 4 * s + q;
    const float a_me = A[i1 + 1][px], gv_me = G[i1 + 1][px];
    const float a0 = A[0][px], gv0 = G[0][px];
    float ga_me, ga0, pcol_me, p0;
    rows_composite_bwd(a_me, gv_me, a0, gv0, o_me, o0, i1, L, ga_me, ga0, pcol_me, p0, gocc_me, g0col, g00);
    G[i1 + 1][px] = ga_me;
    A[i1 + 1][px] = a_me * pcol_me;
    if (i1 == 0) {
      G[0][px] = ga0;
      A[0][px] = a0 * p0;
    }
  }
}

// the rows of the order this lane needs -- its object's and the background's (uniform: scalar registers) -- out of a
// row-major copy staged in LDS (pitch R, padding rows / columns repeat the last real one: a padding layer's alpha is 0)
constexpr int kRowsOccPitch = 20;
__device__ __forceinline__ void rows_stage_order(float* stage, const float* __restrict__ oc, int L) {
  for (int e = threadIdx.x; e < kRowsLP * kRowsOccPitch; e += kBlock) {
    const int r = e / kRowsOccPitch, c = e - r * kRowsOccPitch;
    stage[e] = oc[min(r, L - 1) * L + min(c, L - 1)];
  }
}
__device__ __forceinline__ void rows_order(const float* stage, int i1, float (&o_me)[kRowsLP], float (&o0)[kRowsLP]) {
#pragma unroll
  for (int j = 0; j < kRowsLP; ++j) {
    o_me[j] = stage[(i1 + 1) * kRowsOccPitch + j];
    o0[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(stage[j])));
  }
}

// a value of every lane combined over its 16-lane row (all lanes end up with the result): the butterfly of the products
template <class Op>
__device__ __forceinline__ float row_all(float v, Op op) {
  v = op(v, dpp_mov<0xB1>(v));
  v = op(v, dpp_mov<0x4E>(v));
  v = op(v, dpp_mov<0x141>(v));
  return op(v, dpp_mov<0x140>(v));
}

// grad_occ of the workgroup: the lanes' registers summed over the four pixel rows, then over the four waves through
// `acc` ([4][LP * LP], aliases the tables: the caller has synchronised), one float atomic per entry.
__device__ __forceinline__ void rows_flush_occ(float* acc, float (&gocc_me)[kRowsLP], float g0col, float g00, int lane,
                                               int wave, int L, float* g_occ_unit) {
  constexpr int LP = kRowsLP;
  const int q = lane >> 4, i1 = lane & 15;
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    const float r = rows_sum(gocc_me[j]);
    if (q == 0) acc[wave * LP * LP + (i1 + 1) * LP + j] = r;
  }
  const float r0c = rows_sum(g0col), r00 = rows_sum(g00);
  if (q == 0) acc[wave * LP * LP + (i1 + 1)] = r0c;  // occ[0][i1 + 1]
  if (lane == 0) acc[wave * LP * LP] = r00;          // occ[0][0]
  __syncthreads();
  for (int e = threadIdx.x; e < LP * LP; e += kBlock) {
    const int i = e / LP, j = e % LP;
    if (i < L && j < L)
      atomicAdd(g_occ_unit + i * L + j, (acc[e] + acc[LP * LP + e]) + (acc[2 * LP * LP + e] + acc[3 * LP * LP + e]));
  }
  __syncthreads();
}

// ---- flow_ctx_alpha_bwd for 9 .. 17 layers: same arguments and results as flow_ctx_alpha_bwd_kernel<17>.
// NCP: classes, padded to a multiple of four (20: Cityscapes' 20 and KITTI's 19; 32: the rest).
// The layout filter lives in phase 2 as well: the class probabilities of a pass's four pixels are formed by their
// rows together (lane i1: classes i1 and i1 + 16; maximum and sum by the butterfly), exchanged through 128 floats of
// LDS, and every lane weighs ITS object's class distribution against them -- d / d dist accumulates in the lane's
// registers over all its pixels like grad_occ does (the per-pixel kernels pay a 32-value transpose-reduce per layer).
template <int NCP>
__global__ __launch_bounds__(kBlock, 4) void flow_ctx_alpha_bwd_rows_kernel(
    const float* __restrict__ alpha_lr, const float* __restrict__ input,
    const float* __restrict__ dist, const float* __restrict__ occ, const float* __restrict__ g_a01,
    const float* __restrict__ g_aout, float* __restrict__ g_up, float* __restrict__ g_dist,
    float* __restrict__ g_occ, int T, int Tw, int L, int Nl, int C, int chan_off, int H, int W, int scale,
    int tiles, int tiles_per_block, int groups) {
  const GradOfA01 grad_of_a01(g_a01, g_aout);
  constexpr int LP = kRowsLP;
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const int n_ = blockIdx.x / groups;  // (b, t) with t < Tw
  const int b = n_ / Tw, t = n_ % Tw;
  const int t0 = (blockIdx.x % groups) * tiles_per_block, t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int No = L - 1;
  __shared__ __attribute__((aligned(16))) float tab[4][2][LP][kRowsPitch];
  __shared__ __attribute__((aligned(16))) float prs[4][4][kMaxCls];  // [wave][pixel of the pass][class]
  static_assert(sizeof(tab) >= sizeof(float) * (kRowsLP * kRowsOccPitch + (kRowsLP - 1) * kMaxCls), "staging fits");
  const bool filt = dist != nullptr;
  const int q = lane >> 4, i1 = lane & 15;
  float o_me[LP], o0[LP], gocc_me[LP], g0col = 0.0f, g00 = 0.0f;
  float dist_me[NCP], gd_acc[NCP];
  {
    // the order and this batch entry's class distributions go through the (not yet used) tables into registers
    float* stage = &tab[0][0][0][0];
    float* sdist = stage + kRowsLP * kRowsOccPitch;
    rows_stage_order(stage, occ + ((int64_t)b * T + t) * L * L, L);
    if (filt) dist_stage<LP>(sdist, dist + (int64_t)b * No * Nl, L, Nl);
    __syncthreads();
    rows_order(stage, i1, o_me, o0);
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
      dist_me[c] = filt ? sdist[i1 * kMaxCls + c] : 0.0f;  // row l - 1 = i1 (zeros from class Nl on)
      gd_acc[c] = 0.0f;
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < LP; ++j) gocc_me[j] = 0.0f;
  float(*A)[kRowsPitch] = tab[wave][0];
  float(*G)[kRowsPitch] = tab[wave][1];
  float* pr_row = prs[wave][q];

  for (int tile = t0; tile < t1; ++tile) {
    // (an offset the compiler cannot see through: it would compute the 17 x 3 plane addresses once per kernel and keep
    // them, in ~200 scalar registers it does not have, across the whole loop)
    int z = 0;
    asm volatile("" : "+s"(z));
    const int64_t n = n_ + z;
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HWd;
    {
      // ---- phase 1 (lane = pixel): the upsampled rough alphas and the incoming gradients, coalesced
      const int64_t pc = live ? p : HWd - 1;
      const int y = (int)(pc / Wd), x = (int)(pc - (int64_t)y * Wd);
      const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        const bool real = l < L;
        const int lc = min(l, L - 1);
        A[l][lane] = real ? up_sample(alpha_lr + ((int64_t)n * L + lc) * HW, ut) : 0.0f;
        G[l][lane] = (real && live) ? grad_of_a01(((int64_t)n * L + lc) * HWd + pc) : 0.0f;
      }
    }
    wave_lds_sync();
    // ---- phase 2 (lane = pixel-of-four x object): layout filter, the composite's backward, d / d dist
    const int64_t wp0 = (int64_t)tile * kBlock + wave * kWave;  // first pixel of this wave's 64
    // this lane's two layout logits of a pass's pixel (classes i1 and i1 + 16), loaded ONE PASS AHEAD: a pass that
    // waits for its own loads is a memory round trip long, sixteen times per tile
    const float* lgb = input + (((int64_t)b * T + t + z) * C + chan_off) * HWd;
    const int c1 = i1, c2 = i1 + 16;
    auto logits = [&](int s, float& x1, float& x2) {
      const int64_t pq = min(wp0 + 4 * s + q, HWd - 1);
      x1 = c1 < Nl ? lgb[(int64_t)c1 * HWd + pq] : -INFINITY;
      x2 = c2 < Nl ? lgb[(int64_t)min(c2, Nl - 1) * HWd + pq] : -INFINITY;
    };
    float nx1 = -INFINITY, nx2 = -INFINITY;
    if (filt) logits(0, nx1, nx2);
#ifdef WALDO_ABL_ROWS_NOP2
    if (tiles == -1)
#endif
#pragma unroll 1
    for (int s = 0; s < 16; ++s) {
      const int px = 4 * s + q;
      const float aup_me = A[i1 + 1][px], gv_me = G[i1 + 1][px];
      const float a0 = A[0][px], gv0 = G[0][px];
      float f_me = 1.0f;
      if (filt) {  // uniform
        // class probabilities of this row's pixel (lvd.py:731-735: softmax over the layout logits)
        const float x1 = nx1, x2 = nx2;
        logits(min(s + 1, 15), nx1, nx2);
        const float m = row_all(fmaxf(x1, x2), [](float u, float v) { return fmaxf(u, v); });
        const float e1 = c1 < Nl ? expf(x1 - m) : 0.0f, e2 = c2 < Nl ? expf(x2 - m) : 0.0f;
        const float den = row_all(e1 + e2, [](float u, float v) { return u + v; });
        pr_row[c1] = e1 / den;
        pr_row[c2] = e2 / den;
        wave_lds_sync();
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < NCP; c += 4)
          if (c < Nl) {  // uniform
            const f32x4_o pq4 = *reinterpret_cast<const f32x4_o*>(pr_row + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) d += fabsf(dist_me[c + k] - pq4[k]);
          }
        f_me = 1.0f - d / 2.0f;
      }
      const float a_me = aup_me * f_me;
      float ga_me, ga0, pcol_me, p0;
      rows_composite_bwd(a_me, gv_me, a0, gv0, o_me, o0, i1, L, ga_me, ga0, pcol_me, p0, gocc_me, g0col, g00);
      G[i1 + 1][px] = ga_me * f_me;  // d loss / d (upsampled rough alpha)
      if (i1 == 0) G[0][px] = ga0;
      if (filt && g_dist != nullptr) {
        // d f / d dist[c] = -1/2 sign(dist[c] - pr_c) (classes from Nl on: both are 0 there)
        const float gf = -0.5f * ga_me * aup_me;
#pragma unroll
        for (int c = 0; c < NCP; c += 4)
          if (c < Nl) {
            const f32x4_o pq4 = *reinterpret_cast<const f32x4_o*>(pr_row + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float dd = dist_me[c + k] - pq4[k];
              gd_acc[c + k] += dd == 0.0f ? 0.0f : gf * __builtin_copysignf(1.0f, dd);  // gf sign(dd)
            }
          }
      }
      if (filt) wave_lds_sync();  // the next pass rewrites the probabilities
    }
    wave_lds_sync();
    // ---- phase 3 (lane = pixel): coalesced stores
    if (live) {
#pragma unroll
      for (int l = 0; l < LP; ++l)
        if (l < L) g_up[((int64_t)n * L + l) * HWd + p] = G[l][lane];
    }
    wave_lds_sync();  // the tables are rewritten by the next tile's phase 1
  }
  __syncthreads();
  float* acc = &tab[0][0][0][0];
  if (g_occ != nullptr) rows_flush_occ(acc, gocc_me, g0col, g00, lane, wave, L, g_occ + ((int64_t)b * T + t) * L * L);
  if (filt && g_dist != nullptr) {
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
      const float r = rows_sum(gd_acc[c]);
      if (q == 0) acc[(wave * (LP - 1) + i1) * kMaxCls + c] = r;
    }
    __syncthreads();
    constexpr int kPer = (LP - 1) * kMaxCls;
    for (int e = threadIdx.x; e < kPer; e += kBlock) {
      const int o = e / kMaxCls, c = e % kMaxCls;
      if (o < No && c < Nl && c < NCP)
        atomicAdd(g_dist + ((int64_t)b * No + o) * Nl + c, (acc[e] + acc[kPer + e]) + (acc[2 * kPer + e] + acc[3 * kPer + e]));
    }
  }
}

// ---- flow_ctx_warp_bwd for 9 .. 17 layers: same arguments and results as flow_ctx_warp_bwd_kernel<17>
__global__ __launch_bounds__(kBlock, WALDO_ROWS_WAVES) void flow_ctx_warp_bwd_rows_kernel(
    const float* __restrict__ flow_lr, const float* __restrict__ isobj_lr,
    const float* __restrict__ a01, const int64_t* __restrict__ ctx_ts,
    const int64_t* __restrict__ pred_ts, const float* __restrict__ occ,
    const float* __restrict__ g_flow, const float* __restrict__ g_actx,
    const float* __restrict__ g_dis, float* __restrict__ g_fup, float* __restrict__ g_a01,
    float* __restrict__ g_occ, int T, int Tw, int Tc, int Tp, int L, int H, int W, int scale,
    int tiles, int tiles_per_block, int groups) {
  constexpr int LP = kRowsLP;
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  const int m_ = blockIdx.x / groups;  // (b, tc, tp)
  const int tp = m_ % Tp, b = m_ / (Tc * Tp);
  const int t0 = (blockIdx.x % groups) * tiles_per_block, t1 = min(tiles, t0 + tiles_per_block);
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ __attribute__((aligned(16))) float tab[4][2][LP][kRowsPitch];
  const int m = m_;
  const int ts = __builtin_amdgcn_readfirstlane((int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(Tw - 1)));
  const int tpred = __builtin_amdgcn_readfirstlane((int)min(max(pred_ts[tp], (int64_t)0), (int64_t)(T - 1)));
  const float hw = 0.5f * (float)Wd, hh = 0.5f * (float)Hd;
  const int q = lane >> 4, i1 = lane & 15;
  float o_me[LP], o0[LP], gocc_me[LP], g0col = 0.0f, g00 = 0.0f;
  {
    float* stage = &tab[0][0][0][0];
    rows_stage_order(stage, occ + ((int64_t)b * T + tpred) * L * L, L);
    __syncthreads();
    rows_order(stage, i1, o_me, o0);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < LP; ++j) gocc_me[j] = 0.0f;
  float(*A)[kRowsPitch] = tab[wave][0];
  float(*G)[kRowsPitch] = tab[wave][1];

  for (int tile = t0; tile < t1; ++tile) {
    const int64_t p = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = p < HWd;
    const int64_t pc = live ? p : HWd - 1;
    int amax = 0;
    {
      // (an offset the compiler cannot see through: it would compute the 17 x 5 plane addresses once per kernel and
      // keep them, in ~500 scalar registers it does not have, across the whole loop)
      int z = 0;
      asm volatile("" : "+s"(z));
      const int64_t m = m_ + z;
      const float* ap = a01 + (((int64_t)b * Tw + ts + z) * L) * HWd;
      // ---- phase 1 (lane = pixel): a_l = sample of the context alpha at (pixel + flow_l) times the ghost mask, the
      // disocclusion arg max, and d loss / d v_l
      const int y = (int)(pc / Wd), x = (int)(pc - (int64_t)y * Wd);
      const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
      float gx0, gy0;
      identity_grid(x, y, Wd, Hd, gx0, gy0);
      const float gfx = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2) * HWd + pc] : 0.0f;
      const float gfy = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2 + 1) * HWd + pc] : 0.0f;
      float dis = -INFINITY;
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
        if (real && v > dis) {
          dis = v;
          amax = l;
        }
        float gvl = (real && live) ? fmaf(gfx, fxl, gfy * fyl) : 0.0f;
        if (real && live && g_actx != nullptr) gvl = fmaf(2.0f, g_actx[((int64_t)m * L + lc) * HWd + pc], gvl);
        A[l][lane] = real ? v : 0.0f;
        G[l][lane] = gvl;
        if ((l & (WALDO_ROWS_GROUP - 1)) == WALDO_ROWS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);  // a group of layers' loads in flight, not all seventeen
      }
    }
    wave_lds_sync();
    // ---- phase 2 (lane = pixel-of-four x object)
#ifndef WALDO_ABL_ROWS_NOP2  // timing-only ablation: without the composite's backward (wrong values)
    rows_phase2(A, G, o_me, o0, q, i1, L, gocc_me, g0col, g00);
#endif
    wave_lds_sync();
    {
      int z = 0;
      asm volatile("" : "+s"(z));
      const int64_t m = m_ + z;
      const float* ap = a01 + (((int64_t)b * Tw + ts + z) * L) * HWd;
      // ---- phase 3 (lane = pixel): through the flow composite and through the sample positions (the pixel's
      // geometry is formed again: cheaper than carrying it through phase 2)
      const int y = (int)(pc / Wd), x = (int)(pc - (int64_t)y * Wd);
      const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);
      float gx0, gy0;
      identity_grid(x, y, Wd, Hd, gx0, gy0);
      const float gfx = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2) * HWd + pc] : 0.0f;
      const float gfy = (g_flow != nullptr && live) ? g_flow[((int64_t)m * 2 + 1) * HWd + pc] : 0.0f;
      const float gd = (g_dis != nullptr && live) ? g_dis[(int64_t)m * HWd + pc] : 0.0f;
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        if (l < L) {  // wave-uniform
          const float vjl = A[l][lane];  // the composited alpha v_l
          const float* fl = flow_lr + (((int64_t)m * L + l) * 2) * HW;
          const float fxl = up_sample(fl, ut), fyl = up_sample(fl + HW, ut);
          float ghost = 1.0f;
          if (isobj_lr != nullptr && l >= 1)
            ghost = (up_sample(isobj_lr + ((int64_t)m * (L - 1) + (l - 1)) * HW, ut) > 0.9f) ? 1.0f : 0.0f;
          const Taps t = make_taps(gx0 + fxl, gy0 + fyl, Hd, Wd);
          float ddx, ddy;
          (void)tap_sample_d(ap + (int64_t)l * HWd, t, ddx, ddy);
          const float gs = G[l][lane] + (l == amax ? gd : 0.0f);  // d loss / d a_l
          const float gfxl = fmaf(gs * (ddx * ghost), hw, vjl * gfx);
          const float gfyl = fmaf(gs * (ddy * ghost), hh, vjl * gfy);
          if (live) {
            g_fup[(((int64_t)m * L + l) * 2) * HWd + p] = gfxl;
            g_fup[(((int64_t)m * L + l) * 2 + 1) * HWd + p] = gfyl;
          }
          if (g_a01 != nullptr) {  // d / d a01: bilinear splat of gs * ghost
            const float gsg = live ? gs * ghost : 0.0f;
            if (gsg != 0.0f) {
              float* gp = g_a01 + (((int64_t)b * Tw + ts) * L + l) * HWd;
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
        if ((l & (WALDO_ROWS_GROUP - 1)) == WALDO_ROWS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_sync();  // the tables are rewritten by the next tile's phase 1
  }
  __syncthreads();
  if (g_occ != nullptr)
    rows_flush_occ(&tab[0][0][0][0], gocc_me, g0col, g00, lane, wave, L, g_occ + ((int64_t)b * T + tpred) * L * L);
}

}  // namespace waldo
