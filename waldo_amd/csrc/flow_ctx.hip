// A9: the two full-resolution passes of Warper.grid_to_flow_ctx
// (models/nets/lvd.py:707-828; the unrestricted twin grid_to_flow, :602-705, shares them), fused.
//
// The reference runs the HD part of the method as ~25 elementwise / interpolate / grid_sample /
// softmax / prod launches over (B, Tc, [Tp,] L, [Nl | L,] Hd, Wd) temporaries (2.7 GB for the layout
// filter and 2.4 GB for the L x L occlusion broadcast at B = 1, Tc = 4, Tp = 1, L = 17, 512 x 1024).
// Everything that lives at the LOW resolution (layers warped to the image: alpha, flow, object
// mask; the objects' class distributions) is 1/16 of the pixels and stays with the per-op kernels;
// the two passes over the HD raster are one kernel each, one thread per HD pixel, the L values of
// a pixel in registers:
//
//   flow_ctx_alpha_kernel  (lvd.py:731-766)   x4 bilinear upsampling of the L rough alphas
//       (F.interpolate, align_corners=False) -> layout filter: object o keeps
//       1 - 1/2 sum_n |dist[o][n] - softmax_n(layout logits at the pixel)| of its alpha -> occlusion
//       product a'_j = a_j prod_i (1 - a_i occ[i][j]) -> a' (kept in [0,1] for the second pass) and
//       2a' - 1 (the method's `alpha` output).
//   flow_ctx_warp_kernel   (lvd.py:784-818)   per (b, tc, tp): upsampled per-layer flow ->
//       alpha of context frame ctx_ts[b,tc,tp] sampled at (pixel + flow_l) (bilinear, zeros) ->
//       ghost mask (upsampled warped ones > 0.9) -> disocclusion = max_l -> occlusion product with
//       the predicted frame's order -> flow = sum_l a'_l flow_l; writes flow, 2a' - 1, disocc.
//
// Both are HBM streaming passes: A reads Nl + (L taps from the 16x smaller LR planes) and writes
// 2L floats per HD pixel; B reads ~L gathered alphas and writes L + 3.
#include <type_traits>

#include "flow_ctx_common.hip.h"

namespace waldo {

#ifndef WALDO_FCW_SPARSE
#define WALDO_FCW_SPARSE 1  // wave-uniform skipping of absent layers (flow_ctx_warp_kernel); 0: every layer, every factor
#endif
#ifndef WALDO_FCW_MASK_FIRST
#define WALDO_FCW_MASK_FIRST 1  // 0: whole records of every layer; 1: a layer's mask first, its flow record if wanted;
                                // 2: the masks of ALL layers up front (one LDS round trip), records of wanted layers.
                                // C5 pipeline, A/B on one box (tools_dev/ab_pipeline.sh): 7.67 / 7.55 / 7.80 ms per step
#endif
#ifndef WALDO_FCW_CONST_OUT
#define WALDO_FCW_CONST_OUT 0   // 1: outputs of layers outside the active set as constants, behind a wave-uniform branch
                                // -- measured SLOWER (8.1 against 7.6 ms: twelve more branches cut the store stream up)
#endif
template <int LP, int NCP>
__global__ __launch_bounds__(kBlock) void flow_ctx_alpha_kernel(
    const float* __restrict__ alpha_lr, const float* __restrict__ input,
    const float* __restrict__ dist, const float* __restrict__ occ, float* __restrict__ a01,
    float* __restrict__ alpha_out, unsigned* __restrict__ layer_bits, int T, int Tw, int L, int Nl, int C, int chan_off,
    int H, int W, int scale, int units, int tiles, int nbands) {
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  int n, x, y;  // n = (b, t) with t < Tw
  if (!hd_pixel(units, Hd, Wd, tiles, nbands, n, x, y)) return;
  const int b = n / Tw, t = n % Tw;
  const int64_t p = (int64_t)y * Wd + x;
  // the objects' class distributions of this batch entry: broadcast reads from LDS
  __shared__ __attribute__((aligned(16))) float sdist[(LP - 1) * kMaxCls];
  __shared__ __attribute__((aligned(16))) float occm[OccLds<LP>::kFloats];
  bool tab_bad = false;
  if (dist != nullptr) tab_bad = dist_stage<LP>(sdist, dist + (int64_t)b * (L - 1) * Nl, L, Nl);
  tab_bad |= occ_stage<LP>(occm, occ + ((int64_t)b * T + t) * L * L, L);
  // (the barrier doubles as the vote on non-finite entries of the two tables: see the short cuts below)
  const bool dense = WALDO_FCW_SPARSE ? __syncthreads_or(tab_bad) != 0 : (__syncthreads(), true);
  if (x >= Wd || y >= Hd) return;
  const UpTaps ut = up_taps(y, x, 1.0f / (float)scale, H, W);

  float a[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l)
    a[l] = (l < L) ? up_sample(alpha_lr + ((int64_t)n * L + min(l, L - 1)) * HW, ut) : 0.0f;
  // The wavefront's ACTIVE layers: those whose upsampled alpha is non-zero in some lane (an object's rough alpha is
  // exactly 0 outside its canvas: grid_sample's zeros padding, lvd.py:727).  A layer outside the set keeps alpha 0
  // through the filter (0 * weight) and the product (factor 1 - 0 * occ = 1, result 0 * product): its filter weight
  // and its row and column of the product are skipped.  Exact while the operands are finite; a non-finite alpha in
  // any lane, non-finite layout logits in any lane or a non-finite entry of the order / the class distributions
  // (`dense`) switches back to every layer.
  unsigned active = 0;
  bool wild = false;
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    if (__ballot(a[l] != 0.0f) != 0ull) active |= 1u << l;
    wild |= __ballot(!(fabsf(a[l]) <= 3.0e38f)) != 0ull;
  }
  if (!WALDO_FCW_SPARSE || dense || wild) active = LP >= 32 ? 0xffffffffu : (1u << LP) - 1u;

  if (dist != nullptr) {
    // softmax over the Nl layout logits of this pixel (held in registers)
    const float* lg = input + (((int64_t)b * T + t) * C + chan_off) * HWd + p;
    float pr[NCP];
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
    // (non-finite logits make every filter weight NaN, and 0 * NaN is NaN: no short cuts then)
    if (__ballot(!(den >= 1.0f && den <= 3.0e38f)) != 0ull) active = LP >= 32 ? 0xffffffffu : (1u << LP) - 1u;
#pragma unroll
    for (int l = 1; l < LP; ++l)
      if (active & (1u << l))  // wave-uniform
        a[l] *= 1.0f - dist_l1(sdist + (l - 1) * kMaxCls, pr, Nl) / 2.0f;  // padding: 0 stays 0
  }

  // padding layers carry alpha 0 (factor exactly 1); branch-free so that the arrays stay in registers.
  // Four columns of the order per step (OccLds), two and two on the packed-fp32 pipe.
  typedef float f32x2_w __attribute__((ext_vector_type(2)));
  unsigned nz = 0;  // (layer_bits) bit l: a01 of layer l is non-zero (or NaN) in some pixel of this wavefront's row segment
#pragma unroll
  for (int j = 0; j < LP; j += 4) {
    f32x2_w prd[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
    if ((active >> j) & 0xfu) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        if (active & (1u << i)) {  // wave-uniform
          // (a REAL branch: left alone hipcc if-converts the four packed operations into selects of the factor 1 and
          // evaluates all L x L factors again; an asm statement cannot be executed speculatively)
          asm volatile("");
          const f32x4_o o = occ_quad<LP, false>(occm, i, j);
          const f32x2_w ai = {a[i], a[i]};
          const f32x2_w one = {1.0f, 1.0f};  // 1 - a o in one rounding (v_pk_fma_f32)
          prd[0] = prd[0] * __builtin_elementwise_fma(-ai, (f32x2_w){o[0], o[1]}, one);
          if (j + 2 < LP) prd[1] = prd[1] * __builtin_elementwise_fma(-ai, (f32x2_w){o[2], o[3]}, one);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (j + k >= LP) break;
      const float v = a[j + k] * prd[k >> 1][k & 1];
      if (j + k < L) {
        a01[((int64_t)n * L + j + k) * HWd + p] = v;
        if (alpha_out != nullptr) alpha_out[((int64_t)n * L + j + k) * HWd + p] = v * 2.0f - 1.0f;
        if (layer_bits != nullptr && __ballot(v != 0.0f) != 0ull) nz |= 1u << (j + k);  // (NaN != 0: counts)
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // a quad of columns at a time (bounds the registers)
  }
  // by-product for the second pass on the path WITHOUT a ghost mask (grid_to_flow, lvd.py:602-705): which layers are
  // present at all in this 64-pixel row segment -- one word per (frame, row, segment); flow_ctx_warp_kernel ORs the words
  // its tile's samples can reach and skips the layers that are absent from all of them
  if (layer_bits != nullptr && (threadIdx.x & (kWave - 1)) == 0)
    layer_bits[((int64_t)n * Hd + y) * ((Wd + kHdCols - 1) / kHdCols) + (x / kHdCols)] = nz;
}

// Layout of the staged low-resolution data of one tile (flow_ctx_warp): CELL-major, one 16-byte record per
// layer -- {flow x, flow y, object mask of the layer below the ghost test, unused} -- and the records of
// the padding layers filled with those of layer L - 1.  A tap of a layer is then ONE ds_read_b128 at a
// compile-time offset from the tap's cell (plane-major, as the planes lie in memory, it was three
// ds_read_b32 with a run-time plane offset each: 36 samples x 13 VALU + 4 LDS instructions per pixel).
#ifndef WALDO_FCW_WAVES
#define WALDO_FCW_WAVES 5  // waves per SIMD the tall-tile kernel is compiled for up to 12 layers (80 VGPRs, no spill, since the
                           // unstaged path is compiled out of it and the flows are re-taken: 6.77 -> 5.49 ms per C5 pipeline
                           // step, 3.45 -> 3.01 at C4; at 6, which 24 KB of LDS allow, the same 5.5 ms:
                           // profiles/r04_ab_flow_ctx_warp_reflow_*.txt)
#endif
#ifndef WALDO_FCW_TP_INNER
#define WALDO_FCW_TP_INNER 0  // the Tp units of a (clip, context) innermost in an XCD's tile walk (hd_pixel_rows_grouped)
#endif
#ifndef WALDO_FCW_REFLOW
#define WALDO_FCW_REFLOW 1  // the upsampled flow of a layer is taken AGAIN where its composited alpha is known (active layers only)
                            // instead of kept per layer: 2 LP registers fewer across the occlusion product
#endif
#ifndef WALDO_FCW_ROWS
#define WALDO_FCW_ROWS 4  // pixels per thread of flow_ctx_warp_kernel at scale >= 2 (tile = 4 WALDO_FCW_ROWS x 64 pixels)
#endif
constexpr int kFcwRows = WALDO_FCW_ROWS;
#ifndef WALDO_FCW_CHUNK
#define WALDO_FCW_CHUNK 4  // 4: 116 registers at L = 12 (four waves per SIMD) and 1.92 ms at the C5 size; 6: 140 and 2.04 ms
#endif
// floats of LDS for the staged low-resolution records of a tile with `rows` pixels per thread: 136 cells (4 x 64 tile
// at x 2) up to L = 12; L <= 8: 204 cells (8 x 64 tile at x 2).  The 16 x 64 tile at 9-12 layers only ever fits at
// x 4 and up (6 x 18 cells): sized for exactly that, 24 KB with the order, six workgroups per CU
#ifndef WALDO_FCW_SMALL_LDS
#define WALDO_FCW_SMALL_LDS 1
#endif
constexpr int fcw_cap(int lp, int rows) {
  return lp <= 8 ? 7680 : (lp <= 12 ? ((WALDO_FCW_SMALL_LDS && rows == kFcwRows && kFcwRows == 4) ? 5632 : 7168) : 8192);
}
template <int LP, int R>
struct FcwLds {
  static constexpr int kCell = 4 * LP + 4;  // floats per cell (+ 4: cells of a row on different banks)
  static constexpr int kCap = fcw_cap(LP, R);
};

// where the L planes of alpha_ctx[b, tc, tp] go: element strides of the three unit indices from `alpha_ctx` (the
// planes of one unit are always Hd * Wd apart).  Contiguous (M, L, Hd, Wd): (Tc Tp L, Tp L, L) Hd Wd; inside the
// `raw` tensor of Warper.input_to_output, (B, Tp, Tc', C + L, Hd, Wd), behind the C frame channels of every
// context: base raw + C Hd Wd, strides (Tp Tc' (C + L), C + L, Tc' (C + L)) Hd Wd.
struct ActxLayout {
  int64_t sb, stc, stp;
};

// torch.max / amax return NaN when any element is NaN (lvd.py:803 `alpha_ctx.max(dim=3)[0]`, synthesizer.py:447);
// v_max_f32 returns the other operand.  llvm.maximum = IEEE 754-2019 maximum: v_maximum3_f32 on gfx950, two
// layers per instruction.
__device__ __forceinline__ float nan_max(float a, float b) { return __builtin_elementwise_maximum(a, b); }

// SCORE: also write score[m] = sum_l (alpha_ctx_l + 1) / 2, summed as frame_warp_fuse sums it from the stored
// values (lvd.py:841) -- the frame warp then reads ONE plane per context instead of L.
// R: pixels per thread, kHdRows rows apart (tall tiles, hd_pixel_rows): the tile's staging -- the order (2 L^2 LDS
// entries), the low-resolution patch (one record per cell and layer: three global loads each) and the barrier -- was
// paid per 256 pixels; timing-only ablations put everything but the gathers, the product and the stores at 5 of the
// kernel's 7.8 ms per C5 pipeline step.  With R = 4 a 16 x 64 tile stages 6 x 18 cells where four 4 x 64 tiles staged
// 4 x (3 x 18).
#ifndef WALDO_FCW_COMPACT_CHUNK
#define WALDO_FCW_COMPACT_CHUNK 4  // slots whose loads are in flight together
#endif
#ifndef WALDO_FCW_COMPACT
#define WALDO_FCW_COMPACT 0  // > 0 (variant builds only): slots of the per-TILE compact layer list, round 6's bounded experiment --
                             // bit-identical and 8-13 % SLOWER (tools_dev/dropped/flow_ctx_warp_compact.hip.h); 0: not compiled
#endif
template <int LP, bool SCORE, int R, int KS = 0>
__global__ __launch_bounds__(kBlock, (R > 1 && LP <= 12) ? WALDO_FCW_WAVES : ((R > 1 && LP <= 17) ? 4 : 1)) void flow_ctx_warp_kernel(
    const float* __restrict__ flow_lr, const float* __restrict__ isobj_lr,
    const float* __restrict__ a01, const int64_t* __restrict__ ctx_ts,
    const int64_t* __restrict__ pred_ts, const float* __restrict__ occ, float* __restrict__ flow,
    float* __restrict__ alpha_ctx, ActxLayout lay, float* __restrict__ score, float* __restrict__ disocc,
    float* __restrict__ alpha_max, const unsigned* __restrict__ layer_bits, int* __restrict__ status, int T, int Tw, int Tc,
    int Tp, int L, int H, int W, int scale, int units, int tiles, int nbands) {
  using G = FcwLds<LP, R>;
  typedef float f32x2_w __attribute__((ext_vector_type(2)));
  const int Hd = H * scale, Wd = W * scale;
  const int64_t HWd = (int64_t)Hd * Wd, HW = (int64_t)H * W;
  int m, x, y_first;  // m = (b, tc, tp)
#if WALDO_FCW_TP_INNER
  if (!hd_pixel_rows_grouped<R>(units / Tp, Tp, Hd, Wd, tiles, nbands, m, x, y_first)) return;
#else
  if (!hd_pixel_rows<R>(units, Hd, Wd, tiles, nbands, m, x, y_first)) return;
#endif
  const int tp = m % Tp, b = m / (Tc * Tp);
  const float rscale = 1.0f / (float)scale;
  // frame of the context alpha (clamped: the index comes from device memory; an index outside the window is reported
  // in `status`: checked_frame) and of the order
  // (read through the vector path, the frame indices land in VGPRs and every plane address derived from
  // them becomes per-lane 64-bit arithmetic: they are wave-uniform, say so)
  const int ts = __builtin_amdgcn_readfirstlane(checked_frame(ctx_ts, m, Tw, status, kStatusCtx));
  const int tpred = __builtin_amdgcn_readfirstlane(checked_frame(pred_ts, tp, T, status, kStatusPred));

  __shared__ __attribute__((aligned(16))) float lrimg[G::kCap];
  // the order of the predicted frame from LDS (OccLds): at L = 12 the scalar loads made the kernel issue as
  // many scalar as vector instructions (1480 / 1464 per wavefront)
  __shared__ __attribute__((aligned(16))) float occm[OccLds<LP>::kFloats];
  const bool occ_bad = occ_stage<LP>(occm, occ + ((int64_t)b * T + tpred) * L * L, L);
  // ---- the tile's patch of the low-resolution planes (2 L flow planes, L - 1 object masks)
  const int nob = isobj_lr != nullptr ? L - 1 : 0;
  LrPatch lq = lr_patch(y_first - (int)(threadIdx.x >> 6), x - (int)(threadIdx.x & (kWave - 1)), Hd, Wd, rscale, H, W,
                        kHdRows * R);
  lq.r_lo = __builtin_amdgcn_readfirstlane(lq.r_lo);  // the same in every thread of the workgroup
  lq.c_lo = __builtin_amdgcn_readfirstlane(lq.c_lo);
  lq.nrows = __builtin_amdgcn_readfirstlane(lq.nrows);
  lq.ncols = __builtin_amdgcn_readfirstlane(lq.ncols);
  const int area = lq.nrows * lq.ncols;
  // (uniform.  A tall tile is only launched where every tile's patch fits -- `fits` in flow_ctx_warp_launch bounds
  // the patch of any tile -- so that R > 1 compiles WITHOUT the unstaged path: left in, its per-layer plane addresses
  // were hoisted out of the row loop as 2 LP 64-bit registers and spilled there)
  const bool fits_lds = area <= kBlock && area * G::kCell <= G::kCap;
  if (R > 1 && !fits_lds) return;  // (never taken: see the launcher)
  const bool staged = R > 1 ? true : fits_lds;
  bool flow_bad = false;
  // (layer_bits) the range of the tile's low-resolution flows over all layers: every pixel's upsampled flow is a convex
  // combination of four of these cells
  float fx_lo = INFINITY, fx_hi = -INFINITY, fy_lo = INFINITY, fy_hi = -INFINITY;
  __shared__ float wave_box[kBlock / kWave][4];
  __shared__ unsigned wave_seen[kBlock / kWave];
#if WALDO_FCW_COMPACT
  unsigned mine = 0;  // bit l: this thread staged a cell of layer l whose object mask may pass the ghost test
  __shared__ unsigned wave_bits[kBlock / kWave];
#endif
  if (staged) {
    // thread = (cell, layer group): kBlock / area groups share the layers of a cell
    const int ngrp = kBlock / area;
    const int t = (int)threadIdx.x;
    // t < 256, area / ncols <= 256, the + 0.5: the approximate reciprocal gives the exact quotients
    const int grp = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)area));
    const int cell = t - grp * area;
    const int r = (int)(((float)cell + 0.5f) * __builtin_amdgcn_rcpf((float)lq.ncols));
    const int64_t off = (int64_t)(lq.r_lo + r) * W + lq.c_lo + (cell - r * lq.ncols);
    if (grp < ngrp)
      for (int l = grp; l < LP; l += ngrp) {
        const int lc = min(l, L - 1);
        const float* fl = flow_lr + (((int64_t)m * L + lc) * 2) * HW + off;
        f32x4 rec = {fl[0], fl[HW], 0.0f, 0.0f};
        if (nob && lc >= 1) rec[2] = isobj_lr[((int64_t)m * (L - 1) + (lc - 1)) * HW + off];
#if WALDO_FCW_COMPACT
        // (a pixel's mask is a convex combination of four cells, rounded: 0.8999 keeps a margin below the test's 0.9;
        // a NaN cell counts as present)
        if (KS > 0 && l < L && !(rec[2] <= 0.8999f)) mine |= 1u << l;
#endif
        flow_bad |= !(fabsf(rec[0]) <= 3.0e38f) | !(fabsf(rec[1]) <= 3.0e38f);
        if (layer_bits != nullptr) {  // (uniform)
          fx_lo = fminf(fx_lo, rec[0]), fx_hi = fmaxf(fx_hi, rec[0]);
          fy_lo = fminf(fy_lo, rec[1]), fy_hi = fmaxf(fy_hi, rec[1]);
        }
        *reinterpret_cast<f32x4*>(lrimg + cell * G::kCell + 4 * l) = rec;
      }
  }
  if (layer_bits != nullptr && staged) {  // (uniform) the wavefront's range, one row of wave_box per wavefront
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      fx_lo = fminf(fx_lo, __shfl_xor(fx_lo, d, kWave)), fx_hi = fmaxf(fx_hi, __shfl_xor(fx_hi, d, kWave));
      fy_lo = fminf(fy_lo, __shfl_xor(fy_lo, d, kWave)), fy_hi = fmaxf(fy_hi, __shfl_xor(fy_hi, d, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
      float* wb = wave_box[threadIdx.x >> 6];
      wb[0] = fx_lo, wb[1] = fx_hi, wb[2] = fy_lo, wb[3] = fy_hi;
    }
  }
#if WALDO_FCW_COMPACT
  if (KS > 0) {  // the wavefront's OR of `mine`, one word per wavefront (read behind the barrier below)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine |= (unsigned)__shfl_xor((int)mine, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) wave_bits[threadIdx.x >> 6] = mine;
  }
#endif
  // (the barrier doubles as the vote: a non-finite entry anywhere in the order or in the tile's low-resolution flows
  // switches the skipping below off; a tile whose patch is not staged is not examined: dense)
  const bool dense = WALDO_FCW_SPARSE ? (__syncthreads_or(occ_bad | flow_bad | !staged) != 0) : (__syncthreads(), true);
  // ---- the layers PRESENT around this tile's samples (no ghost mask: Warper.grid_to_flow, lvd.py:602-705).  The
  // first pass left one word per (frame, row, 64-pixel segment) of the context frame's composited alphas: bit l = layer
  // l is non-zero somewhere in the segment (`layer_bits`).  A sample of this tile lands at pixel + flow with the flow
  // inside the range of the staged cells (a convex combination; two pixels of margin for its rounding and for the
  // bilinear footprint), so a layer that is absent from every segment the box [tile + range] touches samples four zero
  // taps in every pixel: value exactly 0, as if it had been sampled -- its upsampling, taps, gathers and its row and
  // column of the product are skipped like a layer behind the ghost mask.  One more barrier per tile.  Not examined
  // (every layer present): tiles that are not staged, a non-finite flow / order entry (`dense`), boxes of more than
  // 2 * kBlock words.
  unsigned present = 0xffffffffu;
  if (layer_bits != nullptr && staged) {  // (uniform)
    unsigned seen = 0xffffffffu;
    if (!dense) {
      const float bx_lo = fminf(fminf(wave_box[0][0], wave_box[1][0]), fminf(wave_box[2][0], wave_box[3][0]));
      const float bx_hi = fmaxf(fmaxf(wave_box[0][1], wave_box[1][1]), fmaxf(wave_box[2][1], wave_box[3][1]));
      const float by_lo = fminf(fminf(wave_box[0][2], wave_box[1][2]), fminf(wave_box[2][2], wave_box[3][2]));
      const float by_hi = fmaxf(fmaxf(wave_box[0][3], wave_box[1][3]), fmaxf(wave_box[2][3], wave_box[3][3]));
      // grid units -> pixels: (Wd / 2) per unit; the tile's pixels [tx0, tx0 + 63] x [ty0, ty0 + 4 R - 1]
      const int tx0 = x - (int)(threadIdx.x & (kWave - 1)), ty0 = y_first - (int)(threadIdx.x >> 6);
      const float hx = 0.5f * (float)Wd, hy = 0.5f * (float)Hd;
      // (clamped in float first: a wild flow must not overflow the conversion)
      const int x0 = (int)fmaxf(fminf(floorf(bx_lo * hx) + (float)(tx0 - 2), (float)Wd), -1.0f);
      const int x1 = (int)fmaxf(fminf(ceilf(bx_hi * hx) + (float)(tx0 + kHdCols + 1), (float)Wd), -1.0f);
      const int y0 = (int)fmaxf(fminf(floorf(by_lo * hy) + (float)(ty0 - 2), (float)Hd), -1.0f);
      const int y1 = (int)fmaxf(fminf(ceilf(by_hi * hy) + (float)(ty0 + kHdRows * R + 1), (float)Hd), -1.0f);
      const int cx0 = max(x0, 0), cx1 = min(x1, Wd - 1), cy0 = max(y0, 0), cy1 = min(y1, Hd - 1);
      const int nseg = (Wd + kHdCols - 1) / kHdCols;
      const int s0 = cx0 / kHdCols, ns = cx1 >= cx0 ? cx1 / kHdCols - s0 + 1 : 0, nr = cy1 >= cy0 ? cy1 - cy0 + 1 : 0;
      const int words = ns * nr;  // (uniform: every input is)
      if (words <= 2 * kBlock) {
        seen = 0;
        const unsigned* lb = layer_bits + ((int64_t)b * Tw + ts) * Hd * nseg;
        for (int e = (int)threadIdx.x; e < words; e += kBlock) {
          const int r = e / ns, c = e - r * ns;
          seen |= lb[(int64_t)(cy0 + r) * nseg + s0 + c];
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) seen |= (unsigned)__shfl_xor((int)seen, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) wave_seen[threadIdx.x >> 6] = seen;
    __syncthreads();
    present = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_seen[0] | wave_seen[1] | wave_seen[2] | wave_seen[3]));
  }
  int rr_first = 0;
#if WALDO_FCW_COMPACT  // (variant builds only: round 6's rejected experiment, tools_dev/dropped/)
#include "flow_ctx_warp_compact.hip.h"
#else
  if (x >= Wd) return;
#endif
#pragma unroll 1
  for (int rr = rr_first; rr < R; ++rr) {
  const int y = y_first + kHdRows * rr;
  if (y >= Hd) break;
  const int64_t p = (int64_t)y * Wd + x;
  const UpTaps ut = up_taps(y, x, rscale, H, W);
  LrTaps lt = lr_taps(y, x, rscale, H, W, lq);
  lt.o00 *= G::kCell, lt.o01 *= G::kCell, lt.o10 *= G::kCell, lt.o11 *= G::kCell;
  float gx0, gy0;  // texel centre of the HD identity grid exactly as get_grid() builds it
  identity_grid(x, y, Wd, Hd, gx0, gy0);

  // branch-free over the padded layer count (a padding layer repeats layer L-1 and its alpha is zeroed):
  // conditional writes to the per-layer arrays would keep them out of registers.  In chunks of up to six
  // layers: the flows, taps and LOADS of the chunk first (twelve eight-byte loads in flight per lane), then
  // its values -- taken two layers at a time (round 2) a wavefront waits L / 2 times for memory, and at
  // three waves per SIMD that wait is what the kernel's time was made of.  Two copies of the loop, one per
  // source of the low-resolution taps.
  // SPARSITY (round 4).  Objects are small: in a wavefront's 64-pixel row segment most layers are absent -- their
  // ghost mask is below the threshold in every lane, or their sampled alpha is 0 in every lane.  Two wave-uniform
  // short cuts, both exact:
  //  * a layer l >= 1 whose upsampled object mask is <= 0.9 in EVERY lane has alpha 0 whatever it samples
  //    (lvd.py:785-802: `alpha_ctx * is_obj`): its taps, its two 8-byte gathers and its bilinear value are skipped;
  //    neither is its flow upsampled: it enters the result as 0 * flow;
  //  * in the occlusion product a layer with alpha == 0 in every lane contributes the factor 1 - 0 * occ = 1 to
  //    every column and its own column's result is 0 * product = 0: rows and columns outside the wavefront's
  //    ACTIVE set are skipped (k^2 instead of L^2 factor evaluations, k ~ 2-4 of 12), and such a layer's outputs
  //    are the constants 2 * 0 - 1 = -1 (alpha_ctx), + 0 (score, flow).
  // "Exact" needs finite operands: 0 * inf would have been NaN.  A non-finite entry of the order or of the tile's
  // low-resolution flows (`dense`, voted at the barrier above) or a non-finite sampled alpha in any lane (`wild`,
  // below) switches everything back to all L layers and all L x L factors, so NaNs propagate exactly as before.
  // (the tall tiles only: at one pixel per thread the low-resolution taps may come from memory -- the LVD recipe, where
  // every layer is active -- and taking them twice costs 34 more loads per pixel: 55 -> 71 us per call there)
  constexpr bool kReflow = WALDO_FCW_REFLOW && R > 1;
  float a[LP], fx[kReflow ? 1 : LP], fy[kReflow ? 1 : LP];
  float dis = -INFINITY;
  const float* ap0 = a01 + (((int64_t)b * Tw + ts) * L) * HWd;  // plane of layer l: + min(l, L - 1) * HWd
  unsigned active = 0;  // wave-uniform: bit l = some lane has a[l] != 0
  bool wild = false;    // some lane sampled a non-finite alpha
  // the object masks of all layers up front (staged tiles): 4 (L - 1) four-byte LDS reads in flight together, one
  // wait; bit l of keep_bits (per lane) = layer l is visible here, bit l of want_bits (wave-uniform) = in some lane
  unsigned keep_bits = 0xffffffffu, want_bits = 0xffffffffu;
  if (WALDO_FCW_MASK_FIRST == 2 && staged && nob) {
    float g[LP];
#pragma unroll
    for (int l = 1; l < LP; ++l)
      g[l] = up_blend(ut, lrimg[lt.o00 + 4 * l + 2], lrimg[lt.o01 + 4 * l + 2], lrimg[lt.o10 + 4 * l + 2],
                      lrimg[lt.o11 + 4 * l + 2]);
#pragma unroll
    for (int l = 1; l < LP; ++l) {
      const bool kp = g[l] > 0.9f;
      if (!kp) keep_bits &= ~(1u << l);
      if (__ballot(kp) == 0ull) want_bits &= ~(1u << l);
    }
  }
  auto layers = [&](auto from_lds) {
    constexpr bool LDS = decltype(from_lds)::value;
    constexpr int CH = LP < WALDO_FCW_CHUNK ? LP : WALDO_FCW_CHUNK;  // layers whose loads are in flight together
#pragma unroll
    for (int l0 = 0; l0 < LP; l0 += CH) {
      PairTaps pt[CH];
      f32x2_p ra[CH], rb[CH];
      bool inter[CH], keep[CH];
      unsigned need = 0;  // wave-uniform: bit k = layer l0 + k is sampled
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int l = l0 + k;
        if (l >= LP) break;
        float fxl = 0.0f, fyl = 0.0f, g = 1.0f;
        const bool masked = l >= 1 && nob;  // uniform: this layer has an object mask
        bool want;
        if (WALDO_FCW_MASK_FIRST == 2 && LDS) {
          keep[k] = (keep_bits >> l) & 1u;
          want = l < L && (WALDO_FCW_SPARSE ? ((want_bits & present) >> l & 1u) != 0 : true);
        } else {
          if (LDS) {
            // the mask alone first (four 4-byte reads): most layers stop here
            if (masked)
              g = up_blend(ut, lrimg[lt.o00 + 4 * l + 2], lrimg[lt.o01 + 4 * l + 2], lrimg[lt.o10 + 4 * l + 2],
                           lrimg[lt.o11 + 4 * l + 2]);
          } else if (masked) {
            g = up_sample(isobj_lr + ((int64_t)m * (L - 1) + max(min(l, L - 1) - 1, 0)) * HW, ut);
          }
          keep[k] = !(masked && !(g > 0.9f));
          // a padding layer (l >= L) is never sampled: its alpha is 0 by definition; nor is a layer that is absent from
          // every segment this tile's samples can reach (`present`)
          want = l < L && (WALDO_FCW_SPARSE ? (__ballot(keep[k]) != 0ull && ((present >> l) & 1u) != 0) : true);
        }
        if (want || ((dense || WALDO_FCW_MASK_FIRST == 0) && l < L)) {  // (dense: the flow of every layer, its product with alpha 0 may be NaN)
          if (LDS) {
            const f32x2_p v00 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o00 + 4 * l);
            const f32x2_p v01 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o01 + 4 * l);
            const f32x2_p v10 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o10 + 4 * l);
            const f32x2_p v11 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o11 + 4 * l);
            fxl = up_blend(ut, v00[0], v01[0], v10[0], v11[0]);
            fyl = up_blend(ut, v00[1], v01[1], v10[1], v11[1]);
          } else {
            const float* fl = flow_lr + (((int64_t)m * L + min(l, L - 1)) * 2) * HW;
            fxl = up_sample(fl, ut);
            fyl = up_sample(fl + HW, ut);
          }
        }
        if (!kReflow) fx[l] = fxl, fy[l] = fyl;
        if (want) {
          need |= 1u << k;
          pt[k] = pair_taps(gx0 + fxl, gy0 + fyl, Hd, Wd, inter[k]);
#ifdef WALDO_ABL_FCW_NOGATHER  // timing-only ablation: one coalesced load instead of the taps
          ra[k] = rb[k] = (f32x2_p){ap0[p], fxl};
#else
          pair_load(ap0 + (int64_t)min(l, L - 1) * HWd, pt[k], ra[k], rb[k]);
#endif
        }
        if (k & 1) __builtin_amdgcn_sched_barrier(0);  // the LDS records of two layers at a time (32 registers)
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int l = l0 + k;
        if (l >= LP) break;
        float v = 0.0f;
        if (need & (1u << k)) {
          v = pair_value(ra[k], rb[k], pt[k], inter[k]);
          v = keep[k] ? v : 0.0f;
          // the value HERE: the wave-uniform border branches cut the loop body into basic blocks, and the
          // compiler sinks this arithmetic to the first use of a[] after the loop -- keeping the taps and loaded
          // pairs of EVERY layer alive to the end (216 registers at L = 12)
          asm volatile("" : "+v"(v));
          if (__ballot(v != 0.0f) != 0ull) active |= 1u << l;
          wild |= __ballot(!(fabsf(v) <= 3.0e38f)) != 0ull;
        }
        if (l < L) dis = nan_max(dis, v);
        a[l] = v;
      }
      __builtin_amdgcn_sched_barrier(0);  // one chunk's loads at a time
    }
  };
  if (R > 1 || staged) layers(std::true_type{});
  else layers(std::false_type{});
  if (!WALDO_FCW_SPARSE || dense || wild) active = L >= 32 ? 0xffffffffu : (1u << L) - 1u;
  disocc[(int64_t)m * HWd + p] = dis;
  float ox = 0.0f, oy = 0.0f;
  float amax = -INFINITY;  // max over the layers of the composited alpha (Synthesizer.predict's disocclusion test)
  float* ac = alpha_ctx + b * lay.sb + ((m / Tp) % Tc) * lay.stc + tp * lay.stp;
  float ssum = 0.0f;
  // four columns j of the order per step, two and two on the packed-fp32 pipe (the product of every column
  // runs over i in the same order as in the other kernels of the path); rows and column quads outside the active
  // set are skipped (see above)
#pragma unroll
  for (int j = 0; j < LP; j += 4) {
    f32x2_w prd[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
#ifdef WALDO_ABL_FCW_NOOCC  // timing-only ablation: without the L x L products
    prd[0][0] = occm[j];
#else
    if ((active >> j) & 0xfu) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        if (active & (1u << i)) {  // wave-uniform
          // (a REAL branch: left alone hipcc if-converts the four packed operations into selects of the factor 1 and
          // evaluates all L x L factors again; an asm statement cannot be executed speculatively)
          asm volatile("");
          const f32x4_o o = occ_quad<LP, false>(occm, i, j);
          const f32x2_w ai = {a[i], a[i]};
          // 1 - a o in one rounding (v_pk_fma_f32): four packed operations per row instead of six
          const f32x2_w one = {1.0f, 1.0f};
          prd[0] = prd[0] * __builtin_elementwise_fma(-ai, (f32x2_w){o[0], o[1]}, one);
          if (j + 2 < LP) prd[1] = prd[1] * __builtin_elementwise_fma(-ai, (f32x2_w){o[2], o[3]}, one);
        }
      }
    }
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (j + k >= LP) break;
      if (!(WALDO_FCW_CONST_OUT || kReflow) || (active & (1u << (j + k)))) {  // wave-uniform
        const float v = a[j + k] * prd[k >> 1][k & 1];
        if (kReflow) {
          // the layer's upsampled flow again (the same expressions as in the sampling loop: the same bits), for the
          // 2-4 layers present in this wavefront's pixels; every layer when the tile is dense / wild
          asm volatile("");  // (a real branch, as above)
          float fxl, fyl;
          if (R > 1 || staged) {
            const int lo = 4 * (j + k);
            const f32x2_p v00 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o00 + lo);
            const f32x2_p v01 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o01 + lo);
            const f32x2_p v10 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o10 + lo);
            const f32x2_p v11 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o11 + lo);
            fxl = up_blend(ut, v00[0], v01[0], v10[0], v11[0]);
            fyl = up_blend(ut, v00[1], v01[1], v10[1], v11[1]);
          } else {
            const float* fl = flow_lr + (((int64_t)m * L + min(j + k, L - 1)) * 2) * HW;
            fxl = up_sample(fl, ut);
            fyl = up_sample(fl + HW, ut);
          }
          ox += v * fxl;
          oy += v * fyl;
        } else {
          ox += v * fx[j + k];
          oy += v * fy[j + k];
        }
#ifndef WALDO_ABL_FCW_NOSTORE
        if (j + k < L) {
          const float av = v * 2.0f - 1.0f;
          ac[p] = av;
          amax = nan_max(amax, av);
          if (SCORE) ssum += (av + 1.0f) / 2.0f;
        }
#endif
      } else if (j + k < L) {
        // alpha 0 in every lane (all operands finite): 2 * 0 - 1, + 0 to the score and to the flow
#ifndef WALDO_ABL_FCW_NOSTORE
        ac[p] = -1.0f;
        amax = nan_max(amax, -1.0f);
#endif
      }
      if (j + k < L) ac += HWd;
    }
    __builtin_amdgcn_sched_barrier(0);  // a quad of columns at a time (bounds the registers)
  }
  flow[((int64_t)m * 2) * HWd + p] = ox;
  flow[((int64_t)m * 2 + 1) * HWd + p] = oy;
  if (alpha_max != nullptr) alpha_max[(int64_t)m * HWd + p] = amax;
  if (SCORE) score[(int64_t)m * HWd + p] = ssum;
  }  // rows of this thread
}

// A10: Warper.input_to_output (models/nets/lvd.py:830-853), forward: warp of the context frames by
// the composited flow + temporal fusion, one thread per HD pixel of one (b, tp).
//   warped[tc][c] = grid_sample(input[b, ctx_ts[b,tc,tp], c], id_hd + flow[b,tc,tp])
//   score[tc]     = sum_l (alpha[b,tc,tp,l] + 1) / 2;  w[tc] = (score + eps) / max(sum_tc |score + eps|, 1e-12)
//   raw[b,tc,tp]  = cat(warped, alpha);  out[b,tp] = sum_tc cat(warped, 2 score - 1)[tc] * w[tc]
// `include_self` (lvd.py:842-845, only when Tp == T) appends the unwarped frame tp as one more context
// with score 1 and alpha 1.  The reference materialises warped, score, the two concatenations and
// the normalised weights as (B,Tc,Tp,.,Hd,Wd) tensors; here the taps and scores of the Tc contexts
// of a pixel stay in registers and every input channel is sampled, written to `raw` and fused
// into `out` in one pass.

#ifndef WALDO_FWF_TILE_COLS
#define WALDO_FWF_TILE_COLS 32  // workgroup tile = 8 rows x 32 columns (HdTile); measured below
#endif
#ifndef WALDO_FWF_BANDS
#define WALDO_FWF_BANDS 8
#endif
#ifndef WALDO_FWF_TP_INNER
#define WALDO_FWF_TP_INNER 1  // the Tp predicted frames of a clip innermost in an XCD's tile walk (HdTile::pixel_grouped)
#endif
#ifndef WALDO_FWF_NT
#define WALDO_FWF_NT 1  // non-temporal stores for out / raw (read next by another kernel, far larger than any cache): -3.5 %
#endif
__device__ __forceinline__ void fwf_store(float* p, float v) {
#if WALDO_FWF_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// Tile shape (tools_dev/ab_hd.py --amp, C5 size, ms at flow amplitudes of 10 / 25 / 50 / 150 px over 32-pixel
// cells): 4 x 64 (a wavefront = one 64-pixel row segment, as the other kernels of this file) 3.86 / 4.89 / 6.53
// / 14.4; 8 x 32 (a wavefront = two rows of 32) 3.64 / 4.11 / 4.89 / 11.3; 16 x 16: 4.56 / 4.93 / 5.55 / 10.7.
// Under a sheared flow the footprint of a long row segment crosses many image rows and every 8-byte pair
// pulls a line of its own; the squarer wavefront keeps the footprint compact, and at 32 columns the stores
// are still whole 128-byte lines.
template <int TCP>
__global__ __launch_bounds__(kBlock) void frame_warp_fuse_kernel(
    const float* __restrict__ input, const float* __restrict__ flow, const float* __restrict__ alpha,
    const float* __restrict__ score, const int64_t* __restrict__ ctx_ts, float* __restrict__ out,
    float* __restrict__ raw, int* __restrict__ status, int T, int Tc, int Tp, int C, int L, int Hd, int Wd, int include_self,
    float eps, int units, int tiles, int nbands) {
  const int64_t HWd = (int64_t)Hd * Wd;
  int n, x, y;  // n = (b, tp)
#if WALDO_FWF_TP_INNER
  if (!HdTile<WALDO_FWF_TILE_COLS>::pixel_grouped(units / Tp, Tp, Hd, Wd, tiles, nbands, n, x, y) || x >= Wd || y >= Hd) return;
#else
  if (!HdTile<WALDO_FWF_TILE_COLS>::pixel(units, Hd, Wd, tiles, nbands, n, x, y) || x >= Wd || y >= Hd) return;
#endif
  const int b = n / Tp, tp = n % Tp;
  const int64_t p = (int64_t)y * Wd + x;
  float gx0, gy0;
  identity_grid(x, y, Wd, Hd, gx0, gy0);

  // taps, score and source frame of every context of this pixel, in registers (branch-free over the
  // padded context count: a padding context repeats context Tc-1 and is never stored or summed)
  const int Tcx = Tc + (include_self ? 1 : 0);
  // The two taps of a row are ONE 8-byte load at the pair origin xb = clamp(x0, 0, Wd - 2) (inside the row;
  // 4-byte aligned: gfx950 takes unaligned dwordx2 loads): half the gather instructions of four single
  // taps.  Within one texel of the left / right border the pair sits one column off the footprint
  // (shift = x0 - xb = -1 / +1) and its elements are re-assigned to the corners; the corner outside the
  // frame carries weight 0 as before.  Interior wavefronts (shift == 0 in every lane, for every context)
  // skip the re-assignment.
  uint32_t ob0[TCP], ob1[TCP];
  int shift[TCP];
  float w00[TCP], w01[TCP], w10[TCP], w11[TCP], sc[TCP];
  const float* frame[TCP];
  float ssum = 0.0f;
  bool shifted = false;
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    const int tcc = min(tc, Tc - 1);
    const bool real = tc < Tc;
    const int64_t m = ((int64_t)b * Tc + tcc) * Tp + tp;
    const float* fl = flow + m * 2 * HWd + p;
    const Taps t = make_taps(gx0 + fl[0], gy0 + fl[HWd], Hd, Wd);
    {
      const int xb = min(max(t.x0, 0), Wd - 2);
      const int cy0 = min(max(t.y0, 0), Hd - 1), cy1 = min(max(t.y0 + 1, 0), Hd - 1);
      ob0[tc] = (uint32_t)(__mul24(cy0, Wd) + xb) * 4u;
      ob1[tc] = (uint32_t)(__mul24(cy1, Wd) + xb) * 4u;
      shift[tc] = t.x0 - xb;
      shifted |= shift[tc] != 0;
    }
    w00[tc] = t.w00;
    w01[tc] = t.w01;
    w10[tc] = t.w10;
    w11[tc] = t.w11;
    const int ts = __builtin_amdgcn_readfirstlane(checked_frame(ctx_ts, m, T, status, kStatusCtx));  // wave-uniform
    frame[tc] = input + ((int64_t)b * T + ts) * C * HWd;
    float s = 0.0f;
    if (score != nullptr) {
      // the alphas already sit in `raw` (waldo_flow_ctx_warp_raw_fwd wrote them there) and their sum came with
      // them: one plane per context instead of L read and L copied
      s = score[m * HWd + p];
    } else {
      const float* al = alpha + m * L * HWd + p;
      float* rw = raw + ((((int64_t)b * Tp + tp) * Tcx + tcc) * (C + L) + C) * HWd + p;
      for (int l = 0; l < L; ++l) {
        const float av = al[(int64_t)l * HWd];
        s += (av + 1.0f) / 2.0f;
        if (real) fwf_store(rw + (int64_t)l * HWd, av);
      }
    }
    sc[tc] = s;
    ssum += real ? fabsf(s + eps) : 0.0f;
  }
  if (include_self) {
    float* rw = raw + ((((int64_t)b * Tp + tp) * Tcx + Tc) * (C + L) + C) * HWd + p;
    for (int l = 0; l < L; ++l) fwf_store(rw + (int64_t)l * HWd, 1.0f);
    ssum += fabsf(1.0f + eps);
  }
  const float den = fmaxf(ssum, 1e-12f);
  const float wself = (1.0f + eps) / den;
  float wt[TCP];
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) wt[tc] = (tc < Tc) ? (sc[tc] + eps) / den : 0.0f;
  const float* self = input + ((int64_t)b * T + min(tp, T - 1)) * C * HWd + p;
  float* rbase = raw + ((int64_t)b * Tp + tp) * Tcx * (C + L) * HWd + p;  // context tc: + tc * (C+L) * HWd
  float* obase = out + ((int64_t)b * Tp + tp) * (C + 1) * HWd + p;
  // Channel loop, software-pipelined by hand: the sixteen tap loads of channel c + 1 are issued BEFORE the
  // stores of channel c.  Vector-memory operations retire in issue order (loads, stores: one counter), so
  // with the stores first every channel's taps would wait for the previous channel's stores to reach
  // memory -- gathers and stores then take turns instead of overlapping (timing ablations at the C5 size:
  // 4.5 ms as written that way, 3.3 without the raw stores, 3.3 without the gathers, 1.7 without both).
  typedef float f32x2_fw __attribute__((ext_vector_type(2)));
  const bool any_shift = __ballot(shifted) != 0ull;  // wave-uniform
  float tv[TCP][4];
  auto load_taps = [&](int c, float (&v)[TCP][4]) {
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc) {
      const float* plane = frame[tc] + (int64_t)c * HWd;
#ifdef WALDO_ABL_FWF_NOGATHER  // timing-only ablation: one coalesced load instead of the four taps
      v[tc][0] = v[tc][1] = v[tc][2] = v[tc][3] = plane[p];
#else
      const f32x2_fw top = *reinterpret_cast<const f32x2_fw*>(reinterpret_cast<const char*>(plane) + ob0[tc]);
      const f32x2_fw bot = *reinterpret_cast<const f32x2_fw*>(reinterpret_cast<const char*>(plane) + ob1[tc]);
      v[tc][0] = top[0];
      v[tc][1] = top[1];
      v[tc][2] = bot[0];
      v[tc][3] = bot[1];
#endif
    }
  };
  // corners of the footprint from the pair elements (see above); a no-op for interior wavefronts
  auto assign = [&](float (&v)[TCP][4]) {
    if (any_shift) {
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc) {
        const float a0 = v[tc][0], a1 = v[tc][1], b0 = v[tc][2], b1 = v[tc][3];
        v[tc][0] = shift[tc] > 0 ? a1 : a0;
        v[tc][1] = shift[tc] < 0 ? a0 : a1;
        v[tc][2] = shift[tc] > 0 ? b1 : b0;
        v[tc][3] = shift[tc] < 0 ? b0 : b1;
      }
    }
  };
  load_taps(0, tv);
  for (int c = 0; c < C; ++c) {
    float nv[TCP][4];
    load_taps(min(c + 1, C - 1), nv);  // the last trip re-reads its own channel: no branch around the loads
    assign(tv);
    const float vself = include_self ? self[(int64_t)c * HWd] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc) {
      const float v = fmaf(tv[tc][3], w11[tc], fmaf(tv[tc][2], w10[tc], fmaf(tv[tc][1], w01[tc], tv[tc][0] * w00[tc])));
#ifndef WALDO_ABL_FWF_NORAW  // timing-only ablation: without the per-context stores
      if (tc < Tc) fwf_store(rbase + ((int64_t)tc * (C + L) + c) * HWd, v);
#endif
      acc += v * wt[tc];
    }
    if (include_self) {
      fwf_store(rbase + ((int64_t)Tc * (C + L) + c) * HWd, vself);
      acc += vself * wself;
    }
    fwf_store(obase + (int64_t)c * HWd, acc);
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc)
#pragma unroll
      for (int k = 0; k < 4; ++k) tv[tc][k] = nv[tc][k];
  }
  float acc = 0.0f;  // the score channel
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) acc += (sc[tc] * 2.0f - 1.0f) * wt[tc];
  if (include_self) acc += wself;  // (1 * 2 - 1) * w
  obase[(int64_t)C * HWd] = acc;
}

// The same pass with the contexts' FOOTPRINTS STAGED IN LDS.  Timing-only ablations of the kernel above at the Cityscapes
// recipe (10.4 ms per pipeline step): 6.8 ms with one coalesced load in place of a context's two pair gathers, 6.5
// without the per-context stores, 2.6 without both -- the stores run at the HBM write rate, the gathers cost as much
// again although the bytes behind them are few (fetch 7.8 GB per launch against 11.7 GB written, rocprofv3): 184
// divergent 8-byte gathers per wavefront, every one a handful of cache-line look-ups, the four wavefronts of a
// tile each pulling the rows they share.  Here a workgroup (an 8 x 32 tile) takes, per context, the BOX of its
// pixels' pair origins -- block-wide min / max of the clamped rows and columns, packed 16-bit, one barrier --
// and, where the box holds at most 1024 texels (91 % of all (tile, context) pairs under real flows,
// profiles/r04_fwf_boxes_*; per context and uniform: the others gather as above), loads it channel by channel as
// ONE 16-byte load per thread (columns from a multiple of four: needs Wd % 4 == 0), parks it in a 4 KB LDS image and
// reads the taps from there -- four coalesced loads per thread and channel instead of eight gathers, every line
// requested once per tile.  Channel c + 1's loads are in flight while channel c is sampled and stored; two LDS-only
// barriers per channel (no vmcnt wait: the stores keep draining).  Same taps, weights and arithmetic: same bits.
#ifndef WALDO_FWF_LDS
#define WALDO_FWF_LDS 1
#endif
constexpr int kFwfCap = 1024;  // texels of one context's staged box = one float4 per thread
// FULL: Tc == TCP and no `include_self` -- every vector-memory operation of the channel loop is then unconditional,
// and the wait for channel c + 1's box can leave channel c's stores in flight (with a store behind a branch the
// compiler must assume it was not issued and waits for everything: gathers and stores take turns again).
#ifndef WALDO_FWF_LDS_WAVES
#define WALDO_FWF_LDS_WAVES 4  // 116 VGPRs, NO scratch.  (Five waves -- 96 VGPRs -- measured the same speed in round 4 and spilled two
                               // dwords: a kernel with scratch inside a replayed HIP graph faulted on this stack)
#endif
#ifndef WALDO_FWF_PER_CONTEXT
#define WALDO_FWF_PER_CONTEXT 1  // Tc = 4: a context whose box does not fit gathers ALONE (0: the whole tile, as in round 4)
#endif
#ifndef WALDO_FWF_LDS_DB
#define WALDO_FWF_LDS_DB 0  // 1: two sets of images, alternating by channel: one barrier per channel instead of two, 33 KB and four
                            // waves per SIMD -- 9.17-9.42 against 9.09-9.19 ms per C5 step with one set at five waves (A/B)
#endif
template <int TCP, bool FULL>
__global__ __launch_bounds__(kBlock, WALDO_FWF_LDS_WAVES) void frame_warp_fuse_lds_kernel(
    const float* __restrict__ input, const float* __restrict__ flow, const float* __restrict__ alpha,
    const float* __restrict__ score, const int64_t* __restrict__ ctx_ts, float* __restrict__ out,
    float* __restrict__ raw, int* __restrict__ status, int T, int Tc_, int Tp, int C, int L, int Hd, int Wd, int include_self_,
    float eps, int units, int tiles, int nbands) {
  typedef float f32x2_fw __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  static_assert(kBlock * 4 == kFwfCap, "one float4 of the box per thread");
  const int Tc = FULL ? TCP : Tc_;
  const bool include_self = FULL ? false : include_self_ != 0;
  const int64_t HWd = (int64_t)Hd * Wd;
  int n, x, y;  // n = (b, tp)
  if (!HdTile<32>::pixel_grouped(units / Tp, Tp, Hd, Wd, tiles, nbands, n, x, y)) return;  // (uniform)
  // a thread beyond the right / bottom edge works on the tile's last pixel of its row / column: it computes and
  // stores the same values to the same addresses as that pixel's own thread (no branch around the stores, every
  // thread reaches the barriers, the box is that of the live pixels)
  x = min(x, Wd - 1);
  y = min(y, Hd - 1);
  const int b = n / Tp, tp = n % Tp;
  const int64_t p = (int64_t)y * Wd + x;
  const int t = (int)threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  float gx0, gy0;
  identity_grid(x, y, Wd, Hd, gx0, gy0);
  __shared__ __attribute__((aligned(16))) float img[WALDO_FWF_LDS_DB ? 2 : 1][TCP][kFwfCap];
  __shared__ int wbox[kBlock / kWave][TCP][2];

  const int Tcx = Tc + (include_self ? 1 : 0);
  uint32_t ob0[TCP], ob1[TCP];  // byte offsets of the two pair origins: in the context's LDS image, or in the plane
  int shift[TCP];
  float w00[TCP], w01[TCP], w10[TCP], w11[TCP], sc[TCP];
  int cyx0[TCP], cy1v[TCP];  // (clamped row y0, pair origin xb) packed; clamped row y1
  const float* frame[TCP];
  float ssum = 0.0f;
  bool shifted = false;
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    const int tcc = min(tc, Tc - 1);  // (a padding context repeats context Tc - 1 and is never stored or summed)
    const bool real = tc < Tc;
    const int64_t m = ((int64_t)b * Tc + tcc) * Tp + tp;
    const float* fl = flow + m * 2 * HWd + p;
    const Taps tp4 = make_taps(gx0 + fl[0], gy0 + fl[HWd], Hd, Wd);
    const int xb = min(max(tp4.x0, 0), Wd - 2);
    const int cy0 = min(max(tp4.y0, 0), Hd - 1), cy1 = min(max(tp4.y0 + 1, 0), Hd - 1);
    cyx0[tc] = (cy0 << 16) | xb;
    cy1v[tc] = cy1;
    shift[tc] = tp4.x0 - xb;
    shifted |= shift[tc] != 0;
    w00[tc] = tp4.w00;
    w01[tc] = tp4.w01;
    w10[tc] = tp4.w10;
    w11[tc] = tp4.w11;
    // the box of this wavefront: (row, column) pairs packed 16 + 16 bits (Hd, Wd < 32768), six exchange steps
    s16x2 lo = {(short)cy0, (short)xb}, hi = {(short)cy1, (short)(xb + 1)};
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const int olo = __shfl_xor(__builtin_bit_cast(int, lo), d, kWave), ohi = __shfl_xor(__builtin_bit_cast(int, hi), d, kWave);
      lo = __builtin_elementwise_min(lo, __builtin_bit_cast(s16x2, olo));
      hi = __builtin_elementwise_max(hi, __builtin_bit_cast(s16x2, ohi));
    }
    if (lane == 0) {
      wbox[wave][tc][0] = __builtin_bit_cast(int, lo);
      wbox[wave][tc][1] = __builtin_bit_cast(int, hi);
    }
    const int ts = __builtin_amdgcn_readfirstlane(checked_frame(ctx_ts, m, T, status, kStatusCtx));  // wave-uniform
    frame[tc] = input + ((int64_t)b * T + ts) * C * HWd;
    float sv = 0.0f;
    if (score != nullptr) {
      sv = score[m * HWd + p];
    } else {
      const float* al = alpha + m * L * HWd + p;
      float* rw = raw + ((((int64_t)b * Tp + tp) * Tcx + tcc) * (C + L) + C) * HWd + p;
      for (int l = 0; l < L; ++l) {
        const float av = al[(int64_t)l * HWd];
        sv += (av + 1.0f) / 2.0f;
        if (real) fwf_store(rw + (int64_t)l * HWd, av);
      }
    }
    sc[tc] = sv;
    ssum += real ? fabsf(sv + eps) : 0.0f;
  }
  if (include_self) {
    float* rw = raw + ((((int64_t)b * Tp + tp) * Tcx + Tc) * (C + L) + C) * HWd + p;
    for (int l = 0; l < L; ++l) fwf_store(rw + (int64_t)l * HWd, 1.0f);
    ssum += fabsf(1.0f + eps);
  }
  lds_barrier();
  // ---- per context (uniform): the tile's box, whether it fits, this thread's float4 of it
  unsigned stage_mask = 0;  // (uniform) bit tc: the context's box fits its image; the other contexts gather
  unsigned mine = 0;       // bit tc: this thread's float4 lies inside the context's box (it is loaded all the same)
  uint32_t goff[TCP];      // float index of that float4 in a plane of the context's frame
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) {
    s16x2 lo = __builtin_bit_cast(s16x2, wbox[0][tc][0]), hi = __builtin_bit_cast(s16x2, wbox[0][tc][1]);
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) {
      lo = __builtin_elementwise_min(lo, __builtin_bit_cast(s16x2, wbox[w][tc][0]));
      hi = __builtin_elementwise_max(hi, __builtin_bit_cast(s16x2, wbox[w][tc][1]));
    }
    const int ylo = __builtin_amdgcn_readfirstlane((int)lo[0]), xlo = __builtin_amdgcn_readfirstlane((int)lo[1]) & ~3;
    const int yhi = __builtin_amdgcn_readfirstlane((int)hi[0]), xhi = __builtin_amdgcn_readfirstlane((int)hi[1]);
    const int pitch4 = (xhi - xlo + 4) >> 2, nrows = yhi - ylo + 1;  // float4s per row (xlo + 4 pitch4 <= Wd: Wd % 4 == 0)
    const int n4 = nrows * pitch4;
    if (n4 <= kBlock) stage_mask |= 1u << tc;
    // t < 256, the + 0.5: the approximate reciprocal gives the exact quotient (pitch4 <= 256 where it matters)
    const int r = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)pitch4));
    if (t < n4) mine |= 1u << tc;
    // (a thread past the box re-reads a float4 of its last row: in bounds, never written to the image)
    goff[tc] = (uint32_t)(__mul24(ylo + min(r, nrows - 1), Wd) + xlo + 4 * min(max(t - r * pitch4, 0), pitch4 - 1));
    const int cy0 = cyx0[tc] >> 16, xb = cyx0[tc] & 0xffff, pitch = 4 * pitch4;
    ob0[tc] = (uint32_t)(__mul24(cy0 - ylo, pitch) + (xb - xlo)) * 4u;
    ob1[tc] = (uint32_t)(__mul24(cy1v[tc] - ylo, pitch) + (xb - xlo)) * 4u;
  }
  const float den = fmaxf(ssum, 1e-12f);
  const float wself = (1.0f + eps) / den;
  float wt[TCP];
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) wt[tc] = (tc < Tc) ? (sc[tc] + eps) / den : 0.0f;
  const float* self = input + ((int64_t)b * T + min(tp, T - 1)) * C * HWd + p;
  float* rbase = raw + ((int64_t)b * Tp + tp) * Tcx * (C + L) * HWd + p;  // context tc: + tc * (C+L) * HWd
  float* obase = out + ((int64_t)b * Tp + tp) * (C + 1) * HWd + p;
  const bool any_shift = __ballot(shifted) != 0ull;  // wave-uniform
  // corners of the footprint from the pair elements (see the kernel above); a no-op for interior wavefronts
  auto assign = [&](float (&v)[TCP][4]) {
    if (any_shift) {
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc) {
        const float a0 = v[tc][0], a1 = v[tc][1], b0 = v[tc][2], b1 = v[tc][3];
        v[tc][0] = shift[tc] > 0 ? a1 : a0;
        v[tc][1] = shift[tc] < 0 ? a0 : a1;
        v[tc][2] = shift[tc] > 0 ? b1 : b0;
        v[tc][3] = shift[tc] < 0 ? b0 : b1;
      }
    }
  };
  auto fuse_store = [&](int c, const float (&v4)[TCP][4]) {
    const float vself = include_self ? self[(int64_t)c * HWd] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc) {
      const float v = fmaf(v4[tc][3], w11[tc], fmaf(v4[tc][2], w10[tc], fmaf(v4[tc][1], w01[tc], v4[tc][0] * w00[tc])));
#ifndef WALDO_ABL_FWF_NORAW
      if (FULL || tc < Tc) fwf_store(rbase + ((int64_t)tc * (C + L) + c) * HWd, v);
#endif
      acc += v * wt[tc];
    }
    if (include_self) {
      fwf_store(rbase + ((int64_t)Tc * (C + L) + c) * HWd, vself);
      acc += vself * wself;
    }
    fwf_store(obase + (int64_t)c * HWd, acc);
  };
#ifdef WALDO_ABL_FWF_ALLSTAGED  // timing-only ablation: no tile gathers (wrong values where a box does not fit)
  stage_mask = (1u << TCP) - 1u;
#endif
  // The channel loop for a compile-time set of staged contexts (bit tc of MASK): a staged context's taps come from
  // its LDS image (one float4 of its box per thread and channel, loaded a channel ahead), the others' straight from
  // memory as two 8-byte pairs, also a channel ahead.  Round 4 had two loops -- every context staged, or the WHOLE
  // tile gathering as soon as one box was too large; under a folded warp (--motion wild) half of all tiles took the
  // second although most of their contexts fit.  Every vector-memory operation of an instance is unconditional (see
  // FULL above), which is why the set is a template parameter and not a run-time test per context.
  auto channel_loop = [&](auto mask_c) {
    constexpr unsigned MASK = decltype(mask_c)::value;
    constexpr bool kAny = MASK != 0;
#pragma unroll
    for (int tc = 0; tc < TCP; ++tc)
      if (!((MASK >> tc) & 1u)) {  // a gathering context: byte offsets of its pair origins in the plane
        const int cy0 = cyx0[tc] >> 16, xb = cyx0[tc] & 0xffff;
        ob0[tc] = (uint32_t)(__mul24(cy0, Wd) + xb) * 4u;
        ob1[tc] = (uint32_t)(__mul24(cy1v[tc], Wd) + xb) * 4u;
      }
    f32x4 box4[TCP];
    float gv[TCP][4], nv[TCP][4];
    auto issue = [&](int c) {  // channel c: this thread's float4 of every staged box, the pairs of the others
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc) {
        const float* plane = frame[tc] + (int64_t)c * HWd;
        if ((MASK >> tc) & 1u) {
          box4[tc] = *reinterpret_cast<const f32x4*>(plane + goff[tc]);
        } else {
          const f32x2_fw top = *reinterpret_cast<const f32x2_fw*>(reinterpret_cast<const char*>(plane) + ob0[tc]);
          const f32x2_fw bot = *reinterpret_cast<const f32x2_fw*>(reinterpret_cast<const char*>(plane) + ob1[tc]);
          nv[tc][0] = top[0];
          nv[tc][1] = top[1];
          nv[tc][2] = bot[0];
          nv[tc][3] = bot[1];
        }
      }
    };
    auto park = [&](int set) {
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc)
        if (((MASK >> tc) & 1u) && ((mine >> tc) & 1u)) *reinterpret_cast<f32x4*>(&img[set][tc][4 * t]) = box4[tc];
    };
    auto take = [&]() {  // the pairs loaded a channel ahead become this channel's
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc)
        if (!((MASK >> tc) & 1u)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) gv[tc][k] = nv[tc][k];
        }
    };
    issue(0);
    park(0);
    take();
    if (kAny) lds_barrier();
    for (int c = 0; c < C; ++c) {
      issue(min(c + 1, C - 1));  // in flight while channel c is sampled and stored (the last trip re-reads its own)
      const int set = WALDO_FWF_LDS_DB ? (c & 1) : 0;
      float tv[TCP][4];
#pragma unroll
      for (int tc = 0; tc < TCP; ++tc) {
        if ((MASK >> tc) & 1u) {
          const char* im = reinterpret_cast<const char*>(&img[set][tc][0]);
          tv[tc][0] = *reinterpret_cast<const float*>(im + ob0[tc]);
          tv[tc][1] = *reinterpret_cast<const float*>(im + ob0[tc] + 4);
          tv[tc][2] = *reinterpret_cast<const float*>(im + ob1[tc]);
          tv[tc][3] = *reinterpret_cast<const float*>(im + ob1[tc] + 4);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) tv[tc][k] = gv[tc][k];
        }
      }
      assign(tv);
      fuse_store(c, tv);
      if (kAny && !WALDO_FWF_LDS_DB) lds_barrier();  // every thread has read channel c's taps
      park(WALDO_FWF_LDS_DB ? (set ^ 1) : 0);  // (waits for the boxes of channel c + 1, not for channel c's stores; the other
                                               // set was last read before the previous trip's barrier)
      take();
      if (kAny) lds_barrier();
    }
  };
  // (uniform dispatch; a context beyond Tc repeats context Tc - 1: same box, same bit)
  if (FULL && TCP == 4 && WALDO_FWF_PER_CONTEXT) {
    switch (stage_mask & 15u) {
#define WALDO_FWF_CASE(M) case M: channel_loop(std::integral_constant<unsigned, M>{}); break;
      WALDO_FWF_CASE(0) WALDO_FWF_CASE(1) WALDO_FWF_CASE(2) WALDO_FWF_CASE(3) WALDO_FWF_CASE(4) WALDO_FWF_CASE(5)
      WALDO_FWF_CASE(6) WALDO_FWF_CASE(7) WALDO_FWF_CASE(8) WALDO_FWF_CASE(9) WALDO_FWF_CASE(10) WALDO_FWF_CASE(11)
      WALDO_FWF_CASE(12) WALDO_FWF_CASE(13) WALDO_FWF_CASE(14) WALDO_FWF_CASE(15)
#undef WALDO_FWF_CASE
    }
  } else if (stage_mask == (1u << TCP) - 1u) {
    channel_loop(std::integral_constant<unsigned, (1u << TCP) - 1u>{});
  } else {
    channel_loop(std::integral_constant<unsigned, 0u>{});
  }
  float acc = 0.0f;  // the score channel
#pragma unroll
  for (int tc = 0; tc < TCP; ++tc) acc += (sc[tc] * 2.0f - 1.0f) * wt[tc];
  if (include_self) acc += wself;  // (1 * 2 - 1) * w
  obase[(int64_t)C * HWd] = acc;
}

static int check_flow_ctx(const char* fn, int64_t N, int L, int H, int W, int scale) {
  // (W * scale >= 2: the gathers read the two taps of a row as one 8-byte pair inside the row, pair_taps())
  if (N < 0 || L < 1 || L > 32 || H < 1 || W < 1 || scale < 1 || scale > 64 || (int64_t)W * scale < 2 ||
      (int64_t)H * scale > 32767 || (int64_t)W * scale > 32767) {
    set_error("%s: bad shape N=%lld L=%d H=%d W=%d scale=%d (need 1<=L<=32, integer scale, 2 <= HD width, HD side < 32768)",
              fn, (long long)N, L, H, W, scale);
    return WALDO_EINVAL;
  }
  if (hd_grid(N, hd_geom(N, H * scale, W * scale)) > 2147483647) {
    set_error("%s: problem too large for one launch", fn);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

#define WALDO_FC_CASE(LPV, KERNEL, ...)                                                      \
  case LPV:                                                                                  \
    hipLaunchKernelGGL((KERNEL<LPV>), dim3((unsigned)hd_grid(N, geom)), dim3(kBlock), 0, st, \
                       __VA_ARGS__, (int)N, geom.tiles, geom.nbands);                        \
    break;

extern "C" int waldo_flow_ctx_alpha_fwd(const float* alpha_lr, const float* input, const float* dist,
                                        const float* occ, float* a01, float* alpha_out, unsigned* layer_bits, int B,
                                        int T, int Tw, int L, int Nl, int C, int chan_off, int H, int W,
                                        int scale, waldo_stream_t stream) {
  const int64_t N = (int64_t)B * Tw;
  int rc = check_flow_ctx("waldo_flow_ctx_alpha_fwd", N, L, H, W, scale);
  if (rc) return rc;
  if (B < 0 || T < 1 || Tw < 1 || Tw > T ||
      (dist != nullptr && (Nl < 1 || Nl > kMaxCls || chan_off < 0 || chan_off + Nl > C))) {
    set_error("waldo_flow_ctx_alpha_fwd: bad frame window Tw=%d of T=%d or class channels [%d, %d) of %d "
              "(at most %d classes)", Tw, T, chan_off, chan_off + Nl, C, kMaxCls);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!alpha_lr || !occ || !a01 || (dist != nullptr && !input)) {
    set_error("waldo_flow_ctx_alpha_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const HdGeom geom = hd_geom(N, H * scale, W * scale);
  // (the class probabilities of a pixel live in registers: compiled for up to kFewCls classes and for kMaxCls)
#define WALDO_FCA_CASE(LPV)                                                                                              \
  case LPV:                                                                                                              \
    if (dist == nullptr || Nl <= kFewCls)                                                                                \
      hipLaunchKernelGGL((flow_ctx_alpha_kernel<LPV, kFewCls>), dim3((unsigned)hd_grid(N, geom)), dim3(kBlock), 0, st,   \
                         alpha_lr, input, dist, occ, a01, alpha_out, layer_bits, T, Tw, L, Nl, C, chan_off, H, W, scale, \
                         (int)N, geom.tiles, geom.nbands);                                                               \
    else                                                                                                                 \
      hipLaunchKernelGGL((flow_ctx_alpha_kernel<LPV, kMaxCls>), dim3((unsigned)hd_grid(N, geom)), dim3(kBlock), 0, st,   \
                         alpha_lr, input, dist, occ, a01, alpha_out, layer_bits, T, Tw, L, Nl, C, chan_off, H, W, scale, \
                         (int)N, geom.tiles, geom.nbands);                                                               \
    break;
  switch (flow_ctx_pad_l(L)) {
    WALDO_FCA_CASE(4)
    WALDO_FCA_CASE(8)
    WALDO_FCA_CASE(12)
    WALDO_FCA_CASE(17)
    WALDO_FCA_CASE(24)
    WALDO_FCA_CASE(32)
  }
#undef WALDO_FCA_CASE
  return launch_status("waldo_flow_ctx_alpha_fwd");
}

#define WALDO_FCW_LAUNCH(LPV, SC, RV)                                                                          \
  hipLaunchKernelGGL((flow_ctx_warp_kernel<LPV, SC, RV, (RV == kFcwRows && kFcwRows > 1 && LPV >= 8 && LPV <= 17) ? WALDO_FCW_COMPACT : 0>), \
                     dim3((unsigned)fcw_grid), dim3(kBlock), 0, st,                                           \
                     flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, flow, alpha_ctx, lay, score, disocc, alpha_max, \
                     layer_bits, status, T, Tw, Tc, Tp, L, H, W, scale, (int)N, geom.tiles, geom.nbands)
#define WALDO_FCW_CASE(LPV)                                        \
  case LPV:                                                       \
    if (rows == 1) {                                              \
      if (score != nullptr) WALDO_FCW_LAUNCH(LPV, true, 1);       \
      else WALDO_FCW_LAUNCH(LPV, false, 1);                       \
    } else if (rows == 2) {                                       \
      if (score != nullptr) WALDO_FCW_LAUNCH(LPV, true, 2);       \
      else WALDO_FCW_LAUNCH(LPV, false, 2);                       \
    } else {                                                      \
      if (score != nullptr) WALDO_FCW_LAUNCH(LPV, true, kFcwRows); \
      else WALDO_FCW_LAUNCH(LPV, false, kFcwRows);                \
    }                                                             \
    break;

static int flow_ctx_warp_launch(const char* fn, const float* flow_lr, const float* isobj_lr, const float* a01,
                                const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ, float* flow,
                                float* alpha_ctx, ActxLayout lay, float* score, float* disocc, float* alpha_max,
                                const unsigned* layer_bits, int* status, int B, int T, int Tw, int Tc, int Tp, int L,
                                int H, int W, int scale, waldo_stream_t stream) {
  const int64_t N = (int64_t)B * Tc * Tp;
  int rc = check_flow_ctx(fn, N, L, H, W, scale);
  if (rc) return rc;
  if (B < 0 || T < 1 || Tw < 1 || Tw > T || Tc < 0 || Tp < 0) {
    set_error("%s: bad frame counts T=%d Tw=%d Tc=%d Tp=%d", fn, T, Tw, Tc, Tp);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!flow_lr || !a01 || !ctx_ts || !pred_ts || !occ || !flow || !alpha_ctx || !disocc) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  // tall tiles where the tall tile's low-resolution patch still fits the staged image (x4 at the Cityscapes recipe:
  // 6 x 18 cells; at x2, the KITTI recipe, 10 x 34 cells do not fit 256 threads / the LDS image, and the unstaged path
  // is far slower: 5.9 against 4.0 ms per C4 pipeline step) -- there two pixels per thread (6 x 34 cells) if that
  // fits; otherwise the 4 x 64 tile of the other kernels
  const int lp = flow_ctx_pad_l(L);
  const int cell_floats = 4 * lp + 4;  // FcwLds<LP, R>::kCell
  auto fits = [&](int r) {
    const int cells = ((kHdRows * r + scale - 1) / scale + 2) * ((kHdCols + scale - 1) / scale + 2);
    return cells <= kBlock && cells * cell_floats <= fcw_cap(lp, r);
  };
  const int rows = scale < 2 ? 1 : (fits(kFcwRows) ? kFcwRows : (fits(2) ? 2 : 1));
  HdGeom geom = hd_geom_rows(N, H * scale, W * scale, rows);
#if WALDO_FCW_TP_INNER
  geom.nbands = 8;
  const int64_t fcw_grid = xcd_grid_banded((int64_t)B * Tc, geom.nbands, geom.tiles, Tp);
#else
  const int64_t fcw_grid = hd_grid(N, geom);
#endif
  if (fcw_grid > 2147483647) {
    set_error("%s: problem too large for one launch", fn);
    return WALDO_EINVAL;
  }
  switch (flow_ctx_pad_l(L)) {
    WALDO_FCW_CASE(4)
    WALDO_FCW_CASE(8)
    WALDO_FCW_CASE(12)
    WALDO_FCW_CASE(17)
    WALDO_FCW_CASE(24)
    WALDO_FCW_CASE(32)
  }
  return launch_status(fn);
}

extern "C" int waldo_flow_ctx_warp_fwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                                       const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ,
                                       float* flow, float* alpha_ctx, float* disocc, float* alpha_max,
                                       const unsigned* layer_bits, int* status, int B, int T, int Tw, int Tc, int Tp,
                                       int L, int H, int W, int scale, waldo_stream_t stream) {
  const int64_t plane = (int64_t)H * scale * W * scale;
  const ActxLayout lay = {(int64_t)Tc * Tp * L * plane, (int64_t)Tp * L * plane, (int64_t)L * plane};
  return flow_ctx_warp_launch("waldo_flow_ctx_warp_fwd", flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, flow,
                              alpha_ctx, lay, nullptr, disocc, alpha_max, layer_bits, status, B, T, Tw, Tc, Tp, L, H, W,
                              scale, stream);
}

extern "C" int waldo_flow_ctx_warp_raw_fwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                                           const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ,
                                           float* flow, float* raw, float* score, float* disocc,
                                           float* alpha_max, const unsigned* layer_bits, int* status, int B, int T,
                                           int Tw, int Tc, int Tp, int L, int H, int W, int scale, int C, int Tcx,
                                           waldo_stream_t stream) {
  if (C < 1 || Tcx < Tc || Tcx > Tc + 1 || !raw || !score) {
    set_error("waldo_flow_ctx_warp_raw_fwd: bad raw layout C=%d Tc'=%d for Tc=%d (need C >= 1, Tc <= Tc' <= Tc + 1, "
              "raw and score)", C, Tcx, Tc);
    return WALDO_EINVAL;
  }
  const int64_t plane = (int64_t)H * scale * W * scale, ctx = (int64_t)(C + L) * plane;
  const ActxLayout lay = {(int64_t)Tp * Tcx * ctx, ctx, (int64_t)Tcx * ctx};
  return flow_ctx_warp_launch("waldo_flow_ctx_warp_raw_fwd", flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, flow,
                              raw + (int64_t)C * plane, lay, score, disocc, alpha_max, layer_bits, status, B, T, Tw, Tc,
                              Tp, L, H, W, scale, stream);
}

static int frame_warp_fuse_launch(const char* fn, const float* input, const float* flow, const float* alpha,
                                  const float* score, const int64_t* ctx_ts, float* out, float* raw, int* status,
                                  int B, int T, int Tc, int Tp, int C, int L, int Hd, int Wd, int include_self,
                                  float eps, waldo_stream_t stream) {
  if (B < 0 || T < 1 || Tc < 1 || Tc + (include_self ? 1 : 0) > kFwMaxCtx || Tp < 1 || C < 1 || L < 1 ||
      Hd < 1 || Wd < 1 || Hd > 32767 || Wd > 32767 || (include_self && Tp != T)) {
    set_error("%s: bad shape B=%d T=%d Tc=%d Tp=%d C=%d L=%d Hd=%d Wd=%d include_self=%d "
              "(at most %d contexts incl. self; include_self needs Tp == T)", fn, B, T, Tc, Tp, C, L, Hd, Wd,
              include_self, kFwMaxCtx);
    return WALDO_EINVAL;
  }
  if (Wd < 2 || Hd < 1) {
    set_error("%s: frames of %d x %d (need at least two columns)", fn, Hd, Wd);
    return WALDO_EINVAL;
  }
  const int64_t units = (int64_t)B * Tp;
  HdGeom geom = HdTile<WALDO_FWF_TILE_COLS>::geom(units, Hd, Wd);
  // Every unit's tiles in 8 bands, one per XCD: the whole chip walks the (b, tp) units IN ORDER instead of eight
  // units side by side, so the Tp units of a clip, which gather from the same Tc context frames, follow each other
  // closely (the frames of one clip, 193 MB at the Cityscapes recipe, are what the 256 MiB Infinity Cache can hold).
  // A/B on one box: 10.94 -> 10.59 ms per C5 pipeline step (-3 %).
  geom.nbands = WALDO_FWF_BANDS;
  if (hd_grid(units, geom) > 2147483647 || xcd_grid_banded(B, geom.nbands, geom.tiles, Tp) > 2147483647) {
    set_error("%s: problem too large for one launch", fn);
    return WALDO_EINVAL;
  }
  if (units == 0) return WALDO_OK;
  if (!input || !flow || (!alpha && !score) || !ctx_ts || !out || !raw) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
#if WALDO_FWF_TP_INNER
  const dim3 grid((unsigned)xcd_grid_banded(B, geom.nbands, geom.tiles, Tp));
#else
  const dim3 grid((unsigned)hd_grid(units, geom));
#endif
#if WALDO_FWF_LDS && WALDO_FWF_TP_INNER && WALDO_FWF_TILE_COLS == 32
  // (16-byte loads of the boxes: rows that start on a multiple of four texels from a 16-byte aligned base)
  if (Wd % 4 == 0 && (reinterpret_cast<uintptr_t>(input) & 15) == 0 && Tc <= 4) {
    // (the context count is a template parameter: a padding context repeats the last real one -- its taps, its box, its
    // loads -- so one context compiled for four did four contexts' work: the LVD recipe's "prev" mode, 114 us per call)
#define WALDO_FWF_LAUNCH(TCPV, FULLV)                                                                                   \
  hipLaunchKernelGGL((frame_warp_fuse_lds_kernel<TCPV, FULLV>), grid, dim3(kBlock), 0, (hipStream_t)stream, input, flow, \
                     alpha, score, ctx_ts, out, raw, status, T, Tc, Tp, C, L, Hd, Wd, include_self, eps, (int)units,     \
                     geom.tiles, geom.nbands)
    if (Tc == 4 && !include_self) WALDO_FWF_LAUNCH(4, true);
    else if (Tc == 1) WALDO_FWF_LAUNCH(1, false);
    else if (Tc == 2) WALDO_FWF_LAUNCH(2, false);
    else WALDO_FWF_LAUNCH(4, false);
#undef WALDO_FWF_LAUNCH
    return launch_status(fn);
  }
#endif
  if (Tc == 1)
    hipLaunchKernelGGL(frame_warp_fuse_kernel<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, input, flow,
                       alpha, score, ctx_ts, out, raw, status, T, Tc, Tp, C, L, Hd, Wd, include_self, eps, (int)units, geom.tiles, geom.nbands);
  else if (Tc <= 4)
    hipLaunchKernelGGL(frame_warp_fuse_kernel<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, input, flow,
                       alpha, score, ctx_ts, out, raw, status, T, Tc, Tp, C, L, Hd, Wd, include_self, eps, (int)units, geom.tiles, geom.nbands);
  else
    hipLaunchKernelGGL(frame_warp_fuse_kernel<8>, grid, dim3(kBlock), 0, (hipStream_t)stream, input, flow,
                       alpha, score, ctx_ts, out, raw, status, T, Tc, Tp, C, L, Hd, Wd, include_self, eps, (int)units, geom.tiles, geom.nbands);
  return launch_status(fn);
}

extern "C" int waldo_frame_warp_fuse_fwd(const float* input, const float* flow, const float* alpha,
                                         const int64_t* ctx_ts, float* out, float* raw, int* status, int B, int T,
                                         int Tc, int Tp, int C, int L, int Hd, int Wd, int include_self,
                                         float eps, waldo_stream_t stream) {
  if (B > 0 && !alpha) {
    set_error("waldo_frame_warp_fuse_fwd: null pointer");
    return WALDO_EINVAL;
  }
  return frame_warp_fuse_launch("waldo_frame_warp_fuse_fwd", input, flow, alpha, nullptr, ctx_ts, out, raw, status, B, T,
                                Tc, Tp, C, L, Hd, Wd, include_self, eps, stream);
}

extern "C" int waldo_frame_warp_fuse_raw_fwd(const float* input, const float* flow, const float* score,
                                             const int64_t* ctx_ts, float* out, float* raw, int* status, int B, int T,
                                             int Tc, int Tp, int C, int L, int Hd, int Wd, int include_self, float eps,
                                             waldo_stream_t stream) {
  if (B > 0 && !score) {
    set_error("waldo_frame_warp_fuse_raw_fwd: null pointer");
    return WALDO_EINVAL;
  }
  return frame_warp_fuse_launch("waldo_frame_warp_fuse_raw_fwd", input, flow, nullptr, score, ctx_ts, out, raw, status,
                                B, T, Tc, Tp, C, L, Hd, Wd, include_self, eps, stream);
}
