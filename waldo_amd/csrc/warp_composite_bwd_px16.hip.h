// K1 of the two-kernel backward, LDS-staged variant (K3 == 19, L <= 8, 4 | W): pixel-major on
// 16 x 16-pixel tiles.
//
// Restates, for one (frame, tile), the backward of TPSWarp -> F.grid_sample -> reduce_comp
// (models/modules/warp.py:57-64, models/nets/lvd.py:100-114 and the autograd the reference gets
// for them) exactly as warp_composite_bwd_px_kernel does, and leaves the same records and cell
// table for the splat kernel.  Differences:
//  * the samples and their derivatives come out of an LDS image of each layer's footprint box,
//    staged with 16-byte loads as in warp_composite_fwd_lds_kernel (the gather variant issues 128
//    4-byte loads per pixel through the vector-memory path); a layer whose box does not fit the
//    image (violent warp) is gathered from memory instead;
//  * a 16 x 16 tile covers two 8 x 16 cells of the footprint table completely, so their boxes
//    (from the ranges of the grid coordinates in the MFMA accumulator layout) and bounds are
//    plain stores: no global atomics, no memset of the table.
#pragma once
// included at the end of warp_composite_kernels.hip.h, after warp_composite_fwd_lds.hip.h

namespace waldo {

struct BoxTaps {
  float w00, w01, w10, w11;      // corner weights, identical to Taps
  float fx, fy;                  // fractional parts
  float v00, v01, v10, v11;      // 1 / 0 validity of the corners
  int xb, yb;                    // origin of the 2x2 block that is read (inside the layer)
  int cs, rs;                    // x0 - xb, y0 - yb: 0 in the interior, +-1 at a clamped border
};

__device__ __forceinline__ BoxTaps make_box_taps(const TapCore& tc, int Hi, int Wi) {
  const Taps t = finish_taps(tc, Hi, Wi);
  BoxTaps p;
  p.w00 = t.w00;
  p.w01 = t.w01;
  p.w10 = t.w10;
  p.w11 = t.w11;
  p.fx = t.fx;
  p.fy = t.fy;
  p.v00 = t.vx0 * t.vy0;
  p.v01 = t.vx1 * t.vy0;
  p.v10 = t.vx0 * t.vy1;
  p.v11 = t.vx1 * t.vy1;
  p.xb = min(max(t.x0, 0), Wi - 2);
  p.yb = min(max(t.y0, 0), Hi - 2);
  p.cs = t.x0 - p.xb;
  p.rs = t.y0 - p.yb;
  return p;
}

__device__ __forceinline__ int pk_max_u16(int a, int b) {
  typedef unsigned short u2 __attribute__((ext_vector_type(2)));
  u2 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  u2 r = __builtin_elementwise_max(x, y);
  int o;
  __builtin_memcpy(&o, &r, 4);
  return o;
}

template <int LP, bool EXL, bool GOCC>
__global__ __launch_bounds__(kBlock, GOCC ? 2 : 3) void warp_composite_bwd_px16_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ,
    const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
    float4* __restrict__ rec, int* __restrict__ cellbox,
    unsigned* __restrict__ cellbound, float* __restrict__ gmap_partial, float* __restrict__ grad_occ,
    int F, int Lrt, int H, int W, int ntx, int ntiles, int ncx, int ncells) {
  static_assert(LP <= 8 && (LP % 2) == 0, "one 16-column MFMA tile of (layer, xy) columns");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int K3 = kGmapK3, KS = (K3 + 3) / 4;
  constexpr int GGC = 16, TP = GGC + 1;
  constexpr int PP1 = kBlock + 1;
  constexpr int kParkFloats = (4 * LP > GGC ? 4 * LP : GGC) * PP1;
  constexpr int kImgFloats = 2 * 4 * kStageCap, kTFloats = 4 * kWave * TP;
  // phase (H) transposes each wave's 64 x 20 basis values through LDS (pitch 21): three slices fit
  // the staged-image region; the fourth goes behind the gg rows when the parked rows reach that far
  // (LP == 8), else the region is sized for four
  constexpr int BP = 21, kBtFloats = kWave * BP;
  constexpr bool kBtInPark = 4 * LP * PP1 >= GGC * PP1 + kBtFloats;
  constexpr int kBtStage = (kBtInPark ? 3 : 4) * kBtFloats;
  constexpr int kStage0 = kImgFloats > kTFloats ? kImgFloats : kTFloats;
  constexpr int kStageFloats = (kStage0 > kBtStage ? kStage0 : kBtStage + 3) / 4 * 4;
  static_assert(kParkFloats % 4 == 0 && kStageFloats % 4 == 0, "16-byte alignment");
  const int L = EXL ? LP : Lrt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int arow = lane & 15, kk = lane >> 4;
  int f, tile;  // frame pinned to an XCD
  if (!xcd_decode(blockIdx.x, F, ntiles, f, tile)) return;

  // LDS:
  //   park   rows 0 .. 4*LP-1 (pitch 257, one column per pixel): the parked tap derivatives; then
  //          rows 0 .. 15 the grid gradients gg (MFMA B operand) / the per-wave MFMA accumulators;
  //   img    the transposition slices of the grid, then the staged layer image (two buffers);
  //   boxred coordinate ranges per wave; wbound contribution bounds per wave.
  __shared__ __attribute__((aligned(16))) float lds[kParkFloats + kStageFloats + 4 * GGC * 2 + 4 * LP];
  float* gg = lds;
  float* img = lds + kParkFloats;
  float* boxred = img + kStageFloats;
  int* wbound = reinterpret_cast<int*>(boxred + 4 * GGC * 2);  // [wave][layer pair]: packed exponents
  const int pix = threadIdx.x;

  const int col0 = (tile % ntx) * kLdsTile, row0 = (tile / ntx) * kLdsTile + wave * 4;
  // 16 x 16 tile: wave w covers rows 4w .. 4w+3, lane -> (row 4w + lane / 16, column lane % 16)
  const bool live = col0 + arow < W && row0 + kk < H;
  const float livef = live ? 1.0f : 0.0f;
  const int64_t p = (int64_t)min(row0 + kk, H - 1) * W + min(col0 + arow, W - 1);
  const float* oc = occ + (int64_t)f * L * L;
  // zero-weight taps of wild (NaN) coordinates may read any word of the image: keep it finite
  for (int i = threadIdx.x; i < kStageFloats; i += kBlock) img[i] = 0.0f;

  // ---- (A) TPS grid of every layer on the matrix pipe (see warp_composite_fwd_lds_kernel)
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  {
    const float* mp = mapping + (int64_t)f * L * K3 * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk;
      const int l = arow >> 1;
      const float m = mp[(min(l, L - 1) * K3 + min(k, K3 - 1)) * 2 + (arow & 1)];
      const float bv = (k < K3 && l < L) ? m : 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t pa = (uint32_t)(min(row0 + g, H - 1) * W + min(col0 + arow, W - 1));
        // 32-bit byte offset from the uniform base: K3 * HW * 4 < 2^32 is checked by the launcher
        const float bs = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + pa) * 4u);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32((k < K3) ? bs : 0.0f, bv, acc[g], 0, 0, 0);
      }
    }
  }
  // ---- (B) range of every grid coordinate over the workgroup's pixels
  {
    float mn = acc[0][0], mx = mn;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mn = fminf(mn, acc[g][r]);
        mx = fmaxf(mx, acc[g][r]);
      }
    mn = fminf(mn, __shfl_xor(mn, 16, kWave));
    mx = fmaxf(mx, __shfl_xor(mx, 16, kWave));
    mn = fminf(mn, __shfl_xor(mn, 32, kWave));
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    if (kk == 0) {
      boxred[(wave * GGC + arow) * 2 + 0] = mn;
      boxred[(wave * GGC + arow) * 2 + 1] = mx;
    }
  }
  __syncthreads();  // the image is zeroed before the slices inside it are written
  // ---- (C) accumulators -> one pixel per lane, through this wave's slice of LDS
  float gxs[LP], gys[LP];
  {
    float* T = img + wave * (kWave * TP);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(16 * g + kk * 4 + r) * TP + arow] = acc[g][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      gxs[l] = T[lane * TP + 2 * l];
      gys[l] = T[lane * TP + 2 * l + 1];
    }
  }
  __syncthreads();  // ranges of all waves visible; the slices (inside the image) are free again
  // ---- (D) box of the 2x2 blocks of every layer: lanes 0..15 turn the range of "their" column
  // (layer, xy) into block origins -- per cell of the footprint table (this tile owns its
  // cells: plain stores in the splat kernel's format (min x, -max x, min y, -max y)), and for
  // the whole tile (the box that is staged; corners to SGPRs)
  int bx0[LP], by0[LP], bw[LP], bh[LP];
  bool fits[LP];
  {
    const int size = (arow & 1) ? H : W;
    int lo_t = 0x7fffffff, hi_t = -1;
    const int l2 = arow >> 1;
    constexpr int kCellsPerTile = kLdsTile / kCellRows, kWavesPerCell = 4 / kCellsPerTile;
    static_assert(kCellsPerTile * kCellRows == kLdsTile && kWavesPerCell * kCellsPerTile == 4, "cell rows: 4, 8 or 16");
#pragma unroll
    for (int c = 0; c < kCellsPerTile; ++c) {
      float mn = boxred[((kWavesPerCell * c) * GGC + arow) * 2 + 0], mx = boxred[((kWavesPerCell * c) * GGC + arow) * 2 + 1];
#pragma unroll
      for (int w = 1; w < kWavesPerCell; ++w) {
        mn = fminf(mn, boxred[((kWavesPerCell * c + w) * GGC + arow) * 2 + 0]);
        mx = fmaxf(mx, boxred[((kWavesPerCell * c + w) * GGC + arow) * 2 + 1]);
      }
      const int lo_c = block_origin(mn, size), hi_c = block_origin(mx, size) + 1;
      lo_t = min(lo_t, lo_c);
      hi_t = max(hi_t, hi_c);
      const int crow = (tile / ntx) * kCellsPerTile + c;
      if (wave == 0 && kk == 0 && l2 < L && crow * ncx < ncells) {
        int* bb = cellbox + (((int64_t)f * L + l2) * ncells + crow * ncx + (tile % ntx)) * 4 + (arow & 1) * 2;
        *reinterpret_cast<int2*>(bb) = make_int2(lo_c, -hi_c);
      }
    }
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int xmin = __builtin_amdgcn_readlane(lo_t, 2 * l), xmax = __builtin_amdgcn_readlane(hi_t, 2 * l);
      const int ymin = __builtin_amdgcn_readlane(lo_t, 2 * l + 1), ymax = __builtin_amdgcn_readlane(hi_t, 2 * l + 1);
      bx0[l] = xmin & ~3;
      by0[l] = ymin;
      bw[l] = ((xmax - bx0[l] + 1) + 3) & ~3;
      bh[l] = ymax - ymin + 1;
      fits[l] = bh[l] * bw[l] <= kStageCap;  // block-uniform
    }
  }

  const float g0 = grad_rgb[(int64_t)f * 3 * HW + p] * livef;
  const float g1 = grad_rgb[(int64_t)f * 3 * HW + HW + p] * livef;
  const float g2 = grad_rgb[(int64_t)f * 3 * HW + 2 * HW + p] * livef;
  const float gmax = fmaxf(fabsf(g0), fmaxf(fabsf(g1), fabsf(g2)));
  float a[LP], G[LP];

  // ---- (E) staged sampling with derivatives.  Wave w moves channel plane w of a layer's box,
  // 16 bytes per lane; a rolling window of kAhead layers is in flight (the load of layer l + kAhead
  // is issued when layer l leaves its registers for LDS); the image is double-buffered, one
  // barrier per layer.
  constexpr int kAhead = LP < WALDO_STAGE_AHEAD ? LP : WALDO_STAGE_AHEAD;
  constexpr int kItems = kStageCap / 4 / kWave;
  f32x4 stg[LP][kItems];  // fully unrolled: a layer's registers live from its load to its LDS store
  auto issue = [&](int l) {
    const int lc = EXL ? l : min(l, L - 1);
    const float* src = layers + (((int64_t)f * L + lc) * 4 + wave) * HW;
    // unconditional loads (items past the box re-read its last item; a box that does not fit
    // reads texel 0): no exec-mask branches, so the loads are issued back to back
    const int bw4 = bw[l] >> 2, n = fits[l] ? bh[l] * bw4 : 1;
    const int ox = fits[l] ? __mul24(by0[l], W) + bx0[l] : 0;
    const float rcp = 1.0f / (float)bw4;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      const int item = min(lane + j * kWave, n - 1);
      const int r = (int)(((float)item + 0.5f) * rcp);  // item, bw4 < 2^9: exact
      const int xg = item - r * bw4;
      const unsigned off = (unsigned)(ox + __mul24(r, W) + 4 * xg);
      stg[l][j] = *reinterpret_cast<const f32x4*>(src + off);
    }
  };
#pragma unroll
  for (int l = 0; l < kAhead; ++l) issue(l);
  {
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      if (!EXL && l >= L) {  // padding layer: inert
        a[l] = 0.0f;
        G[l] = 0.0f;
#pragma unroll
        for (int d = 0; d < 4; ++d) lds[(4 * l + d) * PP1 + pix] = 0.0f;
        continue;
      }
      if (fits[l]) {
        const int n = bh[l] * (bw[l] >> 2);
        f32x4* dst = reinterpret_cast<f32x4*>(img + ((l & 1) * 4 + wave) * kStageCap);
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
          const int item = lane + j * kWave;
          if (item < n) dst[item] = stg[l][j];  // row-major with pitch bw: item = r * bw4 + xg
        }
      }
      if (l + kAhead < LP) issue(l + kAhead);
      __syncthreads();  // buffer l&1 complete; buffer (l+1)&1 no longer read by anyone
      const TapCore tc = tap_core(gxs[l], gys[l], H, W);
      const float* b0 = img + (l & 1) * 4 * kStageCap;
      float sv[4], sx[4], sy[4];
      if (!fits[l]) {  // box larger than the LDS image (violent warp): gather straight from memory
        const Taps tg = finish_taps(tc, H, W);
        const float* base = layers + ((int64_t)f * L + l) * 4 * HW;
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[c] = tap_sample_d(base + c * HW, tg, sx[c], sy[c]);
      } else if (__ballot(!tap_interior(tc, H, W)) == 0ull) {
        // wave-uniform: all corners inside the layer, every validity factor is exactly 1
        const float wx0 = 1.0f - tc.fx, wy0 = 1.0f - tc.fy;
        const float w00 = wx0 * wy0, w01 = tc.fx * wy0, w10 = wx0 * tc.fy, w11 = tc.fx * tc.fy;
        const int idx = (tc.y0 - by0[l]) * bw[l] + (tc.x0 - bx0[l]);  // inside the box
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* pc = b0 + c * kStageCap + idx;
          const float p00 = pc[0], p01 = pc[1], p10 = pc[bw[l]], p11 = pc[bw[l] + 1];
          const float top = fmaf(tc.fx, p01 - p00, p00);
          const float bot = fmaf(tc.fx, p11 - p10, p10);
          sx[c] = fmaf(tc.fy, (p11 - p10) - (p01 - p00), p01 - p00);
          sy[c] = bot - top;
          sv[c] = fmaf(p11, w11, fmaf(p10, w10, fmaf(p01, w01, p00 * w00)));
        }
      } else {
        const BoxTaps t = make_box_taps(tc, H, W);
        // inside the box by construction; the clamp only matters for NaN coordinates
        const int idx = min(max((t.yb - by0[l]) * bw[l] + (t.xb - bx0[l]), 0), kStageCap - bw[l] - 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* pc = b0 + c * kStageCap + idx;
          const float a0 = pc[0], b0v = pc[1], a1 = pc[bw[l]], b1v = pc[bw[l] + 1];
          // rows: block row 0/1 -> corner rows y0 / y0+1 (rs = y0 - yb)
          const float ta = t.rs > 0 ? a1 : a0, tb = t.rs > 0 ? b1v : b0v;  // corner row y0
          const float ua = t.rs < 0 ? a0 : a1, ub = t.rs < 0 ? b0v : b1v;  // corner row y0 + 1
          const float p00 = t.cs > 0 ? tb : ta, p01 = t.cs < 0 ? ta : tb;
          const float p10 = t.cs > 0 ? ub : ua, p11 = t.cs < 0 ? ua : ub;
          const float v00 = p00 * t.v00, v01 = p01 * t.v01, v10 = p10 * t.v10, v11 = p11 * t.v11;
          const float top = fmaf(t.fx, v01 - v00, v00);
          const float bot = fmaf(t.fx, v11 - v10, v10);
          sx[c] = fmaf(t.fy, (v11 - v10) - (v01 - v00), v01 - v00);
          sy[c] = bot - top;
          sv[c] = fmaf(p11, t.w11, fmaf(p10, t.w10, fmaf(p01, t.w01, p00 * t.w00)));
        }
      }
      a[l] = (sv[3] + 1.0f) * 0.5f;
      G[l] = g0 * (sv[0] + 1.0f) + g1 * (sv[1] + 1.0f) + g2 * (sv[2] + 1.0f);
      lds[(4 * l + 0) * PP1 + pix] = fmaf(g2, sx[2], fmaf(g1, sx[1], g0 * sx[0]));
      lds[(4 * l + 1) * PP1 + pix] = sx[3];
      lds[(4 * l + 2) * PP1 + pix] = fmaf(g2, sy[2], fmaf(g1, sy[1], g0 * sy[0]));
      lds[(4 * l + 3) * PP1 + pix] = sy[3];
      if (grad_alpha != nullptr)
        G[l] = fmaf(2.0f * livef, grad_alpha[((int64_t)f * L + l) * HW + p], G[l]);
    }
  }

  // ---- (F) composite backward (lvd.py:100-114): a'_j = a_j prod_i (1 - a_i occ[i][j]).
  // Two layers j per step on the packed-fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of
  // work per VALU issue slot -- this kernel is issue bound); the contributions of even and odd j
  // to d loss / d a_m accumulate separately and are added at the end.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  a[0] = 1.0f;
  float ap[LP], ga[LP];
  {
    f32x2 ga2[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) ga2[l] = (f32x2){0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < LP; j += 2) {
      const int jc0 = EXL ? j : min(j, L - 1), jc1 = EXL ? j + 1 : min(j + 1, L - 1);
      f32x2 tfac[LP], ex[LP];
      f32x2 pre = {1.0f, 1.0f};
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int ic = EXL ? i : min(i, L - 1);
        const f32x2 o = {oc[ic * L + jc0], oc[ic * L + jc1]};
        tfac[i] = (f32x2){1.0f, 1.0f} - (f32x2){a[i], a[i]} * o;
        ex[i] = pre;
        pre = pre * tfac[i];
      }
      f32x2 suf = {1.0f, 1.0f};
#pragma unroll
      for (int i = LP - 1; i >= 0; --i) {
        ex[i] = ex[i] * suf;
        suf = suf * tfac[i];
      }
      ap[j] = a[j] * pre[0];  // 0 for padding layers
      ap[j + 1] = a[j + 1] * pre[1];
      ga[j] = G[j] * pre[0];  // G[j] = d loss / d a'_j
      ga[j + 1] = G[j + 1] * pre[1];
      const f32x2 gaj = (f32x2){G[j], G[j + 1]} * (f32x2){a[j], a[j + 1]};
      float gocc0[LP], gocc1[LP];
#pragma unroll
      for (int m = 0; m < LP; ++m) {
        const int mc = EXL ? m : min(m, L - 1);
        const f32x2 o = {oc[mc * L + jc0], oc[mc * L + jc1]};
        ga2[m] = __builtin_elementwise_fma(-gaj * o, ex[m], ga2[m]);
        if (GOCC) {
          const f32x2 go = -gaj * (f32x2){a[m], a[m]} * ex[m];
          gocc0[m] = go[0];
          gocc1[m] = go[1];
        }
      }
      if (GOCC) {  // compile-time: the reduction costs ~60 registers
        const int m = bitrev6(lane);
        if (EXL || j < L) {
          const float redv = wave_transpose_reduce<LP>(gocc0, lane);
          if (m < L) atomicAdd(grad_occ + (int64_t)f * L * L + m * L + j, redv);
        }
        if (EXL || j + 1 < L) {
          const float redv = wave_transpose_reduce<LP>(gocc1, lane);
          if (m < L) atomicAdd(grad_occ + (int64_t)f * L * L + m * L + j + 1, redv);
        }
      }
    }
#pragma unroll
    for (int l = 0; l < LP; ++l) ga[l] += ga2[l][0] + ga2[l][1];
  }
  // ---- (G) a_m = (s_m3 + 1)/2 for m >= 1; a_0 is the constant 1.  Second half of the records,
  // contribution bounds, and the grid gradient of every layer (read this thread's parked
  // derivatives, then overwrite the same LDS column with gg -- no other thread touches it).
  // The records go out FIRST: phase (H)'s loads complete only after every older store of the wave
  // (vector-memory operations complete in order), so the stores get the rest of this phase to drain.
  if (live) {
#pragma unroll
    for (int l = 0; l < LP; ++l)
      if (EXL || l < L)
        rec[((int64_t)f * L + l) * HW + p] = make_float4(gxs[l], gys[l], ap[l], l >= 1 ? 0.5f * ga[l] : 0.0f);
  }
  {
    float ggx[LP], ggy[LP];
    int eb[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const bool pad = !EXL && l >= L;
      const float gsa = (l >= 1 && !pad) ? 0.5f * ga[l] : 0.0f;
      const float dxr = lds[(4 * l + 0) * PP1 + pix], dxa = lds[(4 * l + 1) * PP1 + pix];
      const float dyr = lds[(4 * l + 2) * PP1 + pix], dya = lds[(4 * l + 3) * PP1 + pix];
      ggx[l] = fmaf(gsa, dxa, ap[l] * dxr) * (0.5f * (float)W);
      ggy[l] = fmaf(gsa, dya, ap[l] * dyr) * (0.5f * (float)H);
      // |tap contribution| <= max(|a'_l| max_c |g_c|, |g_alpha|) < 2^(e - 126), e its biased
      // exponent: bilinear weights are <= 1.  Only e goes into the bound (see below).
      const float cb = (live && !pad) ? fmaxf(fabsf(ap[l]) * gmax, fabsf(gsa)) : 0.0f;
      eb[l] = (int)(__float_as_uint(cb) >> 23);  // cb >= 0: sign bit clear
    }
    // the table's bound (an upper bound of any 16-pixel row sum of the cell): 16 * 2^(e_max - 126)
    // with e_max the largest exponent in the cell -- one packed 16-bit max-reduction per two
    // layers instead of a float row-sum reduction per layer; at most 2x looser (one bit of the
    // splat's 29-bit fixed point)
#pragma unroll
    for (int j = 0; j < LP / 2; ++j) {
      int v = eb[2 * j] | (eb[2 * j + 1] << 16);
      v = pk_max_u16(v, row_ror_i<8>(v));
      v = pk_max_u16(v, row_ror_i<4>(v));
      v = pk_max_u16(v, row_ror_i<2>(v));
      v = pk_max_u16(v, row_ror_i<1>(v));
      v = pk_max_u16(v, __shfl_xor(v, 16, kWave));
      v = pk_max_u16(v, __shfl_xor(v, 32, kWave));
      if (lane == 0) wbound[wave * (LP / 2) + j] = v;
    }
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      gg[(2 * l) * PP1 + pix] = ggx[l];
      gg[(2 * l + 1) * PP1 + pix] = ggy[l];
    }
#pragma unroll
    for (int c = 2 * LP; c < GGC; ++c) gg[c * PP1 + pix] = 0.0f;
  }
  __syncthreads();  // gg rows and the waves' bounds are complete
  {
    constexpr int kCellsPerTile = kLdsTile / kCellRows, kWavesPerCell = 4 / kCellsPerTile;
    if (threadIdx.x < kCellsPerTile * LP) {
      const int c = threadIdx.x / LP, l = threadIdx.x % LP;
      const int crow = (tile / ntx) * kCellsPerTile + c;
      if (l < L && crow * ncx < ncells) {
        int v = wbound[(kWavesPerCell * c) * (LP / 2) + l / 2];
#pragma unroll
        for (int w = 1; w < kWavesPerCell; ++w) v = pk_max_u16(v, wbound[(kWavesPerCell * c + w) * (LP / 2) + l / 2]);
        const int e = (l & 1) ? (v >> 16) & 0xffff : v & 0xffff;
        // 16 * 2^(e - 126) as float bits; nothing but zeros / denormals in the cell: 0
        cellbound[((int64_t)f * L + l) * ncells + crow * ncx + (tile % ntx)] =
            e == 0 ? 0u : (unsigned)min(e + 5, 254) << 23;
      }
    }
  }

  // ---- (H) control-point gradient: basis^T x gg on the MFMA pipe
  // D[k][col] = sum_pix basis[k][pix] * gg[pix][col]; v_mfma_f32_16x16x4_f32: A[row=lane&15]
  // [kk=lane>>4], B[kk=lane>>4][col=lane&15], D[row=(lane>>4)*4+reg][col=lane&15].  Each wave
  // contracts the pixels it produced; the 4 wave results are summed through LDS in a fixed order
  // and stored as this tile's partial (no atomics, deterministic).
  if (gmap_partial != nullptr) {
    // (H1) the A operand wants basis[k][pixel] with k along lane & 15: read that way from memory,
    // an instruction touches 16 rows of basis_t x 16 bytes (0.44 ms of this kernel went into those
    // 32 gathers per lane).  Instead the wave loads its 64 pixels x 20 values as phase (A) does
    // (4 rows x 64 contiguous bytes per instruction) and transposes them through LDS.
    float* Bt = (kBtInPark && wave == 3) ? lds + GGC * PP1 : img + wave * kBtFloats;  // [pixel][k], pitch 21
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t pa = (uint32_t)(min(row0 + g, H - 1) * W + min(col0 + arow, W - 1));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + kk;
        const float v = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + pa) * 4u);
        Bt[(16 * g + arow) * BP + k] = (k < K3) ? v : 0.0f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 macc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) macc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      const int pl = 4 * s4 + kk;  // lane index of the contracted pixel inside this wave
      const int px = wave * kWave + pl;
      float av[2];
      av[0] = Bt[pl * BP + arow];                                     // k = arow
      av[1] = (arow < 4) ? Bt[pl * BP + 16 + min(arow, 3)] : 0.0f;    // k = 16 + arow (19 is the zero pad)
      const float bv = gg[arow * PP1 + px];  // dead pixels carry gg == 0
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv, macc[mt], 0, 0, 0);
    }
    __syncthreads();  // every wave is done reading gg: reuse its bytes for the accumulators
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[(wave * 2 + mt) * 256 + r * 64 + lane] = macc[mt][r];
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * 256; o += kBlock) {
      float sum = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += lds[w * 2 * 256 + o];  // fixed order
      const int mt = o / 256, r = (o >> 6) & 3, ln = o & 63;
      const int k = mt * 16 + (ln >> 4) * 4 + r;
      const int col = ln & 15;
      const int l = col >> 1;
      if (k < K3 && l < L)
        gmap_partial[((int64_t)f * ntiles + tile) * gmap_partial_floats(L) + ((int64_t)l * K3 + k) * 2 +
                     (col & 1)] = sum;
    }
  }
}

}  // namespace waldo
