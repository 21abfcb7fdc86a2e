// Fused WIF hot path for gfx950: TPS grid synthesis -> bilinear backward warp of every
// 4-channel layer -> occlusion / soft-alpha composite (LVD.reduce_comp), forward and backward.
//
// Reference behaviour restated (paths relative to the reference root):
//   models/modules/warp.py:49-55      TPSWarp.forward           grid = basis @ (K^-1 @ [pts;0])
//   torch F.grid_sample defaults      bilinear / zeros / align_corners=False
//   models/nets/lvd.py:100-114        LVD.reduce_comp           a'_j = a_j prod_i (1 - a_i occ_ij)
//
// Kernels (this file and the two it includes at the end)
//   warp_composite_fwd_kernel        generic forward (any K3 <= 32, any width): one thread per
//                                    output pixel, 4x64-pixel tiles, frames looped inside the
//                                    workgroup (the K3 basis values of a pixel stay in registers),
//                                    branch-free layer loop with all tap loads of a group in flight.
//   warp_composite_fwd_lds_kernel    (warp_composite_fwd_lds.hip.h; K3 == 19, 4 | W) forward on
//                                    16x16 tiles: MFMA TPS grid, footprint boxes staged in LDS.
//   warp_composite_bwd_px16_kernel   (warp_composite_bwd_px16.hip.h) K1 of the two-kernel backward.
//   warp_composite_gmap_reduce_kernel  fixed-order sum of K1's control-point partials.
//   warp_composite_splat_kernel      (warp_composite_splat.hip) K2 of the two-kernel backward.
//   warp_composite_bwd_kernel        generic backward (any L <= 32, K3 <= 32, any width): per-tap
//                                    global float atomics; the correctness path for shapes the
//                                    two-kernel backward is not compiled for.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "waldo_common.hip.h"

namespace waldo {

constexpr int kMaxLayers = 32;
constexpr int kMaxK3 = 32;
constexpr int kFwdGroup = 4;  // layers whose tap loads are issued together (forward)
constexpr int kBwdGroup = 2;  // same, tiled backward

// ---------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------
template <int K3P, bool EXK>
__device__ __forceinline__ void load_basis(float (&bas)[K3P], const float* __restrict__ basis_t,
                                           int64_t HW, int64_t p, int K3) {
#pragma unroll
  for (int k = 0; k < K3P; ++k) {
    if constexpr (EXK) {
      bas[k] = basis_t[(int64_t)k * HW + p];
    } else {
      const int kc = min(k, K3 - 1);
      const float v = basis_t[(int64_t)kc * HW + p];
      bas[k] = (k < K3) ? v : 0.0f;
    }
  }
}

// map: (K3,2) wave-uniform.  Sequential fmaf chain in k over the pixel-unit operands of
// scaled_map(): the ONE definition of the grid that every kernel of the fused path shares (the
// MFMA kernels accumulate the same products in the same order: bit-identical coordinates).
template <int K3P, bool EXK>
__device__ __forceinline__ void tps_eval(const float (&bas)[K3P], const float* __restrict__ map,
                                         int K3, int H, int W, float& ix, float& iy) {
  ix = 0.0f;
  iy = 0.0f;
  const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
  const float ow = 0.5f * (float)(W - 1), oh = 0.5f * (float)(H - 1);
#pragma unroll
  for (int k = 0; k < K3P; ++k) {
    const int kc = EXK ? k : min(k, K3 - 1);  // bas[k] == 0 beyond K3
    ix = fmaf(bas[k], scaled_map(map[2 * kc], kc == K3 - 3, hw, ow), ix);
    iy = fmaf(bas[k], scaled_map(map[2 * kc + 1], kc == K3 - 3, hh, oh), iy);
  }
}

__device__ __forceinline__ float opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// pixel of a thread: 2-D tiles of (rows_per_tile x 64) when 64 | W, else linear strips
struct PixelMap {
  int64_t p;   // linear pixel index (clamped into the image for dead lanes)
  bool live;
};

__device__ __forceinline__ PixelMap pixel_of(int tile, int row_in_tile, int lane, int H, int W,
                                             int rows_per_tile, int ntx) {
  PixelMap m;
  const int tx = tile % ntx, ty = tile / ntx;
  const int col = tx * kTileW + lane;
  const int row = ty * rows_per_tile + row_in_tile;
  m.live = (col < W) && (row < H);
  const int cc = min(col, W - 1), rc = min(row, H - 1);
  m.p = (int64_t)rc * W + cc;
  return m;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int LP, int K3P, bool EXL, bool EXK>
__global__ __launch_bounds__(kBlock, (LP <= 8 ? 3 : (LP <= 17 ? 2 : 1))) void warp_composite_fwd_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ, float* __restrict__ rgb,
    float* __restrict__ alpha_out, int F, int Lrt, int H, int W, int K3rt, int frames_per_block,
    int ntx, int ntiles, int nchunks, int nbands, float delta) {
  const int L = EXL ? LP : Lrt;
  const int K3 = EXK ? K3P : K3rt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  int chunk, tile, rest_;  // (frame chunk, band of tiles) pinned to an XCD: neighbours share one L2
  if (!xcd_decode_banded(blockIdx.x, nchunks, nbands, ntiles, 1, chunk, tile, rest_)) return;
  const PixelMap pm = pixel_of(tile, wave, lane, H, W, 4, ntx);
  const int64_t p = pm.p;
  float bas[K3P];
  load_basis<K3P, EXK>(bas, basis_t, HW, p, K3);

  const int f0 = chunk * frames_per_block;
  const int f1 = min(F, f0 + frames_per_block);
  for (int f = f0; f < f1; ++f) {
    float s[LP][4];
    // layers in groups of kFwdGroup: 16 tap loads per layer, and a wave can only have 63 vector
    // memory operations outstanding -- grouping bounds the registers held for loads in flight
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int lc = EXL ? l : min(l, L - 1);  // padding layers re-read layer L-1, then masked
      float ix, iy;
      tps_eval<K3P, EXK>(bas, mapping + ((int64_t)f * L + lc) * K3 * 2, K3, H, W, ix, iy);
      const Taps t = make_taps_px(ix, iy, H, W);
      const float* base = layers + ((int64_t)f * L + lc) * 4 * HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) s[l][c] = tap_sample(base + c * HW, t, delta);
      if (!EXL && l >= L) s[l][3] = -1.0f;  // alpha 0 after (x+1)/2: an inert layer
      if ((l % kFwdGroup) == kFwdGroup - 1) __builtin_amdgcn_sched_barrier(0);
    }
    // composite: a_0 = 1 (lvd.py:105), a_l = (s_l3 + 1) / 2
    float a[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) a[l] = (s[l][3] + 1.0f) * 0.5f;
    a[0] = 1.0f;
    const float* oc = occ + (int64_t)f * L * L;
    float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const int jc = EXL ? j : min(j, L - 1);
      float pr = 1.0f;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int ic = EXL ? i : min(i, L - 1);
        pr *= (1.0f - a[i] * oc[ic * L + jc]);  // a[i] == 0 for padding layers: factor 1
      }
      const float ap = a[j] * pr;  // 0 for padding layers
      r = fmaf(ap, (s[j][0] + 1.0f) * 0.5f, r);
      g = fmaf(ap, (s[j][1] + 1.0f) * 0.5f, g);
      b = fmaf(ap, (s[j][2] + 1.0f) * 0.5f, b);
      if (alpha_out != nullptr && pm.live && (EXL || j < L))
        alpha_out[((int64_t)f * L + j) * HW + p] = 2.0f * ap - 1.0f;
    }
    if (pm.live) {
      float* o = rgb + (int64_t)f * 3 * HW + p;
      o[0] = 2.0f * r - 1.0f;
      o[HW] = 2.0f * g - 1.0f;
      o[2 * HW] = 2.0f * b - 1.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// generic backward (v1): per-tap global float atomics for grad_layers; wave transpose-reduce + one
// atomic per (wave, k, c) for grad_mapping; wave transpose-reduce per column for grad_occ.
// Kept for L > 8 or K3 != 19; the tiled kernel below is the fast path.
// ---------------------------------------------------------------------------------------
template <int LP, int K3P>
__global__ __launch_bounds__(kBlock) void warp_composite_bwd_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ,
    const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
    float* __restrict__ grad_layers, float* __restrict__ grad_mapping,
    float* __restrict__ grad_occ, int F, int L, int H, int W, int K3, float delta) {
  const int64_t HW = (int64_t)H * W;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = p < HW;
  const int64_t pc = live ? p : HW - 1;
  const float livef = live ? 1.0f : 0.0f;
  const int lane = threadIdx.x & (kWave - 1);
  const int f = blockIdx.y;

  float bas[K3P];
  load_basis<K3P, false>(bas, basis_t, HW, pc, K3);

  float s[LP][4], dsx[LP][4], dsy[LP][4], gxs[LP], gys[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    if (l < L) {
      tps_eval<K3P, false>(bas, mapping + ((int64_t)f * L + l) * K3 * 2, K3, H, W, gxs[l], gys[l]);
      Taps t = make_taps_px(gxs[l], gys[l], H, W);
      const float* base = layers + ((int64_t)f * L + l) * 4 * HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) s[l][c] = tap_sample_d(base + c * HW, t, dsx[l][c], dsy[l][c], delta);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) s[l][c] = dsx[l][c] = dsy[l][c] = 0.0f;
      s[l][3] = -1.0f;
      gxs[l] = gys[l] = 0.0f;
    }
  }
  float a[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) a[l] = (s[l][3] + 1.0f) * 0.5f;
  a[0] = 1.0f;

  const float g0 = grad_rgb[(int64_t)f * 3 * HW + pc] * livef;
  const float g1 = grad_rgb[(int64_t)f * 3 * HW + HW + pc] * livef;
  const float g2 = grad_rgb[(int64_t)f * 3 * HW + 2 * HW + pc] * livef;
  const float* oc = occ + (int64_t)f * L * L;

  float ga[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) ga[l] = 0.0f;
  float gs[LP][4];
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    if (j < L) {
      float tfac[LP], ex[LP];
      float pre = 1.0f;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        tfac[i] = (i < L) ? (1.0f - a[i] * oc[i * L + j]) : 1.0f;
        ex[i] = pre;
        pre *= tfac[i];
      }
      float suf = 1.0f;
#pragma unroll
      for (int i = LP - 1; i >= 0; --i) {
        ex[i] *= suf;
        suf *= tfac[i];
      }
      const float P = pre;
      const float ap = a[j] * P;
      gs[j][0] = ap * g0;
      gs[j][1] = ap * g1;
      gs[j][2] = ap * g2;
      float gap = g0 * (s[j][0] + 1.0f) + g1 * (s[j][1] + 1.0f) + g2 * (s[j][2] + 1.0f);
      if (grad_alpha != nullptr)
        gap = fmaf(2.0f * livef, grad_alpha[((int64_t)f * L + j) * HW + pc], gap);
      ga[j] = fmaf(gap, P, ga[j]);
      const float gaj = gap * a[j];
      float gocc[LP];
#pragma unroll
      for (int m = 0; m < LP; ++m) {
        if (m < L) {
          ga[m] = fmaf(-gaj * oc[m * L + j], ex[m], ga[m]);
          gocc[m] = -gaj * a[m] * ex[m];
        } else {
          gocc[m] = 0.0f;
        }
      }
      if (grad_occ != nullptr) {  // wave-uniform branch
        float red = wave_transpose_reduce<LP>(gocc, lane);
        int m = bitrev6(lane);
        if (m < L) atomicAdd(grad_occ + (int64_t)f * L * L + m * L + j, red);
      }
    } else {
      gs[j][0] = gs[j][1] = gs[j][2] = 0.0f;
    }
  }
#pragma unroll
  for (int l = 0; l < LP; ++l) gs[l][3] = (l >= 1) ? 0.5f * ga[l] : 0.0f;

#pragma unroll
  for (int l = 0; l < LP; ++l) {
    if (l < L) {
      Taps t = make_taps_px(gxs[l], gys[l], H, W);
      float* gbase = grad_layers + ((int64_t)f * L + l) * 4 * HW;
      float gix = 0.0f, giy = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gv = gs[l][c];
        gix = fmaf(gv, dsx[l][c], gix);
        giy = fmaf(gv, dsy[l][c], giy);
        float* pl = gbase + c * HW;
        if (live && t.w00 != 0.0f) atomicAdd(pl + (t.o00 >> 2), gv * t.w00);
        if (live && t.w01 != 0.0f) atomicAdd(pl + (t.o01 >> 2), gv * t.w01);
        if (live && t.w10 != 0.0f) atomicAdd(pl + (t.o10 >> 2), gv * t.w10);
        if (live && t.w11 != 0.0f) atomicAdd(pl + (t.o11 >> 2), gv * t.w11);
      }
      if (grad_mapping != nullptr) {  // wave-uniform
        const float ggx = gix * (0.5f * (float)W);
        const float ggy = giy * (0.5f * (float)H);
        float part[2 * K3P];
#pragma unroll
        for (int k = 0; k < K3P; ++k) {
          part[2 * k] = bas[k] * ggx;
          part[2 * k + 1] = bas[k] * ggy;
        }
        float red = wave_transpose_reduce<2 * K3P>(part, lane);
        int idx = bitrev6(lane);
        if (idx < 2 * K3) atomicAdd(grad_mapping + ((int64_t)f * L + l) * K3 * 2 + idx, red);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// two-kernel backward (K3 == 19, L <= 17, 4 | W)
//
//   K1  pixel-major, all layers of a pixel in one thread: re-sample with derivatives, composite
//       backward, control-point gradient as an f32 MFMA contraction basis^T x grid-grad (per-tile
//       partial, summed by a tiny second kernel), and per (pixel, layer) one 16-byte RECORD
//       (grid x, grid y, a'_l, d loss / d s_l3) that is all the splat needs, plus a footprint
//       table per 8x16-pixel CELL: the bounding box of the source texels the cell's bilinear
//       footprints touch and an upper bound of its contribution magnitudes
//       (warp_composite_bwd_px16_kernel, warp_composite_bwd_px16.hip.h: 16x16 tiles, frames looped
//       in the workgroup, samples from an LDS image of each layer's footprint box; needs 4 | W).
//   K2  warp_composite_splat_kernel (warp_composite_splat.hip): one workgroup OWNS a 32x64-texel
//       tile of one layer's gradient plane, visits the cells whose box reaches it, re-derives the
//       taps from the records and sums them in a FIXED-POINT LDS image; plain stores, no global
//       atomics, no zero fill, bitwise reproducible.
//
// Hardware facts this is built around (measured on MI355X, tools_dev/*.hip):
//   * ds_add_f32 (LDS float atomic) retires ~3 cycles PER LANE (195 cycles per wave-instruction);
//     ds_add_u32 / ds_add_rtn_u32 / ds_wrxchg_rtn_b32 run at the ds_write_b32 rate (~5 cycles).
//     The scatter image is therefore 32-bit fixed point with a per-tile power-of-two scale chosen
//     so that no texel can overflow; integer sums are order-independent (bitwise reproducible).
//   * thousands of waves adding floats to the same few hundred addresses run ~14x below the
//     streaming atomic rate: the control-point gradient uses per-tile partials + a reduce.
//   * a pixel tile shares most of its footprint box with its neighbours (skew of the warp + the
//     1-texel bilinear overlap), and shared texels would need atomics; the scatter is therefore
//     organised by SOURCE tile (exclusive ownership), which is what the records buy.
// ---------------------------------------------------------------------------------------
// second stage of the control-point gradient: grad_mapping[f,l,k,c] += sum_tile partial.
// A workgroup sums 32 outputs of one frame: 8 thread groups take an eighth of the tiles each (32
// consecutive floats per tile: 128-byte segments), then the eight partial sums are added in a
// fixed order (deterministic).
constexpr int kRedOut = 32, kRedSlices = kBlock / kRedOut;

static __global__ __launch_bounds__(kBlock) void warp_composite_gmap_reduce_kernel(
    const float* __restrict__ gmap_partial, float* __restrict__ grad_mapping, int F, int L,
    int ntiles, int groups) {
  const int per = L * kGmapK3 * 2;
  const int64_t f = blockIdx.x / groups;
  const int o = (blockIdx.x % groups) * kRedOut + (threadIdx.x % kRedOut);
  const int slice = threadIdx.x / kRedOut;
  __shared__ float red[kRedSlices][kRedOut];
  float s0 = 0.0f, s1 = 0.0f;
  if (o < per) {
    const int t0 = (int)((int64_t)ntiles * slice / kRedSlices), t1 = (int)((int64_t)ntiles * (slice + 1) / kRedSlices);
    const float* src = gmap_partial + f * ntiles * per + o;
    int t = t0;
    for (; t + 1 < t1; t += 2) {
      s0 += src[(int64_t)t * per];
      s1 += src[(int64_t)(t + 1) * per];
    }
    if (t < t1) s0 += src[(int64_t)t * per];
  }
  red[slice][threadIdx.x % kRedOut] = s0 + s1;
  __syncthreads();
  if (slice == 0 && o < per) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kRedSlices; ++k) sum += red[k][threadIdx.x];
    grad_mapping[f * per + o] += sum;
  }
}

// K2 (compiled once, warp_composite_splat.hip)
void launch_splat(const float* rec, const float* grad_rgb, const int* cellbox,
                  const unsigned* cellbound, float* grad_layers, int F, int L, int H, int W,
                  hipStream_t st);

}  // namespace waldo

#include "warp_composite_fwd_lds.hip.h"
#ifdef WALDO_VARIANT_FWD_PIPE  // tools_dev/build_variant.py only: round 5's rejected experiment (tools_dev/dropped/)
#include "warp_composite_fwd_pipe.hip.h"
#endif
#include "warp_composite_bwd_px16.hip.h"

namespace waldo {

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
// frames per workgroup of the kernels that loop over frames: amortises the per-tile basis loads;
// prefer a chunk count divisible by the 8 XCDs (each chunk is pinned to one) while keeping enough
// workgroups to fill the chip several times over
static inline int chunk_frames(int F, int64_t ntiles) {
#ifdef WALDO_ABL_FPB  // timing-only sweep of the frames per workgroup
  return WALDO_ABL_FPB < F ? WALDO_ABL_FPB : F;
#endif
  for (int c = 8; c >= 2; --c) {
    const int chunks = (F + c - 1) / c;
    if (chunks % kXcds == 0 && (int64_t)chunks * ntiles >= 2048) return c;
  }
  return 1;
}

struct TileGeom {
  int ntx, nty, ntiles;
};

static inline TileGeom tile_geom(int H, int W, int rows) {
  TileGeom g;
  g.ntx = (W + kTileW - 1) / kTileW;
  g.nty = (H + rows - 1) / rows;
  g.ntiles = g.ntx * g.nty;
  return g;
}

// inv_kernel / src_pts != nullptr (K3 == 19, staged shapes only: checked by the C-ABI entry point):
// the mapping is computed inside the forward kernel and `mapping` is not read
#ifndef WALDO_FWD_NW
#define WALDO_FWD_NW 4
#endif
template <int LP, int K3P, bool EXK>
static void launch_fwd(const float* layers, const float* basis_t, const float* mapping,
                       const float* inv_kernel, const float* src_pts,
                       const float* occ, float* rgb, float* alpha, int F, int L, int H, int W,
                       int K3, float delta, hipStream_t st) {
  const TileGeom g = tile_geom(H, W, 4);
  const int fpb = chunk_frames(F, g.ntiles);
  const int nchunks = (F + fpb - 1) / fpb;
  // fewer chunks than a multiple of the 8 XCDs (short batches): the tiles of a chunk are cut into
  // bands so that every XCD gets work
  const int nbands = xcd_bands(nchunks);
  dim3 grid((unsigned)xcd_grid_banded(nchunks, nbands, g.ntiles, 1));
  if constexpr (EXK) {
    // LDS-staged sampling needs 16-byte-aligned rows and a 2x2 block inside the layer
    if (!debug_option(WALDO_DEBUG_FWD_PLAIN) && staged_eligible(H, W)) {
      // WALDO_FWD_NW: wavefronts per workgroup of the staged forward, 4 (16 x 16 tiles) or 8 (16 x 32; up to 12 layers)
      constexpr int NW = (WALDO_FWD_NW == 8 && LP <= 12) ? 8 : 4;
      constexpr int TW = kLdsTile * NW / 4;
      const int ntx16 = (W + TW - 1) / TW, nt16 = ntx16 * ((H + kLdsTile - 1) / kLdsTile);
      dim3 grid16((unsigned)xcd_grid_banded(nchunks, nbands, nt16, 1));
      auto go = [&](auto exl, auto fold) {
        constexpr bool EXL = decltype(exl)::value, FOLD = decltype(fold)::value;
        hipLaunchKernelGGL((warp_composite_fwd_lds_kernel<LP, EXL, FOLD, NW>), grid16, dim3(NW * kWave), 0, st, layers,
                           basis_t, mapping, inv_kernel, src_pts, occ, rgb, alpha, F, L, H, W, fpb, ntx16, nt16,
                           nchunks, nbands, delta);
      };
      using T = std::true_type;
      using N = std::false_type;
#ifdef WALDO_VARIANT_FWD_PIPE
      if constexpr (NW == 4 && LP >= 2 && LP <= 8) {
        // a VARIANT build (tools_dev/build_variant.py NAME -DWALDO_VARIANT_FWD_PIPE): round 5's software-pipelined frame
        // loop (tools_dev/dropped/warp_composite_fwd_pipe.hip.h; same bits, 2.5 % slower at the headline shape) takes
        // every launch it can serve -- the product library does not contain it
        if (L == LP) {
          if (src_pts != nullptr)
            hipLaunchKernelGGL((warp_composite_fwd_pipe_kernel<LP, true>), grid16, dim3(NW * kWave), 0, st, layers,
                               basis_t, mapping, inv_kernel, src_pts, occ, rgb, alpha, F, H, W, fpb, ntx16, nt16,
                               nchunks, nbands, delta);
          else
            hipLaunchKernelGGL((warp_composite_fwd_pipe_kernel<LP, false>), grid16, dim3(NW * kWave), 0, st, layers,
                               basis_t, mapping, inv_kernel, src_pts, occ, rgb, alpha, F, H, W, fpb, ntx16, nt16,
                               nchunks, nbands, delta);
          return;
        }
      }
#endif
      if (src_pts != nullptr) {
        if (L == LP) go(T{}, T{}); else go(N{}, T{});
      } else {
        if (L == LP) go(T{}, N{}); else go(N{}, N{});
      }
      return;
    }
  }
  if (L == LP)
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, true, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx, g.ntiles,
                       nchunks, nbands, delta);
  else
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, false, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx, g.ntiles,
                       nchunks, nbands, delta);
}

template <int LP, int K3P>
static void launch_bwd(const float* layers, const float* basis_t, const float* mapping,
                       const float* occ, const float* grad_rgb, const float* grad_alpha,
                       float* grad_layers, float* grad_mapping, float* grad_occ, int F, int L,
                       int H, int W, int K3, float delta, hipStream_t st) {
  const int64_t HW = (int64_t)H * W;
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  dim3 grid(tiles, F);
  hipLaunchKernelGGL((warp_composite_bwd_kernel<LP, K3P>), grid, dim3(kBlock), 0, st, layers,
                     basis_t, mapping, occ, grad_rgb, grad_alpha, grad_layers, grad_mapping,
                     grad_occ, F, L, H, W, K3, delta);
}

// two-kernel backward; workspace = Bwd2Layout
//
// One pass over all frames: K1 -> reduce -> K2.  (Measured and dropped in round 2: passes of a few
// frames over one small workspace, so that the records K1 writes are still in the 256 MiB Infinity
// Cache when K2 reads them -- back to back the short launches lose more to ramp-up and drain than
// the records cost, and with K2 of pass c on a side stream beside K1 of pass c + 1 the two kernels
// only slow each other down; a timing-only build whose records never leave the cache bounds the
// prize at 9 % of the backward.  DESIGN.md section 4.)
template <int LP>
static void launch_bwd2(const float* layers, const float* basis_t, const float* mapping,
                        const float* occ, const float* grad_rgb, const float* grad_alpha,
                        void* workspace, float* grad_layers, float* grad_mapping, float* grad_occ,
                        int F, int L, int H, int W, float delta, hipStream_t st) {
  const Bwd2Layout lo = bwd2_layout(F, L, H, W);
  char* ws = reinterpret_cast<char*>(workspace);
  int* boxes = reinterpret_cast<int*>(ws);
  unsigned* bounds = reinterpret_cast<unsigned*>(ws + lo.box_bytes);
  float4* rec = reinterpret_cast<float4*>(ws + lo.box_bytes + lo.bound_bytes);
  float* part = grad_mapping == nullptr
                    ? nullptr
                    : reinterpret_cast<float*>(ws + lo.box_bytes + lo.bound_bytes + lo.rec_bytes);
  using T = std::true_type;
  using N = std::false_type;
  const int ntiles = lo.ntiles16;
  const int fpb = chunk_frames(F, ntiles);
  const int nchunks = (F + fpb - 1) / fpb;
  // fewer chunks than a multiple of the 8 XCDs: the tiles of a chunk are cut into bands
  const int nbands = xcd_bands(nchunks);
  dim3 grid((unsigned)xcd_grid_banded(nchunks, nbands, ntiles, 1));
  auto go = [&](auto exl, auto gocc) {
    constexpr bool EXL = decltype(exl)::value, GOCC = decltype(gocc)::value;
    hipLaunchKernelGGL((warp_composite_bwd_px16_kernel<LP, EXL, GOCC>), grid, dim3(kBlock), 0, st, layers,
                       basis_t, mapping, occ, grad_rgb, grad_alpha, rec, boxes, bounds, part,
                       grad_occ, F, L, H, W, fpb, lo.ntx16, ntiles, nchunks, nbands, lo.ncx, lo.ncells,
                       delta);
  };
  if (L == LP) {
    if (grad_occ != nullptr) go(T{}, T{}); else go(T{}, N{});
  } else {
    if (grad_occ != nullptr) go(N{}, T{}); else go(N{}, N{});
  }
  if (part != nullptr) {
    const int groups = (int)((gmap_partial_floats(L) + kRedOut - 1) / kRedOut);
    hipLaunchKernelGGL(warp_composite_gmap_reduce_kernel, dim3((unsigned)((int64_t)F * groups)), dim3(kBlock),
                       0, st, part, grad_mapping, F, L, ntiles, groups);
  }
  launch_splat(reinterpret_cast<const float*>(rec), grad_rgb, boxes, bounds, grad_layers, F, L, H, W, st);
}

}  // namespace waldo
