// Fused WIF hot path for gfx950: TPS grid synthesis -> bilinear backward warp of every
// 4-channel layer -> occlusion / soft-alpha composite (LVD.reduce_comp), forward and backward.
//
// Reference behaviour restated (paths relative to the reference root):
//   models/modules/warp.py:49-55      TPSWarp.forward           grid = basis @ (K^-1 @ [pts;0])
//   torch F.grid_sample defaults      bilinear / zeros / align_corners=False
//   models/nets/lvd.py:100-114        LVD.reduce_comp           a'_j = a_j prod_i (1 - a_i occ_ij)
//
// Kernels in this file
//   warp_composite_fwd_kernel   one thread per output pixel, 4x64-pixel tiles (one 64-pixel row
//                               per wavefront), frames looped inside the workgroup so that the
//                               K3 TPS basis values of a pixel stay in registers; branch-free
//                               layer loop so that all 16*L tap loads of a pixel are in flight
//                               together; mapping / occ are wave-uniform (scalar cache).
//   warp_composite_bbox_kernel  backward pre-pass: per (frame, layer, tile) bounding box of the
//                               source texels the tile's bilinear footprints touch.
//   warp_composite_bwd2_kernel  tiled backward (L <= 8, K3 == 19): phase 1 per pixel (all layers:
//                               re-sample with derivatives, composite backward), control-point
//                               gradient as an f32 MFMA contraction basis^T x grid-grad over the
//                               tile's pixels; phase 2 per layer: scatter-add of the tile's tap
//                               contributions into an LDS image of the bounding box, flushed with
//                               PLAIN stores where no other tile's box covers the texel and float
//                               atomics only on the shared rims.
//   warp_composite_bwd_kernel   generic backward (any L <= 32, K3 <= 32): per-tap global atomics.
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

// map: (K3,2) wave-uniform.  Sequential fmaf chain in k: the ONE definition of the grid that
// forward, bounding-box pre-pass and backward share (bit-identical coordinates).
template <int K3P, bool EXK>
__device__ __forceinline__ void tps_eval(const float (&bas)[K3P], const float* __restrict__ map,
                                         int K3, float& gx, float& gy) {
  gx = 0.0f;
  gy = 0.0f;
#pragma unroll
  for (int k = 0; k < K3P; ++k) {
    const int kc = EXK ? k : min(k, K3 - 1);  // bas[k] == 0 beyond K3
    gx = fmaf(bas[k], map[2 * kc], gx);
    gy = fmaf(bas[k], map[2 * kc + 1], gy);
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
    int ntx, int ntiles, int nchunks) {
  const int L = EXL ? LP : Lrt;
  const int K3 = EXK ? K3P : K3rt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  int chunk, tile;  // frame chunk pinned to an XCD: the tiles of a frame share one L2
  if (!xcd_decode(blockIdx.x, nchunks, ntiles, chunk, tile)) return;
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
      float gx, gy;
      tps_eval<K3P, EXK>(bas, mapping + ((int64_t)f * L + lc) * K3 * 2, K3, gx, gy);
      const Taps t = make_taps(gx, gy, H, W);
      const float* base = layers + ((int64_t)f * L + lc) * 4 * HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) s[l][c] = tap_sample(base + c * HW, t);
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
    float* __restrict__ grad_occ, int F, int L, int H, int W, int K3) {
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
      tps_eval<K3P, false>(bas, mapping + ((int64_t)f * L + l) * K3 * 2, K3, gxs[l], gys[l]);
      Taps t = make_taps(gxs[l], gys[l], H, W);
      const float* base = layers + ((int64_t)f * L + l) * 4 * HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) s[l][c] = tap_sample_d(base + c * HW, t, dsx[l][c], dsy[l][c]);
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
      Taps t = make_taps(gxs[l], gys[l], H, W);
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
// two-kernel backward (K3 == 19, L <= 8)
//
//   K1  pixel-major, all layers of a pixel in one thread: re-sample with derivatives, composite
//       backward, control-point gradient as an f32 MFMA contraction basis^T x grid-grad (per-tile
//       partial, summed by a tiny second kernel), and per (pixel, layer) one 16-byte RECORD
//       (grid x, grid y, a'_l, d loss / d s_l3) that is all the splat needs, plus a footprint
//       table per 8x16-pixel CELL: the bounding box of the source texels the cell's bilinear
//       footprints touch and an upper bound of its contribution magnitudes.  Two variants:
//       warp_composite_bwd_px16_kernel (warp_composite_bwd_px16.hip.h; 16x16 tiles, samples from an
//       LDS image of each layer's footprint box, owns its cells) when 4 | W, else
//       warp_composite_bwd_px_kernel below (4x64 tiles, per-tap gathers, cell table by atomics).
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
#ifndef WALDO_PX_WAVES
#define WALDO_PX_WAVES 4
#endif
constexpr int kPxWaves = WALDO_PX_WAVES;       // wavefronts per workgroup of K1 (4: two workgroups
                                               // per CU at 256 VGPRs overlap each other's phases)
constexpr int kPxThreads = kPxWaves * kWave;
static_assert(kPxRows == kPxWaves, "tile = kPxWaves rows x 64 columns, one pixel per thread");
#ifndef WALDO_PX_GROUP
#define WALDO_PX_GROUP 4
#endif
#ifndef WALDO_PX_WPE
#define WALDO_PX_WPE 3  // waves per SIMD the pixel kernel is compiled for (168 VGPRs: no scratch)
#endif
constexpr int kPxGroup = WALDO_PX_GROUP;       // layers whose tap loads are in flight together
constexpr int kPxPix = kPxRows * kTileW;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using short2_ = __attribute__((ext_vector_type(2))) short;

__device__ __forceinline__ int pk_min(int a, int b) {
  short2_ x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  short2_ r = __builtin_elementwise_min(x, y);
  int o;
  __builtin_memcpy(&o, &r, 4);
  return o;
}

// Reductions over each group of 16 consecutive lanes with DPP row rotations (row_ror:8/4/2/1):
// plain VALU moves, no LDS crossbar (ds_bpermute) and no lgkmcnt waits.  A rotation is not an
// xor exchange, but min / + over all 16 lanes only needs every lane to meet every other once.
template <int ROR>
__device__ __forceinline__ int row_ror_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x120 + ROR, 0xf, 0xf, false);
}

__device__ __forceinline__ int group16_pk_min(int v) {
  v = pk_min(v, row_ror_i<8>(v));
  v = pk_min(v, row_ror_i<4>(v));
  v = pk_min(v, row_ror_i<2>(v));
  v = pk_min(v, row_ror_i<1>(v));
  return v;
}

__device__ __forceinline__ float group16_sum(float v) {
  v += __int_as_float(row_ror_i<8>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<4>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<2>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<1>(__float_as_int(v)));
  return v;
}

template <int LP, bool EXL, bool GOCC>
__global__ __launch_bounds__(kPxThreads, GOCC ? 2 : WALDO_PX_WPE) void warp_composite_bwd_px_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ,
    const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
    float4* __restrict__ rec, int* __restrict__ cellbox, unsigned* __restrict__ cellbound,
    float* __restrict__ gmap_partial, float* __restrict__ grad_occ, int F, int Lrt, int H, int W,
    int ntx, int ntiles, int ncx, int ncells) {
  constexpr int K3 = kGmapK3;
  constexpr int NC = 2 * LP;                 // columns of the grid-gradient matrix (layer, xy)
  constexpr int NT = (NC + 15) / 16;         // 16-column MFMA tiles
  constexpr int GGC = NT * 16;               // padded column count of gg
  const int L = EXL ? LP : Lrt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  int f, tile;  // frame pinned to an XCD
  if (!xcd_decode(blockIdx.x, F, ntiles, f, tile)) return;

  // LDS, one array, column-major per pixel with pitch kPxPix + 1 (bank = (row + pixel) mod 32:
  // conflict-free both for a wave writing its 64 pixels of one row and for the MFMA operand reads):
  //   rows 0 .. 4*LP-1   the parked tap derivatives (d s_rgb.g / dx, d s_a / dx, .. / dy) of every
  //                      layer -- they are only needed after the composite, and keeping them in
  //                      registers is what pushed this kernel to 256 VGPRs / 2 waves per SIMD;
  //   rows 0 .. GGC-1    afterwards: the grid gradients gg (MFMA B operand);
  //   then               the per-wave MFMA accumulators.
  constexpr int PP1 = kPxPix + 1;
  constexpr int kParkRows = 4 * LP > GGC ? 4 * LP : GGC;
  constexpr int kAccFloats = kPxWaves * 2 * NT * 256;
  constexpr int kTFloats = kPxWaves * kWave * (GGC + 1);  // per-wave transposition slices (TPS)
  constexpr int kLds0 = kParkRows * PP1 > kAccFloats ? kParkRows * PP1 : kAccFloats;
  constexpr int kLdsFloats = kLds0 > kTFloats ? kLds0 : kTFloats;
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* gg = lds;
  const int pix = threadIdx.x;

  const PixelMap pm = pixel_of(tile, wave, lane, H, W, kPxRows, ntx);
  const float livef = pm.live ? 1.0f : 0.0f;
  const int64_t p = pm.p;
  const float* oc = occ + (int64_t)f * L * L;
  float gxs[LP], gys[LP], ap[LP];
  {
    // (A) TPS grid of every layer on the matrix pipe: for each group of 16 pixels of this wave's
    // row, D[pixel][(layer, xy)] = sum_k basis[pixel][k] * mapping[k][(layer, xy)] with
    // v_mfma_f32_16x16x4_f32 (exact f32 fma chain in k order -- the same numbers as the scalar
    // chain of tps_eval).  Nothing goes through SGPRs (the scalar version kept 38*L mapping values
    // there and spilled hundreds of them to VGPR lanes) and 38*L VALU fmas per pixel disappear.
    // The accumulators are transposed through this wave's slice of LDS.
    constexpr int KS = (K3 + 3) / 4;  // k steps of 4
    const int arow = lane & 15, kk = lane >> 4;
    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[g][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk;
      float bv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 16 + arow, l = col >> 1;
        const float m = (mapping + (int64_t)f * L * K3 * 2)[(min(l, L - 1) * K3 + min(k, K3 - 1)) * 2 + (col & 1)];
        bv[nt] = (k < K3 && l < L) ? m : 0.0f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const PixelMap pq = pixel_of(tile, wave, 16 * g + arow, H, W, kPxRows, ntx);
        // 32-bit byte offset from the uniform base: K3 * HW * 4 < 2^32 is checked by the launcher
        const float bs = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + (uint32_t)pq.p) * 4u);
        const float av = (k < K3) ? bs : 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[g][nt], 0, 0, 0);
      }
    }
    // D[row = (lane>>4)*4 + r][col = lane&15] of group g  ->  T[pixel][col], pitch GGC + 1
    constexpr int TP = GGC + 1;
    float* T = lds + wave * (kWave * TP);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * g + kk * 4 + r) * TP + nt * 16 + arow] = acc[g][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      gxs[l] = T[lane * TP + 2 * l];
      gys[l] = T[lane * TP + 2 * l + 1];
    }
  }
  __syncthreads();  // the transposition slices are inside the park region written below
  const float g0 = grad_rgb[(int64_t)f * 3 * HW + p] * livef;
  const float g1 = grad_rgb[(int64_t)f * 3 * HW + HW + p] * livef;
  const float g2 = grad_rgb[(int64_t)f * 3 * HW + 2 * HW + p] * livef;
  float a[LP], G[LP];
  // footprint-table cell of this lane's 16-pixel group (8 rows x 16 columns of pixels)
  const int tx = tile % ntx, ty = tile / ntx;
  const int prow = ty * kPxRows + wave, pcol = tx * kTileW + lane;
  const int cell = (prow / kCellRows) * ncx + min(pcol, W - 1) / kCellCols;
  const bool leader = (lane & (kCellCols - 1)) == 0;
  const float gmax = fmaxf(fabsf(g0), fmaxf(fabsf(g1), fabsf(g2)));
  // (B) taps with derivatives; footprint box of every layer.  (No sched_barrier grouping here:
  // with the cross-lane box reduction in the loop, hipcc 7.2 mis-schedules the alpha-gradient
  // chain across a sched_barrier -- caught by the L == 8 parity tests.)
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    const int lc = EXL ? l : min(l, L - 1);
    const Taps t = make_taps(gxs[l], gys[l], H, W);
    const float* base = layers + ((int64_t)f * L + lc) * 4 * HW;
    float sx[4], sy[4], sv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) sv[c] = tap_sample_d(base + c * HW, t, sx[c], sy[c]);
    const bool pad = !EXL && l >= L;
    a[l] = pad ? 0.0f : (sv[3] + 1.0f) * 0.5f;
    G[l] = g0 * (sv[0] + 1.0f) + g1 * (sv[1] + 1.0f) + g2 * (sv[2] + 1.0f);
    lds[(4 * l + 0) * PP1 + pix] = fmaf(g2, sx[2], fmaf(g1, sx[1], g0 * sx[0]));
    lds[(4 * l + 1) * PP1 + pix] = sx[3];
    lds[(4 * l + 2) * PP1 + pix] = fmaf(g2, sy[2], fmaf(g1, sy[1], g0 * sy[0]));
    lds[(4 * l + 3) * PP1 + pix] = sy[3];
    if (grad_alpha != nullptr && !pad)
      G[l] = fmaf(2.0f * livef, grad_alpha[((int64_t)f * L + lc) * HW + p], G[l]);
    // a wave can have 63 vector-memory operations outstanding: issue the 16 tap loads of
    // kPxGroup layers together, and make the next group's addresses depend on this group's
    // results so that the registers of at most one group of loads are live at a time
    if ((l % kPxGroup) == kPxGroup - 1 && l + 1 < LP)
      asm volatile("" : "+v"(gxs[l + 1]), "+v"(gys[l + 1]) : "v"(a[l]), "v"(G[l]));
    if (!pad) {  // compile-time for exact L
      // in-range corner of the footprint, as (x, y) packed in 16+16 bits; lanes without any
      // in-range tap carry the neutral element.  Stored negated for the upper corner so that
      // both reduce with a minimum.
      const bool anyx = (t.vx0 + t.vx1) > 0.0f, anyy = (t.vy0 + t.vy1) > 0.0f;
      const bool has = pm.live && anyx && anyy;
      const int xa = t.vx0 > 0.0f ? t.x0 : t.x0 + 1, xb = t.vx1 > 0.0f ? t.x0 + 1 : t.x0;
      const int ya = t.vy0 > 0.0f ? t.y0 : t.y0 + 1, yb = t.vy1 > 0.0f ? t.y0 + 1 : t.y0;
      int lo = has ? (int)(((unsigned)ya << 16) | ((unsigned)xa & 0xffffu)) : 0x7fff7fff;
      int hi = has ? (int)(((unsigned)(-yb) << 16) | ((unsigned)(-xb) & 0xffffu)) : 0x7fff7fff;
      lo = group16_pk_min(lo);
      hi = group16_pk_min(hi);
      if (leader && lo != 0x7fff7fff) {
        int* bb = cellbox + (((int64_t)f * L + l) * ncells + cell) * 4;
        atomicMin(bb + 0, (int)(short)(lo & 0xffff));   // x min
        atomicMin(bb + 1, (int)(short)(hi & 0xffff));   // -(x max)
        atomicMin(bb + 2, lo >> 16);                    // y min
        atomicMin(bb + 3, hi >> 16);                    // -(y max)
      }
    }
  }
  a[0] = 1.0f;
  float ga[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) ga[l] = 0.0f;
#pragma unroll
  for (int j = 0; j < LP; ++j) {
    const int jc = EXL ? j : min(j, L - 1);
    float tfac[LP], ex[LP];
    float pre = 1.0f;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      const int ic = EXL ? i : min(i, L - 1);
      tfac[i] = 1.0f - a[i] * oc[ic * L + jc];
      ex[i] = pre;
      pre *= tfac[i];
    }
    float suf = 1.0f;
#pragma unroll
    for (int i = LP - 1; i >= 0; --i) {
      ex[i] *= suf;
      suf *= tfac[i];
    }
    ap[j] = a[j] * pre;      // 0 for padding layers
    const float gap = G[j];  // d loss / d a'_j
    ga[j] = fmaf(gap, pre, ga[j]);
    const float gaj = gap * a[j];
    float gocc[LP];
#pragma unroll
    for (int m = 0; m < LP; ++m) {
      const int mc = EXL ? m : min(m, L - 1);
      ga[m] = fmaf(-gaj * oc[mc * L + jc], ex[m], ga[m]);
      gocc[m] = -gaj * a[m] * ex[m];
    }
    if (GOCC && (EXL || j < L)) {  // compile-time: the reduction costs ~60 registers
      const float redv = wave_transpose_reduce<LP>(gocc, lane);
      const int m = bitrev6(lane);
      if (m < L) atomicAdd(grad_occ + (int64_t)f * L * L + m * L + j, redv);
    }
  }
  // a_m = (s_m3 + 1)/2 for m >= 1; a_0 is the constant 1.  Second half of the records, cell
  // bounds, and the grid gradient of every layer (read this thread's parked derivatives, then
  // overwrite the same LDS column with gg -- no other thread touches this column).
  float ggx[LP], ggy[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    const bool pad = !EXL && l >= L;
    const float gsa = (l >= 1 && !pad) ? 0.5f * ga[l] : 0.0f;
    const float dxr = lds[(4 * l + 0) * PP1 + pix], dxa = lds[(4 * l + 1) * PP1 + pix];
    const float dyr = lds[(4 * l + 2) * PP1 + pix], dya = lds[(4 * l + 3) * PP1 + pix];
    ggx[l] = fmaf(gsa, dxa, ap[l] * dxr) * (0.5f * (float)W);
    ggy[l] = fmaf(gsa, dya, ap[l] * dyr) * (0.5f * (float)H);
    if (!pad) {
      // |tap contribution| <= max(|a'_l| max_c |g_c|, |g_alpha|): bilinear weights are <= 1.  The
      // table keeps the largest 16-pixel row sum of the cell (atomicMax on the bits of a
      // non-negative float is order-independent: the splat's fixed-point scale is deterministic)
      const float bnd = group16_sum(pm.live ? fmaxf(fabsf(ap[l]) * gmax, fabsf(gsa)) : 0.0f);
      if (leader && prow < H && pcol < W)
        atomicMax(cellbound + ((int64_t)f * L + l) * ncells + cell, __float_as_uint(bnd));
      if (pm.live) rec[((int64_t)f * L + l) * HW + p] = make_float4(gxs[l], gys[l], ap[l], gsa);
    }
  }
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    gg[(2 * l) * PP1 + pix] = ggx[l];
    gg[(2 * l + 1) * PP1 + pix] = ggy[l];
  }
#pragma unroll
  for (int c = NC; c < GGC; ++c) gg[c * PP1 + pix] = 0.0f;
  __syncthreads();  // gg rows are complete

  // ------------------------------------------- control-point gradient: basis^T x gg on the MFMA
  // D[k][col] = sum_pix basis[k][pix] * gg[pix][col]; v_mfma_f32_16x16x4_f32: A[row=lane&15]
  // [kk=lane>>4], B[kk=lane>>4][col=lane&15], D[row=(lane>>4)*4+reg][col=lane&15].  Each wave
  // contracts the pixels it produced; the 8 wave results are summed through LDS in a fixed order
  // and stored as this tile's partial (no atomics, deterministic).
  if (gmap_partial != nullptr) {
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int arow = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      const int pl = 4 * s4 + kk;  // column of the contracted pixel inside this wave's row
      const PixelMap pq = pixel_of(tile, wave, pl, H, W, kPxRows, ntx);
      const int px = wave * kWave + pl;
      float av[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int k = mt * 16 + arow;
        const float v = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + (uint32_t)pq.p) * 4u);
        av[mt] = (k < K3) ? v : 0.0f;  // dead pixels carry gg == 0
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = gg[(nt * 16 + arow) * PP1 + px];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv, acc[mt][nt], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done reading gg: reuse its bytes for the accumulators
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          lds[((wave * 2 + mt) * NT + nt) * 256 + r * 64 + lane] = acc[mt][nt][r];
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * NT * 256; o += kPxThreads) {
      float sum = 0.0f;
#pragma unroll
      for (int w = 0; w < kPxWaves; ++w) sum += lds[w * 2 * NT * 256 + o];  // fixed order
      const int mt = o / (NT * 256), nt = (o / 256) % NT, r = (o >> 6) & 3, ln = o & 63;
      const int k = mt * 16 + (ln >> 4) * 4 + r;
      const int col = nt * 16 + (ln & 15);
      const int l = col >> 1;
      if (k < K3 && l < L)
        gmap_partial[((int64_t)f * ntiles + tile) * gmap_partial_floats(L) + ((int64_t)l * K3 + k) * 2 +
                     (col & 1)] = sum;
    }
  }
}

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
#include "warp_composite_bwd_px16.hip.h"

namespace waldo {

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
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

template <int LP, int K3P, bool EXK>
static void launch_fwd(const float* layers, const float* basis_t, const float* mapping,
                       const float* occ, float* rgb, float* alpha, int F, int L, int H, int W,
                       int K3, hipStream_t st) {
  const TileGeom g = tile_geom(H, W, 4);
  // frames per workgroup: amortises the basis loads; prefer a chunk count divisible by the 8 XCDs
  // (each chunk is pinned to one) while keeping >= 2048 workgroups
  int fpb = 1;
  for (int c = 8; c >= 2; --c) {
    const int chunks = (F + c - 1) / c;
    if (chunks % kXcds == 0 && (int64_t)chunks * g.ntiles >= 2048) {
      fpb = c;
      break;
    }
  }
  const int nchunks = (F + fpb - 1) / fpb;
  dim3 grid((unsigned)xcd_grid(nchunks, g.ntiles));
  if constexpr (EXK) {
    // LDS-staged sampling needs 16-byte-aligned rows and a 2x2 block inside the layer
    static const bool plain = getenv("WALDO_FWD_PLAIN") != nullptr;  // A/B switch for testing
    if (!plain && staged_eligible(H, W)) {
      const int ntx16 = (W + kLdsTile - 1) / kLdsTile, nt16 = ntx16 * ((H + kLdsTile - 1) / kLdsTile);
      dim3 grid16((unsigned)xcd_grid(nchunks, nt16));
      if (L == LP)
        hipLaunchKernelGGL((warp_composite_fwd_lds_kernel<LP, true>), grid16, dim3(kBlock), 0, st, layers,
                           basis_t, mapping, occ, rgb, alpha, F, L, H, W, fpb, ntx16, nt16, nchunks);
      else
        hipLaunchKernelGGL((warp_composite_fwd_lds_kernel<LP, false>), grid16, dim3(kBlock), 0, st, layers,
                           basis_t, mapping, occ, rgb, alpha, F, L, H, W, fpb, ntx16, nt16, nchunks);
      return;
    }
  }
  if (L == LP)
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, true, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx, g.ntiles,
                       nchunks);
  else
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, false, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx, g.ntiles,
                       nchunks);
}

template <int LP, int K3P>
static void launch_bwd(const float* layers, const float* basis_t, const float* mapping,
                       const float* occ, const float* grad_rgb, const float* grad_alpha,
                       float* grad_layers, float* grad_mapping, float* grad_occ, int F, int L,
                       int H, int W, int K3, hipStream_t st) {
  const int64_t HW = (int64_t)H * W;
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  dim3 grid(tiles, F);
  hipLaunchKernelGGL((warp_composite_bwd_kernel<LP, K3P>), grid, dim3(kBlock), 0, st, layers,
                     basis_t, mapping, occ, grad_rgb, grad_alpha, grad_layers, grad_mapping,
                     grad_occ, F, L, H, W, K3);
}

// two-kernel backward; workspace = Bwd2Layout
template <int LP>
static void launch_bwd2(const float* layers, const float* basis_t, const float* mapping,
                        const float* occ, const float* grad_rgb, const float* grad_alpha,
                        void* workspace, float* grad_layers, float* grad_mapping, float* grad_occ,
                        int F, int L, int H, int W, hipStream_t st) {
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
  static const bool gather = getenv("WALDO_BWD_GATHER") != nullptr;  // A/B switch for testing
  const bool staged = staged_eligible(H, W) && !gather;
  int ntiles;
  if (staged) {
    // LDS-staged pixel kernel: owns its cells, writes their table entries itself
    ntiles = lo.ntiles16;
    dim3 grid((unsigned)xcd_grid(F, ntiles));
    auto go = [&](auto exl, auto gocc) {
      constexpr bool EXL = decltype(exl)::value, GOCC = decltype(gocc)::value;
      hipLaunchKernelGGL((warp_composite_bwd_px16_kernel<LP, EXL, GOCC>), grid, dim3(kBlock), 0, st, layers,
                         basis_t, mapping, occ, grad_rgb, grad_alpha, rec, boxes, bounds, part,
                         grad_occ, F, L, H, W, lo.ntx16, ntiles, lo.ncx, lo.ncells);
    };
    if (L == LP) {
      if (grad_occ != nullptr) go(T{}, T{}); else go(T{}, N{});
    } else {
      if (grad_occ != nullptr) go(N{}, T{}); else go(N{}, N{});
    }
  } else {
    // boxes are (min x, -max x, min y, -max y): every component starts at a large positive value;
    // bounds start at +0.0f
    ntiles = lo.ntiles;
    (void)hipMemsetAsync(boxes, 0x7f, (size_t)lo.box_bytes, st);
    (void)hipMemsetAsync(bounds, 0, (size_t)lo.bound_bytes, st);
    dim3 grid((unsigned)xcd_grid(F, ntiles));
    auto go = [&](auto exl, auto gocc) {
      constexpr bool EXL = decltype(exl)::value, GOCC = decltype(gocc)::value;
      hipLaunchKernelGGL((warp_composite_bwd_px_kernel<LP, EXL, GOCC>), grid, dim3(kPxThreads), 0, st,
                         layers, basis_t, mapping, occ, grad_rgb, grad_alpha, rec, boxes, bounds, part,
                         grad_occ, F, L, H, W, lo.ntx, ntiles, lo.ncx, lo.ncells);
    };
    if (L == LP) {
      if (grad_occ != nullptr) go(T{}, T{}); else go(T{}, N{});
    } else {
      if (grad_occ != nullptr) go(N{}, T{}); else go(N{}, N{});
    }
  }
  if (part != nullptr) {
    const int groups = (int)((gmap_partial_floats(L) + kRedOut - 1) / kRedOut);
    hipLaunchKernelGGL(warp_composite_gmap_reduce_kernel, dim3((unsigned)((int64_t)F * groups)), dim3(kBlock),
                       0, st, part, grad_mapping, F, L, ntiles, groups);
  }
  launch_splat(reinterpret_cast<const float*>(rec), grad_rgb,
               boxes, bounds, grad_layers, F, L, H, W, st);
}

}  // namespace waldo
