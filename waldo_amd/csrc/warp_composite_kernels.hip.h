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
#include <type_traits>

#include "waldo_common.hip.h"

namespace waldo {

constexpr int kMaxLayers = 32;
constexpr int kMaxK3 = 32;
constexpr int kTileW = 64;  // one wavefront = 64 consecutive pixels of a row
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
    int ntx) {
  const int L = EXL ? LP : Lrt;
  const int K3 = EXK ? K3P : K3rt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const PixelMap pm = pixel_of(blockIdx.x, wave, lane, H, W, 4, ntx);
  const int64_t p = pm.p;
  float bas[K3P];
  load_basis<K3P, EXK>(bas, basis_t, HW, p, K3);

  const int f0 = blockIdx.y * frames_per_block;
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
// tiled backward: bounding-box pre-pass
//   bbox[((f*L + l)*ntiles + tile)] = (x_min, x_max, y_min, y_max) inclusive, over the IN-RANGE
//   taps of the tile's pixels; empty = (1, 0, 1, 0).
// ---------------------------------------------------------------------------------------
template <int LP, int K3P, bool EXL, bool EXK>
__global__ __launch_bounds__(kBlock) void warp_composite_bbox_kernel(
    const float* __restrict__ basis_t, const float* __restrict__ mapping, int4* __restrict__ bbox,
    int F, int Lrt, int H, int W, int K3rt, int rows_per_tile, int ntx, int ntiles) {
  const int L = EXL ? LP : Lrt;
  const int K3 = EXK ? K3P : K3rt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, f = blockIdx.y;
  __shared__ int red[4][LP][4];
  int xmin[LP], xmax[LP], ymin[LP], ymax[LP];
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    xmin[l] = ymin[l] = 1 << 30;
    xmax[l] = ymax[l] = -(1 << 30);
  }
  for (int r = wave; r < rows_per_tile; r += 4) {
    const PixelMap pm = pixel_of(tile, r, lane, H, W, rows_per_tile, ntx);
    float bas[K3P];
    load_basis<K3P, EXK>(bas, basis_t, HW, pm.p, K3);
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int lc = EXL ? l : min(l, L - 1);
      float gx, gy;
      tps_eval<K3P, EXK>(bas, mapping + ((int64_t)f * L + lc) * K3 * 2, K3, gx, gy);
      const Taps t = make_taps(gx, gy, H, W);
      if (pm.live) {
        const bool anyx = (t.vx0 + t.vx1) > 0.0f, anyy = (t.vy0 + t.vy1) > 0.0f;
        if (anyx && anyy) {
          const int xa = t.vx0 > 0.0f ? t.x0 : t.x0 + 1, xb = t.vx1 > 0.0f ? t.x0 + 1 : t.x0;
          const int ya = t.vy0 > 0.0f ? t.y0 : t.y0 + 1, yb = t.vy1 > 0.0f ? t.y0 + 1 : t.y0;
          xmin[l] = min(xmin[l], xa);
          xmax[l] = max(xmax[l], xb);
          ymin[l] = min(ymin[l], ya);
          ymax[l] = max(ymax[l], yb);
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    const int a = wave_min_i(xmin[l]), b = wave_max_i(xmax[l]);
    const int c = wave_min_i(ymin[l]), d = wave_max_i(ymax[l]);
    if (lane == 0) {
      red[wave][l][0] = a;
      red[wave][l][1] = b;
      red[wave][l][2] = c;
      red[wave][l][3] = d;
    }
  }
  __syncthreads();
  if (threadIdx.x < L) {
    const int l = threadIdx.x;
    int a = 1 << 30, b = -(1 << 30), c = 1 << 30, d = -(1 << 30);
    for (int w = 0; w < 4; ++w) {
      a = min(a, red[w][l][0]);
      b = max(b, red[w][l][1]);
      c = min(c, red[w][l][2]);
      d = max(d, red[w][l][3]);
    }
    if (a > b || c > d) {
      a = 1; b = 0; c = 1; d = 0;
    }
    bbox[((int64_t)f * L + l) * ntiles + tile] = make_int4(a, b, c, d);
  }
}

// ---------------------------------------------------------------------------------------
// tiled backward (K3 == 19, L <= 8)
//
// Hardware facts this kernel is built around (measured on MI355X, tools_dev/lds_atomic_bench.hip):
//   * ds_add_f32 (LDS float atomic) retires ~3 cycles PER LANE (195 cycles per wave-instruction);
//     ds_add_u32 / ds_add_rtn_u32 / ds_wrxchg_rtn_b32 run at the ds_write_b32 rate (~5 cycles per
//     wave-instruction).  The per-layer scatter image is therefore accumulated in 32-bit FIXED
//     POINT with a per-(tile, layer) power-of-two scale chosen so that no texel can overflow;
//     integer sums are also order-independent, so texels owned by one workgroup are bitwise
//     reproducible.
//   * __syncthreads() waits for outstanding global stores / atomics (vmcnt(0)); the layer loop
//     uses a raw s_barrier behind an LDS-only wait so that a layer's flush overlaps the next
//     layer's scatter.
//   * global float atomics from thousands of waves into the same few hundred addresses run ~14x
//     below the streaming atomic rate: the control-point gradient is reduced inside the workgroup
//     and stored as a per-tile partial, summed by a second tiny kernel (deterministic).
// ---------------------------------------------------------------------------------------
#ifndef WALDO_BWD_PP
#define WALDO_BWD_PP 1
#endif
#ifndef WALDO_BWD_WAVES
#define WALDO_BWD_WAVES 8
#endif
constexpr int kBwdPP = WALDO_BWD_PP;        // pixel rows per thread
constexpr int kBwdWaves = WALDO_BWD_WAVES;  // wavefronts per workgroup = pixel rows per pass
constexpr int kBwdThreads = kBwdWaves * kWave;
constexpr int kBwdRows = kBwdWaves * kBwdPP;  // tile = kBwdRows x 64 pixels
constexpr int kBwdPix = kBwdRows * kTileW;
constexpr int kMaxTex = 1280;    // texels of one layer's bounding box kept in LDS (x4 channels)
constexpr int kMaxNb = 24;       // other tiles whose box intersects ours, per layer
constexpr int kGmapK3 = 19;

using f32x4 = __attribute__((ext_vector_type(4))) float;

// LDS-only barrier: waits for this wave's LDS traffic, not for its global stores / atomics
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// bytes of workspace per frame/layer/tile (host side uses the same numbers)
constexpr int64_t kBboxBytes = 16;
__host__ __device__ constexpr int64_t gmap_partial_floats(int L) { return (int64_t)L * kGmapK3 * 2; }

template <int LP, bool EXL, bool GOCC>
__global__ __launch_bounds__(kBwdThreads, kBwdThreads / 256) void warp_composite_bwd2_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ,
    const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
    const int4* __restrict__ bbox, float* __restrict__ gmap_partial,
    float* __restrict__ grad_layers, float* __restrict__ grad_occ, int F, int Lrt, int H, int W,
    int ntx, int ntiles) {
  constexpr int K3 = kGmapK3;
  constexpr int NC = 2 * LP;                 // columns of the grid-gradient matrix (layer, xy)
  constexpr int NT = (NC + 15) / 16;         // 16-column MFMA tiles
  constexpr int GGC = NT * 16;               // padded column count of gg
  constexpr int GGP = GGC + 1;               // odd LDS row pitch: conflict-free lane-strided writes
  const int L = EXL ? LP : Lrt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, f = blockIdx.y;

  // One LDS array (cdna guide: a second __shared__ object can de-pipeline the kernel).
  //   [0, kUnion)        phase 1: gg (grid gradients, [pixel][GGP]) then the per-wave MFMA
  //                      accumulators; phase 2: the two fixed-point scatter images
  //   then persistent:   neighbour boxes of every layer, their counts, per-wave bound partials
  constexpr int kGGFloats = kBwdPix * GGP;
  constexpr int kAccFloats = kBwdWaves * 2 * NT * 256;
  constexpr int kImgWords = 2 * 4 * kMaxTex;
  constexpr int kUnion0 = kGGFloats > kImgWords ? kGGFloats : kImgWords;
  constexpr int kUnion = kUnion0 > kAccFloats ? kUnion0 : kAccFloats;
  constexpr int kNbWords = LP * kMaxNb * 4;
  __shared__ __attribute__((aligned(16))) float lds[kUnion + kNbWords + LP + kBwdWaves * LP];
  float* gg = lds;
  int* img = reinterpret_cast<int*>(lds);
  int* nbl = reinterpret_cast<int*>(lds + kUnion);
  int* nbcount = nbl + kNbWords;
  float* bpart = lds + kUnion + kNbWords + LP;  // [wave][LP] partial bounds

  // ------------------------------------------------ neighbour boxes of every layer, once
  if (threadIdx.x < LP) nbcount[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    if (EXL || l < L) {
      const int4 bb = bbox[((int64_t)f * L + l) * ntiles + tile];
      for (int t = threadIdx.x; t < ntiles; t += kBwdThreads) {
        const int4 ob = bbox[((int64_t)f * L + l) * ntiles + t];
        const bool hit = t != tile && ob.x <= ob.y && max(ob.x, bb.x) <= min(ob.y, bb.y) &&
                         max(ob.z, bb.z) <= min(ob.w, bb.w);
        if (hit) {
          const int slot = atomicAdd(&nbcount[l], 1);
          if (slot < kMaxNb) {
            int* nb = nbl + (l * kMaxNb + slot) * 4;
            nb[0] = ob.x;
            nb[1] = ob.y;
            nb[2] = ob.z;
            nb[3] = ob.w;
          }
        }
      }
    }
  }

  // ------------------------------------------------------------------ phase 1: per pixel
  float ap_[kBwdPP][LP], gsa_[kBwdPP][LP], gx_[kBwdPP][LP], gy_[kBwdPP][LP], gc_[kBwdPP][3];
  bool live_[kBwdPP];
  float bound_[LP];  // sum over this thread's pixels of max_c |contribution| per layer
#pragma unroll
  for (int l = 0; l < LP; ++l) bound_[l] = 0.0f;
  const float* oc = occ + (int64_t)f * L * L;
#pragma unroll
  for (int q = 0; q < kBwdPP; ++q) {
    const PixelMap pm = pixel_of(tile, kBwdWaves * q + wave, lane, H, W, kBwdRows, ntx);
    live_[q] = pm.live;
    const float livef = pm.live ? 1.0f : 0.0f;
    const int64_t p = pm.p;
    float bas[K3];
    load_basis<K3, true>(bas, basis_t, HW, p, K3);
    const float g0 = grad_rgb[(int64_t)f * 3 * HW + p] * livef;
    const float g1 = grad_rgb[(int64_t)f * 3 * HW + HW + p] * livef;
    const float g2 = grad_rgb[(int64_t)f * 3 * HW + 2 * HW + p] * livef;
    gc_[q][0] = g0;
    gc_[q][1] = g1;
    gc_[q][2] = g2;
    const float gmax = fmaxf(fabsf(g0), fmaxf(fabsf(g1), fabsf(g2)));
    float a[LP], G[LP], dxr[LP], dxa[LP], dyr[LP], dya[LP];
    // (A) grid of every layer first: the basis registers die before the tap loads start
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int lc = EXL ? l : min(l, L - 1);
      tps_eval<K3, true>(bas, mapping + ((int64_t)f * L + lc) * K3 * 2, K3, gx_[q][l], gy_[q][l]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (B) taps with derivatives, kBwdGroup layers at a time
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int lc = EXL ? l : min(l, L - 1);
      const Taps t = make_taps(gx_[q][l], gy_[q][l], H, W);
      const float* base = layers + ((int64_t)f * L + lc) * 4 * HW;
      float sx[4], sy[4], sv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) sv[c] = tap_sample_d(base + c * HW, t, sx[c], sy[c]);
      const bool pad = !EXL && l >= L;
      a[l] = pad ? 0.0f : (sv[3] + 1.0f) * 0.5f;
      G[l] = g0 * (sv[0] + 1.0f) + g1 * (sv[1] + 1.0f) + g2 * (sv[2] + 1.0f);
      dxr[l] = fmaf(g2, sx[2], fmaf(g1, sx[1], g0 * sx[0]));
      dyr[l] = fmaf(g2, sy[2], fmaf(g1, sy[1], g0 * sy[0]));
      dxa[l] = sx[3];
      dya[l] = sy[3];
      if (grad_alpha != nullptr && !pad)
        G[l] = fmaf(2.0f * livef, grad_alpha[((int64_t)f * L + lc) * HW + p], G[l]);
      if ((l % kBwdGroup) == kBwdGroup - 1) __builtin_amdgcn_sched_barrier(0);
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
      ap_[q][j] = a[j] * pre;  // 0 for padding layers
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
    // a_m = (s_m3 + 1)/2 for m >= 1; a_0 is the constant 1.  Grid gradient of every layer.
    const int pix = q * kBwdThreads + threadIdx.x;
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const bool pad = !EXL && l >= L;
      gsa_[q][l] = (l >= 1 && !pad) ? 0.5f * ga[l] : 0.0f;
      const float gix = fmaf(gsa_[q][l], dxa[l], ap_[q][l] * dxr[l]);
      const float giy = fmaf(gsa_[q][l], dya[l], ap_[q][l] * dyr[l]);
      gg[pix * GGP + 2 * l] = gix * (0.5f * (float)W);
      gg[pix * GGP + 2 * l + 1] = giy * (0.5f * (float)H);
      // |tap contribution| <= max(|a'_l| max_c |g_c|, |gsa_l|): bilinear weights are <= 1
      bound_[l] += pm.live ? fmaxf(fabsf(ap_[q][l]) * gmax, fabsf(gsa_[q][l])) : 0.0f;
    }
#pragma unroll
    for (int c = NC; c < GGC; ++c) gg[pix * GGP + c] = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
  }
  // per-wave partial of the per-layer bounds (fixed butterfly: deterministic)
#pragma unroll
  for (int l = 0; l < LP; ++l) {
    const float sblk = wave_sum(bound_[l]);
    if (lane == 0) bpart[wave * LP + l] = sblk;
  }
  __syncthreads();  // gg rows, bpart and the neighbour lists are complete

  // ------------------------------------------- control-point gradient: basis^T x gg on the MFMA
  // D[k][col] = sum_pix basis[k][pix] * gg[pix][col]; v_mfma_f32_16x16x4_f32: A[row=lane&15]
  // [kk=lane>>4], B[kk=lane>>4][col=lane&15], D[row=(lane>>4)*4+reg][col=lane&15].  Each wave
  // contracts the pixels it produced; the 8 wave results are summed through LDS and stored as
  // this tile's partial (no atomics).
  if (gmap_partial != nullptr) {
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int arow = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int q = 0; q < kBwdPP; ++q) {
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        const int pl = 4 * s4 + kk;  // column of the contracted pixel inside this wave's row
        const PixelMap pm = pixel_of(tile, kBwdWaves * q + wave, pl, H, W, kBwdRows, ntx);
        const int pix = q * kBwdThreads + wave * kWave + pl;
        float av[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int k = mt * 16 + arow;
          const float v = basis_t[(int64_t)min(k, K3 - 1) * HW + pm.p];
          av[mt] = (k < K3) ? v : 0.0f;  // dead pixels carry gg == 0
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv = gg[pix * GGP + nt * 16 + arow];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv, acc[mt][nt], 0, 0, 0);
        }
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
    for (int o = threadIdx.x; o < 2 * NT * 256; o += kBwdThreads) {
      float sum = 0.0f;
#pragma unroll
      for (int w = 0; w < kBwdWaves; ++w) sum += lds[w * 2 * NT * 256 + o];  // fixed order
      const int mt = o / (NT * 256), nt = (o / 256) % NT, r = (o >> 6) & 3, ln = o & 63;
      const int k = mt * 16 + (ln >> 4) * 4 + r;
      const int col = nt * 16 + (ln & 15);
      const int l = col >> 1;
      if (k < K3 && l < L)
        gmap_partial[((int64_t)f * ntiles + tile) * gmap_partial_floats(L) + ((int64_t)l * K3 + k) * 2 +
                     (col & 1)] = sum;
    }
  }
  __syncthreads();  // gg / accumulators are dead; their bytes become the scatter images

  // ------------------------------------------------------------------ phase 2: per layer
  for (int e = threadIdx.x; e < kImgWords; e += kBwdThreads) img[e] = 0;
  __syncthreads();

#pragma unroll
  for (int l = 0; l < LP; ++l) {
    if (!EXL && l >= L) break;  // uniform
    const int buf = l & 1;
    int* im = img + buf * 4 * kMaxTex;
    const int* nb = nbl + l * kMaxNb * 4;
    const int4 bb = bbox[((int64_t)f * L + l) * ntiles + tile];
    const int bx0 = bb.x, by0 = bb.z;
    const int bw = bb.y - bb.x + 1, bh = bb.w - bb.z + 1;
    const bool empty = bw <= 0 || bh <= 0;
    const bool in_lds = !empty && bw * bh <= kMaxTex;
    float* gbase = grad_layers + ((int64_t)f * L + l) * 4 * HW;
    // fixed-point scale: B = sum over the tile's pixels of max_c |contribution| bounds the
    // magnitude of ANY texel sum; scale = 2^(29 - floor(log2 B)) keeps B * scale < 2^30
    float B = 0.0f;
#pragma unroll
    for (int w = 0; w < kBwdWaves; ++w) B += bpart[w * LP + l];
    const int eB = (int)((__float_as_uint(B) >> 23) & 0xffu) - 127;
    const int es = min(29 - eB, 126);
    const float scale = __uint_as_float((unsigned)(127 + es) << 23);
    const float inv_scale = __uint_as_float((unsigned)(127 - es) << 23);
    const bool any = B > 0.0f;

    // scatter this thread's taps
#pragma unroll
    for (int q = 0; q < kBwdPP; ++q) {
      if (!live_[q] || !any) continue;
      // re-derive the taps from the stored grid point; `opaque` stops the compiler from keeping
      // all 14 tap values of every layer alive since phase 1 instead (register spills)
      const Taps t = make_taps(opaque(gx_[q][l]), opaque(gy_[q][l]), H, W);
      float gv[4];
      gv[0] = ap_[q][l] * gc_[q][0];
      gv[1] = ap_[q][l] * gc_[q][1];
      gv[2] = ap_[q][l] * gc_[q][2];
      gv[3] = gsa_[q][l];
      if (in_lds) {
        const int lx = t.x0 - bx0, ly = t.y0 - by0;  // taps with non-zero weight lie in the box
        const int o00 = ly * bw + lx;
        const bool t00 = t.w00 != 0.0f, t01 = t.w01 != 0.0f, t10 = t.w10 != 0.0f, t11 = t.w11 != 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int* pc = im + c * kMaxTex;
          const float gs = gv[c] * scale;
          if (t00) atomicAdd(pc + o00, __float2int_rn(gs * t.w00));
          if (t01) atomicAdd(pc + o00 + 1, __float2int_rn(gs * t.w01));
          if (t10) atomicAdd(pc + o00 + bw, __float2int_rn(gs * t.w10));
          if (t11) atomicAdd(pc + o00 + bw + 1, __float2int_rn(gs * t.w11));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float* pl = gbase + c * HW;
          if (t.w00 != 0.0f) atomicAdd(pl + (t.o00 >> 2), gv[c] * t.w00);
          if (t.w01 != 0.0f) atomicAdd(pl + (t.o01 >> 2), gv[c] * t.w01);
          if (t.w10 != 0.0f) atomicAdd(pl + (t.o10 >> 2), gv[c] * t.w10);
          if (t.w11 != 0.0f) atomicAdd(pl + (t.o11 >> 2), gv[c] * t.w11);
        }
      }
    }
    lds_barrier();
    // flush: plain stores where the texel is ours alone, atomics on shared rims; re-zero the image
    if (in_lds) {
      const int nnb = nbcount[l];
      const bool all_shared = nnb > kMaxNb;
      const int nn = min(nnb, kMaxNb);
      const float rcp_bw = 1.0f / (float)bw;
      constexpr int kIter = (kMaxTex + kBwdThreads - 1) / kBwdThreads;  // texels per thread
      int ex_[kIter], ey_[kIter];
      unsigned sharedmask = all_shared ? 0xffffffffu : 0u;
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int e = threadIdx.x + it * kBwdThreads;
        // e, bw < 2^11: (e + 0.5) / bw is never within fp32 rounding of an integer
        const int r = (int)(((float)e + 0.5f) * rcp_bw);
        ex_[it] = bx0 + (e - r * bw);
        ey_[it] = by0 + r;
      }
      for (int n = 0; n < nn; ++n) {  // neighbour box: wave-uniform, read once per thread
        const int nx0 = __builtin_amdgcn_readfirstlane(nb[n * 4 + 0]);
        const int nx1 = __builtin_amdgcn_readfirstlane(nb[n * 4 + 1]);
        const int ny0 = __builtin_amdgcn_readfirstlane(nb[n * 4 + 2]);
        const int ny1 = __builtin_amdgcn_readfirstlane(nb[n * 4 + 3]);
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
          const bool hit = ex_[it] >= nx0 && ex_[it] <= nx1 && ey_[it] >= ny0 && ey_[it] <= ny1;
          sharedmask |= hit ? (1u << it) : 0u;
        }
      }
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int e = threadIdx.x + it * kBwdThreads;
        if (e < bw * bh) {
          float* dst = gbase + (int64_t)ey_[it] * W + ex_[it];
          const bool shared = (sharedmask >> it) & 1u;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int iv = im[c * kMaxTex + e];  // row-major inside the box
            im[c * kMaxTex + e] = 0;
            const float v = (float)iv * inv_scale;
            if (shared) {
              if (iv != 0) atomicAdd(dst + c * HW, v);
            } else {
              dst[c * HW] = v;
            }
          }
        }
      }
    }
    // no second barrier: image `buf` is next written by layer l+2, i.e. after the barrier of
    // layer l+1, which every thread reaches only after finishing this flush
  }
}

// second stage of the control-point gradient: grad_mapping[f,l,k,c] += sum_tile partial
static __global__ __launch_bounds__(kBlock) void warp_composite_gmap_reduce_kernel(
    const float* __restrict__ gmap_partial, float* __restrict__ grad_mapping, int F, int L,
    int ntiles) {
  const int per = L * kGmapK3 * 2;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= (int64_t)F * per) return;
  const int64_t f = e / per;
  const int o = (int)(e % per);
  const float* src = gmap_partial + f * ntiles * per + o;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int t = 0;
  for (; t + 3 < ntiles; t += 4) {
    s0 += src[(int64_t)t * per];
    s1 += src[(int64_t)(t + 1) * per];
    s2 += src[(int64_t)(t + 2) * per];
    s3 += src[(int64_t)(t + 3) * per];
  }
  for (; t < ntiles; ++t) s0 += src[(int64_t)t * per];
  grad_mapping[e] += (s0 + s1) + (s2 + s3);
}

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
  int fpb = 4;
  while (fpb > 1 && (int64_t)g.ntiles * ((F + fpb - 1) / fpb) < 2048) fpb >>= 1;
  dim3 grid(g.ntiles, (F + fpb - 1) / fpb);
  if (L == LP)
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, true, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx);
  else
    hipLaunchKernelGGL((warp_composite_fwd_kernel<LP, K3P, false, EXK>), grid, dim3(kBlock), 0, st,
                       layers, basis_t, mapping, occ, rgb, alpha, F, L, H, W, K3, fpb, g.ntx);
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

// tiled backward; `workspace` = [F*L*ntiles int4 boxes | F*ntiles*L*38 float partials]
static inline int64_t bwd2_workspace_bytes(int64_t F, int L, int H, int W) {
  const TileGeom g = tile_geom(H, W, kBwdRows);
  return F * L * g.ntiles * kBboxBytes + F * g.ntiles * gmap_partial_floats(L) * 4;
}

template <int LP>
static void launch_bwd2(const float* layers, const float* basis_t, const float* mapping,
                        const float* occ, const float* grad_rgb, const float* grad_alpha,
                        void* workspace, float* grad_layers, float* grad_mapping, float* grad_occ,
                        int F, int L, int H, int W, hipStream_t st) {
  const TileGeom g = tile_geom(H, W, kBwdRows);
  dim3 grid(g.ntiles, F);
  int4* bb = reinterpret_cast<int4*>(workspace);
  float* part = grad_mapping == nullptr
                    ? nullptr
                    : reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                               (int64_t)F * L * g.ntiles * kBboxBytes);
  auto go = [&](auto exl, auto gocc) {
    constexpr bool EXL = decltype(exl)::value, GOCC = decltype(gocc)::value;
    hipLaunchKernelGGL((warp_composite_bbox_kernel<LP, 19, EXL, true>), grid, dim3(kBlock), 0, st,
                       basis_t, mapping, bb, F, L, H, W, 19, kBwdRows, g.ntx, g.ntiles);
    hipLaunchKernelGGL((warp_composite_bwd2_kernel<LP, EXL, GOCC>), grid, dim3(kBwdThreads), 0, st,
                       layers, basis_t, mapping, occ, grad_rgb, grad_alpha, bb, part, grad_layers,
                       grad_occ, F, L, H, W, g.ntx, g.ntiles);
  };
  using T = std::true_type;
  using N = std::false_type;
  if (L == LP) {
    if (grad_occ != nullptr) go(T{}, T{}); else go(T{}, N{});
  } else {
    if (grad_occ != nullptr) go(N{}, T{}); else go(N{}, N{});
  }
  if (part != nullptr) {
    const int64_t n = (int64_t)F * gmap_partial_floats(L);
    hipLaunchKernelGGL(warp_composite_gmap_reduce_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, part, grad_mapping, F, L, g.ntiles);
  }
}

}  // namespace waldo
