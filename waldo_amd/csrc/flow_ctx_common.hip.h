// Helpers shared by the fused full-resolution passes (flow_ctx.hip) and their backward
// (flow_ctx_bwd.hip): the taps of F.interpolate's bilinear upsampling and the identity grid of the
// HD raster exactly as tools/utils.py:get_grid builds it.
#pragma once
#include "waldo_common.hip.h"

namespace waldo {

// source taps of F.interpolate(mode="bilinear", align_corners=False, scale_factor=s):
// src = max((dst + 0.5) / s - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, size - 1)
struct UpTap {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ UpTap up_tap(int dst, float rscale, int size) {
  const float src = fmaxf(((float)dst + 0.5f) * rscale - 0.5f, 0.0f);
  UpTap t;
  t.i0 = min((int)src, size - 1);
  t.i1 = min(t.i0 + 1, size - 1);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

// the four taps of one HD pixel inside ANY low-resolution plane: byte offsets + weights, computed
// once per thread and shared by all the planes it upsamples (uniform plane base + 32-bit offset)
struct UpTaps {
  uint32_t o00, o01, o10, o11;
  float lx0, lx1, ly0, ly1;
  bool unit;  // scale factor 1 (uniform): the plane itself, as the reference's scale() returns it (lvd.py:175-179)
};

__device__ __forceinline__ UpTaps up_taps(int y, int x, float rscale, int H, int W) {
  const UpTap ty = up_tap(y, rscale, H), tx = up_tap(x, rscale, W);
  UpTaps t;
  t.o00 = (uint32_t)(__mul24(ty.i0, W) + tx.i0) * 4u;
  t.o01 = (uint32_t)(__mul24(ty.i0, W) + tx.i1) * 4u;
  t.o10 = (uint32_t)(__mul24(ty.i1, W) + tx.i0) * 4u;
  t.o11 = (uint32_t)(__mul24(ty.i1, W) + tx.i1) * 4u;
  t.lx0 = tx.l0;
  t.lx1 = tx.l1;
  t.ly0 = ty.l0;
  t.ly1 = ty.l1;
  t.unit = rscale == 1.0f;
  return t;
}

// the weighted sum of F.interpolate's four taps, rows first (six operations; the ONE form every kernel of
// the path evaluates, whether the taps come from memory or from LDS)
__device__ __forceinline__ float up_blend(const UpTaps& t, float v00, float v01, float v10, float v11) {
  const float top = fmaf(t.lx1, v01, t.lx0 * v00);
  const float bot = fmaf(t.lx1, v11, t.lx0 * v10);
  return fmaf(t.ly1, bot, t.ly0 * top);
}

__device__ __forceinline__ float up_sample(const float* __restrict__ plane, const UpTaps& t) {
  // (at scale 1 the taps are i0 = the pixel with weight 1 and a neighbour with weight 0: one load, not four --
  // the LVD recipe trains without a full-resolution raster and every upsampling of its step is this one)
  if (t.unit) return ldb(plane, t.o00);
  return up_blend(t, ldb(plane, t.o00), ldb(plane, t.o01), ldb(plane, t.o10), ldb(plane, t.o11));
}

constexpr int kMaxCls = 32;
// The kernels that hold a pixel's class probabilities in registers are compiled for 20 classes (Cityscapes' 20, KITTI's
// 19) and for kMaxCls: with the count only known at run time every pixel paid 32 exponentials, 32 divisions and -- in
// the backward -- a 32-value transpose-reduce per object, for 20 classes.
constexpr int kFewCls = 20;

// ---------------------------------------------------------------------------------------
// The order (occlusion matrix) of one frame in LDS, padded to LP x LP: by rows ([i][j], row stride kRow)
// and transposed ([j][i]).  A padding row / column repeats the last real one -- its layer's alpha is 0, so
// its factor 1 - a occ is exactly 1 and nothing needs a guard.  The kernels used to read occ[i * L + j]
// through the scalar cache: with the run-time stride that is ~7 SALU instructions per element and, what
// costs more, an s_waitcnt per handful of loads (hoisting L x L of them runs out of SGPRs) -- at L = 17 the
// backward kernels spent most of their 40 us per tile waiting for ~870 scalar loads in turn.  From LDS the
// same values are broadcast ds_read_b128 with compile-time addresses, many in flight.
// ---------------------------------------------------------------------------------------
template <int LP>
struct OccLds {
  static constexpr int kRow = (LP + 3) & ~3;
  static constexpr int kFloats = 2 * LP * kRow;
};

// block-wide; no barrier inside.  Returns whether THIS thread saw a non-finite entry (callers that skip factors
// of the product vote on it: 1 - 0 * inf is not 1).
template <int LP>
__device__ __forceinline__ bool occ_stage(float* occm, const float* __restrict__ oc, int L) {
  constexpr int R = OccLds<LP>::kRow;
  bool bad = false;
  for (int e = threadIdx.x; e < LP * R; e += kBlock) {
    const int r = e / R, c = e - r * R;
    const int rc = min(r, L - 1), cc = min(c, L - 1);
    const float v = oc[rc * L + cc];
    bad |= !(fabsf(v) <= 3.0e38f);
    occm[e] = v;                         // [i = r][j = c]
    occm[LP * R + e] = oc[cc * L + rc];  // [j = r][i = c]
  }
  return bad;
}

typedef float f32x4_o __attribute__((ext_vector_type(4)));

// The objects' class distributions of one batch entry in LDS: one row of kMaxCls floats per layer 1 .. LP - 1
// (row l - 1 = object min(l, L - 1) - 1; zeros from class Nl on, and everywhere when there is no object), so
// that a layer's row sits at a compile-time address and is read four classes at a time -- it was a branch, a
// ds_read_b32 and a wait per (layer, class).  Block-wide; no barrier inside.
// Returns whether THIS thread saw a non-finite value (see occ_stage).
template <int LP>
__device__ __forceinline__ bool dist_stage(float* sdist, const float* __restrict__ dist_b, int L, int Nl) {
  const int No = L - 1;
  bool bad = false;
  for (int e = threadIdx.x; e < (LP - 1) * kMaxCls; e += kBlock) {
    const int r = e / kMaxCls, c = e - r * kMaxCls;
    const float v = (c < Nl && No > 0) ? dist_b[min(r, No - 1) * Nl + c] : 0.0f;
    bad |= !(fabsf(v) <= 3.0e38f);
    sdist[e] = v;
  }
  return bad;
}

// sum_c |dist[row][c] - pr[c]| over the Nl classes in ascending order (pr[c] == 0 from class Nl on: the
// padding of a group of four adds exact zeros)
template <int NCP>
__device__ __forceinline__ float dist_l1(const float* sdist_row, const float (&pr)[NCP], int Nl) {
  static_assert(NCP % 4 == 0 && NCP <= kMaxCls, "classes in groups of four");
  float d = 0.0f;
#pragma unroll
  for (int c = 0; c < NCP; c += 4)
    if (c < Nl) {  // uniform
      const f32x4_o q = *reinterpret_cast<const f32x4_o*>(sdist_row + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) d += fabsf(q[k] - pr[c + k]);
    }
  return d;
}

// elements [c0, c0 + 4) of row r of the row-major (T = false) or transposed (T = true) copy
template <int LP, bool T>
__device__ __forceinline__ f32x4_o occ_quad(const float* occm, int r, int c0) {
  constexpr int R = OccLds<LP>::kRow;
  return *reinterpret_cast<const f32x4_o*>(occm + (T ? LP * R : 0) + r * R + c0);
}


// texel centre (x, y) of the identity grid as get_grid() builds it: torch.linspace(start, end, n)
// with start / end rounded from double, step = (end - start) / (n - 1) in float, and the upper
// half counted down from the end
__device__ __forceinline__ void identity_grid(int x, int y, int Wd, int Hd, float& gx0, float& gy0) {
  const float sx = (float)(-1.0 + 1.0 / (double)Wd), ex = (float)(1.0 - 1.0 / (double)Wd);
  const float sy = (float)(-1.0 + 1.0 / (double)Hd), ey = (float)(1.0 - 1.0 / (double)Hd);
  const float stepx = (Wd > 1) ? (ex - sx) / (float)(Wd - 1) : 0.0f;
  const float stepy = (Hd > 1) ? (ey - sy) / (float)(Hd - 1) : 0.0f;
  gx0 = (x < Wd / 2) ? sx + stepx * (float)x : ex - stepx * (float)(Wd - 1 - x);
  gy0 = (y < Hd / 2) ? sy + stepy * (float)y : ey - stepy * (float)(Hd - 1 - y);
}

// ---------------------------------------------------------------------------------------
// Pixels of a workgroup on the full-resolution raster.  A workgroup covers a TILE of 4 rows x 64
// columns (one row segment per wavefront: every plane access stays a coalesced 256-byte run), and the
// tiles of one outer unit (a frame of the batch) go to ONE XCD in row-major order (xcd_decode_banded;
// placement is for speed only).  With the linear 256-pixel strips these kernels started with,
// consecutive strips went to different XCDs: the tap row a wavefront shares with the one below, the
// low-resolution texels an x S upsampling re-reads S x S times and the 128-byte lines neighbouring
// strips split were fetched once per XCD -- rocprofv3 showed frame_warp_fuse fetching 2.4 x (and
// flow_ctx_warp 4.8 x) the bytes it has to read at the KITTI recipe (profiles/r03_pipeline_C4_*).
// ---------------------------------------------------------------------------------------
constexpr int kHdRows = 4, kHdCols = 64;
static_assert(kHdRows * kHdCols == kBlock && kHdCols == kWave, "one 64-pixel row segment per wavefront");

struct HdGeom {
  int tiles, nbands;
};
inline HdGeom hd_geom(int64_t units, int Hd, int Wd) {
  HdGeom g;
  g.tiles = ((Hd + kHdRows - 1) / kHdRows) * ((Wd + kHdCols - 1) / kHdCols);
  g.nbands = xcd_bands((int)(units % 8 == 0 ? 8 : units % 8));  // 8 / gcd(units, 8)
  return g;
}
inline int64_t hd_grid(int64_t units, const HdGeom& g) { return xcd_grid_banded(units, g.nbands, g.tiles, 1); }

// unit and pixel (x, y) of this thread; false: the whole workgroup has nothing to do (uniform).  The
// pixel may lie outside the raster (x >= Wd or y >= Hd at the right / bottom edge): callers test it.
__device__ __forceinline__ bool hd_pixel(int units, int Hd, int Wd, int tiles, int nbands, int& unit, int& x, int& y) {
  int tile, rest_;
  if (!xcd_decode_banded(blockIdx.x, units, nbands, tiles, 1, unit, tile, rest_)) return false;
  const int ntx = (Wd + kHdCols - 1) / kHdCols;
  const int ty = tile / ntx;
  x = (tile - ty * ntx) * kHdCols + (int)(threadIdx.x & (kWave - 1));
  y = ty * kHdRows + (int)(threadIdx.x >> 6);
  return true;
}

// Tall tiles: a workgroup covers R x kHdRows rows of 64 columns, every thread R pixels kHdRows rows apart (flow_ctx_warp:
// what a tile stages and computes once -- the order, the low-resolution patch, the per-pixel column constants -- is
// shared by R times the pixels).  y is the thread's FIRST row; its others are y + kHdRows * r.
inline HdGeom hd_geom_rows(int64_t units, int Hd, int Wd, int R) {
  HdGeom g;
  g.tiles = ((Hd + kHdRows * R - 1) / (kHdRows * R)) * ((Wd + kHdCols - 1) / kHdCols);
  g.nbands = xcd_bands((int)(units % 8 == 0 ? 8 : units % 8));
  return g;
}
template <int R>
__device__ __forceinline__ bool hd_pixel_rows(int units, int Hd, int Wd, int tiles, int nbands, int& unit, int& x, int& y) {
  int tile, rest_;
  if (!xcd_decode_banded(blockIdx.x, units, nbands, tiles, 1, unit, tile, rest_)) return false;
  const int ntx = (Wd + kHdCols - 1) / kHdCols;
  const int ty = tile / ntx;
  x = (tile - ty * ntx) * kHdCols + (int)(threadIdx.x & (kWave - 1));
  y = ty * (kHdRows * R) + (int)(threadIdx.x >> 6);
  return true;
}

// (units in groups of `inner` that share their sources, the members innermost in an XCD's tile walk: see
// HdTile::pixel_grouped)
template <int R>
__device__ __forceinline__ bool hd_pixel_rows_grouped(int groups, int inner, int Hd, int Wd, int tiles, int nbands,
                                                      int& unit, int& x, int& y) {
  int tile, member, grp;
  if (!xcd_decode_banded(blockIdx.x, groups, nbands, tiles, inner, grp, tile, member)) return false;
  unit = grp * inner + member;
  const int ntx = (Wd + kHdCols - 1) / kHdCols;
  const int ty = tile / ntx;
  x = (tile - ty * ntx) * kHdCols + (int)(threadIdx.x & (kWave - 1));
  y = ty * (kHdRows * R) + (int)(threadIdx.x >> 6);
  return true;
}

// The same with a workgroup of (256 / TC) x TC pixels, a wavefront covering 64 / TC rows of TC columns
// (frame_warp_fuse: WALDO_FWF_TILE_COLS).
template <int TC>
struct HdTile {
  static constexpr int kCols = TC, kRows = kBlock / TC;
  static_assert(TC == 16 || TC == 32 || TC == 64, "tile width");
  static HdGeom geom(int64_t units, int Hd, int Wd) {
    HdGeom g;
    g.tiles = ((Hd + kRows - 1) / kRows) * ((Wd + kCols - 1) / kCols);
    g.nbands = xcd_bands((int)(units % 8 == 0 ? 8 : units % 8));
    return g;
  }
  __device__ static __forceinline__ bool pixel(int units, int Hd, int Wd, int tiles, int nbands, int& unit, int& x,
                                               int& y) {
    int tile, rest_;
    if (!xcd_decode_banded(blockIdx.x, units, nbands, tiles, 1, unit, tile, rest_)) return false;
    const int ntx = (Wd + kCols - 1) / kCols;
    const int ty = tile / ntx;
    x = (tile - ty * ntx) * kCols + (int)(threadIdx.x & (kCols - 1));
    y = ty * kRows + (int)(threadIdx.x / kCols);
    return true;
  }
  // The same with the units in GROUPS of `inner` that share their sources (frame_warp_fuse: the Tp predicted frames of
  // a clip gather from the same Tc context frames): a band of a group's tiles is pinned to an XCD and walked tile by
  // tile with the group's members INNERMOST, so that the `inner` workgroups that read the same footprints run side
  // by side behind one L2.  unit = group * inner + member.
  __device__ static __forceinline__ bool pixel_grouped(int groups, int inner, int Hd, int Wd, int tiles, int nbands,
                                                       int& unit, int& x, int& y) {
    int tile, member, grp;
    if (!xcd_decode_banded(blockIdx.x, groups, nbands, tiles, inner, grp, tile, member)) return false;
    unit = grp * inner + member;
    const int ntx = (Wd + kCols - 1) / kCols;
    const int ty = tile / ntx;
    x = (tile - ty * ntx) * kCols + (int)(threadIdx.x & (kCols - 1));
    y = ty * kRows + (int)(threadIdx.x / kCols);
    return true;
  }
};

// ---------------------------------------------------------------------------------------
// Low-resolution planes of a tile, staged in LDS.  A 4 x 64 tile of the x S raster reads an
// (4 / S + 2) x (64 / S + 2) patch of every low-resolution plane it upsamples; taken tap by tap from
// memory that was 4 loads per plane and pixel (8 L + 4 (L - 1) vector-memory instructions per pixel in
// flow_ctx_warp: the kernels count their memory INSTRUCTIONS as much as their bytes).
// Here every thread loads at most one texel per plane, once, and the taps come out of LDS with the same
// arithmetic as up_sample().  lr_patch() is wave-uniform; a patch of more than 256 texels (S == 1) or
// more than kLrCap floats in all does not qualify (callers keep the direct path).
// ---------------------------------------------------------------------------------------
constexpr int kLrCap = 6144;  // floats of LDS for the staged planes (24 KB: four workgroups per CU and more)

struct LrPatch {
  int r_lo, c_lo, nrows, ncols;
};

__device__ __forceinline__ LrPatch lr_patch(int ty0, int tx0, int Hd, int Wd, float rscale, int H, int W,
                                            int tile_rows = kHdRows) {
  LrPatch q;
  q.r_lo = up_tap(ty0, rscale, H).i0;
  q.c_lo = up_tap(tx0, rscale, W).i0;
  q.nrows = up_tap(min(ty0 + tile_rows - 1, Hd - 1), rscale, H).i1 - q.r_lo + 1;
  q.ncols = up_tap(min(tx0 + kHdCols - 1, Wd - 1), rscale, W).i1 - q.c_lo + 1;
  return q;
}

// planes [0, nplanes) of `src` (plane stride HW floats) into img[plane][nrows][ncols]; needs area <= kBlock.
// No barrier inside.
__device__ __forceinline__ void lr_stage(float* img, const float* __restrict__ src, int nplanes, int64_t HW, int W,
                                         const LrPatch& q) {
  const int area = q.nrows * q.ncols;
  const int t = min((int)threadIdx.x, area - 1);
  // t < 256, ncols < 256, the + 0.5: the approximate reciprocal gives the exact quotient
  const int r = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)q.ncols));
  const uint32_t off = (uint32_t)(__mul24(q.r_lo + r, W) + q.c_lo + (t - __mul24(r, q.ncols))) * 4u;
  if ((int)threadIdx.x < area)
    for (int pl = 0; pl < nplanes; ++pl) img[pl * area + t] = ldb(src + (int64_t)pl * HW, off);
}

// word offsets of a pixel's four taps inside one staged plane + the weights (those of up_taps())
struct LrTaps {
  int o00, o01, o10, o11;
  float lx0, lx1, ly0, ly1;
};

__device__ __forceinline__ LrTaps lr_taps(int y, int x, float rscale, int H, int W, const LrPatch& q) {
  const UpTap ty = up_tap(y, rscale, H), tx = up_tap(x, rscale, W);
  LrTaps t;
  const int r0 = __mul24(ty.i0 - q.r_lo, q.ncols), r1 = __mul24(ty.i1 - q.r_lo, q.ncols);
  t.o00 = r0 + tx.i0 - q.c_lo;
  t.o01 = r0 + tx.i1 - q.c_lo;
  t.o10 = r1 + tx.i0 - q.c_lo;
  t.o11 = r1 + tx.i1 - q.c_lo;
  t.lx0 = tx.l0;
  t.lx1 = tx.l1;
  t.ly0 = ty.l0;
  t.ly1 = ty.l1;
  return t;
}

__device__ __forceinline__ float lr_sample(const float* plane_img, const LrTaps& t) {
  const float top = t.lx0 * plane_img[t.o00] + t.lx1 * plane_img[t.o01];
  const float bot = t.lx0 * plane_img[t.o10] + t.lx1 * plane_img[t.o11];
  return t.ly0 * top + t.ly1 * bot;
}

// ---------------------------------------------------------------------------------------
// Bilinear sample of F.grid_sample (zeros padding, align_corners=False) with the two taps of a row as
// ONE 8-byte load at the pair origin xb = clamp(x0, 0, W - 2): half the gather instructions of four
// single taps (unaligned dwordx2 loads are fine on gfx950).  Same value, bit for bit, as
// tap_sample(plane, make_taps(gx, gy)): within one texel of the left / right border the pair sits one
// column off the footprint and its elements are re-assigned to the corners (corners outside the image
// carry validity 0 as before).  `interior` (wave-uniform: every lane has all four corners inside) skips
// validity and re-assignment.
// ---------------------------------------------------------------------------------------
struct PairTaps {
  uint32_t ob0, ob1;
  float fx, fy;
  int edge;  // border lanes only: validity of the corners (bits 0-3: 00, 01, 10, 11) and x0 - xb + 1 (bits 4-5)
};

__device__ __forceinline__ PairTaps pair_taps(float gx, float gy, int Hi, int Wi, bool& interior) {
  const TapCore c = tap_core(gx, gy, Hi, Wi);
  PairTaps t;
  t.fx = c.fx;
  t.fy = c.fy;
  interior = __ballot(!tap_interior(c, Hi, Wi)) == 0ull;
  if (interior) {
    t.ob0 = (uint32_t)(__mul24(c.y0, Wi) + c.x0) * 4u;
    t.ob1 = t.ob0 + (uint32_t)Wi * 4u;
    t.edge = 15 | (1 << 4);
  } else {
    const Taps f = finish_taps(c, Hi, Wi);
    const int xb = min(max(c.x0, 0), Wi - 2);
    const int cy0 = min(max(c.y0, 0), Hi - 1), cy1 = min(max(c.y0 + 1, 0), Hi - 1);
    t.ob0 = (uint32_t)(__mul24(cy0, Wi) + xb) * 4u;
    t.ob1 = (uint32_t)(__mul24(cy1, Wi) + xb) * 4u;
    // (the validities are 0 / 1: their products are conjunctions)
    t.edge = (f.vx0 * f.vy0 != 0.0f ? 1 : 0) | (f.vx1 * f.vy0 != 0.0f ? 2 : 0) | (f.vx0 * f.vy1 != 0.0f ? 4 : 0) |
             (f.vx1 * f.vy1 != 0.0f ? 8 : 0) | ((c.x0 - xb + 1) << 4);
  }
  return t;
}

typedef float f32x2_p __attribute__((ext_vector_type(2)));

// the two loads and the arithmetic apart, so that a kernel can put the loads of MANY samples in flight
// before it consumes the first (a wavefront that waits for each sample in turn is bound by the memory
// latency times the number of samples)
__device__ __forceinline__ void pair_load(const float* __restrict__ plane, const PairTaps& t, f32x2_p& a, f32x2_p& b) {
  a = *reinterpret_cast<const f32x2_p*>(reinterpret_cast<const char*>(plane) + t.ob0);
  b = *reinterpret_cast<const f32x2_p*>(reinterpret_cast<const char*>(plane) + t.ob1);
}

__device__ __forceinline__ float pair_value(const f32x2_p a, const f32x2_p b, const PairTaps& t, bool interior) {
  float p00 = a[0], p01 = a[1], p10 = b[0], p11 = b[1];
  if (!interior) {  // wave-uniform
    const int shift = (t.edge >> 4) - 1;
    p00 = (shift > 0 ? a[1] : a[0]) * ((t.edge & 1) ? 1.0f : 0.0f);
    p01 = (shift < 0 ? a[0] : a[1]) * ((t.edge & 2) ? 1.0f : 0.0f);
    p10 = (shift > 0 ? b[1] : b[0]) * ((t.edge & 4) ? 1.0f : 0.0f);
    p11 = (shift < 0 ? b[0] : b[1]) * ((t.edge & 8) ? 1.0f : 0.0f);
  }
  const float top = fmaf(t.fx, p01 - p00, p00);
  const float bot = fmaf(t.fx, p11 - p10, p10);
  return fmaf(t.fy, bot - top, top);
}

__device__ __forceinline__ float pair_sample(const float* __restrict__ plane, const PairTaps& t, bool interior) {
  f32x2_p a, b;
  pair_load(plane, t, a, b);
  return pair_value(a, b, t, interior);
}

constexpr int kFwMaxCtx = 8;  // contexts (incl. self) of the fused frame warp

inline int flow_ctx_pad_l(int L) {
  if (L <= 4) return 4;
  if (L <= 8) return 8;
  if (L <= 12) return 12;
  if (L <= 17) return 17;
  if (L <= 24) return 24;
  return 32;
}

}  // namespace waldo
