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
  return t;
}

__device__ __forceinline__ float up_sample(const float* __restrict__ plane, const UpTaps& t) {
  const float top = t.lx0 * ldb(plane, t.o00) + t.lx1 * ldb(plane, t.o01);
  const float bot = t.lx0 * ldb(plane, t.o10) + t.lx1 * ldb(plane, t.o11);
  return t.ly0 * top + t.ly1 * bot;
}

constexpr int kMaxCls = 32;


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
