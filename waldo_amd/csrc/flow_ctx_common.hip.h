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
