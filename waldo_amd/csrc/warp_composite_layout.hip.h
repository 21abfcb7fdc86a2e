// Tile geometry and workspace layout of the tiled backward (host side; shared by the C-ABI dispatcher, which
// answers waldo_warp_composite_bwd_workspace_bytes, and by the launchers in
// warp_composite_kernels.hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace waldo {

constexpr int kGmapK3 = 19;
constexpr int kTileW = 64;                     // pixel tiles of the plain kernels: rows x 64
#ifndef WALDO_CELL_ROWS
#define WALDO_CELL_ROWS 8
#endif
constexpr int kCellRows = WALDO_CELL_ROWS, kCellCols = 16;   // cell of the footprint table (measured: K2 1.11 /
                                               // 1.01 / 1.04 ms for 4 / 8 / 16 rows)
constexpr int kBwd2MaxLayers = 17;             // largest L the two-kernel backward is compiled for
constexpr int kLdsTile = 16;                   // LDS-staged kernels: 16 x 16 pixels per workgroup
constexpr int kStageCap = 512;                 // texels per channel plane of a staged image (forward)

inline int64_t round256(int64_t b) { return ((b + 255) / 256) * 256; }
__host__ __device__ constexpr int64_t gmap_partial_floats(int L) { return (int64_t)L * kGmapK3 * 2; }

// ---- two-kernel backward: [cell boxes | cell bounds | records (grid x, grid y, a', g_alpha: 16 B) |
//                           partials]
// The pixel kernel (K1) runs on 16 x 16-pixel tiles; a tile covers kLdsTile / kCellRows cells of
// one table column.
struct Bwd2Layout {
  int64_t box_bytes, bound_bytes, rec_bytes, part_bytes, total;
  int ntx16, ntiles16;    // 16 x 16 tiles
  int ncx, ncells;
};

// Contribution bounds of a cell as K1 publishes them: two biased exponents, e_rgb | e_alpha << 8.
// e == 0: nothing but zeros / denormals; e == 255: an infinity or NaN in the cell (K2 poisons the
// tiles it reaches); else the cell's largest 16-pixel row sum is < 2^(e - 127) (e already holds
// the + 5 of "16 contributions, each < 2^(exponent - 126)").
__host__ __device__ inline unsigned bound_exponent(int e) {
  return e == 0 ? 0u : (e >= 255 ? 255u : (unsigned)(e + 5 < 254 ? e + 5 : 254));
}
__host__ __device__ inline unsigned pack_bound_exponents(int e_rgb, int e_alpha) {
  return bound_exponent(e_rgb) | (bound_exponent(e_alpha) << 8);
}

inline bool staged_eligible(int H, int W) { return (W % 4) == 0 && H >= 2 && W >= 2; }

inline Bwd2Layout bwd2_layout(int64_t F, int L, int H, int W) {
  Bwd2Layout o;
  o.ntx16 = (W + kLdsTile - 1) / kLdsTile;
  o.ntiles16 = o.ntx16 * ((H + kLdsTile - 1) / kLdsTile);
  o.ncx = (W + kCellCols - 1) / kCellCols;
  o.ncells = o.ncx * ((H + kCellRows - 1) / kCellRows);
  o.box_bytes = round256(F * L * o.ncells * 16);
  o.bound_bytes = round256(F * L * o.ncells * 4);
  o.rec_bytes = 2 * round256(F * L * (int64_t)H * W * 8);
  o.part_bytes = round256(F * o.ntiles16 * gmap_partial_floats(L) * 4);
  o.total = o.box_bytes + o.bound_bytes + o.rec_bytes + o.part_bytes;
  return o;
}

// 0: the shape is served by the generic backward (per-tap atomics), which needs no workspace
inline int64_t bwd_workspace_bytes(int64_t F, int L, int H, int W, int K3) {
  if (K3 != kGmapK3 || L > kBwd2MaxLayers || !staged_eligible(H, W) ||
      (int64_t)H * W * K3 * 4 >= 4294967296ll)
    return 0;
  return bwd2_layout(F, L, H, W).total;
}

}  // namespace waldo
