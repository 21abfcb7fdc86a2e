// K2 of the two-kernel backward: gradient w.r.t. one layer plane, gathered per SOURCE tile.
//
// Restates the input-gradient of F.grid_sample (bilinear, zeros, align_corners=False) -- the
// four-corner scatter of grad * weight that the reference gets from autograd through
// models/nets/lvd.py:548,559 -- for the records the pixel kernel (K1) left behind:
//   records (grid x, grid y in PIXEL units, a'_l, g_alpha) per (frame, layer, pixel); the contribution of pixel p
//   to channel c < 3 of layer l is a'_l * grad_rgb[c][p], to the alpha channel g_alpha.
//
// One workgroup OWNS one 32x64-texel tile S of one layer's gradient plane: it is the only writer
// of those texels, so the whole plane is written with plain, row-coalesced stores -- no global
// atomics, no zero-fill of the output, and a bitwise reproducible result.  The pixels that can
// touch S are found through K1's cell table: per 8x16-pixel cell the bounding box of the texels
// its bilinear footprints reach (the warp's skew over 16 columns is small, so the boxes are
// tight) and an upper bound of the cell's contribution magnitudes.  The workgroup visits the
// cells whose box intersects S, re-derives the taps of their pixels from the records and sums
// the taps that fall into S in a 32-bit FIXED-POINT LDS image: integer LDS atomics run at the
// plain ds_write rate on gfx950 while ds_add_f32 retires ~3 cycles per lane
// (tools_dev/lds_atomic_bench*.hip); integer sums are also order-independent.
//
// Precision contract (also in include/waldo_hip.h): the sums are exact integers of a quantum 2^-s per
// 8x16-texel SUB-BLOCK of the tile and channel GROUP (the three colour planes share one scale, the
// alpha plane has its own), with s chosen from an upper bound B of any texel sum in the sub-block --
// (number of listed cells whose box reaches the sub-block) x (the largest of their bounds: pixels per
// cell x largest contribution in the cell, as a power of two) -- so that no sum can overflow:
// quantum = 2^(ceil(log2 B) - 29).  Every contribution is rounded to the quantum of the sub-block its
// tap lands in, so the error of a texel is at most (number of taps that reach it) / 2 quanta,
// ABSOLUTE for the sub-block and group: a region whose contributions are small gets a fine quantum
// of its own even when the same tile also holds large ones (round 2 had one quantum per tile).  A
// non-finite contribution anywhere in a listed cell turns the whole tile (all four planes) into NaN.
#include <type_traits>

#include "waldo_common.hip.h"

namespace waldo {

#ifndef WALDO_K2_COLS
#define WALDO_K2_COLS 64
#endif
#ifndef WALDO_K2_WAVES
#define WALDO_K2_WAVES 8
#endif
#ifndef WALDO_K2_ROWS
#define WALDO_K2_ROWS 32
#endif
constexpr int kSrcRows = WALDO_K2_ROWS, kSrcCols = WALDO_K2_COLS;  // S tile (64 columns; 32 measured: see DESIGN)
constexpr int kColShift = kSrcCols == 64 ? 6 : 5;
static_assert((1 << kColShift) == kSrcCols, "S tile: 32 or 64 columns");
constexpr int kSrcTex = kSrcRows * kSrcCols;          // 2048 texels (x4 channels)
constexpr int kCellPix = kCellRows * kCellCols;       // 128
constexpr int kCellShift = kCellPix == 64 ? 6 : (kCellPix == 128 ? 7 : 8);
static_assert((1 << kCellShift) == kCellPix, "cell rows: 4, 8 or 16");
constexpr int kG2Waves = WALDO_K2_WAVES;
constexpr int kG2Threads = kG2Waves * kWave;          // 512
constexpr int kMaxHit = 192 * 8 / kCellRows;         // cells listed per tile (else: slow scan)
constexpr int kScanPer = 2;                           // cells per thread and trip of the table scan

// ONE 16-byte load (a native vector: hipcc splits a HIP int4 struct into two 8-byte loads and sinks
// the second behind the short-circuit test of the first -- two dependent round trips per cell)
__device__ __forceinline__ int4 load_box(const int* cellbox, int64_t idx) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 r = reinterpret_cast<const i32x4*>(cellbox)[idx];
  return make_int4(r[0], -r[1], r[2], -r[3]);  // (x min, x max, y min, y max); empty: x min > x max
}

// LDS image of S: [channel][row][kPitch] int32, kPitch = 64: the LDS serves a ds_add as two 32-lane
// groups over 32 banks; a wave covers 4 pixel rows x 16 columns, so the two rows of a group share
// banks (2-way) -- which costs a 4-byte store or add nothing extra on gfx950 (its address / data
// transfer already takes as long as two array cycles) -- and at 34 KB per workgroup a CU holds
// FOUR workgroups instead of the three that the conflict-free pitch 80 (43 KB) allowed: the
// kernel waits on its dependent loads, not on the LDS (K2 -5.5 % at the headline shape).
// Behind every channel plane, one dump word per lane takes the taps that fall outside
// S (the same word offset in every plane, so a tap's four channel adds differ only in the
// instruction's immediate offset).
#ifndef WALDO_K2_RING
#define WALDO_K2_RING 2
#endif
#ifndef WALDO_K2_PITCH
#define WALDO_K2_PITCH WALDO_K2_COLS
#endif
constexpr int kPitch = WALDO_K2_PITCH;
constexpr int kImgWords = kSrcRows * kPitch;           // image words per channel
constexpr int kPlane = kImgWords + kWave;              // + the dump words
constexpr int kDump = kImgWords;                       // + lane
// WALDO_K2_PK64: two channel planes share a 64-bit word per texel (lo = plane 2q, hi = plane 2q + 1) and a
// tap is added with ONE ds_add_u64 per pair: 8 LDS atomics per pixel instead of 16 (tools_dev/r3_micro.hip:
// 52.8 vs 69.1 LDS cycles per pixel-wave).  The signed 32-bit sums decode exactly from the 64-bit total: the
// low half is the low sum modulo 2^32 (no overflow: the scale guarantees it), and the high half carries the
// high sum plus the borrows of negative low halves, -1 per addend, i.e. minus [low sum < 0] in the end.
#ifndef WALDO_K2_PK64
#define WALDO_K2_PK64 1  // measured: backward 2.133 -> 2.092 ms at the headline shape, bit-identical sums
#endif

// taps of one candidate pixel into the S image.  Branch-free: a tap outside S adds to the lane's
// own dump word -- never to a shared address, where same-address adds would serialise.  (A corner
// outside the LAYER has weight 0 and adds 0 wherever it lands.)  This loop is VALU-issue bound
// (rocprofv3: SQ_INSTS_VALU * 4 cycles ~ 3/4 of the kernel), hence two VALU instructions per add:
// the four corner addresses are selected once, the channel planes are immediate offsets.
typedef float f32x2_k2 __attribute__((ext_vector_type(2)));
constexpr int kSubRows = 8, kSubCols = 16;                       // sub-blocks that carry their own scales
constexpr int kSubX = kSrcCols / kSubCols, kSubY = kSrcRows / kSubRows, kSubs = kSubX * kSubY;
static_assert(kSubs <= 16, "the uniform-scale verdict is one DPP row of 16 lanes, one lane per sub-block (group16_pk_min)");
static_assert(kSubX * kSubCols == kSrcCols && kSubY * kSubRows == kSrcRows, "sub-blocks tile S");

// MIXED: the corners take the scales of the sub-blocks they land in (a table read per corner); else
// the tile has one scale per group (the usual case: the sub-blocks' bounds are alike).
template <bool MIXED>
__device__ __forceinline__ void splat_pixel(int* img, int lane, bool on, const Taps& t,
                                            const float4 rec, float g0, float g1, float g2,
                                            const f32x2_k2* sbscale, f32x2_k2 tile_scale, int sx0, int sy0) {
  const int lx0 = t.x0 - sx0, ly0 = t.y0 - sy0;
  const bool cx0 = (unsigned)lx0 < (unsigned)kSrcCols, cx1 = (unsigned)(lx0 + 1) < (unsigned)kSrcCols;
  const bool cy0 = on && (unsigned)ly0 < (unsigned)kSrcRows, cy1 = on && (unsigned)(ly0 + 1) < (unsigned)kSrcRows;
  // byte offsets into the image (one shift for the four corners; 24-bit multiply: |ly0| is small)
  const int base = (__mul24(ly0, kPitch) + lx0) * 4;
  const int dump = (kDump + lane) * 4;
  char* img8 = reinterpret_cast<char*>(img);
  int* a00 = reinterpret_cast<int*>(img8 + ((cx0 && cy0) ? base : dump));
  int* a01 = reinterpret_cast<int*>(img8 + ((cx1 && cy0) ? base + 4 : dump));
  int* a10 = reinterpret_cast<int*>(img8 + ((cx0 && cy1) ? base + 4 * kPitch : dump));
  int* a11 = reinterpret_cast<int*>(img8 + ((cx1 && cy1) ? base + 4 * kPitch + 4 : dump));
  // the sixteen products on the packed-fp32 pipe, two per instruction: (g w00, g w01), (g w10, g w11)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 wt = {t.w00, t.w01}, wb = {t.w10, t.w11};
  f32x2 wt_rgb = wt, wb_rgb = wb, wt_a = wt, wb_a = wb;
  float gv[4];
  if constexpr (MIXED) {
    // scales (colour, alpha) of the sub-blocks the four corners land in (a corner outside S goes to the
    // dump word: any scale will do, the clamp only keeps the table read in range); the weights carry them
    const int kx0 = min(max(lx0 >> 4, 0), kSubX - 1), kx1 = min(max((lx0 + 1) >> 4, 0), kSubX - 1);
    const int ky0 = min(max(ly0 >> 3, 0), kSubY - 1) * kSubX, ky1 = min(max((ly0 + 1) >> 3, 0), kSubY - 1) * kSubX;
    const f32x2_k2 s00 = sbscale[ky0 + kx0], s01 = sbscale[ky0 + kx1], s10 = sbscale[ky1 + kx0], s11 = sbscale[ky1 + kx1];
    wt_rgb = wt * (f32x2){s00[0], s01[0]};
    wb_rgb = wb * (f32x2){s10[0], s11[0]};
    wt_a = wt * (f32x2){s00[1], s01[1]};
    wb_a = wb * (f32x2){s10[1], s11[1]};
    gv[0] = rec.x * g0;
    gv[1] = rec.x * g1;
    gv[2] = rec.x * g2;
    gv[3] = rec.y;
  } else {
    const float as = rec.x * tile_scale[0];
    const f32x2 g01 = (f32x2){as, as} * (f32x2){g0, g1};
    gv[0] = g01[0];
    gv[1] = g01[1];
    gv[2] = as * g2;
    gv[3] = rec.y * tile_scale[1];
  }
#if WALDO_K2_PK64
  // the 64-bit image: pair q of texel word w at 8 * (q * kPlane + w); a00 .. a11 hold 4 * w
  auto add2 = [&](int* a, int q, float vlo, float vhi) {
    const int lo = cvt_round(vlo), hi = cvt_round(vhi);
    const unsigned long long v = (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)(hi + (lo >> 31)) << 32);
    atomicAdd(reinterpret_cast<unsigned long long*>(img) + q * kPlane + (a - img), v);
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x2 g0s = {gv[2 * q], gv[2 * q]}, g1s = {gv[2 * q + 1], gv[2 * q + 1]};
    const f32x2 pt0 = g0s * wt_rgb, pb0 = g0s * wb_rgb;
    const f32x2 pt1 = g1s * (q == 0 ? wt_rgb : wt_a), pb1 = g1s * (q == 0 ? wb_rgb : wb_a);
    add2(a00, q, pt0[0], pt1[0]);
    add2(a01, q, pt0[1], pt1[1]);
    add2(a10, q, pb0[0], pb1[0]);
    add2(a11, q, pb0[1], pb1[1]);
  }
#else
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x2 gs = {gv[c], gv[c]};
    const f32x2 pt = gs * (c < 3 ? wt_rgb : wt_a), pb = gs * (c < 3 ? wb_rgb : wb_a);
    atomicAdd(a00 + c * kPlane, cvt_round(pt[0]));
    atomicAdd(a01 + c * kPlane, cvt_round(pt[1]));
    atomicAdd(a10 + c * kPlane, cvt_round(pb[0]));
    atomicAdd(a11 + c * kPlane, cvt_round(pb[1]));
  }
#endif
}

// does any tap of this pixel fall into S?
__device__ __forceinline__ bool touches(const TapCore& t, int sx0, int sy0) {
  const int lx0 = t.x0 - sx0, ly0 = t.y0 - sy0;
  return lx0 >= -1 && lx0 < kSrcCols && ly0 >= -1 && ly0 < kSrcRows;
}

__global__ __launch_bounds__(kG2Threads, 2) void warp_composite_splat_kernel(
    const float4* __restrict__ rec, const float* __restrict__ grad_rgb,
    const int* __restrict__ cellbox, const unsigned* __restrict__ cellbound,
    float* __restrict__ grad_layers, int F, int L, int H, int W, int nsx, int nstiles, int nbands,
    int ncx, int ncells) {
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  // a FRAME is pinned to one XCD, source-tile-major inside it: the L layer planes of one source
  // tile run back to back.  Their candidate pixels are (nearly) the same neighbourhood of the
  // frame, so its grad_rgb is read once from HBM instead of once per layer (it was 40 % of this
  // kernel's FETCH_SIZE with layer-major order); the records of a plane are still read ~1.8
  // times by neighbouring tiles, from L2.
  int fi, stile, layer;
  if (!xcd_decode_banded(blockIdx.x, F, nbands, nstiles, L, fi, stile, layer)) return;
  const int64_t f = fi;
  const int64_t fl = f * L + layer;
  const int sx0 = (stile % nsx) * kSrcCols, sy0 = (stile / nsx) * kSrcRows;
  const int sx1 = min(sx0 + kSrcCols, W) - 1, sy1 = min(sy0 + kSrcRows, H) - 1;

  __shared__ __attribute__((aligned(16))) int lds[4 * kPlane + kMaxHit + kScanPer * kG2Waves + 5 * kSubs];
  int* img = lds;
  int* hitlist = lds + 4 * kPlane;
  int* wcount = hitlist + kMaxHit;                      // [cell of the trip][wave]: hits
  // per sub-block: listed cells whose box reaches it, the largest of their bounds (float bits; positive
  // floats order like their bits: integer atomics, deterministic), then the scales derived from them
  int* sbcnt = wcount + kScanPer * kG2Waves;
  unsigned* sbmax = reinterpret_cast<unsigned*>(sbcnt + kSubs);          // [group][sub-block]
  f32x2_k2* sbscale = reinterpret_cast<f32x2_k2*>(sbmax + 2 * kSubs);    // (colour, alpha) per sub-block
  static_assert((4 * kPlane + kMaxHit + kScanPer * kG2Waves + 3 * kSubs) % 2 == 0, "8-byte aligned scale table");


  auto reaches = [&](const int4 ob) {  // bitwise: no short-circuit branches between the comparisons
    return (bool)((ob.x <= ob.y) & (ob.x <= sx1) & (ob.y >= sx0) & (ob.z <= sy1) & (ob.w >= sy0));
  };
  // ---- cells whose box reaches S, listed in cell order, and the sums of their contribution
  // bounds (fixed butterfly, fixed wave order: deterministic).  kScanPer cells per thread and
  // trip -- their loads are independent and in flight together, one barrier pair per
  // kScanPer * kG2Threads cells (one trip at 256x512); ballot-based compaction inside a wave.
  int nhit = 0;
  const float cell_rows = (float)kCellRows;  // K1 publishes a bound of a 16-pixel row sum
  for (int c0 = 0; c0 < ncells; c0 += kScanPer * kG2Threads) {
    bool hit[kScanPer];
    unsigned eb[kScanPer];
    int4 box[kScanPer];
    // every load of the trip is issued before the first test (unconditional: the index is clamped);
    // the empty asm pins the bounds' loads here -- left alone hipcc sinks them behind the box test,
    // one more dependent round trip per trip
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
      const int c = min(c0 + u * kG2Threads + (int)threadIdx.x, ncells - 1);
      box[u] = load_box(cellbox, fl * ncells + c);
      eb[u] = cellbound[fl * ncells + c];
    }
    if (c0 == 0) {  // the image is cleared while the first trip's loads are in flight
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      static_assert((4 * kPlane) % 4 == 0, "16-byte zero fill");
      for (int e = threadIdx.x; e < kPlane; e += kG2Threads) reinterpret_cast<i32x4*>(img)[e] = (i32x4){0, 0, 0, 0};
      if (threadIdx.x < 3 * kSubs) sbcnt[threadIdx.x] = 0;  // sbcnt and both halves of sbmax (first added to behind the barrier below)
    }
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) asm volatile("" : "+v"(eb[u]));
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
      hit[u] = (bool)((c0 + u * kG2Threads + (int)threadIdx.x < ncells) & reaches(box[u]));
      eb[u] = hit[u] ? eb[u] : 0u;
    }
    int cnt[kScanPer];
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
      const unsigned long long m = __ballot(hit[u]);
      cnt[u] = __popcll(m);
      if (lane == 0) wcount[u * kG2Waves + wave] = cnt[u];
      cnt[u] = __popcll(m & ((1ull << lane) - 1ull));  // hits of this wave before this lane
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
      int base = nhit;
#pragma unroll
      for (int w = 0; w < kG2Waves; ++w) {
        if (w < wave) base += wcount[u * kG2Waves + w];
        nhit += wcount[u * kG2Waves + w];
      }
      // listed as the cell's pixel origin (row << 16 | column): the division happens once per cell
      const int c = c0 + u * kG2Threads + (int)threadIdx.x;
      if (hit[u] && base + cnt[u] < kMaxHit)
        hitlist[base + cnt[u]] = (((c / ncx) * kCellRows) << 16) | ((c % ncx) * kCellCols);
      if (hit[u]) {
        // the cell's bounds go to every sub-block its box reaches (exponent 255 = infinity: poison)
        const unsigned brgb = __float_as_uint(__uint_as_float((eb[u] & 0xffu) << 23) * cell_rows);
        const unsigned ba = __float_as_uint(__uint_as_float(((eb[u] >> 8) & 0xffu) << 23) * cell_rows);
        const int kx0 = max((box[u].x - sx0) >> 4, 0), kx1 = min((box[u].y - sx0) >> 4, kSubX - 1);
        const int ky0 = max((box[u].z - sy0) >> 3, 0), ky1 = min((box[u].w - sy0) >> 3, kSubY - 1);
        for (int ky = ky0; ky <= ky1; ++ky)
          for (int kx = kx0; kx <= kx1; ++kx) {
            atomicAdd(&sbcnt[ky * kSubX + kx], 1);
            atomicMax(&sbmax[ky * kSubX + kx], brgb);
            atomicMax(&sbmax[kSubs + ky * kSubX + kx], ba);
          }
      }
    }
    __syncthreads();
  }
  // fixed-point scales per sub-block and group: B = (cells that reach the sub-block) x (their largest
  // bound) bounds the magnitude of ANY texel sum there (bilinear weights are <= 1);
  // scale = 2^(29 - floor(log2 B)) keeps |sum| * scale < 2^30
  auto scale_exp = [](float b) {
    const int eB = (int)((__float_as_uint(b) >> 23) & 0xffu) - 127;
    return min(29 - eB, 126);
  };
  // Sub-blocks whose bounds are alike share the tile's coarsest exponent (one scale per group: the corners
  // of a pixel then need no table); only when some sub-block could take a quantum at least 2^4 finer --
  // a region of small gradients next to large ones -- does every sub-block get its own.  Sixteen lanes
  // (one per sub-block) decide, with DPP reductions over their row of 16; the verdict goes through LDS.
  int* verdict = reinterpret_cast<int*>(sbmax);  // reused once the table is read: [mixed, any, poison]
  if (threadIdx.x < kWave) {
    const int k = min((int)threadIdx.x, kSubs - 1);
    const int cnt = sbcnt[k];
    const float n = (float)cnt;
    const float br = n * __uint_as_float(sbmax[k]), ba = n * __uint_as_float(sbmax[kSubs + k]);
    // a sub-block no listed cell reaches receives no tap: it takes no part in the decision
    const int er = scale_exp(br), ea = scale_exp(ba);
    // exponents + 200 are positive 16-bit numbers: packed minima / maxima (as minima of 1000 - x) over
    // the row of 16 lanes with DPP rotations
    int lo = (cnt ? er + 200 : 326) | ((cnt ? ea + 200 : 326) << 16);
    int hi = (cnt ? 800 - er : 1000) | ((cnt ? 800 - ea : 1000) << 16);
    lo = group16_pk_min(lo);
    hi = group16_pk_min(hi);
    int bmi = (int)__float_as_uint(fmaxf(br, ba));  // non-negative floats order like their bits
    bmi = max(bmi, row_ror_i<8>(bmi));
    bmi = max(bmi, row_ror_i<4>(bmi));
    bmi = max(bmi, row_ror_i<2>(bmi));
    bmi = max(bmi, row_ror_i<1>(bmi));
    const float bm = __uint_as_float((unsigned)bmi);
    const int lo_r = (lo & 0xffff) - 200, lo_a = ((lo >> 16) & 0xffff) - 200;
    const int hi_r = 800 - (hi & 0xffff), hi_a = 800 - ((hi >> 16) & 0xffff);  // -200: no sub-block is reached
    const bool mix = (hi_r - lo_r >= 4) || (hi_a - lo_a >= 4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has read sbmax before the verdict overwrites it
    if (threadIdx.x < kSubs)
      sbscale[k] = (f32x2_k2){__uint_as_float((unsigned)(127 + (mix ? er : lo_r)) << 23),
                              __uint_as_float((unsigned)(127 + (mix ? ea : lo_a)) << 23)};
    if (threadIdx.x == 0) {
      verdict[0] = mix;
      verdict[1] = bm > 0.0f;
      verdict[2] = !(bm < __builtin_huge_valf());
      verdict[3] = lo;  // the coarsest exponents (+ 200, packed): the tile's scales when not mixed
    }
  }
  __syncthreads();
  const bool mixed = verdict[0] != 0;  // block-uniform
  // an infinity / NaN in a listed cell (exponent 255 from K1) or bounds that overflow: the sums
  // cannot be represented -- the tile becomes NaN, as loud as the reference's gradient would be
  const bool poison = verdict[2] != 0;
  const float bsum_rgb = verdict[1] ? 1.0f : 0.0f, bsum_a = 0.0f;
  const f32x2_k2 tile_scale = {__uint_as_float((unsigned)(127 - 200 + (verdict[3] & 0xffff)) << 23),
                               __uint_as_float((unsigned)(127 - 200 + ((verdict[3] >> 16) & 0xffff)) << 23)};
  const float* gplane = grad_rgb + f * 3 * HW;
  const float4* rcp = rec + ((int64_t)WALDO_REC_FRAME(f) * L + layer) * HW;

  if ((bsum_rgb > 0.0f || bsum_a > 0.0f) && !poison) {
    if (nhit <= kMaxHit) {
      // candidates: 128 pixels per listed cell; a wave takes 4 rows x 16 columns of one cell.
      // Two candidates per thread are in flight (ping-pong A / B; deeper rings measured slower):
      // the loop is a chain
      // record -> taps -> LDS adds, and with one candidate per trip the kernel mostly waits for
      // the record loads (rocprofv3: SQ_WAIT_ANY ~ 60 % of the wave cycles).
      const int total = nhit * kCellPix;
      struct Cand {
        unsigned p;
        bool livep;
        float4 rc;  // (grid x, grid y, a', g_alpha)
        float g0, g1, g2;
      };
      auto fetch = [&](int i, Cand& k) {
        const int ic = min(i, total - 1);  // past the end: a valid candidate, masked by livep
        const int org = hitlist[ic >> kCellShift];
        const int within = ic & (kCellPix - 1);
        const int py = (org >> 16) + (within >> 4);
        const int px = (org & 0xffff) + (within & 15);
        k.livep = i < total && py < H && px < W;
        k.p = (unsigned)(__mul24(min(py, H - 1), W) + min(px, W - 1));
        // 32-bit byte offsets from uniform bases (HW * 8 < 2^32 is implied by the launcher's check)
#if WALDO_REC_LOAD_NT
        {
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(rcp) + k.p * 16u));
          k.rc = make_float4(r[0], r[1], r[2], r[3]);
        }
#else
        k.rc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(rcp) + k.p * 16u);
#endif
        k.g0 = ldb(gplane, k.p * 4u);
        k.g1 = ldb(gplane + HW, k.p * 4u);
        k.g2 = ldb(gplane + 2 * HW, k.p * 4u);
      };
      auto splat = [&](auto mixed_tag, const Cand& k) {
        constexpr bool MIXED = decltype(mixed_tag)::value;
        const TapCore tc = tap_core_px(k.rc.x, k.rc.y, H, W);
        const bool any = k.livep && touches(tc, sx0, sy0);
        // a wave = 4 rows x 16 columns of one cell: skip the adds when none of its taps reach S
        if (__ballot(any) != 0ull) {
          Taps t;
          if (__ballot(!tap_interior(tc, H, W)) == 0ull) {
            // wave-uniform: all corners inside the layer, every validity factor is exactly 1
            const float wx0 = 1.0f - tc.fx, wy0 = 1.0f - tc.fy;
            t.x0 = tc.x0;
            t.y0 = tc.y0;
            t.w00 = wx0 * wy0;
            t.w01 = tc.fx * wy0;
            t.w10 = wx0 * tc.fy;
            t.w11 = tc.fx * tc.fy;
          } else {
            t = finish_taps(tc, H, W);
          }
          splat_pixel<MIXED>(img, lane, any, t, make_float4(k.rc.z, k.rc.w, k.rc.x, k.rc.y), k.g0, k.g1, k.g2,
                             sbscale, tile_scale, sx0, sy0);
        }
      };
      // ring of kRing candidates per thread in flight (2: with 3 / 4 the backward is 1 % / 1.7 %
      // slower): slot d is consumed, then refilled with the
      // candidate kRing trips ahead (fetch() clamps past the end: harmless, masked by livep).  The
      // steady-state loop has no branch between a splat and the next fetch (a join makes hipcc wait
      // for every outstanding load); the last, partial round is peeled.
      constexpr int kRing = WALDO_K2_RING;
      auto run = [&](auto mixed_tag) {
        Cand k[kRing];
        int i = threadIdx.x;
#pragma unroll
        for (int d = 0; d < kRing; ++d) fetch(i + d * kG2Threads, k[d]);
        while (i + (kRing - 1) * kG2Threads < total) {  // uniform per wave: total is a multiple of 128
#pragma unroll
          for (int d = 0; d < kRing; ++d) {
            splat(mixed_tag, k[d]);
            fetch(i + (d + kRing) * kG2Threads, k[d]);
          }
          i += kRing * kG2Threads;
        }
#pragma unroll
        for (int d = 0; d < kRing - 1; ++d)
          if (i + d * kG2Threads < total) splat(mixed_tag, k[d]);
      };
      if (mixed) run(std::true_type{}); else run(std::false_type{});  // block-uniform
    } else {
      // violent warp (more cells reach S than the list holds): scan every cell, wave-uniformly
      for (int c = 0; c < ncells; ++c) {
        const bool hit = reaches(load_box(cellbox, fl * ncells + c));
        if (!hit || threadIdx.x >= kCellPix) continue;
        const int py = (c / ncx) * kCellRows + (threadIdx.x >> 4);
        const int px = (c % ncx) * kCellCols + (threadIdx.x & 15);
        if (py < H && px < W) {
          const unsigned p = (unsigned)(__mul24(py, W) + px);
          const float4 rc = rcp[p];
          const float4 rec = make_float4(rc.z, rc.w, rc.x, rc.y);
          const Taps t = make_taps_px(rec.z, rec.w, H, W);
          splat_pixel<true>(img, lane, true, t, rec, gplane[p], (gplane + HW)[p], (gplane + 2 * HW)[p],
                            sbscale, tile_scale, sx0, sy0);
        }
      }
    }
  }
  __syncthreads();
  // ---- S is ours alone: plain row-coalesced stores (zeros included)
  float* gbase = grad_layers + fl * 4 * HW;
  if (poison) {  // block-uniform
    const float qnan = __builtin_nanf("");
    for (int e = threadIdx.x; e < kSrcTex; e += kG2Threads) {
      const int y = sy0 + (e >> kColShift), x = sx0 + (e & (kSrcCols - 1));
      if (y < H && x < W) {
#pragma unroll
        for (int c = 0; c < 4; ++c) (gbase + c * HW)[(unsigned)(__mul24(y, W) + x)] = qnan;
      }
    }
    return;
  }
  if ((W & 3) == 0) {  // 16 bytes per lane: four texels of a row (rows are 16-byte aligned)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < kSrcTex / 4; e += kG2Threads) {
      const int y = sy0 + (e >> (kColShift - 2)), x = sx0 + 4 * (e & (kSrcCols / 4 - 1));
      if (y < H && x < W) {
        const unsigned doff = (unsigned)(__mul24(y, W) + x);
        const int li = (e >> (kColShift - 2)) * kPitch + 4 * (e & (kSrcCols / 4 - 1));
        // the four texels of a lane lie in one sub-block; 1 / 2^s = 2^(254 - biased exponent)
        const f32x2_k2 sc = sbscale[((y - sy0) >> 3) * kSubX + ((x - sx0) >> 4)];
        const float inv_rgb = __uint_as_float(0x7f000000u - __float_as_uint(sc[0]));
        const float inv_a = __uint_as_float(0x7f000000u - __float_as_uint(sc[1]));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#if WALDO_K2_PK64
          const i32x4* w64 = reinterpret_cast<const i32x4*>(reinterpret_cast<const long long*>(img) + (c >> 1) * kPlane + li);
          const i32x4 wa = w64[0], wb = w64[1];  // (lo, hi) of texels li .. li + 3
          const i32x4 los = {wa[0], wa[2], wb[0], wb[2]}, his = {wa[1], wa[3], wb[1], wb[3]};
          const i32x4 v = (c & 1) ? his - (los >> 31) : los;
#else
          const i32x4 v = *reinterpret_cast<const i32x4*>(img + c * kPlane + li);
#endif
          const float inv = c < 3 ? inv_rgb : inv_a;
          stream_store16<WALDO_GRAD_STORE_POLICY>(gbase + c * HW, doff * 4u, HW * 4,
                                                  (f32x4){(float)v[0] * inv, (float)v[1] * inv, (float)v[2] * inv, (float)v[3] * inv});
        }
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < kSrcTex; e += kG2Threads) {
    const int y = sy0 + (e >> kColShift), x = sx0 + (e & (kSrcCols - 1));
    if (y < H && x < W) {
      const unsigned doff = (unsigned)(__mul24(y, W) + x);
      const int li = (e >> kColShift) * kPitch + (e & (kSrcCols - 1));
      const f32x2_k2 sc = sbscale[((y - sy0) >> 3) * kSubX + ((x - sx0) >> 4)];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#if WALDO_K2_PK64
        const int lo = img[2 * ((c >> 1) * kPlane + li)], hi = img[2 * ((c >> 1) * kPlane + li) + 1];
        const int v = (c & 1) ? hi - (lo >> 31) : lo;
#else
        const int v = img[c * kPlane + li];
#endif
        (gbase + c * HW)[doff] = (float)v * __uint_as_float(0x7f000000u - __float_as_uint(sc[c < 3 ? 0 : 1]));
      }
    }
  }
}

void launch_splat(const float* rec, const float* grad_rgb, const int* cellbox,
                  const unsigned* cellbound, float* grad_layers, int F, int L, int H, int W,
                  hipStream_t st) {
  const int nsx = (W + kSrcCols - 1) / kSrcCols, nsy = (H + kSrcRows - 1) / kSrcRows;
  const int ncx = (W + kCellCols - 1) / kCellCols, ncy = (H + kCellRows - 1) / kCellRows;
  const int nbands = xcd_bands(F);
  dim3 grid((unsigned)xcd_grid_banded(F, nbands, nsx * nsy, L));
  hipLaunchKernelGGL(warp_composite_splat_kernel, grid, dim3(kG2Threads), 0, st,
                     reinterpret_cast<const float4*>(rec),
                     grad_rgb, cellbox, cellbound, grad_layers, F, L, H, W, nsx, nsx * nsy, nbands, ncx,
                     ncx * ncy);
}

}  // namespace waldo
