// Forward of the fused path with LDS-staged sampling (K3 == 19, 4 | W).
//
// Same sampling / compositing arithmetic as warp_composite_fwd_kernel, different data movement
// and a different pipe for the TPS grid:
//  * the plain kernel fetches every texel four times through the vector-memory path (each lane
//    loads its own 2x2 taps, 4 bytes per lane).  Here a workgroup (16 x 16 pixels, four rows per
//    wavefront -- square tiles: the warp's skew over a 64-pixel-wide tile makes its footprint box
//    ~4x the tile, over 16 pixels ~1.5x) finds, per layer, the bounding box of the 2x2 blocks its
//    pixels read, streams that box ONCE from global memory (every lane two texels of the four
//    channel planes) and takes the bilinear taps out of LDS;
//  * with sampling off the vector-memory path the kernel is VALU-issue bound (rocprofv3:
//    SQ_INSTS_VALU * 4 cycles ~ 80 % of its duration), so the TPS grid of all layers is one
//    v_mfma_f32_16x16x4_f32 chain per 16 pixels (N = (layer, xy) columns), and the boxes come from
//    the min / max of the grid coordinates in the accumulator layout (a lane holds 16 pixels of
//    one column) instead of per-layer cross-lane reductions of tap indices.
// A rolling window of layers is in flight (registers -> LDS -> taps), double-buffered in LDS (one
// barrier per layer).  A box that does not fit the LDS image (violent warp) falls back to
// gathering that layer straight from memory.
//
// Per-(pixel, layer) arithmetic is kept minimal (round 2: what binds the kernel is a mix of VALU,
// LDS and latency at four waves per SIMD -- DESIGN.md section 4):
//  * the grid comes out of the MFMA chain already in PIXEL units (scaled_map(): the
//    un-normalisation of grid_sample is folded into the B operand);
//  * the staged image keeps a texel's four channels together (float4), so a tap arrives as one
//    16-byte LDS read whose halves are packed-fp32 operands: four ds_read_b128 and twelve v_pk_*
//    instructions per (pixel, layer) in the bilinear "lerp" form
//    top + fy (bot - top), top = p00 + fx (p01 - p00), instead of sixteen fmas on four corner
//    weights (same value up to rounding);
//  * whatever is wave-uniform stays scalar: the wave index is read with readfirstlane, plane bases
//    are SGPR pairs and lanes carry 32-bit offsets (122 VGPRs at L = 8: four waves per SIMD).
//
// Tap blocks: instead of clamping the four corners separately, a pixel reads the 2x2 block at
// (xb, yb) = clamp((x0, y0), 0, (W-2, H-2)), which lies inside the layer; the block's cells are
// re-assigned to the corners when x0 / y0 was clamped (only within one texel of the border), and
// corners outside the layer are multiplied by a validity of 0 exactly as in tap_sample().
#pragma once
// included at the end of warp_composite_kernels.hip.h (uses its tps_eval / pixel_of / opaque)

namespace waldo {

#ifndef WALDO_FWD12_WAVES
#define WALDO_FWD12_WAVES 4  // waves per SIMD the L = 12 forward is compiled for (125 VGPRs without a spill once the
                             // per-thread indices are re-materialised per frame; L = 9 .. 11 spills there: 3)
#endif
#ifndef WALDO_FWD8_WAVES
#define WALDO_FWD8_WAVES 3   // register cap of the L <= 8 forward (it needs 109: four waves per SIMD)
#endif
// Wave priority (s_setprio) around the phases that ISSUE memory operations, so that a wavefront about to put loads or
// stores in flight does not queue behind the other workgroups' tap arithmetic.  Level 0: off.  The masks pick the
// phases: bit 0 a layer's staging stores + the next box loads, bit 1 a frame's first box loads (and K1's output-gradient
// loads), bit 2 the output stores (K1: the records), bit 3 K1's MFMA contraction.
// Measured (round 5, tools_dev/ab_bench.py, three boxes, same bits): K1 with bits 0 + 1: backward 2.050 -> 2.005 ms on
// one box, 2.045 -> 2.035 and 2.060 -> 2.047 on two others (levels 1 / 2 / 3 alike); with the record stores or the MFMA
// phase as well: nothing more; the forward: no change with any of its bits (+1 % with bit 2); K2 around its candidate
// loads: +2.5 % (worse).  So: K1 alone, its loads alone.
#ifndef WALDO_STAGE_PRIO
#define WALDO_STAGE_PRIO 1
#endif
#ifndef WALDO_FWD_PRIO_MASK
#define WALDO_FWD_PRIO_MASK 0
#endif
#ifndef WALDO_K1_PRIO_MASK
#define WALDO_K1_PRIO_MASK 3
#endif
#define WALDO_PRIO_ON(mask, bit) do { if (WALDO_STAGE_PRIO && ((mask) & (bit))) __builtin_amdgcn_s_setprio(WALDO_STAGE_PRIO); } while (0)
#define WALDO_PRIO_OFF(mask, bit) do { if (WALDO_STAGE_PRIO && ((mask) & (bit))) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef WALDO_STAGE_AHEAD
#define WALDO_STAGE_AHEAD 2  // layers whose box loads are in flight at a time (measured 2 / 3 / 4 / 6 / 8:
                             // fwd 0.726 / 0.728 / 0.736 / 0.836 / 0.990 ms, bwd 2.222 / 2.225 / 2.242 / 2.58 / 2.59)
#endif

// Diagnostic build only (-DWALDO_FWD_STAMPS): s_memtime stamps of wave 0 at the phase boundaries of the first
// frame of a workgroup's chunk (tools_dev/fwd_stamps.py); never compiled into the product library.
#ifdef WALDO_FWD_STAMPS
constexpr int kFwdStampSlots = 16, kFwdStampBlocks = 4096;
__device__ unsigned long long waldo_fwd_stamps[kFwdStampBlocks * kFwdStampSlots];
#define WALDO_FSTAMP(i)                                                                  \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < kFwdStampBlocks)                                \
      waldo_fwd_stamps[blockIdx.x * kFwdStampSlots + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WALDO_FSTAMP(i) do { } while (0)
#endif

struct BoxTaps {
  float fx, fy;                  // fractional parts
  float v00, v01, v10, v11;      // 1 / 0 validity of the corners
  int xb, yb;                    // origin of the 2x2 block that is read (inside the layer)
  int cs, rs;                    // x0 - xb, y0 - yb: 0 in the interior, +-1 at a clamped border
};

__device__ __forceinline__ BoxTaps make_box_taps(const TapCore& tc, int Hi, int Wi) {
  const Taps t = finish_taps(tc, Hi, Wi);
  BoxTaps p;
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

// first texel column / row of the 2x2 block a coordinate reads: the xb / yb of make_box_taps
// (same instruction chain; monotonic in c, so the block origins of a set of pixels lie between
// the origins of the set's smallest and largest coordinate)
__device__ __forceinline__ int block_origin(float i, int size) {  // i: pixel units
  i = __builtin_amdgcn_fmed3f(i, -2.0f, (float)size + 1.0f);
  return min(max((int)floorf(i), 0), size - 2);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- staged image of one layer's footprint box: float4 texels (all four channels of a texel in
// 16 bytes), row-major with pitch bw.  A tap is ONE 16-byte LDS read (ds_read_b128: 4 LDS cycles
// per wave, against 8 for the ds_read2_b64 of a two-channel image) that arrives as two packed-fp32
// operands.  Staging item = two consecutive texels of a row: a lane loads them from the four
// channel planes (8 bytes each), transposes with four v_pk_mov and writes two texels.
constexpr int kImgBufFloats = 4 * kStageCap;  // one layer image

struct StageRegs {
  f32x2_t c0, c1, c2, c3;  // texels (t, t + 1) of the four channel planes
};

// (a.lo, b.lo) and (a.hi, b.hi) of two register pairs: v_pk_mov_b32 takes its low result from
// src0[op_sel[0]] and its high result from src1[op_sel[1]] (hipcc 7.2 does not form it by itself)
__device__ __forceinline__ f32x2_t pk_lo(f32x2_t a, f32x2_t b) {
  f32x2_t d;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2_t pk_hi(f32x2_t a, f32x2_t b) {
  f32x2_t d;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ void stage_store(float* imgbuf, int item, const StageRegs& r) {
  f32x4_t* dst = reinterpret_cast<f32x4_t*>(imgbuf + 8 * item);
  dst[0] = __builtin_shufflevector(pk_lo(r.c0, r.c1), pk_lo(r.c2, r.c3), 0, 1, 2, 3);
  dst[1] = __builtin_shufflevector(pk_hi(r.c0, r.c1), pk_hi(r.c2, r.c3), 0, 1, 2, 3);
}

// 8 bytes at a 32-bit byte offset from a wave-uniform base (scalar base + VGPR offset addressing)
__device__ __forceinline__ f32x2_t ld8(const float* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const f32x2_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

// the four taps at texel index idx (row-major, pitch bw) of a layer image, as channel pairs
struct PairBlock {
  f32x2_t p00[2], p01[2], p10[2], p11[2];  // [pair]: channels (0, 1) and (2, 3)
};

__device__ __forceinline__ PairBlock read_block(const float* imgbuf, int idx, int bw) {
  const f32x4_t* r0 = reinterpret_cast<const f32x4_t*>(imgbuf + 4 * idx);
  const f32x4_t* r1 = r0 + bw;
  const f32x4_t t00 = r0[0], t01 = r0[1], t10 = r1[0], t11 = r1[1];
  PairBlock b;
  b.p00[0] = __builtin_shufflevector(t00, t00, 0, 1);
  b.p00[1] = __builtin_shufflevector(t00, t00, 2, 3);
  b.p01[0] = __builtin_shufflevector(t01, t01, 0, 1);
  b.p01[1] = __builtin_shufflevector(t01, t01, 2, 3);
  b.p10[0] = __builtin_shufflevector(t10, t10, 0, 1);
  b.p10[1] = __builtin_shufflevector(t10, t10, 2, 3);
  b.p11[0] = __builtin_shufflevector(t11, t11, 0, 1);
  b.p11[1] = __builtin_shufflevector(t11, t11, 2, 3);
  return b;
}

// the block was read at the clamped origin (xb, yb); re-assign its cells to the corners of the
// footprint when x0 / y0 was clamped (cs = x0 - xb, rs = y0 - yb: 0 in the interior, +-1 within
// one texel of the border; corners outside the layer carry zero weight / validity)
__device__ __forceinline__ PairBlock assign_corners(const PairBlock& raw, int cs, int rs) {
  PairBlock o;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x2_t a0 = raw.p00[q], b0 = raw.p01[q], a1 = raw.p10[q], b1 = raw.p11[q];
    const f32x2_t ta = rs > 0 ? a1 : a0, tb = rs > 0 ? b1 : b0;  // corner row y0
    const f32x2_t ua = rs < 0 ? a0 : a1, ub = rs < 0 ? b0 : b1;  // corner row y0 + 1
    o.p00[q] = cs > 0 ? tb : ta;
    o.p01[q] = cs < 0 ? ta : tb;
    o.p10[q] = cs > 0 ? ub : ua;
    o.p11[q] = cs < 0 ? ua : ub;
  }
  return o;
}

// bilinear value of a pair in the lerp form
__device__ __forceinline__ f32x2_t lerp2(const f32x2_t p00, const f32x2_t p01, const f32x2_t p10,
                                         const f32x2_t p11, float fx, float fy) {
  const f32x2_t fx2 = {fx, fx}, fy2 = {fy, fy};
  const f32x2_t top = __builtin_elementwise_fma(fx2, p01 - p00, p00);
  const f32x2_t bot = __builtin_elementwise_fma(fx2, p11 - p10, p10);
  return __builtin_elementwise_fma(fy2, bot - top, top);
}

// At least 2 waves per SIMD (<= 256 VGPRs) for every LP: with 1 (LP >= 24 wants ~310 registers)
// hipcc 7.2 parks MFMA accumulator components in AGPRs and reads some of them back wrong
// (tools_dev/dbg_fwd.py: columns 4k of tile rows 0 and 3, layers 8..15); spilling is correct.
// FOLD: `mapping` is not read; the workgroup computes the TPS mapping of its frame's layers itself,
// mapping = inverse_kernel (19 x 19) @ [src_pts (16 x 2); 0] (warp.py:52-53), with the fma order of
// tps_mapping_fwd_kernel (bit-identical values), one frame ahead, into an LDS table -- the
// forward-only path then is ONE kernel (at 8 frames of 128 x 128 the separate mapping kernel and
// its launch gap were a quarter of the call).
// Measured and dropped in round 4 ("all layers staged" for short launches: every layer's box loads issued at once,
// LP images in LDS, ONE barrier per tile-frame instead of L): bit-identical, and no faster -- 12.2 us against 11.8 at
// BASELINE config C2.  tools_dev/fwd_stamps.py shows why: of a C2 tile-frame's ~10 us a third is the first memory
// round trip (basis operand, mapping fold), a sixth the MFMA grid + transposition, and the L staged layer steps cost
// 4.4 us rolling against 3.1 + 1.8 (issue) all at once.  DESIGN.md section 4c.
// NW: wavefronts per workgroup.  4: a 16 x 16 tile (wave w = rows 4w .. 4w+3).  8: a 16 x 32 tile, waves 4 .. 7 on its
// right half -- the footprint box of the wider tile has less halo per pixel (round 4's experiment: DESIGN.md
// section 4); the staged image holds 128 NW texels, one two-texel item per lane as before.
template <int LP, bool EXL, bool FOLD, int NW = 4>
__global__ __launch_bounds__(NW * kWave, (NW == 8 ? 1 : 1) * (LP <= 8 ? WALDO_FWD8_WAVES : (LP <= 12 ? (EXL ? WALDO_FWD12_WAVES : 3) : 2))) void warp_composite_fwd_lds_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ inv_kernel,
    const float* __restrict__ src_pts, const float* __restrict__ occ, float* __restrict__ rgb,
    float* __restrict__ alpha_out, int F, int Lrt, int H, int W, int frames_per_block, int ntx,
    int ntiles, int nchunks, int nbands, float delta) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vectors stay in registers
  constexpr int K3 = 19, KS = (K3 + 3) / 4;
  constexpr int NC = 2 * LP, NT = (NC + 15) / 16, GGC = NT * 16, TP = GGC + 1;
  constexpr int kThreads = NW * kWave, kCap = kStageCap * NW / 4, kBuf = 4 * kCap;  // texels / floats of one layer image
  constexpr int kImgFloats = 2 * kBuf;            // two buffers of float4 texels
  constexpr int kTFloats = NW * kWave * TP;       // per-wave transposition slices of the grid
  constexpr int kMain = kImgFloats > kTFloats ? kImgFloats : kTFloats;
  const int L = EXL ? LP : Lrt;
  const int64_t HW = (int64_t)H * W;
  // wave index as a scalar: everything derived from it (tile rows, the staged channel pair, the
  // plane pointers of the staging loads) stays in SGPRs
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int arow = lane & 15, kk = lane >> 4;
  int chunk, tile, rest_;
  if (!xcd_decode_banded(blockIdx.x, nchunks, nbands, ntiles, 1, chunk, tile, rest_)) return;
  WALDO_FSTAMP(0);
  const int col0 = (tile % ntx) * (kLdsTile * NW / 4) + (wave >> 2) * kLdsTile;
  const int row0 = (tile / ntx) * kLdsTile + (wave & 3) * 4;
  // 16 x 16 tile: wave w covers rows 4w .. 4w+3, lane -> (row 4w + lane / 16, column lane % 16)
  PixelMap pm;
  pm.live = col0 + arow < W && row0 + kk < H;
  pm.p = (int64_t)min(row0 + kk, H - 1) * W + min(col0 + arow, W - 1);
  const int64_t p = pm.p;

  constexpr int kMapFloats = FOLD ? 2 * LP * K3 * 2 : 0;  // two frames' mapping tables
  __shared__ __attribute__((aligned(16))) float lds[kMain + NW * GGC * 2 + kMapFloats];
  float* img = lds;
  float* boxred = lds + kMain;  // [wave][column][min, max]
  float* smap = boxred + NW * GGC * 2;
  // mapping of frame fm into table fm & 1: entry e = (layer * K3 + k) * 2 + xy
  auto fold_mapping = [&](int fm) {
    if constexpr (FOLD) {
      for (int e = threadIdx.x; e < L * K3 * 2; e += kThreads) {
        const int c = e & 1, r = (e >> 1) % K3, l = (e >> 1) / K3;
        const float* row = inv_kernel + r * K3;
        const float* x = src_pts + ((int64_t)fm * L + l) * (K3 - 3) * 2 + c;
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < K3 - 3; ++n) acc = fmaf(row[n], x[2 * n], acc);
        smap[(fm & 1) * (LP * K3 * 2) + e] = acc;
      }
    }
  };
  // The FIRST frame's table, ahead of the frame loop, where nothing else is live yet: a thread's entries (L * K3 * 2
  // <= 2 kBlock up to 12 layers) are loaded TOGETHER (clamped indices, no branch around the loads) and then summed --
  // as the loop above the second trip, which only the first few threads make, is a second exposed memory round
  // trip that the whole workgroup waits for at the barrier (1.3 us of a C2 tile-frame's 10).  Same fma order per
  // entry: same bits.
  auto fold_first = [&](int fm) {
    if constexpr (FOLD && LP <= 12) {
      constexpr int kTrips = (LP * K3 * 2 + kThreads - 1) / kThreads;
      const int n_ent = L * K3 * 2;
      float rv[kTrips][K3 - 3], xv[kTrips][K3 - 3];
#pragma unroll
      for (int q = 0; q < kTrips; ++q) {
        const int e = min((int)threadIdx.x + q * kThreads, n_ent - 1);
        const int c = e & 1, r = (e >> 1) % K3, l = (e >> 1) / K3;
        const float* row = inv_kernel + r * K3;
        const float* x = src_pts + ((int64_t)fm * L + l) * (K3 - 3) * 2 + c;
#pragma unroll
        for (int n = 0; n < K3 - 3; ++n) {
          rv[q][n] = row[n];
          xv[q][n] = x[2 * n];
        }
      }
#pragma unroll
      for (int q = 0; q < kTrips; ++q) {
        const int e = (int)threadIdx.x + q * kThreads;
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < K3 - 3; ++n) acc = fmaf(rv[q][n], xv[q][n], acc);
        if (e < n_ent) smap[(fm & 1) * (LP * K3 * 2) + e] = acc;
      }
    } else {
      fold_mapping(fm);
    }
  };
  // zero-weight taps of wild (NaN) coordinates may read any word of the image: keep it finite
  static_assert(kMain % 4 == 0, "cleared sixteen bytes at a time");
  for (int i = threadIdx.x; i < kMain / 4; i += kThreads) reinterpret_cast<f32x4*>(lds)[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  // MFMA A operand, v_mfma_f32_16x16x4_f32: A[row = lane & 15][k = lane >> 4]; row = pixel column
  // arow of tile row g of this wave, k = 4 * ks + kk.  Kept across the frames of the chunk.
  float av[4][KS];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t pa = (uint32_t)(min(row0 + g, H - 1) * W + min(col0 + arow, W - 1));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk;
      // 32-bit byte offset from the uniform base: K3 * HW * 4 < 2^32 is checked by the launcher
      const float bs = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + pa) * 4u);
      av[g][ks] = (k < K3) ? bs : 0.0f;
    }
  }
  const int f0 = chunk * frames_per_block;
  const int f1 = min(F, f0 + frames_per_block);
  if (f0 < f1) fold_first(f0);
  __syncthreads();
  WALDO_FSTAMP(1);  // LDS cleared, basis operand loaded, first mapping folded

  // pixel-unit grid: the MFMA column of this lane is an x column (even) or a y column (odd)
  const float half_size = 0.5f * (float)((arow & 1) ? H : W);
  const float half_size_m1 = 0.5f * (float)(((arow & 1) ? H : W) - 1);
  const int lane_k = lane, arow_k = arow, kk_k = kk;
  for (int f = f0; f < f1; ++f) {
    // The per-thread indices are re-materialised every frame: left visible as loop invariants, hipcc 7.2
    // computes every LDS / global address derived from them once per kernel (dozens of VGPRs of
    // base + constant that it then keeps live across the whole frame loop, or spills).
    int lane = lane_k, arow = arow_k, kk = kk_k;
    asm volatile("" : "+v"(lane), "+v"(arow), "+v"(kk));
    if (f == f0 + 1) WALDO_FSTAMP(8);  // (stamps 8 .. 12: the SECOND frame of the chunk, the loop's steady state)
    // ---- (A) TPS grid of every layer on the matrix pipe, in pixel units (scaled_map):
    // D[pixel][(layer, xy)] = sum_k basis[pixel][k] * mapping[k][(layer, xy)]
    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[g][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const float* mp = FOLD ? smap + (f & 1) * (LP * K3 * 2) : mapping + (int64_t)f * L * K3 * 2;
    {
      // every B-operand load of the frame is issued before the first use: left alone hipcc sinks the
      // last k-step's load behind its (k < K3) predicate -- a second dependent round trip at the head
      // of every frame (the empty asm pins the loads; they return in order anyway)
      float mraw[KS][NT];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int k = 4 * ks + kk, col = nt * 16 + arow, l = col >> 1;
          mraw[ks][nt] = mp[(min(l, L - 1) * K3 + min(k, K3 - 1)) * 2 + (col & 1)];
        }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(mraw[ks][nt]));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + kk;
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = nt * 16 + arow, l = col >> 1;
          bv[nt] = (k < K3 && l < L) ? scaled_map(mraw[ks][nt], k == K3 - 3, half_size, half_size_m1) : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][ks], bv[nt], acc[g][nt], 0, 0, 0);
      }
    }
    // the next frame's table (the other one: its last reader was the previous frame's phase (A),
    // barriers ago; its first reader comes after this frame's closing barrier)
    if (f + 1 < f1) fold_mapping(f + 1);
    // ---- (B) range of every grid coordinate over the workgroup's pixels.  In the accumulator
    // layout a lane holds 16 pixels of ONE column (layer, xy): 30 min/max + two cross-row steps
    // per 16 columns, instead of a cross-lane reduction per layer and coordinate.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float mn = acc[0][nt][0], mx = mn;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          mn = fminf(mn, acc[g][nt][r]);
          mx = fmaxf(mx, acc[g][nt][r]);
        }
      mn = rows_min(mn);
      mx = rows_max(mx);
      if (kk == 0) {
        boxred[(wave * GGC + nt * 16 + arow) * 2 + 0] = mn;
        boxred[(wave * GGC + nt * 16 + arow) * 2 + 1] = mx;
      }
    }
    // ---- (C) accumulators -> one pixel per lane, through this wave's slice of LDS:
    // D[row = (lane >> 4) * 4 + r][col = lane & 15] of tile row g -> T[16 g + row][col]
    float gx[LP], gy[LP];
    {
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
        gx[l] = T[lane * TP + 2 * l];
        gy[l] = T[lane * TP + 2 * l + 1];
      }
    }
    __syncthreads();  // ranges of all waves visible; the slices (inside the image) are free again
    if (f == f0) WALDO_FSTAMP(2);  // grid on MFMA, ranges, transposition
    if (f == f0 + 1) WALDO_FSTAMP(9);
    // ---- (D) box of the 2x2 blocks of every layer: lanes 0..15 of every wave turn the range of
    // "their" column into block origins, then the corners go to SGPRs
    int lo_t[NT], hi_t[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float mn = boxred[(nt * 16 + arow) * 2 + 0], mx = boxred[(nt * 16 + arow) * 2 + 1];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        mn = fminf(mn, boxred[(w * GGC + nt * 16 + arow) * 2 + 0]);
        mx = fmaxf(mx, boxred[(w * GGC + nt * 16 + arow) * 2 + 1]);
      }
      const int size = (arow & 1) ? H : W;
      lo_t[nt] = block_origin(mn, size);
      hi_t[nt] = block_origin(mx, size) + 1;
    }
    int bx0[LP], by0[LP], bw[LP], bh[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      const int nt = (2 * l) / 16, ln = (2 * l) % 16;
      const int xmin = __builtin_amdgcn_readlane(lo_t[nt], ln), xmax = __builtin_amdgcn_readlane(hi_t[nt], ln);
      const int ymin = __builtin_amdgcn_readlane(lo_t[nt], ln + 1), ymax = __builtin_amdgcn_readlane(hi_t[nt], ln + 1);
      bx0[l] = xmin & ~3;
      by0[l] = ymin;
      bw[l] = ((xmax - bx0[l] + 1) + 3) & ~3;
      bh[l] = ymax - ymin + 1;
#ifdef WALDO_ABL_NOFALLBACK  // timing-only ablation: oversize boxes are cut to the cap (wrong values)
      bw[l] = min(bw[l], 128);
      bh[l] = min(bh[l], kCap / bw[l]);
#endif
    }

    // ---- (E) staging: every lane moves one item (two texels, four planes) of a layer's box.  A
    // rolling window of kAhead layers is in flight (the load of layer l + kAhead is issued when
    // layer l leaves its registers for LDS): memory latency is exposed once per frame; then each
    // layer goes registers -> LDS -> taps; the image is double-buffered, one barrier per layer.
    constexpr int kAhead = LP < WALDO_STAGE_AHEAD ? LP : WALDO_STAGE_AHEAD;
    static_assert(kCap / 2 == kThreads, "one box item per lane");
    int item_l = threadIdx.x;
    asm volatile("" : "+v"(item_l));
    float s[LP][4];
    StageRegs stg[LP];  // fully unrolled: a layer's registers live from its load to its LDS store
    auto issue = [&](int l) {
      const int lc = EXL ? l : min(l, L - 1);
      const float* src = layers + ((int64_t)WALDO_LAYER_FRAME(f) * L + lc) * 4 * HW;
      // unconditional loads (items past the box re-read its last item; a box that does not fit
      // reads texel 0): no exec-mask branches, so the loads are issued back to back
      const bool fits = bh[l] * bw[l] <= kCap;
      const int bw2 = bw[l] >> 1, n = fits ? bh[l] * bw2 : 1;
      const int ox = fits ? __mul24(by0[l], W) + bx0[l] : 0;
      // item, bw2 < 2^9 and the +0.5: the approximate reciprocal (1 ulp) gives the exact quotient
      const float rcp = __builtin_amdgcn_rcpf((float)bw2);
      const int item = min(item_l, n - 1);
      const int r = (int)(((float)item + 0.5f) * rcp);
      const int xh = item - __mul24(r, bw2);
      const unsigned off = (unsigned)(ox + __mul24(r, W) + 2 * xh) * 4u;  // bytes; HW * 4 < 2^32 (launcher)
      stg[l].c0 = ld8(src, off);
      stg[l].c1 = ld8(src + HW, off);
      stg[l].c2 = ld8(src + 2 * HW, off);
      stg[l].c3 = ld8(src + 3 * HW, off);
    };
    WALDO_PRIO_ON(WALDO_FWD_PRIO_MASK, 2);
#pragma unroll
    for (int l = 0; l < kAhead; ++l) issue(l);
    WALDO_PRIO_OFF(WALDO_FWD_PRIO_MASK, 2);
    if (f == f0) WALDO_FSTAMP(3);  // boxes, first loads issued
    if (f == f0 + 1) WALDO_FSTAMP(10);
    {
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        if (!EXL && l >= L) {  // padding layer: inert
          s[l][0] = s[l][1] = s[l][2] = 0.0f;
          s[l][3] = -1.0f;
          continue;
        }
        const bool fits = bh[l] * bw[l] <= kCap;  // block-uniform
        WALDO_PRIO_ON(WALDO_FWD_PRIO_MASK, 1);
        if (fits) {
          const int n = bh[l] * (bw[l] >> 1);
          if (item_l < n)  // row-major with pitch bw: item = r * bw2 + xh, texel 2 * item
            stage_store(img + (l & 1) * kBuf, item_l, stg[l]);
        }
        if (l + kAhead < LP) issue(l + kAhead);
        WALDO_PRIO_OFF(WALDO_FWD_PRIO_MASK, 1);
        __syncthreads();  // buffer l&1 complete; buffer (l+1)&1 no longer read by anyone
        if (fits) {
          const TapCore tc = tap_core_px(gx[l], gy[l], H, W);
          const float* b0 = img + (l & 1) * kBuf;
          f32x2_t sv[2];
          if (__ballot(!tap_interior(tc, H, W)) == 0ull) {
            // wave-uniform: all corners inside the layer, every validity factor is exactly 1
            const int idx = __mul24(tc.y0 - by0[l], bw[l]) + (tc.x0 - bx0[l]);  // inside the box
            const PairBlock pb = read_block(b0, idx, bw[l]);
#pragma unroll
            for (int q = 0; q < 2; ++q) sv[q] = lerp2(pb.p00[q], pb.p01[q], pb.p10[q], pb.p11[q], tc.fx, tc.fy);
          } else {
            const BoxTaps t = make_box_taps(tc, H, W);
            // inside the box by construction; the clamp only matters for NaN coordinates
            const int idx = min(max(__mul24(t.yb - by0[l], bw[l]) + (t.xb - bx0[l]), 0), kCap - bw[l] - 2);
            const PairBlock pb = assign_corners(read_block(b0, idx, bw[l]), t.cs, t.rs);
            // delta padding (lvd.py:548,559): shift the corner values before their validity
            const f32x2_t d2 = {delta, delta};
#pragma unroll
            for (int q = 0; q < 2; ++q)
              sv[q] = lerp2((pb.p00[q] + d2) * t.v00, (pb.p01[q] + d2) * t.v01, (pb.p10[q] + d2) * t.v10,
                            (pb.p11[q] + d2) * t.v11, t.fx, t.fy) - d2;
          }
          s[l][0] = sv[0][0];
          s[l][1] = sv[0][1];
          s[l][2] = sv[1][0];
          s[l][3] = sv[1][1];
        } else {  // box larger than the LDS image (violent warp): gather straight from memory
          const Taps t = make_taps_px(gx[l], gy[l], H, W);
          const float* base = layers + ((int64_t)f * L + l) * 4 * HW;
#pragma unroll
          for (int c = 0; c < 4; ++c) s[l][c] = tap_sample(base + c * HW, t, delta);
        }
      }
    }

    if (f == f0) WALDO_FSTAMP(5);  // every layer sampled
    if (f == f0 + 1) WALDO_FSTAMP(11);
    // ---- composite: a_0 = 1 (lvd.py:105), a_l = (s_l3 + 1) / 2
    float a[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) a[l] = (s[l][3] + 1.0f) * 0.5f;
    a[0] = 1.0f;
    const float* oc = occ + (int64_t)f * L * L;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    // two layers j per step: the occlusion products run on the packed-fp32 pipe (v_pk_mul_f32 /
    // v_pk_add_f32, two lanes' worth of work per VALU issue slot -- this kernel is issue bound)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float apv[LP];
#pragma unroll
    for (int j = 0; j < LP; j += 2) {
      const int j1 = j + 1 < LP ? j + 1 : j;
      const int jc0 = EXL ? j : min(j, L - 1), jc1 = EXL ? j1 : min(j1, L - 1);
      f32x2 pr = {1.0f, 1.0f};
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int ic = EXL ? i : min(i, L - 1);
        const f32x2 o = {oc[ic * L + jc0], oc[ic * L + jc1]};
        const f32x2 av = {a[i], a[i]};
        pr = pr * ((f32x2){1.0f, 1.0f} - av * o);
      }
      apv[j] = a[j] * pr[0];
      if (j + 1 < LP) apv[j + 1] = a[j + 1] * pr[1];
    }
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const float ap = apv[j];
      r = fmaf(ap, (s[j][0] + 1.0f) * 0.5f, r);
      g = fmaf(ap, (s[j][1] + 1.0f) * 0.5f, g);
      b = fmaf(ap, (s[j][2] + 1.0f) * 0.5f, b);
      if (alpha_out != nullptr && pm.live && (EXL || j < L))
        alpha_out[((int64_t)f * L + j) * HW + p] = 2.0f * ap - 1.0f;
    }
    WALDO_PRIO_ON(WALDO_FWD_PRIO_MASK, 4);
    if (pm.live) {
      float* o = rgb + (int64_t)f * 3 * HW + p;
      o[0] = 2.0f * r - 1.0f;
      o[HW] = 2.0f * g - 1.0f;
      o[2 * HW] = 2.0f * b - 1.0f;
    }
    WALDO_PRIO_OFF(WALDO_FWD_PRIO_MASK, 4);
    __syncthreads();  // boxred and the image buffers are re-used by the next frame
    if (f == f0) WALDO_FSTAMP(6);  // composite, stores issued, closing barrier
    if (f == f0 + 1) WALDO_FSTAMP(12);
  }
}

}  // namespace waldo
