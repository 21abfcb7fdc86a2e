// K1 of the two-kernel backward (K3 == 19, 4 | W): pixel-major on 16 x 16-pixel tiles, frames
// looped inside the workgroup.
//
// Restates, for one (frame, tile), the backward of TPSWarp -> F.grid_sample -> reduce_comp
// (models/modules/warp.py:57-64, models/nets/lvd.py:100-114 and the autograd the reference gets
// for them) and leaves, for the splat kernel (K2, warp_composite_splat.hip):
//   * one 16-byte record (grid x, grid y in pixel units, a'_l, d loss / d s_l3) per (frame, layer,
//     pixel);
//   * the footprint table: per 8 x 16-pixel CELL the bounding box of the source texels its
//     bilinear footprints touch and the exponents of an upper bound of its contribution
//     magnitudes (colour planes and alpha plane separately);
//   * the control-point gradient of the tile (f32 MFMA contraction basis^T x grid-grad) as a
//     per-tile partial that a small reduce kernel sums in a fixed order.
// Data movement:
//   * a workgroup keeps ONE tile and walks over a chunk of frames: the tile's 256 x 19 TPS basis
//     values are loaded once, into registers in the MFMA A-operand order of the grid phase (rows =
//     pixels, contraction over k); the control-point contraction wants them transposed (rows = k,
//     contraction over pixels) and gets them through LDS from those same registers -- before, each
//     (frame, tile) read them twice from memory (2.2 GB per launch at the headline shape);
//   * the samples and their derivatives come out of an LDS image of each layer's footprint box,
//     staged as in warp_composite_fwd_lds_kernel (float4 texels); a layer whose box does not fit
//     the image (violent warp) is gathered from memory instead;
//   * the tile OWNS its cells of the footprint table: plain stores, no atomics, no memset.
// The 4 L tap derivatives of a pixel are parked in LDS between sampling and the composite backward
// (L <= 8; in registers above, where the LDS rows would leave one workgroup per CU).
#pragma once
// included at the end of warp_composite_kernels.hip.h, after warp_composite_fwd_lds.hip.h

namespace waldo {

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

// biased exponent of a non-negative float (255: infinity or NaN)
__device__ __forceinline__ int exponent_of(float v) { return (int)(__float_as_uint(v) >> 23) & 0xff; }

#ifndef WALDO_K1_STAGE_AHEAD
#define WALDO_K1_STAGE_AHEAD 3  // layers whose box loads are in flight (2 / 3: backward 2.19 / 2.16 ms; the
                                // forward, with fewer registers to spare at four waves per SIMD, keeps 2)
#endif

template <int LP>
struct Px16Cfg {
  static constexpr int K3 = kGmapK3, KS = (K3 + 3) / 4;
  static constexpr int NC = 2 * LP, NT = (NC + 15) / 16, GGC = NT * 16, TP = GGC + 1;
  // tap derivatives parked in LDS (L <= 8) or kept in registers / scratch: above 8 layers the LDS
  // rows would leave ONE workgroup per CU, and at one wave per SIMD hipcc 7.2 allocates AGPRs as
  // extra registers and reads MFMA results back wrong through them (see warp_composite_fwd_lds.hip.h)
  static constexpr bool kPark = LP <= 8;
  static constexpr int PP1 = kBlock + 1;         // pitch of the per-pixel columns (park rows; sizing only)
  // Operands of phase (H) in LDS, per wave: [row][kk = pixel & 3][s = pixel >> 2] with a row pitch of 68
  // floats -- the sixteen values a lane feeds to the sixteen MFMA steps are CONTIGUOUS (four ds_read_b128
  // instead of sixteen ds_read_b32; 68 = 64 + 4: the 16 lanes of a read hit 64 different banks)
  static constexpr int OP = 68;                  // row pitch of an operand table
  static constexpr int kBtFloats = 20 * OP;      // one wave's basis table: rows k = 0 .. 19
  static constexpr int kImgFloats = 2 * kImgBufFloats;           // two buffers of float4 texels
  static constexpr int kTFloats = 4 * kWave * TP;                // transposition slices of the grid
  static constexpr int kGgFloats = 4 * GGC * OP;                 // grid gradients (MFMA B operand), one table per wave
  static constexpr int kAccFloats = 4 * 2 * NT * 256;            // per-wave MFMA accumulators
  static constexpr int kParkRows = 4 * LP > GGC ? 4 * LP : GGC;
  static constexpr int kPark0 = kParkRows * PP1 > kAccFloats ? kParkRows * PP1 : kAccFloats;
  static constexpr int kParkFloats = kPark ? (kPark0 + 3) / 4 * 4 : 0;
  // parked: the four basis slices of phase (H) go into the stage region -- three of them when the
  // fourth fits behind the gg rows inside the park region; in registers: gg, then the four slices
  static constexpr bool kBtInPark = kPark && kParkFloats >= kGgFloats + kBtFloats;
  static constexpr int kBtBase = kPark ? 0 : kGgFloats;
  static constexpr int kBtEnd = kBtBase + (kBtInPark ? 3 : 4) * kBtFloats;
  static constexpr int kStage0 = kImgFloats > kTFloats ? kImgFloats : kTFloats;
  static constexpr int kStage1 = kBtEnd > kStage0 ? kBtEnd : kStage0;
  static constexpr int kStage2 = (!kPark && kAccFloats > kStage1) ? kAccFloats : kStage1;
  static constexpr int kStageFloats = (kStage2 + 3) / 4 * 4;
  static constexpr int kLdsFloats = kParkFloats + kStageFloats + 4 * GGC * 2 + 4 * LP;
  // registers: L <= 8 fits three waves per SIMD (168 VGPRs); the grad_occ variant and larger L get 256
  static constexpr int kWavesPerSimd = LP <= 8 ? 3 : 2;
};

// Diagnostic build only (-DWALDO_K1_STAMPS): s_memtime stamps of wave 0 at the phase boundaries of
// the LAST frame of a workgroup's chunk, into a buffer of the code object that no kernel reads
// (tools_dev/k1_stamps.py); never compiled into the product library.
#ifdef WALDO_K1_STAMPS
constexpr int kStampSlots = 24, kStampBlocks = 8192;
__device__ unsigned long long waldo_k1_stamps[kStampBlocks * kStampSlots];
#define WALDO_STAMP(i)                                                                            \
  do {                                                                                            \
    if (threadIdx.x == 0 && f == f1 - 1 && blockIdx.x < kStampBlocks)                             \
      waldo_k1_stamps[blockIdx.x * kStampSlots + (i)] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#else
#define WALDO_STAMP(i) do { } while (0)
#endif

template <int LP, bool EXL, bool GOCC>
__global__ __launch_bounds__(kBlock, (GOCC ? 2 : Px16Cfg<LP>::kWavesPerSimd)) void warp_composite_bwd_px16_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ occ,
    const float* __restrict__ grad_rgb, const float* __restrict__ grad_alpha,
    float4* __restrict__ rec, int* __restrict__ cellbox,
    unsigned* __restrict__ cellbound, float* __restrict__ gmap_partial, float* __restrict__ grad_occ,
    int F, int Lrt, int H, int W, int frames_per_block, int ntx, int ntiles, int nchunks, int nbands,
    int ncx, int ncells, float delta) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  using C = Px16Cfg<LP>;
  constexpr int K3 = C::K3, KS = C::KS, NT = C::NT, GGC = C::GGC, TP = C::TP, OP = C::OP;
  constexpr bool kPark = C::kPark;
  const int L = EXL ? LP : Lrt;
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int arow = lane & 15, kk = lane >> 4;
  int chunk, tile, rest_;  // (frame chunk, band of tiles) pinned to an XCD
  if (!xcd_decode_banded(blockIdx.x, nchunks, nbands, ntiles, 1, chunk, tile, rest_)) return;

  // LDS:
  //   park   (parked variant) rows 0 .. 4*LP-1, one column per pixel: the parked tap derivatives;
  //          then rows 0 .. GGC-1 the grid gradients gg (MFMA B operand) / the MFMA accumulators;
  //   stage  per frame, in turn: the transposition slices of the grid, the staged layer image (two
  //          buffers), the transposed basis of phase (H) -- and, without parking, gg in front of it;
  //   boxred coordinate ranges per wave; wbound contribution-bound exponents per wave.
#ifndef WALDO_ABL_K1_LDS_PAD  // timing-only ablation: extra LDS floats per workgroup (occupancy sweep)
#define WALDO_ABL_K1_LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) float lds[C::kLdsFloats + WALDO_ABL_K1_LDS_PAD];
  float* park = lds;
  float* img = lds + C::kParkFloats;
  float* gg = kPark ? park : img;
  float* boxred = img + C::kStageFloats;
  int* wbound = reinterpret_cast<int*>(boxred + 4 * GGC * 2);  // [wave][layer]: e_rgb | e_alpha << 16
  const int pix = threadIdx.x;

  const int col0 = (tile % ntx) * kLdsTile, row0 = (tile / ntx) * kLdsTile + wave * 4;
  // 16 x 16 tile: wave w covers rows 4w .. 4w+3, lane -> (row 4w + lane / 16, column lane % 16)
  const bool live = col0 + arow < W && row0 + kk < H;
  const float livef = live ? 1.0f : 0.0f;
  const int64_t p = (int64_t)min(row0 + kk, H - 1) * W + min(col0 + arow, W - 1);
  // zero-weight taps of wild (NaN) coordinates may read any word of the image: keep it finite
  for (int i = threadIdx.x; i < C::kStageFloats; i += kBlock) img[i] = 0.0f;

  // MFMA A operand of the grid, v_mfma_f32_16x16x4_f32: A[row = lane & 15][k = lane >> 4]; row =
  // pixel column arow of tile row g of this wave, k = 4 * ks + kk.  Kept across the frames of the
  // chunk (4 rows x 64 contiguous bytes per load instruction).
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
  __syncthreads();

  // pixel-unit grid: the MFMA column of this lane is an x column (even) or a y column (odd)
  const float half_size = 0.5f * (float)((arow & 1) ? H : W);
  const float half_size_m1 = 0.5f * (float)(((arow & 1) ? H : W) - 1);
  const int f0 = chunk * frames_per_block;
  const int f1 = min(F, f0 + frames_per_block);
  const int lane_k = lane, arow_k = arow, kk_k = kk, pix_k = pix;
  for (int f = f0; f < f1; ++f) {
    // The per-thread indices are re-materialised every frame.  Left visible as loop invariants, hipcc 7.2
    // computes every LDS / global address derived from them ONCE per kernel -- dozens of VGPRs of
    // base + constant -- and spills what does not fit; each reload then sits behind an s_waitcnt vmcnt(0),
    // i.e. behind the frame's record stores (vector-memory operations retire in issue order).
    int lane = lane_k, arow = arow_k, kk = kk_k, pix = pix_k;
    asm volatile("" : "+v"(lane), "+v"(arow), "+v"(kk), "+v"(pix));
    const float* oc = occ + (int64_t)f * L * L;
    WALDO_STAMP(0);
    // ---- (A) TPS grid of every layer on the matrix pipe, in pixel units (see
    // warp_composite_fwd_lds_kernel):
    // D[pixel][(layer, xy)] = sum_k basis[pixel][k] * mapping[k][(layer, xy)]
    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[g][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const float* mp = mapping + (int64_t)f * L * K3 * 2;
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
    // ---- (B) range of every grid coordinate over the wave's pixels
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
    // ---- (C) accumulators -> one pixel per lane, through this wave's slice of LDS
    float gxs[LP], gys[LP];
    {
      float* T = img + wave * (kWave * TP);
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
    __syncthreads();  // ranges of all waves visible; the slices (inside the image) are free again
    WALDO_STAMP(1);
    // ---- (D) box of the 2x2 blocks of every layer: lanes 0..15 turn the range of "their" column
    // (layer, xy) into block origins -- per cell of the footprint table (this tile owns its cells:
    // plain stores in the splat kernel's format (min x, -max x, min y, -max y)) and for the whole
    // tile (the box that is staged; corners to SGPRs)
    int bx0[LP], by0[LP], bw[LP], bh[LP];
    bool fits[LP];
    {
      const int size = (arow & 1) ? H : W;
      constexpr int kCellsPerTile = kLdsTile / kCellRows, kWavesPerCell = 4 / kCellsPerTile;
      static_assert(kCellsPerTile * kCellRows == kLdsTile && kWavesPerCell * kCellsPerTile == 4, "cell rows: 4, 8 or 16");
      int lo_t[NT], hi_t[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        lo_t[nt] = 0x7fffffff;
        hi_t[nt] = -1;
        const int l2 = (nt * 16 + arow) >> 1;
#pragma unroll
        for (int c = 0; c < kCellsPerTile; ++c) {
          float mn = boxred[((kWavesPerCell * c) * GGC + nt * 16 + arow) * 2 + 0];
          float mx = boxred[((kWavesPerCell * c) * GGC + nt * 16 + arow) * 2 + 1];
#pragma unroll
          for (int w = 1; w < kWavesPerCell; ++w) {
            mn = fminf(mn, boxred[((kWavesPerCell * c + w) * GGC + nt * 16 + arow) * 2 + 0]);
            mx = fmaxf(mx, boxred[((kWavesPerCell * c + w) * GGC + nt * 16 + arow) * 2 + 1]);
          }
          const int lo_c = block_origin(mn, size), hi_c = block_origin(mx, size) + 1;
          lo_t[nt] = min(lo_t[nt], lo_c);
          hi_t[nt] = max(hi_t[nt], hi_c);
          const int crow = (tile / ntx) * kCellsPerTile + c;
          if (wave == 0 && kk == 0 && l2 < L && crow * ncx < ncells) {
            int* bb = cellbox + (((int64_t)f * L + l2) * ncells + crow * ncx + (tile % ntx)) * 4 + (arow & 1) * 2;
            *reinterpret_cast<int2*>(bb) = make_int2(lo_c, -hi_c);
          }
        }
      }
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
        bh[l] = min(bh[l], kStageCap / bw[l]);
#endif
        fits[l] = bh[l] * bw[l] <= kStageCap;  // block-uniform
      }
    }

    WALDO_PRIO_ON(WALDO_K1_PRIO_MASK, 2);
    const float g0 = grad_rgb[(int64_t)f * 3 * HW + p] * livef;
    const float g1 = grad_rgb[(int64_t)f * 3 * HW + HW + p] * livef;
    const float g2 = grad_rgb[(int64_t)f * 3 * HW + 2 * HW + p] * livef;
    // |g0| + |g1| + |g2| >= max_c |g_c| and, unlike a max, keeps a NaN / infinity visible
    const float gsum = fabsf(g0) + fabsf(g1) + fabsf(g2);
    float a[LP], G[LP];
    float dxr[kPark ? 1 : LP], dxa[kPark ? 1 : LP], dyr[kPark ? 1 : LP], dya[kPark ? 1 : LP];

    // ---- (E) staged sampling with derivatives.  Every lane moves one item (two texels, four
    // planes) of a layer's box into the float4-texel LDS image (see warp_composite_fwd_lds_kernel);
    // a rolling window of kAhead layers is in flight (the load of layer l + kAhead is issued when
    // layer l leaves its registers for LDS); the image is double-buffered, one barrier per layer.
    constexpr int kAhead = LP < WALDO_K1_STAGE_AHEAD ? LP : WALDO_K1_STAGE_AHEAD;
    int item_l = threadIdx.x;
    asm volatile("" : "+v"(item_l));
    StageRegs stg[LP];  // fully unrolled: a layer's registers live from its load to its LDS store
    auto issue = [&](int l) {
      const int lc = EXL ? l : min(l, L - 1);
      const float* src = layers + ((int64_t)WALDO_LAYER_FRAME(f) * L + lc) * 4 * HW;
      // unconditional loads (items past the box re-read its last item; a box that does not fit
      // reads texel 0): no exec-mask branches, so the loads are issued back to back
      const int bw2 = bw[l] >> 1, n = fits[l] ? bh[l] * bw2 : 1;
      const int ox = fits[l] ? __mul24(by0[l], W) + bx0[l] : 0;
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
#pragma unroll
    for (int l = 0; l < kAhead; ++l) issue(l);
    WALDO_PRIO_OFF(WALDO_K1_PRIO_MASK, 2);
    WALDO_STAMP(2);
    {
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        if (!EXL && l >= L) {  // padding layer: inert
          a[l] = 0.0f;
          G[l] = 0.0f;
          if constexpr (kPark) {
            reinterpret_cast<f32x4*>(park)[l * kBlock + pix] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          } else {
            dxr[l] = dxa[l] = dyr[l] = dya[l] = 0.0f;
          }
          continue;
        }
        WALDO_PRIO_ON(WALDO_K1_PRIO_MASK, 1);
        if (fits[l]) {
          const int n = bh[l] * (bw[l] >> 1);
          if (item_l < n)  // row-major with pitch bw: item = r * bw2 + xh, texel 2 * item
            stage_store(img + (l & 1) * kImgBufFloats, item_l, stg[l]);
        }
        if (l + kAhead < LP) issue(l + kAhead);
        WALDO_PRIO_OFF(WALDO_K1_PRIO_MASK, 1);
        // the incoming alpha gradient of this layer travels with the staging loads (loaded where it
        // is used, the wave would wait for it with vmcnt(0) -- and with it for every box load in
        // flight)
        float gal = 0.0f;
        if (grad_alpha != nullptr) gal = grad_alpha[((int64_t)f * L + l) * HW + p];
        __syncthreads();  // buffer l&1 complete; buffer (l+1)&1 no longer read by anyone
        if (l < 8) WALDO_STAMP(3 + l);
        const TapCore tc = tap_core_px(gxs[l], gys[l], H, W);
        const float* b0 = img + (l & 1) * kImgBufFloats;
        float sv[4], sx[4], sy[4];
        if (!fits[l]) {  // box larger than the LDS image (violent warp): gather straight from memory
          const Taps tg = finish_taps(tc, H, W);
          const float* base = layers + ((int64_t)f * L + l) * 4 * HW;
#pragma unroll
          for (int c = 0; c < 4; ++c) sv[c] = tap_sample_d(base + c * HW, tg, sx[c], sy[c], delta);
        } else {
          PairBlock pb;
          float fx, fy, shift = 0.0f;
          if (__ballot(!tap_interior(tc, H, W)) == 0ull) {
            // wave-uniform: all corners inside the layer, every validity factor is exactly 1
            const int idx = __mul24(tc.y0 - by0[l], bw[l]) + (tc.x0 - bx0[l]);  // inside the box
            pb = read_block(b0, idx, bw[l]);
            fx = tc.fx;
            fy = tc.fy;
          } else {
            const BoxTaps t = make_box_taps(tc, H, W);
            // inside the box by construction; the clamp only matters for NaN coordinates
            const int idx = min(max(__mul24(t.yb - by0[l], bw[l]) + (t.xb - bx0[l]), 0), kStageCap - bw[l] - 2);
            pb = assign_corners(read_block(b0, idx, bw[l]), t.cs, t.rs);
            // delta padding: corner values shifted before their validity (the interior path needs
            // no shift: with every corner valid the weights sum to 1 and the shift cancels)
            const f32x2_t d2 = {delta, delta};
            shift = delta;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              pb.p00[q] = (pb.p00[q] + d2) * t.v00;
              pb.p01[q] = (pb.p01[q] + d2) * t.v01;
              pb.p10[q] = (pb.p10[q] + d2) * t.v10;
              pb.p11[q] = (pb.p11[q] + d2) * t.v11;
            }
            fx = t.fx;
            fy = t.fy;
          }
          // value and derivatives w.r.t. the pixel coordinates, two channels per instruction
          const f32x2_t fx2 = {fx, fx}, fy2 = {fy, fy};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x2_t d0 = pb.p01[q] - pb.p00[q], d1 = pb.p11[q] - pb.p10[q];
            const f32x2_t top = __builtin_elementwise_fma(fx2, d0, pb.p00[q]);
            const f32x2_t bot = __builtin_elementwise_fma(fx2, d1, pb.p10[q]);
            const f32x2_t dy = bot - top;
            const f32x2_t dx = __builtin_elementwise_fma(fy2, d1 - d0, d0);
            const f32x2_t v = __builtin_elementwise_fma(fy2, dy, top);
            sv[2 * q] = v[0] - shift;
            sv[2 * q + 1] = v[1] - shift;
            sx[2 * q] = dx[0];
            sx[2 * q + 1] = dx[1];
            sy[2 * q] = dy[0];
            sy[2 * q + 1] = dy[1];
          }
        }
        a[l] = (sv[3] + 1.0f) * 0.5f;
        G[l] = g0 * (sv[0] + 1.0f) + g1 * (sv[1] + 1.0f) + g2 * (sv[2] + 1.0f);
        const float vxr = fmaf(g2, sx[2], fmaf(g1, sx[1], g0 * sx[0]));
        const float vyr = fmaf(g2, sy[2], fmaf(g1, sy[1], g0 * sy[0]));
        if constexpr (kPark) {
          // one 16-byte texel per (layer, pixel): ONE ds_write_b128 here and one ds_read_b128 in phase (G)
          // instead of four 4-byte accesses each way (8 of K1's 26 LDS instructions per pixel-layer)
          reinterpret_cast<f32x4*>(park)[l * kBlock + pix] = (f32x4){vxr, sx[3], vyr, sy[3]};
        } else {
          dxr[l] = vxr;
          dxa[l] = sx[3];
          dyr[l] = vyr;
          dya[l] = sy[3];
        }
        G[l] = fmaf(2.0f * livef, gal, G[l]);
      }
    }

    WALDO_STAMP(11);
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
        const bool has1 = j + 1 < LP;  // odd LP: the last pair is a single layer
        const int j1 = has1 ? j + 1 : j;
        const int jc0 = EXL ? j : min(j, L - 1), jc1 = EXL ? j1 : min(j1, L - 1);
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
        ga[j] = G[j] * pre[0];  // G[j] = d loss / d a'_j
        if (has1) {
          ap[j + 1] = a[j + 1] * pre[1];
          ga[j + 1] = G[j + 1] * pre[1];
        }
        const f32x2 gaj = (f32x2){G[j] * a[j], has1 ? G[j1] * a[j1] : 0.0f};
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
          if (has1 && (EXL || j + 1 < L)) {
            const float redv = wave_transpose_reduce<LP>(gocc1, lane);
            if (m < L) atomicAdd(grad_occ + (int64_t)f * L * L + m * L + j + 1, redv);
          }
        }
      }
#pragma unroll
      for (int l = 0; l < LP; ++l) ga[l] += ga2[l][0] + ga2[l][1];
    }
    WALDO_STAMP(12);
    // ---- (G) a_m = (s_m3 + 1)/2 for m >= 1; a_0 is the constant 1.  Records, contribution bounds
    // and the grid gradient of every layer.  The records go out FIRST: the stores get the rest of
    // this phase to drain.
    WALDO_PRIO_ON(WALDO_K1_PRIO_MASK, 4);
#ifndef WALDO_ABL_NO_REC_STORE  // timing-only ablation: K1 without its record stores (wrong gradients)
    if (live) {
#else
    if (live && H < 0) {
#endif
#pragma unroll
      for (int l = 0; l < LP; ++l)
        if (EXL || l < L)
          stream_store16<WALDO_REC_STORE_POLICY>(reinterpret_cast<float*>(rec + ((int64_t)WALDO_REC_FRAME(f) * L + l) * HW),
                                                 (uint32_t)p * 16u, HW * 16,
                                                 (f32x4){gxs[l], gys[l], ap[l], l >= 1 ? 0.5f * ga[l] : 0.0f});
    }
    WALDO_PRIO_OFF(WALDO_K1_PRIO_MASK, 4);
    if constexpr (!kPark) lds_barrier();  // every wave is done sampling: gg overlays the staged image
    {
      float ggx[LP], ggy[LP];
      int eb[LP];
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        const bool pad = !EXL && l >= L;
        const float gsa = (l >= 1 && !pad) ? 0.5f * ga[l] : 0.0f;
        float vxr, vxa, vyr, vya;
        if constexpr (kPark) {
          const f32x4 pk = reinterpret_cast<const f32x4*>(park)[l * kBlock + pix];
          vxr = pk[0];
          vxa = pk[1];
          vyr = pk[2];
          vya = pk[3];
        } else {
          vxr = dxr[l];
          vxa = dxa[l];
          vyr = dyr[l];
          vya = dya[l];
        }
        ggx[l] = fmaf(gsa, vxa, ap[l] * vxr) * (0.5f * (float)W);
        ggy[l] = fmaf(gsa, vya, ap[l] * vyr) * (0.5f * (float)H);
        // |tap contribution| to a colour plane <= |a'_l| sum_c |g_c| < 2^(e - 126), e the biased
        // exponent of that product; to the alpha plane <= |g_alpha| likewise: bilinear weights are
        // <= 1.  Only the exponents go into the table, one packed 16-bit max-reduction per layer;
        // 255 (infinity / NaN) survives the reduction and poisons the tiles the cell reaches.
        const float cb = (live && !pad) ? fabsf(ap[l]) * gsum : 0.0f;
        const float ca = (live && !pad) ? fabsf(gsa) : 0.0f;
        int v = exponent_of(cb) | (exponent_of(ca) << 16);
        v = pk_max_u16(v, row_ror_i<8>(v));
        v = pk_max_u16(v, row_ror_i<4>(v));
        v = pk_max_u16(v, row_ror_i<2>(v));
        v = pk_max_u16(v, row_ror_i<1>(v));
        eb[l] = (int)rows_combine_u((unsigned)v, [](unsigned x, unsigned y) { return (unsigned)pk_max_u16((int)x, (int)y); });
      }
      // every lane now holds every layer's bound: one masked store region for all of them
      if (lane == 0) {
#pragma unroll
        for (int l = 0; l < LP; ++l) wbound[wave * LP + l] = eb[l];
      }
      // this thread's column of gg.  With parking gg overlays the parked texels -- other threads' too (the
      // texels are [layer][pixel], gg is [column][pixel]): every thread has to be done reading them first
      if constexpr (kPark) lds_barrier();
      // (operand table of this wave, Px16Cfg::OP: column c of pixel `lane` at [c][lane & 3][lane >> 2])
      float* ggw = gg + wave * GGC * OP + (lane & 3) * 16 + (lane >> 2);
#pragma unroll
      for (int l = 0; l < LP; ++l) {
        ggw[(2 * l) * OP] = ggx[l];
        ggw[(2 * l + 1) * OP] = ggy[l];
      }
#pragma unroll
      for (int c = 2 * LP; c < GGC; ++c) ggw[c * OP] = 0.0f;
    }
    __syncthreads();  // gg rows and the waves' bounds are complete
    WALDO_STAMP(13);
    {
      constexpr int kCellsPerTile = kLdsTile / kCellRows, kWavesPerCell = 4 / kCellsPerTile;
      if (threadIdx.x < kCellsPerTile * LP) {
        const int c = threadIdx.x / LP, l = threadIdx.x % LP;
        const int crow = (tile / ntx) * kCellsPerTile + c;
        if (l < L && crow * ncx < ncells) {
          int v = wbound[(kWavesPerCell * c) * LP + l];
#pragma unroll
          for (int w = 1; w < kWavesPerCell; ++w) v = pk_max_u16(v, wbound[(kWavesPerCell * c + w) * LP + l]);
          cellbound[((int64_t)f * L + l) * ncells + crow * ncx + (tile % ntx)] =
              pack_bound_exponents(v & 0xffff, (v >> 16) & 0xffff);
        }
      }
    }

    // ---- (H) control-point gradient: basis^T x gg on the MFMA pipe
    // D[k][col] = sum_pix basis[k][pix] * gg[pix][col]; v_mfma_f32_16x16x4_f32: A[row=lane&15]
    // [kk=lane>>4], B[kk=lane>>4][col=lane&15], D[row=(lane>>4)*4+reg][col=lane&15].  Each wave
    // contracts the pixels it produced; the 4 wave results are summed through LDS in a fixed order
    // and stored as this tile's partial (no atomics, deterministic).
    if (gmap_partial != nullptr) {
      WALDO_PRIO_ON(WALDO_K1_PRIO_MASK, 8);
      // (H1) the A operand wants basis[k][pixel] with k along lane & 15: the wave writes the 64 x 20
      // values it holds in the grid phase's operand order into its LDS slice [pixel][k]
      float* Bw = (C::kBtInPark && wave == 3) ? park + C::kGgFloats : img + C::kBtBase + wave * C::kBtFloats;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)  // basis[k = 4 ks + kk] of the wave's pixel 16 g + arow
          Bw[(4 * ks + kk) * OP + (arow & 3) * 16 + 4 * g + (arow >> 2)] = av[g][ks];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x4 macc[2][NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) macc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      // step s4 contracts the wave's pixels 4 s4 + kk: a lane's operands of all sixteen steps are 16
      // contiguous floats of its table row (same pairs, same order as one 4-byte read per step: same bits)
      f32x4 a0[4], a1[4], bq[NT][4];
      {
        const f32x4* r0 = reinterpret_cast<const f32x4*>(Bw + arow * OP + kk * 16);                  // k = arow
        const f32x4* r1 = reinterpret_cast<const f32x4*>(Bw + (16 + min(arow, 3)) * OP + kk * 16);  // k = 16 + arow
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a0[q] = r0[q];
          a1[q] = r1[q];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4* rb = reinterpret_cast<const f32x4*>(gg + (wave * GGC + nt * 16 + arow) * OP + kk * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[nt][q] = rb[q];  // dead pixels carry gg == 0
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        float ah[2];
        ah[0] = a0[s4 >> 2][s4 & 3];
        ah[1] = (arow < 4) ? a1[s4 >> 2][s4 & 3] : 0.0f;  // (k = 19 is the zero pad)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv = bq[nt][s4 >> 2][s4 & 3];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            macc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[mt], bv, macc[mt][nt], 0, 0, 0);
        }
      }
      WALDO_PRIO_OFF(WALDO_K1_PRIO_MASK, 8);
      WALDO_STAMP(14);
      __syncthreads();  // every wave is done reading gg: reuse its bytes for the accumulators
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gg[((wave * 2 + mt) * NT + nt) * 256 + r * 64 + lane] = macc[mt][nt][r];
      __syncthreads();
      // wave-uniform base + 32-bit index: a per-thread 64-bit address here was hoisted out of the frame
      // loop, spilled, and its reload (s_waitcnt vmcnt(0)) made every frame wait for the record stores
      float* gp = gmap_partial + ((int64_t)f * ntiles + tile) * gmap_partial_floats(L);
#pragma unroll
      for (int o0 = 0; o0 < 2 * NT * 256; o0 += kBlock) {
        const int o = o0 + (int)threadIdx.x;
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += gg[w * 2 * NT * 256 + o];  // fixed order
        const int mt = o / (NT * 256), nt = (o / 256) % NT, r = (o >> 6) & 3, ln = o & 63;
        const int k = mt * 16 + (ln >> 4) * 4 + r;
        const int col = nt * 16 + (ln & 15);
        const int l = col >> 1;
        if (k < K3 && l < L) gp[(l * K3 + k) * 2 + (col & 1)] = sum;
      }
    }
    __syncthreads();  // the stage region, boxred and wbound are re-used by the next frame
    WALDO_STAMP(15);
  }
}

}  // namespace waldo
