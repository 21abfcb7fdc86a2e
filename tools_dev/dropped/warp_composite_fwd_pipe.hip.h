// Round-5 experiment: the staged forward (warp_composite_fwd_lds.hip.h) with its FRAME LOOP
// SOFTWARE-PIPELINED.  Same arithmetic, same instruction chains per value: the same bits
// (tests/test_gpu_parity.py::test_staged_forward_equals_plain_forward and test_pipelined_forward_same_bits).
//
// The product kernel of round 4 walks the frames of a chunk strictly one after the other:
//     (A) grid on MFMA  (B) ranges  (C) transposition  |barrier|  (D) boxes, first loads issued
//     (E) L x { LDS store, issue l + kAhead, |barrier|, taps }   composite, stores   |barrier|
// so between the last staged layer of frame f and the first box loads of frame f + 1 -- composite, stores, two
// barriers, the whole (A)-(D) of the next frame: by the stamps of tools_dev/fwd_stamps.py about 6.5 k of a tile-frame's
// 15 k cycles -- the workgroup has NO layer load in flight.  Here frame f + 1's (A), (B) and the WRITE half of (C) run
// before the barrier of frame f's last layer (the transposition slices have LDS of their own instead of overlaying
// the image buffers: 17 KB more at L = 8, still four workgroups per CU), all its boxes are derived right behind that
// barrier and the first kAhead loads issued BEFORE the last layer is sampled and the frame composited; the grid is
// read back from the wave's own slice at the top of the next frame:
//     E(f): layers 0 .. L-2 as before;  layer L-1: LDS store, A(f+1), fold(f+2), B(f+1), C-write(f+1), |barrier|,
//     D(f+1), ISSUE the first loads, taps of layer L-1;  composite(f), stores;  C-read(f+1).
// With slices of their own the two barriers around (C) / at the end of a frame are gone: L barriers per frame instead
// of L + 2.  Live across the composite: kAhead staging items more than the round-4 kernel.  Only for a compile-time
// layer count (L == LP) of at most 8.
#pragma once
// included behind warp_composite_fwd_lds.hip.h (uses its staging helpers)

namespace waldo {

#ifndef WALDO_FWD_PIPE
#define WALDO_FWD_PIPE 1  // compiled in (L == LP in {4, 8}, 4 | W) behind WALDO_DEBUG_FWD_PIPELINED; 0: not compiled
#endif
#ifndef WALDO_FWDP_PREFETCH_MAP
#define WALDO_FWDP_PREFETCH_MAP 1  // mapping loads of the next frame issued kAhead layers before its phase (A)
#endif
#ifndef WALDO_FWDP8_WAVES
#define WALDO_FWDP8_WAVES 4
#endif

template <int LP, bool FOLD>
__global__ __launch_bounds__(4 * kWave, LP <= 8 ? WALDO_FWDP8_WAVES : (LP <= 12 ? WALDO_FWD12_WAVES : 2)) void warp_composite_fwd_pipe_kernel(
    const float* __restrict__ layers, const float* __restrict__ basis_t,
    const float* __restrict__ mapping, const float* __restrict__ inv_kernel,
    const float* __restrict__ src_pts, const float* __restrict__ occ, float* __restrict__ rgb,
    float* __restrict__ alpha_out, int F, int H, int W, int frames_per_block, int ntx,
    int ntiles, int nchunks, int nbands, float delta) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int L = LP, NW = 4;
  constexpr int K3 = 19, KS = (K3 + 3) / 4;
  constexpr int NC = 2 * LP, NT = (NC + 15) / 16, GGC = NT * 16, TP = GGC + 1;
  constexpr int kThreads = NW * kWave, kCap = kStageCap, kBuf = 4 * kCap;
  constexpr int kImgFloats = 2 * kBuf;
  constexpr int kTFloats = NW * kWave * TP;  // per-wave transposition slices: LDS of their own
  constexpr int kMain = kImgFloats;
  static_assert(LP >= 2 && LP % 2 == 0, "the next frame's ranges are published one layer barrier ahead");
  const int64_t HW = (int64_t)H * W;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane_k = threadIdx.x & (kWave - 1), arow_k = lane_k & 15, kk_k = lane_k >> 4;
  int chunk, tile, rest_;
  if (!xcd_decode_banded(blockIdx.x, nchunks, nbands, ntiles, 1, chunk, tile, rest_)) return;
  WALDO_FSTAMP(0);
  const int col0 = (tile % ntx) * kLdsTile;
  const int row0 = (tile / ntx) * kLdsTile + wave * 4;
  PixelMap pm;
  pm.live = col0 + arow_k < W && row0 + kk_k < H;
  pm.p = (int64_t)min(row0 + kk_k, H - 1) * W + min(col0 + arow_k, W - 1);
  const int64_t p = pm.p;

  constexpr int kMapFloats = FOLD ? 2 * LP * K3 * 2 : 0;
  __shared__ __attribute__((aligned(16))) float lds[kMain + kTFloats + NW * GGC * 2 + kMapFloats];
  float* img = lds;
  float* tslice = lds + kMain;
  float* boxred = tslice + kTFloats;  // [wave][column][min, max]
  float* smap = boxred + NW * GGC * 2;
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
  auto fold_first = [&](int fm) {  // (see the product kernel: a thread's entries loaded together, then summed)
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
  static_assert(kMain % 4 == 0, "cleared sixteen bytes at a time");
  for (int i = threadIdx.x; i < kMain / 4; i += kThreads) reinterpret_cast<f32x4*>(lds)[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  float av[4][KS];  // MFMA A operand (the TPS basis of the tile's pixels), kept across the frames of the chunk
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint32_t pa = (uint32_t)(min(row0 + g, H - 1) * W + min(col0 + arow_k, W - 1));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk_k;
      const float bs = ldb(basis_t, ((uint32_t)min(k, K3 - 1) * (uint32_t)HW + pa) * 4u);
      av[g][ks] = (k < K3) ? bs : 0.0f;
    }
  }
  const int f0 = chunk * frames_per_block;
  const int f1 = min(F, f0 + frames_per_block);
  if (f0 >= f1) return;
  fold_first(f0);
  __syncthreads();
  WALDO_FSTAMP(1);

  const float half_size = 0.5f * (float)((arow_k & 1) ? H : W);
  const float half_size_m1 = 0.5f * (float)(((arow_k & 1) ? H : W) - 1);
  // per-thread indices, re-materialised per frame (the product kernel's note on loop-invariant code motion)
  int lane = lane_k, arow = arow_k, kk = kk_k;

  // ---- (A) TPS grid of every layer of frame f on the matrix pipe, in pixel units
  // the B operand of frame f: this lane's K3 / 4 entries of the mapping (from the folded LDS table, or from memory:
  // WALDO_FWDP_PREFETCH_MAP issues those loads a few layers ahead of phase (A), behind the frame's last box loads)
  auto load_map = [&](int f, float (&mraw)[KS][NT]) {
    const float* mp = FOLD ? smap + (f & 1) * (LP * K3 * 2) : mapping + (int64_t)f * L * K3 * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int k = 4 * ks + kk, col = nt * 16 + arow, l = col >> 1;
        mraw[ks][nt] = mp[(min(l, L - 1) * K3 + min(k, K3 - 1)) * 2 + (col & 1)];
      }
  };
  auto phase_a = [&](float (&mraw)[KS][NT], f32x4 (&acc)[4][NT]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[g][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
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
  };
  // ---- (B) range of every grid coordinate over this wave's pixels -> boxred
  auto phase_b = [&](const f32x4 (&acc)[4][NT]) {
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
  };
  // ---- (C) accumulators -> one pixel per lane, through this wave's slice of LDS: the write half, then (a frame
  // later) the read half -- same wave, other lanes
  auto phase_c_write = [&](const f32x4 (&acc)[4][NT]) {
    float* T = tslice + wave * (kWave * TP);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * g + kk * 4 + r) * TP + nt * 16 + arow] = acc[g][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto phase_c_read = [&](float (&gx)[LP], float (&gy)[LP]) {
    const float* T = tslice + wave * (kWave * TP);
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      gx[l] = T[lane * TP + 2 * l];
      gy[l] = T[lane * TP + 2 * l + 1];
    }
    // (the next write to the slice, a frame's layers later, must stay behind these reads)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // ---- (D) box of the 2x2 blocks of the layers [0, NL) from the published ranges
  auto phase_d = [&](auto nl, int (&bx0)[LP], int (&by0)[LP], int (&bw)[LP], int (&bh)[LP]) {
    constexpr int NL = decltype(nl)::value;
    constexpr int NTL = (2 * NL + 15) / 16;  // column groups that hold those layers
    int lo_t[NTL], hi_t[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
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
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int nt = (2 * l) / 16, ln = (2 * l) % 16;
      const int xmin = __builtin_amdgcn_readlane(lo_t[nt], ln), xmax = __builtin_amdgcn_readlane(hi_t[nt], ln);
      const int ymin = __builtin_amdgcn_readlane(lo_t[nt], ln + 1), ymax = __builtin_amdgcn_readlane(hi_t[nt], ln + 1);
      bx0[l] = xmin & ~3;
      by0[l] = ymin;
      bw[l] = ((xmax - bx0[l] + 1) + 3) & ~3;
      bh[l] = ymax - ymin + 1;
    }
  };
  int item_l = threadIdx.x;
  // one staging item (two texels of the four channel planes) of layer l's box of frame f -> r
  auto issue = [&](int f, int l, int bx0, int by0, int bw, int bh, StageRegs& r) {
    const float* src = layers + ((int64_t)f * L + l) * 4 * HW;
    const bool fits = bh * bw <= kCap;
    const int bw2 = bw >> 1, n = fits ? bh * bw2 : 1;
    const int ox = fits ? __mul24(by0, W) + bx0 : 0;
    const float rcp = __builtin_amdgcn_rcpf((float)bw2);
    const int item = min(item_l, n - 1);
    const int rr = (int)(((float)item + 0.5f) * rcp);
    const int xh = item - __mul24(rr, bw2);
    const unsigned off = (unsigned)(ox + __mul24(rr, W) + 2 * xh) * 4u;
    r.c0 = ld8(src, off);
    r.c1 = ld8(src + HW, off);
    r.c2 = ld8(src + 2 * HW, off);
    r.c3 = ld8(src + 3 * HW, off);
  };

  constexpr int kAhead = LP < WALDO_STAGE_AHEAD ? LP : WALDO_STAGE_AHEAD;
  constexpr bool kMapAhead = !FOLD && WALDO_FWDP_PREFETCH_MAP != 0 && kAhead < LP;
  static_assert(kCap / 2 == kThreads, "one box item per lane");
  f32x4 acc[4][NT];
  float gx[LP], gy[LP];
  int bx0[LP], by0[LP], bw[LP], bh[LP];
  StageRegs stg[LP];  // fully unrolled: a layer's registers live from its load to its LDS store

  // ---- prologue: the first frame's grid, ranges, boxes and first loads
  float mraw[KS][NT];
  load_map(f0, mraw);
  phase_a(mraw, acc);
  if (f0 + 1 < f1) fold_mapping(f0 + 1);
  phase_b(acc);
  phase_c_write(acc);
  __syncthreads();
  WALDO_FSTAMP(2);
  phase_d(std::integral_constant<int, LP>{}, bx0, by0, bw, bh);
#pragma unroll
  for (int l = 0; l < kAhead; ++l) issue(f0, l, bx0[l], by0[l], bw[l], bh[l], stg[l]);
  WALDO_FSTAMP(3);

  for (int f = f0; f < f1; ++f) {
    lane = lane_k, arow = arow_k, kk = kk_k, item_l = threadIdx.x;
    asm volatile("" : "+v"(lane), "+v"(arow), "+v"(kk), "+v"(item_l));
    const bool more = f + 1 < f1;  // block-uniform
    if (f == f0 + 1) WALDO_FSTAMP(8);  // (stamps 8 .. 12: the SECOND frame of the chunk, the loop's steady state)
    phase_c_read(gx, gy);
    float s[LP][4];
#pragma unroll
    for (int l = 0; l < LP; ++l) {
      // this layer's box (the arrays are overwritten with the next frame's behind the last layer's barrier)
      const int lbx0 = bx0[l], lby0 = by0[l], lbw = bw[l], lbh = bh[l];
      const bool fits = lbh * lbw <= kCap;  // block-uniform
      if (fits) {
        const int n = lbh * (lbw >> 1);
        if (item_l < n) stage_store(img + (l & 1) * kBuf, item_l, stg[l]);
      }
      if (l + kAhead < LP) issue(f, l + kAhead, bx0[l + kAhead], by0[l + kAhead], bw[l + kAhead], bh[l + kAhead], stg[l + kAhead]);
      if (kMapAhead && l == LP - 1 - kAhead && more) load_map(f + 1, mraw);  // behind this frame's last box loads
      if (l == LP - 1 && more) {
        // the next frame's grid and ranges, published by the barrier this layer needs anyway
        if (!kMapAhead) load_map(f + 1, mraw);
        phase_a(mraw, acc);
        if (f + 2 < f1) fold_mapping(f + 2);  // table (f + 2) & 1: last read by phase (A) of frame f, barriers ago
        phase_b(acc);
        phase_c_write(acc);
      }
      __syncthreads();  // buffer l&1 complete; buffer (l+1)&1 no longer read by anyone
      if (l == LP - 1 && more) {
        // the next frame's boxes; the loads of its first kAhead layers are in flight from here on
        phase_d(std::integral_constant<int, LP>{}, bx0, by0, bw, bh);
#pragma unroll
        for (int q = 0; q < kAhead; ++q) issue(f + 1, q, bx0[q], by0[q], bw[q], bh[q], stg[q]);
      }
      if (fits) {
        const TapCore tc = tap_core_px(gx[l], gy[l], H, W);
        const float* b0 = img + (l & 1) * kBuf;
        f32x2_t sv[2];
        if (__ballot(!tap_interior(tc, H, W)) == 0ull) {
          const int idx = __mul24(tc.y0 - lby0, lbw) + (tc.x0 - lbx0);
          const PairBlock pb = read_block(b0, idx, lbw);
#pragma unroll
          for (int q = 0; q < 2; ++q) sv[q] = lerp2(pb.p00[q], pb.p01[q], pb.p10[q], pb.p11[q], tc.fx, tc.fy);
        } else {
          const BoxTaps t = make_box_taps(tc, H, W);
          const int idx = min(max(__mul24(t.yb - lby0, lbw) + (t.xb - lbx0), 0), kCap - lbw - 2);
          const PairBlock pb = assign_corners(read_block(b0, idx, lbw), t.cs, t.rs);
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
    if (f == f0) WALDO_FSTAMP(5);
    if (f == f0 + 1) WALDO_FSTAMP(11);
    // ---- composite: a_0 = 1 (lvd.py:105), a_l = (s_l3 + 1) / 2
    float a[LP];
#pragma unroll
    for (int l = 0; l < LP; ++l) a[l] = (s[l][3] + 1.0f) * 0.5f;
    a[0] = 1.0f;
    const float* oc = occ + (int64_t)f * L * L;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float apv[LP];
#pragma unroll
    for (int j = 0; j < LP; j += 2) {
      const int j1 = j + 1 < LP ? j + 1 : j;
      f32x2 pr = {1.0f, 1.0f};
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const f32x2 o = {oc[i * L + j], oc[i * L + j1]};
        const f32x2 avv = {a[i], a[i]};
        pr = pr * ((f32x2){1.0f, 1.0f} - avv * o);
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
      if (alpha_out != nullptr && pm.live) alpha_out[((int64_t)f * L + j) * HW + p] = 2.0f * ap - 1.0f;
    }
    if (pm.live) {
      float* o = rgb + (int64_t)f * 3 * HW + p;
      o[0] = 2.0f * r - 1.0f;
      o[HW] = 2.0f * g - 1.0f;
      o[2 * HW] = 2.0f * b - 1.0f;
    }
    if (f == f0) WALDO_FSTAMP(6);
    if (f == f0 + 1) WALDO_FSTAMP(12);
    // (no closing barrier: the next frame's layer 0 goes into buffer 0, last read at layer L - 2, a barrier ago; the
    // ranges and the mapping tables are a layer barrier apart from their readers as well)
  }
}

}  // namespace waldo
