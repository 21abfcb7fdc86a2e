// Round 6's bounded experiment, NOT in the product: flow_ctx_warp_kernel's row loop over a per-TILE
// COMPACT LIST of the present layers.  This file is the body of that path; csrc/flow_ctx.hip includes it inside the
// kernel when built with -DWALDO_FCW_COMPACT=<slots> (tools_dev/build_variant.py compactN --only flow_ctx
// -DWALDO_FCW_COMPACT=N [-DWALDO_FCW_COMPACT_CHUNK=M]).
//
// Bit-identical to the product kernel (tests/test_gpu_warper.py, test_gpu_pipeline.py, test_demo.py with --waldo-lib) and
// SLOWER: C5 pipeline, A/B on one box (profiles/r06_ab_flow_ctx_warp_compact_list.txt), flow_ctx_warp per step 5.33-5.51 ms
// (product) against 6.05-6.17 (8 slots), 5.90-6.13 (6 slots), 5.85-6.16 (4 slots).  The kernel that contains both loops
// needs 92-96 VGPRs (the product: 79, six waves per SIMD) and spills 170-230 SGPRs; a slot pays run-time LDS offsets (four
// v_add per tap set), 64-bit scalar plane addresses and an extra barrier + a K x K copy of the order per tile -- more than
// the mask test, ballot and branch an ABSENT layer costs the unrolled loop.  See HISTORY.md, Appendix B.
  if constexpr (KS == 0) {
    if (x >= Wd) return;
  } else {
    // ---- round 6's experiment: the tile's PRESENT layers as a compact list of at most KS slots.
    // Layer 0 and every object whose staged mask passes 0.8999 in some cell of the tile's patch; a layer outside the list
    // fails the ghost test in every pixel of the tile (alpha 0: outputs -1 / + 0, factor 1 in every product).  The row
    // loop then runs over SLOTS -- values in registers by slot, the layer of a slot in an SGPR (plane addresses, LDS
    // record offsets, the order's entries at run-time addresses) -- instead of over all LP layers with a wave-uniform
    // branch each.  Same expressions in the same order per value: the same bits as the loop below, which serves every
    // tile with more than KS present layers, every dense tile, and a row that samples a non-finite alpha (`wild`).
    unsigned tile_bits = nob ? ((wave_bits[0] | wave_bits[1] | wave_bits[2] | wave_bits[3]) | 1u)
                             : (L >= 32 ? 0xffffffffu : (1u << L) - 1u);
    tile_bits = (unsigned)__builtin_amdgcn_readfirstlane((int)tile_bits);
    const int K = __builtin_popcount(tile_bits);
    const bool compact = !dense && staged && K <= KS;  // (uniform over the workgroup)
    __shared__ __attribute__((aligned(16))) float occs[KS * KS];  // the order restricted to the list: [slot i][slot j]
    if (compact) {
      constexpr int RO = OccLds<LP>::kRow;
      if (threadIdx.x < KS * KS) {
        const int i = (int)threadIdx.x / KS, j = (int)threadIdx.x % KS;
        int li = 0, lj = 0;
        unsigned rem = tile_bits;
        for (int k = 0; k < KS && rem; ++k) {  // the k-th set bit
          const int l = __builtin_ctz(rem);
          rem &= rem - 1u;
          if (k == i) li = l;
          if (k == j) lj = l;
        }
        occs[threadIdx.x] = (i < K && j < K) ? occm[li * RO + lj] : 0.0f;
      }
      __syncthreads();
    }
    if (x >= Wd) return;
    if (compact) {
      int lid[KS];
      {
        unsigned rem = tile_bits;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          lid[k] = rem ? __builtin_ctz(rem) : 0;
          rem &= rem - 1u;
        }
      }
      bool done = true;
#pragma unroll 1
      for (int rr = 0; rr < R; ++rr) {
        const int y = y_first + kHdRows * rr;
        if (y >= Hd) break;
        const int64_t p = (int64_t)y * Wd + x;
        const UpTaps ut = up_taps(y, x, rscale, H, W);
        LrTaps lt = lr_taps(y, x, rscale, H, W, lq);
        lt.o00 *= G::kCell, lt.o01 *= G::kCell, lt.o10 *= G::kCell, lt.o11 *= G::kCell;
        float gx0, gy0;
        identity_grid(x, y, Wd, Hd, gx0, gy0);
        const float* ap0 = a01 + (((int64_t)b * Tw + ts) * L) * HWd;
        float a[KS];
        float dis = -INFINITY;
        unsigned active = 0;  // bit k: slot k has a non-zero value in some lane
        bool wild = false;
        constexpr int CH = KS < WALDO_FCW_COMPACT_CHUNK ? KS : WALDO_FCW_COMPACT_CHUNK;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += CH) {
          if (k0 < K) {  // (uniform)
            PairTaps pt[CH];
            f32x2_p ra[CH], rb[CH];
            bool inter[CH], keep[CH];
            unsigned need = 0;
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
              const int k = k0 + kk;
              keep[kk] = false;
              if (k < KS && k < K) {  // (uniform)
                asm volatile("");
                const int l = lid[k];
                const int b00 = lt.o00 + 4 * l, b01 = lt.o01 + 4 * l, b10 = lt.o10 + 4 * l, b11 = lt.o11 + 4 * l;
                float g = 1.0f;
                const bool masked = l >= 1 && nob;
                if (masked) g = up_blend(ut, lrimg[b00 + 2], lrimg[b01 + 2], lrimg[b10 + 2], lrimg[b11 + 2]);
                keep[kk] = !(masked && !(g > 0.9f));
                if (__ballot(keep[kk]) != 0ull) {
                  const f32x2_p v00 = *reinterpret_cast<const f32x2_p*>(lrimg + b00);
                  const f32x2_p v01 = *reinterpret_cast<const f32x2_p*>(lrimg + b01);
                  const f32x2_p v10 = *reinterpret_cast<const f32x2_p*>(lrimg + b10);
                  const f32x2_p v11 = *reinterpret_cast<const f32x2_p*>(lrimg + b11);
                  const float fxl = up_blend(ut, v00[0], v01[0], v10[0], v11[0]);
                  const float fyl = up_blend(ut, v00[1], v01[1], v10[1], v11[1]);
                  need |= 1u << kk;
                  pt[kk] = pair_taps(gx0 + fxl, gy0 + fyl, Hd, Wd, inter[kk]);
                  pair_load(ap0 + (int64_t)l * HWd, pt[kk], ra[kk], rb[kk]);
                }
              }
            }
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
              const int k = k0 + kk;
              float v = 0.0f;
              if (k < KS && (need & (1u << kk))) {
                v = pair_value(ra[kk], rb[kk], pt[kk], inter[kk]);
                v = keep[kk] ? v : 0.0f;
                asm volatile("" : "+v"(v));
                if (__ballot(v != 0.0f) != 0ull) active |= 1u << k;
                wild |= __ballot(!(fabsf(v) <= 3.0e38f)) != 0ull;
              }
              if (k < KS && k < K) dis = nan_max(dis, v);
              if (k < KS) a[k] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
          } else {
#pragma unroll
            for (int kk = 0; kk < CH; ++kk)
              if (k0 + kk < KS) a[k0 + kk] = 0.0f;
          }
        }
        if (wild) {  // (uniform) a non-finite sample: this row and the rest through the loop over all layers
          rr_first = rr;
          done = false;
          break;
        }
        if (K < L) dis = nan_max(dis, 0.0f);  // the layers outside the list: alpha 0
        disocc[(int64_t)m * HWd + p] = dis;
        float ox = 0.0f, oy = 0.0f, amax = -INFINITY, ssum = 0.0f;
        float* acb = alpha_ctx + b * lay.sb + ((m / Tp) % Tc) * lay.stc + tp * lay.stp + p;
        unsigned stored = 0;  // bit l: layer l's plane has received its value
        // four column slots per step, two and two on the packed-fp32 pipe, the rows in slot (= layer) order: the factors
        // and the order of the multiplications of the loop over all layers
#pragma unroll
        for (int j = 0; j < KS; j += 4) {
          f32x2_w prd[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
          if ((active >> j) & 0xfu) {  // (uniform)
#pragma unroll
            for (int i = 0; i < KS; ++i) {
              if (active & (1u << i)) {  // (uniform)
                asm volatile("");
                const f32x4_o o = *reinterpret_cast<const f32x4_o*>(occs + i * KS + j);
                const f32x2_w ai = {a[i], a[i]};
                const f32x2_w one = {1.0f, 1.0f};
                prd[0] = prd[0] * __builtin_elementwise_fma(-ai, (f32x2_w){o[0], o[1]}, one);
                if (j + 2 < KS) prd[1] = prd[1] * __builtin_elementwise_fma(-ai, (f32x2_w){o[2], o[3]}, one);
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (j + k >= KS) break;
            if (active & (1u << (j + k))) {  // (uniform) slot j + k is in the list and non-zero somewhere
              asm volatile("");
              const int lj = lid[j + k];
              const float v = a[j + k] * prd[k >> 1][k & 1];
              {  // the layer's upsampled flow again (the expressions of the sampling loop: the same bits)
                const f32x2_p v00 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o00 + 4 * lj);
                const f32x2_p v01 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o01 + 4 * lj);
                const f32x2_p v10 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o10 + 4 * lj);
                const f32x2_p v11 = *reinterpret_cast<const f32x2_p*>(lrimg + lt.o11 + 4 * lj);
                ox += v * up_blend(ut, v00[0], v01[0], v10[0], v11[0]);
                oy += v * up_blend(ut, v00[1], v01[1], v10[1], v11[1]);
              }
              const float av = v * 2.0f - 1.0f;
              acb[(int64_t)lj * HWd] = av;
              amax = nan_max(amax, av);
              if (SCORE) ssum += (av + 1.0f) / 2.0f;
              stored |= 1u << lj;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if ((int)__builtin_popcount(stored) < L) amax = nan_max(amax, -1.0f);
#pragma unroll
        for (int l = 0; l < LP; ++l)
          if (l < L && !(stored & (1u << l))) acb[(int64_t)l * HWd] = -1.0f;  // (uniform) alpha 0 in every lane: 2 * 0 - 1
        flow[((int64_t)m * 2) * HWd + p] = ox;
        flow[((int64_t)m * 2 + 1) * HWd + p] = oy;
        if (alpha_max != nullptr) alpha_max[(int64_t)m * HWd + p] = amax;
        if (SCORE) score[(int64_t)m * HWd + p] = ssum;
      }
      if (done) return;
    }
  }
