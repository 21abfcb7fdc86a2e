// K2 of the two-kernel backward: gradient w.r.t. one layer plane, gathered per SOURCE tile.
//
// Restates the input-gradient of F.grid_sample (bilinear, zeros, align_corners=False) -- the
// four-corner scatter of grad * weight that the reference gets from autograd through
// models/nets/lvd.py:548,559 -- for the records the pixel kernel (K1) left behind:
//   records (grid x, grid y, a'_l, g_alpha) per (frame, layer, pixel); the contribution of pixel p
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
#include "waldo_common.hip.h"

namespace waldo {

constexpr int kSrcRows = 32, kSrcCols = 64;          // S tile
constexpr int kSrcTex = kSrcRows * kSrcCols;          // 2048 texels (x4 channels)
constexpr int kCellPix = kCellRows * kCellCols;       // 128
constexpr int kCellShift = kCellPix == 64 ? 6 : (kCellPix == 128 ? 7 : 8);
static_assert((1 << kCellShift) == kCellPix, "cell rows: 4, 8 or 16");
constexpr int kG2Waves = 8;
constexpr int kG2Threads = kG2Waves * kWave;          // 512
constexpr int kMaxHit = 192 * 8 / kCellRows;                          // cells listed per tile (else: slow scan)

__device__ __forceinline__ int4 load_box(const int* cellbox, int64_t idx) {
  const int4 r = reinterpret_cast<const int4*>(cellbox)[idx];
  return make_int4(r.x, -r.y, r.z, -r.w);  // (x min, x max, y min, y max); empty: x min > x max
}

// LDS image of S: [channel][row][kPitch] int32.  kPitch = 80 = 16 (mod 32): the LDS serves a
// wave-instruction as two 32-lane halves over 32 banks; a wave covers 4 pixel rows x 16 columns,
// so with this pitch the two rows of a half land on disjoint bank ranges (pitch 64 is a 4-way
// conflict).  Behind every channel plane, one dump word per lane takes the taps that fall outside
// S (the same word offset in every plane, so a tap's four channel adds differ only in the
// instruction's immediate offset).
constexpr int kPitch = 80;
constexpr int kImgWords = kSrcRows * kPitch;           // image words per channel
constexpr int kPlane = kImgWords + kWave;              // + the dump words
constexpr int kDump = kImgWords;                       // + lane

// taps of one candidate pixel into the S image.  Branch-free: a tap outside S adds to the lane's
// own dump word -- never to a shared address, where same-address adds would serialise.  (A corner
// outside the LAYER has weight 0 and adds 0 wherever it lands.)  This loop is VALU-issue bound
// (rocprofv3: SQ_INSTS_VALU * 4 cycles ~ 3/4 of the kernel), hence two VALU instructions per add:
// the four corner addresses are selected once, the channel planes are immediate offsets.
__device__ __forceinline__ void splat_pixel(int* img, int lane, bool on, const Taps& t,
                                            const float4 rec, float g0, float g1, float g2,
                                            float scale, int sx0, int sy0) {
  const int lx0 = t.x0 - sx0, ly0 = t.y0 - sy0;
  const bool cx0 = (unsigned)lx0 < (unsigned)kSrcCols, cx1 = (unsigned)(lx0 + 1) < (unsigned)kSrcCols;
  const bool cy0 = on && (unsigned)ly0 < (unsigned)kSrcRows, cy1 = on && (unsigned)(ly0 + 1) < (unsigned)kSrcRows;
  const int base = ly0 * kPitch + lx0;
  const int dump = kDump + lane;
  int* a00 = img + ((cx0 && cy0) ? base : dump);
  int* a01 = img + ((cx1 && cy0) ? base + 1 : dump);
  int* a10 = img + ((cx0 && cy1) ? base + kPitch : dump);
  int* a11 = img + ((cx1 && cy1) ? base + kPitch + 1 : dump);
  float gv[4];
  gv[0] = rec.x * g0;
  gv[1] = rec.x * g1;
  gv[2] = rec.x * g2;
  gv[3] = rec.y;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float gs = gv[c] * scale;
    atomicAdd(a00 + c * kPlane, cvt_round(gs * t.w00));
    atomicAdd(a01 + c * kPlane, cvt_round(gs * t.w01));
    atomicAdd(a10 + c * kPlane, cvt_round(gs * t.w10));
    atomicAdd(a11 + c * kPlane, cvt_round(gs * t.w11));
  }
}

// does any tap of this pixel fall into S?
__device__ __forceinline__ bool touches(const TapCore& t, int sx0, int sy0) {
  const int lx0 = t.x0 - sx0, ly0 = t.y0 - sy0;
  return lx0 >= -1 && lx0 < kSrcCols && ly0 >= -1 && ly0 < kSrcRows;
}

__global__ __launch_bounds__(kG2Threads, 2) void warp_composite_splat_kernel(
    const float4* __restrict__ rec, const float* __restrict__ grad_rgb,
    const int* __restrict__ cellbox, const unsigned* __restrict__ cellbound,
    float* __restrict__ grad_layers, int F, int L, int H, int W, int nsx, int nstiles, int ncx,
    int ncells) {
  const int64_t HW = (int64_t)H * W;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  // a FRAME is pinned to one XCD, source-tile-major inside it: the L layer planes of one source
  // tile run back to back.  Their candidate pixels are (nearly) the same neighbourhood of the
  // frame, so its grad_rgb is read once from HBM instead of once per layer (it was 40 % of this
  // kernel's FETCH_SIZE with layer-major order); the records of a plane are still read ~1.8
  // times by neighbouring tiles, from L2.
  int fi, rest;
  if (!xcd_decode(blockIdx.x, F, L * nstiles, fi, rest)) return;
  const int64_t f = fi;
  const int64_t fl = f * L + rest % L;
  const int stile = rest / L;
  const int sx0 = (stile % nsx) * kSrcCols, sy0 = (stile / nsx) * kSrcRows;
  const int sx1 = min(sx0 + kSrcCols, W) - 1, sy1 = min(sy0 + kSrcRows, H) - 1;

  __shared__ __attribute__((aligned(16))) int lds[4 * kPlane + kMaxHit + 2 * kG2Waves + 4];
  int* img = lds;
  int* hitlist = lds + 4 * kPlane;
  int* wcount = hitlist + kMaxHit;                      // hits per wave (current chunk)
  float* wbound = reinterpret_cast<float*>(wcount + kG2Waves);

  {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    static_assert((4 * kPlane) % 4 == 0, "16-byte zero fill");
    for (int e = threadIdx.x; e < kPlane; e += kG2Threads) reinterpret_cast<i32x4*>(img)[e] = (i32x4){0, 0, 0, 0};
  }

  // ---- cells whose box reaches S, listed in cell order (deterministic), and the sum of their
  // contribution bounds.  Chunks of kG2Threads cells; ballot-based compaction inside a wave.
  int nhit = 0;
  float bsum = 0.0f;
  const float cell_rows = (float)kCellRows;
  for (int c0 = 0; c0 < ncells; c0 += kG2Threads) {
    const int c = c0 + threadIdx.x;
    bool hit = false;
    float bnd = 0.0f;
    if (c < ncells) {
      const int4 ob = load_box(cellbox, fl * ncells + c);
      hit = ob.x <= ob.y && ob.x <= sx1 && ob.y >= sx0 && ob.z <= sy1 && ob.w >= sy0;
      // K1 publishes the largest 16-pixel row sum of the cell; a cell has kCellRows rows
      if (hit) bnd = __uint_as_float(cellbound[fl * ncells + c]) * cell_rows;
    }
    const unsigned long long m = __ballot(hit);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    float wsum = bnd;  // fixed butterfly: deterministic
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, kWave);
    if (lane == 0) {
      wcount[wave] = __popcll(m);
      wbound[wave] = wsum;
    }
    __syncthreads();
    int base = nhit;
#pragma unroll
    for (int w = 0; w < kG2Waves; ++w) {
      if (w < wave) base += wcount[w];
      nhit += wcount[w];
      bsum += wbound[w];
    }
    // listed as the cell's pixel origin (row << 16 | column): the division happens once per cell
    if (hit && base + before < kMaxHit)
      hitlist[base + before] = (((c / ncx) * kCellRows) << 16) | ((c % ncx) * kCellCols);
    __syncthreads();
  }
  // fixed-point scale: bsum bounds the magnitude of ANY texel sum (bilinear weights are <= 1);
  // scale = 2^(29 - floor(log2 bsum)) keeps |sum| * scale < 2^30
  const int eB = (int)((__float_as_uint(bsum) >> 23) & 0xffu) - 127;
  const int es = min(29 - eB, 126);
  const float scale = __uint_as_float((unsigned)(127 + es) << 23);
  const float inv_scale = __uint_as_float((unsigned)(127 - es) << 23);
  const float* gplane = grad_rgb + f * 3 * HW;
  const float4* rcp = rec + fl * HW;

  if (bsum > 0.0f) {
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
        k.rc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(rcp) + k.p * 16u);
        k.g0 = ldb(gplane, k.p * 4u);
        k.g1 = ldb(gplane + HW, k.p * 4u);
        k.g2 = ldb(gplane + 2 * HW, k.p * 4u);
      };
      auto splat = [&](const Cand& k) {
        const TapCore tc = tap_core(k.rc.x, k.rc.y, H, W);
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
          splat_pixel(img, lane, any, t, make_float4(k.rc.z, k.rc.w, k.rc.x, k.rc.y), k.g0, k.g1, k.g2,
                      scale, sx0, sy0);
        }
      };
      Cand ka, kb;
      int i = threadIdx.x;
      if (i < total) fetch(i, ka);
      while (i < total) {  // uniform per wave: total is a multiple of 128, the stride of 512
        fetch(i + kG2Threads, kb);
        splat(ka);
        if (i + kG2Threads >= total) break;
        fetch(i + 2 * kG2Threads, ka);
        splat(kb);
        i += 2 * kG2Threads;
      }
    } else {
      // violent warp (more cells reach S than the list holds): scan every cell, wave-uniformly
      for (int c = 0; c < ncells; ++c) {
        const int4 ob = load_box(cellbox, fl * ncells + c);
        const bool hit = ob.x <= ob.y && ob.x <= sx1 && ob.y >= sx0 && ob.z <= sy1 && ob.w >= sy0;
        if (!hit || threadIdx.x >= kCellPix) continue;
        const int py = (c / ncx) * kCellRows + (threadIdx.x >> 4);
        const int px = (c % ncx) * kCellCols + (threadIdx.x & 15);
        if (py < H && px < W) {
          const unsigned p = (unsigned)(__mul24(py, W) + px);
          const float4 rc = rcp[p];
          const float4 rec = make_float4(rc.z, rc.w, rc.x, rc.y);
          const Taps t = make_taps(rec.z, rec.w, H, W);
          splat_pixel(img, lane, true, t, rec, gplane[p], (gplane + HW)[p], (gplane + 2 * HW)[p],
                      scale, sx0, sy0);
        }
      }
    }
  }
  __syncthreads();
  // ---- S is ours alone: plain row-coalesced stores (zeros included)
  float* gbase = grad_layers + fl * 4 * HW;
  if ((W & 3) == 0) {  // 16 bytes per lane: four texels of a row (rows are 16-byte aligned)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < kSrcTex / 4; e += kG2Threads) {
      const int y = sy0 + (e >> 4), x = sx0 + 4 * (e & 15);
      if (y < H && x < W) {
        const unsigned doff = (unsigned)(__mul24(y, W) + x);
        const int li = (e >> 4) * kPitch + 4 * (e & 15);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const i32x4 v = *reinterpret_cast<const i32x4*>(img + c * kPlane + li);
          *reinterpret_cast<f32x4*>(gbase + c * HW + doff) =
              (f32x4){(float)v[0] * inv_scale, (float)v[1] * inv_scale, (float)v[2] * inv_scale, (float)v[3] * inv_scale};
        }
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < kSrcTex; e += kG2Threads) {
    const int y = sy0 + (e >> 6), x = sx0 + (e & 63);
    if (y < H && x < W) {
      const unsigned doff = (unsigned)(__mul24(y, W) + x);
      const int li = (e >> 6) * kPitch + (e & 63);
#pragma unroll
      for (int c = 0; c < 4; ++c) (gbase + c * HW)[doff] = (float)img[c * kPlane + li] * inv_scale;
    }
  }
}

void launch_splat(const float* rec, const float* grad_rgb, const int* cellbox,
                  const unsigned* cellbound, float* grad_layers, int F, int L, int H, int W,
                  hipStream_t st) {
  const int nsx = (W + kSrcCols - 1) / kSrcCols, nsy = (H + kSrcRows - 1) / kSrcRows;
  const int ncx = (W + kCellCols - 1) / kCellCols, ncy = (H + kCellRows - 1) / kCellRows;
  dim3 grid((unsigned)xcd_grid(F, (int64_t)L * nsx * nsy));
  hipLaunchKernelGGL(warp_composite_splat_kernel, grid, dim3(kG2Threads), 0, st,
                     reinterpret_cast<const float4*>(rec),
                     grad_rgb, cellbox, cellbound, grad_layers, F, L, H, W, nsx, nsx * nsy, ncx,
                     ncx * ncy);
}

}  // namespace waldo
