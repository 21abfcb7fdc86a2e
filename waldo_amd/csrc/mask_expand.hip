// f3: the mask dilation of WIF.inpaint -- `expand` (tools/utils.py:300-323; its callers wif.py:77, 108-112, 204):
// `num` rounds of one-pixel growth towards the south, north, east and west IN THAT ORDER, every step over the whole
// plane and seeing the result of the step before it:
//
//     south:  m[y, x] = max(m[y, x], alpha * m[y - 1, x])   (y >= 1)        north:  ... alpha * m[y + 1, x]   (y <= H - 2)
//     east:   m[y, x] = max(m[y, x], alpha * m[y, x - 1])   (x >= 1)        west:   ... alpha * m[y, x + 1]   (x <= W - 2)
//
// with torch.maximum's NaN rule (a NaN on either side wins).  The hard form (`soft=False`) is the same recurrence on
// m != 0 with alpha = 1: an OR.  The framework spelled one step as two launches over the plane (the combination, the
// copy back): 30 rounds = 240 launches for ONE shadow mask; this is one launch.
//
// A workgroup owns a 64 x 64 tile of one plane and keeps the tile with a `num`-pixel apron in LDS.  What a round does
// to a pixel depends on the pixels within Chebyshev distance 1 before the round, so after r rounds the apron is stale
// r pixels deep (it never saw what lies beyond it) and after `num` rounds exactly the tile itself is right -- the
// stale ring is not updated any more (round r works on the window shrunk by r - 1).  The recurrence is run LITERALLY,
// step by step, so every input (negative values, NaN, alpha > 1) gives the framework's bits: a thread owns one column
// of the window for the south + north pair (walking down with the old value of the row above and the south-stepped
// values of its row and the next one in registers, it writes in place), then one row for the east + west pair; two
// barriers per round.
#include "waldo_common.hip.h"

namespace waldo {

constexpr int kExpTile = 64;      // the tile a workgroup owns
constexpr int kExpMaxNum = 30;    // rounds per launch (the apron): (64 + 60)^2 floats = 61.5 KB of LDS
constexpr int kExpSide = kExpTile + 2 * kExpMaxNum;
constexpr int kExpPitch = kExpSide + 1;  // odd: a thread per ROW walks conflict-free
constexpr int kExpThreads = 128;

// torch.maximum(d, alpha * s): a NaN on either side wins.  HARD: both are 0.0 / 1.0 and alpha is 1 -- an OR of the bits
// (one instruction; fmaxf costs three: the compiler quiets both operands first).
template <bool HARD>
__device__ __forceinline__ float expand_combine(float d, float s, float alpha) {
  if (HARD) return __uint_as_float(__float_as_uint(d) | __float_as_uint(s));
  const float v = alpha * s;
  return __builtin_isunordered(d, v) ? d + v : fmaxf(d, v);
}

// one line (a column: step = pitch; a row: step = 1) of `n` cells; `lo`: the first cell may look at a cell before it
// (it is not the image's first row / column), `hi`: the last may look at one after it.  `back` = the south / east step,
// `fwd` = the north / west step, in that order, as whole-line steps:  s(i) = combine(m(i), m(i - 1)),
// out(i) = combine(s(i), s(i + 1)).  The walk writes out(i) in place BEHIND itself and reads ahead of itself, eight
// cells per trip so that the eight LDS reads are in flight together (one read per cell, each waiting for the store
// before it, made a 30-round growth 0.69 ms); trips that lie inside the line -- all but the last -- carry no per-cell
// tests (a wavefront issues one vector instruction per four cycles and two wavefronts walk a tile: the instruction
// count per cell IS the kernel's time).
template <bool HARD, bool BACK, bool FWD>
__device__ __forceinline__ void expand_line_steps(float* p, int step, int n, bool lo, bool hi, float alpha) {
  constexpr int kTrip = 8;
  const float before = lo ? p[-step] : 0.0f;      // m(-1), old
  float cur = p[0];                               // m(i), old
  float s_cur = (BACK && lo) ? expand_combine<HARD>(cur, before, alpha) : cur;
  const int last = hi ? n : n - 1;                // the last cell that may be read
  int i0 = 0;
  for (; i0 + kTrip <= last; i0 += kTrip) {       // cells i0 .. i0 + 7 all have a next cell
    float nxt[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) nxt[k] = p[(i0 + k + 1) * step];  // m(i + 1), old
#pragma unroll
    for (int k = 0; k < kTrip; ++k) {
      const float s_next = BACK ? expand_combine<HARD>(nxt[k], cur, alpha) : nxt[k];
      p[(i0 + k) * step] = FWD ? expand_combine<HARD>(s_cur, s_next, alpha) : s_cur;
      cur = nxt[k];
      s_cur = s_next;
    }
  }
  for (; i0 < n; i0 += kTrip) {
    float nxt[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) nxt[k] = (i0 + k + 1 <= last) ? p[(i0 + k + 1) * step] : 0.0f;
#pragma unroll
    for (int k = 0; k < kTrip; ++k) {
      const int i = i0 + k;
      if (i < n) {
        const bool has_next = i + 1 <= last;
        const float s_next = (BACK && has_next) ? expand_combine<HARD>(nxt[k], cur, alpha) : nxt[k];
        p[i * step] = (FWD && has_next) ? expand_combine<HARD>(s_cur, s_next, alpha) : s_cur;
        cur = nxt[k];
        s_cur = s_next;
      }
    }
  }
}

template <bool HARD>
__device__ __forceinline__ void expand_line(float* p, int step, int n, bool lo, bool hi, bool back, bool fwd,
                                            float alpha) {
  if (n <= 0) return;
  if (back && fwd)
    expand_line_steps<HARD, true, true>(p, step, n, lo, hi, alpha);
  else if (back)
    expand_line_steps<HARD, true, false>(p, step, n, lo, hi, alpha);
  else
    expand_line_steps<HARD, false, true>(p, step, n, lo, hi, alpha);
}

template <bool HARD>
__global__ __launch_bounds__(kExpThreads) void mask_expand_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                    int H, int W, int num, int steps, float alpha) {
  __shared__ float img[kExpSide * kExpPitch];
  const int tx0 = blockIdx.x * kExpTile, ty0 = blockIdx.y * kExpTile;
  const int64_t plane = (int64_t)blockIdx.z * H * W;
  // the window in image coordinates, clipped to the image: [wx0, wx1) x [wy0, wy1); LDS cell (0, 0) = (tx0 - num, ty0 - num)
  const int ox = tx0 - num, oy = ty0 - num;
  const int side = kExpTile + 2 * num;
  const int wx0 = max(ox, 0), wx1 = min(ox + side, W), wy0 = max(oy, 0), wy1 = min(oy + side, H);
  // (a window row is at most 124 pixels: one per thread; eight rows' loads in flight -- one row per trip, each trip
  // waiting for its load, was 60 us of a launch)
  static_assert(kExpSide <= kExpThreads, "a thread per window column");
  {
    const int x = wx0 + (int)threadIdx.x;
    for (int y0 = wy0; y0 < wy1; y0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (x < wx1 && y0 + k < wy1) ? in[plane + (int64_t)(y0 + k) * W + x] : 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (x < wx1 && y0 + k < wy1)
          img[(y0 + k - oy) * kExpPitch + (x - ox)] = HARD ? (v[k] != 0.0f ? 1.0f : 0.0f) : v[k];
    }
  }
  __syncthreads();
  const bool south = steps & 1, north = steps & 2, east = steps & 4, west = steps & 8;
  for (int r = 0; r < num; ++r) {
    // this round's window: the apron's stale ring (r deep, only on sides that are not the image's own edge) is left alone
    const int ax0 = max(ox + r, 0), ax1 = min(ox + side - r, W), ay0 = max(oy + r, 0), ay1 = min(oy + side - r, H);
    if (south || north) {
      for (int x = ax0 + (int)threadIdx.x; x < ax1; x += kExpThreads)
        expand_line<HARD>(img + (ay0 - oy) * kExpPitch + (x - ox), kExpPitch, ay1 - ay0, ay0 > wy0, ay1 < wy1, south, north,
                    alpha);
      __syncthreads();
    }
    if (east || west) {
      for (int y = ay0 + (int)threadIdx.x; y < ay1; y += kExpThreads)
        expand_line<HARD>(img + (y - oy) * kExpPitch + (ax0 - ox), 1, ax1 - ax0, ax0 > wx0, ax1 < wx1, east, west, alpha);
      __syncthreads();
    }
  }
  const int ex1 = min(tx0 + kExpTile, W), ey1 = min(ty0 + kExpTile, H);
  for (int y = ty0; y < ey1; ++y)
    for (int x = tx0 + (int)threadIdx.x; x < ex1; x += kExpThreads)
      out[plane + (int64_t)y * W + x] = img[(y - oy) * kExpPitch + (x - ox)];
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_mask_expand_fwd(const float* mask, float* out, float* scratch, int64_t planes, int H, int W,
                                     int num, int steps, int soft, float alpha, waldo_stream_t stream) {
  if (planes < 0 || H < 1 || W < 1 || num < 0 || steps < 0 || steps > 15 || planes > 65535) {
    set_error("waldo_mask_expand_fwd: bad arguments planes=%lld H=%d W=%d num=%d steps=%d", (long long)planes, H, W, num,
              steps);
    return WALDO_EINVAL;
  }
  if (planes == 0) return WALDO_OK;
  if (!mask || !out || (num > kExpMaxNum && !scratch)) {
    set_error("waldo_mask_expand_fwd: null pointer%s", mask && out ? " (more than 30 rounds need the scratch plane set)" : "");
    return WALDO_EINVAL;
  }
  if (mask == out || (scratch && (scratch == out || scratch == mask))) {
    set_error("waldo_mask_expand_fwd: mask, out and scratch must be different buffers (tiles read their neighbours' pixels)");
    return WALDO_EINVAL;
  }
  const dim3 grid((unsigned)((W + kExpTile - 1) / kExpTile), (unsigned)((H + kExpTile - 1) / kExpTile), (unsigned)planes);
  // rounds compose: expand(m, a + b) = expand(expand(m, a), b); passes of at most 30 rounds ping-pong between `out`
  // and `scratch` so that the last one lands in `out` (a hard mask is 0 / 1 after the first pass: binarising it again
  // changes nothing)
  const int passes = num == 0 ? 1 : (num + kExpMaxNum - 1) / kExpMaxNum;
  const float a = soft ? alpha : 1.0f;
  const float* src = mask;
  int left = num;
  for (int p = 0; p < passes; ++p) {
    float* dst = ((passes - 1 - p) % 2 == 0) ? out : scratch;
    const int n = min(left, kExpMaxNum);
    if (soft)
      mask_expand_kernel<false><<<grid, dim3(kExpThreads), 0, (hipStream_t)stream>>>(src, dst, H, W, n, steps, a);
    else
      mask_expand_kernel<true><<<grid, dim3(kExpThreads), 0, (hipStream_t)stream>>>(src, dst, H, W, n, steps, a);
    src = dst;
    left -= n;
  }
  return launch_status("waldo_mask_expand_fwd");
}
