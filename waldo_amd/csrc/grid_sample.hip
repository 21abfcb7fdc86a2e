// A4/A5: bilinear backward warp, y = grid_sample(x + delta, grid) - delta with the PyTorch
// defaults (bilinear, zeros, align_corners=False) -- the primitive behind
// Warper.obj_to_output / bg_to_output / obj_from_input / bg_from_input
// (models/nets/lvd.py:502-559) and the HD gathers of grid_to_flow_ctx / input_to_output
// (lvd.py:801,837).  One thread per output pixel, looping over the C channel planes so the tap
// weights are computed once; a wavefront covers 64 consecutive output pixels.
#include "waldo_common.hip.h"

namespace waldo {

__device__ __forceinline__ int64_t in_index(int64_t n, int64_t outer_div, int64_t inner) {
  return (n / outer_div) * inner + (n % inner);
}

// Where output map n goes: slot (n / group) * stride + offset + n % group of an output tensor of (slots, C, Ho, Wo)
// maps.  (N, 1, 0) is the plain (N, C, Ho, Wo) output.  Warper.layer_to_output (lvd.py:533-537) concatenates the
// warped background (one map per frame) and the warped objects (No per frame) along the layer axis: with
// (group, stride, offset) = (1, L, 0) and (No, L, 1) the two calls write straight into the concatenated tensor.
struct OutSlots {
  int64_t group, stride, offset;
  __device__ __forceinline__ int64_t slot(int64_t n) const { return (n / group) * stride + offset + n % group; }
};

// The sampled image is `scale * input + bias` (Warper.grid_to_flow warps `(alpha + 1) / 2`, lvd.py:602-606 / 716-720,
// without the image being written first).  Sampling is linear and the weights of the in-range taps sum to wsum:
// sample(s x + b + delta) - delta = s sample(x) + (b + delta) wsum - delta.  (1, 0): the plain call, the same bits as
// before there was a PreAffine -- fmaf(1, v, shift) = v + shift, delta + 0 = delta.
struct PreAffine {
  float scale, bias;
};

// The sample of an all-ones image at the same taps: tap_sample() on the constant 1 (the same operations on the
// validity products: the bits grid_sample(ones, grid) gives).  Warper.grid_to_flow_ctx asks for it next to the layer
// flows (`is_obj = obj_to_output(ones) > 0.9`, lvd.py:785-791): the same grids, so the mask is a by-product.
__device__ __forceinline__ float tap_sample_ones(const Taps& t) {
  const float v00 = t.vx0 * t.vy0, v01 = t.vx1 * t.vy0, v10 = t.vx0 * t.vy1, v11 = t.vx1 * t.vy1;
  const float top = fmaf(t.fx, v01 - v00, v00);
  const float bot = fmaf(t.fx, v11 - v10, v10);
  return fmaf(t.fy, bot - top, top);
}

__global__ __launch_bounds__(kBlock) void grid_sample2d_fwd_kernel(
    const float* __restrict__ input, const float* __restrict__ grid, float* __restrict__ output,
    float* __restrict__ mask_out, int64_t N, int C, int Hi, int Wi, int64_t HWo, int tiles, float delta,
    int64_t outer_div, int64_t inner, int64_t g_outer_div, int64_t g_inner, OutSlots os, PreAffine pre) {
  const int64_t n = blockIdx.x / tiles;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HWo) return;
  const float2 g = *reinterpret_cast<const float2*>(grid + (in_index(n, g_outer_div, g_inner) * HWo + p) * 2);
  const Taps t = make_taps(g.x, g.y, Hi, Wi);
  if (mask_out != nullptr) mask_out[n * HWo + p] = tap_sample_ones(t);
  const int64_t HWi = (int64_t)Hi * Wi;
  const float* in = input + in_index(n, outer_div, inner) * C * HWi;
  float* out = output + os.slot(n) * C * HWo + p;
  // sum of the weights of the in-range taps: sample(x + delta) = sample(x) + delta * wsum
  const float wsum = (t.w00 + t.w01) + (t.w10 + t.w11);
  const float shift = fmaf(delta + pre.bias, wsum, -delta);
  for (int c = 0; c < C; ++c) out[(int64_t)c * HWo] = fmaf(pre.scale, tap_sample(in + (int64_t)c * HWi, t), shift);
}

// Sum of v over the lanes from this one to the end of its RUN (consecutive lanes with the same scatter address;
// `stop`: the run ends at this lane).  After the step of distance d a lane that has not met its run's end has summed d
// more lanes, all of its run, so lane + d exists: no range check.
// Runs end at the 16-lane ROW boundaries (WALDO_GS_RUN_ROWS): the four steps are DPP row shifts -- two VALU operations
// each, no LDS crossbar -- where six steps of two ds_bpermute round trips each made the hot wavefronts (the few whose
// pixels land inside the object's canvas) crawl; a run cut by a row boundary costs one more atomic.  The runs are
// 2 - 4 lanes long where a 64 x 64 canvas covers ~100 frame pixels.
#ifndef WALDO_GS_RUN_ROWS
#define WALDO_GS_RUN_ROWS 1
#endif
constexpr int kRunSpan = WALDO_GS_RUN_ROWS ? 16 : kWave;
template <int D>
__device__ __forceinline__ int row_from_above(int v) {  // lane i <- lane i + D of its row (row_shl:D; past the row: 0)
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + D, 0xf, 0xf, true);
}
__device__ __forceinline__ float run_sum(float v, bool stop) {
#if WALDO_GS_RUN_ROWS
  int st = stop ? 1 : 0;
#define WALDO_GS_STEP(D)                                                         \
  {                                                                              \
    const float vo = __int_as_float(row_from_above<D>(__float_as_int(v)));       \
    const int so = row_from_above<D>(st);                                        \
    if (!st) {                                                                   \
      v += vo;                                                                   \
      st = so;                                                                   \
    }                                                                            \
  }
  WALDO_GS_STEP(1) WALDO_GS_STEP(2) WALDO_GS_STEP(4) WALDO_GS_STEP(8)
#undef WALDO_GS_STEP
  return v;
#else
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const float vo = __shfl_down(v, d, kWave);
    const int so = __shfl_down((int)stop, d, kWave);
    if (!stop) {
      v += vo;
      stop = so != 0;
    }
  }
  return v;
#endif
}

#ifndef WALDO_GS_RUNS
#define WALDO_GS_RUNS 1  // the scatter of grad_input summed over runs of equal addresses inside a wavefront first
#endif
__global__ __launch_bounds__(kBlock) void grid_sample2d_bwd_kernel(
    const float* __restrict__ input, const float* __restrict__ grid,
    const float* __restrict__ grad_output, float* __restrict__ grad_input,
    float* __restrict__ grad_grid, int64_t N, int C, int Hi, int Wi, int64_t HWo, int tiles,
    float delta, int64_t outer_div, int64_t inner, OutSlots gos, PreAffine pre) {
  const int64_t n = blockIdx.x / tiles;
  const int64_t p_ = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  // (a thread past the last pixel works on the last one with zero weights: it takes part in the wavefront's sums)
  const bool live = p_ < HWo;
  if (!WALDO_GS_RUNS && !live) return;
  const int64_t p = live ? p_ : HWo - 1;
  const float2 g = *reinterpret_cast<const float2*>(grid + (n * HWo + p) * 2);
  const Taps t = make_taps(g.x, g.y, Hi, Wi);
  const int64_t HWi = (int64_t)Hi * Wi;
  const int64_t nin = in_index(n, outer_div, inner);
  const float* in = input + nin * C * HWi;
  const float* go = grad_output + gos.slot(n) * C * HWo + p;  // (the gradient of a tensor written through OutSlots)
  const float db = delta + pre.bias;
  float gix = 0.0f, giy = 0.0f;
  const float m00 = t.vx0 * t.vy0, m01 = t.vx1 * t.vy0, m10 = t.vx0 * t.vy1, m11 = t.vx1 * t.vy1;
  // ---- the scatter.  A wavefront is 64 consecutive output pixels; where the output is finer than the input (object
  // canvases of 64 x 64 warped into a 128 x 256 frame) neighbouring pixels hit the same input texel and their float
  // atomics queue up at one address: the kernel is bound by them (58 us per call at the LVD recipe, 18 with the
  // scatter compiled out).  Runs of equal addresses in consecutive lanes are summed with shuffles first and the
  // run's first lane issues ONE atomic (an address that comes back later in the wavefront starts a new run: still
  // correct).  Sums in a different order than one atomic per lane -- which had no fixed order either.
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t key[4] = {t.o00, t.o01, t.o10, t.o11};
  const float wq[4] = {live ? t.w00 : 0.0f, live ? t.w01 : 0.0f, live ? t.w10 : 0.0f, live ? t.w11 : 0.0f};
  bool stop[4], head[4];
  const bool scatter = grad_input != nullptr &&
                       (!WALDO_GS_RUNS || __ballot(wq[0] != 0.0f || wq[1] != 0.0f || wq[2] != 0.0f || wq[3] != 0.0f) != 0ull);
  if (WALDO_GS_RUNS && scatter) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#if WALDO_GS_RUN_ROWS  // (the neighbours inside the row; the row's first / last lane starts / ends a run whatever they hold)
      const uint32_t nxt = (uint32_t)row_from_above<1>((int)key[q]);
      const uint32_t prv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[q], 0x111, 0xf, 0xf, true);  // row_shr:1
#else
      const uint32_t nxt = (uint32_t)__shfl_down((int)key[q], 1, kWave), prv = (uint32_t)__shfl_up((int)key[q], 1, kWave);
#endif
      stop[q] = (lane & (kRunSpan - 1)) == kRunSpan - 1 || nxt != key[q];
      head[q] = (lane & (kRunSpan - 1)) == 0 || prv != key[q];
    }
  }
  for (int c = 0; c < C; ++c) {
    const float gv = go[(int64_t)c * HWo];
    const float gvs = gv * pre.scale;  // d out / d input texel = scale * weight
    if (grad_grid != nullptr) {
      const float* pl = in + (int64_t)c * HWi;
      const float v00 = fmaf(pre.scale, ldb(pl, t.o00), db) * m00, v01 = fmaf(pre.scale, ldb(pl, t.o01), db) * m01;
      const float v10 = fmaf(pre.scale, ldb(pl, t.o10), db) * m10, v11 = fmaf(pre.scale, ldb(pl, t.o11), db) * m11;
      const float ddx = fmaf(t.fy, (v11 - v10) - (v01 - v00), v01 - v00);
      const float top = fmaf(t.fx, v01 - v00, v00);
      const float bot = fmaf(t.fx, v11 - v10, v10);
      gix = fmaf(gv, ddx, gix);
      giy = fmaf(gv, bot - top, giy);
    }
#ifndef WALDO_ABL_GS_NOATOMIC  // timing-only ablation: without the scatter
    if (scatter) {  // (wave-uniform)
      float* gp = grad_input + (nin * C + c) * HWi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (WALDO_GS_RUNS) {
          const float sum = run_sum(gvs * wq[q], stop[q]);
          if (head[q] && sum != 0.0f) atomicAdd(gp + (key[q] >> 2), sum);
        } else if (wq[q] != 0.0f) {
          atomicAdd(gp + (key[q] >> 2), gvs * wq[q]);
        }
      }
    }
#endif
  }
  if (grad_grid != nullptr && live) {
    float2* o = reinterpret_cast<float2*>(grad_grid + (n * HWo + p) * 2);
    *o = make_float2(gix * (0.5f * (float)Wi), giy * (0.5f * (float)Hi));
  }
}

// Measured and dropped in round 4: the scatter summed in LDS first.  The kernel above is bound by its global atomics
// (at the LVD recipe 58 us per call against 18 us with the scatter compiled out, -DWALDO_ABL_GS_NOATOMIC), so a
// variant gave every workgroup a BAND of output rows of one map, an LDS window of the input rows the band's taps
// reach (ds_add_f32), and one coalesced global atomic per non-zero texel of the window at the end -- correct (it
// passed the parity tests against the per-tap form and the oracle) and SLOWER: 84 / 74 / 77 / 72 us per call with
// bands of 16 / 4 / 2 / 8 rows (profiles/r04_ab_grid_sample_bwd_window.txt).  A band is a serial loop of dependent
// round trips per wavefront (grid -> taps -> store) where the per-pixel kernel keeps 20 000 workgroups in flight,
// the grid is read twice (window placement), and ds_add_f32 retires ~3 cycles per lane.

// Four consecutive output pixels per thread (HWo % 4 == 0): the grid and the outputs move as 16-byte vectors
// and a lane has four pixels' tap loads in flight: forward -12 % (15.6 -> 13.6 us per call at the LVD recipe).
// The BACKWARD was tried both ways and stays at one pixel per thread (57 us): four neighbouring pixels per
// thread make every tap instruction of a wave span four times the footprint (95 us), four pixels 256 apart
// keep the footprint and still lose (64 us); an early exit for wavefronts whose footprints all lie outside the input
// (93 % of them at the LVD recipe), alone or per step of the four-pixel form, changes nothing either (56 us): the
// kernel is not waiting for its taps.  Per pixel the arithmetic is that of the kernel above.
typedef float f32x4_gs __attribute__((ext_vector_type(4)));
typedef float f32x2_gs __attribute__((ext_vector_type(2)));

// Round 4, two things about the taps.  (1) The two taps of a row are ONE 8-byte load at the pair origin xb =
// clamp(x0, 0, Wi - 2) (4-byte aligned: gfx950 takes unaligned dwordx2 loads): half the gather instructions; within one
// texel of the left / right border the pair sits a column off the footprint and its elements are re-assigned to the
// corners (pair_value() of flow_ctx_common.hip.h, restated on a Taps: the same bits as tap_sample()).  (2) A wavefront
// none of whose 256 pixels has a corner inside the input -- object canvases warped into the frame cover a part of it
// -- loads nothing: zeros-padding gives exactly `- delta` there.  At the KITTI recipe the warp of the object flows
// writes 1.3 GB per call and took 0.55 ms.
#ifndef WALDO_GS_PAIRS
#define WALDO_GS_PAIRS 1
#endif
struct PairOff {
  uint32_t ob0, ob1;
  int shift;  // x0 - xb: -1 / 0 / +1 (beyond that every corner is invalid)
};
__device__ __forceinline__ PairOff pair_off(const Taps& t, int Hi, int Wi) {
  const int xb = min(max(t.x0, 0), Wi - 2);
  const int cy0 = min(max(t.y0, 0), Hi - 1), cy1 = min(max(t.y0 + 1, 0), Hi - 1);
  PairOff q;
  q.ob0 = (uint32_t)(__mul24(cy0, Wi) + xb) * 4u;
  q.ob1 = (uint32_t)(__mul24(cy1, Wi) + xb) * 4u;
  q.shift = t.x0 - xb;
  return q;
}
// tap_sample(plane, t) from the two pairs: the corner values picked from the pair elements, then its operations
__device__ __forceinline__ float pair_sample(const f32x2_gs a, const f32x2_gs b, const Taps& t, int shift) {
  const float p00 = shift > 0 ? a[1] : a[0], p01 = shift < 0 ? a[0] : a[1];
  const float p10 = shift > 0 ? b[1] : b[0], p11 = shift < 0 ? b[0] : b[1];
  const float v00 = p00 * (t.vx0 * t.vy0), v01 = p01 * (t.vx1 * t.vy0);
  const float v10 = p10 * (t.vx0 * t.vy1), v11 = p11 * (t.vx1 * t.vy1);
  const float top = fmaf(t.fx, v01 - v00, v00);
  const float bot = fmaf(t.fx, v11 - v10, v10);
  return fmaf(t.fy, bot - top, top);
}

__global__ __launch_bounds__(kBlock) void grid_sample2d_fwd4_kernel(
    const float* __restrict__ input, const float* __restrict__ grid, float* __restrict__ output,
    float* __restrict__ mask_out, int64_t N, int C, int Hi, int Wi, int64_t HWo, int tiles, float delta,
    int64_t outer_div, int64_t inner, int64_t g_outer_div, int64_t g_inner, OutSlots os, PreAffine pre) {
  const int64_t n = blockIdx.x / tiles;
  const int64_t p = ((int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x) * 4;
  if (p >= HWo) return;
  const f32x4_gs* gp = reinterpret_cast<const f32x4_gs*>(grid + (in_index(n, g_outer_div, g_inner) * HWo + p) * 2);
  const f32x4_gs g0 = gp[0], g1 = gp[1];
  Taps t[4];
  t[0] = make_taps(g0[0], g0[1], Hi, Wi);
  t[1] = make_taps(g0[2], g0[3], Hi, Wi);
  t[2] = make_taps(g1[0], g1[1], Hi, Wi);
  t[3] = make_taps(g1[2], g1[3], Hi, Wi);
  const int64_t HWi = (int64_t)Hi * Wi;
  const float* in = input + in_index(n, outer_div, inner) * C * HWi;
  float* out = output + os.slot(n) * C * HWo + p;
  float shift[4];
  bool touches = false;  // some corner of some pixel of this lane lies inside the input
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float wsum = (t[q].w00 + t[q].w01) + (t[q].w10 + t[q].w11);
    shift[q] = fmaf(delta + pre.bias, wsum, -delta);
    touches |= (t[q].vx0 + t[q].vx1) * (t[q].vy0 + t[q].vy1) != 0.0f;
  }
  const bool pairs = WALDO_GS_PAIRS && Wi >= 2;  // (uniform)
  if (pairs && __ballot(touches) == 0ull) {      // (wave-uniform) every corner outside: 0 * texel + shift
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<f32x4_gs*>(out + (int64_t)c * HWo) = (f32x4_gs){shift[0], shift[1], shift[2], shift[3]};
    if (mask_out != nullptr) *reinterpret_cast<f32x4_gs*>(mask_out + n * HWo + p) = (f32x4_gs){0.0f, 0.0f, 0.0f, 0.0f};
    return;
  }
  if (pairs) {
    PairOff po[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) po[q] = pair_off(t[q], Hi, Wi);
    for (int c = 0; c < C; ++c) {
      const char* plane = reinterpret_cast<const char*>(in + (int64_t)c * HWi);
      f32x2_gs a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = *reinterpret_cast<const f32x2_gs*>(plane + po[q].ob0);
        b[q] = *reinterpret_cast<const f32x2_gs*>(plane + po[q].ob1);
      }
      f32x4_gs o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaf(pre.scale, pair_sample(a[q], b[q], t[q], po[q].shift), shift[q]);
      *reinterpret_cast<f32x4_gs*>(out + (int64_t)c * HWo) = o;
    }
  } else {
    for (int c = 0; c < C; ++c) {
      f32x4_gs o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaf(pre.scale, tap_sample(in + (int64_t)c * HWi, t[q]), shift[q]);
      *reinterpret_cast<f32x4_gs*>(out + (int64_t)c * HWo) = o;
    }
  }
  if (mask_out != nullptr) {
    f32x4_gs o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = tap_sample_ones(t[q]);
    *reinterpret_cast<f32x4_gs*>(mask_out + n * HWo + p) = o;
  }
}

static int check_gs(const char* fn, int64_t N, int C, int Hi, int Wi, int Ho, int Wo,
                    int64_t outer_div, int64_t inner) {
  if (N < 0 || C < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || outer_div < 1 || inner < 1) {
    set_error("%s: bad shape N=%lld C=%d in=%dx%d out=%dx%d outer_div=%lld inner=%lld", fn,
              (long long)N, C, Hi, Wi, Ho, Wo, (long long)outer_div, (long long)inner);
    return WALDO_EINVAL;
  }
  const int64_t tiles = ((int64_t)Ho * Wo + kBlock - 1) / kBlock;
  if ((int64_t)Hi * Wi > 1073741823 || N * tiles > 2147483647) {
    set_error("%s: problem too large for one launch", fn);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

static int grid_sample2d_fwd_launch(const char* fn, const float* input, const float* grid, float* output, float* mask_out,
                                    int64_t N, int C, int Hi, int Wi, int Ho, int Wo, float delta, int64_t outer_div,
                                    int64_t inner, int64_t grid_outer_div, int64_t grid_inner, OutSlots os,
                                    PreAffine pre, waldo_stream_t stream) {
  if (os.group < 1 || os.stride < os.group || os.offset < 0 || os.offset + os.group > os.stride) {
    set_error("%s: bad output slots (group %lld, stride %lld, offset %lld)", fn, (long long)os.group,
              (long long)os.stride, (long long)os.offset);
    return WALDO_EINVAL;
  }
  int rc = check_gs(fn, N, C, Hi, Wi, Ho, Wo, outer_div, inner);
  if (rc) return rc;
  if (grid_outer_div < 1 || grid_inner < 1) {
    set_error("%s: bad grid broadcast (%lld, %lld)", fn, (long long)grid_outer_div, (long long)grid_inner);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!input || !grid || !output) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
  const int64_t HWo = (int64_t)Ho * Wo;
  if (HWo % 4 == 0) {
    const int tiles4 = (int)((HWo / 4 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(grid_sample2d_fwd4_kernel, dim3((unsigned)(N * tiles4)), dim3(kBlock), 0,
                       (hipStream_t)stream, input, grid, output, mask_out, N, C, Hi, Wi, HWo, tiles4, delta, outer_div, inner,
                       grid_outer_div, grid_inner, os, pre);
    return launch_status(fn);
  }
  const int tiles = (int)((HWo + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(grid_sample2d_fwd_kernel, dim3((unsigned)(N * tiles)), dim3(kBlock), 0,
                     (hipStream_t)stream, input, grid, output, mask_out, N, C, Hi, Wi, HWo, tiles, delta,
                     outer_div, inner, grid_outer_div, grid_inner, os, pre);
  return launch_status(fn);
}

extern "C" int waldo_grid_sample2d_fwd(const float* input, const float* grid, float* output,
                                       int64_t N, int C, int Hi, int Wi, int Ho, int Wo,
                                       float delta, int64_t outer_div, int64_t inner,
                                       int64_t grid_outer_div, int64_t grid_inner, waldo_stream_t stream) {
  return grid_sample2d_fwd_launch("waldo_grid_sample2d_fwd", input, grid, output, nullptr, N, C, Hi, Wi, Ho, Wo, delta,
                                  outer_div, inner, grid_outer_div, grid_inner, OutSlots{N > 0 ? N : 1, N > 0 ? N : 1, 0},
                                  PreAffine{1.0f, 0.0f}, stream);
}

extern "C" int waldo_grid_sample2d_ex_fwd(const float* input, const float* grid, float* output, float* mask_out,
                                          int64_t N, int C, int Hi, int Wi, int Ho, int Wo, float delta,
                                          int64_t outer_div, int64_t inner, int64_t grid_outer_div,
                                          int64_t grid_inner, int64_t out_group, int64_t out_stride,
                                          int64_t out_offset, float pre_scale, float pre_bias,
                                          waldo_stream_t stream) {
  return grid_sample2d_fwd_launch("waldo_grid_sample2d_ex_fwd", input, grid, output, mask_out, N, C, Hi, Wi, Ho, Wo,
                                  delta, outer_div, inner, grid_outer_div, grid_inner,
                                  OutSlots{out_group, out_stride, out_offset}, PreAffine{pre_scale, pre_bias}, stream);
}

static int grid_sample2d_bwd_launch(const char* fn, const float* input, const float* grid, const float* grad_output,
                                    float* grad_input, float* grad_grid, int64_t N, int C, int Hi, int Wi, int Ho,
                                    int Wo, float delta, int64_t outer_div, int64_t inner, OutSlots gos, PreAffine pre,
                                    waldo_stream_t stream) {
  if (gos.group < 1 || gos.stride < gos.group || gos.offset < 0 || gos.offset + gos.group > gos.stride) {
    set_error("%s: bad gradient slots (group %lld, stride %lld, offset %lld)", fn, (long long)gos.group,
              (long long)gos.stride, (long long)gos.offset);
    return WALDO_EINVAL;
  }
  int rc = check_gs(fn, N, C, Hi, Wi, Ho, Wo, outer_div, inner);
  if (rc) return rc;
  if (N == 0) return WALDO_OK;
  if (!input || !grid || !grad_output) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
  if (!grad_input && !grad_grid) return WALDO_OK;
  const int64_t HWo = (int64_t)Ho * Wo;
  const int tiles = (int)((HWo + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(grid_sample2d_bwd_kernel, dim3((unsigned)(N * tiles)), dim3(kBlock), 0,
                     (hipStream_t)stream, input, grid, grad_output, grad_input, grad_grid, N, C,
                     Hi, Wi, HWo, tiles, delta, outer_div, inner, gos, pre);
  return launch_status(fn);
}

extern "C" int waldo_grid_sample2d_bwd(const float* input, const float* grid,
                                       const float* grad_output, float* grad_input,
                                       float* grad_grid, int64_t N, int C, int Hi, int Wi, int Ho,
                                       int Wo, float delta, int64_t outer_div, int64_t inner,
                                       waldo_stream_t stream) {
  return grid_sample2d_bwd_launch("waldo_grid_sample2d_bwd", input, grid, grad_output, grad_input, grad_grid, N, C, Hi,
                                  Wi, Ho, Wo, delta, outer_div, inner, OutSlots{N > 0 ? N : 1, N > 0 ? N : 1, 0},
                                  PreAffine{1.0f, 0.0f}, stream);
}

extern "C" int waldo_grid_sample2d_ex_bwd(const float* input, const float* grid, const float* grad_output,
                                          float* grad_input, float* grad_grid, int64_t N, int C, int Hi, int Wi,
                                          int Ho, int Wo, float delta, int64_t outer_div, int64_t inner,
                                          int64_t gout_group, int64_t gout_stride, int64_t gout_offset,
                                          float pre_scale, float pre_bias, waldo_stream_t stream) {
  return grid_sample2d_bwd_launch("waldo_grid_sample2d_ex_bwd", input, grid, grad_output, grad_input, grad_grid, N, C,
                                  Hi, Wi, Ho, Wo, delta, outer_div, inner,
                                  OutSlots{gout_group, gout_stride, gout_offset}, PreAffine{pre_scale, pre_bias}, stream);
}
