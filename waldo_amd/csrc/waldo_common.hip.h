// Shared device/host helpers for the gfx950 WIF warp/composite kernels.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf(), so
// that the kernels which re-evaluate the TPS grid and the bilinear taps (forward, bounding-box
// pre-pass, backward) produce bit-identical coordinates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/waldo_hip.h"
#include "warp_composite_layout.hip.h"

// The WALDO_ABL_* switches below and in the kernels are TIMING-ONLY ablations: several of them compute wrong values
// on purpose (aliased frames, stores or gathers compiled out).  None of them may reach a product build: a translation
// unit compiled with one of them and without -DWALDO_TIMING_ONLY_BUILD does not compile, and a library built with
// that flag reports waldo_version() == 0 (csrc/runtime.hip), which waldo_amd._lib.load() refuses unless the library
// was named explicitly (use_library / bench.py --lib).  tools_dev/build_variant.py passes the flag by itself.
#if (defined(WALDO_ABL_NOFALLBACK) || defined(WALDO_ABL_FPB) || defined(WALDO_ABL_FCW_NOGATHER) ||              \
     defined(WALDO_ABL_FCW_NOOCC) || defined(WALDO_ABL_FCW_NOSTORE) || defined(WALDO_ABL_FWF_NOGATHER) ||       \
     defined(WALDO_ABL_FWF_NORAW) || defined(WALDO_ABL_FWF_ALLSTAGED) || defined(WALDO_ABL_FCB_NOATOMIC) ||     \
     defined(WALDO_ABL_GS_NOATOMIC) || defined(WALDO_ABL_REC_ALIAS) || defined(WALDO_ABL_LAYER_ALIAS) ||        \
     defined(WALDO_ABL_K1_LDS_PAD) || defined(WALDO_ABL_NO_REC_STORE) || defined(WALDO_ABL_ROWS_NOP2)) &&                                       \
    !defined(WALDO_TIMING_ONLY_BUILD)
#error "WALDO_ABL_* timing-only ablations need -DWALDO_TIMING_ONLY_BUILD (tools_dev/build_variant.py): not a product build"
#endif

namespace waldo {

void set_error(const char* fmt, ...);
int launch_status(const char* what);
bool debug_option(int option);  // waldo_set_debug_option (tests only; all off by default)

// Fills and copies inside the entry points are KERNELS, never hipMemsetAsync / hipMemcpyAsync: captured into a HIP graph
// those become memset / memcpy NODES, and on this stack (ROCm 7.2, gfx950) the replay of a graph with a large memset node
// that followed an eager kernel on the same stream ended in a memory access fault (round 5: the 107 MB "no winner" fill
// of the grid inversion, tools_dev/repro_graph_parts.py); kernel nodes replay.  One workgroup per 4 KB.
void fill_words(void* dst, unsigned word, size_t bytes, hipStream_t st);  // bytes % 4 == 0, dst 4-byte aligned
void copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t st);

constexpr int kWave = 64;   // gfx950 wavefront
constexpr int kBlock = 256; // 4 waves, one per SIMD of a CU

// ---------------------------------------------------------------------------------------
// Frame indices from device memory (ctx_ts / pred_ts: the reference's gather_time, models/nets/lvd.py:462-467, where
// `tensor.gather(1, ts)` RAISES for an index outside the time axis).  The kernels index with them directly: the value
// is clamped for memory safety and a violation is REPORTED in the caller's sticky status words (include/waldo_hip.h:
// "Frame-index status": words `slot`, `slot + 1` = the limit that was violated (never 0), an offending value) instead
// of passing silently.  The status words may be pinned host memory (the caller reads them without a device
// synchronisation): plain system-scope stores, error path only; every reporting lane writes the same limit.
// ---------------------------------------------------------------------------------------
constexpr int kStatusCtx = 0, kStatusPred = 2;  // word pairs of WALDO_INDEX_STATUS_WORDS

__device__ __forceinline__ int checked_frame(const int64_t* __restrict__ ts, int64_t i, int limit,
                                             int* __restrict__ status, int slot) {
  const int64_t raw = ts[i];
  const int64_t c = min(max(raw, (int64_t)0), (int64_t)(limit - 1));
  if (status != nullptr && raw != c) {
    __hip_atomic_store(status + slot, limit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(status + slot + 1, (int)min(max(raw, (int64_t)INT32_MIN), (int64_t)INT32_MAX), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return (int)c;
}

// ---------------------------------------------------------------------------------------
// XCD-aware work mapping.  MI355X deals consecutive workgroup ids round-robin over its 8 XCDs,
// each with a private 4 MiB L2 (observed behaviour; used for SPEED only, never for correctness).
// Work = nA outer units (frames / frame chunks / layer planes) x nB inner tiles.  Outer unit a is
// pinned to XCD a % 8, so every tile of a frame -- and the halo texels neighbouring tiles share --
// goes through ONE L2 instead of being re-fetched from HBM by several.  Launch 8*ceil(nA/8)*nB
// workgroups; those decoding to a >= nA exit at once.
// ---------------------------------------------------------------------------------------
constexpr int kXcds = 8;

__host__ __device__ inline int64_t xcd_grid(int64_t nA, int64_t nB) {
  return kXcds * ((nA + kXcds - 1) / kXcds) * nB;
}

__device__ __forceinline__ bool xcd_decode(unsigned bid, int nA, int nB, int& a, int& b) {
  const int xcd = (int)(bid & (kXcds - 1));
  const int j = (int)(bid >> 3);
  const int al = j / nB;
  b = j - al * nB;
  a = al * kXcds + xcd;
  return a < nA;
}

// Short launches (a few frames: the passes of the two-kernel backward) would leave XCDs idle with
// one outer unit per frame: a frame's nT tiles are cut into nbands bands of contiguous tiles
// (whole tile rows when nbands divides the row count) and the unit pinned to an XCD is
// (frame, band).  nbands = 1 is xcd_decode.
__host__ __device__ inline int xcd_bands(int nA) {  // smallest band count that makes nA * nbands a multiple of 8
  int g = nA, b = kXcds;
  while (b) { const int t = g % b; g = b; b = t; }
  return kXcds / g;
}
__device__ __forceinline__ bool xcd_decode_banded(unsigned bid, int nA, int nbands, int nT, int inner,
                                                  int& a, int& tile, int& rest) {
  const int tpb = (nT + nbands - 1) / nbands;
  int unit, j;
  if (!xcd_decode(bid, nA * nbands, tpb * inner, unit, j)) return false;
  a = unit / nbands;
  const int t = j / inner;
  rest = j - t * inner;
  tile = (unit - a * nbands) * tpb + t;
  return tile < nT;
}
__host__ inline int64_t xcd_grid_banded(int64_t nA, int nbands, int64_t nT, int64_t inner) {
  return xcd_grid(nA * nbands, ((nT + nbands - 1) / nbands) * inner);
}

// ---------------------------------------------------------------------------------------
// Bilinear taps of grid_sample(mode=bilinear, padding_mode=zeros, align_corners=False).
//   ix = ((x + 1) * W - 1) / 2 ; corners (x0,y0) .. (x0+1,y0+1); a corner outside the image
//   contributes nothing: its weight is zeroed and its address clamped into the image, so every
//   load is in bounds.  Offsets are unsigned BYTE offsets inside one Hi*Wi plane so that loads
//   use the scalar-base + 32-bit-VGPR-offset addressing form.
// ---------------------------------------------------------------------------------------
struct Taps {
  float w00, w01, w10, w11;        // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1), zero when outside
  uint32_t o00, o01, o10, o11;     // byte offsets inside the plane (always in bounds)
  float fx, fy;                    // fractional parts
  float vx0, vx1, vy0, vy1;        // 1.0f / 0.0f validity of column x0, x1 and row y0, y1
  int x0, y0;                      // integer corner (may be -1 .. W / H: unclamped)
};

__device__ __forceinline__ float unnormalize(float c, int size) {
  return ((c + 1.0f) * (float)size - 1.0f) * 0.5f;
}

// integer corner and fractional parts; everything else about a tap set derives from these
struct TapCore {
  float fx, fy;
  int x0, y0;  // may be -2 .. W + 1 / H + 1: unclamped
};

__device__ __forceinline__ TapCore tap_core(float gx, float gy, int Hi, int Wi) {
  float ix = unnormalize(gx, Wi);
  float iy = unnormalize(gy, Hi);
  // keep the float->int conversion defined for wild / NaN coordinates; anything clamped here has
  // all four corners outside the image anyway
  ix = __builtin_amdgcn_fmed3f(ix, -2.0f, (float)Wi + 1.0f);  // (NaN -> -2, as fminf(fmaxf()) gave)
  iy = __builtin_amdgcn_fmed3f(iy, -2.0f, (float)Hi + 1.0f);
  const float x0f = floorf(ix), y0f = floorf(iy);
  TapCore c;
  c.fx = ix - x0f;
  c.fy = iy - y0f;
  c.x0 = (int)x0f;
  c.y0 = (int)y0f;
  return c;
}

// ---------------------------------------------------------------------------------------
// Pixel-unit grids.  The fused kernels evaluate the TPS grid directly in UNNORMALISED coordinates:
//   ix = ((gx + 1) W - 1) / 2 = gx (W / 2) + (W - 1) / 2,   gx = sum_k basis_k m_k
// by scaling the mapping column with W / 2 and adding (W - 1) / 2 to the coefficient of the
// constant basis function (k == K3 - 3: tgt_grid_repr = [phi, 1, x, y], warp.py:36-37) -- the
// un-normalisation of grid_sample then costs nothing per (pixel, layer).  scaled_map() is the ONE
// definition of that operand: the MFMA kernels and the scalar chain of tps_eval() use it, so their
// coordinates are bit-identical.  (Against the reference's two-step evaluation the coordinate
// moves by rounding only, ~1e-5 px at 512 px -- the size of the fp32 noise of its own matmul.)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float scaled_map(float m, bool is_const_term, float half_size, float half_size_m1) {
  return is_const_term ? fmaf(m, half_size, half_size_m1) : m * half_size;
}

// tap_core() for coordinates that are already unnormalised
__device__ __forceinline__ TapCore tap_core_px(float ix, float iy, int Hi, int Wi) {
  // keep the float->int conversion defined for wild / NaN coordinates; anything clamped here has
  // all four corners outside the image anyway
  ix = __builtin_amdgcn_fmed3f(ix, -2.0f, (float)Wi + 1.0f);
  iy = __builtin_amdgcn_fmed3f(iy, -2.0f, (float)Hi + 1.0f);
  const float x0f = floorf(ix), y0f = floorf(iy);
  TapCore c;
  c.fx = ix - x0f;
  c.fy = iy - y0f;
  c.x0 = (int)x0f;
  c.y0 = (int)y0f;
  return c;
}

// all four corners inside the layer: every validity factor is exactly 1
__device__ __forceinline__ bool tap_interior(const TapCore& c, int Hi, int Wi) {
  return (unsigned)c.x0 < (unsigned)(Wi - 1) && (unsigned)c.y0 < (unsigned)(Hi - 1);
}

__device__ __forceinline__ Taps finish_taps(const TapCore& c, int Hi, int Wi) {
  Taps t;
  t.fx = c.fx;
  t.fy = c.fy;
  const int x0 = c.x0, y0 = c.y0;
  int x1 = x0 + 1, y1 = y0 + 1;
  t.x0 = x0;
  t.y0 = y0;
  t.vx0 = (x0 >= 0 && x0 < Wi) ? 1.0f : 0.0f;
  t.vx1 = (x1 >= 0 && x1 < Wi) ? 1.0f : 0.0f;
  t.vy0 = (y0 >= 0 && y0 < Hi) ? 1.0f : 0.0f;
  t.vy1 = (y1 >= 0 && y1 < Hi) ? 1.0f : 0.0f;
  int cx0 = min(max(x0, 0), Wi - 1), cx1 = min(max(x1, 0), Wi - 1);
  int cy0 = min(max(y0, 0), Hi - 1), cy1 = min(max(y1, 0), Hi - 1);
  // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): Hi, Wi < 2^15
  const int r0 = __mul24(cy0, Wi), r1 = __mul24(cy1, Wi);
  t.o00 = (uint32_t)(r0 + cx0) * 4u;
  t.o01 = (uint32_t)(r0 + cx1) * 4u;
  t.o10 = (uint32_t)(r1 + cx0) * 4u;
  t.o11 = (uint32_t)(r1 + cx1) * 4u;
  float wx0 = (1.0f - t.fx) * t.vx0, wx1 = t.fx * t.vx1;
  float wy0 = (1.0f - t.fy) * t.vy0, wy1 = t.fy * t.vy1;
  t.w00 = wx0 * wy0;
  t.w01 = wx1 * wy0;
  t.w10 = wx0 * wy1;
  t.w11 = wx1 * wy1;
  return t;
}

__device__ __forceinline__ Taps make_taps(float gx, float gy, int Hi, int Wi) {
  return finish_taps(tap_core(gx, gy, Hi, Wi), Hi, Wi);
}

__device__ __forceinline__ Taps make_taps_px(float ix, float iy, int Hi, int Wi) {
  return finish_taps(tap_core_px(ix, iy, Hi, Wi), Hi, Wi);
}

// round(x) to int32 in one instruction (floor(x + 0.5); __float2int_rn is v_rndne + v_cvt)
__device__ __forceinline__ int cvt_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ __forceinline__ float ldb(const float* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Bilinear value in the "lerp" form, top + fy (bot - top) with top = v00 + fx (v01 - v00), on the
// corner values times their validity: the same number as the four-weight sum of grid_sample up to
// rounding, in 6 operations per channel instead of 8, and the ONE form every kernel of the fused
// path uses (the LDS-staged kernels evaluate it on packed channel pairs: bit-identical).
// delta: grid_sample(x + delta) - delta of Warper.obj_to_output / bg_to_output (lvd.py:548,559) --
// the shift is applied to the corner values before their validity, exactly as the reference does;
// with delta == 0 the result has the same bits as without.
__device__ __forceinline__ float tap_sample(const float* __restrict__ plane, const Taps& t, float delta = 0.0f) {
  const float p00 = ldb(plane, t.o00) + delta, p01 = ldb(plane, t.o01) + delta;
  const float p10 = ldb(plane, t.o10) + delta, p11 = ldb(plane, t.o11) + delta;
  const float v00 = p00 * (t.vx0 * t.vy0), v01 = p01 * (t.vx1 * t.vy0);
  const float v10 = p10 * (t.vx0 * t.vy1), v11 = p11 * (t.vx1 * t.vy1);
  const float top = fmaf(t.fx, v01 - v00, v00);
  const float bot = fmaf(t.fx, v11 - v10, v10);
  return fmaf(t.fy, bot - top, top) - delta;
}

// sample + partial derivatives w.r.t. the UNNORMALISED coordinates (ix, iy)
__device__ __forceinline__ float tap_sample_d(const float* __restrict__ plane, const Taps& t,
                                              float& ddx, float& ddy, float delta = 0.0f) {
  const float p00 = ldb(plane, t.o00) + delta, p01 = ldb(plane, t.o01) + delta;
  const float p10 = ldb(plane, t.o10) + delta, p11 = ldb(plane, t.o11) + delta;
  float v00 = p00 * (t.vx0 * t.vy0), v01 = p01 * (t.vx1 * t.vy0);
  float v10 = p10 * (t.vx0 * t.vy1), v11 = p11 * (t.vx1 * t.vy1);
  float top = fmaf(t.fx, v01 - v00, v00);
  float bot = fmaf(t.fx, v11 - v10, v10);
  ddx = fmaf(t.fy, (v11 - v10) - (v01 - v00), v01 - v00);
  ddy = bot - top;
  return fmaf(t.fy, ddy, top) - delta;
}

// ---------------------------------------------------------------------------------------
// Wave-level "transpose reduce": every lane holds NV partial sums v[0..NV); on return lane l
// holds the sum over all 64 lanes of v[bitrev6(l)] (valid when bitrev6(l) < NV).  Costs about NV
// cross-lane moves instead of 6*NV for NV independent butterflies.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int bitrev6(int l) {
  return ((l & 1) << 5) | ((l & 2) << 3) | ((l & 4) << 1) | ((l & 8) >> 1) | ((l & 16) >> 3) |
         ((l & 32) >> 5);
}

template <int N, int D>
struct TransposeReduce {
  __device__ __forceinline__ static float run(float (&v)[N], int lane) {
    constexpr int M = (N + 1) / 2;
    float w[M];
    const bool hi = (lane & D) != 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float a = v[2 * m];
      float b = (2 * m + 1 < N) ? v[2 * m + 1] : 0.0f;
      if constexpr (D == 32 || D == 16) {
        // lanes with bit D clear keep a, the others b, each plus its partner's copy of the same:
        // exactly what the row / half swap of (a, b) lays side by side -- one VALU swap and one add
        // instead of two selects, a ds_bpermute and its wait (3/4 of this function's exchanges)
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        if constexpr (D == 32) {
          const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
          w[m] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else {
          const auto r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
          w[m] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
      } else {
        float keep = hi ? b : a;
        float send = hi ? a : b;
        w[m] = keep + __shfl_xor(send, D, kWave);
      }
    }
    if constexpr (D == 1) {
      return w[0];
    } else {
      return TransposeReduce<M, D / 2>::run(w, lane);
    }
  }
};

template <int N>
__device__ __forceinline__ float wave_transpose_reduce(float (&v)[N], int lane) {
  static_assert(N >= 1 && N <= 64, "at most one value per lane");
  return TransposeReduce<N, 32>::run(v, lane);
}

// ---------------------------------------------------------------------------------------
// Streaming 16-byte stores / loads for data that is written once and read once by ANOTHER kernel
// (the records between K1 and K2, the gradient planes K2 writes): the cache policy is a compile-time
// choice so that variants can be measured against each other (tools_dev/ab_bench.py).
//   policy 0: plain store (line stays in the XCD's L2)   1: sc1, write-through (line dropped)
//          2: nt
// ---------------------------------------------------------------------------------------
typedef float f32x4_s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_s __attribute__((ext_vector_type(4)));

#ifndef WALDO_BUF_STORE_AUX
#define WALDO_BUF_STORE_AUX 16  // gfx942 / gfx950 cache-policy bits of a buffer store: 1 sc0, 2 nt, 16 sc1
#endif
template <int POLICY>
__device__ __forceinline__ void stream_store16(float* uniform_base, uint32_t byte_off, int64_t bytes, f32x4_s v) {
  if constexpr (POLICY == 1) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_base, 0, (int)bytes, 0x00020000);
    u32x4_s u;
    __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (int)byte_off, 0, WALDO_BUF_STORE_AUX);
  } else if constexpr (POLICY == 2) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_s*>(reinterpret_cast<char*>(uniform_base) + byte_off));
  } else {
    *reinterpret_cast<f32x4_s*>(reinterpret_cast<char*>(uniform_base) + byte_off) = v;
  }
}

// measured at the headline shape (backward 2.247 ms with plain stores): records nt 2.210, sc1 2.236;
// gradient planes nt 2.218; both nt 2.193; nt LOADS of the records in K2 2.267 (worse)
// timing-only ablation (-DWALDO_ABL_REC_ALIAS=n, wrong values): every frame's records alias those of
// frame f % n, i.e. the records stay resident in the Infinity Cache
#ifdef WALDO_ABL_REC_ALIAS
#define WALDO_REC_FRAME(f) ((f) % WALDO_ABL_REC_ALIAS)
#else
#define WALDO_REC_FRAME(f) (f)
#endif
// timing-only ablation (-DWALDO_ABL_LAYER_ALIAS=n, wrong values): the staged boxes of every frame are read
// from frame f % n, i.e. the layer reads are served by the caches instead of HBM
#ifdef WALDO_ABL_LAYER_ALIAS
#define WALDO_LAYER_FRAME(f) ((f) % WALDO_ABL_LAYER_ALIAS)
#else
#define WALDO_LAYER_FRAME(f) (f)
#endif
#ifndef WALDO_REC_STORE_POLICY
#define WALDO_REC_STORE_POLICY 2
#endif
#ifndef WALDO_REC_LOAD_NT
#define WALDO_REC_LOAD_NT 0
#endif
#ifndef WALDO_GRAD_STORE_POLICY
#define WALDO_GRAD_STORE_POLICY 0  // K2's gradient planes: plain stores.  (Non-temporal was 2 % faster at three
                                   // workgroups per CU; at four it is 2.3 % slower: 2.14 vs 2.19 ms backward.)
#endif

// ---------------------------------------------------------------------------------------
// Exchanges between the four 16-lane ROWS of a wavefront without the LDS crossbar: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange odd rows of one register with even rows of another
// (halves for the 32 form).  Fed the same value twice they return (A, B) with, in every lane, the
// lane's own value in one and its xor-16 (xor-32) partner's in the other -- for a commutative op
// that is all a butterfly step needs: one VALU instruction instead of a ds_bpermute and its wait.
// ---------------------------------------------------------------------------------------
template <class Op>
__device__ __forceinline__ unsigned rows_combine_u(unsigned v, Op op) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = op(r[0], r[1]);
  r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return op(r[0], r[1]);
}

// min / max of a float over the lanes {l, l ^ 16, l ^ 32, l ^ 48} (the same column of the 4 rows)
__device__ __forceinline__ float rows_min(float v) {
  return __uint_as_float(rows_combine_u(__float_as_uint(v), [](unsigned a, unsigned b) {
    return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b)));
  }));
}
__device__ __forceinline__ float rows_max(float v) {
  return __uint_as_float(rows_combine_u(__float_as_uint(v), [](unsigned a, unsigned b) {
    return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b)));
  }));
}
__device__ __forceinline__ float rows_sum(float v) {
  return __uint_as_float(rows_combine_u(__float_as_uint(v), [](unsigned a, unsigned b) {
    return __float_as_uint(__uint_as_float(a) + __uint_as_float(b));
  }));
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using short2_ = __attribute__((ext_vector_type(2))) short;

__device__ __forceinline__ int pk_min(int a, int b) {
  short2_ x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  short2_ r = __builtin_elementwise_min(x, y);
  int o;
  __builtin_memcpy(&o, &r, 4);
  return o;
}

// Reductions over each group of 16 consecutive lanes with DPP row rotations (row_ror:8/4/2/1):
// plain VALU moves, no LDS crossbar (ds_bpermute) and no lgkmcnt waits.  A rotation is not an
// xor exchange, but min / + over all 16 lanes only needs every lane to meet every other once.
template <int ROR>
__device__ __forceinline__ int row_ror_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x120 + ROR, 0xf, 0xf, false);
}

__device__ __forceinline__ int group16_pk_min(int v) {
  v = pk_min(v, row_ror_i<8>(v));
  v = pk_min(v, row_ror_i<4>(v));
  v = pk_min(v, row_ror_i<2>(v));
  v = pk_min(v, row_ror_i<1>(v));
  return v;
}

__device__ __forceinline__ float group16_sum(float v) {
  v += __int_as_float(row_ror_i<8>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<4>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<2>(__float_as_int(v)));
  v += __int_as_float(row_ror_i<1>(__float_as_int(v)));
  return v;
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic but NOT for its outstanding global
// loads / stores (__syncthreads() also emits s_waitcnt vmcnt(0), which would serialise a prefetch
// that is meant to stay in flight across the barrier)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
  return rows_sum(v);
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
  return v;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
  return v;
}

}  // namespace waldo
