// Class distribution of every object for the layout filter of Warper.grid_to_flow / grid_to_flow_ctx
// (models/nets/lvd.py:624-634 and 731-746), forward + backward.
//
//   win[o][x]   = (alpha[o][x] + 1e-6) * q[o][x],   q[o][x] = sum_n (cls[o][n] + min_cls) * softmax_n(lyt[.][x])[n]
//                                                   (q = 1 without weight_cls)
//   total[o]    = sum_x win[o][x]                   x runs over the Tw frames and H x W pixels of a batch item
//   mean[o][n]  = sum_x win[o][x] * lyt[n][x] / total[o]
//   dist[o][.]  = softmax_n(mean[o][.])
//
// The reference materialises (B, T, No, Nl, H, W) tensors for this; torch restated it as two einsums
// plus softmaxes (0.3 ms of GEMM + elementwise kernels per LVD-recipe step, forward + backward).
// Here: one pass over the pixels per direction.  A workgroup takes 256 pixel-frames: each thread
// stages the class logits (and their softmax) of one of them in LDS; then every wave owns a group of
// four objects over all 256 pixels, a lane keeping the 4 x Nl running sums of its pixels in
// registers.  The lanes are combined by a fixed butterfly and the workgroups' partial sums are added
// in a fixed order by a second, small kernel -- no atomics, bitwise reproducible.  The backward has
// the same reduction shape for grad_cls and is elementwise for grad_alpha; the layout logits are
// data (no gradient: callers whose layout requires one use the framework expression).
#include <math.h>

#include "waldo_common.hip.h"

namespace waldo {

constexpr int kLdGroup = 4;                  // objects per wave
constexpr int kLdWaves = kBlock / kWave;     // 4: the waves of a workgroup = the pixel slices of its chunk
constexpr int kLdChunk = kBlock;             // pixel-frames per workgroup
constexpr int kLdMaxObj = 32, kLdMaxCls = 32;

struct LytView {
  const float* base;
  int64_t batch_stride, frame_stride;  // elements; planes are HW apart, rows contiguous
};

// Phase 1 of both pixel kernels: every thread loads the Nl logits of ONE pixel-frame of the chunk
// (Nl coalesced rows per wave), takes their softmax when the class weighting needs it, and leaves
// both in LDS, class-major ([n][pixel]: conflict-free for the writer and for the readers below).
// The four waves then each take a group of objects over ALL 256 pixels: the softmax (the expensive
// part: ~20 instructions per class) is computed once per pixel instead of once per object group.
template <int NLP>
__device__ __forceinline__ void stage_logits(const LytView& lyt, int b, int64_t i, int64_t npx, int HW,
                                             int Nl, bool want_prob, float* s_lyt, float* s_prob) {
  const int64_t ic = i < npx ? i : npx - 1;
  const int t = (int)(ic / HW), p = (int)(ic - (int64_t)t * HW);
  const float* lp = lyt.base + b * lyt.batch_stride + t * lyt.frame_stride + p;
  float v[NLP];
#pragma unroll
  for (int n = 0; n < NLP; ++n) v[n] = lp[(int64_t)min(n, Nl - 1) * HW];
#pragma unroll
  for (int n = 0; n < NLP; ++n) s_lyt[n * kBlock + threadIdx.x] = n < Nl ? v[n] : 0.0f;
  if (want_prob) {
    float mx = v[0];
#pragma unroll
    for (int n = 1; n < NLP; ++n) mx = fmaxf(mx, n < Nl ? v[n] : mx);
    float den = 0.0f;
#pragma unroll
    for (int n = 0; n < NLP; ++n) {
      v[n] = n < Nl ? expf(v[n] - mx) : 0.0f;
      den += v[n];
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int n = 0; n < NLP; ++n) s_prob[n * kBlock + threadIdx.x] = v[n] * inv;
  }
}

// q = sum_n (cls[n] + min_cls) prob[n], as sum_n cls[n] prob[n] + min_cls sum_n prob[n]: the class
// vector stays a scalar operand (cls is wave-uniform), one fma per class
template <int NLP>
__device__ __forceinline__ float class_weight(const float* __restrict__ c, const float (&pr)[NLP], int Nl,
                                              float min_cls) {
  float q = 0.0f, ps = 0.0f;
#pragma unroll
  for (int n = 0; n < NLP; ++n) {
    q = fmaf(n < Nl ? c[n] : 0.0f, pr[n], q);
    ps += pr[n];
  }
  return fmaf(min_cls, ps, q);
}

// ---- forward, pass 1: partial sums of win * lyt and of win per (workgroup, object)
// partial: (B, No, NLP + 1, chunks) -- the chunk index fastest, so that pass 2 reads rows; column
// NLP is the total
template <int NLP>
__global__ __launch_bounds__(kBlock) void lyt_dist_partial_kernel(
    const float* __restrict__ alpha, LytView lyt, const float* __restrict__ cls,
    float* __restrict__ partial, int Tw, int La, int obj0, int No, int Nl, int HW, int chunks,
    float min_cls) {
  __shared__ float s_lyt[NLP * kBlock];
  __shared__ float s_prob[NLP * kBlock];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t npx = (int64_t)Tw * HW;
  const bool weighted = cls != nullptr;
  stage_logits<NLP>(lyt, b, (int64_t)chunk * kLdChunk + threadIdx.x, npx, HW, Nl, weighted, s_lyt, s_prob);
  __syncthreads();
  const int ngroups = (No + kLdGroup - 1) / kLdGroup;
  for (int grp = wave; grp < ngroups; grp += kLdWaves) {
    float acc[kLdGroup][NLP], tot[kLdGroup];
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j) {
      tot[j] = 0.0f;
#pragma unroll
      for (int n = 0; n < NLP; ++n) acc[j][n] = 0.0f;
    }
    for (int sl = 0; sl < kLdWaves; ++sl) {
      const int px = sl * kWave + lane;
      const int64_t i = (int64_t)chunk * kLdChunk + px;
      const bool live = i < npx;
      const int64_t ic = live ? i : npx - 1;
      const int t = (int)(ic / HW), p = (int)(ic - (int64_t)t * HW);
      const float* ap = alpha + (((int64_t)b * Tw + t) * La + obj0) * HW + p;
      float av[kLdGroup];
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) av[j] = ap[(int64_t)min(grp * kLdGroup + j, No - 1) * HW];
      float lv[NLP], pr[NLP];
#pragma unroll
      for (int n = 0; n < NLP; ++n) {
        lv[n] = s_lyt[n * kBlock + px];
        pr[n] = weighted ? s_prob[n * kBlock + px] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = min(grp * kLdGroup + j, No - 1);
        const float q = weighted ? class_weight<NLP>(cls + ((int64_t)b * No + o) * Nl, pr, Nl, min_cls) : 1.0f;
        const float win = live ? (av[j] + 1e-6f) * q : 0.0f;
        tot[j] += win;
#pragma unroll
        for (int n = 0; n < NLP; ++n) acc[j][n] = fmaf(win, lv[n], acc[j][n]);
      }
    }
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j) {
      const int o = grp * kLdGroup + j;
      float* out = partial + (((int64_t)b * No + min(o, No - 1)) * (NLP + 1)) * chunks + chunk;
      const float ts = wave_sum(tot[j]);
      // all NLP sums in ~2 NLP cross-lane steps: lane l ends up with column bitrev6(l)
      const float cs = wave_transpose_reduce<NLP>(acc[j], lane);
      const int n = bitrev6(lane);
      if (n < NLP && o < No) out[(int64_t)n * chunks] = cs;
      if (lane == 0 && o < No) out[(int64_t)NLP * chunks] = ts;
    }
  }
}

// Sums of the (up to 33) rows of `chunks` partials of one (batch item, object), the same order
// whatever the hardware does: thread t adds the elements t, t + 256, ... of every row (coalesced,
// all loads independent and in flight together), one transpose-reduce joins the lanes, the four
// waves' sums are added in wave order.  s_rows[0 .. ncols) receives the sums.
__device__ __forceinline__ void row_sums(const float* __restrict__ rows, int ncols, int chunks,
                                         float* s_part, float* s_rows) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  float a[kLdMaxCls], last = 0.0f;
#pragma unroll
  for (int n = 0; n < kLdMaxCls; ++n) a[n] = 0.0f;
  for (int c = threadIdx.x; c < chunks; c += kBlock) {
    // branch-free (rows past the last are re-reads of it, masked): behind a branch per row the
    // compiler waits for every load before it issues the next (21 us for this loop instead of ~5)
    float x[kLdMaxCls + 1];
#pragma unroll
    for (int n = 0; n <= kLdMaxCls; ++n) x[n] = rows[(int64_t)min(n, ncols - 1) * chunks + c];
#pragma unroll
    for (int n = 0; n < kLdMaxCls; ++n) a[n] += n < ncols ? x[n] : 0.0f;
    last += kLdMaxCls < ncols ? x[kLdMaxCls] : 0.0f;
  }
  const float cs = wave_transpose_reduce<kLdMaxCls>(a, lane);
  last = wave_sum(last);
  const int n = bitrev6(lane);
  if (n < kLdMaxCls) s_part[wave * (kLdMaxCls + 1) + n] = cs;
  if (lane == 0) s_part[wave * (kLdMaxCls + 1) + kLdMaxCls] = last;
  __syncthreads();
  if (threadIdx.x < ncols) {
    const int k = threadIdx.x;
    s_rows[k] = (s_part[k] + s_part[(kLdMaxCls + 1) + k]) + (s_part[2 * (kLdMaxCls + 1) + k] + s_part[3 * (kLdMaxCls + 1) + k]);
  }
  __syncthreads();
}

// ---- forward, pass 2: partials -> total, mean, dist.  One workgroup per (object, batch item).
__global__ __launch_bounds__(kBlock) void lyt_dist_finish_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ dist,
                                                                 float* __restrict__ mean,
                                                                 float* __restrict__ total, int No,
                                                                 int Nl, int NLP, int chunks) {
  __shared__ float s_part[kLdWaves * (kLdMaxCls + 1)];
  __shared__ float s_rows[kLdMaxCls + 1];
  const int o = blockIdx.x, b = blockIdx.y;
  row_sums(partial + (((int64_t)b * No + o) * (NLP + 1)) * chunks, NLP + 1, chunks, s_part, s_rows);
  const float t = s_rows[NLP];
  const int n = threadIdx.x;
  const float m = n < Nl ? s_rows[n] / t : 0.0f;
  __syncthreads();
  if (n < Nl) {
    s_rows[n] = m;
    mean[((int64_t)b * No + o) * Nl + n] = m;
  }
  if (n == 0) total[(int64_t)b * No + o] = t;
  __syncthreads();
  if (n < Nl) {  // every thread walks the row in the same order: one softmax, the same bits in all
    float mx = s_rows[0];
    for (int k = 1; k < Nl; ++k) mx = fmaxf(mx, s_rows[k]);
    float den = 0.0f;
    for (int k = 0; k < Nl; ++k) den += expf(s_rows[k] - mx);
    dist[((int64_t)b * No + o) * Nl + n] = expf(m - mx) / den;
  }
}

// ---- backward, pass 1: grad_dist -> coefficients of the pixel pass.  With S = mean * total:
//   g_mean = dist * (g - sum_n g dist)          (softmax)
//   coef[o][n] = dL/dS[o][n] = g_mean[o][n] / total[o]
//   coef[o][NLP] = dL/dtotal[o] = -sum_n g_mean[o][n] mean[o][n] / total[o]
__global__ __launch_bounds__(kBlock) void lyt_dist_coef_kernel(const float* __restrict__ grad_dist,
                                                               const float* __restrict__ dist,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ total,
                                                               float* __restrict__ coef, int64_t BO,
                                                               int Nl, int NLP) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= BO) return;
  const float* g = grad_dist + e * Nl;
  const float* d = dist + e * Nl;
  const float* m = mean + e * Nl;
  float dot = 0.0f;
  for (int n = 0; n < Nl; ++n) dot = fmaf(g[n], d[n], dot);
  const float inv = 1.0f / total[e];
  float gt = 0.0f;
  float* c = coef + e * (NLP + 1);
  for (int n = 0; n < NLP; ++n) {
    const float gm = n < Nl ? d[n] * (g[n] - dot) : 0.0f;
    c[n] = gm * inv;
    if (n < Nl) gt = fmaf(gm, m[n], gt);
  }
  c[NLP] = -gt * inv;
}

// ---- backward, pass 2 (pixels): grad_alpha, and the partial sums of grad_cls
//   g_win[o][x] = coef[o][NLP] + sum_n coef[o][n] lyt[n][x]
//   grad_alpha[o][x] = g_win q;   g_q = g_win (alpha + 1e-6);   grad_cls[o][n] = sum_x g_q prob[n][x]
template <int NLP>
__global__ __launch_bounds__(kBlock) void lyt_dist_bwd_kernel(
    const float* __restrict__ alpha, LytView lyt, const float* __restrict__ cls,
    const float* __restrict__ coef, float* __restrict__ grad_alpha, float* __restrict__ partial,
    int Tw, int La, int obj0, int No, int Nl, int HW, int chunks, float min_cls) {
  __shared__ float s_lyt[NLP * kBlock];
  __shared__ float s_prob[NLP * kBlock];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t npx = (int64_t)Tw * HW;
  const bool weighted = cls != nullptr;
  {
    const int64_t i = (int64_t)chunk * kLdChunk + threadIdx.x;
    stage_logits<NLP>(lyt, b, i, npx, HW, Nl, weighted, s_lyt, s_prob);
    // layers in front of the objects (the background) take no part: zero gradient
    if (i < npx) {
      const int t = (int)(i / HW), p = (int)(i - (int64_t)t * HW);
      for (int l = 0; l < obj0; ++l) grad_alpha[(((int64_t)b * Tw + t) * La + l) * HW + p] = 0.0f;
    }
  }
  __syncthreads();
  const int ngroups = (No + kLdGroup - 1) / kLdGroup;
  for (int grp = wave; grp < ngroups; grp += kLdWaves) {
    float acc[kLdGroup][NLP];
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j)
#pragma unroll
      for (int n = 0; n < NLP; ++n) acc[j][n] = 0.0f;
    for (int sl = 0; sl < kLdWaves; ++sl) {
      const int px = sl * kWave + lane;
      const int64_t i = (int64_t)chunk * kLdChunk + px;
      const bool live = i < npx;
      const int64_t ic = live ? i : npx - 1;
      const int t = (int)(ic / HW), p = (int)(ic - (int64_t)t * HW);
      const int64_t aoff = (((int64_t)b * Tw + t) * La + obj0) * HW + p;
      float av[kLdGroup];
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j)
        av[j] = weighted ? alpha[aoff + (int64_t)min(grp * kLdGroup + j, No - 1) * HW] : 0.0f;
      float lv[NLP], pr[NLP];
#pragma unroll
      for (int n = 0; n < NLP; ++n) {
        lv[n] = s_lyt[n * kBlock + px];
        pr[n] = weighted ? s_prob[n * kBlock + px] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = min(grp * kLdGroup + j, No - 1);
        const float* cf = coef + ((int64_t)b * No + o) * (NLP + 1);  // wave-uniform: scalar operands
        float gw = cf[NLP];
#pragma unroll
        for (int n = 0; n < NLP; ++n) gw = fmaf(cf[n], lv[n], gw);  // columns >= Nl of coef are 0
        const float q = weighted ? class_weight<NLP>(cls + ((int64_t)b * No + o) * Nl, pr, Nl, min_cls) : 1.0f;
        const bool on = live && grp * kLdGroup + j < No;
        if (on) grad_alpha[aoff + (int64_t)o * HW] = gw * q;
        if (weighted) {
          const float gq = on ? gw * (av[j] + 1e-6f) : 0.0f;
#pragma unroll
          for (int n = 0; n < NLP; ++n) acc[j][n] = fmaf(gq, pr[n], acc[j][n]);
        }
      }
    }
    if (weighted) {
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = grp * kLdGroup + j;
        float* out = partial + (((int64_t)b * No + min(o, No - 1)) * (NLP + 1)) * chunks + chunk;
        const float cs = wave_transpose_reduce<NLP>(acc[j], lane);
        const int n = bitrev6(lane);
        if (n < NLP && o < No) out[(int64_t)n * chunks] = cs;
      }
    }
  }
}

// ---- backward, pass 3: grad_cls = the partials of a row; one workgroup per (object, batch item)
__global__ __launch_bounds__(kBlock) void lyt_dist_gcls_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ grad_cls, int No,
                                                               int Nl, int NLP, int chunks) {
  __shared__ float s_part[kLdWaves * (kLdMaxCls + 1)];
  __shared__ float s_rows[kLdMaxCls + 1];
  const int o = blockIdx.x, b = blockIdx.y;
  row_sums(partial + (((int64_t)b * No + o) * (NLP + 1)) * chunks, Nl, chunks, s_part, s_rows);
  if (threadIdx.x < Nl) grad_cls[((int64_t)b * No + o) * Nl + threadIdx.x] = s_rows[threadIdx.x];
}

static int check_ld(const char* fn, int64_t B, int Tw, int La, int obj0, int No, int Nl, int H, int W) {
  if (B < 0 || B > 65535 || Tw < 1 || La < 1 || obj0 < 0 || No < 1 || obj0 + No != La ||
      No > kLdMaxObj || Nl < 1 || Nl > kLdMaxCls || H < 1 || W < 1 ||
      (int64_t)Tw * H * W > 2147483647ll - kLdChunk) {
    set_error("%s: bad shape B=%lld Tw=%d layers=%d first object=%d No=%d Nl=%d %dx%d (No, Nl <= 32)", fn,
              (long long)B, Tw, La, obj0, No, Nl, H, W);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

static int ld_chunks(int Tw, int H, int W) { return (int)(((int64_t)Tw * H * W + kLdChunk - 1) / kLdChunk); }

template <int NLP>
static void launch_ld_fwd(const float* alpha, LytView lv, const float* cls, float* partial, int64_t B,
                          int Tw, int La, int obj0, int No, int Nl, int HW, int chunks, float min_cls,
                          hipStream_t st) {
  hipLaunchKernelGGL(lyt_dist_partial_kernel<NLP>, dim3((unsigned)chunks, (unsigned)B), dim3(kBlock), 0,
                     st, alpha, lv, cls, partial, Tw, La, obj0, No, Nl, HW, chunks, min_cls);
}

template <int NLP>
static void launch_ld_bwd(const float* alpha, LytView lv, const float* cls, const float* coef,
                          float* grad_alpha, float* partial, int64_t B, int Tw, int La, int obj0, int No,
                          int Nl, int HW, int chunks, float min_cls, hipStream_t st) {
  hipLaunchKernelGGL(lyt_dist_bwd_kernel<NLP>, dim3((unsigned)chunks, (unsigned)B), dim3(kBlock), 0, st,
                     alpha, lv, cls, coef, grad_alpha, partial, Tw, La, obj0, No, Nl, HW, chunks, min_cls);
}

}  // namespace waldo

using namespace waldo;

#define WALDO_LD_DISPATCH(NLP, CALL) \
  switch (NLP) {                     \
    case 4: CALL(4); break;          \
    case 8: CALL(8); break;          \
    case 12: CALL(12); break;        \
    case 16: CALL(16); break;        \
    case 20: CALL(20); break;        \
    case 24: CALL(24); break;        \
    case 28: CALL(28); break;        \
    default: CALL(32); break;        \
  }

extern "C" int64_t waldo_lyt_dist_workspace_bytes(int64_t B, int Tw, int No, int Nl, int H, int W) {
  if (B < 0 || Tw < 1 || No < 1 || Nl < 1 || H < 1 || W < 1) return -1;
  const int nlp = (Nl + 3) & ~3;
  return (int64_t)sizeof(float) * B * ((int64_t)ld_chunks(Tw, H, W) + 1) * No * (nlp + 1);
}

extern "C" int waldo_lyt_dist_fwd(const float* alpha, const float* lyt, int64_t lyt_batch_stride,
                                  int64_t lyt_frame_stride, const float* cls, float min_cls,
                                  float* dist, float* mean, float* total, float* workspace, int64_t B,
                                  int Tw, int layers, int first_obj, int No, int Nl, int H, int W,
                                  waldo_stream_t stream) {
  int rc = check_ld("waldo_lyt_dist_fwd", B, Tw, layers, first_obj, No, Nl, H, W);
  if (rc) return rc;
  if (B == 0) return WALDO_OK;
  if (!alpha || !lyt || !dist || !mean || !total || !workspace) {
    set_error("waldo_lyt_dist_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nlp = (Nl + 3) & ~3, chunks = ld_chunks(Tw, H, W);
  const LytView lv{lyt, lyt_batch_stride, lyt_frame_stride};
#define CALL(N) launch_ld_fwd<N>(alpha, lv, cls, workspace, B, Tw, layers, first_obj, No, Nl, H * W, chunks, min_cls, st)
  WALDO_LD_DISPATCH(nlp, CALL)
#undef CALL
  hipLaunchKernelGGL(lyt_dist_finish_kernel, dim3((unsigned)No, (unsigned)B), dim3(kBlock), 0, st, workspace,
                     dist, mean, total, No, Nl, nlp, chunks);
  return launch_status("waldo_lyt_dist_fwd");
}

extern "C" int waldo_lyt_dist_bwd(const float* grad_dist, const float* alpha, const float* lyt,
                                  int64_t lyt_batch_stride, int64_t lyt_frame_stride,
                                  const float* cls, float min_cls, const float* dist,
                                  const float* mean, const float* total, float* grad_alpha,
                                  float* grad_cls, float* workspace, int64_t B, int Tw, int layers,
                                  int first_obj, int No, int Nl, int H, int W, waldo_stream_t stream) {
  int rc = check_ld("waldo_lyt_dist_bwd", B, Tw, layers, first_obj, No, Nl, H, W);
  if (rc) return rc;
  if (B == 0) return WALDO_OK;
  if (!grad_dist || !alpha || !lyt || !dist || !mean || !total || !grad_alpha || !workspace ||
      (cls != nullptr) != (grad_cls != nullptr)) {
    set_error("waldo_lyt_dist_bwd: null pointer (grad_cls goes with cls)");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nlp = (Nl + 3) & ~3, chunks = ld_chunks(Tw, H, W);
  const LytView lv{lyt, lyt_batch_stride, lyt_frame_stride};
  // workspace: [coef (B, No, nlp + 1)] [partial (B, chunks, No, nlp + 1)]
  float* coef = workspace;
  float* partial = workspace + B * No * (nlp + 1);
  const int64_t BO = B * No;
  hipLaunchKernelGGL(lyt_dist_coef_kernel, dim3((unsigned)((BO + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                     grad_dist, dist, mean, total, coef, BO, Nl, nlp);
#define CALL(N) launch_ld_bwd<N>(alpha, lv, cls, coef, grad_alpha, partial, B, Tw, layers, first_obj, No, Nl, H * W, chunks, min_cls, st)
  WALDO_LD_DISPATCH(nlp, CALL)
#undef CALL
  if (cls != nullptr)
    hipLaunchKernelGGL(lyt_dist_gcls_kernel, dim3((unsigned)No, (unsigned)B), dim3(kBlock), 0, st, partial,
                       grad_cls, No, Nl, nlp, chunks);
  return launch_status("waldo_lyt_dist_bwd");
}
