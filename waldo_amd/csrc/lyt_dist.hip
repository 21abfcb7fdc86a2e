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
// Here: one pass over the pixels per direction.  A wave owns a group of four objects and 64 pixels
// per trip; a lane keeps the 4 x Nl running sums of its pixels in registers (the class logits of a
// pixel are loaded once per wave as Nl coalesced rows), the lanes are combined by a fixed butterfly,
// and the workgroups' partial sums are added in workgroup order by a second, tiny kernel -- no
// atomics, bitwise reproducible.  The backward needs the same reduction shape for grad_cls and is
// elementwise for grad_alpha; the layout logits are data (no gradient: callers whose layout requires
// one use the framework expression).
#include <math.h>

#include "waldo_common.hip.h"

namespace waldo {

constexpr int kLdGroup = 4;        // objects per wave
constexpr int kLdWaves = kBlock / kWave;
constexpr int kLdTrips = 16;       // pixel trips per workgroup: 1024 pixel-frames per workgroup
constexpr int kLdChunk = kLdTrips * kWave;
constexpr int kLdMaxObj = 32, kLdMaxCls = 32;

struct LytView {
  const float* base;
  int64_t batch_stride, frame_stride;  // elements; planes are HW apart, rows contiguous
};

// softmax of the NLP logits in v (entries >= Nl masked), in place -> probabilities
template <int NLP>
__device__ __forceinline__ void softmax_regs(float (&v)[NLP], int Nl) {
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
  for (int n = 0; n < NLP; ++n) v[n] *= inv;
}

// ---- forward, pass 1: partial sums of win * lyt and of win per (workgroup, object)
// partial: (B, chunks, No, NLP + 1), the last column is the total
template <int NLP>
__global__ __launch_bounds__(kBlock) void lyt_dist_partial_kernel(
    const float* __restrict__ alpha, LytView lyt, const float* __restrict__ cls,
    float* __restrict__ partial, int Tw, int La, int obj0, int No, int Nl, int HW, int chunks,
    float min_cls) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t npx = (int64_t)Tw * HW;
  const int ngroups = (No + kLdGroup - 1) / kLdGroup;
  for (int grp = wave; grp < ngroups; grp += kLdWaves) {
    float acc[kLdGroup][NLP], tot[kLdGroup];
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j) {
      tot[j] = 0.0f;
#pragma unroll
      for (int n = 0; n < NLP; ++n) acc[j][n] = 0.0f;
    }
    for (int it = 0; it < kLdTrips; ++it) {
      const int64_t i = (int64_t)chunk * kLdChunk + it * kWave + lane;
      const bool live = i < npx;
      const int64_t ic = live ? i : npx - 1;
      const int t = (int)(ic / HW), p = (int)(ic - (int64_t)t * HW);
      const float* lp = lyt.base + b * lyt.batch_stride + t * lyt.frame_stride + p;
      float lv[NLP], pr[NLP];
#pragma unroll
      for (int n = 0; n < NLP; ++n) {
        lv[n] = lp[(int64_t)min(n, Nl - 1) * HW];
        pr[n] = lv[n];
      }
      if (cls != nullptr) softmax_regs<NLP>(pr, Nl);
      const float* ap = alpha + (((int64_t)b * Tw + t) * La + obj0) * HW + p;
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = min(grp * kLdGroup + j, No - 1);
        float q = 1.0f;
        if (cls != nullptr) {
          const float* c = cls + ((int64_t)b * No + o) * Nl;  // wave-uniform: scalar loads
          q = 0.0f;
#pragma unroll
          for (int n = 0; n < NLP; ++n) q = fmaf(n < Nl ? c[n] + min_cls : 0.0f, pr[n], q);
        }
        const float win = live ? (ap[(int64_t)o * HW] + 1e-6f) * q : 0.0f;
        tot[j] += win;
#pragma unroll
        for (int n = 0; n < NLP; ++n) acc[j][n] = fmaf(win, lv[n], acc[j][n]);
      }
    }
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j) {
      const int o = grp * kLdGroup + j;
      float* out = partial + (((int64_t)b * chunks + chunk) * No + min(o, No - 1)) * (NLP + 1);
      const float ts = wave_sum(tot[j]);
#pragma unroll
      for (int n = 0; n < NLP; ++n) {
        const float s = wave_sum(acc[j][n]);
        if (lane == 0 && o < No) out[n] = s;
      }
      if (lane == 0 && o < No) out[NLP] = ts;
    }
  }
}

// ---- forward, pass 2: partial sums in workgroup order -> total, mean, dist.  One workgroup per
// batch item, one thread per (object, class) pair at a time.
__global__ __launch_bounds__(kBlock) void lyt_dist_finish_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ dist,
                                                                 float* __restrict__ mean,
                                                                 float* __restrict__ total, int No,
                                                                 int Nl, int NLP, int chunks) {
  __shared__ float s[kLdMaxObj * (kLdMaxCls + 1)];
  const int b = blockIdx.x, pitch = NLP + 1;
  for (int e = threadIdx.x; e < No * pitch; e += kBlock) {
    const float* p = partial + (int64_t)b * chunks * No * pitch + e;
    float a = 0.0f;
    for (int c = 0; c < chunks; ++c) a += p[(int64_t)c * No * pitch];
    s[e] = a;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < No; o += kBlock) {
    const float t = s[o * pitch + NLP];
    float mx = -__builtin_huge_valf();
    for (int n = 0; n < Nl; ++n) {
      const float m = s[o * pitch + n] / t;
      mean[((int64_t)b * No + o) * Nl + n] = m;
      s[o * pitch + n] = m;
      mx = fmaxf(mx, m);
    }
    float den = 0.0f;
    for (int n = 0; n < Nl; ++n) den += expf(s[o * pitch + n] - mx);
    for (int n = 0; n < Nl; ++n) dist[((int64_t)b * No + o) * Nl + n] = expf(s[o * pitch + n] - mx) / den;
    total[(int64_t)b * No + o] = t;
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
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t npx = (int64_t)Tw * HW;
  // layers in front of the objects (the background) take no part: zero gradient
  for (int l = 0; l < obj0; ++l)
    for (int k = threadIdx.x; k < kLdChunk; k += kBlock) {
      const int64_t i = (int64_t)chunk * kLdChunk + k;
      if (i < npx) {
        const int t = (int)(i / HW), p = (int)(i - (int64_t)t * HW);
        grad_alpha[(((int64_t)b * Tw + t) * La + l) * HW + p] = 0.0f;
      }
    }
  const int ngroups = (No + kLdGroup - 1) / kLdGroup;
  for (int grp = wave; grp < ngroups; grp += kLdWaves) {
    float acc[kLdGroup][NLP];
#pragma unroll
    for (int j = 0; j < kLdGroup; ++j)
#pragma unroll
      for (int n = 0; n < NLP; ++n) acc[j][n] = 0.0f;
    for (int it = 0; it < kLdTrips; ++it) {
      const int64_t i = (int64_t)chunk * kLdChunk + it * kWave + lane;
      const bool live = i < npx;
      const int64_t ic = live ? i : npx - 1;
      const int t = (int)(ic / HW), p = (int)(ic - (int64_t)t * HW);
      const float* lp = lyt.base + b * lyt.batch_stride + t * lyt.frame_stride + p;
      float lv[NLP], pr[NLP];
#pragma unroll
      for (int n = 0; n < NLP; ++n) {
        lv[n] = lp[(int64_t)min(n, Nl - 1) * HW];
        pr[n] = lv[n];
      }
      if (cls != nullptr) softmax_regs<NLP>(pr, Nl);
      const int64_t aoff = (((int64_t)b * Tw + t) * La + obj0) * HW + p;
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = min(grp * kLdGroup + j, No - 1);
        const float* cf = coef + ((int64_t)b * No + o) * (NLP + 1);  // wave-uniform
        float gw = cf[NLP];
#pragma unroll
        for (int n = 0; n < NLP; ++n) gw = fmaf(n < Nl ? cf[n] : 0.0f, lv[n], gw);
        float q = 1.0f;
        if (cls != nullptr) {
          const float* c = cls + ((int64_t)b * No + o) * Nl;
          q = 0.0f;
#pragma unroll
          for (int n = 0; n < NLP; ++n) q = fmaf(n < Nl ? c[n] + min_cls : 0.0f, pr[n], q);
        }
        const bool on = live && grp * kLdGroup + j < No;
        if (on) grad_alpha[aoff + (int64_t)o * HW] = gw * q;
        if (cls != nullptr) {
          const float gq = on ? gw * (alpha[aoff + (int64_t)o * HW] + 1e-6f) : 0.0f;
#pragma unroll
          for (int n = 0; n < NLP; ++n) acc[j][n] = fmaf(gq, pr[n], acc[j][n]);
        }
      }
    }
    if (cls != nullptr) {
#pragma unroll
      for (int j = 0; j < kLdGroup; ++j) {
        const int o = grp * kLdGroup + j;
        float* out = partial + (((int64_t)b * chunks + chunk) * No + min(o, No - 1)) * (NLP + 1);
#pragma unroll
        for (int n = 0; n < NLP; ++n) {
          const float s = wave_sum(acc[j][n]);
          if (lane == 0 && o < No) out[n] = s;
        }
      }
    }
  }
}

// ---- backward, pass 3: grad_cls = partial sums in workgroup order
__global__ __launch_bounds__(kBlock) void lyt_dist_gcls_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ grad_cls, int64_t B,
                                                               int No, int Nl, int NLP, int chunks) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= B * No * Nl) return;
  const int n = (int)(e % Nl), o = (int)((e / Nl) % No);
  const int64_t b = e / ((int64_t)Nl * No);
  const int pitch = NLP + 1;
  const float* p = partial + (b * chunks * No + o) * pitch + n;
  float a = 0.0f;
  for (int c = 0; c < chunks; ++c) a += p[(int64_t)c * No * pitch];
  grad_cls[e] = a;
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
  hipLaunchKernelGGL(lyt_dist_finish_kernel, dim3((unsigned)B), dim3(kBlock), 0, st, workspace, dist, mean,
                     total, No, Nl, nlp, chunks);
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
  if (cls != nullptr) {
    const int64_t n = B * No * Nl;
    hipLaunchKernelGGL(lyt_dist_gcls_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       partial, grad_cls, B, No, Nl, nlp, chunks);
  }
  return launch_status("waldo_lyt_dist_bwd");
}
