// A12: fusion epilogue of WIF.forward with ii_score (models/nets/wif.py:49-54):
//   beta = net[:, :, :, 0:3];  w = softmax over Tc of net[:, :, :, 3];
//   a = sigmoid(vid[:, :, :, 4] + 5)   (INPUT channel 4, wif.py:53 -- not a network output; 0 if !ab)
//   out[b,t,c] = sum_tc (a * vid[b,t,tc,c] + beta[b,t,tc,c]) * w[b,t,tc]
// vid (B*T, Tc, C, HW) is the UNet input after the permute of wif.py:39, net (B*T, Tc, Co, HW) its
// output (Co >= 4).  The reference runs ~8 elementwise/softmax/reduce launches over (B,T,Tc,.,H,W)
// temporaries; here one thread owns one pixel of one (b,t), streams the Tc planes twice (max, then
// sum) and writes 3 values.  Pure HBM streaming: reads (5 + 4) * Tc floats, writes 3 per pixel.
#include "waldo_common.hip.h"

namespace waldo {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(kBlock) void wif_fuse_fwd_kernel(const float* __restrict__ vid,
                                                              const float* __restrict__ net,
                                                              float* __restrict__ out, int Tc, int C,
                                                              int Co, int64_t HW, int tiles, int ab) {
  const int64_t n = blockIdx.x / tiles;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float* v = vid + n * Tc * C * HW + p;
  const float* o = net + n * Tc * Co * HW + p;
  float m = -INFINITY;
  for (int t = 0; t < Tc; ++t) m = fmaxf(m, o[((int64_t)t * Co + 3) * HW]);
  float den = 0.0f, acc[3] = {0.0f, 0.0f, 0.0f};
  for (int t = 0; t < Tc; ++t) {
    const float e = expf(o[((int64_t)t * Co + 3) * HW] - m);
    const float a = ab ? sigmoidf(v[((int64_t)t * C + 4) * HW] + 5.0f) : 0.0f;
    den += e;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      acc[c] = fmaf(fmaf(a, v[((int64_t)t * C + c) * HW], o[((int64_t)t * Co + c) * HW]), e, acc[c]);
  }
  const float r = 1.0f / den;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(n * 3 + c) * HW + p] = acc[c] * r;
}

// grad_vid / grad_net are OVERWRITTEN on channels (0,1,2,4) / (0,1,2,3) and zero elsewhere must
// be provided by the caller (the launcher memsets both first).
__global__ __launch_bounds__(kBlock) void wif_fuse_bwd_kernel(
    const float* __restrict__ vid, const float* __restrict__ net, const float* __restrict__ out,
    const float* __restrict__ gout, float* __restrict__ gvid, float* __restrict__ gnet, int Tc, int C,
    int Co, int64_t HW, int tiles, int ab) {
  const int64_t n = blockIdx.x / tiles;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float* v = vid + n * Tc * C * HW + p;
  const float* o = net + n * Tc * Co * HW + p;
  float* gv = gvid ? gvid + n * Tc * C * HW + p : nullptr;
  float* go = gnet ? gnet + n * Tc * Co * HW + p : nullptr;
  float g[3], y[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    g[c] = gout[(n * 3 + c) * HW + p];
    y[c] = out[(n * 3 + c) * HW + p];
  }
  float m = -INFINITY;
  for (int t = 0; t < Tc; ++t) m = fmaxf(m, o[((int64_t)t * Co + 3) * HW]);
  float den = 0.0f;
  for (int t = 0; t < Tc; ++t) den += expf(o[((int64_t)t * Co + 3) * HW] - m);
  const float r = 1.0f / den;
  const float gy = g[0] * y[0] + g[1] * y[1] + g[2] * y[2];
  for (int t = 0; t < Tc; ++t) {
    const float w = expf(o[((int64_t)t * Co + 3) * HW] - m) * r;
    const float a = ab ? sigmoidf(v[((int64_t)t * C + 4) * HW] + 5.0f) : 0.0f;
    float gdot = 0.0f, ga = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = v[((int64_t)t * C + c) * HW];
      const float u = fmaf(a, x, o[((int64_t)t * Co + c) * HW]);  // a*x + beta
      gdot = fmaf(g[c], u, gdot);
      ga = fmaf(g[c] * w, x, ga);
      if (gv) gv[((int64_t)t * C + c) * HW] = g[c] * w * a;
      if (go) go[((int64_t)t * Co + c) * HW] = g[c] * w;
    }
    // softmax: d y / d s_t = w_t (u_t . g - y . g)
    if (go) go[((int64_t)t * Co + 3) * HW] = w * (gdot - gy);
    if (gv) gv[((int64_t)t * C + 4) * HW] = ab ? ga * a * (1.0f - a) : 0.0f;
  }
}

static int check_wif(const char* fn, int64_t N, int Tc, int C, int Co, int64_t HW) {
  if (N < 0 || Tc < 1 || C < 5 || Co < 4 || HW < 1 || N * ((HW + kBlock - 1) / kBlock) > 2147483647) {
    set_error("%s: bad shape N=%lld Tc=%d C=%d Co=%d HW=%lld (need C >= 5, Co >= 4)", fn,
              (long long)N, Tc, C, Co, (long long)HW);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_wif_fuse_fwd(const float* vid, const float* net, float* out, int64_t N, int Tc,
                                  int C, int Co, int64_t HW, int ab, waldo_stream_t stream) {
  int rc = check_wif("waldo_wif_fuse_fwd", N, Tc, C, Co, HW);
  if (rc) return rc;
  if (N == 0) return WALDO_OK;
  if (!vid || !net || !out) {
    set_error("waldo_wif_fuse_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(wif_fuse_fwd_kernel, dim3((unsigned)(N * tiles)), dim3(kBlock), 0,
                     (hipStream_t)stream, vid, net, out, Tc, C, Co, HW, tiles, ab);
  return launch_status("waldo_wif_fuse_fwd");
}

extern "C" int waldo_wif_fuse_bwd(const float* vid, const float* net, const float* out,
                                  const float* grad_out, float* grad_vid, float* grad_net, int64_t N,
                                  int Tc, int C, int Co, int64_t HW, int ab, waldo_stream_t stream) {
  int rc = check_wif("waldo_wif_fuse_bwd", N, Tc, C, Co, HW);
  if (rc) return rc;
  if (N == 0) return WALDO_OK;
  if (!vid || !net || !out || !grad_out) {
    set_error("waldo_wif_fuse_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (grad_vid) fill_words(grad_vid, 0u, sizeof(float) * (size_t)(N * Tc * C * HW), st);
  if (grad_net) fill_words(grad_net, 0u, sizeof(float) * (size_t)(N * Tc * Co * HW), st);
  const int tiles = (int)((HW + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(wif_fuse_bwd_kernel, dim3((unsigned)(N * tiles)), dim3(kBlock), 0, st, vid, net,
                     out, grad_out, grad_vid, grad_net, Tc, C, Co, HW, tiles, ab);
  return launch_status("waldo_wif_fuse_bwd");
}
