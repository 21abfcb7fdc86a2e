// A8: gather_time (models/nets/lvd.py:462-467) and the frame arithmetic the flow synthesis builds on it
// (lvd.py:660-668, 780-787): for the grids of a clip, x (B, T, P, 2),
//
//     out[b, tc, tp] = x[b, ctx_ts[b, tc, tp]]  -  x[b, pred_ts[tp]]          (the layer-space flow)
//     out[b, tc, tp] = x[b, pred_ts[tp]]                                      (the predicted frames' grids,
//                                                                              expanded over the contexts)
//
// optionally written channel-first ((…, P, 2) -> (…, N, 2, HW), the permute + reshape of lvd.py:662-664).
// The reference spells this as gather + advanced indexing + subtraction + permute + reshape: at the LVD
// recipe's step that was 12 framework launches forward and -- advanced indexing differentiates through a
// sort -- more than 20 backward, a fifth of the step's time in the framework's elementwise kernels.  Here
// it is one launch each way.  The backward is a gather: a thread owns one (b, t, pair) of grad_x and sums
// the Tc * Tp output frames that read it (frame tests are workgroup-uniform); no atomics, no zero fill,
// the same bits every run.
#include "waldo_common.hip.h"

namespace waldo {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// out frame m = (b, tc, tp); P pairs per frame; HW > 0: pair q = n * HW + hw goes to planes (n, c) at hw
__global__ __launch_bounds__(kBlock) void time_gather_fwd_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ ctx_ts, const int64_t* __restrict__ pred_ts,
    float* __restrict__ out, int* __restrict__ status, int T, int Tc, int Tp, int64_t P, int64_t HW, int subtract,
    int blocks_per_frame) {
  const int m = blockIdx.x / blocks_per_frame;
  const int tp = m % Tp, b = m / (Tc * Tp);
  const int tpr = checked_frame(pred_ts, tp, T, status, kStatusPred);
  const int ta = ctx_ts != nullptr ? checked_frame(ctx_ts, m, T, status, kStatusCtx) : tpr;
  const f32x2_t* xa = reinterpret_cast<const f32x2_t*>(x) + ((int64_t)b * T + ta) * P;
  const f32x2_t* xs = reinterpret_cast<const f32x2_t*>(x) + ((int64_t)b * T + tpr) * P;
  for (int64_t q = (int64_t)(blockIdx.x % blocks_per_frame) * kBlock + threadIdx.x; q < P;
       q += (int64_t)blocks_per_frame * kBlock) {
    f32x2_t v = xa[q];
    if (subtract) v = v - xs[q];
    if (HW > 0) {
      const int64_t n = q / HW, hw = q - n * HW;
      float* o = out + ((int64_t)m * P + n * HW) * 2 + hw;
      o[0] = v[0];
      o[HW] = v[1];
    } else {
      reinterpret_cast<f32x2_t*>(out)[(int64_t)m * P + q] = v;
    }
  }
}

// grad_x frame u = (b, t)
__global__ __launch_bounds__(kBlock) void time_gather_bwd_kernel(
    const float* __restrict__ g_out, const int64_t* __restrict__ ctx_ts, const int64_t* __restrict__ pred_ts,
    float* __restrict__ g_x, int T, int Tc, int Tp, int64_t P, int64_t HW, int subtract, int blocks_per_frame) {
  const int u = blockIdx.x / blocks_per_frame;
  const int t = u % T, b = u / T;
  for (int64_t q = (int64_t)(blockIdx.x % blocks_per_frame) * kBlock + threadIdx.x; q < P;
       q += (int64_t)blocks_per_frame * kBlock) {
    const int64_t n = HW > 0 ? q / HW : 0, hw = HW > 0 ? q - n * HW : 0;
    f32x2_t acc = {0.0f, 0.0f};
    for (int tc = 0; tc < Tc; ++tc)
      for (int tp = 0; tp < Tp; ++tp) {
        const int m = (b * Tc + tc) * Tp + tp;
        const int tpr = (int)min(max(pred_ts[tp], (int64_t)0), (int64_t)(T - 1));
        const int ta = ctx_ts != nullptr ? (int)min(max(ctx_ts[m], (int64_t)0), (int64_t)(T - 1)) : tpr;
        const int sign = (ta == t ? 1 : 0) - ((subtract && tpr == t) ? 1 : 0);  // uniform
        if (sign == 0) continue;
        f32x2_t g;
        if (HW > 0) {
          const float* o = g_out + ((int64_t)m * P + n * HW) * 2 + hw;
          g[0] = o[0];
          g[1] = o[HW];
        } else {
          g = reinterpret_cast<const f32x2_t*>(g_out)[(int64_t)m * P + q];
        }
        acc = sign > 0 ? acc + g : acc - g;
      }
    reinterpret_cast<f32x2_t*>(g_x)[(int64_t)u * P + q] = acc;
  }
}

// A7: scale(input[:, :Tw, c0:], 1 / S) of Warper.grid_to_flow[_ctx] (lvd.py:611, 716 with scale() of lvd.py:175-179):
// the low-resolution copy of the layout channels the class-distribution filter reads.  F.interpolate(bilinear,
// align_corners=False) by 1 / S, S a power of two: the source position of a pixel is S dst + S / 2 - 1/2, i.e.
// the mean of the 2 x 2 texels in the middle of its S x S block, 0.5 (0.5 a + 0.5 b) + 0.5 (0.5 c + 0.5 d) in
// F.interpolate's association (halving is exact: the same bits whether or not its products are fused).  The
// framework spelled it as a contiguous copy of the channel slice (the slice is strided) and the interpolation
// over that copy: three passes over the 20 full-resolution layout planes of every frame; this reads the two
// middle rows of every block once.
// (one workgroup = 256 consecutive output pixels of ONE plane: the plane's (b, t, c) come from the block index with
// 32-bit arithmetic -- as one flat 64-bit index per thread the five divisions by run-time values made it 184 vector
// instructions per pixel, VALU issue 100 %: a copy kernel bound by integer division)
__global__ __launch_bounds__(kBlock) void downscale_frames_kernel(const float* __restrict__ input,
                                                                  float* __restrict__ out, int T, int Tw, int C,
                                                                  int c0, int H, int W, int S, int tiles) {
  typedef float f32x2_d __attribute__((ext_vector_type(2)));
  const unsigned pl = blockIdx.x / (unsigned)tiles;  // (b, t, c) of the output
  const unsigned e = (blockIdx.x - pl * (unsigned)tiles) * kBlock + threadIdx.x;
  if (e >= (unsigned)(H * W)) return;
  const unsigned y = e / (unsigned)W, x = e - y * (unsigned)W;
  const unsigned Cs = (unsigned)(C - c0);
  const unsigned c = pl % Cs, bt = pl / Cs;
  const unsigned t = bt % (unsigned)Tw;
  const int64_t b = bt / (unsigned)Tw;
  const int64_t Wd = (int64_t)W * S, Hd = (int64_t)H * S;
  const float* src = input + (((b * T + t) * C + c0 + c) * Hd + ((int64_t)y * S + S / 2 - 1)) * Wd + (int64_t)x * S + S / 2 - 1;
  const f32x2_d r0 = *reinterpret_cast<const f32x2_d*>(src);
  const f32x2_d r1 = *reinterpret_cast<const f32x2_d*>(src + Wd);
  out[(int64_t)pl * H * W + e] = 0.5f * (0.5f * r0[0] + 0.5f * r0[1]) + 0.5f * (0.5f * r1[0] + 0.5f * r1[1]);
}

static int check_time_gather(const char* fn, const void* x, const void* pred_ts, const void* out, int B, int T,
                             int Tc, int Tp, int64_t P, int64_t HW, int subtract, const void* ctx_ts) {
  if (B < 0 || T < 1 || Tc < 0 || Tp < 0 || P < 0 || HW < 0 || (HW > 0 && P % HW != 0) ||
      (int64_t)B * Tc * Tp > 0x7fffffff / 64 || (int64_t)B * T > 0x7fffffff / 64) {
    set_error("%s: bad shape B=%d T=%d Tc=%d Tp=%d P=%lld HW=%lld", fn, B, T, Tc, Tp, (long long)P, (long long)HW);
    return WALDO_EINVAL;
  }
  if (subtract && ctx_ts == nullptr) {
    set_error("%s: the difference needs ctx_ts", fn);
    return WALDO_EINVAL;
  }
  if ((int64_t)B * Tc * Tp * P == 0) return -1;  // nothing to do
  if (!x || !pred_ts || !out) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

static int frame_blocks(int64_t frames, int64_t P) {
  // enough workgroups to fill the chip, at most one per kBlock pairs
  const int64_t per = (P + kBlock - 1) / kBlock;
  const int64_t want = (4096 + frames - 1) / max(frames, (int64_t)1);
  return (int)max((int64_t)1, min(per, want));
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_time_gather_fwd(const float* x, const int64_t* ctx_ts, const int64_t* pred_ts, float* out,
                                     int* status, int B, int T, int Tc, int Tp, int64_t P, int64_t HW, int subtract,
                                     waldo_stream_t stream) {
  const int rc = check_time_gather("waldo_time_gather_fwd", x, pred_ts, out, B, T, Tc, Tp, P, HW, subtract, ctx_ts);
  if (rc) return rc < 0 ? WALDO_OK : rc;
  const int64_t frames = (int64_t)B * Tc * Tp;
  const int bpf = frame_blocks(frames, P);
  time_gather_fwd_kernel<<<dim3((unsigned)(frames * bpf)), dim3(kBlock), 0, (hipStream_t)stream>>>(
      x, ctx_ts, pred_ts, out, status, T, Tc, Tp, P, HW, subtract, bpf);
  return launch_status("waldo_time_gather_fwd");
}

extern "C" int waldo_time_gather_bwd(const float* grad_out, const int64_t* ctx_ts, const int64_t* pred_ts,
                                     float* grad_x, int B, int T, int Tc, int Tp, int64_t P, int64_t HW,
                                     int subtract, waldo_stream_t stream) {
  if (B < 0 || T < 1 || P < 0) {
    set_error("waldo_time_gather_bwd: bad shape B=%d T=%d P=%lld", B, T, (long long)P);
    return WALDO_EINVAL;
  }
  if ((int64_t)B * T * P == 0) return WALDO_OK;
  if ((int64_t)Tc * Tp == 0) {  // no output frame reads x: the gradient is zero
    if (!grad_x) {
      set_error("waldo_time_gather_bwd: null pointer");
      return WALDO_EINVAL;
    }
    fill_words(grad_x, 0u, (size_t)B * T * P * 2 * sizeof(float), (hipStream_t)stream);
    return launch_status("waldo_time_gather_bwd");
  }
  const int rc = check_time_gather("waldo_time_gather_bwd", grad_out, pred_ts, grad_x, B, T, Tc, Tp, P, HW, subtract,
                                   ctx_ts);
  if (rc) return rc < 0 ? WALDO_OK : rc;
  const int64_t frames = (int64_t)B * T;
  const int bpf = frame_blocks(frames, P);
  time_gather_bwd_kernel<<<dim3((unsigned)(frames * bpf)), dim3(kBlock), 0, (hipStream_t)stream>>>(
      grad_out, ctx_ts, pred_ts, grad_x, T, Tc, Tp, P, HW, subtract, bpf);
  return launch_status("waldo_time_gather_bwd");
}

extern "C" int waldo_downscale_frames_fwd(const float* input, float* out, int B, int T, int Tw, int C, int c0, int H,
                                          int W, int S, waldo_stream_t stream) {
  if (B < 0 || T < 1 || Tw < 0 || Tw > T || C < 1 || c0 < 0 || c0 >= C || H < 1 || W < 1 || S < 2 || (S & (S - 1)) ||
      (int64_t)H * S > 32767 || (int64_t)W * S > 32767) {
    set_error("waldo_downscale_frames_fwd: bad shape B=%d T=%d Tw=%d C=%d c0=%d H=%d W=%d S=%d (S: a power of two >= 2)",
              B, T, Tw, C, c0, H, W, S);
    return WALDO_EINVAL;
  }
  const int64_t total = (int64_t)B * Tw * (C - c0) * H * W;
  if (total == 0) return WALDO_OK;
  if (!input || !out) {
    set_error("waldo_downscale_frames_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int64_t tiles = ((int64_t)H * W + kBlock - 1) / kBlock, planes = (int64_t)B * Tw * (C - c0);
  if (planes * tiles > 2147483647 || (int64_t)H * W > 2147483647) {
    set_error("waldo_downscale_frames_fwd: problem too large for one launch");
    return WALDO_EINVAL;
  }
  downscale_frames_kernel<<<dim3((unsigned)(planes * tiles)), dim3(kBlock), 0, (hipStream_t)stream>>>(
      input, out, T, Tw, C, c0, H, W, S, (int)tiles);
  return launch_status("waldo_downscale_frames_fwd");
}
