// f3: the per-frame propagation step of WIF.inpaint (models/nets/wif.py:179-214) -- the inpainted reference
// background warped into a predicted frame along the background flow, objects that enter through the image border pasted
// over it, the shadow mask applied, the frame's holes filled from it, and the inputs of the external inpainter prepared
// -- as ONE launch per frame instead of ~45 framework launches over full-resolution planes (three grid_sample calls by
// the same grid, two per entering object, and the mask algebra `1 - (1 - a) * (1 - b)` spelled as rsub / mul / rsub):
//
//     g        = flow + identity                                   (the sampling grid of wif.py:65)
//     w_img    = sample(ref_img, g);   w_mask = sample(ref_mask, g) > 0.9
//     for every entering object (region, look, flow_k):            wif.py:188-194
//         w_region = sample(region, flow_k + identity) > 0.9;   w_look = sample(look, flow_k + identity)
//         w_mask = 1 - (1 - w_mask) (1 - w_region);  todo = 1 - (1 - todo) (1 - w_region)
//         w_img  = (1 - w_region) w_img + w_region w_look
//     w_shadow = sample(shadow, g)  [> 0.9 unless soft];  todo = todo (1 - w_shadow (1 - obj))        wif.py:196-201
//     take = todo w_mask;  img = take w_img + (1 - take) img;  todo = (1 - take) todo                 wif.py:202-205
//     keep = (1 - todo) (1 - obj):  the inpainter gets (keep img, 1 - keep)                           wif.py:210-211
//     (fix_mask: it gets img and 1 - (1 - todo) (1 - obj), which the caller dilates)                   wif.py:207-208
//
// Every product, difference and comparison is taken in the order the framework's elementwise kernels take them (the
// library is built with -ffp-contract=off) and the samples come from the device function waldo_grid_sample2d_fwd uses,
// so the results have the BITS of the composition they replace (tests/test_inpaint.py::
// test_fused_propagation_has_the_bits_of_the_spelled_out_loop).  After the inpainter returns, the frame is
// (1 - todo) img + todo fill (wif.py:214): waldo_inpaint_blend_fwd.
#include "waldo_common.hip.h"

namespace waldo {

struct EnterArg {
  const float* region;  // (B, HW)
  const float* look;    // (B, 3, HW)
  const float* flow;    // (B, HW, 2)
};

constexpr float kMaskThresh = 0.9f;  // `1 - mask_thresh` of wif.py:65, as the float32 scalar the comparison is made with

__global__ __launch_bounds__(kBlock) void inpaint_propagate_kernel(
    const float* __restrict__ flow, const float* __restrict__ ident, const float* __restrict__ ref_img,
    const float* __restrict__ ref_mask, const float* __restrict__ shadow, EnterArg e0, EnterArg e1, int n_enter,
    const float* __restrict__ img, const float* __restrict__ todo_in, const float* __restrict__ obj,
    float* __restrict__ img_out, float* __restrict__ todo_out, float* __restrict__ inp_img,
    float* __restrict__ inp_mask, int H, int W, int soft_shadow, int fix_mask, int tiles) {
  const int64_t b = blockIdx.x / tiles;
  const int64_t HW = (int64_t)H * W;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float idx = ident[2 * p], idy = ident[2 * p + 1];
  const float* fl = flow + (b * HW + p) * 2;
  const Taps t = make_taps(fl[0] + idx, fl[1] + idy, H, W);
  float w_img[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) w_img[c] = tap_sample(ref_img + (b * 3 + c) * HW, t);
  float w_mask = tap_sample(ref_mask + b * HW, t) > kMaskThresh ? 1.0f : 0.0f;
  float todo = todo_in[b * HW + p];
  for (int k = 0; k < n_enter; ++k) {
    const EnterArg e = k == 0 ? e0 : e1;
    const float* fk = e.flow + (b * HW + p) * 2;
    const Taps tk = make_taps(fk[0] + idx, fk[1] + idy, H, W);
    const float w_region = tap_sample(e.region + b * HW, tk) > kMaskThresh ? 1.0f : 0.0f;
    const float not_region = 1.0f - w_region;
    w_mask = 1.0f - (1.0f - w_mask) * not_region;
    todo = 1.0f - (1.0f - todo) * not_region;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w_look = tap_sample(e.look + (b * 3 + c) * HW, tk);
      w_img[c] = not_region * w_img[c] + w_region * w_look;
    }
  }
  const float o = obj[b * HW + p];
  if (shadow != nullptr) {
    float w_shadow = tap_sample(shadow + b * HW, t);
    if (!soft_shadow) w_shadow = w_shadow > kMaskThresh ? 1.0f : 0.0f;
    todo = todo * (1.0f - w_shadow * (1.0f - o));
  }
  const float take = todo * w_mask;
  const float not_take = 1.0f - take;
  float im[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    im[c] = take * w_img[c] + not_take * img[(b * 3 + c) * HW + p];
    img_out[(b * 3 + c) * HW + p] = im[c];
  }
  todo = not_take * todo;
  todo_out[b * HW + p] = todo;
  const float keep = (1.0f - todo) * (1.0f - o);
  inp_mask[b * HW + p] = 1.0f - keep;
  if (!fix_mask) {
#pragma unroll
    for (int c = 0; c < 3; ++c) inp_img[(b * 3 + c) * HW + p] = keep * im[c];
  }
}

// The hole and object masks of the predicted frames (wif.py:60-75):
//     cover = sum_l (alpha_ctx + 1) / 2,  obj = the same over the object layers l >= 1        (per context tc)
//     the last context's, or the maximum over the contexts;  mask = (1 - cover) > thr,  obj_mask = obj > 0.9
// The framework reads the (B, Tc, Tp, L, H, W) tensor -- 1 GB at the Cityscapes recipe with ten predicted frames --
// five times (two shifted copies, two sums, the slice): 1.9 ms of an 8 ms call.  Here every plane is read once.  The
// sums are taken as the framework's reduction takes them over a short strided dimension -- element j into accumulator
// j % 4, the four combined in order ((a0 + a1) + a2) + a3 (tools_dev/sum_order_probe.py: 100 % of 1.5 M sums equal,
// 47 % for a sequential sum) -- because a thresholded sum is a mask pixel.
struct CtxStrides {
  int64_t b, tc, tp, l;  // element strides of alpha_ctx (B, Tc, Tp, L, H, W); the (H, W) planes are contiguous
};

__global__ __launch_bounds__(kBlock) void inpaint_holes_kernel(const float* __restrict__ actx, CtxStrides st,
                                                               float* __restrict__ mask, float* __restrict__ obj_mask,
                                                               int Tc, int Tp, int L, int64_t HW, int last_only,
                                                               float thresh, int tiles) {
  const int64_t u = blockIdx.x / tiles;  // (b, tp)
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const int64_t b = u / Tp, tp = u - b * Tp;
  const float* base = actx + b * st.b + tp * st.tp + p;
  float cover = 0.0f, obj = 0.0f;
  for (int tc = last_only ? Tc - 1 : 0; tc < Tc; ++tc) {
    const float* a = base + tc * st.tc;
    float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f}, oa[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int l = 0; l < L; ++l) {
      const float h = (a[l * st.l] + 1.0f) / 2.0f;
      ca[l & 3] += h;
      if (l > 0) oa[(l - 1) & 3] += h;
    }
    const float c = ((ca[0] + ca[1]) + ca[2]) + ca[3];
    const float o = ((oa[0] + oa[1]) + oa[2]) + oa[3];
    const bool first = last_only || tc == 0;
    cover = (first || c > cover || c != c) ? c : cover;  // torch.max: a NaN wins
    obj = (first || o > obj || o != o) ? o : obj;
  }
  mask[u * HW + p] = (1.0f - cover) > thresh ? 1.0f : 0.0f;
  obj_mask[u * HW + p] = obj > 0.9f ? 1.0f : 0.0f;
}

// frames[t] = (1 - todo) img + todo fill                                                    wif.py:214
__global__ __launch_bounds__(kBlock) void inpaint_blend_kernel(const float* __restrict__ img, const float* __restrict__ todo,
                                                               const float* __restrict__ fill, float* __restrict__ out,
                                                               int64_t HW, int tiles) {
  const int64_t b = blockIdx.x / tiles;
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= HW) return;
  const float td = todo[b * HW + p];
  const float keep = 1.0f - td;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int64_t i = (b * 3 + c) * HW + p;
    out[i] = keep * img[i] + td * fill[i];
  }
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_inpaint_propagate_fwd(const float* flow, const float* ident, const float* ref_img,
                                           const float* ref_mask, const float* shadow, const float* const* enter_region,
                                           const float* const* enter_look, const float* const* enter_flow, int n_enter,
                                           const float* img, const float* todo, const float* obj, float* img_out,
                                           float* todo_out, float* inp_img, float* inp_mask, int64_t B, int H, int W,
                                           int soft_shadow, int fix_mask, waldo_stream_t stream) {
  if (B < 0 || H < 1 || W < 1 || H > 32767 || W > 32767 || n_enter < 0 || n_enter > 2) {
    set_error("waldo_inpaint_propagate_fwd: bad arguments B=%lld H=%d W=%d entering=%d (at most two entering objects)",
              (long long)B, H, W, n_enter);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!flow || !ident || !ref_img || !ref_mask || !img || !todo || !obj || !img_out || !todo_out || !inp_mask ||
      (!fix_mask && !inp_img) || (n_enter > 0 && (!enter_region || !enter_look || !enter_flow))) {
    set_error("waldo_inpaint_propagate_fwd: null pointer");
    return WALDO_EINVAL;
  }
  EnterArg e[2] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  for (int k = 0; k < n_enter; ++k) {
    e[k] = EnterArg{enter_region[k], enter_look[k], enter_flow[k]};
    if (!e[k].region || !e[k].look || !e[k].flow) {
      set_error("waldo_inpaint_propagate_fwd: null pointer (entering object %d)", k);
      return WALDO_EINVAL;
    }
  }
  const int64_t HW = (int64_t)H * W, tiles = (HW + kBlock - 1) / kBlock;
  if (B * tiles > 2147483647) {
    set_error("waldo_inpaint_propagate_fwd: problem too large for one launch");
    return WALDO_EINVAL;
  }
  inpaint_propagate_kernel<<<dim3((unsigned)(B * tiles)), dim3(kBlock), 0, (hipStream_t)stream>>>(
      flow, ident, ref_img, ref_mask, shadow, e[0], e[1], n_enter, img, todo, obj, img_out, todo_out, inp_img, inp_mask, H,
      W, soft_shadow, fix_mask, (int)tiles);
  return launch_status("waldo_inpaint_propagate_fwd");
}

extern "C" int waldo_inpaint_holes_fwd(const float* alpha_ctx, int64_t stride_b, int64_t stride_tc, int64_t stride_tp,
                                       int64_t stride_l, float* mask, float* obj_mask, int64_t B, int Tc, int Tp, int L,
                                       int64_t HW, int last_only, float thresh, waldo_stream_t stream) {
  if (B < 0 || Tc < 1 || Tp < 0 || L < 1 || HW < 1 || stride_b < 0 || stride_tc < 0 || stride_tp < 0 || stride_l < 0) {
    set_error("waldo_inpaint_holes_fwd: bad arguments B=%lld Tc=%d Tp=%d L=%d HW=%lld", (long long)B, Tc, Tp, L,
              (long long)HW);
    return WALDO_EINVAL;
  }
  if (B * Tp == 0) return WALDO_OK;
  if (!alpha_ctx || !mask || !obj_mask) {
    set_error("waldo_inpaint_holes_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int64_t tiles = (HW + kBlock - 1) / kBlock;
  if (B * Tp * tiles > 2147483647) {
    set_error("waldo_inpaint_holes_fwd: problem too large for one launch");
    return WALDO_EINVAL;
  }
  inpaint_holes_kernel<<<dim3((unsigned)(B * Tp * tiles)), dim3(kBlock), 0, (hipStream_t)stream>>>(
      alpha_ctx, CtxStrides{stride_b, stride_tc, stride_tp, stride_l}, mask, obj_mask, Tc, Tp, L, HW, last_only, thresh,
      (int)tiles);
  return launch_status("waldo_inpaint_holes_fwd");
}

extern "C" int waldo_inpaint_blend_fwd(const float* img, const float* todo, const float* fill, float* out, int64_t B,
                                       int64_t HW, waldo_stream_t stream) {
  if (B < 0 || HW < 1) {
    set_error("waldo_inpaint_blend_fwd: bad arguments B=%lld HW=%lld", (long long)B, (long long)HW);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!img || !todo || !fill || !out) {
    set_error("waldo_inpaint_blend_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int64_t tiles = (HW + kBlock - 1) / kBlock;
  if (B * tiles > 2147483647) {
    set_error("waldo_inpaint_blend_fwd: problem too large for one launch");
    return WALDO_EINVAL;
  }
  inpaint_blend_kernel<<<dim3((unsigned)(B * tiles)), dim3(kBlock), 0, (hipStream_t)stream>>>(img, todo, fill, out, HW,
                                                                                              (int)tiles);
  return launch_status("waldo_inpaint_blend_fwd");
}
