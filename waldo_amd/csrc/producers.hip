// Producers of the path's inputs: the small steps that sit between the
// networks and the warp / composite kernels, so that ``estimate_alpha_grid_occ -> decode_output``
// runs without leaving the device or falling back to elementwise framework kernels.
//
//   compute_occ    LVD.compute_occ (models/nets/lvd.py:59-68): pairwise occlusion matrix from the
//                  occlusion scores, forward + backward.
//   alpha_head     ImageDecoder.forward's tail (lvd.py:245-254: + init_bias, tanh and the circular
//                  prior on the alpha channel, x`scale` bilinear upsampling, F.interpolate with
//                  align_corners=False) fused with the padding-mask / remove / freeze arithmetic of
//                  LVD.forward(mode="estimate_alpha_grid_occ") (lvd.py:128-132), forward + backward.
//   pose_affine    the pose heads' affine (models/nets/flp.py:259-273, also lvd.py:440-449):
//                  control points = [base + delta, 1] @ (mul * pose[:6] + bias), forward + backward.
// All three are tiny next to the warp kernels (KBs to a few MB): one thread per output element, no
// atomics (the backward kernels gather), so the gradients are bitwise reproducible.
#include <math.h>

#include "waldo_common.hip.h"

namespace waldo {

// ---------------------------------------------------------------------------------------------
// compute_occ
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void compute_occ_fwd_kernel(const float* __restrict__ score,
                                                                 float* __restrict__ occ, int64_t M,
                                                                 int No, float eps) {
  const int L = No + 1;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= M * L * L) return;
  const int j = (int)(e % L), i = (int)((e / L) % L);
  const int64_t m = e / ((int64_t)L * L);
  float v;
  if (i == 0) {
    v = 0.0f;  // the background occludes nothing
  } else if (j == 0) {
    v = 1.0f;  // everything occludes the background
  } else {
    const float xi = score[m * No + i - 1], xj = score[m * No + j - 1];
    const float si = expf(-(xi * xi)) + eps, sj = expf(-(xj * xj)) + eps;
    v = si / (si + sj) - (i == j ? 0.5f : 0.0f);
  }
  occ[e] = v;
}

// d occ[i+1][j+1] / d s_i = s_j / (s_i + s_j)^2 and d occ[j+1][i+1] / d s_i = -s_j / (s_i + s_j)^2
__global__ __launch_bounds__(kBlock) void compute_occ_bwd_kernel(const float* __restrict__ score,
                                                                 const float* __restrict__ grad_occ,
                                                                 float* __restrict__ grad_score,
                                                                 int64_t M, int No, float eps) {
  const int L = No + 1;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= M * No) return;
  const int i = (int)(e % No);
  const int64_t m = e / No;
  const float xi = score[e];
  const float ei = expf(-(xi * xi)), si = ei + eps;
  const float* g = grad_occ + m * L * L;
  float acc = 0.0f;
  for (int j = 0; j < No; ++j) {
    const float xj = score[m * No + j];
    const float sj = expf(-(xj * xj)) + eps;
    const float den = si + sj;
    acc = fmaf(g[(i + 1) * L + j + 1] - g[(j + 1) * L + i + 1], sj / (den * den), acc);
  }
  grad_score[e] = acc * (-2.0f * xi * ei);
}

// ---------------------------------------------------------------------------------------------
// alpha_head
// ---------------------------------------------------------------------------------------------
struct Lerp1 {
  int i0, i1;
  float l0, l1;
};

// source index of F.interpolate(mode="bilinear", align_corners=False) for an integer up-scale
__device__ __forceinline__ Lerp1 upsample_src(int dst, int in_size, float inv_scale) {
  float src = ((float)dst + 0.5f) * inv_scale - 0.5f;
  src = fmaxf(src, 0.0f);
  Lerp1 r;
  r.i0 = min((int)src, in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

// value of the decoder output before upsampling: + bias; last channel: tanh and the prior blend
__device__ __forceinline__ float head_pre(float v, float bias, bool alpha_ch, const float* prior, int idx) {
  v += bias;
  if (!alpha_ch) return v;
  v = tanhf(v);
  if (prior != nullptr) {
    const float p = prior[idx];
    v = p * 1.0f + (1.0f - p) * v;
  }
  return v;
}

// mode: 0 keep, 1 remove_obj (alpha := -1), 2 freeze_obj (alpha := +1); then the padding mask
__device__ __forceinline__ float head_post(float a, int mode, const float* mask, int idx) {
  if (mode == 1) a = 0.0f * a - 1.0f;
  if (mode == 2) a = 0.0f * a + 1.0f;
  if (mask != nullptr) {
    const float m = mask[idx];
    a = m * a + (1.0f - m) * (-1.0f);
  }
  return a;
}

__global__ __launch_bounds__(kBlock) void alpha_head_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ prior, const float* __restrict__ mask,
    float* __restrict__ out, int64_t N, int C, int h, int w, int scale, float bias, int has_alpha,
    int mode) {
  const int H = h * scale, W = w * scale;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= N * C * H * W) return;
  const int X = (int)(e % W), Y = (int)((e / W) % H);
  const int64_t nc = e / ((int64_t)H * W);
  const int c = (int)(nc % C);
  const bool ach = has_alpha && c == C - 1;
  const float inv = 1.0f / (float)scale;
  const Lerp1 ly = upsample_src(Y, h, inv), lx = upsample_src(X, w, inv);
  const float* xp = x + nc * h * w;
  const float v00 = head_pre(xp[ly.i0 * w + lx.i0], bias, ach, prior, ly.i0 * w + lx.i0);
  const float v01 = head_pre(xp[ly.i0 * w + lx.i1], bias, ach, prior, ly.i0 * w + lx.i1);
  const float v10 = head_pre(xp[ly.i1 * w + lx.i0], bias, ach, prior, ly.i1 * w + lx.i0);
  const float v11 = head_pre(xp[ly.i1 * w + lx.i1], bias, ach, prior, ly.i1 * w + lx.i1);
  const float a = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
  out[e] = (mask != nullptr || mode != 0) ? head_post(a, mode, mask, Y * W + X) : a;
}

// grad_x[y][x] = pre'(x) * sum_Y wy(Y -> y) sum_X wx(X -> x) * post'(Y, X) * grad_out[Y][X]:
// gather over the <= 3*scale output rows / columns whose interpolation touches (y, x)
__global__ __launch_bounds__(kBlock) void alpha_head_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ prior, const float* __restrict__ mask,
    const float* __restrict__ grad_out, float* __restrict__ grad_x, int64_t N, int C, int h, int w,
    int scale, float bias, int has_alpha, int mode) {
  const int H = h * scale, W = w * scale;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= N * C * h * w) return;
  const int xx = (int)(e % w), yy = (int)((e / w) % h);
  const int64_t nc = e / ((int64_t)h * w);
  const int c = (int)(nc % C);
  const bool ach = has_alpha && c == C - 1;
  const float inv = 1.0f / (float)scale;
  float acc = 0.0f;
  if (mode == 0) {  // remove / freeze make the output independent of x
    const float* g = grad_out + nc * H * W;
    const int Y0 = max(0, (yy - 1) * scale), Y1 = min(H - 1, (yy + 2) * scale - 1);
    const int X0 = max(0, (xx - 1) * scale), X1 = min(W - 1, (xx + 2) * scale - 1);
    for (int Y = Y0; Y <= Y1; ++Y) {
      const Lerp1 ly = upsample_src(Y, h, inv);
      const float wy = (ly.i0 == yy ? ly.l0 : 0.0f) + (ly.i1 == yy ? ly.l1 : 0.0f);
      if (wy == 0.0f) continue;
      float row = 0.0f;
      for (int X = X0; X <= X1; ++X) {
        const Lerp1 lx = upsample_src(X, w, inv);
        const float wx = (lx.i0 == xx ? lx.l0 : 0.0f) + (lx.i1 == xx ? lx.l1 : 0.0f);
        const float m = mask != nullptr ? mask[Y * W + X] : 1.0f;
        row = fmaf(wx * m, g[Y * W + X], row);
      }
      acc = fmaf(wy, row, acc);
    }
    if (ach) {
      const float t = tanhf(x[e] + bias);
      const float p = prior != nullptr ? prior[yy * w + xx] : 0.0f;
      acc *= (1.0f - p) * (1.0f - t * t);
    }
  }
  grad_x[e] = acc;
}

// ---------------------------------------------------------------------------------------------
// pose_affine: out[r][p] = [pts_mul * base[p] + mul_delta * pose[r][6 + 2p ..], 1] @ T(r),
//              T(r)[a][b] = mul6[2a + b] * pose[r][2a + b] + bias6[2a + b]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pose_affine_fwd_kernel(
    const float* __restrict__ pose, const float* __restrict__ mul6, const float* __restrict__ bias6,
    const float* __restrict__ base, float* __restrict__ out, int64_t R, int P, float mul_delta,
    float pts_mul) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= R * P) return;
  const int p = (int)(e % P);
  const int64_t r = e / P;
  const float* q = pose + r * (6 + 2 * P);
  float T[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) T[k] = mul6[k] * q[k] + bias6[k];
  const float px = pts_mul * base[2 * p] + mul_delta * q[6 + 2 * p];
  const float py = pts_mul * base[2 * p + 1] + mul_delta * q[6 + 2 * p + 1];
  out[2 * e] = px * T[0] + py * T[2] + T[4];
  out[2 * e + 1] = px * T[1] + py * T[3] + T[5];
}

// one thread per (row, pose component): the six transform components sum over the P points
__global__ __launch_bounds__(kBlock) void pose_affine_bwd_kernel(
    const float* __restrict__ pose, const float* __restrict__ mul6, const float* __restrict__ bias6,
    const float* __restrict__ base, const float* __restrict__ grad_out, float* __restrict__ grad_pose,
    int64_t R, int P, float mul_delta, float pts_mul) {
  const int D = 6 + 2 * P;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= R * D) return;
  const int k = (int)(e % D);
  const int64_t r = e / D;
  const float* q = pose + r * D;
  const float* g = grad_out + r * P * 2;
  float res;
  if (k < 6) {
    const int a = k >> 1, b = k & 1;  // T[a][b]: coordinate a (x, y, 1) feeds output component b
    float acc = 0.0f;
#pragma unroll 8  // P = 128 background points: keep eight steps' loads in flight (same fma order)
    for (int p = 0; p < P; ++p) {
      const float coord = a == 2 ? 1.0f : pts_mul * base[2 * p + a] + mul_delta * q[6 + 2 * p + a];
      acc = fmaf(coord, g[2 * p + b], acc);
    }
    res = acc * mul6[k];
  } else {
    const int p = (k - 6) >> 1, a = (k - 6) & 1;  // d out / d pts[p][a] = T[a][:]
    const float t0 = mul6[2 * a] * q[2 * a] + bias6[2 * a];
    const float t1 = mul6[2 * a + 1] * q[2 * a + 1] + bias6[2 * a + 1];
    res = mul_delta * (g[2 * p] * t0 + g[2 * p + 1] * t1);
  }
  grad_pose[e] = res;
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace waldo

using namespace waldo;

static int check_occ(const char* fn, int64_t M, int No) {
  if (M < 0 || No < 1 || No > 63 || M * (int64_t)(No + 1) * (No + 1) > 2147483647ll * kBlock) {
    set_error("%s: bad shape M=%lld No=%d (need 1<=No<=63)", fn, (long long)M, No);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

extern "C" int waldo_compute_occ_fwd(const float* score, float* occ, int64_t M, int No, float eps,
                                     waldo_stream_t stream) {
  if (int rc = check_occ("waldo_compute_occ_fwd", M, No)) return rc;
  if (M == 0) return WALDO_OK;
  if (!score || !occ) {
    set_error("waldo_compute_occ_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int64_t n = M * (No + 1) * (No + 1);
  hipLaunchKernelGGL(compute_occ_fwd_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, score,
                     occ, M, No, eps);
  return launch_status("waldo_compute_occ_fwd");
}

extern "C" int waldo_compute_occ_bwd(const float* score, const float* grad_occ, float* grad_score,
                                     int64_t M, int No, float eps, waldo_stream_t stream) {
  if (int rc = check_occ("waldo_compute_occ_bwd", M, No)) return rc;
  if (M == 0) return WALDO_OK;
  if (!score || !grad_occ || !grad_score) {
    set_error("waldo_compute_occ_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(compute_occ_bwd_kernel, dim3(blocks_for(M * No)), dim3(kBlock), 0, (hipStream_t)stream,
                     score, grad_occ, grad_score, M, No, eps);
  return launch_status("waldo_compute_occ_bwd");
}

static int check_head(const char* fn, int64_t N, int C, int h, int w, int scale, int mode, const float* mask) {
  if (N < 0 || C < 1 || h < 1 || w < 1 || scale < 1 || scale > 16 || mode < 0 || mode > 2 ||
      N * C * (int64_t)h * w * scale * scale > 2147483647ll * kBlock || (int64_t)h * w * scale * scale > 2147483647ll) {
    set_error("%s: bad shape N=%lld C=%d h=%d w=%d scale=%d mode=%d", fn, (long long)N, C, h, w, scale, mode);
    return WALDO_EINVAL;
  }
  // the padding mask and the remove / freeze modes are the arithmetic of the ONE-channel object alpha
  // (lvd.py:128-132); applied to a C > 1 decoder output they would overwrite its colour channels
  if (C > 1 && (mask != nullptr || mode != 0)) {
    set_error("%s: mask / remove / freeze act on the one-channel object alpha only (C=%d)", fn, C);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

extern "C" int waldo_alpha_head_fwd(const float* x, const float* prior, const float* mask, float* out,
                                    int64_t N, int C, int h, int w, int scale, float bias,
                                    int has_alpha, int mode, waldo_stream_t stream) {
  if (int rc = check_head("waldo_alpha_head_fwd", N, C, h, w, scale, mode, mask)) return rc;
  if (N == 0) return WALDO_OK;
  if (!x || !out) {
    set_error("waldo_alpha_head_fwd: null pointer");
    return WALDO_EINVAL;
  }
  const int64_t n = N * C * (int64_t)h * w * scale * scale;
  hipLaunchKernelGGL(alpha_head_fwd_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, prior,
                     mask, out, N, C, h, w, scale, bias, has_alpha, mode);
  return launch_status("waldo_alpha_head_fwd");
}

extern "C" int waldo_alpha_head_bwd(const float* x, const float* prior, const float* mask,
                                    const float* grad_out, float* grad_x, int64_t N, int C, int h,
                                    int w, int scale, float bias, int has_alpha, int mode,
                                    waldo_stream_t stream) {
  if (int rc = check_head("waldo_alpha_head_bwd", N, C, h, w, scale, mode, mask)) return rc;
  if (N == 0) return WALDO_OK;
  if (!x || !grad_out || !grad_x) {
    set_error("waldo_alpha_head_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(alpha_head_bwd_kernel, dim3(blocks_for(N * C * (int64_t)h * w)), dim3(kBlock), 0,
                     (hipStream_t)stream, x, prior, mask, grad_out, grad_x, N, C, h, w, scale, bias, has_alpha,
                     mode);
  return launch_status("waldo_alpha_head_bwd");
}

// Synthesizer.predict's disocclusion test (models/synthesizer.py:447-450 and 475-478) on the by-product of the fused
// flow pass, m = alpha_ctx.max(dim=3)[0] (B, Tc, Tp, HW):
//   dmax = m.max(dim=1)[0];  dmin = m.min(dim=1)[0];  dmax[dmax - dmin > 1] = 0      -> (B, Tp, HW)
// One pass (Tc planes read, one written) instead of torch's aminmax + subtract + compare + masked_fill over
// (B, Tp, HW) temporaries.  torch.max / min return NaN when any element is NaN and `NaN > 1` is false: IEEE
// maximum / minimum (v_maximum3_f32 / v_minimum3_f32) keep both.
template <typename V>
__global__ __launch_bounds__(kBlock) void disocc_test_kernel(const V* __restrict__ m, V* __restrict__ out, int Tc,
                                                             int Tp, int64_t hw, int tiles) {
  const int64_t u = blockIdx.x / tiles;  // (b, tp)
  const int64_t p = (int64_t)(blockIdx.x % tiles) * kBlock + threadIdx.x;
  if (p >= hw) return;
  const int64_t b = u / Tp, tp = u - b * Tp;
  const V* src = m + (b * Tc * Tp + tp) * hw + p;
  V mx = src[0], mn = mx;
  for (int t = 1; t < Tc; ++t) {
    const V v = src[(int64_t)t * Tp * hw];
    mx = __builtin_elementwise_maximum(mx, v);
    mn = __builtin_elementwise_minimum(mn, v);
  }
  const V d = mx - mn;
  if constexpr (sizeof(V) == 4) {
    out[u * hw + p] = d > 1.0f ? 0.0f : mx;
  } else {
    V r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = d[i] > 1.0f ? 0.0f : mx[i];
    out[u * hw + p] = r;
  }
}

static int check_pose(const char* fn, int64_t R, int P) {
  if (R < 0 || P < 1 || P > 4096 || R * (int64_t)(6 + 2 * P) > 2147483647ll * kBlock) {
    set_error("%s: bad shape R=%lld P=%d", fn, (long long)R, P);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

extern "C" int waldo_pose_affine_fwd(const float* pose, const float* mul6, const float* bias6,
                                     const float* base_pts, float* out, int64_t R, int P,
                                     float mul_delta, float pts_mul, waldo_stream_t stream) {
  if (int rc = check_pose("waldo_pose_affine_fwd", R, P)) return rc;
  if (R == 0) return WALDO_OK;
  if (!pose || !mul6 || !bias6 || !base_pts || !out) {
    set_error("waldo_pose_affine_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(pose_affine_fwd_kernel, dim3(blocks_for(R * P)), dim3(kBlock), 0, (hipStream_t)stream, pose,
                     mul6, bias6, base_pts, out, R, P, mul_delta, pts_mul);
  return launch_status("waldo_pose_affine_fwd");
}

extern "C" int waldo_pose_affine_bwd(const float* pose, const float* mul6, const float* bias6,
                                     const float* base_pts, const float* grad_out, float* grad_pose,
                                     int64_t R, int P, float mul_delta, float pts_mul,
                                     waldo_stream_t stream) {
  if (int rc = check_pose("waldo_pose_affine_bwd", R, P)) return rc;
  if (R == 0) return WALDO_OK;
  if (!pose || !mul6 || !bias6 || !base_pts || !grad_out || !grad_pose) {
    set_error("waldo_pose_affine_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(pose_affine_bwd_kernel, dim3(blocks_for(R * (6 + 2 * P))), dim3(kBlock), 0,
                     (hipStream_t)stream, pose, mul6, bias6, base_pts, grad_out, grad_pose, R, P, mul_delta,
                     pts_mul);
  return launch_status("waldo_pose_affine_bwd");
}

extern "C" int waldo_disocc_test_fwd(const float* layer_max, float* out, int64_t B, int Tc, int Tp, int64_t HW,
                                     waldo_stream_t stream) {
  if (B < 0 || Tc < 1 || Tp < 1 || HW < 1 || B * Tp * ((HW + kBlock - 1) / kBlock) > 2147483647ll) {
    set_error("waldo_disocc_test_fwd: bad shape B=%lld Tc=%d Tp=%d HW=%lld", (long long)B, Tc, Tp, (long long)HW);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!layer_max || !out) {
    set_error("waldo_disocc_test_fwd: null pointer");
    return WALDO_EINVAL;
  }
  typedef float f32x4_d __attribute__((ext_vector_type(4)));
  const bool wide = HW % 4 == 0 && ((uintptr_t)layer_max % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const int64_t n = wide ? HW / 4 : HW;
  const int tiles = (int)((n + kBlock - 1) / kBlock);
  if (wide)
    hipLaunchKernelGGL(disocc_test_kernel<f32x4_d>, dim3((unsigned)(B * Tp * tiles)), dim3(kBlock), 0,
                       (hipStream_t)stream, reinterpret_cast<const f32x4_d*>(layer_max),
                       reinterpret_cast<f32x4_d*>(out), Tc, Tp, n, tiles);
  else
    hipLaunchKernelGGL(disocc_test_kernel<float>, dim3((unsigned)(B * Tp * tiles)), dim3(kBlock), 0,
                       (hipStream_t)stream, layer_max, out, Tc, Tp, n, tiles);
  return launch_status("waldo_disocc_test_fwd");
}
