// Fused WIF hot path -- C-ABI entry points and dispatch over the compiled (LP, K3P) variants.
// Kernels: warp_composite_kernels.hip.h; one translation unit per padded layer count
// (warp_composite_lp*.hip) so that the variants compile in parallel.
#include "waldo_common.hip.h"

namespace waldo {

constexpr int kMaxLayers = 32;
constexpr int kMaxK3 = 32;

#define WALDO_DECL_LP(LPV)                                                                      \
  void wc_fwd_lp##LPV(bool k19, const float* layers, const float* basis_t, const float* mapping, \
                      const float* inv_kernel, const float* src_pts, const float* occ,          \
                      float* rgb, float* alpha, int F, int L, int H, int W, int K3, float delta, \
                      hipStream_t st);                                                          \
  void wc_bwd_lp##LPV(bool k19, const float* layers, const float* basis_t, const float* mapping, \
                      const float* occ, const float* grad_rgb, const float* grad_alpha,         \
                      float* grad_layers, float* grad_mapping, float* grad_occ,                 \
                      void* workspace, int F, int L, int H, int W, int K3, float delta,         \
                      hipStream_t st);
WALDO_DECL_LP(4)
WALDO_DECL_LP(8)
WALDO_DECL_LP(12)
WALDO_DECL_LP(17)
WALDO_DECL_LP(24)
WALDO_DECL_LP(32)

static int check_common(const char* fn, int64_t F, int L, int H, int W, int K3) {
  if (F < 0 || L < 1 || L > kMaxLayers || H < 1 || W < 1 || K3 < 3 || K3 > kMaxK3) {
    set_error("%s: unsupported shape F=%lld L=%d H=%d W=%d K3=%d (need 1<=L<=%d, 3<=K3<=%d)", fn,
              (long long)F, L, H, W, K3, kMaxLayers, kMaxK3);
    return WALDO_EINVAL;
  }
  if (F > 65535 || F * L > 65535 || (int64_t)H * W > (int64_t)2147483647 / 4 || H > 32767 ||
      W > 32767) {
    set_error("%s: F*L=%lld (max 65535 per launch) or H*W=%lld too large", fn, (long long)F * L,
              (long long)H * W);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

#define WALDO_CALL_LP(FN, ...)                    \
  do {                                            \
    if (L <= 4) FN##4(__VA_ARGS__);               \
    else if (L <= 8) FN##8(__VA_ARGS__);          \
    else if (L <= 12) FN##12(__VA_ARGS__);        \
    else if (L <= 17) FN##17(__VA_ARGS__);        \
    else if (L <= 24) FN##24(__VA_ARGS__);        \
    else FN##32(__VA_ARGS__);                     \
  } while (0)

extern "C" int waldo_max_layers(void) { return kMaxLayers; }

extern "C" int64_t waldo_warp_composite_bwd_workspace_bytes(int64_t F, int L, int H, int W,
                                                            int K3) {
  if (F < 0 || L < 1 || H < 1 || W < 1 || debug_option(WALDO_DEBUG_BWD_GENERIC)) return 0;
  return bwd_workspace_bytes(F, L, H, W, K3);
}

extern "C" int waldo_warp_composite_fwd(const float* layers, const float* basis_t,
                                        const float* mapping, const float* occ, float* rgb,
                                        float* alpha, int64_t F, int L, int H, int W, int K3,
                                        float delta, waldo_stream_t stream) {
  int rc = check_common("waldo_warp_composite_fwd", F, L, H, W, K3);
  if (rc) return rc;
  if (F == 0) return WALDO_OK;
  if (!layers || !basis_t || !mapping || !occ || !rgb) {
    set_error("waldo_warp_composite_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  WALDO_CALL_LP(wc_fwd_lp, K3 == 19, layers, basis_t, mapping, nullptr, nullptr, occ, rgb, alpha, (int)F,
                L, H, W, K3, delta, st);
  return launch_status("waldo_warp_composite_fwd");
}

extern "C" int waldo_warp_composite_pts_supported(int L, int H, int W, int N) {
  return N + 3 == kGmapK3 && L >= 1 && L <= kMaxLayers && H >= 1 && W >= 1 && staged_eligible(H, W) &&
         (int64_t)H * W * kGmapK3 * 4 < 4294967296ll && !debug_option(WALDO_DEBUG_FWD_PLAIN);
}

extern "C" int waldo_warp_composite_pts_fwd(const float* layers, const float* basis_t,
                                            const float* inverse_kernel, const float* src_pts,
                                            const float* occ, float* rgb, float* alpha, int64_t F,
                                            int L, int H, int W, int N, float delta,
                                            waldo_stream_t stream) {
  int rc = check_common("waldo_warp_composite_pts_fwd", F, L, H, W, N + 3);
  if (rc) return rc;
  if (!waldo_warp_composite_pts_supported(L, H, W, N)) {
    set_error("waldo_warp_composite_pts_fwd: shape not served (N=%d H=%d W=%d); use waldo_tps_mapping_fwd + "
              "waldo_warp_composite_fwd", N, H, W);
    return WALDO_EINVAL;
  }
  if (F == 0) return WALDO_OK;
  if (!layers || !basis_t || !inverse_kernel || !src_pts || !occ || !rgb) {
    set_error("waldo_warp_composite_pts_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int K3 = N + 3;
  WALDO_CALL_LP(wc_fwd_lp, true, layers, basis_t, nullptr, inverse_kernel, src_pts, occ, rgb, alpha, (int)F, L,
                H, W, K3, delta, st);
  return launch_status("waldo_warp_composite_pts_fwd");
}

extern "C" int waldo_warp_composite_bwd(const float* layers, const float* basis_t,
                                        const float* mapping, const float* occ,
                                        const float* grad_rgb, const float* grad_alpha,
                                        float* grad_layers, float* grad_mapping, float* grad_occ,
                                        void* workspace, int64_t workspace_bytes, int64_t F,
                                        int L, int H, int W, int K3, float delta,
                                        waldo_stream_t stream) {
  int rc = check_common("waldo_warp_composite_bwd", F, L, H, W, K3);
  if (rc) return rc;
  if (F == 0) return WALDO_OK;
  if (!layers || !basis_t || !mapping || !occ || !grad_rgb || !grad_layers) {
    set_error("waldo_warp_composite_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t need = debug_option(WALDO_DEBUG_BWD_GENERIC) ? 0 : bwd_workspace_bytes(F, L, H, W, K3);
  if (workspace != nullptr && (need == 0 || workspace_bytes < need)) {
    if (need != 0) {
      set_error("waldo_warp_composite_bwd: workspace of %lld bytes given, %lld needed",
                (long long)workspace_bytes, (long long)need);
      return WALDO_EINVAL;
    }
    workspace = nullptr;  // shape served by the generic kernel, which needs none
  }
  WALDO_CALL_LP(wc_bwd_lp, K3 == 19, layers, basis_t, mapping, occ, grad_rgb, grad_alpha,
                grad_layers, grad_mapping, grad_occ, workspace, (int)F, L, H, W, K3, delta, st);
  return launch_status("waldo_warp_composite_bwd");
}

#ifdef WALDO_K1_STAMPS
// diagnostic builds only: copies the stamp buffer of the LP = 8 translation unit to the host
namespace waldo { int k1_stamps_read(unsigned long long* dst, int n); }
extern "C" int waldo_debug_k1_stamps(unsigned long long* dst, int n) { return waldo::k1_stamps_read(dst, n); }
#endif
#ifdef WALDO_FWD_STAMPS
namespace waldo { int fwd_stamps_read(unsigned long long* dst, int n); }
extern "C" int waldo_debug_fwd_stamps(unsigned long long* dst, int n) { return waldo::fwd_stamps_read(dst, n); }
#endif
