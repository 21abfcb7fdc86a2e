// Instantiation of the fused warp/composite kernels for padded layer count WALDO_LP.
#include "warp_composite_kernels.hip.h"

#define WALDO_CAT_(a, b) a##b
#define WALDO_CAT(a, b) WALDO_CAT_(a, b)

namespace waldo {

void WALDO_CAT(wc_fwd_lp, WALDO_LP)(bool k19, const float* layers, const float* basis_t,
                                    const float* mapping, const float* inv_kernel,
                                    const float* src_pts, const float* occ, float* rgb,
                                    float* alpha, int F, int L, int H, int W, int K3, float delta,
                                    hipStream_t st) {
  if (k19)
    launch_fwd<WALDO_LP, 19, true>(layers, basis_t, mapping, inv_kernel, src_pts, occ, rgb, alpha, F, L, H, W,
                                   K3, delta, st);
  else
    launch_fwd<WALDO_LP, 32, false>(layers, basis_t, mapping, nullptr, nullptr, occ, rgb, alpha, F, L, H, W,
                                    K3, delta, st);
}

// `workspace` != nullptr selects the two-kernel backward (compiled for L <= kBwd2MaxLayers, K3 == 19)
void WALDO_CAT(wc_bwd_lp, WALDO_LP)(bool k19, const float* layers, const float* basis_t,
                                    const float* mapping, const float* occ, const float* grad_rgb,
                                    const float* grad_alpha, float* grad_layers,
                                    float* grad_mapping, float* grad_occ, void* workspace, int F,
                                    int L, int H, int W, int K3, float delta, hipStream_t st) {
#if WALDO_LP <= 17
  if (k19 && workspace != nullptr) {
    launch_bwd2<WALDO_LP>(layers, basis_t, mapping, occ, grad_rgb, grad_alpha, workspace,
                          grad_layers, grad_mapping, grad_occ, F, L, H, W, delta, st);
    return;
  }
#endif
  if (k19)
    launch_bwd<WALDO_LP, 19>(layers, basis_t, mapping, occ, grad_rgb, grad_alpha, grad_layers,
                             grad_mapping, grad_occ, F, L, H, W, K3, delta, st);
  else
    launch_bwd<WALDO_LP, 32>(layers, basis_t, mapping, occ, grad_rgb, grad_alpha, grad_layers,
                             grad_mapping, grad_occ, F, L, H, W, K3, delta, st);
}

}  // namespace waldo
