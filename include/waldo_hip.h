/*
 * waldo_hip.h -- C ABI of the MI355X (gfx950) WIF warp/composite hot path.
 *
 * Drop-in boundary for the operator layer of 16lemoing/waldo that the hot path lives in
 * (reference files cited per entry point; paths relative to the reference root).  Every entry
 * point is a stream-ordered launcher:
 *   - plain pointers and sizes only (no torch types); all tensors are fp32, contiguous, in the
 *     layout stated below; index tensors are int32/int64 as stated;
 *   - the CALLER owns and allocates every buffer (inputs, outputs, workspaces); the library never
 *     allocates or frees device memory and never synchronises;
 *   - kernels are launched on `stream` (a hipStream_t passed as void*) of the CURRENT device;
 *   - returns 0 on success, a negative WALDO_E* code otherwise; no C++ exception crosses the ABI;
 *     waldo_last_error_string() gives the message of the last failure on the calling thread;
 *   - re-entrant: no global mutable state except the thread-local error string and ONE piece of
 *     process-global state, the debug options of waldo_set_debug_option below (three test-only
 *     switches between kernel variants that compute the same thing; all off unless a test
 *     switches one on; a caller that never calls it has a stateless library).
 *
 * Build: hipcc --offload-arch=gfx950 -shared -fPIC (see waldo_amd/build.py).
 */
#ifndef WALDO_HIP_H
#define WALDO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WALDO_OK 0
#define WALDO_EINVAL (-1)  /* bad shape / null pointer / unsupported size */
#define WALDO_ELAUNCH (-2) /* hipLaunch / runtime error */

typedef void* waldo_stream_t; /* hipStream_t */

/* ABI version: major*1000 + minor. */
int waldo_version(void);
const char* waldo_last_error_string(void);
/* Largest layer count L the fused composite kernels accept. */
int waldo_max_layers(void);

/* Test-only switches between kernel variants that compute the same thing (A/B parity tests of the
 * fast paths against the plain ones).  THE ONE PIECE OF PROCESS-GLOBAL STATE of the library: process-wide,
 * off by default, relaxed atomics; nothing reads the environment.  Kernel variants that were measured and
 * rejected are NOT in the library (tools_dev/dropped/, buildable with tools_dev/build_variant.py). */
#define WALDO_DEBUG_FWD_PLAIN 0   /* fused forward: gather kernel instead of the LDS-staged one */
#define WALDO_DEBUG_IW_PASSES 1   /* grid inversion: one kernel per fill / erosion pass */
#define WALDO_DEBUG_BWD_GENERIC 2 /* fused backward: the generic per-tap-atomics kernel for every shape
                                     (waldo_warp_composite_bwd_workspace_bytes answers 0) */
#define WALDO_DEBUG_COUNT 3
int waldo_set_debug_option(int option, int value);

/* Frame-index status.  The reference's gather_time (models/nets/lvd.py:462-467: `tensor.gather(1, ts)`) RAISES for
 * a frame index outside the time axis.  The forward entry points that index frames with ctx_ts / pred_ts from
 * device memory (waldo_time_gather_fwd, waldo_flow_ctx_warp_fwd / _raw_fwd, waldo_frame_warp_fuse_fwd / _raw_fwd)
 * take `status`: WALDO_INDEX_STATUS_WORDS int32 words owned by the caller (zero them once), device-accessible --
 * device memory, or pinned host memory through waldo_host_device_pointer, which the caller can read WITHOUT
 * synchronising the device.  A kernel that meets an index outside its range clamps it (memory safety) and reports:
 *   status[0] = the limit a ctx_ts entry violated (valid: 0 .. limit-1; never 0), status[1] = an offending value,
 *   status[2], status[3] likewise for pred_ts.
 * The words are STICKY (kernels only ever write non-zero limits); the caller checks and clears them when it
 * chooses -- after a synchronisation for the reference's raise-at-once behaviour, or lazily (a captured HIP graph:
 * after a replay).  status == NULL: out-of-range indices are clamped silently.  The backward entry points read the
 * indices the forward saw and clamp only. */
#define WALDO_INDEX_STATUS_WORDS 4
/* Device pointer of a pinned (page-locked, mapped) host allocation, hipHostGetDevicePointer: WALDO_EINVAL when
 * `host` is not such memory -- so that a status word in host memory fails at set-up, not as a fault in a kernel. */
int waldo_host_device_pointer(void* host, void** device);

/* ---------------------------------------------------------------------------------------
 * A2. Thin-plate-spline grid synthesis -- replaces TPSWarp.forward
 *     (models/modules/warp.py:49-55).  K3 = N + 3.
 *   mapping[b] = inverse_kernel (K3,K3) @ [src_pts[b] (N,2); 0 (3,2)]          (warp.py:52-53)
 *   grid[b,p]  = sum_k basis_t[k,p] * mapping[b,k]                              (warp.py:54)
 * basis_t is the reference buffer `tgt_grid_repr` (HW,K3) stored TRANSPOSED as (K3,HW) so that
 * a wavefront reads 64 consecutive pixels of one basis function in one coalesced request.
 * ------------------------------------------------------------------------------------- */
int waldo_tps_mapping_fwd(const float* inverse_kernel, const float* src_pts, float* mapping,
                          int64_t B, int N, waldo_stream_t stream);
/* grad_src_pts[b,n] = sum_r inverse_kernel[r,n] * grad_mapping[b,r]  (rows n < N only) */
int waldo_tps_mapping_bwd(const float* inverse_kernel, const float* grad_mapping,
                          float* grad_src_pts, int64_t B, int N, waldo_stream_t stream);
/* grid (B,HW,2) */
int waldo_tps_grid_fwd(const float* basis_t, const float* mapping, float* grid, int64_t B,
                       int64_t HW, int K3, waldo_stream_t stream);
/* grad_mapping (B,K3,2) is OVERWRITTEN (the launcher zero-fills it on `stream` first). */
int waldo_tps_grid_bwd(const float* basis_t, const float* grad_grid, float* grad_mapping,
                       int64_t B, int64_t HW, int K3, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A3. Grid inversion by forward splat + hole filling -- replaces InverseWarp.forward
 *     (models/modules/warp.py:71-174; pad == True) and its autograd.
 *   src_grid (B,Hs,Ws,2) layer->image grid;  src_id (Hs,Ws,2), tgt_id (H,W,2) the identity grids
 *   (reference buffers `src_grid`, `tgt_grid`);  gauss (ksize*ksize) the reference buffer `kernel`, ksize = the
 *   module's `kernel_size`: odd, 1 ... 15 (3 in every script: one launch; other sizes run the fill passes one by one);
 *   out (B,H,W,2) image->layer grid.  Hp = H + 2*(niter+1), Wp likewise.
 * Among several samples landing on one cell the lowest sample index wins (the reference's answer
 * under a stable sort; its own tie-break is implementation-defined).
 * Work / saved buffers, all caller-allocated, contents irrelevant on entry:
 *   dxy (B,2,H*W) f32, cell (B,H*W) i32, winner (B,H*W) i32, field_a / field_b (B,2,Hp*Wp) f32,
 *   fill_iter (B,Hp*Wp) u8, denom (B,Hp*Wp) f32, mask_a / mask_b (B,Hp*Wp) u8.
 * The backward needs cell, winner, fill_iter, denom and mask_a (final mask) as the forward left
 * them, plus gfield (B,2,Hp*Wp) f32 scratch; grad_src_grid (B,Hs,Ws,2) is overwritten.
 * ------------------------------------------------------------------------------------- */
int waldo_inverse_warp_fwd(const float* src_grid, const float* src_id, const float* tgt_id,
                           const float* gauss, float* out, float* dxy, int* cell, int* winner,
                           float* field_a, float* field_b, unsigned char* fill_iter, float* denom,
                           unsigned char* mask_a, unsigned char* mask_b, int64_t B, int Hs, int Ws,
                           int H, int W, int niter, int erode, int ksize, waldo_stream_t stream);
/* The same with the tie-break order given (InverseWarp with num_perm > 1, warp.py:91-111, one
 * call per permutation; the results are averaged by the caller -- the fill, the erosion and the
 * final grid are linear in the elected field for a fixed set of occupied cells, and that set does
 * not depend on the order):  order (H*W) i32 a permutation of the samples (one row of the
 * reference buffer `perm`), rank (H*W) i32 its inverse.  Among the samples landing on one cell
 * the one standing first in `order` wins.  waldo_inverse_warp_bwd serves both. */
int waldo_inverse_warp_order_fwd(const float* src_grid, const float* src_id, const float* tgt_id,
                                 const float* gauss, const int* rank, const int* order,
                                 float* out, float* dxy, int* cell, int* winner, float* field_a,
                                 float* field_b, unsigned char* fill_iter, float* denom,
                                 unsigned char* mask_a, unsigned char* mask_b, int64_t B, int Hs,
                                 int Ws, int H, int W, int niter, int erode, int ksize, waldo_stream_t stream);
int waldo_inverse_warp_bwd(const float* grad_out, const float* gauss, const int* cell,
                           const int* winner, const unsigned char* fill_iter, const float* denom,
                           const unsigned char* mask, float* gfield, float* grad_src_grid,
                           int64_t B, int Hs, int Ws, int H, int W, int niter, int ksize,
                           waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A4/A5. Bilinear backward warp -- replaces F.grid_sample(x + delta, grid) - delta with the
 *     defaults (bilinear, zeros, align_corners=False) as used by Warper.obj_to_output /
 *     bg_to_output / obj_from_input / bg_from_input (models/nets/lvd.py:502-559).
 *   input (Nin,C,Hi,Wi), grid (N,Ho,Wo,2), output (N,C,Ho,Wo).
 *   Input batch broadcast (the reference's .expand over T, lvd.py:544,555):
 *       n_in = (n / outer_div) * inner + (n % inner);  pass outer_div = inner = N for identity.
 *   Forward only: the same map for the GRID, n_grid = (n / grid_outer_div) * grid_inner + n % grid_inner,
 *       grid (N_grid,Ho,Wo,2) -- the predicted frames' grids repeated over the Tc contexts
 *       (`[:, pred_ts].unsqueeze(1).expand(-1, Tc, ...)`, lvd.py:665-668) without the Tc copies;
 *       grid_outer_div = grid_inner = N for one grid per output.
 * ------------------------------------------------------------------------------------- */
int waldo_grid_sample2d_fwd(const float* input, const float* grid, float* output, int64_t N,
                            int C, int Hi, int Wi, int Ho, int Wo, float delta,
                            int64_t outer_div, int64_t inner, int64_t grid_outer_div,
                            int64_t grid_inner, waldo_stream_t stream);
/* The same with two extras for Warper.grid_to_flow_ctx (models/nets/lvd.py:785-796), where the reference warps an
 * all-ones canvas, then the object flows and the background flow with the same grids, and concatenates the results:
 *   mask_out (N,1,Ho,Wo) or NULL: the sample of an ALL-ONES image at the same grid -- what Warper.obj_to_output(ones,
 *     grid) returns and the ghost test thresholds at 0.9 (lvd.py:785-791) -- a by-product of the taps (one store
 *     instead of a launch of its own over B*Tc*Tp*No maps);
 *   out_group / out_stride / out_offset: output map n is written to slot (n / out_group) * out_stride + out_offset +
 *     n % out_group of an output tensor of (slots, C, Ho, Wo) maps ((N, N, 0): the plain output).  With (1, L, 0) for
 *     the background and (No, L, 1) for the objects the two calls of Warper.layer_to_output (lvd.py:533-537) write
 *     straight into the (frames, L, C, Ho, Wo) tensor its torch.cat would build;
 *   pre_scale / pre_bias: the image sampled is pre_scale * input + pre_bias -- Warper.grid_to_flow[_ctx] warps
 *     `(obj_alpha + 1) / 2` and `(bg_alpha + 1) / 2` (lvd.py:602-606, 716-720): (0.5, 0.5) warps them without the
 *     two images being written first.  (1, 0): the plain call. */
int waldo_grid_sample2d_ex_fwd(const float* input, const float* grid, float* output, float* mask_out,
                               int64_t N, int C, int Hi, int Wi, int Ho, int Wo, float delta,
                               int64_t outer_div, int64_t inner, int64_t grid_outer_div, int64_t grid_inner,
                               int64_t out_group, int64_t out_stride, int64_t out_offset, float pre_scale,
                               float pre_bias, waldo_stream_t stream);
/* grad_input (Nin,C,Hi,Wi) must be ZERO-FILLED by the caller (accumulated with atomics; may be
 * NULL to skip); grad_grid (N,Ho,Wo,2) is overwritten (may be NULL to skip). */
int waldo_grid_sample2d_bwd(const float* input, const float* grid, const float* grad_output,
                            float* grad_input, float* grad_grid, int64_t N, int C, int Hi, int Wi,
                            int Ho, int Wo, float delta, int64_t outer_div, int64_t inner,
                            waldo_stream_t stream);
/* The backward of a waldo_grid_sample2d_ex_fwd call (one grid per output map): grad_output is the gradient of the
 * WHOLE (slots, C, Ho, Wo) tensor the forward wrote into, and map n reads slot (n / gout_group) * gout_stride +
 * gout_offset + n % gout_group of it -- the backward of Warper.layer_to_output's torch.cat (lvd.py:533-537) without
 * the two slices of the gradient being copied out first; pre_scale / pre_bias as in the forward (grad_input is the
 * gradient of `input`, not of the scaled image). */
int waldo_grid_sample2d_ex_bwd(const float* input, const float* grid, const float* grad_output,
                               float* grad_input, float* grad_grid, int64_t N, int C, int Hi, int Wi,
                               int Ho, int Wo, float delta, int64_t outer_div, int64_t inner,
                               int64_t gout_group, int64_t gout_stride, int64_t gout_offset, float pre_scale,
                               float pre_bias, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A6. Occlusion product / soft-alpha composite -- replaces the
 *     (1 - alpha * occ).prod(dim) * alpha pattern of models/nets/lvd.py:651-652,686,764-765,809
 *     (spec form LVD.reduce_comp, lvd.py:100-114).
 *   alpha (M,L,HW) in [0,1], occ (Mo,L,L) with m_occ = m / occ_div (broadcast of occ over a
 *   trailing group, e.g. Tc), out (M,L,HW):  out[m,j,p] = alpha[m,j,p] * prod_i (1 - alpha[m,i,p]*occ[m_occ,i,j])
 * ------------------------------------------------------------------------------------- */
int waldo_occ_composite_fwd(const float* alpha, const float* occ, float* out, int64_t M, int L,
                            int64_t HW, int64_t occ_div, waldo_stream_t stream);
/* grad_alpha (M,L,HW) overwritten; grad_occ (Mo,L,L) must be ZERO-FILLED by the caller
 * (accumulated with atomics; may be NULL to skip). */
int waldo_occ_composite_bwd(const float* alpha, const float* occ, const float* grad_out,
                            float* grad_alpha, float* grad_occ, int64_t M, int L, int64_t HW,
                            int64_t occ_div, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f2. Producers of the path's inputs: the steps between the networks and the warp kernels.
 *
 * waldo_compute_occ_*: LVD.compute_occ (models/nets/lvd.py:59-68).
 *   score (M,No) -> occ (M,No+1,No+1):  s = exp(-score^2) + eps;
 *   occ[i+1][j+1] = s_i / (s_i + s_j) - [i == j] / 2;  occ[i+1][0] = 1;  occ[0][*] = 0.
 *   Backward: grad_score (M,No) is OVERWRITTEN.
 * waldo_alpha_head_*: ImageDecoder.forward's tail (lvd.py:245-254) + the alpha arithmetic of
 *   LVD.forward(mode="estimate_alpha_grid_occ") (lvd.py:128-132) in one pass.
 *   x (N,C,h,w) the decoder's raw image;  y = x + bias;  on the LAST channel when has_alpha:
 *   y = tanh(y), then y = prior + (1 - prior) * y with prior (h,w) (the `circle` buffer; NULL:
 *   no prior);  out (N,C,h*scale,w*scale) = F.interpolate(y, scale_factor=scale, "bilinear",
 *   align_corners=False);  then mode 1 (remove_obj): out = -1, mode 2 (freeze_obj): out = +1;
 *   then with mask (h*scale,w*scale) (the `obj_alpha_mask` buffer; NULL: none):
 *   out = mask * out - (1 - mask).   mode != 0 and mask act on the ONE-channel object alpha of
 *   lvd.py:128-132: with C > 1 either of them is refused (WALDO_EINVAL).
 *   Backward: grad_x (N,C,h,w) is OVERWRITTEN (gathered: no atomics).
 * waldo_pose_affine_*: the pose heads' affine (models/nets/flp.py:259-273, lvd.py:440-449).
 *   pose (R,6+2P) (after tanh / + last);  T = (mul6 * pose[:6] + bias6) as (3,2);
 *   pts[p] = pts_mul * base_pts[p] + mul_delta * pose[6+2p : 8+2p];  out (R,P,2) = [pts, 1] @ T.
 *   Backward: grad_pose (R,6+2P) is OVERWRITTEN.
 * ------------------------------------------------------------------------------------- */
int waldo_compute_occ_fwd(const float* score, float* occ, int64_t M, int No, float eps,
                          waldo_stream_t stream);
int waldo_compute_occ_bwd(const float* score, const float* grad_occ, float* grad_score, int64_t M,
                          int No, float eps, waldo_stream_t stream);
int waldo_alpha_head_fwd(const float* x, const float* prior, const float* mask, float* out,
                         int64_t N, int C, int h, int w, int scale, float bias, int has_alpha,
                         int mode, waldo_stream_t stream);
int waldo_alpha_head_bwd(const float* x, const float* prior, const float* mask,
                         const float* grad_out, float* grad_x, int64_t N, int C, int h, int w,
                         int scale, float bias, int has_alpha, int mode, waldo_stream_t stream);
int waldo_pose_affine_fwd(const float* pose, const float* mul6, const float* bias6,
                          const float* base_pts, float* out, int64_t R, int P, float mul_delta,
                          float pts_mul, waldo_stream_t stream);
int waldo_pose_affine_bwd(const float* pose, const float* mul6, const float* bias6,
                          const float* base_pts, const float* grad_out, float* grad_pose, int64_t R,
                          int P, float mul_delta, float pts_mul, waldo_stream_t stream);
/* waldo_disocc_test_fwd: the disocclusion test of Synthesizer.predict (models/synthesizer.py:447-450, 475-478)
 *   on layer_max (B,Tc,Tp,HW) = alpha_ctx.max(dim=3)[0] (the `alpha_max` by-product of waldo_flow_ctx_warp_*):
 *   dmax = max over Tc, dmin = min over Tc (NaN-propagating, as torch's);  out (B,Tp,HW) = dmax, 0 where
 *   dmax - dmin > 1.  Inference only (the reference runs it under no_grad). */
int waldo_disocc_test_fwd(const float* layer_max, float* out, int64_t B, int Tc, int Tp, int64_t HW,
                          waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A9: the two full-resolution passes of Warper.grid_to_flow_ctx / grid_to_flow
 * (models/nets/lvd.py:707-828, 602-705); their backward entry points follow A10.  The low-resolution
 * inputs come from waldo_grid_sample2d_fwd (lvd.py:723-728, 784-796); Hd = H*scale, Wd = W*scale.
 *
 * waldo_flow_ctx_alpha_fwd (lvd.py:731-766): bilinear x`scale` upsampling (F.interpolate,
 * align_corners=False) of the rough alphas, layout filter, occlusion product.
 *   alpha_lr (B*Tw,L,H,W) in [0,1]   rough alpha of every layer in image space, frames 0..Tw-1
 *   input    (B,T,C,Hd,Wd)           layout logits in channels [chan_off, chan_off+Nl) (lvd.py:731)
 *   dist     (B,L-1,Nl) or NULL      class distribution of every object (NULL: no filter)
 *   occ      (B,T,L,L)               occlusion order (LVD.compute_occ)
 *   a01      (B*Tw,L,Hd,Wd)          out: a'_j = a_j prod_i (1 - a_i occ[i][j]) in [0,1]
 *   alpha_out same shape or NULL     out: 2a' - 1 (the method's `alpha` / `alpha_unflt`)
 * waldo_flow_ctx_warp_fwd (lvd.py:784-818), M = B*Tc*Tp:
 *   flow_lr  (M,L,2,H,W)             per-layer flow warped to image space (low resolution)
 *   isobj_lr (M,L-1,H,W) or NULL     warped ones of the objects (ghost mask, `allow_ghost` = NULL)
 *   a01      (B*Tw,L,Hd,Wd)          from waldo_flow_ctx_alpha_fwd
 *   ctx_ts   (B,Tc,Tp) int64, pred_ts (Tp) int64, occ (B,T,L,L)
 *   flow (M,2,Hd,Wd), alpha_ctx (M,L,Hd,Wd) in [-1,1], disocc (M,Hd,Wd)      outputs
 * ------------------------------------------------------------------------------------- */
/* layer_bits, optional (NULL to skip), (B*Tw, Hd, ceil(Wd / 64)) uint32: bit l of word (n, y, s) is set when a01 of
 * layer l is non-zero (or NaN) somewhere in columns [64 s, 64 s + 64) of row y of frame n -- the first pass's map of
 * where each layer IS, for the second pass on the path that has no ghost mask (Warper.grid_to_flow, lvd.py:602-705):
 * waldo_flow_ctx_warp_* skip, per tile, the layers that are absent from every segment the tile's samples can reach. */
int waldo_flow_ctx_alpha_fwd(const float* alpha_lr, const float* input, const float* dist,
                             const float* occ, float* a01, float* alpha_out, unsigned* layer_bits, int B, int T,
                             int Tw, int L, int Nl, int C, int chan_off, int H, int W, int scale,
                             waldo_stream_t stream);
/* alpha_max (M,Hd,Wd), optional (NULL to skip): max over the layers of alpha_ctx -- what
 * Synthesizer.predict's disocclusion test takes from it (models/synthesizer.py:447, `alpha_ctx.max(dim=3)[0]`:
 * a pass over the largest tensor but one of the pipeline, here a by-product of writing it). */
/* ctx_ts must lie in [0, Tw), pred_ts in [0, T): violations are reported in `status` ("Frame-index status" above).
 * layer_bits, optional: waldo_flow_ctx_alpha_fwd's by-product for the SAME a01 (NULL: every layer is sampled in every
 * pixel unless the ghost mask excludes it).  The values do not depend on it. */
int waldo_flow_ctx_warp_fwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                            const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ,
                            float* flow, float* alpha_ctx, float* disocc, float* alpha_max,
                            const unsigned* layer_bits, int* status, int B, int T, int Tw, int Tc, int Tp, int L,
                            int H, int W, int scale, waldo_stream_t stream);
/* The same pass for a caller that runs waldo_frame_warp_fuse_raw_fwd next (LVD.forward(mode="decode_output"),
 * lvd.py:141-153, without autograd): the composited context alphas are written straight into the slots they
 * occupy in A10's `raw` tensor, raw[b, tp, tc, C + l] (lvd.py:846: `raw_output = cat(output, alpha)`), instead of
 * into a tensor of their own that the frame warp would read and copy there (L planes read + L written per
 * (b, tc, tp) and full-resolution pixel: 29 % of that kernel's traffic at the Cityscapes recipe), and
 *   score (M,Hd,Wd) = sum_l (alpha_ctx_l + 1) / 2   (lvd.py:841), summed in layer order from the stored values,
 * comes out beside them.  raw (B,Tp,Tc',C+L,Hd,Wd) as A10 lays it out, Tc' = Tc or Tc + 1; only the alpha slots
 * of contexts 0 .. Tc-1 are written.  The reference's alpha_ctx (B,Tc,Tp,L,Hd,Wd) is a strided view of raw. */
int waldo_flow_ctx_warp_raw_fwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                                const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ, float* flow,
                                float* raw, float* score, float* disocc, float* alpha_max,
                                const unsigned* layer_bits, int* status, int B, int T, int Tw, int Tc, int Tp, int L,
                                int H, int W, int scale, int C, int Tcx, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A10: Warper.input_to_output (models/nets/lvd.py:830-853): warp of the context frames
 * by the composited flow and temporal fusion, incl. the include_self branch (lvd.py:842-845).
 *   input (B,T,C,Hd,Wd); flow (B,Tc,Tp,2,Hd,Wd); alpha (B,Tc,Tp,L,Hd,Wd) in [-1,1];
 *   ctx_ts (B,Tc,Tp) int64; Tc' = Tc + (include_self ? 1 : 0) <= 8; include_self needs Tp == T
 *   out (B,Tp,C+1,Hd,Wd)  fused frames, last channel = fused (2 score - 1)
 *   raw (B,Tp,Tc',C+L,Hd,Wd)  per-context warped frames and alphas (the WIF input, wif.py:21), stored
 *                             with the PREDICTED frame ahead of the context: the reference's
 *                             (B,Tc',Tp,...) tensor is the view raw.permute(0,2,1,3,4,5) (strides, no
 *                             copy), and WIF.forward's own permute + contiguous (wif.py:39) -- a copy of
 *                             the largest tensor of the pipeline -- finds it already in place
 * ctx_ts must lie in [0, T): violations are reported in `status` ("Frame-index status" above).
 * Any Wd >= 2 and any alignment give the same bits; with Wd % 4 == 0, a 16-byte aligned `input` and Tc <= 4 the
 * contexts' footprints are staged in LDS (16-byte loads) instead of gathered tap by tap.
 * ------------------------------------------------------------------------------------- */
int waldo_frame_warp_fuse_fwd(const float* input, const float* flow, const float* alpha,
                              const int64_t* ctx_ts, float* out, float* raw, int* status, int B, int T, int Tc,
                              int Tp, int C, int L, int Hd, int Wd, int include_self, float eps,
                              waldo_stream_t stream);
/* A10 behind waldo_flow_ctx_warp_raw_fwd: the alpha slots of `raw` are filled already and `score` (B,Tc,Tp,Hd,Wd)
 * holds their per-context sums; this call reads one score plane per context instead of L alpha planes, writes the C
 * warped channels of every context (and, with include_self, the whole self slot) and `out`.  Same values, bit
 * for bit, as waldo_frame_warp_fuse_fwd on the contiguous alpha tensor. */
int waldo_frame_warp_fuse_raw_fwd(const float* input, const float* flow, const float* score,
                                  const int64_t* ctx_ts, float* out, float* raw, int* status, int B, int T, int Tc,
                                  int Tp, int C, int L, int Hd, int Wd, int include_self, float eps,
                                  waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Backward of A9 / A10 (csrc/flow_ctx_bwd.hip): the reference's live backward path in LVD training
 * (models/synthesizer.py:841 through models/nets/lvd.py:602-853).  Shapes as the forward entry points.
 *
 * waldo_flow_ctx_alpha_bwd:  grad_a01, grad_alpha_out (B*Tw,L,Hd,Wd) = d loss / d a01 and d loss / d alpha_out, the
 *   two outputs of the forward (alpha_out = 2 a01 - 1: the kernel reads grad_a01 + 2 grad_alpha_out; either may be
 *   NULL = zero, not both);  grad_alpha_lr (B*Tw,L,H,W) OVERWRITTEN;
 *   grad_dist (B,L-1,Nl) and grad_occ (B,T,L,L) must be ZERO-FILLED (one float atomic per workgroup
 *   and entry; either may be NULL);  workspace: B*Tw*L*Hd*Wd floats when scale > 1 (the gradient at
 *   the HD raster, transposed-upsampled by a gather pass), unused at scale 1.
 * waldo_flow_ctx_warp_bwd:  grad_flow (M,2,Hd,Wd), grad_alpha_ctx (M,L,Hd,Wd), grad_disocc (M,Hd,Wd),
 *   any may be NULL (zero);  grad_flow_lr (M,L,2,H,W) OVERWRITTEN;  grad_a01 (B*Tw,L,Hd,Wd) must be
 *   ZERO-FILLED (bilinear splat with float atomics, as F.grid_sample's backward; NULL to skip);
 *   grad_occ as above;  workspace: M*L*2*Hd*Wd floats when scale > 1.  The ghost mask (isobj_lr > 0.9)
 *   and the frame indices carry no gradient.
 * waldo_frame_warp_fuse_bwd:  grad_out (B,Tp,C+1,Hd,Wd), grad_raw (B,Tc',Tp,C+L,Hd,Wd), either may be
 *   NULL;  grad_flow (B,Tc,Tp,2,Hd,Wd) and grad_alpha (B,Tc,Tp,L,Hd,Wd) OVERWRITTEN.  The frames
 *   (`input`) are data: no gradient is produced for them.
 * ------------------------------------------------------------------------------------- */
int waldo_flow_ctx_alpha_bwd(const float* alpha_lr, const float* input, const float* dist,
                             const float* occ, const float* grad_a01, const float* grad_alpha_out,
                             float* grad_alpha_lr, float* grad_dist, float* grad_occ, float* workspace, int B,
                             int T, int Tw, int L, int Nl, int C, int chan_off, int H, int W, int scale,
                             waldo_stream_t stream);
int waldo_flow_ctx_warp_bwd(const float* flow_lr, const float* isobj_lr, const float* a01,
                            const int64_t* ctx_ts, const int64_t* pred_ts, const float* occ,
                            const float* grad_flow, const float* grad_alpha_ctx,
                            const float* grad_disocc, float* grad_flow_lr, float* grad_a01,
                            float* grad_occ, float* workspace, int B, int T, int Tw, int Tc, int Tp,
                            int L, int H, int W, int scale, waldo_stream_t stream);
int waldo_frame_warp_fuse_bwd(const float* input, const float* flow, const float* alpha,
                              const int64_t* ctx_ts, const float* grad_out, const float* grad_raw,
                              float* grad_flow, float* grad_alpha, int B, int T, int Tc, int Tp, int C,
                              int L, int Hd, int Wd, int include_self, float eps,
                              waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A9, low-resolution part: the class distribution of every object for the layout filter
 * (models/nets/lvd.py:624-634 and 731-746; csrc/lyt_dist.hip).  Per batch item, with x running over
 * the Tw frames and H x W pixels:
 *   win[o][x]  = (alpha[o][x] + 1e-6) * sum_n (cls[o][n] + min_cls) * softmax_n(lyt[.][x])[n]
 *                (the sum is 1 when cls is NULL: the reference's `cls is None` / no weight_cls case)
 *   mean[o][n] = sum_x win[o][x] lyt[n][x] / sum_x win[o][x];   dist[o][.] = softmax_n(mean[o][.])
 *   alpha  (B,Tw,layers,H,W)   projected alpha in [0,1]; the objects are layers first_obj ..
 *                              first_obj + No - 1 = layers - 1 (layer 0 is the background)
 *   lyt    layout logits at the alpha's raster: plane n of frame t of item b starts at
 *          lyt + b * lyt_batch_stride + t * lyt_frame_stride + n * H * W (elements), so a channel
 *          slice of the (B,T,3+Nl,H,W) input is passed without a copy
 *   cls    (B,No,Nl) or NULL;   dist, mean (B,No,Nl) and total (B,No) out (mean / total are what the
 *          backward needs);   workspace: waldo_lyt_dist_workspace_bytes(), shared by both directions
 * Backward: grad_dist (B,No,Nl) -> grad_alpha (B,Tw,layers,H,W) OVERWRITTEN (zero for the layers in
 * front of the objects) and grad_cls (B,No,Nl) OVERWRITTEN (NULL exactly when cls is).  The layout
 * logits are data (no gradient).  No atomics: per-workgroup partial sums added in workgroup order.
 * No, Nl <= 32.
 * ------------------------------------------------------------------------------------- */
int64_t waldo_lyt_dist_workspace_bytes(int64_t B, int Tw, int No, int Nl, int H, int W);
int waldo_lyt_dist_fwd(const float* alpha, const float* lyt, int64_t lyt_batch_stride,
                       int64_t lyt_frame_stride, const float* cls, float min_cls, float* dist,
                       float* mean, float* total, float* workspace, int64_t B, int Tw, int layers,
                       int first_obj, int No, int Nl, int H, int W, waldo_stream_t stream);
int waldo_lyt_dist_bwd(const float* grad_dist, const float* alpha, const float* lyt,
                       int64_t lyt_batch_stride, int64_t lyt_frame_stride, const float* cls,
                       float min_cls, const float* dist, const float* mean, const float* total,
                       float* grad_alpha, float* grad_cls, float* workspace, int64_t B, int Tw,
                       int layers, int first_obj, int No, int Nl, int H, int W,
                       waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused hot path (BASELINE.json metric): TPS grid (A2) -> bilinear warp of every 4-channel
 * layer (A4) -> LVD.reduce_comp (A6, lvd.py:100-114) in ONE launch.
 *   layers   (F,L,4,H,W) in [-1,1]   channel 3 = alpha; the alpha of layer 0 is taken as +1
 *   basis_t  (K3,H*W)                TPS basis of the output raster (shared by all frames)
 *   mapping  (F*L,K3,2)              from waldo_tps_mapping_fwd
 *   occ      (F,L,L)
 *   rgb      (F,3,H,W)  out, in [-1,1]
 *   alpha    (F,L,H,W)  out, composited alpha in [-1,1]; may be NULL (not written)
 * K3 <= 32, L <= waldo_max_layers().
 * Padding: every layer is sampled as grid_sample(x + delta, grid) - delta (A4; lvd.py:548,559).
 * delta = 0: taps outside a layer contribute 0 (zeros padding on the raw values: alpha 0.5 / grey
 * after the (x + 1) / 2 of reduce_comp) -- the BASELINE pipeline.  delta = 1 is the default of the
 * reference's layer_to_output: out-of-range taps read -1 (alpha 0 / black).  The shift only acts
 * where a footprint leaves the layer; with delta = 0 the results have the same bits as before the
 * parameter existed.
 * Coordinates: the grid is evaluated directly in pixel units (the mapping column scaled by W/2 or
 * H/2 inside the kernels), which moves a sample position by rounding only (~1e-5 px at 512 px).
 * ------------------------------------------------------------------------------------- */
int waldo_warp_composite_fwd(const float* layers, const float* basis_t, const float* mapping,
                             const float* occ, float* rgb, float* alpha, int64_t F, int L, int H,
                             int W, int K3, float delta, waldo_stream_t stream);
/* The same forward from the control points: mapping = inverse_kernel (N+3,N+3) @ [src_pts (F*L,N,2); 0]
 * (A2, warp.py:52-53) is computed inside the kernel (same fma order as waldo_tps_mapping_fwd: the
 * outputs have the same bits as waldo_tps_mapping_fwd + waldo_warp_composite_fwd), so a forward-only
 * call is ONE launch.  Served shapes: waldo_warp_composite_pts_supported() != 0 (N = 16, 4 | W);
 * otherwise EINVAL -- call the two-step path. */
int waldo_warp_composite_pts_supported(int L, int H, int W, int N);
int waldo_warp_composite_pts_fwd(const float* layers, const float* basis_t,
                                 const float* inverse_kernel, const float* src_pts,
                                 const float* occ, float* rgb, float* alpha, int64_t F, int L,
                                 int H, int W, int N, float delta, waldo_stream_t stream);
/* Backward of waldo_warp_composite_fwd.
 *   grad_rgb (F,3,H,W); grad_alpha (F,L,H,W) or NULL;
 *   workspace: scratch of at least waldo_warp_composite_bwd_workspace_bytes() bytes (256-byte
 *       aligned, contents irrelevant).  Non-NULL selects the two-kernel path (K3 == 19, L <= 17,
 *       4 | W: pixel kernel + per-source-tile gather, no global atomics, bitwise reproducible
 *       grad_layers / grad_mapping); NULL -- and every other shape, for which the size query
 *       returns 0 -- selects the generic per-tap-atomics kernel.
 *   Precision of grad_layers on the two-kernel path: a layer's gradient is summed in 32-bit FIXED
 *       POINT per 8x16-texel sub-block (of the 32x64-texel tile a workgroup owns), one power-of-two
 *       quantum for the sub-block's three colour planes and one for its alpha plane, chosen from an
 *       upper bound of the sub-block's sums so that nothing can overflow: quantum ~ 2^-17 of the
 *       largest possible sum of the group in that sub-block.  The error of a texel is therefore
 *       ABSOLUTE per sub-block and group (a few quanta), not relative to the texel: gradients several
 *       orders of magnitude below the largest ones of their 8x16 neighbourhood lose relative
 *       precision (the generic path keeps fp32 relative precision); a region of small gradients
 *       next to a region of large ones keeps its own.  An infinity or NaN among the contributions
 *       that can reach a tile turns that whole tile (all four planes) into NaN -- never into finite
 *       garbage.
 *   grad_layers (F,L,4,H,W): with a workspace it is OVERWRITTEN (every texel written once);
 *       without, it must be ZERO-FILLED by the caller (accumulated with float atomics);
 *   grad_mapping (F*L,K3,2): must be ZERO-FILLED by the caller (accumulated into); NULL to skip;
 *   grad_occ (F,L,L): must be ZERO-FILLED by the caller (float atomics); NULL to skip. */
int64_t waldo_warp_composite_bwd_workspace_bytes(int64_t F, int L, int H, int W, int K3);
int waldo_warp_composite_bwd(const float* layers, const float* basis_t, const float* mapping,
                             const float* occ, const float* grad_rgb, const float* grad_alpha,
                             float* grad_layers, float* grad_mapping, float* grad_occ,
                             void* workspace, int64_t workspace_bytes, int64_t F, int L, int H,
                             int W, int K3, float delta, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A12. Fusion epilogue of WIF.forward with ii_score (models/nets/wif.py:49-54).
 *   vid (N,Tc,C,HW)  the UNet INPUT after the permute of wif.py:39 (N = B*T), C >= 5
 *   net (N,Tc,Co,HW) the UNet output, Co >= 4 (channels 0-2 residual, 3 score)
 *   out (N,3,HW) = sum_tc (sigmoid(vid_4 + 5) * vid_{0..2} + net_{0..2}) * softmax_tc(net_3)
 *   ab == 0 drops the sigmoid term (opt.ii_ab false, wif.py:53).
 * Backward: grad_vid (N,Tc,C,HW) / grad_net (N,Tc,Co,HW) are OVERWRITTEN (zero on the channels
 * the epilogue does not read); either may be NULL.
 * ------------------------------------------------------------------------------------- */
int waldo_wif_fuse_fwd(const float* vid, const float* net, float* out, int64_t N, int Tc, int C,
                       int Co, int64_t HW, int ab, waldo_stream_t stream);
int waldo_wif_fuse_bwd(const float* vid, const float* net, const float* out, const float* grad_out,
                       float* grad_vid, float* grad_net, int64_t N, int Tc, int C, int Co,
                       int64_t HW, int ab, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f3. The mask dilation of WIF.inpaint: expand() (tools/utils.py:300-323; called at models/nets/wif.py:77,
 * 108-112, 204).  `num` rounds; a round is the steps named in `steps` (bit 0 south, 1 north, 2 east, 3 west -- 15 =
 * the reference's dir=None) taken IN THAT ORDER over the whole plane, each seeing the result of the one before:
 *     south: m[y,x] = max(m[y,x], alpha*m[y-1,x])   north: ... alpha*m[y+1,x]   east: ... alpha*m[y,x-1]   west: ... alpha*m[y,x+1]
 * with torch.maximum's NaN rule.  soft == 0: the same recurrence on (mask != 0) with alpha = 1 (the reference's
 * bool branch), written as 0.0 / 1.0.  The recurrence is executed literally inside a workgroup's LDS tile, so the
 * result has the framework's bits for every input.
 *   mask (planes,H,W) -> out (planes,H,W); mask is never written; out / scratch must not alias mask or each other;
 *   scratch (planes,H,W) is needed only when num > 30 (passes of 30 rounds ping-pong through it), else NULL.
 * The masks are data: no backward.
 * ------------------------------------------------------------------------------------- */
int waldo_mask_expand_fwd(const float* mask, float* out, float* scratch, int64_t planes, int H, int W, int num,
                          int steps, int soft, float alpha, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f3. The point-in-polygon test of WIF.inpaint (models/nets/wif.py:228-235: matplotlib.path.Path(corners)
 * .contains_points(pts) -- radius 0, no transform -- for the region an object enters the frame from, wif.py:140-160).
 * matplotlib's own test (the crossings-multiply test over the path's vertices, closed back to the first, in double
 * precision) restated operation by operation, so that a point ON an edge gets matplotlib's answer.
 *   pts (N,2) f32 on the device (x, y);  corners_host (K,2) f64 in HOST memory, read before the call returns
 *   (3 <= K <= 16; fewer than three corners: nothing is inside, as in matplotlib);  out (N) f32: 1.0 inside, 0.0 outside
 *   (a point with a non-finite coordinate is outside).
 * ------------------------------------------------------------------------------------- */
int waldo_points_in_polygon_fwd(const float* pts, const double* corners_host, int K, float* out, int64_t N,
                                waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f3. The per-frame propagation step of WIF.inpaint (models/nets/wif.py:179-214) as one launch: the inpainted reference
 * background warped along the background flow, up to two objects entering through the image border pasted over it, the
 * shadow mask, the fill of the frame's holes, and the inputs of the external inpainter (csrc/inpaint_ops.hip spells the
 * arithmetic; every operation in the framework's order: the bits of the composition it replaces).  All planes (B,.,H,W)
 * f32 contiguous:
 *   flow (B,H,W,2) grid-unit flow of this frame, ident (H,W,2) the identity grid (WIF.src_grid_hd);
 *   ref_img (B,3,H,W), ref_mask (B,1,H,W), shadow (B,1,H,W) or NULL (opt.use_shadows off);
 *   enter_region[k] (B,1,H,W), enter_look[k] (B,3,H,W), enter_flow[k] (B,H,W,2): HOST arrays of n_enter <= 2 device
 *   pointers (read before the call returns);  img (B,3,H,W), todo (B,1,H,W), obj (B,1,H,W) the frame, its hole mask and
 *   its object mask;
 *   out: img_out (B,3,H,W), todo_out (B,1,H,W), inp_mask (B,1,H,W) = 1 - (1 - todo)(1 - obj), and -- fix_mask == 0 --
 *   inp_img (B,3,H,W) = (1 - todo)(1 - obj) img  (fix_mask != 0: the inpainter takes img_out; inp_img may be NULL).
 * waldo_inpaint_blend_fwd: out (B,3,HW) = (1 - todo) img + todo fill  (wif.py:214).
 * ------------------------------------------------------------------------------------- */
int waldo_inpaint_propagate_fwd(const float* flow, const float* ident, const float* ref_img, const float* ref_mask,
                                const float* shadow, const float* const* enter_region, const float* const* enter_look,
                                const float* const* enter_flow, int n_enter, const float* img, const float* todo,
                                const float* obj, float* img_out, float* todo_out, float* inp_img, float* inp_mask,
                                int64_t B, int H, int W, int soft_shadow, int fix_mask, waldo_stream_t stream);
int waldo_inpaint_blend_fwd(const float* img, const float* todo, const float* fill, float* out, int64_t B, int64_t HW,
                            waldo_stream_t stream);
/* The hole and object masks of the predicted frames (wif.py:60-75) in one pass over alpha_ctx (B,Tc,Tp,L,H,W) f32 --
 * given by its element strides over (b, tc, tp, l); the (H,W) planes contiguous, HW = H*W:
 *   cover = sum_l (alpha_ctx + 1) / 2,  obj = the same over l >= 1  (summed as the framework's reduction sums a short
 *   strided dimension: four accumulators j % 4, combined in order); last_only != 0: the last context's, else the maximum
 *   over the contexts (a NaN wins);  mask (B,Tp,HW) = (1 - cover) > thresh,  obj_mask (B,Tp,HW) = obj > 0.9, as 0 / 1.
 *   thresh: 0.1 with opt.fix_thresh, 0.9 without (wif.py:70-73). */
int waldo_inpaint_holes_fwd(const float* alpha_ctx, int64_t stride_b, int64_t stride_tc, int64_t stride_tp,
                            int64_t stride_l, float* mask, float* obj_mask, int64_t B, int Tc, int Tp, int L,
                            int64_t HW, int last_only, float thresh, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A8. gather_time (models/nets/lvd.py:462-467) with the frame arithmetic of the flow synthesis
 * (lvd.py:660-668, 780-787) on a clip's grids x (B,T,P,2) -- P pairs per frame:
 *   subtract != 0:  out[b,tc,tp] = x[b, ctx_ts[b,tc,tp]] - x[b, pred_ts[tp]]   (layer-space flow)
 *   subtract == 0:  out[b,tc,tp] = x[b, ctx_ts[b,tc,tp]]  (gather_time), or, with ctx_ts == NULL,
 *                   x[b, pred_ts[tp]] repeated over the Tc contexts (the `[:, pred_ts].unsqueeze(1).expand`)
 *   HW > 0 (a divisor of P): out is written channel-first, (B,Tc,Tp,P/HW,2,HW) -- the
 *   permute(..., 6, 4, 5) + reshape of lvd.py:662-664; HW == 0: (B,Tc,Tp,P,2).
 * ctx_ts (B,Tc,Tp) / pred_ts (Tp) int64 on the device, valid in [0,T): clamped, and the forward reports a violation
 * in `status` ("Frame-index status" above).
 * Backward: grad_x (B,T,P,2) is OVERWRITTEN with the sum over the output frames that read each input
 * frame (a gather: deterministic, no atomics, no zero fill needed).
 * ------------------------------------------------------------------------------------- */
int waldo_time_gather_fwd(const float* x, const int64_t* ctx_ts, const int64_t* pred_ts, float* out,
                          int* status, int B, int T, int Tc, int Tp, int64_t P, int64_t HW, int subtract,
                          waldo_stream_t stream);
int waldo_time_gather_bwd(const float* grad_out, const int64_t* ctx_ts, const int64_t* pred_ts,
                          float* grad_x, int B, int T, int Tc, int Tp, int64_t P, int64_t HW,
                          int subtract, waldo_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A7. scale(input[:, :Tw, c0:], 1 / S) (models/nets/lvd.py:611 / 716 through scale(), lvd.py:175-179):
 * F.interpolate(bilinear, align_corners=False, scale_factor=1/S) of the channel slice, S a power of two >= 2
 * (1 / S exact).   input (B,T,C,H*S,W*S)  ->  out (B,Tw,C-c0,H,W); the same bits as F.interpolate's device kernel
 * (its CPU kernel associates the four terms differently: last-ulp differences).  The frames are data: no backward.
 * ------------------------------------------------------------------------------------- */
int waldo_downscale_frames_fwd(const float* input, float* out, int B, int T, int Tw, int C, int c0, int H,
                               int W, int S, waldo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WALDO_HIP_H */
