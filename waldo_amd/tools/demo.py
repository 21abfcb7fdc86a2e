"""BASELINE config C1 plumbing: WIF inference on a demo clip, reproducing the call order of the
reference's ``Synthesizer.predict`` (models/synthesizer.py:425-480) around the hot path.

    python -m waldo_amd.tools.demo --clip <dir of frame PNGs> --out <dir> [--dim 128 --num-obj 3]

What is real: the frames and layout maps of the clip (``waldo_amd.tools.io.load_clip``), every
warp / composite / fusion step (the HIP kernels behind ``nets.lvd`` / ``nets.flp`` / ``nets.wif``),
the tensor plumbing between them (``ctx_ts`` / ``pred_ts`` construction, the ``last_n_ctx``
window, the disocclusion bookkeeping, the future-prediction branch).  What is synthetic: the
outputs of the networks that are outside this path (layer / pose estimators,
object decoder conv stack, pose generator, UNet) -- seeded stand-ins with the right shapes:
smooth object blobs for the decoder's raw image, small smooth motions for the pose heads, and a
UNet stand-in that predicts a zero residual and uniform scores.  The same stand-ins feed the CPU
restatement of the chain in tests/test_demo.py, so it is checked end to end, not just exercised.
"""
import argparse
import os
import types

import torch
import torch.nn as nn

from .. import functional as WF
from ..nets import flp
from ..nets.lvd import Warper, decode_output, decoder_tail, estimate_alpha_grid_occ
from ..nets.wif import WIF
from . import io as wio
from .utils import get_grid


def demo_opt(dim=128, aspect_ratio=1.0, num_obj=3, num_lyt=20, **over):
    """The option fields the path reads (Warper: models/nets/lvd.py:472-499; WIF: ii_score, ii_ab),
    at the C1 size: 128 x 128, 4 layers (3 objects + background)."""
    d = dict(latent_shape=[4, 4], obj_shape=[2, 2], time_dropout=0.0, num_obj=num_obj, patch_size=8,
             scale_factor=2, dim=dim, aspect_ratio=aspect_ratio, load_dim=0, num_perm_grid=1,
             normalize_alpha=False, use_lyt_filtering=True, use_lyt_opacity=False, weight_cls=True,
             min_cls=0.05, include_self=False, no_filter=False, allow_ghost=False, num_lyt=num_lyt,
             ii_score=True, ii_ab=True, last_n_ctx=0, no_future=False, pad_obj_alpha=2, use_inpainter=False)
    d.update(over)
    return types.SimpleNamespace(**d)


class UniformFusionUNet(nn.Module):
    """UNet stand-in: zero colour residual (channels 0-2), equal scores (channel 3) -- the fusion of
    wif.py:49-54 then averages the warped context frames with their sigmoid(alpha + 5) weights.  The zeros are
    made once per shape and handed out again (nobody writes to them): the stand-in for a network OUTSIDE the path
    should not put half a millisecond of fills into the pipeline's timing."""

    def __init__(self):
        super().__init__()
        self._zeros = {}

    def forward(self, x):
        key = (x.shape[0], tuple(x.shape[-2:]), x.device, x.dtype)
        z = self._zeros.get(key)
        if z is None:
            z = self._zeros[key] = x.new_zeros(x.shape[0], 4, *x.shape[-2:])
        return z


# Background motion of the stand-ins, as (drift of the 2 x 2 affine part, of the translation, of the per-point
# deltas) per frame step, in the pose heads' units.
#   "wild"        every one of the 6 + 2 Lb background pose values drifts independently (0.02 randn per frame):
#                 with Lb = 128 lattice points 1/8 of the frame apart, neighbouring control points end up a whole
#                 lattice cell apart after ten frames -- a folded warp (local stretch |d ix / dx| of the composited
#                 flow at 512 x 1024: median 1, 90 % quantile 7, 99 % quantile 30).  The stress case; rounds 1-3
#                 benchmarked the C4 / C5 pipelines on it.
#   "calibrated"  the same amplitudes for the rigid part and a tenth of them for the per-point deltas: the local
#                 stretch of the warp stays within what optical flow of real driving footage shows (RAFT flows of
#                 the reference's demo clips, datasets/demo_cityscapes/*_raft_128: |d ix / dx| 10 % / 99 % quantiles
#                 0.92 / 1.04 per frame step, mean |flow| 2.4 px of 256 and max 16 px per step; measured with
#                 tools_dev/flo_stats.py), accumulated over the 4 .. 13 frame steps a clip spans.
BG_MOTION = {"wild": (0.02, 0.02, 0.02), "calibrated": (0.01, 0.02, 0.002)}


def synthetic_network_outputs(opt, b, t, ctx_len, seed=0, device="cpu", motion="wild"):
    """Seeded stand-ins for what the networks outside the path would hand over, on ``device``."""
    g = torch.Generator().manual_seed(seed)
    no, nl = opt.num_obj, opt.num_lyt
    lo = opt.obj_shape[0] * opt.obj_shape[1]
    lb = opt.latent_shape[0] * opt.latent_shape[1]
    ho, wo = opt.obj_shape[0] * opt.patch_size, opt.obj_shape[1] * opt.patch_size
    # object decoder raw image: one soft blob per object (positive inside -> alpha ~ +1 after tanh)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, ho), torch.linspace(-1, 1, wo), indexing="ij")
    rad = 0.5 + 0.3 * torch.rand(b * no, 1, 1, 1, generator=g)
    raw = 4.0 * (rad - (xx ** 2 + yy ** 2).sqrt().view(1, 1, ho, wo) / 1.0)
    # pose heads (after tanh): per-object placement + a slow drift over time, small point deltas
    base = 0.6 * (torch.rand(b, 1, no, 6 + 2 * lo, generator=g) * 2 - 1)
    drift = 0.05 * torch.randn(b, 1, no, 6 + 2 * lo, generator=g) * torch.arange(t).view(1, t, 1, 1)
    obj = base + drift
    obj[..., 6:] = 0.1 * obj[..., 6:]
    obj[..., [0, 3]] = -0.2 + 0.2 * obj[..., [0, 3]]  # scales around bias_obj - 0.2
    obj[..., [1, 2]] = 0.2 * obj[..., [1, 2]]        # small shear
    bg = 0.02 * torch.randn(b, 1, 1, 6 + 2 * lb, generator=g) * torch.arange(t).view(1, t, 1, 1)
    lin, shift, pts = BG_MOTION[motion]  # ("wild" multiplies by 1: the same numbers as ever)
    bg[..., [0, 1, 2, 3]] *= lin / 0.02
    bg[..., [4, 5]] *= shift / 0.02
    bg[..., 6:] *= pts / 0.02
    occ_score = torch.randn(b, t, no, generator=g)
    cls = torch.softmax(2.0 * torch.randn(b, no, nl, generator=g), dim=-1)
    out = dict(raw=raw, pred_obj_pose=obj.reshape(b * t, no, -1), pred_bg_pose=bg.reshape(b * t, 1, -1),
               occ_score=occ_score, cls=cls)
    return {k: v.to(device) for k, v in out.items()}


_CONSTANTS = {}


def _cached(kind, opt, device, make):
    """The path's constant tensors (the pose heads' buffers, the padding mask, the background alpha) live on the
    device ONCE per option set, as a model's registered buffers do: made inside predict() they were ten small
    host-to-device copies per call -- each one a point where the host stops queueing kernels -- and kept the call
    from being captured into a HIP graph."""
    key = (kind, str(device), tuple(opt.obj_shape), tuple(opt.latent_shape), opt.patch_size, opt.scale_factor,
           opt.pad_obj_alpha, opt.dim, float(opt.aspect_ratio))
    if key not in _CONSTANTS:
        _CONSTANTS[key] = make()
    return _CONSTANTS[key]


def pose_buffers(opt, device):
    """Buffers of the pose heads (models/nets/flp.py:119-123) at the demo's option values."""
    return _cached("pose", opt, device, lambda: _pose_buffers(opt, device))


def _pose_buffers(opt, device):
    lo = opt.obj_shape[0] * opt.obj_shape[1]
    lb = opt.latent_shape[0] * opt.latent_shape[1]
    return dict(
        tgt_pts_obj=get_grid(*opt.obj_shape).view(1, 1, lo, 2).to(device),
        tgt_pts_bg=get_grid(*opt.latent_shape).view(1, 1, lb, 2).to(device),
        bias_obj=torch.tensor([[[0.5, 0.0, 0.0, opt.aspect_ratio * 0.5, 0.0, 0.0]]], device=device),
        mul_obj=torch.tensor([[[0.5, 0.5, 0.5, 0.5, 1.0, 1.0]]], device=device),
        bias_bg=torch.tensor([[[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]], device=device))


def obj_alpha_mask(opt, device):
    """lvd.py:27-34: zero border of ``pad_obj_alpha`` decoder pixels around the object canvas."""
    return _cached("mask", opt, device, lambda: _obj_alpha_mask(opt, device))


def _obj_alpha_mask(opt, device):
    ho = opt.obj_shape[0] * opt.patch_size * opt.scale_factor
    wo = opt.obj_shape[1] * opt.patch_size * opt.scale_factor
    po = opt.pad_obj_alpha * opt.scale_factor
    m = torch.ones(ho, wo, device=device)
    if po > 0:
        m[:po] = 0
        m[:, :po] = 0
        m[-po:] = 0
        m[:, -po:] = 0
    return m.view(1, 1, 1, ho, wo)


def _cached_by(key, make):
    if key not in _CONSTANTS:
        _CONSTANTS[key] = make()
    return _CONSTANTS[key]


def _cached_index(values, device):
    """A small index tensor on the device, made once (a fresh host-to-device copy per call would stall the
    launch queue: see _cached)."""
    key = ("index", str(device), tuple(values))
    if key not in _CONSTANTS:
        _CONSTANTS[key] = torch.tensor(list(values), device=device, dtype=torch.int64)
    return _CONSTANTS[key]


def _block_net(opt, net, b, t, b0, b1, sel, device):
    """The stand-ins' outputs for clips b0:b1 at the frames ``sel`` (ascending) of the clip's T."""
    if (b0, b1) == (0, b) and list(sel) == list(range(t)):
        return net
    no = opt.num_obj
    idx = _cached_index(sel, device)

    def frames(x):  # (B * T, ...) -> (nb * T', ...)
        x = x.view(b, t, *x.shape[1:])[b0:b1]
        return x.index_select(1, idx).reshape(-1, *x.shape[2:])

    return dict(raw=net["raw"][b0 * no:b1 * no], pred_obj_pose=frames(net["pred_obj_pose"]),
                pred_bg_pose=frames(net["pred_bg_pose"]), occ_score=net["occ_score"][b0:b1].index_select(1, idx),
                cls=net["cls"][b0:b1])


class SharedContext:
    """What the decodes of one step compute from a clip's CONTEXT alone -- the object alphas (decoder tail + padding
    mask), the context frames' control points, grids, inverted grids and occlusion matrices, and the composited
    full-resolution context alphas (``Warper.context_products``) -- kept from the first decode for the next one.
    ``Synthesizer.predict`` runs them once per decode (synthesizer.py:434-445 and 470-472); its second decode receives
    the context poses "as they came in" (flp.py:275-290), so the products are the same tensors' functions: reused when
    the stand-ins' outputs and the input frames are the SAME tensor objects, unmodified (identity + version counter),
    computed afresh otherwise.  Every value depends on its own (b, t) frame, so a decode that reuses them has the bits
    of one that does not (tests/test_gpu_pipeline.py)."""

    def __init__(self):
        self.key, self.value, self.held = None, None, ()

    @staticmethod
    def _key(tensors):
        return tuple((id(x), None if x.is_inference() else x._version) for x in tensors)

    def get(self, tensors, make):
        key = self._key(tensors)
        if self.key != key or any(a is not b for a, b in zip(self.held, tensors)):
            self.value, self.key, self.held = make(), key, tuple(tensors)  # (held: ids are not recycled meanwhile)
        return self.value


def _points_grids_occ(opt, warper, net, nb, nt):
    """Pose heads' affine (flp.py:259-273), Warper.forward and compute_occ for the ``nt`` frames of ``net``."""
    from ..nets.lvd import compute_occ
    no = opt.num_obj
    lo = opt.obj_shape[0] * opt.obj_shape[1]
    lb = opt.latent_shape[0] * opt.latent_shape[1]
    buf = pose_buffers(opt, net["pred_obj_pose"].device)
    obj_pose = flp.obj_pose_to_points(net["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
    bg_pose = flp.bg_pose_to_points(net["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"])
    grid = warper(obj_pose.view(nb, nt, no, lo, 2), bg_pose.view(nb, nt, 1, lb, 2))
    return grid, compute_occ(net["occ_score"])


def _decode_block(opt, warper, wif, real_input, net, ctx_len, nb, sel, where, shared=None, shared_key=None,
                  out_alpha=None):
    """One decode of predict() (estimate_alpha_grid_occ -> decode_output -> disocclusion test -> WIF fusion,
    synthesizer.py:434-460 / 464-484) for ``nb`` clips on the compact time axis ``sel`` (frame numbers: the context
    frames 0 .. ctx_len - 1, then the other frames whose poses the decode needs -- a frame may stand there twice, with
    the reconstruction's and with the prediction's poses), producing the units at the positions ``where`` of that axis.
    ``net`` holds the stand-ins' outputs for exactly those clips and axis entries (_block_net); ``real_input``
    (nb, >= ctx_len, C, Hd, Wd) at least the context frames.  ``shared`` (a ``SharedContext``): what depends on the
    context alone is taken from / left for the other decodes of the step (``shared_key``: the tensors whose identity
    vouches for it, when ``net`` is a per-block copy of them).  Returns (output (nb, n, 3, Hd, Wd), disocc
    (nb, n, 1, Hd, Wd), inpainted (nb, n, 3, Hd, Wd), flow (nb, Tc, n, 2, Hd, Wd)).  Every kernel of the chain works
    per (b, t) unit (the layout filter's class distribution per clip, over its context frames), so the bits of a frame
    do not depend on which other frames or clips are decoded beside it (tests/test_gpu_pipeline.py)."""
    no = opt.num_obj
    dev = real_input.device
    nt, n = len(sel), len(where)
    assert list(sel[:ctx_len]) == list(range(ctx_len)), "the compact time axis starts with the context frames"
    # `alpha` (2 a' - 1 on the context frames) is dropped by the reconstruction (synthesizer.py:445) and handed by the
    # prediction to net_ii.inpaint (synthesizer.py:472, 484), which reads it only with the inpainter on (wif.py:103):
    # without one the flow pass does not write it (as large as the a' it keeps)
    want_alpha = bool(getattr(opt, "use_inpainter", False))
    mask = obj_alpha_mask(opt, dev)
    bg_alpha = _cached("bg_alpha", opt, dev, lambda: torch.ones(1, 1, opt.dim, int(opt.dim * opt.aspect_ratio), device=dev))

    def frames_of(x, lo, hi):  # (nb * nt, ...) -> (nb * (hi - lo), ...)
        return x.view(nb, nt, *x.shape[1:])[:, lo:hi].reshape(-1, *x.shape[1:])

    def context_part():
        # decoder tail (lvd.py:245-254) + the alpha arithmetic of estimate_alpha_grid_occ (lvd.py:128-132); the context
        # frames' part of its Warper.forward and compute_occ (lvd.py:133-134); what decode_output computes from them
        obj_alpha = decoder_tail(net["raw"], init_bias=0.0, scale_factor=opt.scale_factor)
        obj_alpha = obj_alpha.view(nb, no, 1, *obj_alpha.shape[-2:])
        ctx_net = dict(pred_obj_pose=frames_of(net["pred_obj_pose"], 0, ctx_len),
                       pred_bg_pose=frames_of(net["pred_bg_pose"], 0, ctx_len), occ_score=net["occ_score"][:, :ctx_len])
        # (estimate_alpha_grid_occ with the context frames' poses: occ, masked object alphas, expanded bg alpha, grids)
        lo = opt.obj_shape[0] * opt.obj_shape[1]
        lb = opt.latent_shape[0] * opt.latent_shape[1]
        buf = pose_buffers(opt, dev)
        obj_pose = flp.obj_pose_to_points(ctx_net["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
        bg_pose = flp.bg_pose_to_points(ctx_net["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"])
        occ, obj_alpha, bga, grid = estimate_alpha_grid_occ(warper, obj_alpha, bg_alpha, obj_pose.view(nb, ctx_len, no, lo, 2),
                                                            bg_pose.view(nb, ctx_len, 1, lb, 2), ctx_net["occ_score"],
                                                            obj_alpha_mask=mask)
        prev = warper.return_alpha
        warper.return_alpha = want_alpha
        try:
            products = warper.context_products(real_input, grid, occ, obj_alpha, bga, net["cls"], ctx_len)
        finally:
            warper.return_alpha = prev
        return occ, obj_alpha, bga, grid, products

    if shared is not None:
        key = shared_key if shared_key is not None else (net["raw"], net["pred_obj_pose"], net["pred_bg_pose"],
                                                          net["occ_score"], net["cls"], real_input)
        occ_c, obj_alpha, bga, grid_c, products = shared.get(key, context_part)
        if nt > ctx_len:  # the frames beyond the context: their control points, grids and occlusion matrices
            new_net = dict(pred_obj_pose=frames_of(net["pred_obj_pose"], ctx_len, nt),
                           pred_bg_pose=frames_of(net["pred_bg_pose"], ctx_len, nt), occ_score=net["occ_score"][:, ctx_len:])
            grid_n, occ_n = _points_grids_occ(opt, warper, new_net, nb, nt - ctx_len)
            grid = [torch.cat([a, b], dim=1) for a, b in zip(grid_c, grid_n)]
            occ = torch.cat([occ_c, occ_n], dim=1)
        else:
            grid, occ = list(grid_c), occ_c
    else:
        # nothing to share with another decode: the whole axis in ONE call each of the pose heads' affine, Warper.forward
        # and compute_occ (estimate_alpha_grid_occ, lvd.py:126-135) -- no second set of launches for the context's grids,
        # nothing to concatenate; the context's products from the first ctx_len frames of those grids
        lo = opt.obj_shape[0] * opt.obj_shape[1]
        lb = opt.latent_shape[0] * opt.latent_shape[1]
        buf = pose_buffers(opt, dev)
        obj_alpha = decoder_tail(net["raw"], init_bias=0.0, scale_factor=opt.scale_factor)
        obj_alpha = obj_alpha.view(nb, no, 1, *obj_alpha.shape[-2:])
        obj_pose = flp.obj_pose_to_points(net["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
        bg_pose = flp.bg_pose_to_points(net["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"])
        occ, obj_alpha, bga, grid = estimate_alpha_grid_occ(warper, obj_alpha, bg_alpha, obj_pose.view(nb, nt, no, lo, 2),
                                                            bg_pose.view(nb, nt, 1, lb, 2), net["occ_score"],
                                                            obj_alpha_mask=mask)
        prev = warper.return_alpha
        warper.return_alpha = want_alpha
        try:
            products = warper.context_products(real_input, grid, occ, obj_alpha, bga, net["cls"], ctx_len)
        finally:
            warper.return_alpha = prev
    def make_ctx_ts():  # synthesizer.py:438-442
        ts = torch.arange(ctx_len, device=dev, dtype=torch.int64).view(1, -1, 1).expand(nb, -1, n)
        return WF.normalise_time_index(ts[:, -opt.last_n_ctx:] if opt.last_n_ctx > 0 else ts)

    # the frame indices of a decode, made ONCE per shape and kept (int64, contiguous: as the kernels take them): built
    # afresh per call (synthesizer.py:438-444 does) each would be a small host-to-device copy in the launch queue
    ctx_ts = _cached_by(("ctx_ts", str(dev), ctx_len, nb, n, opt.last_n_ctx), make_ctx_ts)
    pred_ts = _cached_index(where, dev)
    # decode_output + max_l alpha_ctx, which the fused flow pass produces as a by-product (the warper's switch and its
    # result are restored / cleared afterwards: no state is left on the module)
    prev, prev_alpha = warper.keep_alpha_ctx_max, warper.return_alpha
    warper.keep_alpha_ctx_max = True
    warper.return_alpha = want_alpha
    try:
        output, flow, _, alpha, _, raw_output, alpha_ctx = decode_output(warper, real_input, grid, occ, obj_alpha, bga,
                                                                         net["cls"], ctx_ts, pred_ts, ctx_products=products)
        mx = warper.alpha_ctx_max
    finally:
        warper.keep_alpha_ctx_max, warper.return_alpha = prev, prev_alpha
        warper.alpha_ctx_max = None
    # synthesizer.py:447-450: max / min over the contexts of max_l alpha_ctx, dmax[dmax - dmin > 1] = 0 -- one pass
    # over the flow pass's by-product (torch: aminmax, subtract, compare, masked_fill)
    if mx is None:  # == alpha_ctx.max(dim=3)[0] (NaN-propagating, as torch's), without the fused pass's by-product
        mx = alpha_ctx.amax(dim=3)
    dmax = WF.disocc_test(mx)
    if out_alpha is not None:
        out_alpha.append(alpha)
    return output[:, :, :3], dmax.unsqueeze(2), wif(raw_output), flow  # (wif: synthesizer.py:460)


@torch.no_grad()
def predict(opt, warper, wif, real_vid, real_lyt, net, ctx_len):
    """The hot-path part of Synthesizer.predict (models/synthesizer.py:434-480).  real_vid
    (B, T, 3, H, W), real_lyt (B, T, Nl, H, W); ``net`` = synthetic_network_outputs(...).
    Returns a dict of the tensors predict produces."""
    b, t = real_vid.shape[:2]
    every = list(range(t))
    out = {}
    # reconstruct video (synthesizer.py:436-445).  The reference concatenates all T frames and their layouts
    # (synthesizer.py:445: 2.7 GB per step at the Cityscapes recipe, 1 ms); `decode_output` with `restrict_to_ctx` and no
    # `include_self` reads the CONTEXT frames of it only (lvd.py:716-745, 837; Warper._clip_length): those alone are
    # concatenated -- 29 % of the bytes, the same results bit for bit (tests/test_demo.py compares with the restatement
    # that is handed all T frames)
    n_in = t if getattr(opt, "include_self", False) else ctx_len
    real_input = torch.cat([real_vid[:, :n_in], real_lyt[:, :n_in]], dim=2)
    if getattr(opt, "include_self", False) or not MERGE_DECODES or opt.no_future:
        # the reference's two calls, one after the other (include_self: every frame is a context of itself)
        shared = SharedContext() if not getattr(opt, "include_self", False) else None
        rec, dis, inp, _ = _decode_block(opt, warper, wif, real_input, net, ctx_len, b, every, every, shared=shared)
        out["rec_vid"], out["rec_disocc"], out["inp_rec_vid"] = rec, dis, inp
        if not opt.no_future:
            alpha = []
            pred, dis, inp, flow = _decode_block(opt, warper, wif, real_input, net, ctx_len, b, every,
                                                 list(range(ctx_len, t)), shared=shared, out_alpha=alpha)
    else:
        # ONE decode for the reconstruction's and the prediction's units (see decode_units): the context's products once,
        # every full-resolution pass launched once with T + Tp units per clip
        alpha = []
        (rec, dis, inp, _), (pred, dis_p, inp_p, flow) = decode_units(
            opt, warper, wif, real_input, net, ctx_len, b, t, 0, b, every, list(range(ctx_len, t)), out_alpha=alpha)
        out["rec_vid"], out["rec_disocc"], out["inp_rec_vid"] = rec, dis, inp
        dis, inp = dis_p, inp_p
    if not opt.no_future:
        # the pose generator (net_pg, outside the path) returns full-length pose sequences: the context
        # poses as they came in, the future ones predicted (flp.py:275-290) -- here the synthetic poses
        # of all T frames stand for them (synthesizer.py:464-472)
        if alpha and alpha[0] is not None:  # (with opt.use_inpainter: what the prediction hands to net_ii.inpaint, synthesizer.py:484)
            out["pred_alpha"] = alpha[0]
        out["pred_disocc"] = dis
        out["pred_flow"] = flow
        out["pred_vid"] = torch.cat([real_vid[:, :ctx_len], pred], dim=1)
        out["inp_pred_vid"] = torch.cat([real_vid[:, :ctx_len], inp], dim=1)
    return out


# Synthesizer.predict decodes twice: all T frames from the estimated poses (synthesizer.py:434-445), then the T - Tc future
# frames from the pose generator's (synthesizer.py:464-472).  Both decodes read the same context; their units are
# independent of each other, and the pose generator needs nothing the first decode produces -- so both sets of units can go
# through ONE decode_output on one time axis [context | the reconstruction's other frames | the prediction's frames]: every
# full-resolution pass is launched once with all units (a rank's share of a split job is launch-bound: ~150 launches
# become ~85), what depends on the context alone is computed once by construction.  The same bits per unit (every kernel
# of the chain works per (b, t) unit).  False: the reference's two calls, one after the other.
MERGE_DECODES = True


def decode_units(opt, warper, wif, real_input, net, ctx_len, b, t, b0, b1, rec_frames, pred_frames, out_alpha=None):
    """The reconstruction's frames ``rec_frames`` and the prediction's frames ``pred_frames`` (clip-relative frame numbers,
    ascending) of clips b0:b1 in ONE decode.  Returns the two 4-tuples of ``_decode_block`` (either may be None when its
    list is empty)."""
    dev = real_input.device
    rec_new = [f for f in rec_frames if f >= ctx_len]
    sel = list(range(ctx_len)) + rec_new + list(pred_frames)
    where = [f if f < ctx_len else ctx_len + rec_new.index(f) for f in rec_frames] + \
            [ctx_len + len(rec_new) + i for i in range(len(pred_frames))]
    blk = _block_net(opt, net, b, t, b0, b1, sel, dev)
    vid, dis, inp, flow = _decode_block(opt, warper, wif, real_input, blk, ctx_len, b1 - b0, sel, where, out_alpha=out_alpha)
    nr = len(rec_frames)

    def part(lo, hi):
        return (vid[:, lo:hi], dis[:, lo:hi], inp[:, lo:hi], flow[:, :, lo:hi]) if hi > lo else None

    return part(0, nr), part(nr, nr + len(pred_frames))


def rec_unit_order(t, ctx_len):
    """The order in which a clip's T RECONSTRUCTION units are dealt to the ranks (frame numbers).  Two things make units
    unequal: a frame beyond the context costs its rank a control-point grid and a grid inversion of its own (the context
    frames' every rank that shares the clip holds anyway), and a frame far from the context is warped further (larger
    footprint boxes in the frame warp: 1.34 against 1.23 ms per rank and step at the Cityscapes recipe on 8 ranks).
    Dealt in frame order, the rank that got frames 0 .. 6 of a 14-frame clip inverted 7 frames' grids and its neighbour
    11, and the neighbour also predicted the five LATE frames.  Here the context frames are spread evenly among the
    others, and the others are dealt in DESCENDING order: the rank that predicts a clip's early frames (the prediction is
    dealt in frame order) reconstructs its late ones."""
    nc = min(ctx_len, t)
    order, ctx, rest = [], list(range(nc)), list(range(t - 1, nc - 1, -1))
    for i in range(t):  # (a context frame wherever i * nc / t passes an integer)
        if ctx and ((i + 1) * nc // t > i * nc // t or not rest):
            order.append(ctx.pop(0))
        else:
            order.append(rest.pop(0))
    return order


def unit_segments(u0, u1, per_clip, order=None):
    """The units [u0, u1) of a phase's dealing order as runs of clips that decode the SAME frames: [(b0, b1, frames)] --
    a partial first clip, the whole clips in the middle as one batch, a partial last clip.  Unit u is clip u // per_clip,
    position u % per_clip of ``order`` (None: the natural order); ``frames`` are clip-relative unit numbers 0 ..
    per_clip - 1, ASCENDING within a segment -- the order in which ``predict_sharded`` returns them (``local_unit_ids``)."""
    order = list(range(per_clip)) if order is None else list(order)
    segs, u = [], u0
    while u < u1:
        b, i = divmod(u, per_clip)
        if i == 0 and u1 - u >= per_clip:
            nb = (u1 - u) // per_clip
            segs.append((b, b + nb, list(range(per_clip))))
            u += nb * per_clip
        else:
            i1 = min(per_clip, i + (u1 - u))
            segs.append((b, b + 1, sorted(order[i:i1])))
            u += i1 - i
    return segs


def phase_order(phase, t, ctx_len):
    """Dealing order of a phase's per-clip units: reconstruction -> ``rec_unit_order``; prediction -> frame order (every
    predicted frame costs the same)."""
    return rec_unit_order(t, ctx_len) if phase == "rec" else list(range(t - ctx_len))


def local_unit_ids(phase, b, t, ctx_len, rank, world):
    """Natural unit numbers (clip * per_clip + clip-relative unit) of the units ``predict_sharded`` returns for this rank
    and phase, in the order it returns them."""
    from ..dist import shard_range
    per_clip = t if phase == "rec" else t - ctx_len
    u0, u1 = shard_range(b * per_clip, rank, world)
    ids = []
    for b0, b1, frames in unit_segments(u0, u1, per_clip, phase_order(phase, t, ctx_len)):
        ids += [c * per_clip + f for c in range(b0, b1) for f in frames]
    return ids


UNIT_KEYS = {"rec": ("rec_vid", "rec_disocc", "inp_rec_vid"), "pred": ("pred_vid", "pred_disocc", "inp_pred_vid",
                                                                      "pred_flow")}


@torch.no_grad()
def predict_sharded(opt, warper, wif, real_vid, real_lyt, net, ctx_len, rank, world, phases=("rec", "pred")):
    """This rank's share of predict() when ONE job (B clips) is split over ``world`` ranks: the
    (b, t) output units of each decode -- B * T reconstructed frames, B * (T - Tc) predicted ones -- are dealt in
    contiguous blocks of the phase's dealing order (dist.shard_range over ``phase_order``: the reconstruction's spreads
    the context frames over the ranks that share a clip, so that every rank inverts the same number of new grids).  A
    rank keeps the context frames of the clips its block touches and their stand-in network outputs, runs the producers
    and Warper.forward for the context frames and ITS frames only, and decodes its block; what depends on a clip's
    context alone (its grids, the layout filter's class distribution, the first occlusion product) is computed ONCE per
    step and clip range (``SharedContext``) and replicated on every rank that shares the clip, which needs no collective.
    Returns {key: (units, C, Hd, Wd)} with the rank's units in the order of ``local_unit_ids``, for the keys of UNIT_KEYS
    (``pred_vid`` / ``inp_pred_vid``: the predicted frames only; ``pred_flow``: Tc * 2 channels); ``gather_predict`` puts
    the ranks' blocks together into predict()'s dict.  Reference: the data-parallel split of tools/engine.py:63-64, here
    over frames instead of clips so that one clip can use every GPU."""
    from ..dist import shard_range
    if opt.include_self:
        raise ValueError("predict_sharded: include_self appends the predicted frame itself as a context "
                         "(lvd.py:842-845): every rank would need every frame")
    b, t = real_vid.shape[:2]
    dev = real_vid.device
    hd, wd = real_vid.shape[-2:]
    out = {}
    inputs, shared = {}, {}  # clips b0:b1 -> cat of their context frames and layouts / their context's products
    job = (net["raw"], net["pred_obj_pose"], net["pred_bg_pose"], net["occ_score"], net["cls"])
    parts = {k: [] for ph in phases for k in UNIT_KEYS[ph] if not (ph == "pred" and opt.no_future)}

    def clip_input(b0, b1):
        if (b0, b1) not in inputs:
            inputs[(b0, b1)] = torch.cat([real_vid[b0:b1, :ctx_len], real_lyt[b0:b1, :ctx_len]], dim=2)
            shared[(b0, b1)] = SharedContext()
        return inputs[(b0, b1)]

    def keep(phase, nb, res):
        vid, dis, inp, flow = res
        n = vid.shape[1]
        parts[phase + "_vid"].append(vid.reshape(-1, 3, hd, wd))
        parts[phase + "_disocc"].append(dis.reshape(-1, 1, hd, wd))
        parts["inp_" + phase + "_vid"].append(inp.reshape(-1, 3, hd, wd))
        if phase == "pred":  # (nb, Tc, n, 2, Hd, Wd) -> units x (Tc * 2)
            parts["pred_flow"].append(flow.permute(0, 2, 1, 3, 4, 5).reshape(nb * n, -1, hd, wd))

    segs = {}
    for phase in phases:
        if phase == "pred" and opt.no_future:
            continue
        first = 0 if phase == "rec" else ctx_len
        per_clip = t - first
        u0, u1 = shard_range(b * per_clip, rank, world)
        segs[phase] = [(b0, b1, [first + f for f in units])
                       for b0, b1, units in unit_segments(u0, u1, per_clip, phase_order(phase, t, ctx_len))]
    if MERGE_DECODES and len(segs.get("rec", ())) == 1 and len(segs.get("pred", ())) == 1 and \
            segs["rec"][0][:2] == segs["pred"][0][:2]:
        # this rank's reconstruction and prediction units belong to the same clips: ONE decode for both (decode_units)
        b0, b1, rec_frames = segs["rec"][0]
        rec, pred = decode_units(opt, warper, wif, clip_input(b0, b1), net, ctx_len, b, t, b0, b1, rec_frames,
                                 segs["pred"][0][2])
        keep("rec", b1 - b0, rec)
        keep("pred", b1 - b0, pred)
    else:
        for phase, lst in segs.items():
            for b0, b1, frames in lst:
                sel = sorted(set(range(ctx_len)) | set(frames))
                blk = _block_net(opt, net, b, t, b0, b1, sel, dev)
                res = _decode_block(opt, warper, wif, clip_input(b0, b1), blk, ctx_len, b1 - b0, sel,
                                    [sel.index(f) for f in frames], shared=shared[(b0, b1)],
                                    shared_key=job + (inputs[(b0, b1)],))
                keep(phase, b1 - b0, res)
    for k, v in parts.items():
        if v:
            out[k] = v[0] if len(v) == 1 else torch.cat(v, dim=0)
        else:  # a rank past the end of a short job holds no unit
            ch = {"disocc": 1, "flow": 2 * (opt.last_n_ctx or ctx_len)}.get(k.split("_")[-1], 3)
            out[k] = real_vid.new_empty(0, ch, hd, wd)
    return out


def units_to_clips(key, units, b, t, ctx_len, world, real_vid=None):
    """The ranks' unit blocks of one key, concatenated in rank order (what the all-gather returns), shaped as predict()
    returns that key: the reconstruction's units go back from dealing order to frame order (one index copy; the
    prediction's are in frame order already), ``pred_flow`` to (B, Tc, Tp, 2, Hd, Wd), the context frames in front of
    ``pred_vid`` / ``inp_pred_vid`` (``real_vid``)."""
    phase = "rec" if key in UNIT_KEYS["rec"] else "pred"
    per_clip = t if phase == "rec" else t - ctx_len
    if phase == "rec" and world > 1:
        ids = [i for r in range(world) for i in local_unit_ids(phase, b, t, ctx_len, r, world)]
        if ids != list(range(b * per_clip)):
            natural = torch.empty_like(units)
            natural[_cached_index(ids, units.device)] = units
            units = natural
    full = units.view(b, per_clip, *units.shape[1:])
    if key == "pred_flow":  # units x (Tc * 2) -> (B, Tc, Tp, 2, Hd, Wd)
        hd, wd = full.shape[-2:]
        return full.view(b, per_clip, -1, 2, hd, wd).permute(0, 2, 1, 3, 4, 5).contiguous()
    if key in ("pred_vid", "inp_pred_vid") and real_vid is not None:
        return torch.cat([real_vid[:, :ctx_len], full], dim=1)
    return full


def gather_predict(local, real_vid, ctx_len, keys=None, group=None):
    """All-gather the ranks' blocks of ``predict_sharded`` (one collective per key, ALL issued before the first is waited
    for -- dist.all_gather_frames_async: RCCL over xGMI, or gloo in the tests) and shape them as predict() returns them."""
    import torch.distributed as dist
    from ..dist import all_gather_frames_async
    b, t = real_vid.shape[:2]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    keys = list(keys or local.keys())
    pending = {}
    for k in keys:
        per_clip = t if k in UNIT_KEYS["rec"] else t - ctx_len
        pending[k] = all_gather_frames_async(local[k], b * per_clip, group=group)
    return {k: units_to_clips(k, pending[k].wait(), b, t, ctx_len, world, real_vid) for k in keys}


def run(clip_dir, out_dir=None, dim=128, aspect_ratio=1.0, num_obj=3, num_lyt=20, frames=6, ctx_len=4, seed=0,
        device="cuda:0"):
    opt = demo_opt(dim, aspect_ratio, num_obj, num_lyt)
    size = (dim, int(dim * aspect_ratio))
    clip = wio.load_clip(clip_dir, size, num_lyt, max_frames=frames)
    dev = torch.device(device)
    vid, lyt = clip["vid"].unsqueeze(0).to(dev), clip["lyt"].unsqueeze(0).to(dev)
    warper = Warper(opt).to(dev)
    wif = WIF(opt, unet=UniformFusionUNet()).to(dev)
    net = synthetic_network_outputs(opt, 1, vid.shape[1], ctx_len, seed=seed, device=dev)
    res = predict(opt, warper, wif, vid, lyt, net, ctx_len)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        for key in ("rec_vid", "inp_rec_vid", "pred_vid", "inp_pred_vid"):
            if key in res:
                wio.dump_video(res[key][0], os.path.join(out_dir, key + ".gif"))
                wio.dump_image(res[key][0, -1], os.path.join(out_dir, key + "_last.png"))
        if "pred_flow" in res:
            fl = res["pred_flow"][0, -1, 0]
            wio.write_flo(os.path.join(out_dir, "pred_flow_last.flo"), fl)
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--clip", required=True, help="directory with the clip's frame PNGs (demo dataset layout)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--aspect-ratio", type=float, default=1.0)
    ap.add_argument("--num-obj", type=int, default=3)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--ctx-len", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    res = run(args.clip, args.out, args.dim, args.aspect_ratio, args.num_obj, frames=args.frames,
              ctx_len=args.ctx_len, seed=args.seed)
    for k, v in res.items():
        print(f"{k}: {tuple(v.shape)} range [{v.min().item():.3f}, {v.max().item():.3f}] "
              f"finite={bool(torch.isfinite(v).all())}")


if __name__ == "__main__":
    main()
