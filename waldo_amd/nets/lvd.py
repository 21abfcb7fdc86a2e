"""Host-side mirror of the hot-path part of the reference's models/nets/lvd.py.

``Warper`` keeps the reference's constructor, method names, argument meaning, return tuples and
persistent buffer names (``src_pts, tgt_pts, src_grid, src_grid_hd, tgt_grid`` + the four
sub-modules ``tps_obj, invert_obj, tps_bg, invert_bg``), so ``LVD`` / ``Synthesizer`` can
construct and call it unchanged and a reference checkpoint's ``warper.*`` entries load as they
are.  Every resampling (``F.grid_sample`` in the reference), every occlusion product
(``(1 - alpha * occ).prod(dim)``), the TPS grids and the grid inversion run in the hand-written
gfx950 kernels of ``waldo_amd.functional``; what is left in PyTorch is indexing, concatenation,
the small low-resolution softmax / mean of the layout filter and the bilinear ``F.interpolate``
rescale of the layouts.  The full-resolution passes of ``grid_to_flow[_ctx]`` and
``input_to_output`` run fused, forward (csrc/flow_ctx.hip; row f1 of SURVEY.md section 8) and
backward (csrc/flow_ctx_bwd.hip: the reference's live backward path in LVD training); the per-op
composition stays as the fallback for non-integer scales and for frames that require a gradient.
No CPU path: tensors must live on the GPU.

The occlusion products never materialise the reference's (L, L, h, w) broadcast, so the
``fast`` / ``restrict_to_ctx`` memory switches of the reference only change WHAT is returned
(``alpha_unflt`` is None when ``load_dim > 0``, as in the reference), not how it is computed.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as WF
from ..modules.warp import InverseWarp, TPSWarp
from ..tools.utils import get_grid


def compute_occ(occ_score, eps=1e-6):
    """Pairwise occlusion matrix, reference LVD.compute_occ (models/nets/lvd.py:59-68).
    occ_score (B, T, No) -> (B, T, No+1, No+1): occ[i, j] = s_i / (s_i + s_j) - [i == j] / 2 with
    s = exp(-score^2) + eps; first column ones (objects occlude the background), first row zeros.
    One kernel forward, one backward (csrc/producers.hip)."""
    return WF.compute_occ(occ_score, eps)


def decoder_tail(img, circle=None, init_bias=0.0, scale_factor=1, has_alpha=True, use_prior=False,
                 drop_alpha=False, obj_alpha_mask=None, remove_obj=False, freeze_obj=False):
    """The tail of the reference's ImageDecoder.forward (models/nets/lvd.py:245-254) on the raw image
    ``img`` (N, C, h, w) of its conv stack (a network outside this path): ``+ init_bias``; on the
    alpha channel ``tanh`` and, with ``use_prior``, the blend with the ``circle`` buffer;
    ``scale(img, scale_factor)``.  ``obj_alpha_mask`` / ``remove_obj`` / ``freeze_obj`` fold the
    arithmetic of ``LVD.forward(mode="estimate_alpha_grid_occ")`` (lvd.py:128-132) into the same
    kernel -- for the object decoder, whose only channel is alpha."""
    out = WF.alpha_head(img, prior=circle if (has_alpha and use_prior) else None, mask=obj_alpha_mask,
                        scale=scale_factor, bias=init_bias, has_alpha=has_alpha, remove=remove_obj,
                        freeze=freeze_obj)
    return out[:, :-1] if (has_alpha and drop_alpha) else out


def reduce_comp(vid, occ, flow=None):
    """Canonical over-composite, reference LVD.reduce_comp (models/nets/lvd.py:100-114).
    vid (B, T, L, C+1, H, W) in [-1, 1], occ (B, T, L, L) -> composited vid (B, T, C, H, W),
    alpha' (B, T, L, H, W) (both in [-1, 1]) and the composited flow."""
    b, t, nl = vid.shape[:3]
    v = (vid + 1) / 2
    alpha = torch.cat([torch.ones_like(v[:, :, :1, -1]), v[:, :, 1:, -1]], dim=2)  # bg alpha = 1
    h, w = alpha.shape[-2:]
    alpha = WF.occ_composite(alpha.reshape(b * t, nl, h, w), occ.reshape(b * t, nl, nl)).view(b, t, nl, h, w)
    out = (alpha.unsqueeze(3) * v[:, :, :, :-1]).sum(dim=2)
    fl = None
    if flow is not None:
        fl = (alpha[:, :-1].unsqueeze(3) * flow).sum(dim=2)
    return 2 * out - 1, 2 * alpha - 1, fl


def scale(tensor, scale_factor, mode="bilinear"):
    """Reference ``scale`` (lvd.py:175-179): F.interpolate on the trailing (C, h, w) of any rank."""
    if tensor is None or scale_factor == 1:
        return tensor
    lead = tensor.shape[:-3]
    out = F.interpolate(tensor.reshape(-1, *tensor.shape[-3:]), scale_factor=scale_factor, mode=mode)
    return out.reshape(*lead, *out.shape[-3:])


class TimeRepeat:
    """A clip of grids (B, Tp, ...) standing for its copies over Tc contexts, (B * Tc, Tp, ...) -- what
    ``x[:, pred_ts].unsqueeze(1).expand(-1, Tc, ...).reshape(B * Tc, Tp, ...)`` (lvd.py:665-668) materialises.
    ``obj_to_output`` / ``bg_to_output`` read it through the grid map of ``waldo_grid_sample2d_fwd`` instead
    (forward only: with a gradient in play ``Warper._layer_flows`` hands out the expanded tensors)."""

    def __init__(self, grid, repeat):
        self.grid, self.repeat = grid, int(repeat)


def gather_time(tensor, ts):
    """Reference ``gather_time`` (lvd.py:462-467): tensor (B, T, ...), ts (B, Tc, Tp)."""
    b, tc, tp = ts.shape
    idx = ts.reshape(b, tc * tp, *([1] * (tensor.ndim - 2))).expand(-1, -1, *tensor.shape[2:])
    return tensor.gather(1, idx).reshape(b, tc, tp, *tensor.shape[2:])


class Warper(nn.Module):
    """Reference: models/nets/lvd.py:469-870.  ``opt`` fields read: latent_shape, obj_shape,
    time_dropout, num_obj, patch_size, scale_factor, dim, aspect_ratio, load_dim, num_perm_grid,
    normalize_alpha, use_lyt_filtering, use_lyt_opacity, weight_cls, min_cls, include_self,
    no_filter, allow_ghost."""

    def __init__(self, opt, repeat_border=False):
        super().__init__()
        src_pts = get_grid(*opt.latent_shape).view(-1, 2)
        tgt_pts = get_grid(*opt.obj_shape).view(-1, 2)
        self.time_dropout = opt.time_dropout
        self.num_obj = opt.num_obj
        self.latent_obj_size = opt.obj_shape[0] * opt.obj_shape[1]
        self.latent_size = opt.latent_shape[0] * opt.latent_shape[1]
        self.tgt_shape = [int(opt.obj_shape[0] * opt.patch_size * opt.scale_factor),
                          int(opt.obj_shape[1] * opt.patch_size * opt.scale_factor)]
        self.src_shape = [opt.dim, int(opt.dim * opt.aspect_ratio)]
        self.src_shape_hd = [opt.load_dim, int(opt.load_dim * opt.aspect_ratio)] if opt.load_dim > 0 else self.src_shape
        self.register_buffer("src_pts", src_pts)
        self.register_buffer("tgt_pts", tgt_pts)
        self.register_buffer("src_grid", get_grid(*self.src_shape))
        self.register_buffer("src_grid_hd", get_grid(*self.src_shape_hd))
        self.register_buffer("tgt_grid", get_grid(*self.tgt_shape))
        self.tps_obj = TPSWarp(*self.tgt_shape, tgt_pts)
        self.invert_obj = InverseWarp(*self.tgt_shape, *self.src_shape, num_perm=opt.num_perm_grid)
        self.normalize_alpha = opt.normalize_alpha
        self.use_lyt_filtering = opt.use_lyt_filtering
        self.use_lyt_opacity = opt.use_lyt_opacity
        self.weight_cls = opt.weight_cls
        self.min_cls = opt.min_cls
        self.include_self = opt.include_self
        self.fast = opt.load_dim == 0
        self.scale_hd = opt.load_dim / opt.dim if opt.load_dim > 0 else 1
        self.tps_bg = TPSWarp(*self.src_shape, src_pts)
        self.invert_bg = InverseWarp(*self.src_shape, *self.src_shape, num_perm=opt.num_perm_grid)
        self.no_filter = opt.no_filter
        self.allow_ghost = opt.allow_ghost
        # ask the fused flow pass for max_l alpha_ctx as a by-product (read it from .alpha_ctx_max after the call)
        self.keep_alpha_ctx_max = False
        # False: the fused flow pass does not write `alpha` / `alpha_unflt` (2 a' - 1 on the Tw frames: as large as
        # what it keeps) and decode_output returns None for them -- for callers that drop them, as Synthesizer.predict's
        # reconstruction does (synthesizer.py:445: `rec_output, _, _, _, _, raw_output, alpha_ctx = ...`).  Its
        # prediction KEEPS `alpha` (synthesizer.py:472) for net_ii.inpaint, which reads it when the inpainter is on
        # (wif.py:103): leave this True there (tools/demo.py ties it to opt.use_inpainter).  Inference only.
        self.return_alpha = True
        self.alpha_ctx_max = None
        self.fuse_hd = True  # run the full-resolution passes of grid_to_flow[_ctx] / input_to_output fused
        # decode_output without autograd: the flow pass composites alpha_ctx straight into raw_output's slots (False: into
        # a tensor of its own that the frame warp reads and copies -- the same bits, for tests)
        self.raw_slots = True
        # on the paths without a ghost mask the alpha pass maps where each layer is and the flow pass skips the layers that
        # are absent around a tile's samples (WF.flow_ctx_alpha(want_bits=True)); False: every layer in every pixel
        self.layer_occupancy = True
        self._index_status = None

    @property
    def index_status(self):
        """This module's frame-index status words (``_lib.IndexStatus``: pinned host memory the kernels report into when
        ``ctx_ts`` / ``pred_ts`` hold an index outside the time axis, where the reference's ``gather_time`` fails,
        lvd.py:462-467).  Checked without a synchronisation at the start and the end of every flow synthesis / frame
        warp -- an error of an earlier launch surfaces at the next call, as a device-side assert does on the
        reference's GPU path -- and on demand by ``check_time_indices()``."""
        if self._index_status is None:
            from .._lib import IndexStatus
            self._index_status = IndexStatus()
        return self._index_status

    def check_time_indices(self):
        """Wait for the launches queued so far and raise if one of them met a frame index outside its range."""
        self.index_status.check(sync=True)

    # ------------------------------------------------------------------ image -> layer space
    def layer_from_input(self, input, grid):
        return self.obj_from_input(input, grid), self.bg_from_input(input, grid)

    def obj_from_input(self, input, grid):
        tgt_grid_obj = grid[0]
        b, t = input.shape[:2]
        c = input.size(-3)
        ho, wo = self.tgt_shape
        h, w = self.src_shape
        no = self.num_obj
        g = tgt_grid_obj.reshape(b * t * no, ho, wo, 2)
        if input.ndim == 5:  # one image per frame, shared by the objects: broadcast, no copies
            out = WF.grid_sample(input.reshape(b * t, c, h, w), g, broadcast=(no, 1))
        else:
            out = WF.grid_sample(input[:, :, 1:].reshape(b * t * no, c, h, w), g)
        return out.view(b, t, no, c, ho, wo)

    def bg_from_input(self, input, grid):
        tgt_grid_bg = grid[2]
        b, t = input.shape[:2]
        c = input.size(-3)
        h, w = self.src_shape
        src = input if input.ndim == 5 else input[:, :, :1]
        return WF.grid_sample(src.reshape(b * t, c, h, w), tgt_grid_bg.reshape(b * t, h, w, 2)).view(b, t, c, h, w)

    # ------------------------------------------------------------------ layer -> image space
    def layer_to_output(self, obj, bg, grid, delta_bg=1, delta_obj=1, pre=None, return_mask=False):
        """Reference lvd.py:533-537: ``cat([bg_to_output(bg), obj_to_output(obj)], dim=2)`` -- here one op whose two
        launches write the concatenated tensor directly and whose backward reads the two parts of its gradient in
        place (``WF.layers_to_output``).  ``pre`` = (scale, bias): the layers warped are ``scale * obj + bias`` and
        ``scale * bg + bias`` (grid_to_flow's ``(alpha + 1) / 2``, lvd.py:602-606) without being written first.
        ``return_mask``: also the warped all-ones canvas of the objects' grids (B, T, No, 1, H, W)."""
        src_grid_obj, src_grid_bg = grid[1], grid[3]
        if isinstance(src_grid_obj, TimeRepeat) or isinstance(src_grid_bg, TimeRepeat):
            # grids shared by several outputs (inference): the index-mapped launches of obj_ / bg_to_output
            if pre is not None:
                obj, bg = obj * pre[0] + pre[1], bg * pre[0] + pre[1]
            output = self.obj_to_output(obj, grid, delta_obj, return_mask=return_mask)
            cat = torch.cat([self.bg_to_output(bg, grid, delta_bg), output[0] if return_mask else output], dim=2)
            return (cat, output[1]) if return_mask else cat
        b, t, no = src_grid_obj.shape[:3]
        c1 = obj.size(-3)
        ho, wo = self.tgt_shape
        h, w = self.src_shape
        # (B, No, C+1, Ho, Wo) / (B, C+1, H, W): shared over time (lvd.py:544,555), broadcast without copies
        obc = (t * no, no) if obj.ndim == 5 else None
        bbc = (t, 1) if bg.ndim == 4 else None
        out = WF.layers_to_output(obj.reshape(-1, c1, ho, wo), bg.reshape(-1, c1, h, w),
                                  src_grid_obj.reshape(b * t * no, h, w, 2), src_grid_bg.reshape(b * t, h, w, 2),
                                  delta_obj, delta_bg, obc, bbc, pre if pre is not None else (1.0, 0.0), return_mask)
        if return_mask:
            return out[0].view(b, t, no + 1, c1, h, w), out[1].view(b, t, no, 1, h, w)
        return out.view(b, t, no + 1, c1, h, w)

    def obj_to_output(self, obj, grid, delta_obj=1, return_mask=False, into=None):
        """Reference lvd.py:533-549.  ``return_mask``: also the warped all-ones canvas of the same grids,
        ``obj_to_output(ones_like(obj[..., :1, :, :]), grid, delta_obj=0)`` -- a by-product of the taps.
        ``into``: a (frames, L, C+1, H, W) tensor whose layers 1 .. No receive the result (inference only)."""
        src_grid_obj = grid[1]
        c1 = obj.size(-3)
        ho, wo = self.tgt_shape
        h, w = self.src_shape
        if isinstance(src_grid_obj, TimeRepeat):  # grids (B, Tp, No, ...) for outputs (B * Tc, Tp, No, ...)
            rep, sg = src_grid_obj.repeat, src_grid_obj.grid
            b0, t, no = sg.shape[:3]
            out = WF.grid_sample(obj.reshape(b0 * rep * t * no, c1, ho, wo), sg.reshape(b0 * t * no, h, w, 2),
                                 delta=delta_obj, grid_repeat=(b0 * rep * t * no, rep * t * no, t * no),
                                 return_mask=return_mask, out=None if into is None else (into, no, no + 1, 1))
            if into is not None:
                return out[1].view(b0 * rep, t, no, 1, h, w) if return_mask else None
            if return_mask:
                return out[0].view(b0 * rep, t, no, c1, h, w), out[1].view(b0 * rep, t, no, 1, h, w)
            return out.view(b0 * rep, t, no, c1, h, w)
        b, t, no = src_grid_obj.shape[:3]
        g = src_grid_obj.reshape(b * t * no, h, w, 2)
        if obj.ndim == 5:  # (B, No, C+1, Ho, Wo) shared over time (lvd.py:544): n_in = b*No + o
            out = WF.grid_sample(obj.reshape(b * no, c1, ho, wo), g, delta=delta_obj, broadcast=(t * no, no),
                                 return_mask=return_mask)
        else:
            out = WF.grid_sample(obj.reshape(b * t * no, c1, ho, wo), g, delta=delta_obj, return_mask=return_mask)
        if return_mask:
            return out[0].view(b, t, no, c1, h, w), out[1].view(b, t, no, 1, h, w)
        return out.view(b, t, no, c1, h, w)

    def bg_to_output(self, bg, grid, delta_bg=1, eps=1e-6, into=None):
        src_grid_bg = grid[3]
        c1 = bg.size(-3)
        h, w = self.src_shape
        if isinstance(src_grid_bg, TimeRepeat):
            rep, sg = src_grid_bg.repeat, src_grid_bg.grid
            b0, t = sg.shape[:2]
            out = WF.grid_sample(bg.reshape(b0 * rep * t, c1, h, w), sg.reshape(b0 * t, h, w, 2), delta=delta_bg,
                                 grid_repeat=(b0 * rep * t, rep * t, t),
                                 out=None if into is None else (into, 1, into.size(-4), 0))
            if into is not None:
                return None
            return out.view(b0 * rep, t, 1, c1, h, w)
        b, t = src_grid_bg.shape[:2]
        g = src_grid_bg.reshape(b * t, h, w, 2)
        if bg.ndim == 4:  # (B, C+1, H, W) shared over time (lvd.py:555): n_in = b
            out = WF.grid_sample(bg.reshape(b, c1, h, w), g, delta=delta_bg, broadcast=(t, 1))
        else:
            out = WF.grid_sample(bg.reshape(b * t, c1, h, w), g, delta=delta_bg)
        return out.view(b, t, 1, c1, h, w)

    def _occ_product(self, alpha, occ):
        """alpha (..., L, h, w) in [0, 1], occ (..., L, L) with the same leading dims ->
        alpha_j * prod_i (1 - alpha_i occ[i, j])."""
        lead = alpha.shape[:-3]
        nl, h, w = alpha.shape[-3:]
        out = WF.occ_composite(alpha.reshape(-1, nl, h, w), occ.reshape(-1, nl, nl))
        return out.view(*lead, nl, h, w)

    def alpha_to_alpha(self, obj_alpha, bg_alpha, grid, occ):
        """Reference lvd.py:561-573 (only reached through the unused ``decode_layer`` mode)."""
        src_grid_obj = grid[1]
        b, t, no = src_grid_obj.shape[:3]
        a = ((self.layer_to_output(obj_alpha, bg_alpha, grid) + 1) / 2).squeeze(3)   # B T L H W
        occ = occ.reshape(b, t, no + 1, no + 1)
        factor = torch.stack([(1 - a * occ[:, :, :, j, None, None]).prod(dim=2) for j in range(no + 1)],
                             dim=2).unsqueeze(3)                                  # prod_i (1 - a_i occ_ij)
        output_alpha = self._occ_product(a, occ).unsqueeze(3)
        obj_occ, bg_occ = self.layer_from_input(factor, grid)
        return (obj_occ * (obj_alpha.unsqueeze(1) + 1) - 1, bg_occ * (bg_alpha.unsqueeze(1) + 1) - 1,
                output_alpha)

    # ------------------------------------------------------------------ flow helpers (WIF.inpaint)
    # (the reference indexes the frame with a LIST, ``tgt_grid_bg[:, [ref]]``: an index tensor built on the host and
    # copied to the device -- a copy that waits for the queue to drain, 5 ms into WIF.inpaint at 512 x 1024; a slice
    # of one frame is the same tensor without it)
    @staticmethod
    def _frame(x, i):
        return x.narrow(1, i % x.shape[1], 1)

    def grid_to_bg_flow_from_ref_to_pred(self, grid, ctx_len, ref):
        _, _, tgt_grid_bg, src_grid_bg = grid
        bg_flow = (self._frame(tgt_grid_bg, ref) - tgt_grid_bg[:, ctx_len:]).permute(0, 1, 4, 2, 3)
        bg_flow = self.bg_to_output(bg_flow, [None, None, None, src_grid_bg[:, ctx_len:]], delta_bg=0).squeeze(2)
        return scale(bg_flow, self.scale_hd).permute(0, 1, 3, 4, 2)

    def grid_to_obj_flow_from_ref_to_pred(self, grid, ctx_len, ref, obj_id):
        tgt_grid_obj, src_grid_obj, _, _ = grid
        if tgt_grid_obj.shape[0] == 1:  # (WIF.inpaint: one clip; the list indices of lvd.py:586-588 as slices)
            one = tgt_grid_obj.narrow(2, obj_id % tgt_grid_obj.shape[2], 1)
            obj_flow = self._frame(one, ref) - one[:, ctx_len:]
            src_one = src_grid_obj.narrow(2, obj_id % src_grid_obj.shape[2], 1)
        else:  # the reference's expression as it stands (its two lists pair up: (B, 1, ...) against (B, Tp, 1, ...))
            obj_flow = tgt_grid_obj[:, [ref], [obj_id]] - tgt_grid_obj[:, ctx_len:, [obj_id]]
            src_one = src_grid_obj[:, :, [obj_id]]
        obj_flow = obj_flow.permute(0, 1, 2, 5, 3, 4)  # B T 1 2 Ho Wo
        b, t = obj_flow.shape[:2]
        h, w = self.src_shape
        g = src_one[:, ctx_len:].reshape(b * t, h, w, 2)
        out = WF.grid_sample(obj_flow.reshape(b * t, 2, *self.tgt_shape), g).view(b, t, 2, h, w)
        return scale(out, self.scale_hd).permute(0, 1, 3, 4, 2)

    def grid_to_bg_flow_from_ctx_to_ref(self, grid, ctx_len, ref):
        _, _, tgt_grid_bg, src_grid_bg = grid
        bg_flow = (tgt_grid_bg[:, :ctx_len] - self._frame(tgt_grid_bg, ref)).permute(0, 1, 4, 2, 3)
        g = self._frame(src_grid_bg, ref).expand(-1, ctx_len, -1, -1, -1)
        bg_flow = self.bg_to_output(bg_flow, [None, None, None, g], delta_bg=0).squeeze(2)
        return scale(bg_flow, self.scale_hd).permute(0, 1, 3, 4, 2)

    # ------------------------------------------------------------------ flow / alpha synthesis
    def _lyt_dist(self, alpha, lyt, cls):
        """Class distribution of every object for the layout filter (lvd.py:624-634 / 731-746).
        alpha (B,Tw,L,1,H,W) with the background at layer 0, lyt (B,Tw,Nl,H,W), cls (B,No,Nl) or
        None -> (B,No,Nl).  One pass of csrc/lyt_dist.hip; a layout that requires a gradient (no
        script has one) takes the framework expression."""
        if not (cls is None or self.weight_cls):
            return cls
        if torch.is_grad_enabled() and lyt.requires_grad:
            return self._lyt_dist_torch(alpha[:, :, 1:], lyt, cls)
        return WF.lyt_dist(alpha.squeeze(3), lyt, cls if self.weight_cls else None, self.min_cls, first_obj=1)

    def _lyt_dist_torch(self, alpha_obj, lyt, cls):
        """The same distribution as a framework expression, differentiable w.r.t. the layout as the reference's
        is (lvd.py:737-744); checked against the kernel and the CPU restatement's layout gradient in
        tests/test_gpu_warper.py::test_lyt_dist_layout_gradient_branch."""
        win = alpha_obj.squeeze(3) + 1e-6                                       # B Tw No H W
        if self.weight_cls:
            win = win * torch.einsum("bon,btnhw->btohw", cls + self.min_cls, lyt.softmax(dim=2))
        total = win.sum(dim=(1, 3, 4))                                          # B No
        mean = torch.einsum("btohw,btnhw->bon", win, lyt) / total.unsqueeze(2)
        return mean.softmax(dim=2)                                              # B No Nl

    def _lyt_alpha(self, alpha, lyt, hd_lyt, cls):
        """Layout filter (lvd.py:624-639 / 731-751).  alpha (B,Tw,L,1,H,W), lyt (B,Tw,Nl,H,W),
        hd_lyt (B,Tw,Nl,Hd,Wd), cls (B,No,Nl) or None -> (B,Tw,No,1,Hd,Wd).  The reference builds a
        (B,Tw,No,Nl,Hd,Wd) tensor; this loops over the objects instead."""
        no = alpha.shape[2] - 1
        hd_prob = hd_lyt.softmax(dim=2)
        dist = self._lyt_dist(alpha, lyt, cls)
        out = [1 - (dist[:, None, o, :, None, None] - hd_prob).abs().sum(dim=2, keepdim=True) / 2
               for o in range(no)]
        return torch.stack(out, dim=2)

    def _fused_ok(self, tensors, nl, ncls):
        """The fused HD passes (forward and backward kernels) need an integer upsampling factor and
        frames that are data: ``tensors[0]`` (the input video / layout) must not require a gradient."""
        s = self.scale_hd
        return (not (torch.is_grad_enabled() and tensors[0] is not None and tensors[0].requires_grad)
                and float(s) == int(s) and int(s) >= 1 and nl <= 32 and 1 <= ncls <= 32
                and self.src_shape_hd[0] == self.src_shape[0] * int(s)
                and self.src_shape_hd[1] == self.src_shape[1] * int(s))

    def _clip_length(self, input, occ, nl, ctx_only):
        """Frames per clip T on the time axis that ``grid`` / ``occ`` / ``pred_ts`` share.  The reference hands
        ``decode_output`` an ``input`` of all T frames (synthesizer.py:439-445) of which the ``restrict_to_ctx`` path
        reads the first Tc only (lvd.py:716-745, 837); a caller that shards the predicted frames over ranks
        (tools/demo.py:predict_sharded) passes just those context frames: T then comes from ``occ`` (B, T, L, L)."""
        t = input.size(1)
        if occ.ndim == 4 and occ.size(1) != t:
            if not ctx_only or self.include_self:
                raise ValueError(f"input holds {t} frames per clip, occ {occ.size(1)}: an input of the context "
                                 "frames alone needs restrict_to_ctx and no include_self")
            t = occ.size(1)
        return t

    def _layer_flows(self, grid, ctx_ts, pred_ts):
        """The layer-space flows between the context and the predicted frames and the predicted frames'
        source grids repeated over the contexts (lvd.py:660-668 / 780-787):

            obj_flow = gather_time(tgt_grid_obj, ctx_ts) - tgt_grid_obj[:, pred_ts].unsqueeze(1)
            obj_flow = obj_flow.permute(0, 1, 2, 3, 6, 4, 5).view(B * Tc, Tp, No, 2, Ho, Wo)     (bg alike)
            src_grid_obj[:, pred_ts].unsqueeze(1).expand(-1, Tc, ...).view(B * Tc, Tp, No, H, W, 2)

        each as one ``waldo_time_gather`` launch (forward and backward) instead of the gather /
        advanced-index / subtract / permute / copy kernels of the spelled-out form."""
        tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg = grid
        b, _, no = src_grid_obj.shape[:3]
        tc, tp = ctx_ts.size(1), pred_ts.size(0)
        h, w = self.src_shape
        ho, wo = self.tgt_shape
        st = self.index_status
        obj_flow = WF.time_gather(tgt_grid_obj, ctx_ts, pred_ts, subtract=True, channel_first=True, status=st)
        bg_flow = WF.time_gather(tgt_grid_bg.unsqueeze(2), ctx_ts, pred_ts, subtract=True, channel_first=True, status=st)
        if torch.is_grad_enabled() and (src_grid_obj.requires_grad or src_grid_bg.requires_grad or
                                        tgt_grid_obj.requires_grad or tgt_grid_bg.requires_grad):
            sgo = WF.time_gather(src_grid_obj, None, pred_ts, num_ctx=tc, status=st).reshape(b * tc, tp, no, h, w, 2)
            sgb = WF.time_gather(src_grid_bg, None, pred_ts, num_ctx=tc, status=st).reshape(b * tc, tp, h, w, 2)
        else:  # inference: the Tc copies are never made (TimeRepeat)
            sgo = TimeRepeat(WF.time_gather(src_grid_obj, None, pred_ts, num_ctx=1, status=st).reshape(b, tp, no, h, w, 2), tc)
            sgb = TimeRepeat(WF.time_gather(src_grid_bg, None, pred_ts, num_ctx=1, status=st).reshape(b, tp, h, w, 2), tc)
        return obj_flow.reshape(b * tc, tp, no, 2, ho, wo), bg_flow.reshape(b * tc, tp, 2, h, w), sgo, sgb

    def _composited_alphas(self, input, grid, occ, obj_alpha, bg_alpha, cls, tw, filt, want_bits):
        """First half of the fused flow synthesis (lvd.py:716-766 / 602-652): the rough alphas of frames 0 .. tw - 1 warped
        to the image, the layout filter's class distribution, and the full-resolution pass that upsamples, filters and
        composites them.  Returns ``(a01, alpha_out, layer_bits)`` (WF.flow_ctx_alpha; ``layer_bits`` None with autograd).  With ``restrict_to_ctx`` these depend on the
        CONTEXT frames alone (their grids, frames and occlusion matrices): ``context_products``."""
        tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg = grid
        b, _, no = src_grid_obj.shape[:3]
        nl = no + 1
        h, w = self.src_shape
        hd, wd = self.src_shape_hd
        s = int(self.scale_hd)
        # the rough alphas of the Tw frames that are used (lvd.py:716-722 warps all T and slices: with four contexts of
        # fourteen frames ten of them for nothing); (x + 1) / 2 folded into the taps
        ga = grid if tw >= src_grid_obj.size(1) else [None, src_grid_obj[:, :tw], None, src_grid_bg[:, :tw]]
        alpha = self.layer_to_output(obj_alpha, bg_alpha, ga, delta_bg=0, delta_obj=0, pre=(0.5, 0.5))  # B Tw L 1 H W
        dist = None
        if filt:
            if s >= 2 and s & (s - 1) == 0 and input.is_cuda and hd % s == 0 and wd % s == 0:
                lyt = WF.downscale_frames(input, tw, 3, s)  # the same bits in one pass (waldo_downscale_frames_fwd)
            else:
                lyt = scale(input[:, :tw, 3:], 1 / self.scale_hd)
            dist = self._lyt_dist(alpha, lyt, cls)
        occ = occ.reshape(b, -1, nl, nl)
        # (an input of the context frames alone -- _clip_length -- goes with the occlusion matrices of those frames)
        occ_in = occ if input.size(1) == occ.size(1) else occ[:, :input.size(1)]
        # (want_bits: where each layer IS in the composited alphas -- the second pass's substitute for the ghost mask,
        # asked for on the paths that have none)
        res = WF.flow_ctx_alpha(alpha.reshape(b * tw, nl, h, w), input, dist, occ_in, tw, 3, s,
                                want_alpha=self.return_alpha, want_bits=want_bits)
        return res if want_bits else (*res, None)

    def context_products(self, input, grid, occ, obj_alpha, bg_alpha, cls, num_ctx):
        """What ``grid_to_flow_ctx`` (``restrict_to_ctx``) computes from the CONTEXT frames alone: the composited
        full-resolution alphas of frames 0 .. num_ctx - 1.  ``input`` (B, >= num_ctx, C, Hd, Wd), ``grid`` / ``occ`` with
        at least the context frames on their time axis.  A caller that decodes the same context twice -- the
        reconstruction and the prediction of ``Synthesizer.predict`` (synthesizer.py:445, 472), or two blocks of one clip's
        frames on one rank -- computes them once and hands them to ``decode_output(..., ctx_products=...)``; every value
        depends on its own (b, t) frame only, so the bits are those of the call that computes them itself.
        Inference only (no gradient flows through the hand-over)."""
        if not (self.fuse_hd and self._fused_ok([input], grid[1].shape[2] + 1, input.size(2) - 3)):
            return None
        with torch.no_grad():
            g = [x[:, :num_ctx] if x is not None else None for x in grid]
            oc = occ.reshape(occ.shape[0], -1, *occ.shape[-2:])[:, :num_ctx]
            return self._composited_alphas(input[:, :num_ctx], g, oc, obj_alpha, bg_alpha, cls, num_ctx, True,
                                           bool(self.allow_ghost and self.layer_occupancy))

    def _flow_fused(self, input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only, into_raw=None,
                    ctx_products=None):
        """_flow_common with the two full-resolution passes fused (csrc/flow_ctx.hip); everything at
        the low resolution goes through the same per-op kernels as the unfused path.  ``into_raw``
        (decode_output, no autograd; a list): alpha_ctx is written into the slots it will occupy in
        input_to_output's ``raw`` tensor and comes back as a view of it; the list receives the ``WF.RawSlots``
        that ``WF.frame_warp_fuse_raw`` takes (WF.flow_ctx_warp_into_raw).  ``ctx_products``: ``context_products``'
        result for the same context (``ctx_only``): the first half is not run again."""
        tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg = grid
        b, _, no = src_grid_obj.shape[:3]
        tc, tp = ctx_ts.size(1), pred_ts.size(0)
        nl = no + 1
        t = self._clip_length(input, occ, nl, ctx_only)
        h, w = self.src_shape
        hd, wd = self.src_shape_hd
        ho, wo = self.tgt_shape
        s = int(self.scale_hd)
        tw = tc if ctx_only else t
        occ = occ.reshape(b, t, nl, nl)
        if ctx_products is not None and ctx_only:
            a01, alpha_out, layer_bits = ctx_products
            if tuple(a01.shape) != (b * tw, nl, hd, wd):
                raise ValueError(f"ctx_products hold alphas of shape {tuple(a01.shape)}, this decode needs "
                                 f"{(b * tw, nl, hd, wd)}")
        else:
            a01, alpha_out, layer_bits = self._composited_alphas(input, grid, occ, obj_alpha, bg_alpha, cls, tw,
                                                                 ctx_only or not self.no_filter,
                                                                 self.layer_occupancy and s >= 2 and
                                                                 not (ctx_only and not self.allow_ghost))

        obj_flow, bg_flow, sgo, sgb = self._layer_flows(grid, ctx_ts, pred_ts)
        gridp = [None, sgo, None, sgb]
        is_obj = None
        if ctx_only and not self.allow_ghost:
            # the warped all-ones canvas of the ghost test (lvd.py:785-791) comes out of the SAME grids the object
            # flows are warped with one statement later (lvd.py:792): a by-product of that call's taps instead of
            # a launch of its own over B * Tc * Tp * No maps
            if isinstance(sgo, TimeRepeat):  # inference: both warps write straight into the concatenated tensor
                flow_lr = input.new_empty(b * tc, tp, nl, 2, h, w)
                is_obj = self.obj_to_output(obj_flow, gridp, delta_obj=0, return_mask=True, into=flow_lr)
                self.bg_to_output(bg_flow, gridp, 0, into=flow_lr)
            else:
                flow_lr, is_obj = self.layer_to_output(obj_flow, bg_flow, gridp, delta_bg=0, delta_obj=0, return_mask=True)
            is_obj = is_obj.reshape(b * tc * tp, no, h, w)
        else:
            flow_lr = self.layer_to_output(obj_flow, bg_flow, gridp, delta_bg=0, delta_obj=0)
        if into_raw is not None:
            res = WF.flow_ctx_warp_into_raw(flow_lr.reshape(b * tc * tp, nl, 2, h, w), is_obj, a01, ctx_ts, pred_ts,
                                            occ, tw, s, input.size(2), self.include_self and tp == t,
                                            layer_max=self.keep_alpha_ctx_max, status=self.index_status,
                                            layer_bits=layer_bits if is_obj is None else None)
            into_raw.append(res[4])
        else:
            res = WF.flow_ctx_warp(flow_lr.reshape(b * tc * tp, nl, 2, h, w), is_obj, a01, ctx_ts, pred_ts, occ, tw, s,
                                   layer_max=self.keep_alpha_ctx_max, status=self.index_status,
                                   layer_bits=layer_bits if is_obj is None else None)
        flow, alpha_ctx, disocc = res[:3]
        # by-product for Synthesizer.predict's disocclusion test (synthesizer.py:447: alpha_ctx.max(dim=3)[0])
        self.alpha_ctx_max = res[3].view(b, tc, tp, hd, wd) if self.keep_alpha_ctx_max else None
        alpha_out = alpha_out.view(b, tw, nl, hd, wd) if alpha_out is not None else None
        return (flow.view(b, tc, tp, 2, hd, wd), (alpha_out if self.fast else None), alpha_out,
                (alpha_ctx if into_raw is not None else alpha_ctx.view(b, tc, tp, nl, hd, wd)),
                disocc.view(b, tc, tp, 1, hd, wd))

    def _flow_common(self, input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only, into_raw=None,
                     ctx_products=None):
        # int64 + contiguous ONCE for every op below; what earlier launches reported about their indices surfaces here
        ctx_ts, pred_ts = WF.normalise_time_index(ctx_ts), WF.normalise_time_index(pred_ts)
        self.index_status.check()
        if self.fuse_hd and self._fused_ok([input, occ, obj_alpha, bg_alpha, cls, *grid],
                                           grid[1].shape[2] + 1, input.size(2) - 3):
            no_grad = not (torch.is_grad_enabled() and any(
                x is not None and x.requires_grad for x in (occ, obj_alpha, bg_alpha, cls, *grid)))
            raw_ok = into_raw is not None and self.raw_slots and no_grad and \
                self._frame_warp_fused(input, ctx_ts.size(1), pred_ts.size(0))
            return self._flow_fused(input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only,
                                    into_raw=into_raw if raw_ok else None, ctx_products=ctx_products if no_grad else None)
        self.alpha_ctx_max = None  # (only the fused pass produces it)
        tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg = grid
        b, _, no = src_grid_obj.shape[:3]
        tc, tp = ctx_ts.size(1), pred_ts.size(0)
        nl = no + 1
        t = self._clip_length(input, occ, nl, ctx_only)
        h, w = self.src_shape
        hd, wd = self.src_shape_hd
        ho, wo = self.tgt_shape
        hd_input = input
        input = scale(hd_input, 1 / self.scale_hd)
        win = slice(0, tc) if ctx_only else slice(0, t)

        # rough alpha of every layer in image space (objects / background shared over time)
        alpha = self.layer_to_output(obj_alpha, bg_alpha, grid, delta_bg=0, delta_obj=0, pre=(0.5, 0.5))  # of (x + 1) / 2
        alpha = alpha[:, win]                                                   # B Tw L 1 H W
        filt = ctx_only or not self.no_filter
        if filt:
            lyt_alpha = self._lyt_alpha(alpha, input[:, win, 3:], hd_input[:, win, 3:], cls)
        alpha = scale(alpha, self.scale_hd)
        if filt:
            alpha = torch.cat([alpha[:, :, :1], alpha[:, :, 1:] * lyt_alpha], dim=2)
        occ = occ.reshape(b, t, nl, nl)
        alpha = self._occ_product(alpha.squeeze(3), occ[:, win]).unsqueeze(3)    # B Tw L 1 Hd Wd
        alpha_unflt = alpha

        # per-layer flow in layer space between context and predicted frames, warped to the image
        obj_flow, bg_flow, sgo, sgb = self._layer_flows(grid, ctx_ts, pred_ts)
        gridp = [None, sgo, None, sgb]
        is_obj = 1
        if ctx_only and not self.allow_ghost:
            ones = torch.ones(b * tc, tp, no, 1, ho, wo, device=input.device, dtype=input.dtype)
            is_obj = self.obj_to_output(ones, gridp, delta_obj=0)
            is_obj = (scale(is_obj, self.scale_hd) > 0.9).to(input.dtype).view(b, tc, tp, no, 1, hd, wd)
            is_obj = torch.cat([torch.ones_like(is_obj[:, :, :, :1]), is_obj], dim=3)
        flow = self.layer_to_output(obj_flow, bg_flow, gridp, delta_bg=0, delta_obj=0)
        flow = scale(flow.view(b, tc, tp, nl, 2, h, w), self.scale_hd)           # B Tc Tp L 2 Hd Wd
        samp = self.src_grid_hd + flow.permute(0, 1, 2, 3, 5, 6, 4).reshape(b * tc * tp * nl, hd, wd, 2)

        # context alpha warped by the flow, second occlusion product, flow compositing
        alpha_ctx = gather_time(alpha, ctx_ts).reshape(b * tc * tp * nl, 1, hd, wd)
        alpha_ctx = WF.grid_sample(alpha_ctx, samp).reshape(b, tc, tp, nl, 1, hd, wd) * is_obj
        disocc = alpha_ctx.max(dim=3)[0]
        occ_p = occ[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1)
        alpha_ctx = self._occ_product(alpha_ctx.squeeze(4), occ_p).unsqueeze(4)
        flow = (alpha_ctx * flow).sum(dim=3)

        alpha_unflt = alpha_unflt.squeeze(-3) * 2 - 1
        alpha = alpha.squeeze(-3) * 2 - 1
        alpha_ctx = alpha_ctx.squeeze(-3) * 2 - 1
        return flow, (alpha_unflt if self.fast else None), alpha, alpha_ctx, disocc

    def grid_to_flow(self, input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts):
        """Reference lvd.py:602-705 (training path: alpha composited on all T frames)."""
        return self._flow_common(input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, False)

    def grid_to_flow_ctx(self, input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts):
        """Reference lvd.py:707-828 (restrict_to_ctx inference path)."""
        return self._flow_common(input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, True)

    def _frame_warp_fused(self, input, tc, tp):
        """input_to_output runs as the fused kernel (csrc/flow_ctx.hip:frame_warp_fuse_kernel)."""
        self_slot = self.include_self and tp == input.size(1)
        return self.fuse_hd and tc + int(self_slot) <= WF.MAX_FUSE_CTX and \
            not (torch.is_grad_enabled() and input.requires_grad)

    def input_to_output(self, input, alpha, flow, ctx_ts, eps=1e-6):
        """Reference lvd.py:830-853."""
        b, tc, tp = flow.shape[:3]
        ctx_ts = WF.normalise_time_index(ctx_ts)
        self_slot = self.include_self and tp == input.size(1)
        if self._frame_warp_fused(input, tc, tp):
            st = self.index_status
            st.check()
            return WF.frame_warp_fuse(input, flow, alpha, ctx_ts, include_self=self_slot, eps=eps, status=st)
        hd, wd = self.src_shape_hd
        c = input.size(-3)
        samp = self.src_grid_hd + flow.permute(0, 1, 2, 4, 5, 3).reshape(b * tc * tp, hd, wd, 2)
        output = WF.grid_sample(gather_time(input, ctx_ts).reshape(b * tc * tp, c, hd, wd), samp)
        output = output.reshape(b, tc, tp, c, hd, wd)
        score = ((alpha + 1) / 2).sum(dim=3, keepdim=True)
        if self.include_self and tp == input.size(1):
            score = torch.cat([score, torch.ones_like(score[:, :1])], dim=1)
            alpha = torch.cat([alpha, torch.ones_like(alpha[:, :1])], dim=1)
            output = torch.cat([output, input.unsqueeze(1)], dim=1)
        raw_output = torch.cat([output, alpha], dim=3)
        output = torch.cat([output, score * 2 - 1], dim=3)
        score = F.normalize(score + eps, p=1, dim=1)
        return (output * score).sum(dim=1), raw_output

    # ------------------------------------------------------------------ the four grids
    def forward(self, obj_pose, bg_pose, invert=True):
        """Reference lvd.py:855-870: (obj_pose (B,T,No,Lo,2), bg_pose (B,T,1,Lb,2)) ->
        (tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg)."""
        b, t, no = obj_pose.shape[:3]
        lo, lb = self.latent_obj_size, self.latent_size
        tgt_grid_obj = self.tps_obj(obj_pose.reshape(b * t * no, lo, 2))
        src_grid_obj = self.invert_obj(tgt_grid_obj) if invert else None
        tgt_grid_obj = tgt_grid_obj.view(b, t, no, *tgt_grid_obj.shape[1:])
        src_grid_obj = src_grid_obj.view(b, t, no, *src_grid_obj.shape[1:]) if invert else None
        tgt_grid_bg = self.tps_bg(bg_pose.reshape(b * t, lb, 2))
        src_grid_bg = self.invert_bg(tgt_grid_bg, erode=False) if invert else None
        tgt_grid_bg = tgt_grid_bg.view(b, t, *tgt_grid_bg.shape[1:])
        src_grid_bg = src_grid_bg.view(b, t, *src_grid_bg.shape[1:]) if invert else None
        return tgt_grid_obj, src_grid_obj, tgt_grid_bg, src_grid_bg


# ---------------------------------------------------------------------- LVD.forward glue (A11)
def estimate_alpha_grid_occ(warper, obj_alpha, bg_alpha, obj_pose, bg_pose, occ_score, obj_alpha_mask=None,
                            remove_obj=False, freeze_obj=False):
    """The warp-path part of ``LVD.forward(mode="estimate_alpha_grid_occ")`` (lvd.py:126-135).
    ``obj_alpha`` (B, No, 1, Ho, Wo) is the object decoder's output (``self.decoder(x_obj)``, a conv
    net outside this path), ``bg_alpha`` the model's (1, 1, H, W) parameter; ``obj_alpha_mask`` the
    padding mask of lvd.py:132.  Returns ``(occ, obj_alpha, bg_alpha, grid)``."""
    bg_alpha = bg_alpha.expand(obj_alpha.size(0), -1, -1, -1)
    if remove_obj or freeze_obj or obj_alpha_mask is not None:
        ho, wo = obj_alpha.shape[-2:]
        mask = obj_alpha_mask.expand(1, 1, 1, ho, wo).reshape(ho, wo) if torch.is_tensor(obj_alpha_mask) else None
        obj_alpha = WF.alpha_head(obj_alpha.reshape(-1, 1, ho, wo), mask=mask, has_alpha=False,
                                  remove=remove_obj, freeze=freeze_obj).view(obj_alpha.shape)
    grid = warper(obj_pose, bg_pose)
    return compute_occ(occ_score), obj_alpha, bg_alpha, grid


def decode_output(warper, input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, restrict_to_ctx=True,
                  use_disocc=False, ctx_products=None):
    """``LVD.forward(mode="decode_output")`` (lvd.py:141-153): flow / alpha synthesis, frame warp and
    temporal fusion, the ``use_disocc`` concatenation (lvd.py:148-151) and the split of the score
    channel.  Returns ``(output, flow, alpha_unflt, alpha, raw_alpha, raw_output, alpha_ctx)``.

    ``input`` holds all T frames as in the reference -- or, with ``restrict_to_ctx`` and no ``include_self``, just the
    context frames the path reads (``Warper._clip_length``).  Without autograd and without ``use_disocc``, ``alpha_ctx``
    is a strided VIEW into ``raw_output``'s storage (the reference returns two tensors; the values are the same): an
    in-place write to either shows in the other, and the view keeps the whole buffer alive -- clone it to detach.
    ``ctx_products``: ``Warper.context_products`` of the same context frames (``restrict_to_ctx``, no autograd), computed
    once by a caller that decodes that context more than once."""
    ctx_ts, pred_ts = WF.normalise_time_index(ctx_ts), WF.normalise_time_index(pred_ts)  # shared by both calls
    # (without autograd the context alphas are composited straight into raw_output's slots: `slots` receives what
    # the frame warp needs to know about them)
    slots = []
    flow, alpha_unflt, alpha, alpha_ctx, disocc = warper._flow_common(input, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts,
                                                                      pred_ts, restrict_to_ctx, into_raw=slots,
                                                                      ctx_products=ctx_products)
    if slots:
        output, raw_output = WF.frame_warp_fuse_raw(input, flow, slots[0], ctx_ts, status=warper.index_status)
    else:
        output, raw_output = warper.input_to_output(input, alpha_ctx, flow, ctx_ts)
    warper.index_status.check()  # (no synchronisation: whatever has been reported by now)
    # (ONE split instead of two slices of `output`: backward is a concatenation of the two gradients, where two
    # SliceBackward nodes each zero-fill a buffer of the full size and autograd adds them)
    output, raw_alpha = torch.split(output, [output.size(2) - 1, 1], dim=2)
    if use_disocc:
        if warper.include_self:
            disocc = torch.cat([disocc, torch.ones_like(disocc[:, :1])], dim=1)
        raw_output = torch.cat([raw_output, disocc], dim=3)
    return output, flow, alpha_unflt, alpha, raw_alpha, raw_output, alpha_ctx
