"""The warp-path part of one LVD training step at the reference's recipe -- the one place where the
reference differentiates THROUGH the path (models/synthesizer.py:815-841, scripts/cityscapes/train_lvd.sh:
128 x 256 with no full-resolution raster, 16 objects, 5-frame clips, ``ctx_mode "prev"``, ``include_self``,
layout filtering with weighted classes):

    decoder tail -> pose heads' affine -> estimate_alpha_grid_occ (TPS grids + grid inversion, compute_occ)
    -> decode_output (grid_to_flow + input_to_output) -> loss -> backward

with the tensors the networks outside the path would hand over as seeded leaves (decoder logits, pose head
outputs, occlusion scores, class logits).  ``bench.py --config LVD`` times it.
"""
import types

import torch

from .. import functional as WF
from ..nets import Warper, decode_output, estimate_alpha_grid_occ, flp
from ..nets.lvd import decoder_tail
from .utils import get_grid


def lvd_opt(**over):
    d = dict(latent_shape=[8, 16], obj_shape=[4, 4], time_dropout=0.0, num_obj=16, patch_size=16, scale_factor=1,
             dim=128, aspect_ratio=2, load_dim=0, num_perm_grid=1, normalize_alpha=False, use_lyt_filtering=True,
             use_lyt_opacity=True, weight_cls=True, min_cls=0.1, include_self=True, no_filter=False, allow_ghost=False)
    d.update(over)
    return types.SimpleNamespace(**d)


class LvdStep:
    """``clips`` clips of 5 frames resident on ``device``; ``__call__`` runs forward, loss and backward once and
    returns the loss (the leaves' ``.grad`` hold the gradients)."""

    frames, num_lyt = 5, 20

    def __init__(self, clips, device, seed=0):
        self.opt = o = lvd_opt()
        self.clips = b = clips
        t, no, nl = self.frames, o.num_obj, self.num_lyt
        lo, lb = o.obj_shape[0] * o.obj_shape[1], o.latent_shape[0] * o.latent_shape[1]
        h, w = o.dim, int(o.dim * o.aspect_ratio)
        ho = o.obj_shape[0] * o.patch_size
        self.warper = Warper(o).to(device)
        g = torch.Generator(device=device).manual_seed(seed)
        self.raw = torch.randn(b * no, 1, ho, ho, generator=g, device=device, requires_grad=True)
        self.pose_o = (0.3 * torch.randn(b * t, no, 6 + 2 * lo, generator=g, device=device)).requires_grad_()
        self.pose_b = (0.05 * torch.randn(b * t, 1, 6 + 2 * lb, generator=g, device=device)).requires_grad_()
        self.score = torch.randn(b, t, no, generator=g, device=device, requires_grad=True)
        self.cls_logit = torch.randn(b, no, nl, generator=g, device=device, requires_grad=True)
        self.inp = torch.randn(b, t, 3 + nl, h, w, generator=g, device=device)
        self.base_o = get_grid(*o.obj_shape).view(1, 1, lo, 2).to(device)
        self.base_b = get_grid(*o.latent_shape).view(1, 1, lb, 2).to(device)
        self.mul6 = torch.tensor([[[0.25, 0.25, 0.25, 0.25, 1.0, 1.0]]], device=device)
        self.bias_o = torch.tensor([[[0.25, 0.0, 0.0, 0.5, 0.0, 0.0]]], device=device)
        self.bias_b = torch.tensor([[[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]], device=device)
        self.bg_alpha = torch.ones(1, 1, h, w, device=device)
        # ctx_mode "prev" (synthesizer.py:833-835): every frame is predicted from the one before it
        self.ctx_ts = torch.roll(torch.arange(t, device=device), 1).view(1, 1, t).expand(b, -1, -1).contiguous()
        # (every frame is predicted, in order: an index MARKED as 0 .. T-1 lets time_gather hand out the clip itself)
        self.pred_ts = WF.arange_index(t, device)
        self.leaves = [self.raw, self.pose_o, self.pose_b, self.score, self.cls_logit]
        self.shape = (b, t, no, lo, lb, ho)

    def __call__(self):
        b, t, no, lo, lb, ho = self.shape
        for x in self.leaves:
            x.grad = None
        obj_alpha = decoder_tail(self.raw, init_bias=5.0).view(b, no, 1, ho, ho)
        obj_pose = flp.obj_pose_to_points(torch.tanh(self.pose_o), self.base_o, self.mul6, self.bias_o, 0.2)
        bg_pose = flp.bg_pose_to_points(torch.tanh(self.pose_b), self.base_b, self.bias_b, 1.2)
        occ, oa, ba, grid = estimate_alpha_grid_occ(self.warper, obj_alpha, self.bg_alpha,
                                                    obj_pose.view(b, t, no, lo, 2), bg_pose.view(b, t, 1, lb, 2),
                                                    self.score)
        out = decode_output(self.warper, self.inp, grid, occ, oa, ba, self.cls_logit.softmax(-1), self.ctx_ts,
                            self.pred_ts, restrict_to_ctx=False)
        loss = out[0].square().mean() + out[1].square().mean() + out[3].mean()
        loss.backward()
        return loss

    def grads_finite(self):
        return all(bool(torch.isfinite(x.grad).all()) for x in self.leaves)
