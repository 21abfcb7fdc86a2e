"""Drop-in for the hot-path part of the reference's models/nets/wif.py: ``WIF.forward`` and the
warp / mask arithmetic of ``WIF.inpaint``.

The UNet itself (models/modules/conv.py, MIOpen convolutions) is out of scope: the constructor
takes any ``nn.Module`` mapping (N, C_in, H, W) -> (N, 4|5, H, W) (or builds nothing when None is
given and ``forward`` is called with precomputed network outputs through ``fuse``).  The fusion
arithmetic around it runs in one hand-written gfx950 kernel (``waldo_wif_fuse_*``)."""
import torch
import torch.nn as nn

from .. import functional as WF
from ..tools.utils import get_grid

expand = WF.mask_expand  # tools/utils.py:300-323 as one launch (waldo_amd.tools.utils.expand states the steps)

_MASK_T = 0.1  # `mask_thresh` of wif.py:65: a warped / composited mask counts above 1 - _MASK_T


def point_in_polygon(pts, corners):
    """(1, H, W, 2) pixel coordinates inside the polygon `corners` -> (1, 1, H, W) bool: the reference's
    ``matplotlib.path.Path(corners).contains_points`` (wif.py:228-235) -- which it runs on the host, behind a copy of
    every pixel coordinate -- as one launch of matplotlib's own crossings test in double precision
    (``WF.points_in_polygon``; against matplotlib itself in tests/test_inpaint.py)."""
    b, h, w, _ = pts.shape
    if b != 1:
        raise ValueError("point_in_polygon: batch size 1 only (as the reference)")
    return WF.points_in_polygon(pts, corners).view(b, 1, h, w)


class WIF(nn.Module):
    """``forward(vid)`` with vid (B, Tc, T, C, H, W) -> (B, T, 3, H, W), reference wif.py:37-57.

    opt fields read: ``ii_score`` (must be true: the score-fusion variant every script uses),
    ``ii_ab``."""

    def __init__(self, opt, unet=None):
        super().__init__()
        self.score = opt.ii_score
        self.ab = opt.ii_ab
        self.opt = opt
        self.unet = unet
        self.fuse_propagate = True  # WIF.inpaint's per-frame propagation as one launch (False: the spelled-out loop)
        if hasattr(opt, "dim"):  # the HD identity grid `inpaint` warps against (wif.py:29-31)
            shape = [opt.dim, int(opt.dim * opt.aspect_ratio)]
            if getattr(opt, "load_dim", 0) > 0:
                shape = [opt.load_dim, int(opt.load_dim * opt.aspect_ratio)]
            self.register_buffer("src_grid_hd", get_grid(*shape), persistent=False)

    def get_last_layer(self):
        return self.unet.from_emb.weight

    def fuse(self, vid_t, net_out):
        """vid_t (B, T, Tc, C, H, W) (already permuted), net_out (B, T, Tc, Co, H, W)."""
        return WF.wif_fuse(vid_t, net_out, ab=self.ab)

    def forward(self, vid):
        b, tc, t, c, h, w = vid.shape
        vid = vid.permute(0, 2, 1, 3, 4, 5).contiguous()
        if not self.score:
            out = self.unet(vid.reshape(b * t, tc * c, h, w))
            return out.reshape(b, t, -1, h, w)
        out = self.unet(vid.reshape(b * t * tc, c, h, w))
        return self.fuse(vid, out.reshape(b, t, tc, -1, h, w))

    # ------------------------------------------------------------------ inpaint (wif.py:58-226)
    def _warp(self, x, flow):
        """grid_sample(x, flow + identity) with flow (B, Hd, Wd, 2) in grid units."""
        return WF.grid_sample(x, flow + self.src_grid_hd)

    def _warp_mask(self, m, flow):
        return (self._warp(m, flow) > 1 - _MASK_T).float()

    def _holes(self, alpha_ctx):
        """Disocclusion and object masks of the predicted frames (wif.py:60-79), (B, Tp, 1, H, W)."""
        o = self.opt
        if self.fuse_propagate and alpha_ctx.is_cuda and alpha_ctx.dtype == torch.float32:
            # one pass over alpha_ctx instead of five (csrc/inpaint_ops.hip): the same mask pixels
            mask, obj_mask = WF.inpaint_holes(alpha_ctx, last_only=o.ii_last_only, fix_thresh=o.fix_thresh)
            if o.use_expansion:
                mask = expand(mask, num=o.num_expansion) * (1 - obj_mask)
            return mask, obj_mask
        cover = ((alpha_ctx + 1) / 2).sum(dim=3, keepdim=True)
        obj = ((alpha_ctx[:, :, :, 1:] + 1) / 2).sum(dim=3, keepdim=True)
        if o.ii_last_only:
            cover, obj = cover[:, -1], obj[:, -1]
        else:
            cover, obj = cover.max(dim=1)[0], obj.max(dim=1)[0]
        mask = 1 - cover
        mask = (mask > _MASK_T).float() if o.fix_thresh else (mask > 1 - _MASK_T).float()
        obj_mask = (obj > 0.9).float()
        if o.use_expansion:
            mask = expand(mask, num=o.num_expansion) * (1 - obj_mask)
        return mask, obj_mask

    def _reference_background(self, inpainter, frames, mask, obj_mask, alpha, real_vid, ctx_len, warper, grid, ref):
        """The reference frame with the background behind its objects gathered from the context
        frames, then inpainted (wif.py:96-128).  Returns (ref_img, ref_mask, shadow_mask)."""
        o = self.opt
        ctx_to_ref = warper.grid_to_bg_flow_from_ctx_to_ref(grid, ctx_len, ref)
        ref_img = frames[ref].squeeze(1)
        behind = obj_mask[:, ref]
        shadow = None
        for t2 in range(ctx_len - 1, -1, -1):
            seen = (alpha[:, t2, :1] > 1 - _MASK_T).float()
            w_img = self._warp(real_vid[:, t2], ctx_to_ref[:, t2])
            w_seen = self._warp_mask(seen, ctx_to_ref[:, t2])
            if o.use_shadows and t2 == ctx_len - 1:
                shadow = ((w_img - ref_img).abs().mean(dim=1, keepdim=True) > 0.25).float() * w_seen * (1 - behind)
                shadow = 1 - expand(1 - shadow, num=5)
                shadow = expand(shadow, num=5)  # drops small regions
                shadow[:, :, :int(shadow.size(2) * 0.4)] = 0
                shadow = expand(shadow, num=30, soft=o.soft_shadow)
            take = behind * w_seen
            ref_img = take * w_img + (1 - take) * ref_img
            behind = (1 - take) * behind
            if o.ii_last_only:
                break
        ref_mask = 1 - (1 - mask[:, ref]) * (1 - behind)
        if o.fix_mask:
            ref_img = inpainter(ref_img, ref_mask, is_masked=False)
        else:
            ref_img = inpainter((1 - mask[:, ref]) * (1 - behind) * ref_img, ref_mask)
        return ref_img, ref_mask, shadow

    def _border_objects(self, inpainter, raw_output, alpha_ctx, pred_flow, ctx_len, warper, grid, ref):
        """Objects cut by the left / right image border in the last prediction: polygon mask of the
        region they enter from, inpainted appearance, and their flow to every predicted frame
        (wif.py:130-172).  Returns a list of (mask, appearance, flow (B, Tp, Hd, Wd, 2))."""
        border = 3  # pixels
        h, w = self.src_grid_hd.shape[1:3]

        def to_px(g):
            return torch.stack([(g[..., 0] * w + w - 1) / 2, (g[..., 1] * h + h - 1) / 2], dim=-1)

        pred_px = to_px(pred_flow[:, -1, -1].permute(0, 2, 3, 1) + self.src_grid_hd)
        orig_px = to_px(self.src_grid_hd)
        all_obj = (((alpha_ctx[:, :, -1, 1:] + 1) / 2).max(dim=1)[0] > 0.9).float()     # B No H W
        out = []
        for side, at_border in (("left", pred_px[..., 0] < border), ("right", pred_px[..., 0] >= w - border)):
            hit = at_border.float().unsqueeze(1) * all_obj
            if not hit.sum() > 0:
                continue
            obj_id = int(hit.flatten(start_dim=2).sum(-1).argmax(dim=1)[0])
            sel = hit[:, obj_id].bool()
            bv, ov = pred_px[sel], orig_px[sel]
            # (the six extrema in ONE device -> host read; the reference takes a `float()` of each)
            by0, by1, ox0, ox1, oy0, oy1 = torch.stack([bv[:, 1].min(), bv[:, 1].max(), ov[:, 0].min(), ov[:, 0].max(),
                                                        ov[:, 1].min(), ov[:, 1].max()]).tolist()
            if side == "left":
                corners = [(0, by0), (0, by1), (ox1, oy1), (ox1, oy0)]
            else:
                corners = [(ox0, oy0), (ox0, oy1), (w - 1, by1), (w - 1, by0)]
            region = point_in_polygon(orig_px, corners).float()
            look = inpainter((1 - region) * raw_output[:, -1, -1, :3], region)
            out.append((region, look, warper.grid_to_obj_flow_from_ref_to_pred(grid, ctx_len, ref, obj_id)))
        return out

    def inpaint(self, inpainter, raw_output, alpha, alpha_ctx, real_vid, pred_flow, ctx_len, warper, grid):
        """``WIF.inpaint`` (wif.py:58-226): fuse the warped context frames, inpaint the disoccluded
        background ONCE in a reference frame (the last prediction) and propagate it to the other
        predicted frames along the background flow.  `inpainter(img, mask, ...)` is the external
        MAT network (out of scope: any callable with the reference's signature).
        Returns (B, ctx_len + Tp, 3, H, W)."""
        o = self.opt
        if o.use_inpainter:
            mask, obj_mask = self._holes(alpha_ctx)
        if not o.loop_ii:
            vid = self.forward(raw_output)
            if o.use_inpainter:
                for t in range(vid.size(1)):
                    if o.inpaint_obj:
                        keep = (1 - mask[:, t]) * (1 - obj_mask[:, t])
                        # (the reference calls a `self.inpainter` it never sets here, wif.py:218)
                        fill = inpainter(keep * vid[:, t], 1 - keep)
                        vid[:, t] = (1 - mask[:, t]) * vid[:, t] + mask[:, t] * fill
                    else:
                        vid[:, t] = inpainter((1 - mask[:, t]) * vid[:, t], mask[:, t])
            return torch.cat([real_vid[:, :ctx_len], vid], dim=1)

        tp = raw_output.size(2)
        frames = [self.forward(raw_output[:, :, t:t + 1]) for t in range(tp)]
        if o.use_inpainter:
            if not (o.inpaint_obj and o.propagate_unique):
                raise AssertionError("loop_ii with an inpainter needs inpaint_obj and propagate_unique (wif.py:85-86)")
            ref = -1  # the last predicted frame is inpainted and propagated to the earlier ones
            ref_to_pred = warper.grid_to_bg_flow_from_ref_to_pred(grid, ctx_len, ref)
            ref_img, ref_mask, shadow = self._reference_background(inpainter, frames, mask, obj_mask, alpha,
                                                                   real_vid, ctx_len, warper, grid, ref)
            entering = self._border_objects(inpainter, raw_output, alpha_ctx, pred_flow, ctx_len, warper, grid,
                                            ref) if o.propagate_obj else []
            # (the kernels carry no gradient: under autograd with a differentiable frame the spelled-out loop runs)
            fused = (self.fuse_propagate and len(entering) <= 2 and all(x.dtype == torch.float32 for x in (ref_img, mask))
                     and not (torch.is_grad_enabled() and any(x.requires_grad for x in (*frames, ref_img))))
            for t in range(tp):
                img, todo = frames[t].squeeze(1), mask[:, t]
                if fused:  # the rest of this loop body as two launches around the inpainter (csrc/inpaint_ops.hip)
                    img, todo, inp_img, inp_mask = WF.inpaint_propagate(
                        ref_to_pred[:, t], self.src_grid_hd, ref_img, ref_mask, shadow if o.use_shadows else None,
                        [(region, look, flow[:, t]) for region, look, flow in entering], img, todo, obj_mask[:, t],
                        soft_shadow=o.soft_shadow, fix_mask=o.fix_mask)
                    if o.fix_mask:
                        fill = inpainter(inp_img, expand(inp_mask, 3), exp=False, is_masked=False)
                    else:
                        fill = inpainter(inp_img, inp_mask)
                    frames[t] = WF.inpaint_blend(img, todo, fill).unsqueeze(1)
                    continue
                w_img = self._warp(ref_img, ref_to_pred[:, t])
                w_mask = self._warp_mask(ref_mask, ref_to_pred[:, t])
                for region, look, flow in entering:
                    w_region = self._warp_mask(region, flow[:, t])
                    w_look = self._warp(look, flow[:, t])
                    w_mask = 1 - (1 - w_mask) * (1 - w_region)
                    todo = 1 - (1 - todo) * (1 - w_region)
                    w_img = (1 - w_region) * w_img + w_region * w_look
                obj_t = obj_mask[:, t]
                if o.use_shadows:
                    w_shadow = self._warp(shadow, ref_to_pred[:, t])
                    if not o.soft_shadow:
                        w_shadow = (w_shadow > 1 - _MASK_T).float()
                    todo = todo * (1 - w_shadow * (1 - obj_t))
                take = todo * w_mask
                img = take * w_img + (1 - take) * img
                todo = (1 - take) * todo
                if o.fix_mask:
                    fill = inpainter(img, expand(1 - (1 - todo) * (1 - obj_t), 3), exp=False, is_masked=False)
                else:
                    keep = (1 - todo) * (1 - obj_t)
                    fill = inpainter(keep * img, 1 - keep)
                frames[t] = ((1 - todo) * img + todo * fill).unsqueeze(1)
        return torch.cat([real_vid[:, :ctx_len], torch.cat(frames, dim=1)], dim=1)
