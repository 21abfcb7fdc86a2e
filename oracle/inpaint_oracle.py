"""CPU restatement of WIF.inpaint (models/nets/wif.py:58-235) -- TEST INFRASTRUCTURE ONLY.

Flat, in the reference's own order of operations; every resampling goes through the oracle's
``grid_sample`` restatement and the A14 flow helpers of ``warper_oracle``.  Pinned by
``tests/golden/wif_inpaint_*.npz`` (outputs of the reference itself, ``oracle/make_golden.py``).
"""
import torch

from . import warper_oracle as WO
from . import wif_oracle as O


def expand(mask, num=1, dir=None, soft=False, alpha=0.97):
    """tools/utils.py:300-323 (does not modify its argument)."""
    mask = mask.clone() if soft else mask.bool()
    for _ in range(num):
        for d, dst, src in (("south", (slice(1, None), slice(None)), (slice(None, -1), slice(None))),
                            ("north", (slice(None, -1), slice(None)), (slice(1, None), slice(None))),
                            ("east", (slice(None), slice(1, None)), (slice(None), slice(None, -1))),
                            ("west", (slice(None), slice(None, -1)), (slice(None), slice(1, None)))):
            if dir and dir != d:
                continue
            a = mask[(slice(None), slice(None)) + dst]
            b = mask[(slice(None), slice(None)) + src]
            mask[(slice(None), slice(None)) + dst] = torch.maximum(a, alpha * b) if soft else (a | b)
    return mask if soft else mask.float()


def stub_inpainter(img, mask, exp=True, is_masked=True):
    """Deterministic stand-in for the MAT inpainter in tests and golden generation: fills the
    hole with a smooth function of the visible part."""
    vis = (img * (1 - mask)).sum(dim=(2, 3), keepdim=True) / (1 - mask).sum(dim=(2, 3), keepdim=True).clamp_min(1.0)
    ramp = torch.linspace(-0.2, 0.2, img.shape[-1], device=img.device).view(1, 1, 1, -1)
    return img * (1 - mask) + mask * (vis + ramp)


def point_in_polygon(pts, corners):
    import matplotlib.path as mplt_path
    b, h, w, _ = pts.shape
    inside = mplt_path.Path(corners).contains_points(pts.reshape(-1, 2).numpy())
    return torch.from_numpy(inside).view(b, 1, h, w)


def wif_inpaint(opt, cfg, forward, inpainter, raw_output, alpha, alpha_ctx, real_vid, pred_flow, ctx_len, grid):
    """`forward(vid)` is WIF.forward (UNet + fusion), `cfg` the WarperCfg of the flows."""
    hd, wd = cfg.src_shape_hd
    ident = O.get_grid(hd, wd)
    thr = 0.1

    def warp(x, flow):
        return O.grid_sample(x, flow + ident)

    if opt.use_inpainter:
        cover = ((alpha_ctx + 1) / 2).sum(dim=3, keepdim=True)
        objs = ((alpha_ctx[:, :, :, 1:] + 1) / 2).sum(dim=3, keepdim=True)
        cover, objs = (cover[:, -1], objs[:, -1]) if opt.ii_last_only else (cover.max(dim=1)[0], objs.max(dim=1)[0])
        mask = 1 - cover
        mask = (mask > thr).float() if opt.fix_thresh else (mask > 1 - thr).float()
        obj_mask = (objs > 0.9).float()
        if opt.use_expansion:
            mask = expand(mask, num=opt.num_expansion) * (1 - obj_mask)
    if not opt.loop_ii:
        vid = forward(raw_output).clone()
        if opt.use_inpainter:
            for t in range(vid.shape[1]):
                if opt.inpaint_obj:
                    keep = (1 - mask[:, t]) * (1 - obj_mask[:, t])
                    vid[:, t] = (1 - mask[:, t]) * vid[:, t] + mask[:, t] * inpainter(keep * vid[:, t], 1 - keep)
                else:
                    vid[:, t] = inpainter((1 - mask[:, t]) * vid[:, t], mask[:, t])
        return torch.cat([real_vid[:, :ctx_len], vid], dim=1)

    tp = raw_output.shape[2]
    frames = [forward(raw_output[:, :, t:t + 1]) for t in range(tp)]
    if opt.use_inpainter:
        assert opt.inpaint_obj and opt.propagate_unique
        ref = -1
        r2p = WO.grid_to_bg_flow_from_ref_to_pred(cfg, grid, ctx_len, ref)
        c2r = WO.grid_to_bg_flow_from_ctx_to_ref(cfg, grid, ctx_len, ref)
        ref_img = frames[ref].squeeze(1)
        omr = obj_mask[:, ref]
        shadow = None
        for t2 in range(ctx_len - 1, -1, -1):
            cm = (alpha[:, t2, :1] > 1 - thr).float()
            wi = warp(real_vid[:, t2], c2r[:, t2])
            wm = (warp(cm, c2r[:, t2]) > 1 - thr).float()
            if opt.use_shadows and t2 == ctx_len - 1:
                shadow = ((wi - ref_img).abs().mean(dim=1, keepdim=True) > 0.25).float() * wm * (1 - omr)
                shadow = 1 - expand(1 - shadow, num=5)
                shadow = expand(shadow, num=5)
                shadow[:, :, :int(shadow.shape[2] * 0.4)] = 0
                shadow = expand(shadow, num=30, soft=opt.soft_shadow)
            inter = omr * wm
            ref_img = inter * wi + (1 - inter) * ref_img
            omr = (1 - inter) * omr
            if opt.ii_last_only:
                break
        ref_mask = 1 - (1 - mask[:, ref]) * (1 - omr)
        if opt.fix_mask:
            ref_img = inpainter(ref_img, ref_mask, is_masked=False)
        else:
            ref_img = inpainter((1 - mask[:, ref]) * (1 - omr) * ref_img, ref_mask)
        sides = []
        if opt.propagate_obj:
            border = 3
            pg = pred_flow[:, -1, -1].permute(0, 2, 3, 1) + ident
            pg = torch.stack([(pg[..., 0] * wd + wd - 1) / 2, (pg[..., 1] * hd + hd - 1) / 2], dim=-1)
            og = torch.stack([(ident[..., 0] * wd + wd - 1) / 2, (ident[..., 1] * hd + hd - 1) / 2], dim=-1)
            allobj = (((alpha_ctx[:, :, -1, 1:] + 1) / 2).max(dim=1)[0] > 0.9).float()
            for left in (True, False):
                at = (pg[..., 0] < border) if left else (pg[..., 0] >= wd - border)
                hit = at.float().unsqueeze(1) * allobj
                if not hit.sum() > 0:
                    continue
                oid = int(hit.flatten(start_dim=2).sum(-1).argmax(dim=1)[0])
                sel = hit[:, oid].bool()
                bv, ov = pg[sel], og[sel]
                if left:
                    corners = [(0, float(bv[:, 1].min())), (0, float(bv[:, 1].max())),
                               (float(ov[:, 0].max()), float(ov[:, 1].max())), (float(ov[:, 0].max()), float(ov[:, 1].min()))]
                else:
                    corners = [(float(ov[:, 0].min()), float(ov[:, 1].min())), (float(ov[:, 0].min()), float(ov[:, 1].max())),
                               (wd - 1, float(bv[:, 1].max())), (wd - 1, float(bv[:, 1].min()))]
                region = point_in_polygon(og, corners).float()
                look = inpainter((1 - region) * raw_output[:, -1, -1, :3], region)
                sides.append((region, look, WO.grid_to_obj_flow_from_ref_to_pred(cfg, grid, ctx_len, ref, oid)))
        for t in range(tp):
            img = frames[t].squeeze(1)
            cur = mask[:, t]
            wi = warp(ref_img, r2p[:, t])
            wm = (warp(ref_mask, r2p[:, t]) > 1 - thr).float()
            for region, look, fl in sides:
                wr = (warp(region, fl[:, t]) > 1 - thr).float()
                wl = warp(look, fl[:, t])
                wm = 1 - (1 - wm) * (1 - wr)
                cur = 1 - (1 - cur) * (1 - wr)
                wi = (1 - wr) * wi + wr * wl
            ot = obj_mask[:, t]
            if opt.use_shadows:
                ws = warp(shadow, r2p[:, t])
                if not opt.soft_shadow:
                    ws = (ws > 1 - thr).float()
                cur = cur * (1 - ws * (1 - ot))
            inter = cur * wm
            img = inter * wi + (1 - inter) * img
            cur = (1 - inter) * cur
            if opt.fix_mask:
                fill = inpainter(img, expand(1 - (1 - cur) * (1 - ot), 3), exp=False, is_masked=False)
            else:
                fill = inpainter((1 - cur) * (1 - ot) * img, 1 - (1 - cur) * (1 - ot))
            frames[t] = ((1 - cur) * img + cur * fill).unsqueeze(1)
    return torch.cat([real_vid[:, :ctx_len], torch.cat(frames, dim=1)], dim=1)
