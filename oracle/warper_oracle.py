"""CPU oracle for the Warper / WIF glue around the hot ops -- TEST INFRASTRUCTURE, NOT PRODUCT.

Functional PyTorch-CPU restatement of ``models/nets/lvd.py:Warper`` (lines 469-870), the two LVD
modes that call it (lvd.py:126-153) and the WIF fusion arithmetic (models/nets/wif.py:37-57).
Pinned like ``wif_oracle``: golden vectors generated from the reference by
``oracle/make_golden.py`` + live differential tests when /root/reference is present.

Written as free functions over an explicit ``WarperCfg`` instead of an nn.Module with hidden
state, with the per-layer composites spelled as loops (``wif_oracle.occlusion_product``) -- the
reference materialises (L, L, h, w) broadcasts instead.  Shapes use the reference's letters:
B batch, T frames, Tc context frames, Tp predicted frames, No objects, L = No + 1 layers,
(H, W) low-res raster, (Hd, Wd) high-res raster, (Ho, Wo) object canvas.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import wif_oracle as O


@dataclass
class WarperCfg:
    """The option fields Warper reads (lvd.py:472-499), with the derived shapes."""
    latent_shape: tuple
    obj_shape: tuple
    num_obj: int
    patch_size: int
    scale_factor: float
    dim: int
    aspect_ratio: float
    load_dim: int
    weight_cls: bool = False
    min_cls: float = 0.0
    include_self: bool = False
    no_filter: bool = False
    allow_ghost: bool = False

    @property
    def tgt_shape(self):  # object canvas (Ho, Wo), lvd.py:478
        return (int(self.obj_shape[0] * self.patch_size * self.scale_factor),
                int(self.obj_shape[1] * self.patch_size * self.scale_factor))

    @property
    def src_shape(self):  # (H, W), lvd.py:479
        return (self.dim, int(self.dim * self.aspect_ratio))

    @property
    def src_shape_hd(self):  # (Hd, Wd), lvd.py:480
        return (self.load_dim, int(self.load_dim * self.aspect_ratio)) if self.load_dim > 0 else self.src_shape

    @property
    def scale_hd(self):  # lvd.py:495
        return self.load_dim / self.dim if self.load_dim > 0 else 1

    @property
    def fast(self):  # lvd.py:494
        return self.load_dim == 0

    @classmethod
    def from_opt(cls, opt):
        return cls(tuple(opt.latent_shape), tuple(opt.obj_shape), opt.num_obj, opt.patch_size,
                   opt.scale_factor, opt.dim, opt.aspect_ratio, opt.load_dim, opt.weight_cls,
                   opt.min_cls, opt.include_self, opt.no_filter, opt.allow_ghost)


def rescale(t, factor):
    """``scale()`` of lvd.py:175-179: bilinear F.interpolate on the last two dims of any-rank t."""
    if factor == 1:
        return t
    lead = t.shape[:-3]
    out = F.interpolate(t.reshape(-1, *t.shape[-3:]), scale_factor=factor, mode="bilinear")
    return out.reshape(*lead, *out.shape[-3:])


def gather_time(t, ts):
    """lvd.py:462-467: t (B, T, ...), ts (B, Tc, Tp) long -> (B, Tc, Tp, ...)."""
    b, tc, tp = ts.shape
    idx = ts.reshape(b, tc * tp)
    out = torch.stack([t[i].index_select(0, idx[i]) for i in range(b)])
    return out.reshape(b, tc, tp, *t.shape[2:])


def layer_flows(grid, ctx_ts, pred_ts):
    """The layer-space flows and the predicted frames' grids as ``Warper.grid_to_flow[_ctx]`` spells them
    (lvd.py:660-668 / 779-787), the expressions ``waldo_time_gather_*`` replaces:
    (obj_flow (B,Tc,Tp,No,2,Ho,Wo), bg_flow (B,Tc,Tp,2,H,W), sgo (B,Tc,Tp,No,H,W,2), sgb (B,Tc,Tp,H,W,2))."""
    tgo, sgo, tgb, sgb = grid
    tc = ctx_ts.shape[1]
    obj_flow = (gather_time(tgo, ctx_ts) - tgo[:, pred_ts].unsqueeze(1)).permute(0, 1, 2, 3, 6, 4, 5)
    bg_flow = (gather_time(tgb, ctx_ts) - tgb[:, pred_ts].unsqueeze(1)).permute(0, 1, 2, 5, 3, 4)
    sgo_p = sgo[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1, -1, -1)
    sgb_p = sgb[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1, -1)
    return obj_flow, bg_flow, sgo_p, sgb_p


# --------------------------------------------------------------------------------------
# A13: the four grids
# --------------------------------------------------------------------------------------
def warper_grids(cfg, obj_pose, bg_pose, invert=True):
    """Warper.forward, lvd.py:855-870.  obj_pose (B,T,No,Lo,2), bg_pose (B,T,1,Lb,2) ->
    tgt_grid_obj (B,T,No,Ho,Wo,2), src_grid_obj (B,T,No,H,W,2), tgt_grid_bg (B,T,H,W,2),
    src_grid_bg (B,T,H,W,2)."""
    b, t, no = obj_pose.shape[:3]
    ho, wo = cfg.tgt_shape
    h, w = cfg.src_shape
    inv_o, rep_o = O.tps_init(ho, wo, O.get_grid(*cfg.obj_shape).view(-1, 2))
    inv_b, rep_b = O.tps_init(h, w, O.get_grid(*cfg.latent_shape).view(-1, 2))
    tgo = O.tps_grid(inv_o, rep_o, obj_pose.reshape(b * t * no, -1, 2), ho, wo)
    tgb = O.tps_grid(inv_b, rep_b, bg_pose.reshape(b * t, -1, 2), h, w)
    sgo = sgb = None
    if invert:
        sgo = O.inverse_warp(tgo, (h, w)).view(b, t, no, h, w, 2)
        sgb = O.inverse_warp(tgb, (h, w), erode=False).view(b, t, h, w, 2)
    return tgo.view(b, t, no, ho, wo, 2), sgo, tgb.view(b, t, h, w, 2), sgb


# --------------------------------------------------------------------------------------
# A4/A5: layer <-> image resampling
# --------------------------------------------------------------------------------------
def obj_to_output(cfg, obj, src_grid_obj, delta=1):
    """lvd.py:537-548.  obj (B,[T],No,C,Ho,Wo) sampled at src_grid_obj (B,T,No,H,W,2)."""
    b, t, no, h, w, _ = src_grid_obj.shape
    if obj.ndim == 5:
        obj = obj.unsqueeze(1).expand(-1, t, -1, -1, -1, -1)
    c = obj.shape[3]
    out = O.grid_sample_delta(obj.reshape(b * t * no, c, *obj.shape[-2:]),
                              src_grid_obj.reshape(b * t * no, h, w, 2), delta)
    return out.view(b, t, no, c, h, w)


def bg_to_output(cfg, bg, src_grid_bg, delta=1):
    """lvd.py:550-559.  bg (B,[T],C,H,W) sampled at src_grid_bg (B,T,H,W,2) -> (B,T,1,C,H,W)."""
    b, t, h, w, _ = src_grid_bg.shape
    if bg.ndim == 4:
        bg = bg.unsqueeze(1).expand(-1, t, -1, -1, -1)
    c = bg.shape[2]
    out = O.grid_sample_delta(bg.reshape(b * t, c, h, w), src_grid_bg.reshape(b * t, h, w, 2), delta)
    return out.view(b, t, 1, c, h, w)


def layer_to_output(cfg, obj, bg, grid, delta_bg=1, delta_obj=1):
    """lvd.py:533-535: background first, then the objects, along dim 2."""
    _, sgo, _, sgb = grid
    return torch.cat([bg_to_output(cfg, bg, sgb, delta_bg), obj_to_output(cfg, obj, sgo, delta_obj)], dim=2)


def layer_from_input(cfg, inp, grid):
    """lvd.py:502-531: image -> layer space.  inp (B,T,C,H,W) (or (B,T,L,C,H,W) per layer)."""
    tgo, _, tgb, _ = grid
    b, t = inp.shape[:2]
    no = cfg.num_obj
    h, w = cfg.src_shape
    ho, wo = cfg.tgt_shape
    c = inp.shape[-3]
    if inp.ndim == 5:
        for_obj = inp.view(b * t, 1, c, h, w).expand(-1, no, -1, -1, -1)
        for_bg = inp
    else:
        for_obj = inp[:, :, 1:]
        for_bg = inp[:, :, :1]
    obj = O.grid_sample(for_obj.reshape(b * t * no, c, h, w), tgo.reshape(b * t * no, ho, wo, 2))
    bg = O.grid_sample(for_bg.reshape(b * t, c, h, w), tgb.reshape(b * t, h, w, 2))
    return obj.view(b, t, no, c, ho, wo), bg.view(b, t, c, h, w)


def alpha_to_alpha(cfg, obj_alpha, bg_alpha, grid, occ):
    """lvd.py:561-573."""
    _, sgo, _, _ = grid
    b, t, no = sgo.shape[:3]
    oa = obj_alpha.unsqueeze(1).expand(-1, t, -1, -1, -1, -1)
    ba = bg_alpha.unsqueeze(1).expand(-1, t, -1, -1, -1)
    out = (layer_to_output(cfg, oa, ba, grid) + 1) / 2          # B T L 1 H W
    prod = _occ_factor(out.squeeze(3), occ).unsqueeze(3)        # prod_i (1 - a_i occ_ij)
    output_alpha = prod * out
    obj_occ, bg_occ = layer_from_input(cfg, prod, grid)
    return obj_occ * (oa + 1) - 1, bg_occ * (ba + 1) - 1, output_alpha


def _occ_factor(alpha, occ):
    """prod_i (1 - alpha_i occ[i, j]) for every j; alpha (..., L, h, w), occ (..., L, L)."""
    nl = alpha.shape[-3]
    outs = []
    for j in range(nl):
        p = torch.ones_like(alpha[..., 0, :, :])
        for i in range(nl):
            p = p * (1 - alpha[..., i, :, :] * occ[..., i, j, None, None])
        outs.append(p)
    return torch.stack(outs, dim=-3)


# --------------------------------------------------------------------------------------
# A14: flow helpers used by WIF.inpaint
# --------------------------------------------------------------------------------------
def grid_to_bg_flow_from_ref_to_pred(cfg, grid, ctx_len, ref):
    """lvd.py:575-582."""
    _, _, tgb, sgb = grid
    fl = (tgb[:, [ref]] - tgb[:, ctx_len:]).permute(0, 1, 4, 2, 3)
    fl = bg_to_output(cfg, fl, sgb[:, ctx_len:], 0).squeeze(2)
    return rescale(fl, cfg.scale_hd).permute(0, 1, 3, 4, 2)


def grid_to_obj_flow_from_ref_to_pred(cfg, grid, ctx_len, ref, obj_id):
    """lvd.py:584-591."""
    tgo, sgo, _, _ = grid
    fl = (tgo[:, [ref], [obj_id]] - tgo[:, ctx_len:, [obj_id]]).permute(0, 1, 2, 5, 3, 4)
    b, t, _, h, w, _ = sgo[:, ctx_len:, [obj_id]].shape
    out = O.grid_sample(fl.reshape(b * t, 2, *fl.shape[-2:]),
                        sgo[:, ctx_len:, [obj_id]].reshape(b * t, h, w, 2)).view(b, t, 2, h, w)
    return rescale(out, cfg.scale_hd).permute(0, 1, 3, 4, 2)


def grid_to_bg_flow_from_ctx_to_ref(cfg, grid, ctx_len, ref):
    """lvd.py:593-600."""
    _, _, tgb, sgb = grid
    fl = (tgb[:, :ctx_len] - tgb[:, [ref]]).permute(0, 1, 4, 2, 3)
    fl = bg_to_output(cfg, fl, sgb[:, [ref]].repeat(1, ctx_len, 1, 1, 1), 0).squeeze(2)
    return rescale(fl, cfg.scale_hd).permute(0, 1, 3, 4, 2)


# --------------------------------------------------------------------------------------
# A9: flow / alpha synthesis
# --------------------------------------------------------------------------------------
def lyt_dist(alpha_obj, lyt, cls, weight_cls, min_cls):
    """Class distribution of every object (lvd.py:627-634 / 737-744): the layout logits averaged
    over the object's alpha window (optionally weighted by how well each pixel's class agrees with
    the predicted class vector), then a softmax.
    alpha_obj (B,Tw,No,1,H,W), lyt (B,Tw,Nl,H,W), cls (B,No,Nl) or None -> (B,No,Nl)."""
    win = alpha_obj + 1e-6                                              # B Tw No 1 H W
    if weight_cls:
        prob = lyt.softmax(dim=2)                                       # B Tw Nl H W
        wcls = torch.einsum("bon,btnhw->btohw", cls + min_cls, prob).unsqueeze(3)
        win = win * wcls
    total = win.sum(dim=(1, 4, 5))                                      # B No 1
    mean = torch.einsum("btohw,btnhw->bon", win.squeeze(3), lyt) / total  # B No Nl
    return mean.softmax(dim=2)                                          # B No Nl


def _lyt_alpha(cfg, alpha_obj, lyt, hd_lyt, cls):
    """Layout filter (lvd.py:624-639 / 731-751): per object, 1 - 0.5 * L1 distance between the
    object's class distribution and the per-pixel class distribution at HD.
    alpha_obj (B,Tw,No,1,H,W), lyt (B,Tw,Nl,H,W), hd_lyt (B,Tw,Nl,Hd,Wd) -> (B,Tw,No,1,Hd,Wd)."""
    no = alpha_obj.shape[2]
    hd_prob = hd_lyt.softmax(dim=2)                                     # B Tw Nl Hd Wd
    if cls is None or cfg.weight_cls:
        dist = lyt_dist(alpha_obj, lyt, cls, cfg.weight_cls, cfg.min_cls)
    else:
        dist = cls                                                      # B No Nl
    out = []
    for o in range(no):
        d = (dist[:, None, o, :, None, None] - hd_prob).abs().sum(dim=2, keepdim=True)  # B Tw 1 Hd Wd
        out.append(1 - d / 2)
    return torch.stack(out, dim=2)                                      # B Tw No 1 Hd Wd


def grid_to_flow_ctx(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts):
    """Warper.grid_to_flow_ctx, lvd.py:707-828 (restrict_to_ctx inference path)."""
    return _grid_to_flow(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only=True)


def grid_to_flow(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts):
    """Warper.grid_to_flow, lvd.py:602-705 (training path: alpha composited on all T frames)."""
    return _grid_to_flow(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only=False)


def _grid_to_flow(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, ctx_only):
    tgo, sgo, tgb, sgb = grid
    b, _, no = sgo.shape[:3]
    tc, tp, t = ctx_ts.shape[1], pred_ts.shape[0], inp.shape[1]
    nl_layers = no + 1
    h, w = cfg.src_shape
    hd, wd = cfg.src_shape_hd
    ho, wo = cfg.tgt_shape
    hd_inp = inp
    lo_inp = rescale(hd_inp, 1 / cfg.scale_hd)
    win = slice(0, tc) if ctx_only else slice(0, t)          # frames alpha is composited on
    to_ctx = (lambda x: gather_time(x, ctx_ts))

    # 1. rough alpha of every layer in image space (delta 0), lvd.py:617-621 / 723-728
    oa = ((obj_alpha + 1) / 2).unsqueeze(1).expand(-1, t, -1, -1, -1, -1)
    ba = ((bg_alpha + 1) / 2).unsqueeze(1).expand(-1, t, -1, -1, -1)
    alpha = layer_to_output(cfg, oa, ba, grid, 0, 0)[:, win]          # B Tw L 1 H W
    # 2./3. layout filter and HD upsampling, lvd.py:624-647 / 731-760
    filt = (not cfg.no_filter) or ctx_only  # grid_to_flow_ctx ignores no_filter
    if filt:
        la = _lyt_alpha(cfg, alpha[:, :, 1:], lo_inp[:, win, 3:], hd_inp[:, win, 3:], cls)
    alpha = rescale(alpha, cfg.scale_hd)
    if filt:
        alpha = torch.cat([alpha[:, :, :1], alpha[:, :, 1:] * la], dim=2)
    # 4. occlusion composite #1, lvd.py:650-653 / 763-766
    occ6 = occ.reshape(b, t, nl_layers, nl_layers)
    alpha = O.occlusion_product(alpha.squeeze(3), occ6[:, win]).unsqueeze(3)   # B Tw L 1 Hd Wd
    alpha_unflt = alpha
    # 5. per-layer flow in layer space between context and predicted frames, lvd.py:663-667 / 779-782
    obj_flow = (to_ctx(tgo) - tgo[:, pred_ts].unsqueeze(1)).permute(0, 1, 2, 3, 6, 4, 5)  # B Tc Tp No 2 Ho Wo
    bg_flow = (to_ctx(tgb) - tgb[:, pred_ts].unsqueeze(1)).permute(0, 1, 2, 5, 3, 4)      # B Tc Tp 2 H W
    sgo_p = sgo[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1, -1, -1).reshape(b * tc, tp, no, h, w, 2)
    sgb_p = sgb[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1, -1).reshape(b * tc, tp, h, w, 2)
    gridp = (None, sgo_p, None, sgb_p)
    # 6. ghost suppression (ctx variant only), lvd.py:784-791
    is_obj = 1
    if ctx_only and not cfg.allow_ghost:
        ones = torch.ones(b * tc, tp, no, 1, ho, wo, dtype=sgo_p.dtype)
        io = obj_to_output(cfg, ones, sgo_p, 0)
        io = (rescale(io, cfg.scale_hd) > 0.9).to(io.dtype).view(b, tc, tp, no, 1, hd, wd)
        is_obj = torch.cat([torch.ones_like(io[:, :, :, :1]), io], dim=3)
    # 7. flow of every layer warped to image space and upsampled, lvd.py:670-674 / 792-796
    flow = layer_to_output(cfg, obj_flow.reshape(b * tc, tp, no, 2, ho, wo),
                           bg_flow.reshape(b * tc, tp, 2, h, w), gridp, 0, 0)
    flow = rescale(flow.view(b, tc, tp, nl_layers, 2, h, w), cfg.scale_hd)      # B Tc Tp L 2 Hd Wd
    samp = O.get_grid(hd, wd).to(flow.dtype) + flow.permute(0, 1, 2, 3, 5, 6, 4).reshape(-1, hd, wd, 2)
    # 8. context alpha warped by the flow, lvd.py:677-681 / 799-803
    actx = to_ctx(alpha)
    actx = O.grid_sample(actx.reshape(-1, 1, hd, wd), samp).reshape(b, tc, tp, nl_layers, 1, hd, wd) * is_obj
    disocc = actx.max(dim=3)[0]
    # 9. occlusion composite #2 with the occlusion order of the predicted frames, lvd.py:684-693 / 806-816
    occp = occ6[:, pred_ts].unsqueeze(1).expand(-1, tc, -1, -1, -1)           # B Tc Tp L L
    actx = O.occlusion_product(actx.squeeze(4), occp).unsqueeze(4)
    # 10. soft-alpha flow compositing, lvd.py:696 / 818
    flow = (actx * flow).sum(dim=3)
    alpha_out = alpha.squeeze(3) * 2 - 1
    unflt = alpha_unflt.squeeze(3) * 2 - 1 if cfg.fast else None
    return flow, unflt, alpha_out, actx.squeeze(4) * 2 - 1, disocc


# --------------------------------------------------------------------------------------
# A10: frame warp + temporal fusion
# --------------------------------------------------------------------------------------
def input_to_output(cfg, inp, alpha, flow, ctx_ts, eps=1e-6):
    """Warper.input_to_output, lvd.py:830-853."""
    b, tc, tp = flow.shape[:3]
    hd, wd = cfg.src_shape_hd
    c = inp.shape[-3]
    samp = O.get_grid(hd, wd).to(flow.dtype) + flow.permute(0, 1, 2, 4, 5, 3).reshape(b * tc * tp, hd, wd, 2)
    warped = O.grid_sample(gather_time(inp, ctx_ts).reshape(b * tc * tp, c, hd, wd), samp)
    warped = warped.reshape(b, tc, tp, c, hd, wd)
    score = ((alpha + 1) / 2).sum(dim=3, keepdim=True)
    if cfg.include_self and tp == inp.shape[1]:
        score = torch.cat([score, torch.ones_like(score[:, :1])], dim=1)
        alpha = torch.cat([alpha, torch.ones_like(alpha[:, :1])], dim=1)
        warped = torch.cat([warped, inp.unsqueeze(1)], dim=1)
    raw = torch.cat([warped, alpha], dim=3)
    out = torch.cat([warped, score * 2 - 1], dim=3)
    wgt = score + eps
    wgt = wgt / wgt.abs().sum(dim=1, keepdim=True).clamp_min(1e-12)
    return (out * wgt).sum(dim=1), raw


def decode_output(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts, restrict_to_ctx,
                  use_disocc=False):
    """LVD.forward(mode="decode_output"), lvd.py:141-153."""
    fn = grid_to_flow_ctx if restrict_to_ctx else grid_to_flow
    flow, unflt, alpha, actx, disocc = fn(cfg, inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
    out, raw = input_to_output(cfg, inp, actx, flow, ctx_ts)
    raw_alpha = out[:, :, -1:]
    if use_disocc:
        if cfg.include_self:
            disocc = torch.cat([disocc, torch.ones_like(disocc[:, :1])], dim=1)
        raw = torch.cat([raw, disocc], dim=3)
    return out[:, :, :-1], flow, unflt, alpha, raw_alpha, raw, actx


# --------------------------------------------------------------------------------------
# A12: WIF fusion epilogue
# --------------------------------------------------------------------------------------
def wif_fuse(vid, net_out, ab=True):
    """The fusion arithmetic of WIF.forward with ii_score (models/nets/wif.py:49-54).
    vid (B, T, Tc, C, H, W) is the UNet input after the permute of wif.py:39; net_out
    (B, T, Tc, 4|5, H, W) the UNet output.  beta = out[0:3], score = softmax_Tc(out[3]);
    the blending weight sigma(x_4 + 5) uses INPUT channel 4 (wif.py:53), not a UNet output."""
    beta = net_out[:, :, :, :3]
    score = net_out[:, :, :, 3:4].softmax(dim=2)
    a = torch.sigmoid(vid[:, :, :, 4:5] + 5) if ab else 0
    return ((a * vid[:, :, :, :3] + beta) * score).sum(dim=2)
