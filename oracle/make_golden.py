"""Generate golden vectors from the REFERENCE ITSELF (run in the build container only).

    python oracle/make_golden.py

Imports the reference's own modules from /root/reference through ``oracle/ref_import.py``
(nothing is copied), runs them on seeded inputs on the CPU and stores inputs + outputs (+
autograd gradients) as small .npz fixtures under tests/golden/.  The fixtures are data; this
script is the provenance.  InverseWarp vectors are produced with ``Tensor.sort`` forced stable
(see ``ref_import.stable_sort``): the reference's own tie-break is implementation-defined.
"""
import os
import types
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
warnings.filterwarnings("ignore", message="Default grid_sample")


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(out)}")


def lvd_stub(ns, num_obj):
    """An LVD instance with only what compute_occ / reduce_comp touch (lvd.py:49)."""
    lvd = ns.LVD.__new__(ns.LVD)
    torch.nn.Module.__init__(lvd)
    lvd.register_buffer("diag", torch.eye(num_obj, num_obj)[None, None, :, :])
    return lvd


def gen_tps(ns):
    g = torch.Generator().manual_seed(1)
    for tag, (h, w, kh, kw) in {"k16": (16, 32, 4, 4), "k32": (12, 20, 4, 8)}.items():
        ctrl = ns.get_grid(kh, kw).view(-1, 2)
        mod = ns.TPSWarp(h, w, ctrl)
        pts = (ctrl.view(1, -1, 2) + 0.1 * torch.randn(5, kh * kw, 2, generator=g)).requires_grad_()
        grid = mod(pts)
        wgt = torch.randn(grid.shape, generator=g)
        (grid * wgt).sum().backward()
        save(f"tps_{tag}", h=h, w=w, ctrl=ctrl, pts=pts, grid=grid, wgt=wgt, grad_pts=pts.grad,
             inverse_kernel=mod.inverse_kernel, tgt_grid_repr=mod.tgt_grid_repr)


def gen_grid_sample(ns):
    g = torch.Generator().manual_seed(2)
    n, c, hi, wi, ho, wo = 3, 5, 9, 14, 11, 13
    x = torch.randn(n, c, hi, wi, generator=g, requires_grad=True)
    # grid with in-range, border-straddling and far-out-of-range samples
    grid = (torch.rand(n, ho, wo, 2, generator=g) * 2.6 - 1.3)
    grid[0, 0, 0] = torch.tensor([-1.0, -1.0])
    grid[0, 0, 1] = torch.tensor([1.0, 1.0])
    grid[0, 0, 2] = torch.tensor([5.0, -7.0])
    grid[0, 0, 3] = torch.tensor([-1.0 + 1.0 / wi, 1.0 - 1.0 / hi])
    grid.requires_grad_()
    for delta in (0.0, 1.0):
        x.grad = grid.grad = None
        out = F.grid_sample(x + delta, grid) - delta
        wgt = torch.randn(out.shape, generator=g)
        (out * wgt).sum().backward()
        save(f"grid_sample_d{int(delta)}", x=x, grid=grid, delta=delta, out=out, wgt=wgt,
             grad_x=x.grad, grad_grid=grid.grad)


def gen_occ_comp(ns):
    g = torch.Generator().manual_seed(3)
    b, t, no, c, h, w = 2, 3, 4, 3, 6, 10
    lvd = lvd_stub(ns, no)
    score = torch.randn(b, t, no, generator=g, requires_grad=True)
    occ = lvd.compute_occ(score)
    vid = (torch.rand(b, t, no + 1, c + 1, h, w, generator=g) * 2 - 1).requires_grad_()
    flow = torch.randn(b, t - 1, no + 1, 2, h, w, generator=g)
    out, alpha, fl = lvd.reduce_comp(vid, occ, flow)
    w1 = torch.randn(out.shape, generator=g)
    w2 = torch.randn(alpha.shape, generator=g)
    ((out * w1).sum() + (alpha * w2).sum()).backward()
    save("occ_comp", score=score, occ=occ, vid=vid, out=out, alpha=alpha, w1=w1, w2=w2,
         grad_vid=vid.grad, grad_score=score.grad)


def gen_warp_composite(ns):
    """The synthetic hot path of SURVEY.md 8(d) assembled from the reference's own ops:
    TPSWarp.forward -> F.grid_sample -> LVD.reduce_comp."""
    g = torch.Generator().manual_seed(4)
    for tag, (f, nl, h, w, sigma) in {"small": (3, 5, 16, 24, 0.08), "l8": (2, 8, 32, 32, 0.05),
                                      "big_warp": (2, 3, 20, 28, 0.5)}.items():
        ctrl = ns.get_grid(4, 4).view(-1, 2)
        tps = ns.TPSWarp(h, w, ctrl)
        lvd = lvd_stub(ns, nl - 1)
        layers = (torch.rand(f, nl, 4, h, w, generator=g) * 2 - 1).requires_grad_()
        pts = (ctrl.view(1, 16, 2) + sigma * torch.randn(f * nl, 16, 2, generator=g)).requires_grad_()
        score = torch.randn(f, 1, nl - 1, generator=g, requires_grad=True)
        occ = lvd.compute_occ(score)  # f 1 L L
        grid = tps(pts)
        warped = F.grid_sample(layers.view(f * nl, 4, h, w), grid).view(f, 1, nl, 4, h, w)
        rgb, alpha, _ = lvd.reduce_comp(warped, occ, torch.zeros(f, 0, nl, 2, h, w))
        w1 = torch.randn(rgb.shape, generator=g)
        w2 = torch.randn(alpha.shape, generator=g)
        ((rgb * w1).sum() + (alpha * w2).sum()).backward(retain_graph=True)
        both = dict(grad_layers=layers.grad.clone(), grad_pts=pts.grad.clone(),
                    grad_score=score.grad.clone())
        layers.grad = pts.grad = score.grad = None
        # the benchmark's loss: rgb only
        rgb.square().mean().backward()
        save(f"warp_composite_{tag}", layers=layers, pts=pts, score=score, occ=occ[:, 0],
             ctrl=ctrl, rgb=rgb[:, 0], alpha=alpha[:, 0], w1=w1[:, 0], w2=w2[:, 0],
             grad_layers=both["grad_layers"], grad_pts=both["grad_pts"],
             grad_score=both["grad_score"], grad_layers_sq=layers.grad, grad_pts_sq=pts.grad)


def gen_warp_composite_delta(ns):
    """The same path with the reference's "-delta" padding (Warper.obj_to_output / bg_to_output,
    lvd.py:548,559: F.grid_sample(x + delta, grid) - delta; layer_to_output's default is 1)."""
    g = torch.Generator().manual_seed(44)
    for tag, (f, nl, h, w, sigma, delta) in {"delta1": (3, 5, 16, 24, 0.15, 1.0),
                                             "delta1_big": (2, 3, 20, 28, 0.5, 1.0),
                                             "delta_half": (2, 8, 32, 32, 0.2, 0.5)}.items():
        ctrl = ns.get_grid(4, 4).view(-1, 2)
        tps = ns.TPSWarp(h, w, ctrl)
        lvd = lvd_stub(ns, nl - 1)
        layers = (torch.rand(f, nl, 4, h, w, generator=g) * 2 - 1).requires_grad_()
        pts = (ctrl.view(1, 16, 2) + sigma * torch.randn(f * nl, 16, 2, generator=g)).requires_grad_()
        score = torch.randn(f, 1, nl - 1, generator=g, requires_grad=True)
        occ = lvd.compute_occ(score)
        grid = tps(pts)
        warped = (F.grid_sample(layers.view(f * nl, 4, h, w) + delta, grid) - delta).view(f, 1, nl, 4, h, w)
        rgb, alpha, _ = lvd.reduce_comp(warped, occ, torch.zeros(f, 0, nl, 2, h, w))
        w1 = torch.randn(rgb.shape, generator=g)
        w2 = torch.randn(alpha.shape, generator=g)
        ((rgb * w1).sum() + (alpha * w2).sum()).backward()
        save(f"warp_composite_{tag}", layers=layers, pts=pts, score=score, occ=occ[:, 0], ctrl=ctrl,
             delta=delta, rgb=rgb[:, 0], alpha=alpha[:, 0], w1=w1[:, 0], w2=w2[:, 0],
             grad_layers=layers.grad, grad_pts=pts.grad, grad_score=score.grad)


def gen_inverse_warp(ns):
    g = torch.Generator().manual_seed(5)
    ctrl = ns.get_grid(4, 4).view(-1, 2)
    cases = {"obj": (8, 8, 16, 32, True, 0.6, 0.1), "bg": (16, 32, 16, 32, False, 1.0, 0.06),
             "obj2": (12, 20, 24, 40, True, 0.5, 0.15)}
    for tag, (hs, ws, ht, wt, erode, shrink, sigma) in cases.items():
        inv = ns.InverseWarp(hs, ws, ht, wt)
        tps = ns.TPSWarp(hs, ws, ctrl)
        pts = ctrl.view(1, 16, 2) * shrink + sigma * torch.randn(4, 16, 2, generator=g)
        src_grid = tps(pts).detach().requires_grad_()
        with R.stable_sort():
            out = inv(src_grid, erode=erode)
        wgt = torch.randn(out.shape, generator=g)
        # holes carry the constant (2W, 2H) offset: they get no gradient; fine
        (out * wgt).sum().backward()
        save(f"inverse_warp_{tag}", src_grid=src_grid, out=out, wgt=wgt, grad_src_grid=src_grid.grad,
             hs=hs, ws=ws, ht=ht, wt=wt, erode=erode)


def gen_inverse_warp_kernel_size(ns):
    """kernel_size 5 and 7 (warp.py:58-63, 140-146: the fill's Gaussian window; no script passes one, the module takes
    it)."""
    g = torch.Generator().manual_seed(25)
    ctrl = ns.get_grid(4, 4).view(-1, 2)
    for tag, (hs, ws, ht, wt, erode, shrink, sigma, ks, niter) in {
            "k5": (8, 8, 16, 32, True, 0.55, 0.1, 5, 5), "k7": (12, 20, 24, 40, False, 0.5, 0.15, 7, 4)}.items():
        inv = ns.InverseWarp(hs, ws, ht, wt, kernel_size=ks)
        tps = ns.TPSWarp(hs, ws, ctrl)
        pts = ctrl.view(1, 16, 2) * shrink + sigma * torch.randn(3, 16, 2, generator=g)
        src_grid = tps(pts).detach().requires_grad_()
        with R.stable_sort():
            out = inv(src_grid, niter=niter, erode=erode)
        wgt = torch.randn(out.shape, generator=g)
        (out * wgt).sum().backward()
        save(f"inverse_warp_{tag}", src_grid=src_grid, out=out, wgt=wgt, grad_src_grid=src_grid.grad,
             hs=hs, ws=ws, ht=ht, wt=wt, erode=erode, kernel_size=ks, niter=niter)


def gen_inverse_warp_perm(ns):
    """num_perm > 1 (warp.py:91-111): the module's own randperm buffer goes into the fixture."""
    g = torch.Generator().manual_seed(15)
    ctrl = ns.get_grid(4, 4).view(-1, 2)
    for tag, (hs, ws, ht, wt, erode, shrink, sigma, nperm) in {
            "perm3": (8, 8, 16, 32, True, 0.5, 0.12, 3), "perm4": (12, 20, 24, 40, False, 0.45, 0.15, 4)}.items():
        torch.manual_seed(31)  # the perm buffer draws from the global generator
        inv = ns.InverseWarp(hs, ws, ht, wt, num_perm=nperm)
        tps = ns.TPSWarp(hs, ws, ctrl)
        pts = ctrl.view(1, 16, 2) * shrink + sigma * torch.randn(3, 16, 2, generator=g)
        src_grid = tps(pts).detach().requires_grad_()
        with R.stable_sort():
            out = inv(src_grid, erode=erode)
        wgt = torch.randn(out.shape, generator=g)
        (out * wgt).sum().backward()
        save(f"inverse_warp_{tag}", src_grid=src_grid, out=out, wgt=wgt, grad_src_grid=src_grid.grad,
             perm=inv.perm.to(torch.int32), hs=hs, ws=ws, ht=ht, wt=wt, erode=erode)


def gen_warper(ns):
    """Warper.forward -> grid_to_flow_ctx -> input_to_output (the inference chain of
    LVD decode_output, lvd.py:141-153) and grid_to_flow (training), plus WIF.forward's fusion."""
    g = torch.Generator().manual_seed(6)
    opt = R.warper_opt(num_obj=2, weight_cls=True, min_cls=0.05)
    ref = ns.Warper(opt)
    b, t, no, nl = 1, 3, 2, 3
    obj_pose = ns.get_grid(2, 2).view(1, 1, 1, 4, 2) * 0.5 + 0.15 * torch.randn(b, t, no, 4, 2, generator=g)
    bg_pose = ns.get_grid(2, 4).view(1, 1, 1, 8, 2) + 0.05 * torch.randn(b, t, 1, 8, 2, generator=g)
    with R.stable_sort():
        grid = ref(obj_pose, bg_pose)
    hd, wd, h, w, ho, wo = 32, 64, 16, 32, 8, 8
    inp = torch.randn(b, t, 3 + nl, hd, wd, generator=g)
    score = torch.randn(b, t, no, generator=g)
    occ = lvd_stub(ns, no).compute_occ(score)
    obj_alpha = torch.rand(b, no, 1, ho, wo, generator=g) * 2 - 1
    bg_alpha = torch.ones(b, 1, h, w)
    cls = torch.rand(b, no, nl, generator=g).softmax(-1)
    ctx_ts = torch.tensor([[[0, 1], [1, 0]]])
    pred_ts = torch.tensor([2, 1])
    fc = ref.grid_to_flow_ctx(inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
    ft = ref.grid_to_flow(inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
    out, raw = ref.input_to_output(inp, fc[3], fc[0], ctx_ts)
    save("warper_chain", obj_pose=obj_pose, bg_pose=bg_pose, tgo=grid[0], sgo=grid[1], tgb=grid[2],
         sgb=grid[3], inp=inp, occ=occ, obj_alpha=obj_alpha, bg_alpha=bg_alpha, cls=cls,
         ctx_ts=ctx_ts, pred_ts=pred_ts, c_flow=fc[0], c_alpha=fc[2], c_alpha_ctx=fc[3],
         c_disocc=fc[4], t_flow=ft[0], t_alpha=ft[2], t_alpha_ctx=ft[3], t_disocc=ft[4], out=out,
         raw=raw)
    # WIF.forward with a 1x1-conv stand-in for the UNet
    bb, tc, tt, c, hh, ww = 2, 3, 2, 12, 8, 16
    vid = torch.randn(bb, tc, tt, c, hh, ww, generator=g)
    wif = ns.WIF.__new__(ns.WIF)
    torch.nn.Module.__init__(wif)
    wif.score, wif.ab = True, True
    lin = torch.nn.Conv2d(c, 5, 1)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.3)
        lin.bias.copy_(torch.randn(5, generator=g) * 0.1)
    wif.unet = lin
    y = wif(vid)
    save("wif_forward", vid=vid, weight=lin.weight, bias=lin.bias, out=y)


def inpaint_opt(**over):
    d = dict(ii_score=True, ii_ab=True, use_inpainter=True, ii_last_only=False, fix_thresh=True,
             use_expansion=True, num_expansion=2, loop_ii=True, inpaint_obj=True, propagate_unique=True,
             use_shadows=True, soft_shadow=False, fix_mask=False, propagate_obj=True)
    d.update(over)
    return types.SimpleNamespace(**d)


INPAINT_CASES = {
    "a": dict(),
    "b": dict(fix_mask=True, soft_shadow=True, propagate_obj=False, ii_last_only=True, fix_thresh=False,
              use_expansion=False),
    "c": dict(loop_ii=False, inpaint_obj=False),
    "d": dict(loop_ii=False, inpaint_obj=True, use_expansion=False),
}


def inpaint_inputs(ns_get_grid, seed=9):
    """Synthetic but structured inputs of WIF.inpaint: smooth frames, blobby alphas (so that the
    thresholded masks have regions), one object touching the left image border."""
    g = torch.Generator().manual_seed(seed)
    b, ctx_len, tp, no, nl = 1, 2, 2, 2, 3
    t = ctx_len + tp
    hd, wd = 32, 64

    def smooth(*shape, lo=4):
        x = torch.randn(*shape[:-2], shape[-2] // lo, shape[-1] // lo, generator=g)
        lead = x.shape[:-3]
        y = torch.nn.functional.interpolate(x.reshape(-1, *x.shape[-3:]), size=shape[-2:], mode="bilinear")
        return y.reshape(*lead, *y.shape[-3:])

    nlay = no + 1
    c = 3 + nl + nlay
    obj_pose = ns_get_grid(2, 2).view(1, 1, 1, 4, 2) * 0.5 + 0.1 * torch.randn(b, t, no, 4, 2, generator=g)
    bg_pose = ns_get_grid(2, 4).view(1, 1, 1, 8, 2) + 0.04 * torch.randn(b, t, 1, 8, 2, generator=g)
    raw_output = smooth(b, ctx_len, tp, c, hd, wd)
    real_vid = smooth(b, t, 3, hd, wd).clamp(-1, 1)
    alpha = (2.5 * smooth(b, ctx_len, nlay, hd, wd)).tanh()
    alpha[:, :, 0] = alpha[:, :, 0] * 0.3 + 0.7                      # background mostly visible
    alpha_ctx = (2.5 * smooth(b, ctx_len, tp, nlay, hd, wd)).tanh() * 0.5 - 0.45
    alpha_ctx[:, :, :, 0] = (2.0 * smooth(b, ctx_len, tp, hd, wd) + 0.5).tanh()
    alpha_ctx[:, :, -1, 1, 8:20, 0:5] = 0.95                          # an object at the left border
    pred_flow = 0.05 * smooth(b, ctx_len, tp, 2, hd, wd)
    pred_flow[:, -1, -1, 0, 8:20, 0:5] = -0.1                         # ... moving out of the frame
    return dict(obj_pose=obj_pose, bg_pose=bg_pose, raw_output=raw_output, real_vid=real_vid, alpha=alpha,
                alpha_ctx=alpha_ctx, pred_flow=pred_flow, ctx_len=ctx_len, channels=c)


def gen_inpaint(ns):
    """WIF.inpaint (wif.py:58-226) with a 1x1-conv UNet stand-in and the deterministic stub
    inpainter of oracle/inpaint_oracle.py, for four option sets."""
    from oracle.inpaint_oracle import stub_inpainter
    d = inpaint_inputs(ns.get_grid)
    wopt = R.warper_opt(num_obj=2)
    warper = ns.Warper(wopt)
    with R.stable_sort():
        grid = warper(d["obj_pose"], d["bg_pose"])
    g = torch.Generator().manual_seed(10)
    lin = torch.nn.Conv2d(d["channels"], 5, 1)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.3)
        lin.bias.copy_(torch.randn(5, generator=g) * 0.1)
    save("wif_inpaint_inputs", tgo=grid[0], sgo=grid[1], tgb=grid[2], sgb=grid[3], weight=lin.weight,
         bias=lin.bias, **{k: v for k, v in d.items() if torch.is_tensor(v)})
    for tag, over in INPAINT_CASES.items():
        opt = inpaint_opt(**over)
        wif = ns.WIF.__new__(ns.WIF)
        torch.nn.Module.__init__(wif)
        wif.score, wif.ab, wif.opt, wif.unet = True, True, opt, lin
        wif.src_grid_hd = ns.get_grid(32, 64)
        wif.inpainter = stub_inpainter  # the non-loop branch calls an attribute the class never sets
        with torch.no_grad():
            out = wif.inpaint(stub_inpainter, d["raw_output"].clone(), d["alpha"], d["alpha_ctx"], d["real_vid"],
                              d["pred_flow"], d["ctx_len"], warper, grid)
        save(f"wif_inpaint_{tag}", out=out)


def gen_producers(ns):
    """Row f2: the producers of the path's inputs, run through the reference's OWN methods with stub
    `self` objects that replace the networks around them by fixed tensors:
      * ImageDecoder.forward (lvd.py:239-255): `norm` = identity, `to_img` = a fixed raw image;
      * LVD.forward(mode="estimate_alpha_grid_occ") (lvd.py:126-135): `decoder` = the decoder output
        above, `warper` = a no-op; it calls the real compute_occ;
    The pose affine (flp.py:259-273) sits in the middle of PoseDecoder.forward: gen_pose_affine below."""
    g = torch.Generator().manual_seed(11)
    b, no, lo, c_tok = 3, 4, 16, 8
    ho, wo, sf = 16, 16, 4  # decoder raw image 16x16 (obj_shape 4x4 x patch 4), x4 upsampling
    for tag, (use_prior, remove, freeze, masked) in {"plain": (False, False, False, False),
                                                     "prior_mask": (True, False, False, True),
                                                     "remove": (True, True, False, True),
                                                     "freeze": (False, False, True, True)}.items():
        raw = torch.randn(b * no, 1, ho, wo, generator=g).requires_grad_()
        dec = types.SimpleNamespace(norm=lambda x: x, to_img=lambda x, **kw: raw, latent_shape={lo: [4, 4]},
                                    init_bias=5 if tag == "plain" else 0.25, has_alpha=True, use_prior=use_prior,
                                    circle=ns.get_circle([ho, wo], p=0.75).float().view(1, 1, ho, wo),
                                    scale_factor=sf)
        tokens = torch.zeros(b, no, lo, c_tok)
        Ho, Wo, Po = ho * sf, wo * sf, 3 * sf
        mask = torch.ones(Ho, Wo)
        mask[:Po] = 0
        mask[:, :Po] = 0
        mask[-Po:] = 0
        mask[:, -Po:] = 0
        score = torch.randn(b, 2, no, generator=g).requires_grad_()
        lvd = lvd_stub(ns, no)
        lvd.decoder = lambda x: ns.ImageDecoder.forward(dec, x)
        lvd.bg_alpha = torch.ones(1, 1, 8, 16)
        lvd.remove_obj, lvd.freeze_obj = remove, freeze
        lvd.obj_alpha_mask = mask.view(1, 1, 1, Ho, Wo) if masked else 1
        lvd.warper = lambda a, bb: None
        occ, obj_alpha, bg_alpha, _ = ns.LVD.forward(lvd, x_obj=tokens, obj_pose=None, bg_pose=None,
                                                     occ_score=score, mode="estimate_alpha_grid_occ")
        w1 = torch.randn(obj_alpha.shape, generator=g)
        w2 = torch.randn(occ.shape, generator=g)
        ((obj_alpha * w1).sum() + (occ * w2).sum()).backward()
        save(f"producers_{tag}", raw=raw, circle=dec.circle, init_bias=float(dec.init_bias), scale_factor=sf,
             use_prior=int(use_prior), remove=int(remove), freeze=int(freeze), masked=int(masked), mask=mask,
             score=score, occ=occ, obj_alpha=obj_alpha, w1=w1, w2=w2,
             grad_raw=raw.grad if raw.grad is not None else torch.zeros_like(raw), grad_score=score.grad)


def gen_pose_affine(ns):
    """Row f2, the pose heads' affine (models/nets/flp.py:252-273): the reference's OWN
    ``PoseDecoder.forward`` run with a stub `self` whose transformer is empty (no blocks, identity norm)
    and whose two linear heads return fixed tensors -- lines 252-273 then execute as they stand on
    known inputs: tanh of the head outputs, the optional ``+ last``, transform / delta / ``pts @ transform``
    for objects and background, and the scatter into the pose sequences.  Saved: the head outputs, the
    buffers, the returned poses of the predicted frames and the gradients of a weighted sum of them
    with respect to the head outputs."""
    g = torch.Generator().manual_seed(19)
    b, t, ctx_len, no, c = 2, 5, 2, 3, 4
    lat, lat_obj = 8, 4  # background grid 2 x 4, object grid 2 x 2
    for tag, use_last in (("plain", False), ("last", True)):
        head_obj = torch.randn(b * (t - ctx_len), no, 6 + 2 * lat_obj + 1, generator=g).requires_grad_()
        head_bg = torch.randn(b * (t - ctx_len), 1, 6 + 2 * lat, generator=g).requires_grad_()
        last_obj = 0.1 * torch.randn(b, no, 6 + 2 * lat_obj, generator=g)
        last_bg = 0.1 * torch.randn(b, 1, 6 + 2 * lat, generator=g)
        stub = types.SimpleNamespace(
            latent_size=lat, latent_obj_size=lat_obj, num_obj=no, embed_dim=c, cat_z=False, modulate_noise=False,
            self_blocks=[], cross_blocks=[], norm=lambda x: x,
            obj_head=lambda x: head_obj.view(-1, no * (6 + 2 * lat_obj + 1)),
            bg_head=lambda x: head_bg.view(-1, 6 + 2 * lat), use_last=use_last,
            mul_obj=torch.tensor([[[0.25, 0.25, 0.25, 0.25, 1.0, 1.0]]]), bias_obj=torch.tensor([[[0.25, 0.0, 0.0, 0.5, 0.0, 0.0]]]),
            mul_delta_obj=0.2, tgt_pts_obj=ns.get_grid(2, 2).view(1, 1, lat_obj, 2),
            bg_mul=1.2, tgt_pts_bg=ns.get_grid(2, 4).view(1, 1, lat, 2), bias_bg=torch.tensor([[[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]]))
        ctx_mask = (torch.arange(t).view(1, -1) < ctx_len).expand(b, -1)
        x = torch.zeros(b, t, no + 1, c)
        obj_in = torch.randn(b, t, no, lat_obj, 2, generator=g)
        bg_in = torch.randn(b, t, 1, lat, 2, generator=g)
        occ_in = torch.randn(b, t, no, generator=g)
        obj_pose, bg_pose, occ_score = ns.PoseDecoder.forward(stub, obj_in, bg_in, occ_in, x, ctx_mask, last_obj, last_bg)
        pred = ~ctx_mask
        assert torch.equal(obj_pose[ctx_mask], obj_in[ctx_mask]) and torch.equal(bg_pose[ctx_mask], bg_in[ctx_mask])
        po, pb = obj_pose[pred], bg_pose[pred]              # (B (T - ctx), No, Lo, 2), (B (T - ctx), 1, L, 2)
        w1 = torch.randn(po.shape, generator=g)
        w2 = torch.randn(pb.shape, generator=g)
        ((po * w1).sum() + (pb * w2).sum()).backward()
        save(f"pose_affine_{tag}", head_obj=head_obj, head_bg=head_bg, last_obj=last_obj, last_bg=last_bg,
             use_last=int(use_last), frames_per_clip=t - ctx_len, mul_obj=stub.mul_obj, bias_obj=stub.bias_obj,
             mul_delta_obj=float(stub.mul_delta_obj), tgt_pts_obj=stub.tgt_pts_obj, bg_mul=float(stub.bg_mul),
             tgt_pts_bg=stub.tgt_pts_bg, bias_bg=stub.bias_bg, obj_pose=po, bg_pose=pb, occ_score=occ_score[pred],
             w1=w1, w2=w2, grad_head_obj=head_obj.grad, grad_head_bg=head_bg.grad)


def gen_demo_clip():
    """BASELINE config C1: six frames of the in-tree demo clip (datasets/demo_cityscapes, munster),
    reduced 4x with PIL so that the fixture stays small -- frames bilinear (as the loader's own
    resize), class maps nearest -- in the demo set's directory layout, plus the matching flow file
    names' first flow (already under tests/golden/demo_flow.flo).  Data files, not code."""
    import PIL.Image
    src = os.path.join(R.REF_ROOT, "datasets", "demo_cityscapes")
    dst = os.path.join(OUT, "demo_clip")
    city = os.path.join("val", "munster")
    names = sorted(os.listdir(os.path.join(src, "leftImg8bit_sequence_512", city)))[:6]
    for sub, mode, resample in (("leftImg8bit_sequence_512", "RGB", PIL.Image.BILINEAR),
                                ("leftImg8bit_sequence_deeplabv3_512", None, PIL.Image.NEAREST)):
        os.makedirs(os.path.join(dst, sub, city), exist_ok=True)
        for n in names:
            img = PIL.Image.open(os.path.join(src, sub, city, n))
            if mode:
                img = img.convert(mode)
            img = img.resize((256, 128), resample)
            img.save(os.path.join(dst, sub, city, n), optimize=True)
    total = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(dst) for f in fs)
    print(f"demo_clip: {len(names)} frames + layouts, {total / 1024:.1f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = R.load()
    only = set(sys.argv[1:])  # e.g. `make_golden.py inverse_warp_perm`; none = all
    for fn in (gen_tps, gen_grid_sample, gen_occ_comp, gen_warp_composite, gen_warp_composite_delta, gen_inverse_warp,
               gen_inverse_warp_kernel_size, gen_inverse_warp_perm, gen_warper, gen_inpaint, gen_producers, gen_pose_affine):
        if not only or fn.__name__[4:] in only:
            fn(ns)
    if not only or "demo_clip" in only:
        gen_demo_clip()


if __name__ == "__main__":
    main()
