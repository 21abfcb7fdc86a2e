"""GPU parity of the rows around the fused path: InverseWarp (A3), Warper (A9/A10/A13/A14),
compute_occ / reduce_comp (A6) and the WIF fusion epilogue (A12) -- against golden vectors from
the reference and against the CPU oracle on seeded inputs.  Tolerance 1e-4 (north star)."""
import types

import pytest
import torch

from oracle import warper_oracle as WO
from oracle import wif_oracle as O

pytestmark = pytest.mark.gpu

from parity import TOL, close  # noqa: E402  (tests/parity.py: 1e-4 + the MEASURED fp32 noise of the reference)


def dbl(x):
    """The same fp32 values in float64 (lists / tuples element-wise, index tensors and None as they are):
    the oracle evaluated on them is the `exact` argument of close()."""
    if isinstance(x, (list, tuple)):
        return type(x)(dbl(v) for v in x)
    return x.double() if torch.is_tensor(x) and x.is_floating_point() else x


def opt_ns(**over):
    d = dict(latent_shape=[2, 4], obj_shape=[2, 2], time_dropout=False, num_obj=3, patch_size=4,
             scale_factor=1, dim=16, aspect_ratio=2, load_dim=32, num_perm_grid=1,
             normalize_alpha=False, use_lyt_filtering=False, use_lyt_opacity=False,
             weight_cls=False, min_cls=0.0, include_self=False, no_filter=False, allow_ghost=False)
    d.update(over)
    return types.SimpleNamespace(**d)


# ----------------------------------------------------------------------------- A3
@pytest.mark.parametrize("tag", ["obj", "bg", "obj2"])
def test_inverse_warp_golden(dev, golden, tag):
    import waldo_amd
    g = golden(f"inverse_warp_{tag}")
    mod = waldo_amd.InverseWarp(int(g["hs"]), int(g["ws"]), int(g["ht"]), int(g["wt"])).to(dev)
    sg = g["src_grid"].to(dev).requires_grad_()
    out = mod(sg, erode=bool(g["erode"]))
    close(out, g["out"], what="out")
    (out * g["wgt"].to(dev)).sum().backward()
    close(sg.grad, g["grad_src_grid"], rel=True, what="grad_src_grid")


@pytest.mark.parametrize("tag", ["k5", "k7"])
def test_inverse_warp_kernel_size_golden(dev, golden, tag):
    """``InverseWarp(kernel_size=5 / 7)`` (warp.py:58-63, 140-146; no script passes one) against the reference's own
    outputs and gradients: the fill passes run one by one with the K x K window."""
    import waldo_amd
    g = golden(f"inverse_warp_{tag}")
    mod = waldo_amd.InverseWarp(int(g["hs"]), int(g["ws"]), int(g["ht"]), int(g["wt"]),
                                kernel_size=int(g["kernel_size"])).to(dev)
    sg = g["src_grid"].to(dev).requires_grad_()
    out = mod(sg, niter=int(g["niter"]), erode=bool(g["erode"]))
    close(out, g["out"], what="out")
    (out * g["wgt"].to(dev)).sum().backward()
    close(sg.grad, g["grad_src_grid"], rel=True, what="grad_src_grid")


@pytest.mark.parametrize("ks,cfg", [(5, (8, 8, 40, 70, 6, True)), (7, (16, 32, 16, 32, 3, False)), (9, (6, 6, 70, 40, 2, True)),
                                    (1, (8, 8, 8, 8, 0, False)), (5, (64, 64, 128, 256, 5, True))])
def test_inverse_warp_kernel_size_random(dev, ks, cfg):
    """Other odd kernel sizes and rasters against the oracle, forward and backward (kernel_size 1 with no fill pass: the
    1 x 1 window would divide by the empty neighbourhood's zero weight, in the reference too)."""
    import waldo_amd
    hs, ws, ht, wt, niter, erode = cfg
    torch.manual_seed(ks * 100 + wt)
    ctrl = O.get_grid(4, 4).view(-1, 2)
    inv, rep = O.tps_init(hs, ws, ctrl)
    pts = ctrl.view(1, 16, 2) * 0.6 + 0.12 * torch.randn(3, 16, 2)
    sg = O.tps_grid(inv, rep, pts, hs, ws).requires_grad_()
    ref = O.inverse_warp(sg, (ht, wt), niter=niter, erode=erode, kernel_size=ks)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    mod = waldo_amd.InverseWarp(hs, ws, ht, wt, kernel_size=ks).to(dev)
    s2 = sg.detach().to(dev).requires_grad_()
    out = mod(s2, niter=niter, erode=erode)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(s2.grad, sg.grad, rel=True, what="grad")


@pytest.mark.parametrize("tag", ["perm3", "perm4"])
def test_inverse_warp_num_perm_golden(dev, golden, tag):
    """num_perm > 1 (warp.py:91-111) against the reference's own output for its own perm buffer:
    the first sample in perm[p] order wins each contested cell, the P fields are averaged."""
    import waldo_amd
    g = golden(f"inverse_warp_{tag}")
    perm = g["perm"]
    mod = waldo_amd.InverseWarp(int(g["hs"]), int(g["ws"]), int(g["ht"]), int(g["wt"]),
                                num_perm=perm.shape[0]).to(dev)
    mod.perm.copy_(perm)
    sg = g["src_grid"].to(dev).requires_grad_()
    out = mod(sg, erode=bool(g["erode"]))
    close(out, g["out"], what="out")
    (out * g["wgt"].to(dev)).sum().backward()
    close(sg.grad, g["grad_src_grid"], rel=True, what="grad_src_grid")
    # the tie-break matters in these cases: sample-index order gives a different field
    one = waldo_amd.InverseWarp(int(g["hs"]), int(g["ws"]), int(g["ht"]), int(g["wt"])).to(dev)
    assert (one(sg.detach(), erode=bool(g["erode"])).cpu() - g["out"]).abs().max() > 1e-3


def test_inverse_warp_num_perm_random(dev):
    """Recipe-sized object map, 3 random orders, against the oracle."""
    import waldo_amd
    torch.manual_seed(77)
    ctrl = O.get_grid(4, 4).view(-1, 2)
    inv, rep = O.tps_init(64, 64, ctrl)
    pts = ctrl.view(1, 16, 2) * 0.55 + 0.12 * torch.randn(2, 16, 2)
    sg = O.tps_grid(inv, rep, pts, 64, 64).requires_grad_()
    mod = waldo_amd.InverseWarp(64, 64, 128, 256, num_perm=3).to(dev)
    ref = O.inverse_warp(sg, (128, 256), perm=mod.perm.cpu())
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    s2 = sg.detach().to(dev).requires_grad_()
    out = mod(s2)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(s2.grad, sg.grad, rel=True, what="grad")


@pytest.mark.parametrize("cfg", [(8, 8, 16, 32, 5, True), (16, 32, 16, 32, 5, False), (64, 64, 128, 256, 5, True),
                                 (7, 9, 13, 21, 3, True), (8, 8, 8, 8, 0, False), (4, 4, 40, 40, 8, True),
                                 (16, 16, 40, 70, 1, True), (16, 32, 40, 70, 2, True), (8, 8, 40, 70, 6, True),
                                 (6, 6, 70, 40, 7, True)])
def test_inverse_warp_random(dev, cfg):
    """Object- and background-shaped maps (incl. the recipe's 64x64 -> 128x256), ragged sizes,
    niter 0 and > 5, strong shrink (most of the target unfilled) -- bit-level cell decisions
    (round-half-even, lowest index wins) make any slip an O(1) error.  niter 1 .. 8 cover the forms the one-launch
    kernel takes its passes in: bit rows for the erosion and the rings (3 .. 7), bit rows for the erosion alone
    (1, 2: the rings' rows do not fit behind the list), bytes (8: a region row is wider than 64 cells)."""
    import waldo_amd
    hs, ws, ht, wt, niter, erode = cfg
    torch.manual_seed(hs * 31 + wt)
    ctrl = O.get_grid(4, 4).view(-1, 2)
    inv, rep = O.tps_init(hs, ws, ctrl)
    pts = ctrl.view(1, 16, 2) * 0.6 + 0.12 * torch.randn(3, 16, 2)
    sg = O.tps_grid(inv, rep, pts, hs, ws).requires_grad_()
    ref = O.inverse_warp(sg, (ht, wt), niter=niter, erode=erode)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    mod = waldo_amd.InverseWarp(hs, ws, ht, wt).to(dev)
    s2 = sg.detach().to(dev).requires_grad_()
    out = mod(s2, niter=niter, erode=erode)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(s2.grad, sg.grad, rel=True, what="grad")
    # the one-launch tile + halo kernels (forward fill / erosion, backward fill passes) against the
    # per-pass kernels they replace: same arithmetic in the same order, bit for bit
    from waldo_amd import _lib
    lib = _lib.load()
    assert lib.waldo_set_debug_option(_lib.DEBUG_IW_PASSES, 1) == 0
    try:
        s3 = sg.detach().to(dev).requires_grad_()
        out3 = mod(s3, niter=niter, erode=erode)
        (out3 * wgt.to(dev)).sum().backward()
    finally:
        lib.waldo_set_debug_option(_lib.DEBUG_IW_PASSES, 0)
    assert torch.equal(out3, out), "fused forward differs from the per-pass kernels"
    assert torch.equal(s3.grad, s2.grad), "fused backward differs from the per-pass kernels"


def test_inverse_warp_seeded_fuzz(dev):
    """Forty more inversions from a seeded generator: source rasters of 3 ... 40, targets of 3 ... 150 pixels a side
    (minifications and magnifications, every remainder against the tile + halo kernels' regions), 0 ... 9 fill passes,
    erosion on and off, shrinking and expanding maps, 1 ... 5 maps per call: forward and backward against the oracle and
    the one-launch kernels against the per-pass kernels bit for bit, as ``test_inverse_warp_random``."""
    import random
    import waldo_amd
    from waldo_amd import _lib
    lib = _lib.load()
    rng = random.Random(77)
    for case in range(40):
        hs, ws = rng.randint(3, 40), rng.randint(3, 40)
        ht, wt = rng.randint(3, 150), rng.randint(3, 150)
        niter, erode, n = rng.randint(0, 9), rng.random() < 0.6, rng.randint(1, 5)
        shrink, jitter = rng.choice([0.3, 0.6, 0.9, 1.0, 1.2]), rng.choice([0.02, 0.1, 0.25])
        what = f"case {case}: {hs}x{ws} -> {ht}x{wt}, niter {niter}, erode {erode}, {n} maps, shrink {shrink}, jitter {jitter}"
        torch.manual_seed(case)
        ctrl = O.get_grid(4, 4).view(-1, 2)
        inv, rep = O.tps_init(hs, ws, ctrl)
        pts = ctrl.view(1, 16, 2) * shrink + jitter * torch.randn(n, 16, 2)
        sg = O.tps_grid(inv, rep, pts, hs, ws).requires_grad_()
        ref = O.inverse_warp(sg, (ht, wt), niter=niter, erode=erode)
        wgt = torch.randn(ref.shape)
        (ref * wgt).sum().backward()
        mod = waldo_amd.InverseWarp(hs, ws, ht, wt).to(dev)
        s2 = sg.detach().to(dev).requires_grad_()
        out = mod(s2, niter=niter, erode=erode)
        (out * wgt.to(dev)).sum().backward()
        assert lib.waldo_set_debug_option(_lib.DEBUG_IW_PASSES, 1) == 0
        try:
            s3 = sg.detach().to(dev).requires_grad_()
            out3 = mod(s3, niter=niter, erode=erode)
            (out3 * wgt.to(dev)).sum().backward()
        finally:
            lib.waldo_set_debug_option(_lib.DEBUG_IW_PASSES, 0)
        try:
            close(out, ref, what="out")
            close(s2.grad, sg.grad, rel=True, what="grad")
            assert torch.equal(out3, out), "fused forward differs from the per-pass kernels"
            assert torch.equal(s3.grad, s2.grad), "fused backward differs from the per-pass kernels"
        except AssertionError as exc:
            raise AssertionError(f"{what}: {exc}") from exc


def test_inverse_warp_identity_and_errors(dev):
    import waldo_amd
    mod = waldo_amd.InverseWarp(12, 20, 12, 20).to(dev)
    ident = O.get_grid(12, 20).to(dev)
    close(mod(ident, erode=False), ident.cpu(), 1e-6, what="identity")
    with pytest.raises(ValueError):
        mod(ident, pad=False)
    with pytest.raises(ValueError):  # an even window changes the raster under conv2d(padding=k // 2): fails in the reference
        waldo_amd.InverseWarp(8, 8, 8, 8, kernel_size=4).to(dev)(O.get_grid(8, 8).to(dev))
    two = waldo_amd.InverseWarp(12, 20, 12, 20, num_perm=2).to(dev)  # no collisions: order is moot
    close(two(ident, erode=False), ident.cpu(), 1e-6, what="identity, num_perm=2")
    assert mod(torch.zeros(0, 12, 20, 2, device=dev)).shape == (0, 12, 20, 2)


# ----------------------------------------------------------------------------- A6
def test_compute_occ_and_reduce_comp_golden(dev, golden):
    from waldo_amd.nets import compute_occ, reduce_comp
    g = golden("occ_comp")
    score = g["score"].to(dev).requires_grad_()
    occ = compute_occ(score)
    close(occ, g["occ"], 1e-6, what="occ")
    vid = g["vid"].to(dev).requires_grad_()
    out, alpha, _ = reduce_comp(vid, occ)
    close(out, g["out"], what="out")
    close(alpha, g["alpha"], what="alpha")
    ((out * g["w1"].to(dev)).sum() + (alpha * g["w2"].to(dev)).sum()).backward()
    close(vid.grad, g["grad_vid"], rel=True, what="grad_vid")
    close(score.grad, g["grad_score"], rel=True, what="grad_score")


# ----------------------------------------------------------------------------- A9 / A10 / A13 / A14
def _warper_inputs(cfg, b, t, nl, seed):
    g = torch.Generator().manual_seed(seed)
    no = cfg.num_obj
    lo = cfg.obj_shape[0] * cfg.obj_shape[1]
    lb = cfg.latent_shape[0] * cfg.latent_shape[1]
    obj_pose = O.get_grid(*cfg.obj_shape).view(1, 1, 1, lo, 2) * 0.5 + 0.15 * torch.randn(b, t, no, lo, 2, generator=g)
    bg_pose = O.get_grid(*cfg.latent_shape).view(1, 1, 1, lb, 2) + 0.05 * torch.randn(b, t, 1, lb, 2, generator=g)
    (hd, wd), (h, w), (ho, wo) = cfg.src_shape_hd, cfg.src_shape, cfg.tgt_shape
    # smooth "frames": the chain samples them at flow-displaced positions
    lo_res = torch.randn(b * t, 3 + nl, hd // 4, wd // 4, generator=g)
    inp = torch.nn.functional.interpolate(lo_res, size=(hd, wd), mode="bilinear").view(b, t, 3 + nl, hd, wd)
    occ = O.compute_occ(torch.randn(b, t, no, generator=g))
    obj_alpha = torch.rand(b, no, 1, ho, wo, generator=g) * 2 - 1
    bg_alpha = torch.ones(b, 1, h, w)
    cls = torch.rand(b, no, nl, generator=g).softmax(-1)
    return obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls


def test_warper_chain_golden(dev, golden):
    """Warper.forward -> grid_to_flow[_ctx] -> input_to_output vs the reference's own outputs."""
    from waldo_amd.nets import Warper
    g = golden("warper_chain")
    wp = Warper(opt_ns(num_obj=2, weight_cls=True, min_cls=0.05)).to(dev)
    d = {k: v.to(dev) for k, v in g.items()}
    grid = wp(d["obj_pose"], d["bg_pose"])
    for a, name in zip(grid, ("tgo", "sgo", "tgb", "sgb")):
        close(a, g[name], what=name)
    args = (d["inp"], grid, d["occ"], d["obj_alpha"], d["bg_alpha"], d["cls"], d["ctx_ts"], d["pred_ts"])
    for pre, r in (("c_", wp.grid_to_flow_ctx(*args)), ("t_", wp.grid_to_flow(*args))):
        close(r[0], g[pre + "flow"], what=pre + "flow")
        assert r[1] is None  # load_dim > 0: alpha_unflt withheld, as in the reference
        close(r[2], g[pre + "alpha"], what=pre + "alpha")
        close(r[3], g[pre + "alpha_ctx"], what=pre + "alpha_ctx")
        close(r[4], g[pre + "disocc"], what=pre + "disocc")
    out, raw = wp.input_to_output(d["inp"], d["c_alpha_ctx"], d["c_flow"], d["ctx_ts"])
    # frames sampled at flow-displaced positions: the allowance on top of 1e-4 is the distance of the
    # reference's own fp32 result from the float64 evaluation of the same step on the same inputs
    cfg = WO.WarperCfg.from_opt(opt_ns(num_obj=2, weight_cls=True, min_cls=0.05))
    out64, raw64 = WO.input_to_output(cfg, g["inp"].double(), g["c_alpha_ctx"].double(), g["c_flow"].double(), g["ctx_ts"])
    close(out, g["out"], what="out", exact=out64)
    close(raw, g["raw"], what="raw", exact=raw64)


@pytest.mark.parametrize("over", [dict(), dict(weight_cls=True, min_cls=0.1), dict(load_dim=0),
                                  dict(allow_ghost=True), dict(no_filter=True),
                                  dict(num_obj=7, obj_shape=[4, 4], latent_shape=[4, 8], dim=32, load_dim=64)])
def test_warper_against_oracle(dev, over):
    """Every Warper method, fwd and bwd, on seeded inputs; the last case is recipe-shaped
    (16 object / 32 background control points, 8 layers, x2 high-res)."""
    from waldo_amd.nets import Warper
    opt = opt_ns(**over)
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    b, t, nl = 2, 4, 5
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=3)
    ctx_ts = torch.tensor([[[0, 1], [1, 0]], [[1, 1], [0, 0]]])
    pred_ts = torch.tensor([2, 3])
    # (i) the four grids.  The TPS grids agree to fp32 rounding; the INVERTED grids are a
    # discontinuous function of them (round-to-cell, winner election), so they are compared on
    # IDENTICAL inputs: the HIP inversion of the oracle's TPS grids vs the oracle's inversion.
    with torch.no_grad():
        grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
        grid_h = wp(obj_pose.to(dev), bg_pose.to(dev))
        close(grid_h[0], grid_o[0], what="tgo")
        close(grid_h[2], grid_o[2], what="tgb")
        no, (h, w), (ho, wo) = cfg.num_obj, cfg.src_shape, cfg.tgt_shape
        close(wp.invert_obj(grid_o[0].reshape(-1, ho, wo, 2).to(dev)).view(b, t, no, h, w, 2), grid_o[1], 1e-6, what="sgo")
        close(wp.invert_bg(grid_o[2].reshape(-1, h, w, 2).to(dev), erode=False).view(b, t, h, w, 2), grid_o[3], 1e-6, what="sgb")
    # (ii) the chain downstream of the grids, fwd and bwd, on the oracle's grids
    leaves = [x.clone().requires_grad_() for x in (*grid_o, obj_alpha)]
    ro = WO.grid_to_flow_ctx(cfg, inp, leaves[:4], occ, leaves[4], bg_alpha, cls, ctx_ts, pred_ts)
    out_o, raw_o = WO.input_to_output(cfg, inp, ro[3], ro[0], ctx_ts)
    torch.manual_seed(1)
    wgt = torch.randn(out_o.shape)
    (out_o * wgt).sum().backward()
    # the same chain in float64 from the same fp32 inputs: the `exact` side of every comparison below
    l64 = [x.detach().double().requires_grad_() for x in (*grid_o, obj_alpha)]
    r64 = WO.grid_to_flow_ctx(cfg, inp.double(), l64[:4], occ.double(), l64[4], bg_alpha.double(), cls.double(), ctx_ts, pred_ts)
    out64, raw64 = WO.input_to_output(cfg, inp.double(), r64[3], r64[0], ctx_ts)
    (out64 * wgt.double()).sum().backward()
    dl = [x.clone().to(dev).requires_grad_() for x in (*grid_o, obj_alpha)]
    grid_h = dl[:4]
    args = (inp.to(dev), grid_h, occ.to(dev), dl[4], bg_alpha.to(dev), cls.to(dev), ctx_ts.to(dev), pred_ts.to(dev))
    rh = wp.grid_to_flow_ctx(*args)
    for x, y, z, name in zip(rh, ro, r64, ("flow", "alpha_unflt", "alpha", "alpha_ctx", "disocc")):
        close(x, y, what="ctx:" + name, exact=z)
    out_h, raw_h = wp.input_to_output(inp.to(dev), rh[3], rh[0], ctx_ts.to(dev))
    # frames sampled at positions that are themselves the fp32 result of the chain above: the allowance
    # is what that costs the fp32 ORACLE against its float64 self, measured
    close(out_h, out_o, what="out", exact=out64)
    close(raw_h, raw_o, what="raw", exact=raw64)
    (out_h * wgt.to(dev)).sum().backward()
    for x, y, z, name in zip(dl, leaves, l64, ("grad tgo", "grad sgo", "grad tgb", "grad sgb", "grad obj_alpha")):
        close(x.grad, y.grad, rel=True, what=name, exact=z.grad)
    grid_o = [x.detach() for x in leaves[:4]]
    grid_h = [x.detach() for x in dl[:4]]
    # training variant and the helpers
    rt_o = WO.grid_to_flow(cfg, inp, grid_o, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
    rt_h = wp.grid_to_flow(inp.to(dev), grid_h, occ.to(dev), obj_alpha.to(dev), bg_alpha.to(dev),
                           cls.to(dev), ctx_ts.to(dev), pred_ts.to(dev))
    for x, y, name in zip(rt_h, rt_o, ("flow", "alpha_unflt", "alpha", "alpha_ctx", "disocc")):
        close(x, y, what="train:" + name)
    gd = [x.detach() for x in grid_h]
    go = [x.detach() for x in grid_o]
    close(wp.grid_to_bg_flow_from_ref_to_pred(gd, 2, 1), WO.grid_to_bg_flow_from_ref_to_pred(cfg, go, 2, 1), what="A14a")
    close(wp.grid_to_obj_flow_from_ref_to_pred(gd, 2, 1, 2), WO.grid_to_obj_flow_from_ref_to_pred(cfg, go, 2, 1, 2), what="A14b")
    close(wp.grid_to_bg_flow_from_ctx_to_ref(gd, 2, 3), WO.grid_to_bg_flow_from_ctx_to_ref(cfg, go, 2, 3), what="A14c")
    h, w = cfg.src_shape
    x5 = torch.randn(b, t, 4, h, w)
    for x, y in zip(wp.layer_from_input(x5.to(dev), gd), WO.layer_from_input(cfg, x5, go)):
        close(x, y, what="layer_from_input")
    for x, y in zip(wp.alpha_to_alpha(obj_alpha.to(dev), bg_alpha.to(dev), gd, occ.to(dev)),
                    WO.alpha_to_alpha(cfg, obj_alpha, bg_alpha, go, occ)):
        close(x, y, what="alpha_to_alpha")


@pytest.mark.parametrize("over", [dict(num_obj=16, obj_shape=[2, 2], dim=8, load_dim=32),
                                  dict(num_obj=11, dim=16, load_dim=48, allow_ghost=True),
                                  dict(num_obj=3, dim=16, load_dim=16, weight_cls=True, min_cls=0.05)])
@pytest.mark.parametrize("ctx_only", [True, False])
def test_flow_ctx_fused_passes(dev, over, ctx_only):
    """The fused full-resolution passes of grid_to_flow[_ctx] (csrc/flow_ctx.hip)
    against the CPU oracle and against the unfused per-op path: the recipe's L = 17 layers and
    Nl = 20 classes at x4, a non-power-of-two x3, and x1 (load_dim == dim)."""
    from waldo_amd.nets import Warper
    opt = opt_ns(**over)
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    b, t, nl = 2, 3, 20
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=11)
    ctx_ts = torch.tensor([[[0], [1]], [[1], [1]]])
    pred_ts = torch.tensor([2])
    with torch.no_grad():
        grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
        fn_o = WO.grid_to_flow_ctx if ctx_only else WO.grid_to_flow
        ro = fn_o(cfg, inp, grid_o, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
        args = (inp.to(dev), [x.to(dev) for x in grid_o], occ.to(dev), obj_alpha.to(dev), bg_alpha.to(dev),
                cls.to(dev), ctx_ts.to(dev), pred_ts.to(dev))
        fn_h = wp.grid_to_flow_ctx if ctx_only else wp.grid_to_flow
        assert wp.fuse_hd and wp._fused_ok(list(args[:1]), cfg.num_obj + 1, nl)
        rf = fn_h(*args)
        assert wp.alpha_ctx_max is None
        # the by-product Synthesizer.predict's disocclusion test asks for: max over the layers of alpha_ctx
        wp.keep_alpha_ctx_max = True
        rf2 = fn_h(*args)
        assert all(torch.equal(x, y) for x, y in zip(rf, rf2) if x is not None)
        assert torch.equal(wp.alpha_ctx_max, rf[3].amax(dim=3))
        close(wp.alpha_ctx_max, ro[3].amax(dim=3), what="alpha_ctx_max vs oracle")
        wp.fuse_hd = False
        ru = fn_h(*args)
        assert wp.alpha_ctx_max is None
    for x, y, z, name in zip(rf, ro, ru, ("flow", "alpha_unflt", "alpha", "alpha_ctx", "disocc")):
        close(x, y, what="fused vs oracle:" + name)
        close(x, z, what="fused vs unfused:" + name)


def test_fused_hd_passes_seeded_fuzz(dev):
    """Thirty recipes drawn from a seeded generator -- 1 ... 16 objects, 2 ... 32 layout classes, 8 ... 24-pixel layer
    rasters upsampled x1 ... x5 (tile remainders of every kind), 1 ... 3 clips, 1 ... 3 context and 1 ... 3 predicted
    frames with random (repeating) context indices, the option sets that change the kernels' paths (ghost mask, class
    weighting, no filter) -- through ``grid_to_flow_ctx`` / ``grid_to_flow`` and ``input_to_output``
    without autograd (what ``predict`` and ``inpaint`` run: composited alphas written straight into the raw slots,
    occupancy map, staged frame warp) against the oracle, fp32 and fp64, through ``close``."""
    import random
    from waldo_amd.nets import Warper
    rng = random.Random(6)
    fused = 0
    for case in range(30):
        dim = rng.choice([8, 12, 16, 20, 24])
        s = rng.choice([1, 2, 2, 3, 4, 4, 5])
        over = dict(num_obj=rng.randint(1, 16), dim=dim, load_dim=dim * s, obj_shape=rng.choice([[2, 2], [2, 2], [3, 3]]),
                    allow_ghost=rng.random() < 0.3, weight_cls=rng.random() < 0.4, no_filter=rng.random() < 0.2)
        over["min_cls"] = 0.05 if over["weight_cls"] else 0.0
        opt = opt_ns(**over)
        cfg = WO.WarperCfg.from_opt(opt)
        wp = Warper(opt).to(dev)
        b, tc, tp = rng.randint(1, 3), rng.randint(1, 3), rng.randint(1, 3)
        t, nl = tc + tp, rng.choice([2, 5, 13, 20, 21, 32])
        ctx_only = rng.random() < 0.6
        obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=100 + case)
        g = torch.Generator().manual_seed(case)
        ctx_ts = torch.randint(0, tc, (b, tc, tp), generator=g)
        pred_ts = torch.arange(tc, t)
        what = f"case {case}: {over} b={b} tc={tc} tp={tp} nl={nl} ctx_only={ctx_only}"
        with torch.no_grad():
            grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
            fn_o = WO.grid_to_flow_ctx if ctx_only else WO.grid_to_flow
            ro = fn_o(cfg, inp, grid_o, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
            r64 = fn_o(cfg, *dbl((inp, grid_o, occ, obj_alpha, bg_alpha, cls)), ctx_ts, pred_ts)
            out_o, raw_o = WO.input_to_output(cfg, inp, ro[3], ro[0], ctx_ts)
            out64, raw64 = WO.input_to_output(cfg, inp.double(), r64[3], r64[0], ctx_ts)
            args = (inp.to(dev), [x.to(dev) for x in grid_o], occ.to(dev), obj_alpha.to(dev), bg_alpha.to(dev),
                    cls.to(dev), ctx_ts.to(dev), pred_ts.to(dev))
            fused += bool(wp.fuse_hd and wp._fused_ok(list(args[:1]), cfg.num_obj + 1, nl))
            rh = (wp.grid_to_flow_ctx if ctx_only else wp.grid_to_flow)(*args)
            out_h, raw_h = wp.input_to_output(inp.to(dev), rh[3], rh[0], ctx_ts.to(dev))
        try:
            for x, y, z, name in zip(rh, ro, r64, ("flow", "alpha_unflt", "alpha", "alpha_ctx", "disocc")):
                if y is None:
                    assert x is None, name
                    continue
                close(x, y, what=name, exact=z)
            close(out_h, out_o, what="out", exact=out64)
            close(raw_h, raw_o, what="raw", exact=raw64)
        except AssertionError as exc:
            raise AssertionError(f"{what}: {exc}") from exc
    assert fused == 30, fused


def _flow_ctx_warp_spelled_out(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, s):
    """lvd.py:784-818 with framework ops, on whatever device the inputs are: the statement the kernel's short cuts
    (absent layers skipped per wavefront) must not be distinguishable from, NaNs included."""
    import torch.nn.functional as F
    m, nl, _, h, w = flow_lr.shape
    b, tc, tp = ctx_ts.shape
    hd, wd = h * s, w * s
    up = (lambda x: F.interpolate(x, scale_factor=s, mode="bilinear")) if s > 1 else (lambda x: x)
    flow = up(flow_lr.reshape(m * nl, 2, h, w)).view(m, nl, 2, hd, wd)
    xs = O.get_grid(hd, wd).to(flow_lr.device)                                          # 1 Hd Wd 2
    samp = xs.unsqueeze(1) + flow.permute(0, 1, 3, 4, 2)                               # M L Hd Wd 2
    src = a01.view(b, tw, nl, hd, wd)[torch.arange(b, device=a01.device).view(b, 1, 1), ctx_ts]   # B Tc Tp L Hd Wd
    val = F.grid_sample(src.reshape(m * nl, 1, hd, wd), samp.reshape(m * nl, hd, wd, 2), align_corners=False)
    val = val.view(m, nl, hd, wd)
    if isobj is not None:
        keep = (up(isobj) > 0.9).to(val.dtype)
        val = torch.cat([val[:, :1], torch.where(keep > 0, val[:, 1:], torch.zeros_like(val[:, 1:]))], dim=1)
    dis = val.max(dim=1)[0]
    oc = occ[:, pred_ts].unsqueeze(1).expand(b, tc, tp, nl, nl).reshape(m, nl, nl)       # [i][j]
    fac = 1 - val.unsqueeze(2) * oc.view(m, nl, nl, 1, 1)                               # M i j Hd Wd
    prod = torch.ones_like(val)
    for i in range(nl):                                                                 # the kernel's order over i
        prod = prod * fac[:, i]
    actx = val * prod
    out_flow = (actx.unsqueeze(2) * flow).sum(dim=1)
    return out_flow, actx * 2 - 1, dis


@pytest.mark.parametrize("poison,hw", [("none", (32, 64)), ("occ", (32, 64)), ("flow", (32, 64)), ("alpha", (32, 64)),
                                       ("soft", (32, 64)), ("mask", (32, 64)), ("soft", (9, 20)),
                                       ("none", (9, 20)), ("alpha", (9, 20))])  # 36 x 80: ragged 16 x 64 tiles
def test_flow_ctx_warp_skips_absent_layers_exactly(dev, poison, hw):
    """The wavefront-level short cuts of flow_ctx_warp_kernel (a layer whose object mask is off in all 64 lanes is
    not sampled; layers with alpha 0 in all lanes leave the occlusion product): on a scene of SMALL objects --
    most layers absent from most wavefronts -- the results equal the spelled-out expression, and a NaN / inf in the
    order, in the low-resolution flows or in the context alphas lands exactly where the expression puts it."""
    from waldo_amd import functional as WF
    b, t, tc, tp, nl, s, tw = 1, 3, 2, 2, 12, 4, 2
    h, w = hw
    hd, wd = h * s, w * s
    g = torch.Generator(device=dev).manual_seed(21)
    m = b * tc * tp
    flow_lr = 0.05 * torch.randn(m, nl, 2, h, w, generator=g, device=dev)
    # every object: a disc of a few cells somewhere in the frame; mask 1 inside, 0 outside
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    cy = torch.rand(m, nl - 1, 1, 1, generator=g, device=dev) * h
    cx = torch.rand(m, nl - 1, 1, 1, generator=g, device=dev) * w
    isobj = (((yy - cy) ** 2 + (xx - cx) ** 2) < (36 if h >= 32 else 9)).float()
    if poison == "soft":
        # masks that fall off over a few cells: cells between the coarse row test's 0.5 and the 0.9 of lvd.py:785,
        # upsampled values on both sides of 0.9 inside one wavefront
        isobj = (1.3 - ((yy - cy) ** 2 + (xx - cx) ** 2).sqrt() / (6.0 if h >= 32 else 3.0)).clamp(0.0, 1.0)
    elif poison == "mask":
        isobj[1, 3, h // 2, :] = float("nan")       # NaN > 0.9 is false: the layer is dropped there, as if absent
        isobj[2, 7, 2:5, 3:9] = float("nan")
    a01 = torch.rand(b * tw, nl, hd, wd, generator=g, device=dev)
    a01[:, 1:] *= (torch.rand(b * tw, nl - 1, hd, wd, generator=g, device=dev) > 0.5)   # exact zeros inside objects too
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    ctx_ts = torch.randint(0, tw, (b, tc, tp), generator=g, device=dev)
    pred_ts = torch.tensor([2, 0], device=dev)
    if poison == "occ":
        occ[0, 2, 3, 5] = float("nan")      # frame 2 is predicted frame 0: a NaN factor in column 5 of every pixel
        occ[0, 0, 7, 1] = float("inf")
    elif poison == "flow":
        flow_lr[1, 4, 0, h // 3, w // 3] = float("nan")    # an object that is absent there: 0 * NaN all the same
        flow_lr[2, 0, 1, 5, 5] = float("inf")
    elif poison == "alpha":
        # (NaN only: an INFINITE texel under a zero tap weight is NaN in the four-weight form of F.grid_sample and
        # may be finite in the lerp form of the kernels -- a difference of the sampling form, not of the skipping)
        a01[0, 6, hd // 3:hd // 2, wd // 3:wd // 2] = float("nan")
        a01[1, 0, 5, 7] = float("nan")
    with torch.no_grad():
        flow, actx, dis, amax = WF.flow_ctx_warp(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, s, layer_max=True)
        rflow, ractx, rdis = _flow_ctx_warp_spelled_out(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, s)
    for x, y, name in ((flow, rflow, "flow"), (actx, ractx, "alpha_ctx"), (dis, rdis, "disocc"), (amax, ractx.amax(dim=1), "max")):
        assert torch.equal(torch.isnan(x), torch.isnan(y)), f"{poison}: NaNs of {name} differ"
        ok = ~torch.isnan(y)
        # +-inf must agree exactly; finite values to the chain's tolerance
        assert torch.equal(torch.isinf(x[ok]), torch.isinf(y[ok])), f"{poison}: infinities of {name} differ"
        fin = ok & ~torch.isinf(y)
        close(x[fin], y[fin], what=f"{poison}: {name}")
    if poison in ("none", "soft", "mask"):
        # absent layers come out as the exact constants
        gone = (actx == -1.0).float().mean().item()
        assert gone > 0.4, gone


@pytest.mark.parametrize("nl", [2, 4, 8, 12, 13, 17, 24])
@pytest.mark.parametrize("scale,hw", [(1, (20, 70)), (2, (19, 40)), (3, (11, 30)), (4, (9, 20)), (5, (7, 15)), (8, (5, 9))])
def test_flow_ctx_warp_tile_shapes_agree_with_the_launcher(dev, nl, scale, hw):
    """flow_ctx_warp over the scales and layer counts that make its launcher pick different tiles (16 x 64, 8 x 64 or
    4 x 64 pixels per workgroup, staged or not: `fits` in flow_ctx_warp_launch against the kernels' LDS images, which
    the tall-tile instances take on trust) on rasters with ragged tiles: every output against the spelled-out
    expression."""
    from waldo_amd import functional as WF
    h, w = hw
    b, t, tc, tp, tw = 1, 3, 2, 2, 2
    hd, wd = h * scale, w * scale
    g = torch.Generator(device=dev).manual_seed(100 * nl + scale)
    m = b * tc * tp
    flow_lr = 0.05 * torch.randn(m, nl, 2, h, w, generator=g, device=dev)
    isobj = (torch.rand(m, nl - 1, h, w, generator=g, device=dev) > 0.6).float() if nl > 1 else None
    a01 = torch.rand(b * tw, nl, hd, wd, generator=g, device=dev)
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    ctx_ts = torch.randint(0, tw, (b, tc, tp), generator=g, device=dev)
    pred_ts = torch.tensor([2, 0], device=dev)
    with torch.no_grad():
        flow, actx, dis, amax = WF.flow_ctx_warp(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, scale, layer_max=True)
        rflow, ractx, rdis = _flow_ctx_warp_spelled_out(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, scale)
    close(flow, rflow, what="flow")
    close(actx, ractx, what="alpha_ctx")
    close(dis, rdis, what="disocc")
    close(amax, ractx.amax(dim=1), what="max")


@pytest.mark.parametrize("cfg", [(17, 4, 32, 64, 0.02), (17, 4, 32, 64, 0.5), (12, 2, 40, 72, 0.1), (5, 4, 16, 48, 2.5),
                                 (8, 3, 24, 40, 0.05)])
@pytest.mark.parametrize("poison", ["none", "alpha"])
def test_layer_occupancy_map_skips_absent_layers_exactly(dev, cfg, poison):
    """The path WITHOUT a ghost mask (Warper.grid_to_flow, lvd.py:602-705: what train_wif.sh runs): the alpha pass leaves
    `layer_bits` -- per (frame, row, 64-pixel segment) the layers that are non-zero there -- and the flow pass skips, per
    tile, the layers that are absent from every segment its samples can reach (round 6).  (i) the map is what it says;
    (ii) every output of the flow pass has the same bits with and without it, for small, large and off-frame flows
    (a box of several hundred words, a box beyond the frame) and with a NaN in the source alphas (NaN counts as present)."""
    from waldo_amd import _lib, functional as WF
    nl, s, h, w, amp = cfg
    b, t, tw, tc, tp, ncls = 2, 3, 3, 2, 2, 6
    hd, wd = h * s, w * s
    g = torch.Generator(device=dev).manual_seed(nl * 7 + s)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    cy = torch.rand(b * tw, nl, 1, 1, generator=g, device=dev) * h
    cx = torch.rand(b * tw, nl, 1, 1, generator=g, device=dev) * w
    alpha_lr = torch.rand(b * tw, nl, h, w, generator=g, device=dev) * (((yy - cy) ** 2 + (xx - cx) ** 2) < 9).float()
    alpha_lr[:, 0] = 1.0
    alpha_lr[0, 0, : h // 2] = 0.0                       # (the background absent from half a frame as well)
    inp = torch.randn(b, t, 3 + ncls, hd, wd, generator=g, device=dev)
    dist = torch.rand(b, nl - 1, ncls, generator=g, device=dev).softmax(dim=2)
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    if poison == "alpha":
        alpha_lr[1, 3, 2, 5] = float("nan")
    with torch.no_grad():
        a01, _, bits = WF.flow_ctx_alpha(alpha_lr, inp, dist, occ, tw, 3, s, want_alpha=False, want_bits=True)
        a01_b, alpha_b = WF.flow_ctx_alpha(alpha_lr, inp, dist, occ, tw, 3, s)
    assert torch.equal(torch.nan_to_num(a01, nan=7.0), torch.nan_to_num(a01_b, nan=7.0))
    nseg = (wd + 63) // 64
    assert bits.shape == (b * tw, hd, nseg) and bits.dtype == torch.int32
    pad = nseg * 64 - wd
    nzp = torch.nn.functional.pad((a01 != 0).to(torch.int64), (0, pad)).view(b * tw, nl, hd, nseg, 64).amax(dim=-1)
    want = (nzp << torch.arange(nl, device=dev).view(1, nl, 1, 1)).sum(dim=1)
    assert torch.equal(bits.to(torch.int64), want)
    m = b * tc * tp
    fl = amp * torch.randn(m, nl, 2, max(h // 8, 1), max(w // 8, 1), generator=g, device=dev)
    flow_lr = torch.nn.functional.interpolate(fl.view(m * nl, 2, *fl.shape[-2:]), size=(h, w), mode="bilinear").view(m, nl, 2, h, w)
    ctx_ts = torch.randint(0, tw, (b, tc, tp), generator=g, device=dev)
    pred_ts = torch.randint(0, t, (tp,), generator=g, device=dev)
    with torch.no_grad():
        plain = WF.flow_ctx_warp(flow_lr, None, a01, ctx_ts, pred_ts, occ, tw, s, layer_max=True)
        skipping = WF.flow_ctx_warp(flow_lr, None, a01, ctx_ts, pred_ts, occ, tw, s, layer_max=True, layer_bits=bits)
        raw_plain = WF.flow_ctx_warp_into_raw(flow_lr, None, a01, ctx_ts, pred_ts, occ, tw, s, 5, False, layer_max=True)
        raw_skip = WF.flow_ctx_warp_into_raw(flow_lr, None, a01, ctx_ts, pred_ts, occ, tw, s, 5, False, layer_max=True,
                                             layer_bits=bits)
    for x, y, name in list(zip(plain, skipping, ("flow", "alpha_ctx", "disocc", "alpha max"))) + \
            list(zip(raw_plain[:4], raw_skip[:4], ("raw: flow", "raw: alpha_ctx", "raw: disocc", "raw: alpha max"))):
        assert torch.equal(torch.isnan(x), torch.isnan(y)), f"{name}: NaNs differ"
        assert torch.equal(torch.nan_to_num(x, nan=7.0), torch.nan_to_num(y, nan=7.0)), name
    assert torch.equal(raw_plain[4].score.isnan(), raw_skip[4].score.isnan())
    assert torch.equal(torch.nan_to_num(raw_plain[4].score, nan=7.0), torch.nan_to_num(raw_skip[4].score, nan=7.0))
    if poison == "none" and amp < 1.0:
        assert (plain[1] == -1).float().mean() > 0.5    # most layers ARE absent from most pixels: something to skip
    with pytest.raises(_lib.WaldoHipError, match="layer_bits"):
        WF.flow_ctx_warp(flow_lr, None, a01, ctx_ts, pred_ts, occ, tw, s, layer_bits=bits[:, :-1].contiguous())


@pytest.mark.parametrize("poison,ncls", [("none", 20), ("occ", 20), ("dist", 20), ("logits", 20), ("alpha", 20),
                                         ("none", 21), ("logits", 32), ("none", 5)])
def test_flow_ctx_alpha_skips_absent_layers_exactly(dev, poison, ncls):
    """The same short cuts in flow_ctx_alpha_kernel (layers whose upsampled alpha is 0 in all 64 lanes skip their
    filter weight and leave the occlusion product) against the spelled-out expression of lvd.py:731-766, with most
    objects absent from most wavefronts, and with a NaN / inf in the order, the class distributions, the layout
    logits or the rough alphas.  20 classes and fewer run the kFewCls instances of the kernel, 21 .. 32 the kMaxCls ones."""
    import torch.nn.functional as F
    from waldo_amd import functional as WF
    b, t, tw, nl, h, w, s = 2, 3, 2, 12, 16, 32, 4
    hd, wd = h * s, w * s
    g = torch.Generator(device=dev).manual_seed(33)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    cy = torch.rand(b * tw, nl, 1, 1, generator=g, device=dev) * h
    cx = torch.rand(b * tw, nl, 1, 1, generator=g, device=dev) * w
    alpha_lr = torch.rand(b * tw, nl, h, w, generator=g, device=dev) * (((yy - cy) ** 2 + (xx - cx) ** 2) < 16).float()
    alpha_lr[:, 0] = 1.0                                                    # the background is everywhere
    inp = torch.randn(b, t, 3 + ncls, hd, wd, generator=g, device=dev) * 2
    dist = torch.rand(b, nl - 1, ncls, generator=g, device=dev).softmax(dim=2)
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    if poison == "occ":
        occ[0, 1, 4, 2] = float("nan")
        occ[1, 0, 0, 7] = float("inf")
    elif poison == "dist":
        dist[0, 5, 3] = float("nan")
    elif poison == "logits":
        inp[1, 1, 3 + 4, 10:20, 30:50] = float("nan")
        inp[0, 0, 3 + 1, 5, 5] = float("inf")
    elif poison == "alpha":
        alpha_lr[1, 6, 3, 4] = float("nan")
    with torch.no_grad():
        a01, alpha = WF.flow_ctx_alpha(alpha_lr, inp, dist, occ, tw, 3, s)
        a = F.interpolate(alpha_lr, scale_factor=s, mode="bilinear")                        # B*Tw L Hd Wd
        prob = inp[:, :tw, 3:].softmax(dim=2).reshape(b * tw, 1, ncls, hd, wd)
        d = dist.view(b, 1, nl - 1, ncls, 1, 1).expand(-1, tw, -1, -1, -1, -1).reshape(b * tw, nl - 1, ncls, 1, 1)
        wgt = 1 - (d - prob).abs().sum(dim=2) / 2
        a = torch.cat([a[:, :1], a[:, 1:] * wgt], dim=1)
        oc = occ[:, :tw].reshape(b * tw, nl, nl)
        prod = torch.ones_like(a)
        for i in range(nl):
            prod = prod * (1 - a[:, i:i + 1] * oc[:, i].view(b * tw, nl, 1, 1))
        ref = a * prod
    for x, y, name in ((a01, ref, "a01"), (alpha, ref * 2 - 1, "alpha")):
        assert torch.equal(torch.isnan(x), torch.isnan(y)), f"{poison}: NaNs of {name} differ"
        fin = torch.isfinite(y)
        assert torch.equal(torch.isinf(x[~torch.isnan(y)]), torch.isinf(y[~torch.isnan(y)])), f"{poison}: infinities of {name}"
        close(x[fin], y[fin], what=f"{poison}: {name}")
    if poison == "none":
        assert (a01 == 0).float().mean().item() > 0.5


@pytest.mark.parametrize("nl,s,ncls", [(5, 2, 20), (12, 1, 20), (17, 1, 20), (17, 1, 21), (8, 2, 32)])
def test_flow_ctx_alpha_backward_takes_both_output_gradients(dev, nl, s, ncls):
    """waldo_flow_ctx_alpha_bwd reads d loss / d a01 and d loss / d alpha_out (= 2 a01 - 1) and sums them itself:
    a loss on both outputs gives the gradients of the same loss on `a01` alone with the two gradients added first
    (g_a01 + 2 g_alpha_out, what the wrapper used to do in two passes) -- grad_alpha_lr bit for bit (it is written,
    not accumulated), grad_dist / grad_occ up to the order of their float atomics; and each output alone.  Class
    counts of 20 (the kFewCls instances) and 21 / 32 (the kMaxCls instances, wave_transpose_reduce<32> included)."""
    from waldo_amd import functional as WF
    b, t, tw, h, w = 2, 3, 2, 16, 32
    hd, wd = h * s, w * s
    g = torch.Generator(device=dev).manual_seed(35 + nl)
    alpha_lr = torch.rand(b * tw, nl, h, w, generator=g, device=dev)
    inp = torch.randn(b, t, 3 + ncls, hd, wd, generator=g, device=dev) * 2
    dist = torch.rand(b, nl - 1, ncls, generator=g, device=dev).softmax(dim=2)
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    w1 = torch.randn(b * tw, nl, hd, wd, generator=g, device=dev)
    w2 = torch.randn(b * tw, nl, hd, wd, generator=g, device=dev)

    def grads(loss):
        leaves = [x.clone().requires_grad_() for x in (alpha_lr, dist, occ)]
        a01, alpha = WF.flow_ctx_alpha(leaves[0], inp, leaves[1], leaves[2], tw, 3, s)
        loss(a01, alpha).backward()
        return [x.grad for x in leaves]

    for name, two, one in (
            ("both", lambda a01, alpha: (a01 * w1).sum() + (alpha * w2).sum(), lambda a01, alpha: (a01 * (w1 + 2.0 * w2)).sum()),
            ("alpha_out alone", lambda a01, alpha: (alpha * w2).sum(), lambda a01, alpha: (a01 * (2.0 * w2)).sum()),
            ("a01 alone", lambda a01, alpha: (a01 * w1).sum(), lambda a01, alpha: (a01 * w1).sum() + 0.0 * alpha.sum())):
        got, want = grads(two), grads(one)
        assert torch.equal(got[0], want[0]), f"{name}: grad_alpha_lr"
        close(got[1], want[1], 1e-5, rel=True, what=f"{name}: grad_dist (float atomics)")
        close(got[2], want[2], 1e-5, rel=True, what=f"{name}: grad_occ (float atomics)")


@pytest.mark.parametrize("over,ctx_only,include_self", [
    (dict(num_obj=3, dim=16, load_dim=0), False, True),                       # the LVD recipe's shape: x1, ctx "prev"
    (dict(num_obj=16, obj_shape=[2, 2], dim=8, load_dim=32), True, False),   # L = 17, x4, ghost mask
    (dict(num_obj=5, dim=16, load_dim=48, allow_ghost=True), True, False),   # x3
    (dict(num_obj=3, dim=16, load_dim=32, no_filter=True), False, False),    # no layout filter
    (dict(num_obj=4, dim=16, load_dim=16, weight_cls=True, min_cls=0.05), False, True),
    # 9 .. 17 layers (the <12> / <17> instances; L = 11 pads): the layout filter with weighted classes, x2 and x1, ghost
    # mask on and off
    (dict(num_obj=10, dim=16, load_dim=32, use_lyt_filtering=True, weight_cls=True, min_cls=0.05), True, False),
    (dict(num_obj=16, obj_shape=[2, 2], dim=16, load_dim=0, use_lyt_filtering=True, use_lyt_opacity=True), False, True),
    (dict(num_obj=8, dim=16, load_dim=16, no_filter=True), False, False),
])
def test_fused_hd_backward(dev, over, ctx_only, include_self):
    """Backward of the fused full-resolution passes (csrc/flow_ctx_bwd.hip) against the CPU oracle's
    autograd AND against the per-op HIP path, for every differentiable input of the chain
    grid_to_flow[_ctx] -> input_to_output: the four grids, the object alphas, the occlusion matrix
    and the class distributions; loss over every output (flow, both alphas, disocc, fused frames,
    raw frames)."""
    _hd_backward_case(dev, opt_ns(include_self=include_self, **over), ctx_only, include_self, b=2, t=3, nl=6, seed=17)


def test_fused_hd_backward_seeded_fuzz(dev):
    """Ten more recipes from a seeded generator for the backward of the fused passes (1 ... 16 objects, x1 ... x3,
    2 ... 32 classes, 2 ... 4 frames, either context mode, the option sets): every differentiable input's gradient
    against the oracle's autograd in fp32 and fp64, as ``test_fused_hd_backward`` -- with ``close``'s allowance for a
    few elements downstream of a kink (an earlier draw of sixteen recipes had two -- one of 6.3 M
    |dist - prob| terms of the layout filter with a margin of 2.4e-9, which the kernel's softmax rounds to the other side
    and which moves the 32 class gradients of that object by 1.4e-3 of their scale, and a background-grid gradient 4 %
    over its bound)."""
    import random
    rng = random.Random(11)
    failures = []
    for case in range(10):
        dim = rng.choice([8, 12, 16])
        s = rng.choice([0, 1, 2, 2, 3])
        include_self = rng.random() < 0.5
        over = dict(num_obj=rng.randint(1, 16), dim=dim, load_dim=dim * s, obj_shape=rng.choice([[2, 2], [3, 3]]),
                    allow_ghost=rng.random() < 0.3, weight_cls=rng.random() < 0.5, no_filter=rng.random() < 0.2,
                    use_lyt_filtering=rng.random() < 0.5, use_lyt_opacity=rng.random() < 0.5)
        over["min_cls"] = 0.05 if over["weight_cls"] else 0.0
        ctx_only = (not include_self) and rng.random() < 0.5
        b, t, nl = (rng.randint(1, 2), rng.randint(2, 4), rng.choice([2, 6, 20, 21, 32])) if include_self else \
            (2, 3, rng.choice([2, 6, 20, 21, 32]))
        try:
            _hd_backward_case(dev, opt_ns(include_self=include_self, **over), ctx_only, include_self, b=b, t=t, nl=nl,
                              seed=200 + case, per_op=False, outliers=0.05)
        except AssertionError as exc:
            failures.append(f"case {case}: {over} include_self={include_self} ctx_only={ctx_only} b={b} t={t} "
                            f"nl={nl}: {exc}")
    assert not failures, "\n".join(failures)


def test_fused_hd_backward_at_recipe_size(dev):
    """The same comparison at the size the backward really runs at: the LVD recipe
    (scripts/cityscapes/train_lvd.sh, models/synthesizer.py:826-841): 128 x 256 with no HD raster,
    16 objects (L = 17 -- the <17> kernel variants, which spill at this size), 20 layout classes, 5 frames,
    ctx_mode "prev", include_self, weighted layout filter with opacity; grid_to_flow -> input_to_output
    against the oracle's autograd in fp32 and fp64."""
    opt = opt_ns(num_obj=16, obj_shape=[4, 4], patch_size=16, latent_shape=[8, 16], dim=128, load_dim=0,
                 aspect_ratio=2, use_lyt_filtering=True, use_lyt_opacity=True, weight_cls=True, min_cls=0.1,
                 include_self=True)
    # fused kernels only: they are what runs at this size.  (The per-op composition of the same chain sums
    # its object-alpha gradient with float atomics over thousands of samples and lands 4.5 x the fp32
    # oracle's own distance from float64 here -- 4 % of the largest entry, 0.9 % for the oracle, 6e-5 for
    # the fused kernels; it is checked at the sizes it is used at, in test_fused_hd_backward.)
    _hd_backward_case(dev, opt, False, True, b=1, t=5, nl=20, seed=23, per_op=False)


def _hd_backward_case(dev, opt, ctx_only, include_self, b, t, nl, seed, per_op=True, outliers=0.0):
    from waldo_amd.nets import Warper
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=seed)
    if include_self:  # ctx_mode "prev" of synthesizer.py:826-828
        ctx_ts = torch.roll(torch.arange(t), 1).view(1, 1, t).expand(b, -1, -1).contiguous()
        pred_ts = torch.arange(t)
    else:
        assert b == 2
        ctx_ts = torch.tensor([[[0], [1]], [[1], [1]]])
        pred_ts = torch.tensor([2])
    with torch.no_grad():
        grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
    names = ("tgo", "sgo", "tgb", "sgb", "obj_alpha", "occ", "cls")

    def run(fn_flow, fn_out, to, leaves, dtype=torch.float32):
        g4, oa, oc, cl = leaves[:4], leaves[4], leaves[5], leaves[6]
        r = fn_flow(inp.to(to), g4, oc, oa, bg_alpha.to(to), cl, ctx_ts.to(to), pred_ts.to(to))
        out, raw = fn_out(inp.to(to), r[3], r[0], ctx_ts.to(to))
        torch.manual_seed(2)
        loss = 0
        for x in (r[0], r[2], r[3], r[4], out, raw):
            loss = loss + (x * torch.randn(x.shape).to(to, dtype)).sum()
        loss.backward()
        return [x.grad for x in leaves], (r, out, raw)

    base = [*grid_o, obj_alpha, occ, cls]
    lo = [x.clone().requires_grad_() for x in base]
    fo = WO.grid_to_flow_ctx if ctx_only else WO.grid_to_flow
    g_o, _ = run(lambda *a: fo(cfg, *a), lambda *a: WO.input_to_output(cfg, *a), "cpu", lo)
    g_64, _ = run(lambda *a: fo(cfg, *dbl(a)), lambda *a: WO.input_to_output(cfg, *dbl(a)), "cpu",
                  [x.clone().double().requires_grad_() for x in base], dtype=torch.float64)
    grads = {}
    for fused in ((True, False) if per_op else (True,)):
        wp.fuse_hd = fused
        lh = [x.clone().to(dev).requires_grad_() for x in base]
        fh = wp.grid_to_flow_ctx if ctx_only else wp.grid_to_flow
        grads[fused], _ = run(fh, wp.input_to_output, dev, lh)
    wp.fuse_hd = True
    for i, name in enumerate(names):
        if g_o[i] is None:
            assert grads[True][i] is None or grads[True][i].abs().max() == 0, name
            continue
        # 1e-4 of the largest gradient + 4 x the fp32 oracle's distance from its float64 self (gradients
        # reach the loss through sample POSITIONS: a frame / alpha edge turns fp32 position noise into value noise)
        close(grads[True][i], g_o[i], rel=True, what=f"fused vs oracle: grad {name}", exact=g_64[i], outliers=outliers)
        if per_op:
            close(grads[False][i], g_o[i], rel=True, what=f"per-op vs oracle: grad {name}", exact=g_64[i])


def _recipe_size_case(dev, ctx_only, recipe="cityscapes"):
    """The reference's real Cityscapes recipe R (scripts/cityscapes/train_wif.sh:12-14,28): L = 17 layers, Nl = 20
    classes, 128x256 -> 512x1024, B = 1, Tc = 4, Tp = 1 -- the fused passes of grid_to_flow_ctx (``ctx_only``:
    --s_restrict_to_ctx, what demo.sh / test.sh pass: alphas composited on the Tc context frames, ghost mask) or of the
    UNRESTRICTED grid_to_flow (what train_wif.sh runs: alphas composited on all T frames, no ghost mask), and
    input_to_output, at full size against the CPU oracle in fp32 and fp64 (which materialises the reference's multi-GB
    broadcasts; ~1-2 min on the host)."""
    from waldo_amd.nets import Warper
    if recipe == "kitti":
        # scripts/kitti/*.sh (BASELINE config 4): 128 x 416 layers -> 256 x 832 frames (x2: the <8, ., 2> instances, a
        # width of 13 x 64 pixels), 7 objects + background, 8 x 26 background control points, two predicted frames
        opt = opt_ns(num_obj=7, obj_shape=[4, 4], patch_size=16, latent_shape=[8, 26], dim=128, load_dim=256,
                     aspect_ratio=3.25, use_lyt_filtering=True, weight_cls=True, min_cls=0.1)
        t, tp, tag = 6, 2, "KITTI size"
    elif recipe == "c5":
        # BASELINE config 5 (scripts/cityscapes/demo.sh / test.sh): the Cityscapes raster with 11 objects + background
        # (the <12, ., 4> instances of the C5 pipeline), class weighting, two predicted frames
        opt = opt_ns(num_obj=11, obj_shape=[4, 4], patch_size=16, latent_shape=[8, 16], dim=128, load_dim=512,
                     aspect_ratio=2, use_lyt_filtering=True, weight_cls=True, min_cls=0.1)
        t, tp, tag = 6, 2, "C5 size"
    else:
        opt = opt_ns(num_obj=16, obj_shape=[4, 4], patch_size=16, latent_shape=[8, 16], dim=128, load_dim=512,
                     aspect_ratio=2, use_lyt_filtering=True)
        t, tp, tag = 5, 1, "R size"
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    b, nl = 1, 20
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=21 if ctx_only else 22)
    ctx_ts = torch.arange(4).view(1, 4, 1).expand(1, 4, tp).contiguous()
    pred_ts = torch.arange(4, t)
    fo = WO.grid_to_flow_ctx if ctx_only else WO.grid_to_flow
    tag += ", restricted" if ctx_only else ", UNRESTRICTED"
    with torch.no_grad():
        grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
        ro = fo(cfg, inp, grid_o, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
        out_o, raw_o = WO.input_to_output(cfg, inp, ro[3], ro[0], ctx_ts)
        # float64 evaluation of the same two steps on the same fp32 inputs (the second on the fp32 flow /
        # alphas, like out_o): the measured allowance of close()
        r64 = fo(cfg, *dbl((inp, grid_o, occ, obj_alpha, bg_alpha, cls)), ctx_ts, pred_ts)
        out64, raw64 = WO.input_to_output(cfg, inp.double(), ro[3].double(), ro[0].double(), ctx_ts)
        args = (inp.to(dev), [x.to(dev) for x in grid_o], occ.to(dev), obj_alpha.to(dev), bg_alpha.to(dev),
                cls.to(dev), ctx_ts.to(dev), pred_ts.to(dev))
        assert wp.fuse_hd and wp._fused_ok(list(args[:1]), cfg.num_obj + 1, nl)
        rf = (wp.grid_to_flow_ctx if ctx_only else wp.grid_to_flow)(*args)
        # the frame warp on the ORACLE's flow / alpha: fed its own flow, a 1e-4 difference in grid units
        # is 0.05 px at 1024 columns and moves a frame value by up to ~1e-2 -- that would measure
        # the flow's tolerance a second time, not this kernel
        out_f, raw_f = wp.input_to_output(args[0], ro[3].to(dev), ro[0].to(dev), args[6])
    wp.check_time_indices()
    hd, wd = cfg.src_shape_hd
    nlay = cfg.num_obj + 1
    assert rf[0].shape == (1, 4, tp, 2, hd, wd) and rf[3].shape == (1, 4, tp, nlay, hd, wd)
    assert rf[2].shape == (1, 4 if ctx_only else t, nlay, hd, wd)  # `alpha`: on the context frames / on all T frames
    for x, y, z, name in zip(rf, ro, r64, ("flow", "alpha_unflt", "alpha", "alpha_ctx", "disocc")):
        if y is None:
            assert x is None, name
            continue
        # alpha_ctx / disocc sample the composited HD alpha at flow-displaced positions: at 1024 px a
        # position is good to ~1e-5 grid units in fp32 on either side, which a steep alpha edge turns
        # into ~1e-4 of value -- in the fp32 oracle too, which is what `exact` measures
        close(x, y, what=f"{tag}, fused vs oracle: " + name, exact=z)
    close(out_f, out_o, what=f"{tag}: output", exact=out64)
    close(raw_f, raw_o, what=f"{tag}: raw_output", exact=raw64)


def test_fused_hd_passes_at_recipe_size(dev):
    """grid_to_flow_ctx (the RESTRICTED path of demo.sh / test.sh, lvd.py:707-828) + input_to_output at the recipe's
    full size.  (train_wif.sh itself runs the unrestricted twin: the next test.)"""
    _recipe_size_case(dev, ctx_only=True)


def test_fused_hd_passes_unrestricted_at_recipe_size(dev):
    """grid_to_flow (lvd.py:602-705: the path scripts/cityscapes/train_wif.sh takes, it does not pass
    --s_restrict_to_ctx) + input_to_output at the recipe's full size: the <17> tall-tile / four-pixels-per-thread
    instances with the alphas composited on all T = 5 frames and no ghost mask."""
    _recipe_size_case(dev, ctx_only=False)


@pytest.mark.parametrize("ctx_only", [True, False])
def test_fused_hd_passes_at_kitti_size(dev, ctx_only):
    """The same comparison at BASELINE config 4's shape -- the KITTI recipe: 128 x 416 -> 256 x 832 (x2), L = 8, 20
    classes with class weighting, four context and two predicted frames -- restricted and unrestricted: the instances the
    C4 pipeline runs (``flow_ctx_warp_kernel<8, ., 2>``, ``flow_ctx_alpha_kernel<8, 20>``, ``frame_warp_fuse_lds_kernel<4>``
    on a raster 13 tiles wide) against the oracle in fp32 and fp64."""
    _recipe_size_case(dev, ctx_only, recipe="kitti")


def test_fused_hd_passes_at_c5_size(dev):
    """... and at BASELINE config 5's own shape: 512 x 1024 with L = 12 (the C5 pipeline's ``flow_ctx_warp_kernel<12, .,
    4>`` / ``flow_ctx_alpha_kernel<12, 20>``), the restricted path demo.sh / test.sh take."""
    _recipe_size_case(dev, True, recipe="c5")


@pytest.mark.parametrize("include_self", [False, True])
@pytest.mark.parametrize("shape", [(2, 3, 2, 3, 7, 5, 16, 32), (1, 2, 4, 2, 23, 17, 24, 48), (1, 3, 6, 1, 4, 3, 8, 16)])
def test_frame_warp_fuse(dev, include_self, shape):
    """Fused Warper.input_to_output (A10, csrc/flow_ctx.hip) vs the oracle and the per-op path,
    with and without the include_self branch (lvd.py:842-845: Tp == T appends the unwarped frame)."""
    from waldo_amd.nets import Warper
    b, t, tc, tp, c, nl, hd, wd = shape
    if include_self:
        tp = t
    opt = opt_ns(num_obj=nl - 1, dim=hd, load_dim=hd, include_self=include_self)
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    g = torch.Generator().manual_seed(5)
    lo = torch.randn(b * t, c, hd // 4, wd // 4, generator=g)
    inp = torch.nn.functional.interpolate(lo, size=(hd, wd), mode="bilinear").view(b, t, c, hd, wd)
    flow = 0.2 * torch.randn(b, tc, tp, 2, hd, wd, generator=g)
    alpha = torch.rand(b, tc, tp, nl, hd, wd, generator=g) * 2 - 1
    ctx_ts = torch.randint(0, t, (b, tc, tp), generator=g)
    out_o, raw_o = WO.input_to_output(cfg, inp, alpha, flow, ctx_ts)
    with torch.no_grad():
        args = (inp.to(dev), alpha.to(dev), flow.to(dev), ctx_ts.to(dev))
        out_f, raw_f = wp.input_to_output(*args)
        wp.fuse_hd = False
        out_u, raw_u = wp.input_to_output(*args)
    close(out_f, out_o, what="out vs oracle")
    close(raw_f, raw_o, what="raw vs oracle")
    close(out_f, out_u, what="out vs per-op")
    close(raw_f, raw_u, what="raw vs per-op")


@pytest.mark.parametrize("amp_px,tc", [(10.0, 2), (50.0, 2), (150.0, 2), (50.0, 1)])
def test_frame_warp_fuse_at_256x512(dev, amp_px, tc):
    """input_to_output at a 256 x 512 raster with flows of 10 and 50 px amplitude, smooth over 32-pixel cells (the
    regimes the C4 / C5 pipelines operate in; the small-raster test above cannot leave a tile's neighbourhood), and of
    150 px (neighbouring cells land far apart: the folded warp of the pipelines' `--motion wild`, local stretch up to
    ~10): forward against the oracle in fp32 and fp64, backward (flow and alpha gradients) against its autograd."""
    from waldo_amd import functional as WF
    b, t, tp, c, nl, hd, wd = 1, 3, 2, 5, 3, 256, 512  # (tc = 1: the <1> instances of the forward and backward kernels)
    opt = opt_ns(num_obj=nl - 1, dim=hd, aspect_ratio=2.0, load_dim=hd)
    cfg = WO.WarperCfg.from_opt(opt)
    g = torch.Generator().manual_seed(int(amp_px))
    lo = torch.randn(b * t, c, hd // 4, wd // 4, generator=g)
    inp = torch.nn.functional.interpolate(lo, size=(hd, wd), mode="bilinear").view(b, t, c, hd, wd)
    # grid units: amp_px pixels of the 512-wide raster, smooth over 32-pixel cells, plus a shear across the frame
    fl = (amp_px * 2 / wd) * torch.randn(b * tc * tp, 2, hd // 32, wd // 32, generator=g)
    flow = torch.nn.functional.interpolate(fl, size=(hd, wd), mode="bilinear").view(b, tc, tp, 2, hd, wd)
    flow = flow + (amp_px * 2 / wd) * torch.linspace(-1, 1, hd).view(1, 1, 1, 1, hd, 1)
    alpha = torch.rand(b, tc, tp, nl, hd, wd, generator=g) * 2 - 1
    ctx_ts = torch.randint(0, t, (b, tc, tp), generator=g)
    w_out = torch.randn(b, tp, c + 1, hd, wd, generator=g)
    w_raw = torch.randn(b, tc, tp, c + nl, hd, wd, generator=g)

    def ref(dtype):
        f, a = flow.clone().to(dtype).requires_grad_(), alpha.clone().to(dtype).requires_grad_()
        out, raw = WO.input_to_output(cfg, inp.to(dtype), a, f, ctx_ts)
        ((out * w_out.to(dtype)).sum() + (raw * w_raw.to(dtype)).sum()).backward()
        return out.detach(), raw.detach(), f.grad, a.grad

    r32, r64 = ref(torch.float32), ref(torch.float64)
    f, a = flow.to(dev).requires_grad_(), alpha.to(dev).requires_grad_()
    out, raw = WF.frame_warp_fuse(inp.to(dev), f, a, ctx_ts.to(dev))
    close(out, r32[0], what=f"{amp_px} px: out", exact=r64[0])
    close(raw, r32[1], what=f"{amp_px} px: raw", exact=r64[1])
    ((out * w_out.to(dev)).sum() + (raw * w_raw.to(dev)).sum()).backward()
    close(f.grad, r32[2], rel=True, what=f"{amp_px} px: grad_flow", exact=r64[2])
    close(a.grad, r32[3], rel=True, what=f"{amp_px} px: grad_alpha", exact=r64[3])


def test_frame_warp_fuse_seeded_fuzz(dev):
    """Twenty shapes of ``input_to_output`` from a seeded generator -- rasters of 5 ... 90 by 5 ... 160 pixels, 1 ... 4
    contexts, 1 ... 3 predicted frames (or the include_self branch), 1 ... 26 channels, 1 ... 17 alpha layers, flows of
    1 ... 60 pixels (smooth, with a shear: the staged boxes and the per-context gather fallback both occur), random
    context indices: forward with and without autograd (the staged no-grad kernel) and both gradients against the
    oracle in fp32 and fp64."""
    import random
    from waldo_amd import functional as WF
    rng = random.Random(4242)
    for case in range(20):
        hd, wd = rng.randint(5, 90), rng.randint(5, 160)
        b, tc, c, nl = rng.randint(1, 2), rng.randint(1, 4), rng.randint(1, 26), rng.randint(1, 17)
        include_self = rng.random() < 0.3
        t = rng.randint(2, 4)
        tp = t if include_self else rng.randint(1, 3)
        amp_px = rng.choice([1.0, 4.0, 15.0, 60.0])
        what = (f"case {case}: {hd}x{wd} b={b} t={t} tc={tc} tp={tp} c={c} nl={nl} include_self={include_self} "
                f"amp={amp_px}")
        opt = opt_ns(num_obj=nl - 1, dim=hd, aspect_ratio=(wd + 0.5) / hd, load_dim=hd, include_self=include_self)
        cfg = WO.WarperCfg.from_opt(opt)
        g = torch.Generator().manual_seed(case)
        lo = torch.randn(b * t, c, max(hd // 4, 2), max(wd // 4, 2), generator=g)
        inp = torch.nn.functional.interpolate(lo, size=(hd, wd), mode="bilinear").view(b, t, c, hd, wd)
        fl = (amp_px * 2 / wd) * torch.randn(b * tc * tp, 2, max(hd // 16, 2), max(wd // 16, 2), generator=g)
        flow = torch.nn.functional.interpolate(fl, size=(hd, wd), mode="bilinear").view(b, tc, tp, 2, hd, wd)
        flow = flow + (amp_px * 2 / wd) * torch.linspace(-1, 1, hd).view(1, 1, 1, 1, hd, 1)
        alpha = torch.rand(b, tc, tp, nl, hd, wd, generator=g) * 2 - 1
        ctx_ts = torch.randint(0, t, (b, tc, tp), generator=g)
        n_out = tc + 1 if include_self else tc
        w_out = torch.randn(b, tp, c + 1, hd, wd, generator=g)
        w_raw = torch.randn(b, n_out, tp, c + nl, hd, wd, generator=g)

        def ref(dtype):
            f, a = flow.clone().to(dtype).requires_grad_(), alpha.clone().to(dtype).requires_grad_()
            out, raw = WO.input_to_output(cfg, inp.to(dtype), a, f, ctx_ts)
            ((out * w_out.to(dtype)).sum() + (raw * w_raw.to(dtype)).sum()).backward()
            return out.detach(), raw.detach(), f.grad, a.grad

        try:
            r32, r64 = ref(torch.float32), ref(torch.float64)
            f, a = flow.to(dev).requires_grad_(), alpha.to(dev).requires_grad_()
            out, raw = WF.frame_warp_fuse(inp.to(dev), f, a, ctx_ts.to(dev), include_self=include_self)
            close(out, r32[0], what="out", exact=r64[0])
            close(raw, r32[1], what="raw", exact=r64[1])
            ((out * w_out.to(dev)).sum() + (raw * w_raw.to(dev)).sum()).backward()
            close(f.grad, r32[2], rel=True, what="grad_flow", exact=r64[2], outliers=0.001)
            close(a.grad, r32[3], rel=True, what="grad_alpha", exact=r64[3])
            with torch.no_grad():  # the staged kernel of the no-grad path: the same values
                out2, raw2 = WF.frame_warp_fuse(inp.to(dev), flow.to(dev), alpha.to(dev), ctx_ts.to(dev),
                                                include_self=include_self)
            close(out2, r32[0], what="out (no grad)", exact=r64[0])
            close(raw2, r32[1], what="raw (no grad)", exact=r64[1])
        except AssertionError as exc:
            raise AssertionError(f"{what}: {exc}") from exc


@pytest.mark.parametrize("tc,include_self", [(4, False), (2, False), (3, True), (4, True), (1, False), (1, True)])
@pytest.mark.parametrize("hw", [(64, 128), (37, 100), (8, 36), (128, 256)])
@pytest.mark.parametrize("amp_px", [3.0, 40.0])
def test_frame_warp_fuse_staged_boxes_same_bits(dev, tc, include_self, hw, amp_px):
    """frame_warp_fuse with the contexts' footprint boxes staged in LDS (frame_warp_fuse_lds_kernel: rows that start on a
    multiple of four texels from a 16-byte aligned base) against the pair gathers of frame_warp_fuse_kernel -- taken
    when the frames' base address is NOT 16-byte aligned, which a view one float into a buffer arranges: the same bits
    for `out` and `raw`, with four contexts (every vector-memory operation unconditional) and with fewer / the unwarped
    frame itself as a context, on rasters with ragged tiles (threads past the edge duplicate the last pixel's stores),
    under flows whose boxes fit (3 px) and flows of which some tiles' boxes do not (40 px over 16-pixel cells: those
    tiles gather)."""
    from waldo_amd import functional as WF
    hd, wd = hw
    b, t, c, nl = 2, 5, 6, 3
    tp = t if include_self else 3
    g = torch.Generator(device=dev).manual_seed(hd + tc)
    buf = torch.randn(b * t * c * hd * wd + 1, generator=g, device=dev)
    inp = buf[:-1].view(b, t, c, hd, wd)
    inp_off = buf[1:].view(b, t, c, hd, wd)
    inp_off.copy_(inp.clone())
    inp = inp_off.clone()                                       # 16-byte aligned copy of the same values
    assert inp.data_ptr() % 16 == 0 and inp_off.data_ptr() % 16 == 4
    fl = (amp_px * 2 / wd) * torch.randn(b * tc * tp, 2, max(hd // 16, 1), max(wd // 16, 1), generator=g, device=dev)
    flow = torch.nn.functional.interpolate(fl, size=(hd, wd), mode="bilinear").view(b, tc, tp, 2, hd, wd).contiguous()
    flow[0, 0, 0, :, : hd // 2] += 2.5                          # a block of samples outside the frame
    alpha = torch.rand(b, tc, tp, nl, hd, wd, generator=g, device=dev) * 2 - 1
    ctx_ts = torch.randint(0, t, (b, tc, tp), generator=g, device=dev)
    with torch.no_grad():
        out, raw = WF.frame_warp_fuse(inp, flow, alpha, ctx_ts, include_self=include_self)
        out_g, raw_g = WF.frame_warp_fuse(inp_off, flow, alpha, ctx_ts, include_self=include_self)
    assert torch.equal(out, out_g), (out - out_g).abs().max().item()
    assert torch.equal(raw, raw_g), (raw - raw_g).abs().max().item()


@pytest.mark.parametrize("include_self", [False, True])
@pytest.mark.parametrize("shape", [(2, 3, 2, 2, 7, 5, 4, 16, 32, 2, True), (1, 4, 4, 3, 23, 12, 8, 32, 4, 4, False),
                                   (1, 2, 5, 1, 4, 3, 8, 16, 1, 1, True)])
def test_alpha_ctx_written_into_raw_slots(dev, include_self, shape):
    """waldo_flow_ctx_warp_raw_fwd + waldo_frame_warp_fuse_raw_fwd (the context alphas composited straight into
    raw_output's slots, one score plane per context) == waldo_flow_ctx_warp_fwd + waldo_frame_warp_fuse_fwd,
    bit for bit, for every output; frame_warp_fuse on the alpha VIEW (the long way: it reads and copies the alphas)
    gives the same bits again, also after the caller edited the view in place."""
    from waldo_amd import functional as WF
    b, t, tc, tp, c, nl, h, w, s, tw, ghost = shape
    if include_self:
        tp = t
    hd, wd = h * s, w * s
    g = torch.Generator(device=dev).manual_seed(nl * 10 + tc)
    m = b * tc * tp
    flow_lr = 0.1 * torch.randn(m, nl, 2, h, w, generator=g, device=dev)
    isobj = torch.rand(m, nl - 1, h, w, generator=g, device=dev) * 1.2 if ghost else None
    a01 = torch.rand(b * tw, nl, hd, wd, generator=g, device=dev)
    occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
    ctx_ts = torch.randint(0, tw, (b, tc, tp), generator=g, device=dev)
    pred_ts = torch.randint(0, t, (tp,), generator=g, device=dev)
    inp = torch.randn(b, t, c, hd, wd, generator=g, device=dev)
    with torch.no_grad():
        flow, actx, dis, amax = WF.flow_ctx_warp(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, s, layer_max=True)
        out, raw = WF.frame_warp_fuse(inp, flow.view(b, tc, tp, 2, hd, wd), actx.view(b, tc, tp, nl, hd, wd), ctx_ts,
                                      include_self=include_self)
        flow2, actx2, dis2, amax2, slots = WF.flow_ctx_warp_into_raw(flow_lr, isobj, a01, ctx_ts, pred_ts, occ, tw, s, c,
                                                                     include_self, layer_max=True)
        assert tuple(actx2.shape) == (b, tc, tp, nl, hd, wd) and not actx2.is_contiguous()
        out2, raw2 = WF.frame_warp_fuse_raw(inp, flow2.view(b, tc, tp, 2, hd, wd), slots, ctx_ts)
        assert raw2.data_ptr() == slots.raw.data_ptr()  # the short way: raw is the tensor the alphas went into
        for x, y, name in ((flow, flow2, "flow"), (actx.view(b, tc, tp, nl, hd, wd), actx2, "alpha_ctx"), (dis, dis2, "disocc"),
                           (amax, amax2, "alpha max"), (out, out2, "out"), (raw, raw2, "raw")):
            assert torch.equal(x, y), name
        # the score plane is what frame_warp_fuse would have summed
        sc = ((actx.view(b, tc, tp, nl, hd, wd)[:, :, :, 0] + 1) / 2)
        for l in range(1, nl):
            sc = sc + (actx.view(b, tc, tp, nl, hd, wd)[:, :, :, l] + 1) / 2
        assert torch.equal(slots.score, sc)
        # the strided view is an ordinary tensor: the long way on it (a caller of Warper.input_to_output with the
        # alpha_ctx that decode_output returned), before and after an in-place edit
        out3, raw3 = WF.frame_warp_fuse(inp, flow2.view(b, tc, tp, 2, hd, wd), actx2, ctx_ts, include_self=include_self)
        assert raw3.data_ptr() != slots.raw.data_ptr() and torch.equal(out3, out) and torch.equal(raw3, raw)
        actx2[:, 0, 0, 0] = 0.25
        out4, raw4 = WF.frame_warp_fuse(inp, flow2.view(b, tc, tp, 2, hd, wd), actx2, ctx_ts, include_self=include_self)
        ref = actx.view(b, tc, tp, nl, hd, wd).clone()
        ref[:, 0, 0, 0] = 0.25
        out5, raw5 = WF.frame_warp_fuse(inp, flow.view(b, tc, tp, 2, hd, wd), ref, ctx_ts, include_self=include_self)
        assert torch.equal(out4, out5) and torch.equal(raw4, raw5)


@pytest.mark.parametrize("shape", [(2, 5, 3, 4, 3, 6, 5, 7), (1, 3, 1, 1, 1, 2, 2, 3), (2, 4, 5, 2, 2, 8, 16, 8)])
def test_time_gather_against_the_spelled_out_expressions(dev, shape):
    """waldo_time_gather_* against gather_time / [:, pred_ts] / subtract / permute / expand as the reference
    writes them (oracle.layer_flows): values bit for bit (a subtraction and copies), gradients -- sums over
    the output frames that read an input frame, repeated indices included -- against autograd's."""
    from waldo_amd import functional as WF
    from waldo_amd._lib import WaldoHipError
    b, t, tc, tp, no, ho, h, w = shape
    g = torch.Generator().manual_seed(b * 100 + t)
    grid = [torch.randn(b, t, no, ho, ho + 1, 2, generator=g), torch.randn(b, t, no, h, w, 2, generator=g),
            torch.randn(b, t, h, w, 2, generator=g), torch.randn(b, t, h, w, 2, generator=g)]
    ctx_ts = torch.randint(0, t, (b, tc, tp), generator=g)   # repeats on purpose
    pred_ts = torch.randint(0, t, (tp,), generator=g)
    ref_in = [x.clone().requires_grad_() for x in grid]
    ref = WO.layer_flows(ref_in, ctx_ts, pred_ts)
    hip_in = [x.to(dev).requires_grad_() for x in grid]
    cd, pd = ctx_ts.to(dev), pred_ts.to(dev)
    hip = (WF.time_gather(hip_in[0], cd, pd, subtract=True, channel_first=True),
           WF.time_gather(hip_in[2].unsqueeze(2), cd, pd, subtract=True, channel_first=True).squeeze(3),
           WF.time_gather(hip_in[1], None, pd, num_ctx=tc), WF.time_gather(hip_in[3], None, pd, num_ctx=tc))
    ws = [torch.randn(r.shape, generator=g) for r in ref]
    for r, o in zip(ref, hip):
        assert tuple(o.shape) == tuple(r.shape)
        assert torch.equal(o.detach().cpu(), r.detach())
    sum((r * wt).sum() for r, wt in zip(ref, ws)).backward()
    sum((o * wt.to(dev)).sum() for o, wt in zip(hip, ws)).backward()
    for name, r, o in zip(("tgt_grid_obj", "src_grid_obj", "tgt_grid_bg", "src_grid_bg"), ref_in, hip_in):
        close(o.grad.cpu(), r.grad, 1e-5, what=f"grad {name}")
    # plain gather_time, and the index checks the reference's gather() makes
    out = WF.time_gather(hip_in[3].detach(), cd, pd)
    assert torch.equal(out.cpu(), WO.gather_time(grid[3], ctx_ts))
    with pytest.raises(WaldoHipError):
        WF.time_gather(hip_in[3].detach(), cd + t, pd)
    with pytest.raises(WaldoHipError):
        WF.time_gather(hip_in[3].detach(), None, pd, subtract=True)
    empty = WF.time_gather(hip_in[3].detach()[:0], cd[:0], pd)
    assert tuple(empty.shape) == (0, tc, tp, h, w, 2)


def test_time_gather_of_every_frame_in_order_is_the_clip_itself(dev):
    """`x[:, pred_ts]` for one context with pred_ts = 0 .. T-1 (the LVD recipe's ctx_mode "prev": every frame is
    predicted): for an index MADE by `WF.arange_index` -- known on the host to be the identity, no device read --
    `time_gather` hands out a VIEW of the clip: the same values and gradients as the kernel's copy.  A plain
    `torch.arange`, the marked index after an in-place change, a permutation, a shorter index or two contexts go
    through the kernel."""
    from waldo_amd import functional as WF
    b, t, no, h, w = 2, 5, 3, 6, 7
    g = torch.Generator(device=dev).manual_seed(8)
    x = torch.randn(b, t, no, h, w, 2, generator=g, device=dev, requires_grad=True)
    wgt = torch.randn(b, 1, t, no, h, w, 2, generator=g, device=dev)
    pred = WF.arange_index(t, dev)
    assert pred.dtype == torch.int64 and WF.normalise_time_index(pred) is pred
    out = WF.time_gather(x, None, pred, num_ctx=1)
    assert out.shape == (b, 1, t, no, h, w, 2) and out.data_ptr() == x.data_ptr()
    (out * wgt).sum().backward()
    g_view = x.grad.clone()
    x.grad = None
    perm = torch.tensor([1, 0, 2, 3, 4], device=dev)
    for idx, nctx in ((perm, 1), (pred[:3], 1), (pred, 2), (torch.arange(t, device=dev), 1), (pred + 0, 1)):
        o = WF.time_gather(x, None, idx, num_ctx=nctx)
        assert o.data_ptr() != x.data_ptr()
        assert torch.equal(o, x[:, idx].unsqueeze(1).expand(-1, nctx, *([-1] * 5)))
    pred.mul_(0)                                                   # the same tensor object, a new version: frame 0 five times
    rep = WF.time_gather(x, None, pred, num_ctx=1)
    assert rep.data_ptr() != x.data_ptr() and torch.equal(rep, x[:, :1].expand(-1, t, -1, -1, -1, -1).unsqueeze(1))
    # the kernel's gradient for the identity index (forced through the kernel by a second context) is the view's
    both = WF.time_gather(x, None, torch.arange(t, device=dev), num_ctx=2)
    (both[:, :1] * wgt).sum().backward()
    assert torch.equal(x.grad, g_view)


@pytest.mark.parametrize("shape", [(2, 3, 7, 16, 32, 2, 2, 3), (1, 2, 23, 32, 64, 4, 2, 3), (1, 1, 4, 8, 8, 8, 1, 0)])
def test_downscale_frames_matches_interpolate(dev, shape):
    """waldo_downscale_frames_fwd == scale(input[:, :Tw, c0:], 1 / S) (lvd.py:611 through lvd.py:175-179), bit for bit."""
    from waldo_amd import functional as WF
    from waldo_amd._lib import WaldoHipError
    b, t, c, hd, wd, s, tw, c0 = shape
    g = torch.Generator().manual_seed(c * 10 + s)
    inp = torch.randn(b, t, c, hd, wd, generator=g)
    ref = WO.rescale(inp[:, :tw, c0:], 1 / s)
    out = WF.downscale_frames(inp.to(dev), tw, c0, s)
    assert tuple(out.shape) == tuple(ref.shape) == (b, tw, c - c0, hd // s, wd // s)
    close(out, ref, 1e-6, what="vs the oracle's F.interpolate")  # (the CPU kernel associates the four terms differently)
    # the launches it replaces -- the framework's copy of the slice + its bilinear kernel on the device -- bit for bit
    per_op = torch.nn.functional.interpolate(inp.to(dev)[:, :tw, c0:].reshape(-1, c - c0, hd, wd), scale_factor=1 / s,
                                             mode="bilinear").view(out.shape)
    assert torch.equal(out, per_op)
    with pytest.raises(WaldoHipError):
        WF.downscale_frames(inp.to(dev), tw, c0, 3)


def test_time_indices_outside_the_window_are_refused(dev):
    """gather_time (lvd.py:462-467) fails for an index outside the time axis; the fused kernels index with
    ctx_ts / pred_ts directly: they clamp (memory safety) and REPORT in the caller's status words
    (include/waldo_hip.h: "Frame-index status").  A stand-alone call (no status given) checks before it returns."""
    from waldo_amd import _lib, functional as WF
    b, t, tc, tp, c, nl, hd, wd = 1, 3, 2, 1, 4, 3, 8, 16
    inp = torch.randn(b, t, c, hd, wd, device=dev)
    flow = torch.zeros(b, tc, tp, 2, hd, wd, device=dev)
    alpha = torch.zeros(b, tc, tp, nl, hd, wd, device=dev)
    ok = torch.tensor([[[0], [2]]], device=dev)
    WF.frame_warp_fuse(inp, flow, alpha, ok)
    for bad, shown in ((torch.tensor([[[0], [3]]], device=dev), "index 3"), (torch.tensor([[[-1], [1]]], device=dev), "index -1")):
        with pytest.raises(_lib.WaldoHipError, match=rf"ctx_ts holds the {shown}, valid range is \[0, 2\]"):
            WF.frame_warp_fuse(inp, flow, alpha, bad)
    WF.frame_warp_fuse(inp, flow, alpha, ok)  # (the words were cleared by the raise)
    flow_lr = torch.zeros(b * tc * tp, nl, 2, hd, wd, device=dev)
    a01 = torch.rand(b * 2, nl, hd, wd, device=dev)  # window of tw = 2 frames
    occ = torch.zeros(b, t, nl, nl, device=dev)
    pred = torch.tensor([2], device=dev)
    WF.flow_ctx_warp(flow_lr, None, a01, torch.tensor([[[0], [1]]], device=dev), pred, occ, 2, 1)
    with pytest.raises(_lib.WaldoHipError, match=r"ctx_ts holds the index 2, valid range is \[0, 1\]"):  # inside T, outside the window
        WF.flow_ctx_warp(flow_lr, None, a01, ok, pred, occ, 2, 1)
    with pytest.raises(_lib.WaldoHipError, match=r"pred_ts holds the index 3, valid range is \[0, 2\]"):
        WF.flow_ctx_warp(flow_lr, None, a01, torch.tensor([[[0], [1]]], device=dev), torch.tensor([3], device=dev), occ, 2, 1)
    with pytest.raises(_lib.WaldoHipError, match="valid range"):
        WF.flow_ctx_warp_into_raw(flow_lr, None, a01, ok, pred, occ, 2, 1, c, False)
    # a caller with status words of its own decides when to look: nothing raises at the call ...
    st = _lib.IndexStatus()
    out_bad, _ = WF.frame_warp_fuse(inp, flow, alpha, torch.tensor([[[0], [7]]], device=dev), status=st)
    out_clamped, _ = WF.frame_warp_fuse(inp, flow, alpha, torch.tensor([[[0], [2]]], device=dev), status=st)
    assert torch.equal(out_bad, out_clamped)  # (memory-safe: the frame was clamped)
    with pytest.raises(_lib.WaldoHipError, match=r"ctx_ts holds the index 7"):
        st.check(sync=True)
    st.check(sync=True)  # sticky until read, then clear


def test_time_indices_are_checked_in_a_captured_step(dev):
    """The same validation when the step is a HIP graph: the check is IN the kernels, so a bad index copied into
    the graph's static index tensor is reported by the replay that meets it (the host-side check of rounds 3-5
    could not run during capture and the kernels clamped silently)."""
    from waldo_amd import _lib
    from waldo_amd.graphs import GraphedCall
    from waldo_amd.nets import Warper, decode_output, estimate_alpha_grid_occ
    opt = opt_ns()
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    b, t, nl = 2, 3, 5
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=8)
    occ_score = torch.randn(b, t, cfg.num_obj, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        occ_d, oa, ba, grid = estimate_alpha_grid_occ(wp, obj_alpha.to(dev), bg_alpha[:1].to(dev), obj_pose.to(dev),
                                                      bg_pose.to(dev), occ_score.to(dev))
    grid = [x.contiguous() for x in grid]
    ctx_ts = torch.arange(2, device=dev).view(1, 2, 1).expand(b, 2, 1).contiguous()
    pred_ts = torch.tensor([2], device=dev)
    inp_d, cls_d = inp.to(dev), cls.to(dev)

    def step(ctx, pred):
        return decode_output(wp, inp_d, grid, occ_d, oa, ba, cls_d, ctx, pred)[0]

    with torch.no_grad():
        eager = step(ctx_ts, pred_ts).clone()
    graphed = GraphedCall(step, ctx_ts, pred_ts)
    assert torch.equal(graphed(ctx_ts, pred_ts), eager)
    graphed.check()
    # (a replay looks at the words without waiting: the report comes from the call or, at the latest, from check())
    with pytest.raises(_lib.WaldoHipError, match=r"pred_ts holds the index 5, valid range is \[0, 2\]"):
        graphed(ctx_ts, torch.tensor([5], device=dev))  # (copied into the graph's static pred_ts)
        graphed.check()
    bad_ctx = ctx_ts.clone()
    bad_ctx[1, 0, 0] = 2  # inside T, outside the context window of restrict_to_ctx
    with pytest.raises(_lib.WaldoHipError, match=r"ctx_ts holds the index 2, valid range is \[0, 1\]"):
        graphed(bad_ctx, pred_ts)
        graphed.check()
    assert torch.equal(graphed(ctx_ts, pred_ts), eager)
    graphed.check()
    # eagerly through the module: reported at the latest by check_time_indices()
    with pytest.raises(_lib.WaldoHipError, match=r"pred_ts holds the index 4"), torch.no_grad():
        step(ctx_ts, torch.tensor([4], device=dev))
        wp.check_time_indices()


def test_index_validation_under_inference_mode(dev):
    """Tensors made under torch.inference_mode() have no version counter: the device-side validation does not care --
    an in-place change of the caller's index tensor is seen by the next launch -- and the whole decode runs under it."""
    from waldo_amd import _lib, functional as WF
    from waldo_amd.nets import Warper, decode_output, estimate_alpha_grid_occ
    b, t, tc, tp, c, nl, hd, wd = 1, 3, 2, 1, 4, 3, 8, 16
    with torch.inference_mode():
        inp = torch.randn(b, t, c, hd, wd, device=dev)
        flow = torch.zeros(b, tc, tp, 2, hd, wd, device=dev)
        alpha = torch.zeros(b, tc, tp, nl, hd, wd, device=dev)
        ok = torch.arange(tc, device=dev).view(1, tc, 1)  # an inference tensor
        assert ok.is_inference()
        WF.frame_warp_fuse(inp, flow, alpha, ok)
        ok.fill_(t)  # written in place, unseen by any version counter: the next call must notice
        with pytest.raises(_lib.WaldoHipError, match="valid range"):
            WF.frame_warp_fuse(inp, flow, alpha, ok)
        idx = WF.normalise_time_index(torch.arange(tc, device=dev).view(1, tc, 1).expand(b, tc, tp))
        assert idx.is_contiguous() and idx.dtype == torch.int64
        WF.frame_warp_fuse(inp, flow, alpha, idx)
        assert WF.time_gather(inp.new_zeros(b, t, 2, 2), None, WF.arange_index(t, dev), num_ctx=1).shape == (b, 1, t, 2, 2)
    # the chain of test_decode_output_glue under inference mode == under no_grad, bit for bit
    opt = opt_ns()
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    bb, tt, nll = 2, 3, 5
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, bb, tt, nll, seed=8)
    occ_score = torch.randn(bb, tt, cfg.num_obj, generator=torch.Generator().manual_seed(4))
    outs = []
    for mode in (torch.no_grad, torch.inference_mode):
        with mode():
            ctx_ts = torch.arange(2, device=dev).view(1, 2, 1).expand(bb, 2, tt)  # expanded view, as synthesizer.py:438
            pred_ts = torch.arange(tt, device=dev)
            occ_d, oa, ba, grid = estimate_alpha_grid_occ(wp, obj_alpha.to(dev), bg_alpha[:1].to(dev), obj_pose.to(dev),
                                                          bg_pose.to(dev), occ_score.to(dev))
            outs.append(decode_output(wp, inp.to(dev), grid, occ_d, oa, ba, cls.to(dev), ctx_ts, pred_ts))
    wp.check_time_indices()
    for x, y in zip(*outs):
        assert (x is None and y is None) or torch.equal(x, y)


@pytest.mark.parametrize("restrict,use_disocc,include_self", [(True, False, False), (True, True, False),
                                                              (False, True, False), (True, True, True)])
def test_decode_output_glue(dev, restrict, use_disocc, include_self):
    """LVD.forward(mode="decode_output") / "estimate_alpha_grid_occ" glue (A11, lvd.py:126-153) vs the
    oracle's restatement, incl. the use_disocc / include_self branches; the downstream chain runs on
    the oracle's grids (grid inversion is compared on identical inputs elsewhere)."""
    from waldo_amd.nets import Warper, decode_output, estimate_alpha_grid_occ
    opt = opt_ns(include_self=include_self)
    cfg = WO.WarperCfg.from_opt(opt)
    wp = Warper(opt).to(dev)
    b, t, nl = 2, 3, 5
    obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=8)
    tp = t if include_self else 1
    ctx_ts = torch.randint(0, 2, (b, 2, tp), generator=torch.Generator().manual_seed(2))
    pred_ts = torch.arange(t) if include_self else torch.tensor([2])
    occ_score = torch.randn(b, t, cfg.num_obj, generator=torch.Generator().manual_seed(4))
    mask = (torch.rand(1, 1, 1, *cfg.tgt_shape, generator=torch.Generator().manual_seed(6)) > 0.2).float()
    with torch.no_grad():
        grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
        ref = WO.decode_output(cfg, inp, grid_o, O.compute_occ(occ_score), mask * obj_alpha + (1 - mask) * -1.0,
                               bg_alpha, cls, ctx_ts, pred_ts, restrict, use_disocc)
        ref64 = WO.decode_output(cfg, *dbl((inp, grid_o, O.compute_occ(occ_score), mask * obj_alpha + (1 - mask) * -1.0,
                                            bg_alpha, cls)), ctx_ts, pred_ts, restrict, use_disocc)
        occ_h, oa_h, ba_h, grid_h = estimate_alpha_grid_occ(wp, obj_alpha.to(dev), bg_alpha[:1].to(dev),
                                                            obj_pose.to(dev), bg_pose.to(dev), occ_score.to(dev),
                                                            obj_alpha_mask=mask.to(dev))
        close(occ_h, O.compute_occ(occ_score), what="occ")
        close(grid_h[0], grid_o[0], what="tgo")
        got = decode_output(wp, inp.to(dev), [x.to(dev) for x in grid_o], occ_h, oa_h, ba_h, cls.to(dev),
                            ctx_ts.to(dev), pred_ts.to(dev), restrict, use_disocc)
    for x, y, z, name in zip(got, ref, ref64, ("output", "flow", "alpha_unflt", "alpha", "raw_alpha", "raw_output", "alpha_ctx")):
        close(x, y, what=name, exact=z)
    # a caller that drops `alpha` / `alpha_unflt` (Synthesizer.predict does) can have the flow pass not write them:
    # None in their place, every other output bit for bit
    wp.return_alpha = False
    with torch.no_grad():
        lean = decode_output(wp, inp.to(dev), [x.to(dev) for x in grid_o], occ_h, oa_h, ba_h, cls.to(dev),
                             ctx_ts.to(dev), pred_ts.to(dev), restrict, use_disocc)
    for x, y, name in zip(lean, got, ("output", "flow", "alpha_unflt", "alpha", "raw_alpha", "raw_output", "alpha_ctx")):
        if name in ("alpha_unflt", "alpha") and wp.fuse_hd and x is None:
            continue
        assert (x is None and y is None) or torch.equal(x, y), name
    assert lean[3] is None or not wp._fused_ok([inp.to(dev)], nl, inp.size(2) - 3)


def test_decode_output_glue_seeded_fuzz(dev):
    """Fourteen more configurations of the ``estimate_alpha_grid_occ`` -> ``decode_output`` glue from a seeded generator
    (1 ... 12 objects, x1 ... x4 rasters, 2 ... 21 classes, 1 ... 2 clips, the restrict / use_disocc / include_self
    branches, the option sets, padding masks) against the oracle in fp32 and fp64 -- the seven outputs of
    lvd.py:141-153, the lean form (``return_alpha = False``) bit for bit."""
    import random
    from waldo_amd.nets import Warper, decode_output, estimate_alpha_grid_occ
    rng = random.Random(19)
    names = ("output", "flow", "alpha_unflt", "alpha", "raw_alpha", "raw_output", "alpha_ctx")
    for case in range(14):
        include_self = rng.random() < 0.3
        restrict = True if include_self else rng.random() < 0.6
        use_disocc = rng.random() < 0.5
        dim, s = rng.choice([8, 12, 16]), rng.choice([0, 1, 2, 3, 4])
        over = dict(num_obj=rng.randint(1, 12), dim=dim, load_dim=dim * s, allow_ghost=rng.random() < 0.3,
                    weight_cls=rng.random() < 0.4, no_filter=rng.random() < 0.2, include_self=include_self)
        over["min_cls"] = 0.05 if over["weight_cls"] else 0.0
        opt = opt_ns(**over)
        cfg = WO.WarperCfg.from_opt(opt)
        wp = Warper(opt).to(dev)
        b, tc = rng.randint(1, 2), rng.randint(1, 3)
        t = tc + rng.randint(1, 2)
        nl = rng.choice([2, 5, 20, 21])
        what = f"case {case}: {over} restrict={restrict} use_disocc={use_disocc} b={b} t={t} tc={tc} nl={nl}"
        obj_pose, bg_pose, inp, occ, obj_alpha, bg_alpha, cls = _warper_inputs(cfg, b, t, nl, seed=300 + case)
        g = torch.Generator().manual_seed(case)
        tp = t if include_self else t - tc
        ctx_ts = torch.randint(0, tc, (b, tc, tp), generator=g)
        pred_ts = torch.arange(t) if include_self else torch.arange(tc, t)
        occ_score = torch.randn(b, t, cfg.num_obj, generator=g)
        mask = (torch.rand(1, 1, 1, *cfg.tgt_shape, generator=g) > 0.2).float()
        masked = mask * obj_alpha + (1 - mask) * -1.0
        try:
            with torch.no_grad():
                grid_o = WO.warper_grids(cfg, obj_pose, bg_pose)
                ref = WO.decode_output(cfg, inp, grid_o, O.compute_occ(occ_score), masked, bg_alpha, cls, ctx_ts, pred_ts,
                                       restrict, use_disocc)
                ref64 = WO.decode_output(cfg, *dbl((inp, grid_o, O.compute_occ(occ_score), masked, bg_alpha, cls)), ctx_ts,
                                         pred_ts, restrict, use_disocc)
                occ_h, oa_h, ba_h, grid_h = estimate_alpha_grid_occ(wp, obj_alpha.to(dev), bg_alpha[:1].to(dev),
                                                                    obj_pose.to(dev), bg_pose.to(dev), occ_score.to(dev),
                                                                    obj_alpha_mask=mask.to(dev))
                close(occ_h, O.compute_occ(occ_score), what="occ")
                close(grid_h[0], grid_o[0], what="tgo")
                close(grid_h[2], grid_o[2], what="tgb")
                args = (inp.to(dev), [x.to(dev) for x in grid_o], occ_h, oa_h, ba_h, cls.to(dev), ctx_ts.to(dev),
                        pred_ts.to(dev), restrict, use_disocc)
                got = decode_output(wp, *args)
                wp.return_alpha = False
                lean = decode_output(wp, *args)
            for x, y, z, name in zip(got, ref, ref64, names):
                close(x, y, what=name, exact=z)
            for x, y, name in zip(lean, got, names):
                if name in ("alpha_unflt", "alpha") and x is None:
                    continue
                assert (x is None and y is None) or torch.equal(x, y), "lean " + name
        except AssertionError as exc:
            raise AssertionError(f"{what}: {exc}") from exc


def test_warper_state_dict_names(dev):
    """Buffer names / shapes survive, so a reference checkpoint's warper.* entries load."""
    from waldo_amd.nets import Warper
    names = {k: tuple(v.shape) for k, v in Warper(opt_ns()).state_dict().items()}
    expect = {"src_pts", "tgt_pts", "src_grid", "src_grid_hd", "tgt_grid", "tps_obj.inverse_kernel",
              "tps_obj.pad", "tps_obj.tgt_grid_repr", "invert_obj.kernel", "invert_obj.src_grid",
              "invert_obj.tgt_grid", "invert_obj.x_grid", "invert_obj.y_grid", "invert_obj.perm",
              "tps_bg.inverse_kernel", "tps_bg.pad", "tps_bg.tgt_grid_repr", "invert_bg.kernel",
              "invert_bg.src_grid", "invert_bg.tgt_grid", "invert_bg.x_grid", "invert_bg.y_grid",
              "invert_bg.perm"}
    assert set(names) == expect
    assert names["tps_bg.tgt_grid_repr"] == (16 * 32, 8 + 3) and names["invert_obj.perm"] == (1, 16 * 32)


# ----------------------------------------------------------------------------- A12
def test_wif_forward_golden(dev, golden):
    from waldo_amd.nets import WIF
    g = golden("wif_forward")
    lin = torch.nn.Conv2d(12, 5, 1)
    with torch.no_grad():
        lin.weight.copy_(g["weight"])
        lin.bias.copy_(g["bias"])
    wif = WIF(types.SimpleNamespace(ii_score=True, ii_ab=True), unet=lin).to(dev)
    close(wif(g["vid"].to(dev)), g["out"], what="wif")


def test_wif_fuse_at_recipe_size(dev):
    """waldo_wif_fuse_fwd / _bwd at the shape they exist for (wif.py:37-57 at train_wif.sh's sizes): 40-channel
    raw_output (3 + 20 + 17) of 512 x 1024 frames, Tc = 4 contexts, one predicted frame of two clips, the UNet's 5
    outputs (ii_score + ii_ab) -- forward, grad_vid and grad_net against the oracle's autograd."""
    from waldo_amd import functional as WF
    b, t, tc, c, h, w, co = 2, 1, 4, 40, 512, 1024, 5
    g = torch.Generator().manual_seed(40)
    vid = (torch.rand(b, t, tc, c, h, w, generator=g) * 2 - 1).requires_grad_()
    vid.data[:, :, :, 4] = 5.0 * torch.randn(b, t, tc, h, w, generator=g)   # the layout logit the gate reads (wif.py:53)
    net = torch.randn(b, t, tc, co, h, w, generator=g).requires_grad_()
    ref = WO.wif_fuse(vid, net, ab=True)
    wgt = torch.randn(ref.shape, generator=g)
    (ref * wgt).sum().backward()
    v2, n2 = vid.detach().to(dev).requires_grad_(), net.detach().to(dev).requires_grad_()
    out = WF.wif_fuse(v2, n2, ab=True)
    assert out.shape == (b, t, 3, h, w)
    close(out, ref, what="R size: out")
    (out * wgt.to(dev)).sum().backward()
    close(v2.grad, vid.grad, rel=True, what="R size: grad_vid")
    close(n2.grad, net.grad, rel=True, what="R size: grad_net")
    # the training step's form: raw_output carries no gradient (it was made under no_grad) -- grad_net alone, same bits
    n3 = net.detach().to(dev).requires_grad_()
    (WF.wif_fuse(vid.detach().to(dev), n3, ab=True) * wgt.to(dev)).sum().backward()
    assert torch.equal(n3.grad, n2.grad)


@pytest.mark.parametrize("shape", [(1, 1, 1, 5, 3, 5, 4), (2, 3, 4, 40, 16, 32, 5), (1, 2, 5, 8, 33, 65, 4)])
@pytest.mark.parametrize("ab", [True, False])
def test_wif_fuse_random(dev, shape, ab):
    from waldo_amd import functional as WF
    b, t, tc, c, h, w, co = shape
    torch.manual_seed(c)
    vid = torch.randn(b, t, tc, c, h, w, requires_grad=True)
    net = torch.randn(b, t, tc, co, h, w, requires_grad=True)
    ref = WO.wif_fuse(vid, net, ab=ab)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    v2, n2 = vid.detach().to(dev).requires_grad_(), net.detach().to(dev).requires_grad_()
    out = WF.wif_fuse(v2, n2, ab=ab)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(v2.grad, vid.grad, rel=True, what="grad_vid")
    close(n2.grad, net.grad, rel=True, what="grad_net")

