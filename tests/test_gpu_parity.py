"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerance: BASELINE.json north_star -- outputs within 1e-4 fp32 of the reference on identical
inputs.  Gradients are sums of many such terms; they are checked to 1e-4 RELATIVE to the
largest gradient magnitude of the tensor (float atomics reorder the sums).
"""
import pytest
import torch

from oracle import wif_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def close(a, b, tol=TOL, rel=False, what=""):
    a = a.detach().cpu()
    b = b.detach().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    scale = max(1.0, b.abs().max().item()) if rel else 1.0
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e}"


def test_native_library_is_loaded(dev):
    from waldo_amd import _lib
    lib = _lib.load()
    assert lib.waldo_version() >= 1000
    maps = open("/proc/self/maps").read()
    assert "libwaldo_hip.so" in maps


# ----------------------------------------------------------------------------- A2
@pytest.mark.parametrize("tag", ["k16", "k32"])
def test_tps_golden(dev, golden, tag):
    import waldo_amd
    g = golden(f"tps_{tag}")
    h, w = int(g["h"]), int(g["w"])
    mod = waldo_amd.TPSWarp(h, w, g["ctrl"]).to(dev)
    assert torch.equal(mod.inverse_kernel.cpu(), g["inverse_kernel"])
    assert torch.equal(mod.tgt_grid_repr.cpu(), g["tgt_grid_repr"])
    pts = g["pts"].to(dev).requires_grad_()
    grid = mod(pts)
    close(grid, g["grid"], what="grid")
    (grid * g["wgt"].to(dev)).sum().backward()
    close(pts.grad, g["grad_pts"], rel=True, what="grad_pts")


def test_tps_bg_sized(dev):
    """Background-sized TPS of the real recipe: N = 128 control points (8x16), K3 = 131."""
    import waldo_amd
    ctrl = O.get_grid(8, 16).view(-1, 2)
    h, w = 32, 64
    mod = waldo_amd.TPSWarp(h, w, ctrl).to(dev)
    inv, rep = O.tps_init(h, w, ctrl)
    torch.manual_seed(0)
    pts = (ctrl.view(1, -1, 2) + 0.02 * torch.randn(11, 128, 2)).requires_grad_()
    ref = O.tps_grid(inv, rep, pts, h, w)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    p2 = pts.detach().to(dev).requires_grad_()
    out = mod(p2)
    close(out, ref, what="grid")
    (out * wgt.to(dev)).sum().backward()
    close(p2.grad, pts.grad, tol=2e-4, rel=True, what="grad_pts")


def test_tps_identity(dev):
    import waldo_amd
    ctrl = O.get_grid(4, 4).view(-1, 2)
    mod = waldo_amd.TPSWarp(24, 40, ctrl).to(dev)
    out = mod(ctrl.view(1, 16, 2).to(dev))
    close(out, O.get_grid(24, 40), 1e-5)


# ----------------------------------------------------------------------------- A4/A5
@pytest.mark.parametrize("delta", [0, 1])
def test_grid_sample_golden(dev, golden, delta):
    from waldo_amd import functional as WF
    g = golden(f"grid_sample_d{delta}")
    x = g["x"].to(dev).requires_grad_()
    grid = g["grid"].to(dev).requires_grad_()
    out = WF.grid_sample(x, grid, delta=float(g["delta"]))
    close(out, g["out"], what="out")
    (out * g["wgt"].to(dev)).sum().backward()
    close(x.grad, g["grad_x"], rel=True, what="grad_x")
    close(grid.grad, g["grad_grid"], rel=True, what="grad_grid")


@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 1, 1), (2, 3, 7, 5, 300, 1), (4, 2, 64, 64, 128, 256),
                                   (6, 23, 16, 32, 16, 32)])
def test_grid_sample_random(dev, shape):
    from waldo_amd import functional as WF
    n, c, hi, wi, ho, wo = shape
    torch.manual_seed(n * 7 + c)
    x = torch.randn(n, c, hi, wi, requires_grad=True)
    grid = (torch.rand(n, ho, wo, 2) * 2.4 - 1.2).requires_grad_()
    ref = O.grid_sample_delta(x, grid, 0.5)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    x2, g2 = x.detach().to(dev).requires_grad_(), grid.detach().to(dev).requires_grad_()
    out = WF.grid_sample(x2, g2, delta=0.5)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(x2.grad, x.grad, rel=True, what="grad_x")
    close(g2.grad, grid.grad, rel=True, what="grad_grid")


def test_grid_sample_broadcast(dev):
    """Input expanded over time as in Warper.obj_to_output (lvd.py:544): (B,1,No,..) -> (B,T,No,..)"""
    from waldo_amd import functional as WF
    b, t, no, c, ho, wo, h, w = 2, 3, 4, 2, 8, 8, 12, 10
    torch.manual_seed(3)
    obj = torch.randn(b, no, c, ho, wo, requires_grad=True)
    grid = (torch.rand(b * t * no, h, w, 2) * 2.2 - 1.1)
    exp = obj.view(b, 1, no, c, ho, wo).expand(-1, t, -1, -1, -1, -1).reshape(b * t * no, c, ho, wo)
    ref = O.grid_sample_delta(exp, grid, 1.0)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    o2 = obj.detach().to(dev).requires_grad_()
    out = WF.grid_sample(o2.view(b * no, c, ho, wo), grid.to(dev), delta=1.0, broadcast=(t * no, no))
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(o2.grad, obj.grad, rel=True, what="grad_obj")


def test_grid_sample_empty(dev):
    from waldo_amd import functional as WF
    out = WF.grid_sample(torch.zeros(0, 3, 4, 4, device=dev), torch.zeros(0, 5, 5, 2, device=dev))
    assert out.shape == (0, 3, 5, 5)


# ----------------------------------------------------------------------------- A6
@pytest.mark.parametrize("nl", [1, 2, 5, 8, 9, 12, 17, 21, 32])
def test_occ_composite(dev, nl):
    from waldo_amd import functional as WF
    torch.manual_seed(nl)
    m, h, w, div = 6, 9, 31, 3
    alpha = torch.rand(m, nl, h, w, requires_grad=True)
    occ = torch.rand(m // div, nl, nl, requires_grad=True)
    occ_full = occ.repeat_interleave(div, dim=0)
    ref = O.occlusion_product(alpha, occ_full)
    wgt = torch.randn(ref.shape)
    (ref * wgt).sum().backward()
    a2, o2 = alpha.detach().to(dev).requires_grad_(), occ.detach().to(dev).requires_grad_()
    out = WF.occ_composite(a2, o2, occ_div=div)
    close(out, ref, what="out")
    (out * wgt.to(dev)).sum().backward()
    close(a2.grad, alpha.grad, rel=True, what="grad_alpha")
    close(o2.grad, occ.grad, rel=True, what="grad_occ")


def test_occ_composite_golden(dev, golden):
    from waldo_amd import functional as WF
    g = golden("occ_comp")
    v = (g["vid"] + 1) / 2
    alpha = v[:, :, :, -1].clone()
    alpha[:, :, 0] = 1.0
    b, t, nl, h, w = alpha.shape
    out = WF.occ_composite(alpha.view(b * t, nl, h, w).to(dev), g["occ"].view(b * t, nl, nl).to(dev))
    close(out.view(b, t, nl, h, w) * 2 - 1, g["alpha"], what="alpha")


# ----------------------------------------------------------------------------- fused path
def _run_fused(dev, layers, pts, occ, ctrl, w1, w2):
    from waldo_amd import functional as WF
    import waldo_amd
    f, nl, _, h, w = layers.shape
    tps = waldo_amd.TPSWarp(h, w, ctrl).to(dev)
    l2 = layers.detach().to(dev).requires_grad_()
    p2 = pts.detach().to(dev).requires_grad_()
    o2 = occ.detach().to(dev).requires_grad_()
    rgb, alpha = WF.warp_composite(l2, p2, o2, tps.inverse_kernel, tps.basis_t, return_alpha=True)
    ((rgb * w1.to(dev)).sum() + (alpha * w2.to(dev)).sum()).backward()
    return rgb, alpha, l2.grad, p2.grad, o2.grad


@pytest.mark.parametrize("tag", ["small", "l8", "big_warp"])
def test_warp_composite_golden(dev, golden, tag):
    g = golden(f"warp_composite_{tag}")
    occ = g["occ"].clone().requires_grad_()
    rgb, alpha, gl, gp, go = _run_fused(dev, g["layers"], g["pts"], occ, g["ctrl"], g["w1"], g["w2"])
    close(rgb, g["rgb"], what="rgb")
    close(alpha, g["alpha"], what="alpha")
    close(gl, g["grad_layers"], rel=True, what="grad_layers")
    close(gp, g["grad_pts"], tol=3e-4, rel=True, what="grad_pts")
    # d loss / d score through compute_occ, chained on the CPU from the kernel's d loss / d occ
    score = g["score"].clone().requires_grad_()
    O.compute_occ(score)[:, 0].backward(go.cpu())
    close(score.grad, g["grad_score"], tol=3e-4, rel=True, what="grad_score")


def test_warp_composite_golden_bench_loss(dev, golden):
    """The benchmark's loss (rgb.square().mean()) and the rgb-only output path."""
    from waldo_amd import functional as WF
    import waldo_amd
    g = golden("warp_composite_l8")
    f, nl, _, h, w = g["layers"].shape
    tps = waldo_amd.TPSWarp(h, w, g["ctrl"]).to(dev)
    l2 = g["layers"].to(dev).requires_grad_()
    p2 = g["pts"].to(dev).requires_grad_()
    rgb = WF.warp_composite(l2, p2, g["occ"].to(dev), tps.inverse_kernel, tps.basis_t)
    close(rgb, g["rgb"], what="rgb")
    rgb.square().mean().backward()
    close(l2.grad, g["grad_layers_sq"], tol=1e-4 * g["grad_layers_sq"].abs().max().item(), what="gl")
    close(p2.grad, g["grad_pts_sq"], tol=3e-4 * g["grad_pts_sq"].abs().max().item(), what="gp")


@pytest.mark.parametrize("cfg", [
    dict(f=1, nl=1, h=5, w=7), dict(f=2, nl=2, h=16, w=16), dict(f=3, nl=4, h=17, w=33),
    dict(f=2, nl=5, h=24, w=40), dict(f=5, nl=8, h=32, w=48), dict(f=2, nl=9, h=20, w=20),
    dict(f=2, nl=12, h=16, w=24), dict(f=1, nl=17, h=16, w=32), dict(f=1, nl=20, h=12, w=12),
    dict(f=1, nl=32, h=8, w=16), dict(f=2, nl=8, h=16, w=16, k=3), dict(f=2, nl=3, h=16, w=16, k=5),
    dict(f=40, nl=4, h=16, w=16),
])
def test_warp_composite_random(dev, cfg):
    f, nl, h, w = cfg["f"], cfg["nl"], cfg["h"], cfg["w"]
    k = cfg.get("k", 4)
    torch.manual_seed(f * 100 + nl)
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, k_side=k, seed=nl, sigma=0.1) \
        if nl > 1 else (None,) * 5
    ctrl = O.get_grid(k, k).view(-1, 2)
    if nl == 1:
        layers = torch.rand(f, 1, 4, h, w) * 2 - 1
        pts = ctrl.view(1, -1, 2) + 0.1 * torch.randn(f, k * k, 2)
        occ = torch.zeros(f, 1, 1)
        inv, rep = O.tps_init(h, w, ctrl)
    layers = layers.clone().requires_grad_()
    pts = pts.clone().requires_grad_()
    occ = occ.clone().requires_grad_()
    rgb, alpha = O.warp_composite(layers, pts, occ, inv, rep)
    w1, w2 = torch.randn(rgb.shape), torch.randn(alpha.shape)
    ((rgb * w1).sum() + (alpha * w2).sum()).backward()
    rgb2, alpha2, gl, gp, go = _run_fused(dev, layers, pts, occ, ctrl, w1, w2)
    close(rgb2, rgb, what="rgb")
    close(alpha2, alpha, what="alpha")
    close(gl, layers.grad, rel=True, what="grad_layers")
    close(gp, pts.grad, tol=3e-4, rel=True, what="grad_pts")
    close(go, occ.grad, tol=3e-4, rel=True, what="grad_occ")


def test_warp_composite_empty(dev):
    from waldo_amd import functional as WF
    import waldo_amd
    tps = waldo_amd.TPSWarp(8, 8, O.get_grid(4, 4).view(-1, 2)).to(dev)
    rgb = WF.warp_composite(torch.zeros(0, 3, 4, 8, 8, device=dev), torch.zeros(0, 16, 2, device=dev),
                            torch.zeros(0, 3, 3, device=dev), tps.inverse_kernel, tps.basis_t)
    assert rgb.shape == (0, 3, 8, 8)


def test_warp_composite_rejects_bad_shapes(dev):
    from waldo_amd import functional as WF
    from waldo_amd._lib import WaldoHipError
    import waldo_amd
    tps = waldo_amd.TPSWarp(8, 8, O.get_grid(4, 4).view(-1, 2)).to(dev)
    with pytest.raises(WaldoHipError):
        WF.warp_composite(torch.zeros(1, 33, 4, 8, 8, device=dev), torch.zeros(33, 16, 2, device=dev),
                          torch.zeros(1, 33, 33, device=dev), tps.inverse_kernel, tps.basis_t)


# ----------------------------------------------------------------------------- full size
@pytest.mark.parametrize("h,w,nl", [(128, 128, 8), (256, 512, 8)])
def test_warp_composite_full_size(dev, h, w, nl):
    """BASELINE.json sizes: (i) oracle on the same seeded inputs for a few frames, (ii) the
    size-independent properties: identity control points => plain composite of the unwarped
    layers; layer 0 alone => its own rgb."""
    from waldo_amd import functional as WF
    import waldo_amd
    f = 3
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=11)
    ctrl = O.get_grid(4, 4).view(-1, 2)
    tps = waldo_amd.TPSWarp(h, w, ctrl).to(dev)
    ld, pd, od = layers.to(dev), pts.to(dev), occ.to(dev)
    rgb, alpha = WF.warp_composite(ld, pd, od, tps.inverse_kernel, tps.basis_t, return_alpha=True)
    ref_rgb, ref_alpha = O.warp_composite(layers, pts, occ, inv, rep)
    close(rgb, ref_rgb, what="rgb")
    close(alpha, ref_alpha, what="alpha")
    # (ii) identity warp
    ident = ctrl.view(1, 16, 2).expand(f * nl, -1, -1).contiguous().to(dev)
    rgb_i = WF.warp_composite(ld, ident, od, tps.inverse_kernel, tps.basis_t)
    plain, _, _ = O.reduce_comp(layers.unsqueeze(1), occ.unsqueeze(1))
    close(rgb_i, plain[:, 0], what="identity warp")
    # (iii) all object alphas at -1 (transparent) => output is layer 0's rgb
    l0 = layers.clone()
    l0[:, 1:, 3] = -1.0
    rgb_0 = WF.warp_composite(l0.to(dev), ident, od, tps.inverse_kernel, tps.basis_t)
    close(rgb_0, layers[:, 0, :3], what="background only")


def test_warp_composite_bwd_full_size(dev):
    """fwd+bwd at the headline size against the oracle's autograd (2 frames)."""
    from waldo_amd import functional as WF
    import waldo_amd
    f, nl, h, w = 2, 8, 256, 512
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=5)
    layers.requires_grad_()
    pts.requires_grad_()
    O.warp_composite(layers, pts, occ, inv, rep)[0].square().mean().backward()
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    l2 = layers.detach().to(dev).requires_grad_()
    p2 = pts.detach().to(dev).requires_grad_()
    WF.warp_composite(l2, p2, occ.to(dev), tps.inverse_kernel, tps.basis_t).square().mean().backward()
    close(l2.grad, layers.grad, tol=1e-4 * layers.grad.abs().max().item(), what="grad_layers")
    close(p2.grad, pts.grad, tol=1e-3 * pts.grad.abs().max().item(), what="grad_pts")
