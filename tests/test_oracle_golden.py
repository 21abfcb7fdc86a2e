"""CPU: the oracle (oracle/wif_oracle.py) against golden vectors produced by the reference."""
import pytest
import torch

from oracle import wif_oracle as O

TOL = 1e-6  # oracle vs reference on the same CPU: same ops, same order -> (near) bit-equal


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, err


@pytest.mark.parametrize("tag", ["k16", "k32"])
def test_tps(golden, tag):
    g = golden(f"tps_{tag}")
    h, w = int(g["h"]), int(g["w"])
    inv, rep = O.tps_init(h, w, g["ctrl"])
    assert torch.equal(inv, g["inverse_kernel"])
    assert torch.equal(rep, g["tgt_grid_repr"])
    pts = g["pts"].clone().requires_grad_()
    grid = O.tps_grid(inv, rep, pts, h, w)
    close(grid, g["grid"])
    (grid * g["wgt"]).sum().backward()
    close(pts.grad, g["grad_pts"], 1e-4)


@pytest.mark.parametrize("delta", [0, 1])
@pytest.mark.parametrize("explicit", [False, True])
def test_grid_sample(golden, delta, explicit):
    g = golden(f"grid_sample_d{delta}")
    x = g["x"].clone().requires_grad_()
    grid = g["grid"].clone().requires_grad_()
    out = O.grid_sample_delta(x, grid, float(g["delta"]), explicit=explicit)
    close(out, g["out"], 2e-6)
    (out * g["wgt"]).sum().backward()
    close(x.grad, g["grad_x"], 1e-5)
    close(grid.grad, g["grad_grid"], 1e-4)


def test_occ_and_composite(golden):
    g = golden("occ_comp")
    score = g["score"].clone().requires_grad_()
    occ = O.compute_occ(score)
    close(occ, g["occ"])
    vid = g["vid"].clone().requires_grad_()
    out, alpha, _ = O.reduce_comp(vid, occ)
    close(out, g["out"], 1e-6)
    close(alpha, g["alpha"], 1e-6)
    ((out * g["w1"]).sum() + (alpha * g["w2"]).sum()).backward()
    close(vid.grad, g["grad_vid"], 1e-5)
    close(score.grad, g["grad_score"], 1e-4)


def test_occ_properties():
    occ = O.compute_occ(torch.randn(2, 3, 5))
    assert torch.all(occ[:, :, 0, :] == 0) and torch.all(occ[:, :, 1:, 0] == 1)
    d = torch.diagonal(occ, dim1=2, dim2=3)
    assert torch.all(d[..., 1:].abs() < 1e-6)
    off = occ[:, :, 1:, 1:] + occ[:, :, 1:, 1:].transpose(2, 3)
    eye = torch.eye(5, dtype=torch.bool)
    assert torch.allclose(off[:, :, ~eye], torch.ones(()), atol=1e-6)


@pytest.mark.parametrize("tag", ["small", "l8", "big_warp"])
@pytest.mark.parametrize("explicit", [False, True])
def test_warp_composite(golden, tag, explicit):
    g = golden(f"warp_composite_{tag}")
    layers = g["layers"].clone().requires_grad_()
    pts = g["pts"].clone().requires_grad_()
    score = g["score"].clone().requires_grad_()
    f, nl, _, h, w = layers.shape
    inv, rep = O.tps_init(h, w, g["ctrl"])
    occ = O.compute_occ(score)[:, 0]
    close(occ, g["occ"])
    rgb, alpha = O.warp_composite(layers, pts, occ, inv, rep, explicit=explicit)
    close(rgb, g["rgb"], 2e-6)
    close(alpha, g["alpha"], 2e-6)
    ((rgb * g["w1"]).sum() + (alpha * g["w2"]).sum()).backward()
    close(layers.grad, g["grad_layers"], 1e-5)
    close(pts.grad, g["grad_pts"], 2e-3 * max(1.0, g["grad_pts"].abs().max().item()))
    close(score.grad, g["grad_score"], 1e-3)


@pytest.mark.parametrize("tag", ["delta1", "delta1_big", "delta_half"])
@pytest.mark.parametrize("explicit", [False, True])
def test_warp_composite_delta(golden, tag, explicit):
    """The "-delta" padding of Warper.obj_to_output / bg_to_output (lvd.py:548,559) on the fused path:
    the reference's F.grid_sample(x + delta) - delta -> reduce_comp, forward and autograd."""
    g = golden(f"warp_composite_{tag}")
    layers = g["layers"].clone().requires_grad_()
    pts = g["pts"].clone().requires_grad_()
    score = g["score"].clone().requires_grad_()
    f, nl, _, h, w = layers.shape
    inv, rep = O.tps_init(h, w, g["ctrl"])
    occ = O.compute_occ(score)[:, 0]
    rgb, alpha = O.warp_composite(layers, pts, occ, inv, rep, explicit=explicit, delta=float(g["delta"]))
    close(rgb, g["rgb"], 2e-6)
    close(alpha, g["alpha"], 2e-6)
    ((rgb * g["w1"]).sum() + (alpha * g["w2"]).sum()).backward()
    close(layers.grad, g["grad_layers"], 1e-5)
    close(pts.grad, g["grad_pts"], 2e-3 * max(1.0, g["grad_pts"].abs().max().item()))
    close(score.grad, g["grad_score"], 1e-3)
    # the padding matters in these cases
    plain, _ = O.warp_composite(g["layers"], g["pts"], g["occ"], inv, rep)
    assert (plain - g["rgb"]).abs().max() > 1e-2


@pytest.mark.parametrize("tag", ["obj", "bg", "obj2"])
def test_inverse_warp(golden, tag):
    g = golden(f"inverse_warp_{tag}")
    sg = g["src_grid"].clone().requires_grad_()
    out = O.inverse_warp(sg, (int(g["ht"]), int(g["wt"])), erode=bool(g["erode"]))
    close(out, g["out"], 1e-6)
    (out * g["wgt"]).sum().backward()
    close(sg.grad, g["grad_src_grid"], 1e-5)


@pytest.mark.parametrize("tag", ["k5", "k7"])
def test_inverse_warp_kernel_size(golden, tag):
    """kernel_size 5 / 7 (warp.py:58-63, 140-146): the reference's own outputs for a wider Gaussian fill window."""
    g = golden(f"inverse_warp_{tag}")
    sg = g["src_grid"].clone().requires_grad_()
    out = O.inverse_warp(sg, (int(g["ht"]), int(g["wt"])), niter=int(g["niter"]), erode=bool(g["erode"]),
                         kernel_size=int(g["kernel_size"]))
    close(out, g["out"], 1e-6)
    (out * g["wgt"]).sum().backward()
    close(sg.grad, g["grad_src_grid"], 1e-5)
    three = O.inverse_warp(sg.detach(), (int(g["ht"]), int(g["wt"])), niter=int(g["niter"]), erode=bool(g["erode"]))
    assert (three - g["out"]).abs().max() > 1e-4  # the window matters in these cases


@pytest.mark.parametrize("tag", ["perm3", "perm4"])
def test_inverse_warp_num_perm(golden, tag):
    """num_perm > 1: the reference's output for its own randperm buffer (stable sort)."""
    g = golden(f"inverse_warp_{tag}")
    sg = g["src_grid"].clone().requires_grad_()
    out = O.inverse_warp(sg, (int(g["ht"]), int(g["wt"])), erode=bool(g["erode"]), perm=g["perm"])
    close(out, g["out"], 1e-6)
    (out * g["wgt"]).sum().backward()
    close(sg.grad, g["grad_src_grid"], 1e-5)
    plain = O.inverse_warp(sg.detach(), (int(g["ht"]), int(g["wt"])), erode=bool(g["erode"]))
    assert (plain - g["out"]).abs().max() > 1e-3  # the order matters in these cases


def test_inverse_warp_identity():
    """InverseWarp(identity) = identity, fully filled (SURVEY.md section 4 invariant)."""
    ident = O.get_grid(12, 20)
    out = O.inverse_warp(ident, (12, 20), erode=False)
    assert (out - ident).abs().max() < 1e-6


def test_tps_identity_fixed_point():
    """Control points at rest => the TPS grid is get_grid (warp.py:38-55 with src == tgt)."""
    ctrl = O.get_grid(4, 4).view(-1, 2)
    inv, rep = O.tps_init(10, 14, ctrl)
    grid = O.tps_grid(inv, rep, ctrl.view(1, 16, 2), 10, 14)
    assert (grid - O.get_grid(10, 14)).abs().max() < 1e-5


def test_warper_chain(golden):
    """Warper.forward -> grid_to_flow[_ctx] -> input_to_output against the reference's outputs."""
    from oracle import warper_oracle as WO
    g = golden("warper_chain")
    cfg = WO.WarperCfg((2, 4), (2, 2), 2, 4, 1, 16, 2, 32, weight_cls=True, min_cls=0.05)
    grid = WO.warper_grids(cfg, g["obj_pose"], g["bg_pose"])
    for a, name in zip(grid, ("tgo", "sgo", "tgb", "sgb")):
        close(a, g[name], 1e-6)
    args = (g["inp"], grid, g["occ"], g["obj_alpha"], g["bg_alpha"], g["cls"], g["ctx_ts"], g["pred_ts"])
    fc = WO.grid_to_flow_ctx(cfg, *args)
    ft = WO.grid_to_flow(cfg, *args)
    for pre, r in (("c_", fc), ("t_", ft)):
        close(r[0], g[pre + "flow"], 2e-6)
        close(r[2], g[pre + "alpha"], 2e-6)
        close(r[3], g[pre + "alpha_ctx"], 2e-6)
        close(r[4], g[pre + "disocc"], 2e-6)
    out, raw = WO.input_to_output(cfg, g["inp"], g["c_alpha_ctx"], g["c_flow"], g["ctx_ts"])
    close(out, g["out"], 2e-6)
    close(raw, g["raw"], 2e-6)


def test_wif_forward(golden):
    from oracle import warper_oracle as WO
    g = golden("wif_forward")
    vid = g["vid"].permute(0, 2, 1, 3, 4, 5)
    b, t, tc, c, h, w = vid.shape
    net = torch.nn.functional.conv2d(vid.reshape(-1, c, h, w), g["weight"], g["bias"])
    close(WO.wif_fuse(vid, net.reshape(b, t, tc, 5, h, w)), g["out"], 1e-6)


# ----------------------------------------------------------------------------- f2: producers
@pytest.mark.parametrize("tag", ["plain", "prior_mask", "remove", "freeze"])
def test_producers_oracle_vs_reference(golden, tag):
    """oracle/producers_oracle.py + wif_oracle.compute_occ against the reference's own
    ImageDecoder.forward / LVD.forward(mode="estimate_alpha_grid_occ") outputs and gradients."""
    from oracle import producers_oracle as PO
    g = golden(f"producers_{tag}")
    raw = g["raw"].clone().requires_grad_()
    score = g["score"].clone().requires_grad_()
    a = PO.decoder_tail(raw, g["circle"], float(g["init_bias"]), int(g["scale_factor"]), True, bool(g["use_prior"]))
    a = PO.alpha_arithmetic(a, g["mask"] if int(g["masked"]) else None, bool(g["remove"]), bool(g["freeze"]))
    a = a.view(g["obj_alpha"].shape)
    occ = O.compute_occ(score)
    assert torch.allclose(a, g["obj_alpha"], atol=1e-6)
    assert torch.allclose(occ, g["occ"], atol=1e-6)
    ((a * g["w1"]).sum() + (occ * g["w2"]).sum()).backward()
    assert torch.allclose(score.grad, g["grad_score"], atol=1e-5)
    graw = raw.grad if raw.grad is not None else torch.zeros_like(raw)
    assert torch.allclose(graw, g["grad_raw"], atol=1e-5)


def test_pose_affine_oracle_vs_independent_formula():
    """flp.py:259-273 restated two ways: the oracle (cat + matmul, as the reference writes it) and a
    float64 einsum of [pts, 1] @ T written from the formula -- the reference cannot run these lines
    in isolation (oracle/producers_oracle.py)."""
    from oracle import producers_oracle as PO
    torch.manual_seed(4)
    r, no, p = 3, 5, 16
    pose = torch.tanh(torch.randn(r, no, 6 + 2 * p))
    mul6 = torch.tensor([0.5, 0.5, 0.5, 0.5, 1.0, 1.0])
    bias6 = torch.tensor([0.25, 0.0, 0.0, 0.5, 0.0, 0.0])
    base = O.get_grid(4, 4).view(p, 2)
    out = PO.pose_affine(pose, mul6, bias6, base, mul_delta=0.3, pts_mul=1.0)
    p64 = pose.double()
    T = (mul6.double() * p64[..., :6] + bias6.double()).view(r, no, 3, 2)
    pts = base.double() + 0.3 * p64[..., 6:].view(r, no, p, 2)
    ref = torch.einsum("rnpa,rnab->rnpb", pts, T[:, :, :2]) + T[:, :, 2:3]
    assert out.shape == (r, no, p, 2)
    assert torch.allclose(out.double(), ref, atol=1e-6)


def _pose_inputs(g):
    """The tensors PoseDecoder.forward hands to flp.py:259-273, rebuilt from the golden's head outputs:
    tanh (flp.py:256) and the optional ``+ last`` of the clip (flp.py:257-259)."""
    fpc = int(g["frames_per_clip"])
    head_obj = g["head_obj"].clone().requires_grad_()
    head_bg = g["head_bg"].clone().requires_grad_()
    obj, bg = head_obj[:, :, :-1].tanh(), head_bg.tanh()
    if int(g["use_last"]):
        obj = obj + g["last_obj"].repeat_interleave(fpc, dim=0)
        bg = bg + g["last_bg"].repeat_interleave(fpc, dim=0)
    return head_obj, head_bg, obj, bg


@pytest.mark.parametrize("tag", ["plain", "last"])
def test_pose_affine_golden(golden, tag):
    """Pins oracle.producers_oracle.pose_affine on the reference's OWN PoseDecoder.forward
    (models/nets/flp.py:252-273, run by oracle/make_golden.py:gen_pose_affine with an empty transformer
    and fixed head outputs): control points of the predicted frames and the gradients back to the heads."""
    from oracle import producers_oracle as PO
    g = golden(f"pose_affine_{tag}")
    head_obj, head_bg, obj, bg = _pose_inputs(g)
    po = PO.pose_affine(obj, g["mul_obj"], g["bias_obj"], g["tgt_pts_obj"].view(-1, 2), float(g["mul_delta_obj"]), 1.0)
    pb = PO.pose_affine(bg, torch.ones(6), g["bias_bg"], g["tgt_pts_bg"].view(-1, 2), 1.0, float(g["bg_mul"]))
    assert torch.allclose(po, g["obj_pose"], atol=1e-6) and torch.allclose(pb, g["bg_pose"], atol=1e-6)
    assert torch.equal(head_obj[:, :, -1].detach(), g["occ_score"])  # flp.py:256: the last channel is the score
    ((po * g["w1"]).sum() + (pb * g["w2"]).sum()).backward()
    assert torch.allclose(head_obj.grad, g["grad_head_obj"], atol=1e-6)
    assert torch.allclose(head_bg.grad, g["grad_head_bg"], atol=1e-6)
