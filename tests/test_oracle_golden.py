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


@pytest.mark.parametrize("tag", ["obj", "bg", "obj2"])
def test_inverse_warp(golden, tag):
    g = golden(f"inverse_warp_{tag}")
    sg = g["src_grid"].clone().requires_grad_()
    out = O.inverse_warp(sg, (int(g["ht"]), int(g["wt"])), erode=bool(g["erode"]))
    close(out, g["out"], 1e-6)
    (out * g["wgt"]).sum().backward()
    close(sg.grad, g["grad_src_grid"], 1e-5)


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
