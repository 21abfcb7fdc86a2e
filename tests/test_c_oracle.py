"""The plain-C restatement of the fused path (oracle/wif_oracle.c, double precision) against the
REFERENCE's own outputs and gradients (golden vectors) and against the torch restatement in fp64."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import wif_oracle as O


def _tps_init64(h, w, ctrl):
    """TPSWarp.__init__ (models/modules/warp.py:21-47) in double precision, exact pixel grid."""
    c = ctrl.double()
    n = c.shape[0]
    fk = torch.zeros(n + 3, n + 3, dtype=torch.float64)
    fk[:n, :n] = O.kernel_distance(c, c)
    fk[:n, n] = 1
    fk[n, :n] = 1
    fk[:n, n + 1:] = c
    fk[n + 1:, :n] = c.t()
    xs = -1.0 + (2.0 * torch.arange(w, dtype=torch.float64) + 1.0) / w
    ys = -1.0 + (2.0 * torch.arange(h, dtype=torch.float64) + 1.0) / h
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    g = torch.stack([xx, yy], dim=-1).view(-1, 2)
    rep = torch.cat([O.kernel_distance(g, c), torch.ones(h * w, 1, dtype=torch.float64), g], dim=1)
    return torch.inverse(fk), rep


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("tag", ["small", "l8", "big_warp"])
def test_c_oracle_vs_reference_golden(golden, tag):
    g = {k: v.numpy() for k, v in golden(f"warp_composite_{tag}").items()}
    r = C.fused(g["layers"], g["pts"], g["occ"], g["ctrl"], g["w1"], g["w2"])
    # the reference ran in fp32: agreement to its own rounding noise
    assert np.abs(r["rgb"] - g["rgb"]).max() < 1e-4
    assert np.abs(r["alpha"] - g["alpha"]).max() < 1e-4
    assert rel(r["grad_layers"], g["grad_layers"]) < 2e-4
    assert rel(r["grad_pts"], g["grad_pts"]) < 2e-3
    q = C.fused(g["layers"], g["pts"], g["occ"], g["ctrl"], loss_sq=True)
    assert rel(q["grad_layers"], g["grad_layers_sq"]) < 2e-4
    assert rel(q["grad_pts"], g["grad_pts_sq"]) < 2e-3


@pytest.mark.parametrize("tag", ["delta1", "delta1_big", "delta_half"])
def test_c_oracle_delta_vs_reference_golden(golden, tag):
    """grid_sample(x + delta) - delta (lvd.py:548,559) in the C restatement vs the reference."""
    g = {k: v.numpy() for k, v in golden(f"warp_composite_{tag}").items()}
    r = C.fused(g["layers"], g["pts"], g["occ"], g["ctrl"], g["w1"], g["w2"], delta=float(g["delta"]))
    # the reference's fp32 grid under a sigma = 0.5 warp of white noise moves values by ~1e-4
    assert np.abs(r["rgb"] - g["rgb"]).max() < 2e-4
    assert np.abs(r["alpha"] - g["alpha"]).max() < 2e-4
    assert rel(r["grad_layers"], g["grad_layers"]) < 2e-4
    assert rel(r["grad_pts"], g["grad_pts"]) < 2e-3


@pytest.mark.parametrize("shape", [(2, 3, 9, 13), (1, 8, 16, 24), (2, 1, 5, 7)])
def test_c_oracle_vs_torch_fp64(shape):
    f, nl, h, w = shape
    layers, pts, occ, _, _ = O.make_synthetic(f, nl, h, w, seed=3, sigma=0.2)
    ctrl = O.get_grid(4, 4).view(-1, 2)
    torch.manual_seed(1)
    w1, w2 = torch.randn(f, 3, h, w), torch.randn(f, nl, h, w)
    inv, rep = _tps_init64(h, w, ctrl)
    l64, p64, o64 = layers.double().requires_grad_(), pts.double().requires_grad_(), occ.double().requires_grad_()
    rgb, alpha = O.warp_composite(l64, p64, o64, inv, rep)
    ((rgb * w1.double()).sum() + (alpha * w2.double()).sum()).backward()
    r = C.fused(layers, pts, occ, ctrl, w1, w2)
    # the C side takes fp32 control points / layers (as the ABI does); everything downstream is double
    assert np.abs(r["rgb"] - rgb.detach().numpy()).max() < 1e-9
    assert np.abs(r["alpha"] - alpha.detach().numpy()).max() < 1e-9
    assert rel(r["grad_layers"], l64.grad.numpy()) < 1e-9
    assert rel(r["grad_pts"], p64.grad.numpy()) < 1e-8
    assert rel(r["grad_occ"], o64.grad.numpy()) < 1e-9
