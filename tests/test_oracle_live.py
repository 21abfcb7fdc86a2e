"""CPU, build container only: the oracle against the REFERENCE ITSELF, imported from
/root/reference (skipped where the tree is absent, e.g. on the GPU box)."""
import warnings

import pytest
import torch

from oracle import ref_import as R
from oracle import warper_oracle as WO
from oracle import wif_oracle as O

pytestmark = [pytest.mark.live_ref,
              pytest.mark.skipif(not R.available(), reason="reference tree not present")]
warnings.filterwarnings("ignore", message="Default grid_sample")


@pytest.fixture(scope="module")
def ns():
    return R.load()


def close(a, b, tol=1e-6):
    assert (a is None) == (b is None)
    if a is None:
        return
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol


def test_buffers_bit_equal(ns):
    for h, w in [(4, 4), (8, 16), (128, 256), (3, 7)]:
        assert torch.equal(ns.get_grid(h, w), O.get_grid(h, w))
    assert torch.equal(ns.get_gaussian_kernel(3), O.get_gaussian_kernel(3))
    ctrl = ns.get_grid(4, 4).view(-1, 2)
    ref = ns.TPSWarp(16, 32, ctrl)
    inv, rep = O.tps_init(16, 32, ctrl)
    assert torch.equal(ref.inverse_kernel, inv) and torch.equal(ref.tgt_grid_repr, rep)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_inverse_warp_random(ns, seed):
    torch.manual_seed(seed)
    ctrl = ns.get_grid(4, 4).view(-1, 2)
    hs, ws, ht, wt = [(8, 8, 16, 32), (16, 32, 16, 32), (12, 20, 24, 40)][seed]
    inv = ns.InverseWarp(hs, ws, ht, wt)
    sg = ns.TPSWarp(hs, ws, ctrl)(ctrl.view(1, 16, 2) * 0.7 + 0.1 * torch.randn(3, 16, 2))
    for erode in (True, False):
        with R.stable_sort():
            ref = inv(sg, erode=erode)
        assert torch.equal(ref, O.inverse_warp(sg, (ht, wt), erode=erode))


@pytest.mark.parametrize("over", [dict(), dict(weight_cls=True, min_cls=0.1), dict(load_dim=0),
                                  dict(allow_ghost=True, include_self=True), dict(no_filter=True)])
def test_warper_against_reference(ns, over):
    torch.manual_seed(0)
    opt = R.warper_opt(num_obj=3, **over)
    ref = ns.Warper(opt)
    cfg = WO.WarperCfg.from_opt(opt)
    b, t, no, nl = 2, 4, 3, 5
    obj_pose = ns.get_grid(2, 2).view(1, 1, 1, 4, 2) * 0.5 + 0.15 * torch.randn(b, t, no, 4, 2)
    bg_pose = ns.get_grid(2, 4).view(1, 1, 1, 8, 2) + 0.05 * torch.randn(b, t, 1, 8, 2)
    with R.stable_sort():
        grid = ref(obj_pose, bg_pose)
    for a, c in zip(grid, WO.warper_grids(cfg, obj_pose, bg_pose)):
        close(a, c)
    (hd, wd), (h, w), (ho, wo) = cfg.src_shape_hd, cfg.src_shape, cfg.tgt_shape
    inp = torch.randn(b, t, 3 + nl, hd, wd)
    occ = O.compute_occ(torch.randn(b, t, no))
    obj_alpha = torch.rand(b, no, 1, ho, wo) * 2 - 1
    bg_alpha = torch.ones(b, 1, h, w)
    cls = torch.rand(b, no, nl).softmax(-1)
    ctx_ts = torch.tensor([[[0, 1], [1, 0]], [[1, 1], [0, 0]]])
    pred_ts = torch.tensor([2, 3])
    for name in ("grid_to_flow_ctx", "grid_to_flow"):
        for c in ((cls,) if opt.weight_cls else (cls, None)):
            r = getattr(ref, name)(inp, grid, occ, obj_alpha, bg_alpha, c, ctx_ts, pred_ts)
            o = getattr(WO, name)(cfg, inp, grid, occ, obj_alpha, bg_alpha, c, ctx_ts, pred_ts)
            for x, y in zip(r, o):
                close(x, y, 2e-6)
    ro = ref.input_to_output(inp, r[3], r[0], ctx_ts)
    oo = WO.input_to_output(cfg, inp, r[3], r[0], ctx_ts)
    close(ro[0], oo[0])
    close(ro[1], oo[1])
    if cfg.include_self:  # needs Tp == T
        ctx2 = torch.roll(torch.arange(t), 1).view(1, 1, t).expand(b, 1, t)
        pr2 = torch.arange(t)
        r2 = ref.grid_to_flow(inp, grid, occ, obj_alpha, bg_alpha, cls, ctx2, pr2)  # LVD training path
        ro = ref.input_to_output(inp, r2[3], r2[0], ctx2)
        oo = WO.input_to_output(cfg, inp, r2[3], r2[0], ctx2)
        close(ro[0], oo[0])
        close(ro[1], oo[1])
    for x, y in zip(ref.alpha_to_alpha(obj_alpha, bg_alpha, grid, occ),
                    WO.alpha_to_alpha(cfg, obj_alpha, bg_alpha, grid, occ)):
        close(x, y)
    x5 = torch.randn(b, t, 4, h, w)
    for x, y in zip(ref.layer_from_input(x5, grid), WO.layer_from_input(cfg, x5, grid)):
        close(x, y)
    close(ref.grid_to_bg_flow_from_ref_to_pred(grid, 2, 1), WO.grid_to_bg_flow_from_ref_to_pred(cfg, grid, 2, 1))
    close(ref.grid_to_obj_flow_from_ref_to_pred(grid, 2, 1, 2),
          WO.grid_to_obj_flow_from_ref_to_pred(cfg, grid, 2, 1, 2))
    close(ref.grid_to_bg_flow_from_ctx_to_ref(grid, 2, 3), WO.grid_to_bg_flow_from_ctx_to_ref(cfg, grid, 2, 3))


def test_wif_forward_against_reference(ns):
    """WIF.forward with a stand-in UNet (the convolutions are out of scope): the fusion arithmetic
    around it, including the input-channel-4 blending weight (wif.py:53)."""
    import types
    torch.manual_seed(0)
    b, tc, t, c, h, w = 2, 3, 2, 12, 8, 16
    vid = torch.randn(b, tc, t, c, h, w)
    wif = ns.WIF.__new__(ns.WIF)
    torch.nn.Module.__init__(wif)
    wif.score, wif.ab = True, True
    lin = torch.nn.Conv2d(c, 5, 1)
    wif.unet = lin
    ref = ns.WIF.forward(wif, vid)
    v = vid.permute(0, 2, 1, 3, 4, 5)
    out = lin(v.reshape(b * t * tc, c, h, w)).reshape(b, t, tc, 5, h, w)
    close(ref, WO.wif_fuse(v, out), 1e-6)


def test_expand_equals_reference_bit_for_bit(ns):
    """tools/utils.py:300-323 against the product's restatement (waldo_amd/tools/utils.py): hard and
    soft masks, every `dir`, several rounds -- and the product must leave its argument alone."""
    from waldo_amd.tools.utils import expand
    torch.manual_seed(5)
    hard = (torch.rand(2, 3, 17, 23) > 0.9).float()
    soft = torch.rand(2, 1, 19, 21) * (torch.rand(2, 1, 19, 21) > 0.8)
    for d in (None, "south", "north", "east", "west"):
        for num in (1, 2, 5):
            keep = hard.clone()
            assert torch.equal(expand(hard, num, dir=d), ns.expand(hard.clone(), num, dir=d))
            assert torch.equal(hard, keep)
            keep = soft.clone()
            assert torch.equal(expand(soft, num, dir=d, soft=True), ns.expand(soft.clone(), num, dir=d, soft=True))
            assert torch.equal(soft, keep)
    b = hard.bool()
    keep = b.clone()
    expand(b, 3)
    assert torch.equal(b, keep)
