"""BASELINE config C1 (demo_cityscapes clip, 128x128, 4 layers): the plumbing driver
waldo_amd/tools/demo.py -- the call order of Synthesizer.predict (models/synthesizer.py:434-480) on
the committed six-frame clip -- on the HIP path against the SAME chain restated with the CPU oracle,
stage by stage.  Grid inversion rounds positions to cells, so fp32-level differences upstream flip
a few cells: the inverted grids alone are compared with a robust measure (share of
deviating pixels + mean error); everything DOWNSTREAM of them is compared on identical grids -- the restated chain
is fed the grids the HIP path produced -- and held to the chain bound of tests/parity.py (1e-4 + the measured
fp32 noise of the restatement itself), like every other chain test."""
import os

import pytest
import torch

from oracle import producers_oracle as PO
from oracle import warper_oracle as WO
from oracle import wif_oracle as O
from parity import close  # tests/parity.py

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CLIP = os.path.join(HERE, "golden", "demo_clip", "leftImg8bit_sequence_512", "val", "munster")


def robust(a, b, what, tol=2e-3, share=0.02, mean_tol=1e-3):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    d = (a - b).abs()
    bad = (d > tol).double().mean().item()
    print(f"[demo] {what}: max {d.max().item():.3e} mean {d.mean().item():.3e} share>{tol:g}: {bad:.4f}")
    assert bad <= share and d.mean().item() <= mean_tol, (what, bad, d.mean().item())


def oracle_predict(opt, vid, lyt, net, ctx_len, grid=None, dtype=torch.float32):
    """demo.predict restated with oracle functions only (CPU), in ``dtype``; ``grid``: the four grids to use
    instead of the restatement's own (so that both sides of a comparison invert nothing differently)."""
    given = grid
    vid, lyt = vid.to(dtype), lyt.to(dtype)
    net = {k: v.to(dtype) for k, v in net.items()}
    from waldo_amd.tools import demo
    cfg = WO.WarperCfg.from_opt(opt)
    b, t = vid.shape[:2]
    no = opt.num_obj
    lo, lb = opt.obj_shape[0] * opt.obj_shape[1], opt.latent_shape[0] * opt.latent_shape[1]
    buf = {k: v.to(dtype) for k, v in demo.pose_buffers(opt, "cpu").items()}
    mask = demo.obj_alpha_mask(opt, "cpu").to(dtype)
    bg_alpha = torch.ones(b, 1, *cfg.src_shape, dtype=dtype)

    def stage(pop, pbp, score, nt):
        obj_pose = PO.pose_affine(pop, buf["mul_obj"], buf["bias_obj"], buf["tgt_pts_obj"].view(lo, 2))
        bg_pose = PO.pose_affine(pbp, torch.ones(6, dtype=dtype), buf["bias_bg"], buf["tgt_pts_bg"].view(lb, 2))
        oa = PO.decoder_tail(net["raw"], None, 0.0, opt.scale_factor)
        oa = PO.alpha_arithmetic(oa.view(b, no, 1, *oa.shape[-2:]), mask)
        if given is not None:
            grid = [g.to(dtype) for g in given]
        else:
            grid = WO.warper_grids(cfg, obj_pose.view(b, nt, no, lo, 2), bg_pose.view(b, nt, 1, lb, 2))
        return obj_pose, oa, O.compute_occ(score), grid

    out = {}
    obj_pose, oa, occ, grid = stage(net["pred_obj_pose"], net["pred_bg_pose"], net["occ_score"], t)
    out["obj_pose"], out["obj_alpha"], out["occ"], out["grid"] = obj_pose, oa, occ, grid
    ctx_ts = torch.arange(ctx_len).view(1, -1, 1).expand(b, -1, t).contiguous()
    inp = torch.cat([vid, lyt], 2)
    dec = WO.decode_output(cfg, inp, grid, occ, oa, bg_alpha, net["cls"], ctx_ts, torch.arange(t), True, False)
    out["rec_vid"] = dec[0][:, :, :3]
    raw = dec[5]
    vt = raw.permute(0, 2, 1, 3, 4, 5)
    out["inp_rec_vid"] = WO.wif_fuse(vt, torch.zeros(*vt.shape[:3], 4, *vt.shape[-2:], dtype=dtype), ab=True)
    tp = t - ctx_len
    oa2, occ2, grid2 = oa, occ, grid  # full-length "predicted" pose sequences: the same synthetic poses
    ctx_ts = torch.arange(ctx_len).view(1, -1, 1).expand(b, -1, tp).contiguous()
    dec = WO.decode_output(cfg, inp, grid2, occ2, oa2, bg_alpha, net["cls"], ctx_ts, torch.arange(ctx_len, t), True, False)
    out["pred_vid"] = torch.cat([vid[:, :ctx_len], dec[0][:, :, :3]], 1)
    vt = dec[5].permute(0, 2, 1, 3, 4, 5)
    out["inp_pred_vid"] = torch.cat([vid[:, :ctx_len], WO.wif_fuse(vt, torch.zeros(*vt.shape[:3], 4, *vt.shape[-2:], dtype=dtype), True)], 1)
    return out


def test_demo_clip_predict_chain(dev, tmp_path):
    from waldo_amd.nets import flp
    from waldo_amd.nets.lvd import Warper
    from waldo_amd.nets.wif import WIF
    from waldo_amd.tools import demo, io as wio
    opt = demo.demo_opt(dim=128, aspect_ratio=1.0, num_obj=3, num_lyt=20)
    clip = wio.load_clip(CLIP, (128, 128), 20, max_frames=6)
    vid, lyt = clip["vid"].unsqueeze(0), clip["lyt"].unsqueeze(0)
    ctx_len = 4
    net = demo.synthetic_network_outputs(opt, 1, 6, ctx_len, seed=0)
    ref = oracle_predict(opt, vid, lyt, net, ctx_len)

    warper = Warper(opt).to(dev)
    wif = WIF(opt, unet=demo.UniformFusionUNet()).to(dev)
    netd = {k: v.to(dev) for k, v in net.items()}
    got = demo.predict(opt, warper, wif, vid.to(dev), lyt.to(dev), netd, ctx_len)
    # the stages in front of the grid inversion: tight
    buf = demo.pose_buffers(opt, dev)
    pts = flp.obj_pose_to_points(netd["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
    assert (pts.cpu() - ref["obj_pose"]).abs().max() <= 1e-6
    grid = warper(pts.view(1, 6, 3, 4, 2), flp.bg_pose_to_points(netd["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"]).view(1, 6, 1, 16, 2))
    assert (grid[0].cpu() - ref["grid"][0]).abs().max() <= 1e-4
    robust(grid[1], ref["grid"][1], "inverted object grids", tol=1e-3, share=0.02, mean_tol=5e-2)
    robust(grid[3], ref["grid"][3], "inverted background grid", tol=1e-3, share=0.02, mean_tol=5e-2)
    # the frames predict produces, against the chain restated ON THE SAME GRIDS (fp32 and fp64): the chain bound
    hip_grid = [g.cpu() for g in grid]
    same32 = oracle_predict(opt, vid, lyt, net, ctx_len, grid=hip_grid)
    same64 = oracle_predict(opt, vid, lyt, net, ctx_len, grid=hip_grid, dtype=torch.float64)
    for key in ("rec_vid", "inp_rec_vid", "pred_vid", "inp_pred_vid"):
        assert torch.isfinite(got[key]).all()
        close(got[key], same32[key], what=key, exact=same64[key])
        robust(got[key], ref[key], key)  # and, loosely, against the chain that inverted its own grids
    assert got["rec_vid"].shape == (1, 6, 3, 128, 128) and got["pred_vid"].shape == (1, 6, 3, 128, 128)
    assert torch.equal(got["pred_vid"][:, :ctx_len].cpu(), vid[:, :ctx_len])
    # reconstruction from 4 context frames of a real clip: not a blank image
    assert got["inp_rec_vid"].std() > 0.1
    # and the command-line driver writes its results
    res = demo.run(CLIP, str(tmp_path / "out"), frames=6, ctx_len=4)
    names = set(os.listdir(tmp_path / "out"))
    assert {"rec_vid.gif", "inp_pred_vid.gif", "pred_vid_last.png", "pred_flow_last.flo"} <= names
    assert torch.equal(res["rec_vid"], got["rec_vid"])


def test_kitti_size_predict_chain(dev):
    """The same stage-by-stage comparison for ONE clip of BASELINE config 4 at its own size -- the KITTI recipe of
    waldo_amd/tools/pipeline.py (256 x 832 frames over 128 x 416 layers, 8 layers, 9 frames, 4 contexts), exactly what
    ``bench.py --config C4 --pipeline`` times: ``demo.predict`` (merged decodes, raw slots, occupancy map, staged frame warp)
    against the chain restated with the CPU oracle on the grids the HIP path produced, fp32 and fp64, through ``close``;
    the inverted grids themselves with the robust measure."""
    from waldo_amd.nets import flp
    from waldo_amd.tools import demo
    from waldo_amd.tools.pipeline import RECIPES, Pipeline
    pipe = Pipeline("C4", 1, dev, seed=12)
    opt, ctx_len, t = pipe.opt, pipe.ctx_len, pipe.frames
    vid, lyt = pipe.vid.cpu(), pipe.lyt.cpu()
    net = {k: v.cpu() for k, v in pipe.net.items()}
    with torch.no_grad():
        got = pipe()
        buf = demo.pose_buffers(opt, dev)
        no, lo = opt.num_obj, opt.obj_shape[0] * opt.obj_shape[1]
        lb = opt.latent_shape[0] * opt.latent_shape[1]
        pts = flp.obj_pose_to_points(pipe.net["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
        bgp = flp.bg_pose_to_points(pipe.net["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"])
        grid = pipe.warper(pts.view(1, t, no, lo, 2), bgp.view(1, t, 1, lb, 2))
        hip_grid = [g.cpu() for g in grid]
        # the restatement's own grids (the inversion is a discontinuous function of them: the robust measure)
        cfg = WO.WarperCfg.from_opt(opt)
        own = WO.warper_grids(cfg, pts.cpu().view(1, t, no, lo, 2), bgp.cpu().view(1, t, 1, lb, 2))
        assert (grid[0].cpu() - own[0]).abs().max() <= 1e-4
        robust(grid[1], own[1], "inverted object grids (KITTI size)", tol=1e-3, share=0.02, mean_tol=5e-2)
        robust(grid[3], own[3], "inverted background grid (KITTI size)", tol=1e-3, share=0.02, mean_tol=5e-2)
        same32 = oracle_predict(opt, vid, lyt, net, ctx_len, grid=hip_grid)
        same64 = oracle_predict(opt, vid, lyt, net, ctx_len, grid=hip_grid, dtype=torch.float64)
    hd, wd = RECIPES["C4"][2], int(RECIPES["C4"][2] * RECIPES["C4"][1])
    for key in ("rec_vid", "inp_rec_vid", "pred_vid", "inp_pred_vid"):
        assert got[key].shape == (1, t, 3, hd, wd) and torch.isfinite(got[key]).all()
        close(got[key], same32[key], what=key + " (KITTI size)", exact=same64[key])
    assert torch.equal(got["pred_vid"][:, :ctx_len].cpu(), vid[:, :ctx_len])
