"""GPU: the C4 and C5 recipes of waldo_amd/tools/pipeline.py (what ``bench.py --config C4 / C5 --pipeline`` times) run
end to end on one clip at their own sizes: shapes as Synthesizer.predict produces them (models/synthesizer.py:434-472),
finite values, the context frames passed through untouched, bitwise the same on a second run, and -- the decode
without autograd composites the context alphas straight into raw_output -- the same frames as the two-tensor path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_c4_pipeline_one_clip(dev):
    from waldo_amd.tools.pipeline import RECIPES, Pipeline
    pipe = Pipeline("C4", 1, dev, seed=3)
    t, ctx = RECIPES["C4"][5], RECIPES["C4"][6]
    out = pipe()
    hd, wd = 256, 832
    assert out["rec_vid"].shape == (1, t, 3, hd, wd) and out["inp_rec_vid"].shape == (1, t, 3, hd, wd)
    assert out["pred_vid"].shape == (1, t, 3, hd, wd) and out["inp_pred_vid"].shape == (1, t, 3, hd, wd)
    assert out["pred_flow"].shape == (1, ctx, t - ctx, 2, hd, wd)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    assert torch.equal(out["inp_pred_vid"][:, :ctx], pipe.vid[:, :ctx])
    again = pipe()
    for k in ("inp_pred_vid", "inp_rec_vid", "pred_flow"):
        assert torch.equal(out[k], again[k]), k
    alg = pipe.hd_algorithmic_bytes()
    assert {"waldo_flow_ctx_alpha_fwd", "waldo_flow_ctx_warp_raw_fwd", "waldo_frame_warp_fuse_raw_fwd",
            "waldo_wif_fuse_fwd"} <= set(alg) and all(v > 0 for v in alg.values())


@pytest.mark.parametrize("motion", ["calibrated", "wild"])
def test_c5_pipeline_one_clip(dev, motion):
    """The Cityscapes recipe at its own size (512 x 1024, 12 layers, 14 frames, 4 contexts), both stand-in motions."""
    from waldo_amd import functional as WF
    from waldo_amd.tools.pipeline import RECIPES, Pipeline
    pipe = Pipeline("C5", 1, dev, seed=5, motion=motion)
    t, ctx = RECIPES["C5"][5], RECIPES["C5"][6]
    out = pipe()
    hd, wd = 512, 1024
    for k in ("rec_vid", "inp_rec_vid", "pred_vid", "inp_pred_vid"):
        assert out[k].shape == (1, t, 3, hd, wd), k
    assert out["pred_flow"].shape == (1, ctx, t - ctx, 2, hd, wd)
    assert out["rec_disocc"].shape == (1, t, 1, hd, wd) and out["pred_disocc"].shape == (1, t - ctx, 1, hd, wd)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    assert torch.equal(out["inp_pred_vid"][:, :ctx], pipe.vid[:, :ctx])
    assert out["inp_rec_vid"].std() > 0.05  # not a blank frame
    again = pipe()
    for k in ("inp_pred_vid", "inp_rec_vid", "pred_flow", "rec_disocc"):
        assert torch.equal(out[k], again[k]), k
    # the same predict with the raw-slot short cut switched off (alpha_ctx in a tensor of its own, read and copied by
    # the frame warp): bit for bit the same products
    long_way, calls = WF.frame_warp_fuse, []

    def counted(*a, **kw):
        calls.append(1)
        return long_way(*a, **kw)

    WF.frame_warp_fuse = counted
    pipe.warper.raw_slots = False
    try:
        plain = pipe()
    finally:
        WF.frame_warp_fuse = long_way
        pipe.warper.raw_slots = True
    from waldo_amd.tools import demo
    # (reconstruction and prediction went the long way: in ONE decode when predict() merges them, demo.MERGE_DECODES)
    assert len(calls) == (1 if demo.MERGE_DECODES else 2)
    # ... and the reference's two calls one after the other give the same bits as the merged decode
    demo.MERGE_DECODES = not demo.MERGE_DECODES
    try:
        other = pipe()
    finally:
        demo.MERGE_DECODES = not demo.MERGE_DECODES
    for k in out:
        assert torch.equal(out[k], other[k]), f"merged / separate decodes differ in {k}"
    for k in out:
        assert torch.equal(out[k], plain[k]), k


def test_predict_replays_from_one_hip_graph(dev):
    """The whole hot-path part of predict() -- producers, both Warper.forward calls, both decodes, both WIF fusions,
    ~200 launches -- is capturable into ONE HIP graph (the library launches on the caller's stream and never
    synchronises; the path's constant tensors live on the device; the kernels validate the frame indices themselves) and the
    replay has the same bits as the eager call, also after the clip changes."""
    from waldo_amd.graphs import GraphedCall
    from waldo_amd.tools import demo
    from waldo_amd.tools.pipeline import Pipeline, synthetic_clip
    pipe = Pipeline("C4", 1, dev, seed=9)

    def run(vid, lyt):
        out = demo.predict(pipe.opt, pipe.warper, pipe.wif, vid, lyt, pipe.net, pipe.ctx_len)
        return out["inp_pred_vid"], out["inp_rec_vid"], out["pred_flow"], out["pred_disocc"]

    with torch.no_grad():
        graphed = GraphedCall(run, pipe.vid, pipe.lyt)
        other = synthetic_clip(pipe.opt, 1, pipe.frames, 10, dev)
        for vid, lyt in ((pipe.vid, pipe.lyt), other, (pipe.vid, pipe.lyt)):
            eager = run(vid, lyt)
            for x, y in zip(graphed(vid, lyt), eager):
                assert torch.equal(x, y)


def _assemble(pipe_full, blocks, key):
    """The ranks' unit blocks of one key, in rank order, shaped as predict() returns that key (what gather_predict does
    with the all-gathered blocks: the reconstruction's units go back from their dealing order to frame order)."""
    from waldo_amd.tools import demo
    b, t, ctx = pipe_full.clips, pipe_full.frames, pipe_full.ctx_len
    full = torch.cat([blk[key] for blk in blocks], dim=0)
    per_clip = t if key in demo.UNIT_KEYS["rec"] else t - ctx
    assert full.shape[0] == b * per_clip, (key, full.shape)
    return demo.units_to_clips(key, full, b, t, ctx, len(blocks), pipe_full.vid)


@pytest.mark.parametrize("name,clips,worlds", [("C4", 1, (2, 5, 8)), ("C4", 2, (3, 4)), ("C5", 1, (8,)),
                                               ("C4", 3, (2, 7)), ("C5", 2, (3, 5))])
def test_predict_split_by_output_frames_has_the_same_bits(dev, name, clips, worlds):
    """One job split over `world` ranks by (b, t) output units (demo.predict_sharded): every rank decodes
    its contiguous block from the clip's context frames and its OWN frames' poses only, and the blocks put together
    are bit for bit what the single-rank predict() returns -- ragged splits (5 predicted frames over 2 ranks, 9
    reconstructed over 8, ranks with no unit at all) and blocks that cross a clip boundary (2 clips over 3 ranks)
    included.  All ranks run one after the other in this process; the collective is tests/test_gpu_dist.py's."""
    from waldo_amd.tools.pipeline import Pipeline
    full = Pipeline(name, clips, dev, seed=4)
    ref = full()
    for world in worlds:
        blocks = []
        for r in range(world):
            full.shard = (r, world)  # (the same job, another rank's share)
            blocks.append(full())
        full.shard = None
        for key, want in ref.items():
            got = _assemble(full, blocks, key)
            assert got.shape == want.shape, (world, key, got.shape, want.shape)
            assert torch.equal(got, want), (world, key, (got - want).abs().max().item())
