"""GPU: the C4 recipe of waldo_amd/tools/pipeline.py (what ``bench.py --config C4 --pipeline`` times) runs
end to end on one clip: shapes as Synthesizer.predict produces them (models/synthesizer.py:434-472),
finite values, the context frames passed through untouched, and bitwise the same on a second run."""
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
