"""Row f4: on-disk formats of the demo clips (host logic; no GPU)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from waldo_amd.tools import io as wio

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO = os.path.join(HERE, "golden", "demo_flow.flo")  # a data file of the reference's demo set
REF = "/root/reference"


def test_read_flo_demo_file():
    flow = wio.read_flo(DEMO)
    assert flow.shape == (2, 128, 256) and flow.dtype == torch.float32
    raw = wio.read_flo(DEMO, normalize=False)
    assert torch.equal(flow[0], 2.0 * raw[0] / 256) and torch.equal(flow[1], 2.0 * raw[1] / 128)
    assert torch.isfinite(flow).all() and flow.abs().max() < 2.0


def test_flo_round_trip_and_errors(tmp_path):
    g = torch.Generator().manual_seed(0)
    flow = torch.randn(2, 7, 11, generator=g)
    p = tmp_path / "a.flo"
    wio.write_flo(p, flow)
    back = wio.read_flo(p)
    assert torch.allclose(back, flow, atol=1e-6)
    wio.write_flo(p, flow, normalized=False)
    assert torch.equal(wio.read_flo(p, normalize=False), flow)
    (tmp_path / "bad.flo").write_bytes(b"XXXX" + b"\0" * 16)
    with pytest.raises(ValueError):
        wio.read_flo(tmp_path / "bad.flo")
    data = open(p, "rb").read()
    (tmp_path / "short.flo").write_bytes(data[:-8])
    with pytest.raises(ValueError):
        wio.read_flo(tmp_path / "short.flo")


def test_layout_to_logits():
    cm = torch.tensor([[0, 1, 2], [2, 255, 1]])
    lg = wio.layout_to_logits(cm, 4, remap_lyt=(255, 3))
    assert lg.shape == (4, 2, 3)
    assert torch.equal(lg.argmax(0), torch.tensor([[0, 1, 2], [2, 3, 1]]))
    assert set(lg.unique().tolist()) == {-5.0, 5.0} and torch.equal((lg > 0).sum(0), torch.ones(2, 3, dtype=torch.long))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_formats_against_reference_loader():
    """The reference's own loader methods (data/base_dataset.py, executed from where it lies under
    stubbed torchvision / tools.utils parents) on the same files."""
    import importlib.util
    import PIL.Image

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img)
            a = a[None] if a.ndim == 2 else a.transpose(2, 0, 1)
            return torch.from_numpy(a.copy()).float() / 255

    stubs = {
        "torchvision": types.ModuleType("torchvision"),
        "torchvision.transforms": types.ModuleType("torchvision.transforms"),
        "torchvision.datasets": types.ModuleType("torchvision.datasets"),
        "torchvision.datasets.video_utils": types.ModuleType("torchvision.datasets.video_utils"),
        "tools": types.ModuleType("tools"),
        "tools.utils": types.ModuleType("tools.utils"),
    }
    stubs["torchvision.transforms"].ToTensor = ToTensor
    stubs["torchvision"].transforms = stubs["torchvision.transforms"]
    stubs["torchvision.datasets.video_utils"].VideoClips = object
    for name in ("get_vprint", "serialize", "deserialize"):
        setattr(stubs["tools.utils"], name, lambda *a, **k: None)
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("ref_base_dataset", os.path.join(REF, "data", "base_dataset.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ds = object.__new__(mod.BaseDataset)
        ds.opt = types.SimpleNamespace(remap_lyt=[], num_lyt=20, flow_dim=0)
        ref_flow = ds.load_flow_path(DEMO)
        assert torch.equal(wio.read_flo(DEMO), ref_flow)
        lyt_dir = os.path.join(REF, "datasets", "demo_cityscapes", "leftImg8bit_sequence_deeplabv3_512", "val", "munster")
        p = os.path.join(lyt_dir, sorted(os.listdir(lyt_dir))[0])
        ref_lyt = ds.load_layout_path(p)
        ids = torch.from_numpy(np.asarray(PIL.Image.open(p)).copy()).long()
        assert torch.equal(wio.layout_to_logits(ids, 20), ref_lyt)
        assert torch.equal(wio.read_layout(p, 20), ref_lyt)
        # frames: load_rgb_path + the inference transform of get_transform (Resize on the PIL image,
        # ToTensor, Normalize(0.5, 0.5)) with stand-ins for the three torchvision classes it composes
        tv = stubs["torchvision.transforms"]
        tv.Resize = lambda size, method: (lambda img: img.resize((size[1], size[0]), method))
        tv.Normalize = lambda mean, std: (lambda x: (x - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1))

        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        tv.Compose = Compose
        img_dir = lyt_dir.replace("_deeplabv3_512", "_512")
        pi = os.path.join(img_dir, sorted(os.listdir(img_dir))[0])
        for dim, ar in ((128, 1.0), (128, 2.0), (512, 2.0)):
            ref_img = ds.load_rgb_path(pi, mod.get_transform(dim, aspect_ratio=ar))
            assert torch.equal(wio.read_rgb(pi, (dim, int(dim * ar))), ref_img), (dim, ar)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


CLIP = os.path.join(HERE, "golden", "demo_clip", "leftImg8bit_sequence_512", "val", "munster")


def test_load_clip_and_dumps(tmp_path):
    """The committed six-frame demo clip (C1) through load_clip, and the result writers."""
    import PIL.Image
    clip = wio.load_clip(CLIP, (64, 128), 20, max_frames=5)
    assert clip["vid"].shape == (5, 3, 64, 128) and clip["lyt"].shape == (5, 20, 64, 128)
    assert clip["flow"] is None and clip["names"][0].endswith("000001_leftImg8bit.png")
    assert clip["vid"].min() >= -1 and clip["vid"].max() <= 1 and clip["vid"].std() > 0.1
    assert torch.equal((clip["lyt"] > 0).sum(1), torch.ones(5, 64, 128, dtype=torch.long))
    wio.dump_image(clip["vid"][0], tmp_path / "f.png")
    back = wio.read_rgb(tmp_path / "f.png")
    assert (back - clip["vid"][0]).abs().max() <= 1.0 / 255 + 1e-6
    wio.dump_video(clip["vid"], str(tmp_path / "v.gif"))
    assert PIL.Image.open(tmp_path / "v.gif").n_frames == 5
    wio.dump_video(clip["vid"], str(tmp_path / "frames"))
    assert sorted(os.listdir(tmp_path / "frames")) == [f"{i:04d}.png" for i in range(5)]
    with pytest.raises(ValueError):
        wio.dump_video(clip["vid"], str(tmp_path / "v.mp4"))
    with pytest.raises(ValueError):
        wio.read_layout(os.path.join(CLIP.replace("_512", "_deeplabv3_512"), clip["names"][0]), 5)
