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
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
