"""On-disk formats of the demo clips that feed the path.

Host-side plumbing, Python in the reference (data/base_dataset.py:173-208) and here:
``read_flo`` / ``write_flo`` for the Middlebury ``.flo`` optical-flow files (magic ``PIEH``, int32
width, int32 height, float32 H*W*2 interleaved (u, v)), normalised to grid units the way
``load_flow_path`` does (u * 2 / W, v * 2 / H), and ``layout_to_logits`` for the palette class maps
(one-hot over ``num_lyt`` classes mapped to +-5 logits, with the optional class remapping).
"""
import numpy as np
import torch

_MAGIC = b"PIEH"


def read_flo(path, normalize=True):
    """-> float32 tensor (2, H, W).  Reference: data/base_dataset.py:185-203 (without the
    augmentation branches, which are identities at inference)."""
    with open(path, "rb") as fh:
        if fh.read(4) != _MAGIC:
            raise ValueError(f"{path}: not a .flo file (bad magic)")
        width = int(np.frombuffer(fh.read(4), np.int32)[0])
        height = int(np.frombuffer(fh.read(4), np.int32)[0])
        if width <= 0 or height <= 0:
            raise ValueError(f"{path}: bad size {width}x{height}")
        data = np.frombuffer(fh.read(width * height * 8), np.float32)
    if data.size != width * height * 2:
        raise ValueError(f"{path}: truncated ({data.size} of {width * height * 2} floats)")
    flow = torch.from_numpy(data.reshape(height, width, 2).copy()).permute(2, 0, 1).contiguous()
    if normalize:
        flow[0] = 2.0 * flow[0] / width
        flow[1] = 2.0 * flow[1] / height
    return flow


def write_flo(path, flow, normalized=True):
    """flow (2, H, W); ``normalized`` undoes the grid-unit scaling of ``read_flo``."""
    flow = flow.detach().cpu().float()
    _, height, width = flow.shape
    if normalized:
        flow = torch.stack([flow[0] * width / 2.0, flow[1] * height / 2.0])
    with open(path, "wb") as fh:
        fh.write(_MAGIC)
        np.array([width, height], np.int32).tofile(fh)
        flow.permute(1, 2, 0).contiguous().numpy().astype(np.float32).tofile(fh)


def layout_to_logits(class_map, num_lyt, remap_lyt=()):
    """class_map (H, W) or (1, H, W) integer class ids -> (num_lyt, H, W) logits: +5 for the pixel's
    class, -5 elsewhere.  ``remap_lyt`` = flat (src, tgt, src, tgt, ...) as ``opt.remap_lyt``.
    Reference: data/base_dataset.py:173-183."""
    layout = torch.as_tensor(class_map).long().clone()
    if layout.ndim == 2:
        layout = layout.unsqueeze(0)
    for i in range(len(remap_lyt) // 2):
        layout[layout == remap_lyt[2 * i]] = remap_lyt[2 * i + 1]
    onehot = torch.zeros(num_lyt, *layout.shape[-2:]).scatter_(0, layout, 1)
    return 5 * (onehot * 2 - 1)
