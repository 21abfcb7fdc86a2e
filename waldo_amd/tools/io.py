"""On-disk formats of the demo clips that feed the path.

Host-side plumbing, Python in the reference (data/base_dataset.py:168-208, 330-370;
tools/utils.py:250-264) and here:
``read_flo`` / ``write_flo`` for the Middlebury ``.flo`` optical-flow files (magic ``PIEH``, int32
width, int32 height, float32 H*W*2 interleaved (u, v)), normalised to grid units the way
``load_flow_path`` does (u * 2 / W, v * 2 / H); ``layout_to_logits`` for the palette class maps
(one-hot over ``num_lyt`` classes mapped to +-5 logits, with the optional class remapping);
``read_rgb`` / ``read_layout`` for the frame and class-map PNGs with the inference-time transform
of ``get_transform`` (bilinear resize of the image, nearest resize of the one-hot layout,
[-1, 1] normalisation); ``load_clip`` for a directory of the in-tree demo layout
(``leftImg8bit_sequence_512/<split>/<city>/*.png`` with ``_deeplabv3_512`` / ``_raft_128``
siblings); ``dump_image`` / ``dump_video`` for results.
"""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F

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


def read_rgb(path, size=None):
    """RGB frame -> float32 (3, H, W) in [-1, 1].  ``size`` = (H, W): PIL bilinear resize first, as
    ``transforms.Resize(size, PIL.Image.BILINEAR)`` does on a PIL image; then ToTensor and
    Normalize(0.5, 0.5).  Reference: data/base_dataset.py:168-172, 344-345, 364-369."""
    import PIL.Image
    img = PIL.Image.open(path).convert("RGB")
    if size is not None and (img.height, img.width) != tuple(size):
        img = img.resize((int(size[1]), int(size[0])), PIL.Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    return (x - 0.5) / 0.5


def read_layout(path, num_lyt, size=None, remap_lyt=()):
    """Palette class-map PNG -> (num_lyt, H, W) logits in {-5, +5}: class ids as ``ToTensor() * 255``
    reads them, one-hot at the file's resolution, nearest resize of the one-hot planes to ``size``
    (the reference resizes the one-hot TENSOR with the NEAREST transform), then 5 (2 x - 1).
    Reference: data/base_dataset.py:173-183."""
    import PIL.Image
    cm = torch.from_numpy(np.asarray(PIL.Image.open(path)).copy()).long()
    if cm.ndim != 2:
        raise ValueError(f"{path}: expected a single-channel class map, got shape {tuple(cm.shape)}")
    for i in range(len(remap_lyt) // 2):
        cm[cm == remap_lyt[2 * i]] = remap_lyt[2 * i + 1]
    if cm.max() >= num_lyt:
        raise ValueError(f"{path}: class id {int(cm.max())} >= num_lyt {num_lyt}")
    onehot = torch.zeros(num_lyt, *cm.shape).scatter_(0, cm.unsqueeze(0), 1)
    if size is not None and tuple(cm.shape) != tuple(size):
        onehot = F.interpolate(onehot.unsqueeze(0), size=tuple(int(v) for v in size), mode="nearest")[0]
    return 5 * (onehot * 2 - 1)


def load_clip(frames_dir, size, num_lyt, max_frames=None, layout_dir=None, flow_dir=None, remap_lyt=()):
    """A demo clip -> dict(vid (T, 3, H, W) in [-1, 1], lyt (T, num_lyt, H, W) logits,
    flow (T, 2, h, w) or None, names).  ``frames_dir`` holds the frame PNGs in order; the layout /
    flow siblings default to the reference's directory naming
    (.../leftImg8bit_sequence_512/... -> ..._deeplabv3_512 / ..._raft_128).  The first frame of a
    clip has no flow file: it gets zeros (data/video_dataset.py pads the same way)."""
    names = sorted(glob.glob(os.path.join(frames_dir, "*.png")))[:max_frames]
    if not names:
        raise ValueError(f"no frames under {frames_dir}")

    def sibling(tag):
        head, tail = frames_dir, []
        while head and not os.path.basename(head).startswith("leftImg8bit_sequence"):
            head, t = os.path.split(head)
            tail.insert(0, t)
            if not t:
                return None
        base = os.path.basename(head)
        return os.path.join(os.path.dirname(head), base.rsplit("_", 1)[0] + tag, *tail)

    layout_dir = layout_dir or sibling("_deeplabv3_512")
    flow_dir = flow_dir if flow_dir is not None else sibling("_raft_128")
    vid = torch.stack([read_rgb(n, size) for n in names])
    lyt = torch.stack([read_layout(os.path.join(layout_dir, os.path.basename(n)), num_lyt, size, remap_lyt)
                       for n in names])
    flow = None
    if flow_dir and os.path.isdir(flow_dir):
        flows = []
        for n in names:
            fp = os.path.join(flow_dir, os.path.basename(n)[:-4] + ".flo")
            flows.append(read_flo(fp) if os.path.exists(fp) else None)
        shape = next((f.shape for f in flows if f is not None), None)
        if shape is not None:
            flow = torch.stack([f if f is not None else torch.zeros(shape) for f in flows])
    return {"vid": vid, "lyt": lyt, "flow": flow, "names": [os.path.basename(n) for n in names]}


def _to_uint8(tensor, span):
    lo, hi = span if span is not None else (-1.0, 1.0)
    x = ((tensor.detach().float().cpu() - lo) / (hi - lo)).clamp_(0, 1)
    return (x * 255.0 + 0.5).to(torch.uint8)


def dump_image(tensor, path, span=None):
    """(3, H, W) in ``span`` (default [-1, 1]) -> PNG.  Reference: tools/utils.py:250-255."""
    import PIL.Image
    PIL.Image.fromarray(_to_uint8(tensor, span).permute(1, 2, 0).numpy()).save(path)


def dump_video(tensor, path, span=None, fps=4):
    """(T, 3, H, W) in ``span`` -> an animation.  The reference writes mp4 through
    torchvision.io.write_video (tools/utils.py:258-264), which needs PyAV / ffmpeg; neither is in
    this image, so the container format follows the extension PIL can write: ``.gif`` / ``.png``
    (APNG) / ``.webp``; a directory path gets one PNG per frame."""
    import PIL.Image
    frames = [PIL.Image.fromarray(f.permute(1, 2, 0).numpy()) for f in _to_uint8(tensor, span)]
    if os.path.isdir(path) or not os.path.splitext(path)[1]:
        os.makedirs(path, exist_ok=True)
        for i, f in enumerate(frames):
            f.save(os.path.join(path, f"{i:04d}.png"))
        return
    if path.lower().endswith(".mp4"):
        raise ValueError("dump_video: no mp4 encoder in this image (torchvision.io / PyAV absent); "
                         "use .gif, .png (APNG), .webp or a directory")
    frames[0].save(path, save_all=True, append_images=frames[1:], duration=int(1000 / fps), loop=0)
