"""Host-side helpers the hot path needs from the reference's tools/utils.py.

Init-time only (buffers); bit-identical to the reference so that checkpoints interchange.
"""
import math

import torch


def get_grid(height, width):
    """(1, H, W, 2) pixel-centre normalised coordinates, last dim (x, y): the texel centres of
    grid_sample(align_corners=False).  Reference: tools/utils.py:293-297."""
    xs = torch.linspace(-1.0 + 1.0 / width, 1.0 - 1.0 / width, width)
    ys = torch.linspace(-1.0 + 1.0 / height, 1.0 - 1.0 / height, height)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([xx, yy], dim=-1).unsqueeze(0).contiguous()


def get_gaussian_kernel(k, sigma_div=6):
    """Normalised k x k Gaussian with sigma = k / sigma_div.  Reference: tools/utils.py:273-291."""
    coords = torch.arange(k)
    xg = coords.repeat(k).view(k, k)
    yg = xg.t()
    mean = (k - 1) / 2.0
    variance = (k / sigma_div) ** 2.0
    g = (1.0 / (2.0 * math.pi * variance)) * torch.exp(
        -((xg - mean) ** 2.0 + (yg - mean) ** 2.0) / (2 * variance))
    return g / g.sum()


# one-pixel growth steps of expand(), in the reference's order: (name, tensor dim, shift)
_GROW_STEPS = (("south", 2, 1), ("north", 2, -1), ("east", 3, 1), ("west", 3, -1))


def expand(mask, num=1, dir=None, soft=False, alpha=0.97):
    """Mask dilation with the semantics of the reference's tools/utils.py:300-323: ``num`` rounds of
    one-pixel growth towards the south, north, east and west IN THAT ORDER, each step seeing the
    result of the previous one (a round with ``dir=None`` is a 3x3 box dilation; ``dir`` keeps one
    step).  A step combines every pixel with its neighbour ``shift`` pixels back along the step's
    axis: OR for hard masks (returned as float 0 / 1), ``max(pixel, alpha * neighbour)`` for
    ``soft`` ones.  Never modifies its argument (the reference's soft branch does; so would a
    bool input through ``.bool()``)."""
    if soft:
        out = mask.clone()

        def combine(dst, src):
            return torch.maximum(dst, alpha * src)
    else:
        out = mask.to(torch.bool, copy=True)
        combine = torch.logical_or
    steps = [st for st in _GROW_STEPS if not dir or dir == st[0]]
    for _ in range(num):
        for _, dim, shift in steps:
            n = out.shape[dim] - 1
            dst = out.narrow(dim, max(shift, 0), n)
            src = out.narrow(dim, max(-shift, 0), n)
            dst.copy_(combine(dst, src))  # the combination is evaluated before it is written back
    return out if soft else out.float()
