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


def expand(mask, num=1, dir=None, soft=False, alpha=0.97):
    """Mask dilation of the reference (tools/utils.py:300-323): ``num`` rounds of one-pixel growth
    to the south, north, east and west IN THAT ORDER (each direction sees the result of the
    previous one, so a round with ``dir=None`` is a 3x3 box dilation).  Hard masks are OR-ed
    (returned as float 0/1); ``soft`` masks take ``max(mask, alpha * neighbour)``.  Unlike the
    reference's soft branch this does not modify its argument."""
    if soft:
        mask = mask.clone()
        for _ in range(num):
            if not dir or dir == "south":
                mask[:, :, 1:, :] = torch.maximum(mask[:, :, 1:, :], alpha * mask[:, :, :-1, :])
            if not dir or dir == "north":
                mask[:, :, :-1, :] = torch.maximum(mask[:, :, :-1, :], alpha * mask[:, :, 1:, :])
            if not dir or dir == "east":
                mask[:, :, :, 1:] = torch.maximum(mask[:, :, :, 1:], alpha * mask[:, :, :, :-1])
            if not dir or dir == "west":
                mask[:, :, :, :-1] = torch.maximum(mask[:, :, :, :-1], alpha * mask[:, :, :, 1:])
        return mask
    m = mask.bool()
    for _ in range(num):
        if not dir or dir == "south":
            m[:, :, 1:, :] = m[:, :, 1:, :] | m[:, :, :-1, :]
        if not dir or dir == "north":
            m[:, :, :-1, :] = m[:, :, :-1, :] | m[:, :, 1:, :]
        if not dir or dir == "east":
            m[:, :, :, 1:] = m[:, :, :, 1:] | m[:, :, :, :-1]
        if not dir or dir == "west":
            m[:, :, :, :-1] = m[:, :, :, :-1] | m[:, :, :, 1:]
    return m.float()
