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
