"""dev: footprint boxes of frame_warp_fuse's tiles under the flows the C4 / C5 pipeline really hands it.
For every (unit, context) and tile shape: the box [floor(min ix), floor(max ix) + 1] x [.. iy ..] of the sample
positions, clamped to the frame; prints the distribution of box texels relative to the tile's pixels and the share
of (tile, context) pairs under a few LDS caps.

    python tools_dev/fwf_boxes.py [--config C5 --clips 1]"""
import argparse
import sys

import torch

sys.path.insert(0, '.')
from waldo_amd import functional as WF  # noqa: E402
from waldo_amd.tools import pipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='C5')
ap.add_argument('--clips', type=int, default=1)
ap.add_argument('--motion', default='calibrated')
a = ap.parse_args()
dev = torch.device('cuda:0')
calls = []
orig = WF.frame_warp_fuse


def spy(input, flow, alpha, ctx_ts, include_self=False, eps=1e-6):
    calls.append(flow.detach())
    return orig(input, flow, alpha, ctx_ts, include_self=include_self, eps=eps)


WF.frame_warp_fuse = spy
p = pipeline.Pipeline(a.config, a.clips, dev, motion=a.motion)
p()
torch.cuda.synchronize()
for ci, flow in enumerate(calls):
    b, tc, tp, _, hd, wd = flow.shape
    print(f'call {ci}: flow {tuple(flow.shape)}  |flow_x| px mean {flow[:, :, :, 0].abs().mean().item() * wd / 2:.1f} '
          f'max {flow[:, :, :, 0].abs().max().item() * wd / 2:.1f}  |flow_y| px mean '
          f'{flow[:, :, :, 1].abs().mean().item() * hd / 2:.1f} max {flow[:, :, :, 1].abs().max().item() * hd / 2:.1f}')
    xs = (torch.arange(wd, device=dev) * 2 + 1) / wd - 1
    ys = (torch.arange(hd, device=dev) * 2 + 1) / hd - 1
    ix = ((flow[:, :, :, 0] + xs.view(1, 1, 1, 1, wd) + 1) * wd - 1) / 2
    iy = ((flow[:, :, :, 1] + ys.view(1, 1, 1, hd, 1) + 1) * hd - 1) / 2
    # local stretch: |d ix / dx| and |d iy / dy|
    dx = (ix[..., 1:] - ix[..., :-1]).abs()
    dy = (iy[..., 1:, :] - iy[..., :-1, :]).abs()
    q = torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev)
    print('   d ix/dx quantiles 10/50/90/99 %:', [round(v, 2) for v in torch.quantile(dx.flatten()[::97], q).tolist()],
          ' d iy/dy:', [round(v, 2) for v in torch.quantile(dy.flatten()[::97], q).tolist()])
    for th, tw in ((8, 32), (16, 32), (16, 16), (8, 64), (32, 32), (16, 64)):
        def box(v, size):
            t = v.reshape(-1, hd // th, th, wd // tw, tw)
            lo = t.amin(dim=(2, 4)).floor().clamp(0, size - 1)
            hi = (t.amax(dim=(2, 4)).floor() + 1).clamp(0, size - 1)
            # a tile entirely outside the frame: nothing to stage
            outside = (t.amax(dim=(2, 4)) < -1) | (t.amin(dim=(2, 4)) > size)
            return lo, hi, outside
        xl, xh, xo = box(ix, wd)
        yl, yh, yo = box(iy, hd)
        cols, rows = (xh - xl + 1), (yh - yl + 1)
        area = cols * rows
        area = torch.where(xo | yo, torch.zeros_like(area), area)
        rel = area / (th * tw)
        n = rel.numel()
        line = f'   tile {th:2d}x{tw:2d}: box/tile mean {rel.mean().item():.2f} median {rel.median().item():.2f}'
        for cap in (512, 1024, 2048, 4096):
            ok = area <= cap
            # traffic if boxes <= cap are staged (their texels, rows rounded to 32-byte sectors) and the rest gathered
            line += f' | cap {cap}: {100.0 * ok.sum().item() / n:.1f} % staged, box/tile of those {rel[ok].mean().item():.2f}'
        print(line)
        rows_cap = rows[(area <= 2048) & (area > 0)]
        cols_cap = cols[(area <= 2048) & (area > 0)]
        print(f'      (cap 2048) rows mean {rows_cap.mean().item():.1f} max {rows_cap.max().item():.0f}  cols mean '
              f'{cols_cap.mean().item():.1f} max {cols_cap.max().item():.0f}')
