#!/usr/bin/env python3
"""dev: how many object layers pass the ghost test per 64-pixel row segment (a wavefront's row) and per 16 x 64 tile in
the C4 / C5 pipeline's flow pass (the sets flow_ctx_warp's skipping works on).   python tools_dev/fcw_layer_stats.py C5"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd import functional as WF  # noqa: E402
from waldo_amd.tools.pipeline import Pipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
motion = sys.argv[2] if len(sys.argv) > 2 else "calibrated"
dev = torch.device("cuda:0")
pipe = Pipeline(name, 1, dev, motion=motion)
seen = []
orig = WF.flow_ctx_warp_into_raw


def spy(flow_lr, is_obj, *a, **k):
    seen.append(is_obj)
    return orig(flow_lr, is_obj, *a, **k)


WF.flow_ctx_warp_into_raw = spy
with torch.no_grad():
    pipe()
s = int(pipe.opt.load_dim // pipe.opt.dim)
for i, m in enumerate(seen):
    up = F.interpolate(m, scale_factor=s, mode="bilinear") > 0.9      # (M, No, Hd, Wd)
    M, no, hd, wd = up.shape
    seg = up.view(M, no, hd, wd // 64, 64).any(-1)                      # per 64-pixel row segment
    tile = seg.view(M, no, hd // 16, 16, wd // 64).any(3)               # per 16 x 64 tile
    px = up.float().sum(1)
    print(f"{name} {motion} decode {i}: units {M}, objects {no}; layers passing per pixel {px.mean().item():.2f}, "
          f"per row segment {seg.float().sum(1).mean().item():.2f}, per tile {tile.float().sum(1).mean().item():.2f}; "
          f"segments with none {100 * (seg.sum(1) == 0).float().mean().item():.0f} %, tiles with none "
          f"{100 * (tile.sum(1) == 0).float().mean().item():.0f} %")
