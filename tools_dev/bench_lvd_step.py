#!/usr/bin/env python3
"""dev: the warp-path part of one LVD training step at the reference's recipe
(scripts/cityscapes/train_lvd.sh: 128x256, no HD raster, 16 objects, 5 frames, ctx_mode "prev",
include_self, layout filtering with weighted classes; models/synthesizer.py:815-841) under
autograd: estimate_alpha_grid_occ -> decode_output -> loss -> backward.  Prints one JSON line."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd.nets import Warper, decode_output, estimate_alpha_grid_occ  # noqa: E402
from waldo_amd.nets.lvd import decoder_tail  # noqa: E402
from waldo_amd.nets import flp  # noqa: E402
from waldo_amd.tools.utils import get_grid  # noqa: E402

dev = torch.device("cuda:0")
if "--lib" in sys.argv:  # A/B runs: another build of the library (tools_dev/run_lvd_ab.sh)
    from waldo_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.use_library(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
b = int(_pos[0]) if len(_pos) > 0 else 2
iters = int(_pos[1]) if len(_pos) > 1 else 10
t, no, nl, lo, lb = 5, 16, 20, 16, 128
opt = types.SimpleNamespace(latent_shape=[8, 16], obj_shape=[4, 4], time_dropout=0.0, num_obj=no, patch_size=16,
                            scale_factor=1, dim=128, aspect_ratio=2, load_dim=0, num_perm_grid=1,
                            normalize_alpha=False, use_lyt_filtering=True, use_lyt_opacity=True, weight_cls=True,
                            min_cls=0.1, include_self=True, no_filter=False, allow_ghost=False)
wp = Warper(opt).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
raw = torch.randn(b * no, 1, 64, 64, generator=g, device=dev, requires_grad=True)
pose_o = (0.3 * torch.randn(b * t, no, 6 + 2 * lo, generator=g, device=dev)).requires_grad_()
pose_b = (0.05 * torch.randn(b * t, 1, 6 + 2 * lb, generator=g, device=dev)).requires_grad_()
score = torch.randn(b, t, no, generator=g, device=dev, requires_grad=True)
cls_logit = torch.randn(b, no, nl, generator=g, device=dev, requires_grad=True)
inp = torch.randn(b, t, 3 + nl, 128, 256, generator=g, device=dev)
base_o = get_grid(4, 4).view(1, 1, lo, 2).to(dev)
base_b = get_grid(8, 16).view(1, 1, lb, 2).to(dev)
mul6 = torch.tensor([[[0.25, 0.25, 0.25, 0.25, 1.0, 1.0]]], device=dev)
bias_o = torch.tensor([[[0.25, 0.0, 0.0, 0.5, 0.0, 0.0]]], device=dev)
bias_b = torch.tensor([[[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]], device=dev)
bg_alpha = torch.ones(1, 1, 128, 256, device=dev)
ctx_ts = torch.roll(torch.arange(t, device=dev), 1).view(1, 1, t).expand(b, -1, -1).contiguous()
pred_ts = torch.arange(t, device=dev)
leaves = [raw, pose_o, pose_b, score, cls_logit]


def step():
    for x in leaves:
        x.grad = None
    obj_alpha = decoder_tail(raw, init_bias=5.0).view(b, no, 1, 64, 64)
    obj_pose = flp.obj_pose_to_points(torch.tanh(pose_o), base_o, mul6, bias_o, 0.2).view(b, t, no, lo, 2)
    bg_pose = flp.bg_pose_to_points(torch.tanh(pose_b), base_b, bias_b, 1.2).view(b, t, 1, lb, 2)
    occ, oa, ba, grid = estimate_alpha_grid_occ(wp, obj_alpha, bg_alpha, obj_pose, bg_pose, score)
    out = decode_output(wp, inp, grid, occ, oa, ba, cls_logit.softmax(-1), ctx_ts, pred_ts, restrict_to_ctx=False)
    loss = out[0].square().mean() + out[1].square().mean() + out[3].mean()
    loss.backward()
    return loss


graph = "--graph" in sys.argv
for _ in range(3):
    step()
torch.cuda.synchronize()
if graph:
    # the whole step -- forward, loss, backward -- captured once and replayed (the library launches on the
    # current stream and never synchronises; its host-side index checks are skipped during capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        loss = step()
    run = g_.replay
else:
    run = step
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out_ = run()
    loss = out_ if out_ is not None else loss
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
print(json.dumps({"config": f"LVD recipe step: B={b} T={t} L={no + 1} Nl={nl} 128x256, ctx prev + include_self" + (", one HIP graph" if graph else ""),
                  "ms_per_step": round(ms, 3), "loss": float(loss),
                  "grads_finite": all(bool(torch.isfinite(x.grad).all()) for x in leaves)}))
