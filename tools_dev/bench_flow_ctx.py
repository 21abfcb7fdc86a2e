#!/usr/bin/env python3
"""dev: Warper.grid_to_flow_ctx at the reference's Cityscapes recipe R (L = 17, Nl = 20,
128x256 -> 512x1024, Tc = 4, Tp = 1), fused HD passes vs the per-op path.  Prints one JSON line."""
import json, sys, time, types
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd.nets import Warper
from waldo_amd.nets.lvd import compute_occ
from waldo_amd.tools.utils import get_grid

dev = torch.device('cuda:0')
opt = types.SimpleNamespace(latent_shape=[8, 16], obj_shape=[4, 4], time_dropout=False, num_obj=16, patch_size=16,
                            scale_factor=1, dim=128, aspect_ratio=2, load_dim=512, num_perm_grid=1,
                            normalize_alpha=False, use_lyt_filtering=False, use_lyt_opacity=False,
                            weight_cls=False, min_cls=0.0, include_self=False, no_filter=False, allow_ghost=False)
b, t, tc, tp, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 5, 4, 1, 20
wp = Warper(opt).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
no, lo, lb = 16, 16, 128
obj_pose = get_grid(4, 4).view(1, 1, 1, lo, 2).to(dev) * 0.5 + 0.1 * torch.randn(b, t, no, lo, 2, generator=g, device=dev)
bg_pose = get_grid(8, 16).view(1, 1, 1, lb, 2).to(dev) + 0.02 * torch.randn(b, t, 1, lb, 2, generator=g, device=dev)
inp = torch.randn(b, t, 3 + nl, 512, 1024, generator=g, device=dev)
occ = compute_occ(torch.randn(b, t, no, generator=g, device=dev))
obj_alpha = torch.rand(b, no, 1, 64, 64, generator=g, device=dev) * 2 - 1
bg_alpha = torch.ones(b, 1, 128, 256, device=dev)
cls = torch.rand(b, no, nl, generator=g, device=dev).softmax(-1)
ctx_ts = torch.arange(tc, device=dev).view(1, tc, 1).expand(b, tc, tp).contiguous()
pred_ts = torch.tensor([tc], device=dev)
res = {"config": f"R: B={b} Tc={tc} Tp={tp} L=17 Nl={nl} 128x256->512x1024"}
with torch.no_grad():
    grid = wp(obj_pose, bg_pose)
    args = (inp, grid, occ, obj_alpha, bg_alpha, cls, ctx_ts, pred_ts)
    for name, fused in (("fused", True), ("per_op", False)):
        wp.fuse_hd = fused
        for _ in range(3):
            out = wp.grid_to_flow_ctx(*args)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            out = wp.grid_to_flow_ctx(*args)
        e1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = round(e0.elapsed_time(e1) / n, 3)
        res[name + "_peak_extra_MB"] = round((torch.cuda.max_memory_allocated() - base) / 2**20, 1)
    wp.fuse_hd = True
    flow, _, alpha, alpha_ctx, disocc = wp.grid_to_flow_ctx(*args)
    for name, fused in (("i2o_fused", True), ("i2o_per_op", False)):
        wp.fuse_hd = fused
        for _ in range(3):
            out = wp.input_to_output(inp, alpha_ctx, flow, ctx_ts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = wp.input_to_output(inp, alpha_ctx, flow, ctx_ts)
        e1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = round(e0.elapsed_time(e1) / 10, 3)
hwd = 512 * 1024
alg = b * tc * ((nl + 2 * 17) * hwd * 4) + b * tc * tp * ((17 + 17 + 3) * hwd * 4)
res["hd_alg_bytes"] = alg
res["calls_per_s_fused"] = round(1e3 / res["fused_ms"], 1)
print(json.dumps(res))
