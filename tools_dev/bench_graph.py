#!/usr/bin/env python3
"""dev: the launch-bound C2 configuration (B*T = 8 frames, L = 8, 128x128, forward only) eagerly vs
replayed from a captured HIP graph (torch.cuda.CUDAGraph around the C-ABI launches)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waldo_amd
from waldo_amd import functional as WF
from waldo_amd.tools.utils import get_grid

dev = torch.device("cuda:0")
f, nl, h, w = 8, 8, 128, 128
g = torch.Generator(device=dev).manual_seed(0)
layers = torch.rand(f, nl, 4, h, w, generator=g, device=dev) * 2 - 1
pts = get_grid(4, 4).view(1, 16, 2).to(dev) + 0.05 * torch.randn(f * nl, 16, 2, generator=g, device=dev)
occ = torch.rand(f, nl, nl, generator=g, device=dev) * 0.5
tps = waldo_amd.TPSWarp(h, w, get_grid(4, 4).view(-1, 2)).to(dev)


def run():
    with torch.no_grad():
        return WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = run().clone()
eager_us = timeit(run)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        run()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = run()
graph.replay()
torch.cuda.synchronize()
same = bool(torch.equal(out, ref))
graph_us = timeit(graph.replay)
print(json.dumps({"config": "C2: 8 frames, L=8, 128x128, fwd", "eager_us": round(eager_us, 1),
                  "graph_us": round(graph_us, 1), "identical": same,
                  "frames_per_s_eager": round(f / eager_us * 1e6), "frames_per_s_graph": round(f / graph_us * 1e6)}))
