"""dev: the headline step (C3: forward, rgb.square().mean(), backward) eagerly -- with and without the per-call event
pairs of _lib.KernelTimer -- and replayed from one HIP graph.   python tools_dev/c3_graph.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
import waldo_amd  # noqa: E402
from waldo_amd import _lib, functional as WF  # noqa: E402
from waldo_amd.tools.utils import get_grid  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
frames, nl, h, w = 112, 8, 256, 512
tps = waldo_amd.TPSWarp(h, w, get_grid(4, 4).view(-1, 2)).to(dev)
layers, pts, occ = bench.synth(frames, nl, h, w, dev, seed=0, sigma=0.05)
layers.requires_grad_()
pts.requires_grad_()


def step():
    layers.grad = None
    pts.grad = None
    rgb = WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)
    rgb.square().mean().backward()


def timed(run, n=steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(10):
    step()
bench.settle_interpreter()
for rnd in range(3):
    e = timed(step)
    with _lib.KernelTimer():
        ek = timed(step)
    print(f"round {rnd}: eager {e:.4f} ms   eager with event pairs {ek:.4f} ms", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g0 = layers.grad.clone()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
print("same gradient bits from the replay:", bool(torch.equal(layers.grad, g0)))
for rnd in range(3):
    print(f"round {rnd}: eager {timed(step):.4f} ms   graph replay {timed(graph.replay):.4f} ms", flush=True)
