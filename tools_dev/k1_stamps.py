"""dev: phase timeline of K1 from the s_memtime stamps of a -DWALDO_K1_STAMPS build
(tools_dev/_variants/stamps.so): median cycles between phase boundaries over the workgroups, for the
last frame of each workgroup's chunk, while the whole chip runs the headline launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import waldo_amd  # noqa: E402
from waldo_amd import _lib, functional as WF  # noqa: E402
_lib.use_library(sys.argv[1] if len(sys.argv) > 1 else "tools_dev/_variants/stamps.so")
from waldo_amd.tools.utils import get_grid  # noqa: E402

dev = torch.device("cuda:0")
F, L, H, W = 112, 8, 256, 512
g = torch.Generator(device=dev).manual_seed(0)
layers = (torch.rand(F, L, 4, H, W, generator=g, device=dev) * 2 - 1).requires_grad_()
pts = (get_grid(4, 4).view(1, 16, 2).to(dev) + 0.05 * torch.randn(F * L, 16, 2, generator=g, device=dev)).requires_grad_()
occ = torch.rand(F, L, L, generator=g, device=dev) * 0.5
tps = waldo_amd.TPSWarp(H, W, get_grid(4, 4).view(-1, 2)).to(dev)
for _ in range(3):
    layers.grad = pts.grad = None
    rgb = WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)
    rgb.square().mean().backward()
torch.cuda.synchronize()
lib = _lib.load()
n_slots, n_blocks = 24, 8192
buf = (ctypes.c_ulonglong * (n_slots * n_blocks))()
lib.waldo_debug_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.waldo_debug_k1_stamps(buf, n_slots * n_blocks)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_blocks, n_slots).astype(np.int64)
valid = st[:, 15] > st[:, 0]
st = st[valid]
names = {0: "frame start", 1: "grid on MFMA, ranges, transposition (A-C)", 2: "boxes, table stores, grad loads, first issue (D)",
         3: "layer 0 staged + barrier", 4: "layer 1", 5: "layer 2", 6: "layer 3", 7: "layer 4", 8: "layer 5", 9: "layer 6", 10: "layer 7",
         11: "last layer sampled", 12: "composite backward (F)", 13: "records, bounds, gg (G) + barrier", 14: "control-point MFMA (H)",
         15: "reduce, partial store, end barrier"}
print(f"{valid.sum()} workgroups; s_memtime ticks (100 MHz constant clock on gfx9: x ~21-24 for shader cycles)")
prev = 0
tot = np.median(st[:, 15] - st[:, 0])
for i in range(1, 16):
    d = st[:, i] - st[:, prev]
    print(f"  {names[i]:52s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}  share {np.median(d) / tot * 100:5.1f} %")
    prev = i
print(f"  frame total median {tot:.0f} ticks")
