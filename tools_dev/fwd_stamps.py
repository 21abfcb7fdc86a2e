"""dev: phase timeline of the staged forward at BASELINE config C2 (8 frames of 128 x 128, L = 8: 512 workgroups of
one tile-frame each) from the s_memtime stamps of a -DWALDO_FWD_STAMPS build:

    python tools_dev/build_variant.py fstamps --only warp_composite_lp8,warp_composite -DWALDO_FWD_STAMPS
    python tools_dev/fwd_stamps.py tools_dev/_variants/fstamps.so

Prints, over the workgroups: the spread of their start times, and the median / p90 duration of every phase."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import waldo_amd  # noqa: E402
from waldo_amd import _lib, functional as WF  # noqa: E402
_lib.use_library(sys.argv[1])
from waldo_amd.tools.utils import get_grid  # noqa: E402

dev = torch.device("cuda:0")
F, L, H, W = 8, 8, 128, 128
if len(sys.argv) > 2 and sys.argv[2] == "C3":  # the headline shape: 7 frames per workgroup, steady-state stamps 8 .. 12
    F, L, H, W = 112, 8, 256, 512

g = torch.Generator(device=dev).manual_seed(0)
layers = torch.rand(F, L, 4, H, W, generator=g, device=dev) * 2 - 1
pts = get_grid(4, 4).view(1, 16, 2).to(dev) + 0.05 * torch.randn(F * L, 16, 2, generator=g, device=dev)
occ = torch.rand(F, L, L, generator=g, device=dev) * 0.5
tps = waldo_amd.TPSWarp(H, W, get_grid(4, 4).view(-1, 2)).to(dev)
lib = _lib.load()
if len(sys.argv) > 3:  # a debug option of the C ABI (round 6: the pipelined forward is a variant build, tools_dev/dropped/README.md)
    assert lib.waldo_set_debug_option(int(sys.argv[3]), 1) == 0
with torch.no_grad():
    for _ in range(5):
        WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)
torch.cuda.synchronize()
n_slots, n_blocks = 16, 4096
buf = (ctypes.c_ulonglong * (n_slots * n_blocks))()
lib.waldo_debug_fwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.waldo_debug_fwd_stamps(buf, n_slots * n_blocks) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_blocks, n_slots).astype(np.int64)
st = st[st[:, 6] > st[:, 0]]
names = {1: "LDS cleared, basis operand, mapping folded + barrier", 2: "grid on MFMA, ranges, transposition + barrier",
         3: "boxes, first box loads issued", 5: "layers staged / sampled",
         6: "composite, stores, closing barrier"}
t0 = st[:, 0].min()
print(f"{len(st)} workgroups; ticks of s_memtime (100 MHz: 10 ns)")
print(f"  workgroup start after the first: median {np.median(st[:, 0] - t0):.0f}  p90 {np.percentile(st[:, 0] - t0, 90):.0f}  max {(st[:, 0] - t0).max()}")
print(f"  workgroup end   after the first start: median {np.median(st[:, 6] - t0):.0f}  max {(st[:, 6] - t0).max()}")
prev = 0
for i in (1, 2, 3, 5, 6):
    d = st[:, i] - st[:, prev]
    print(f"  {names[i]:56s} median {np.median(d):6.0f}  p90 {np.percentile(d, 90):6.0f}")
    prev = i
print(f"  workgroup lifetime median {np.median(st[:, 6] - st[:, 0]):.0f}")
if F > 8:
    s2 = st[st[:, 12] > st[:, 8]]
    print(f"second frame of the chunk ({len(s2)} workgroups): top of the loop -> all layers sampled -> end of the frame")
    for a, b, what in ((8, 9, "grid, ranges, transposition + barrier (serial kernel only)"),
                       (9, 10, "boxes, first loads issued (serial kernel only)"),
                       (8, 11, "top of the frame -> every layer sampled"), (11, 12, "composite, stores (+ closing barrier)"),
                       (8, 12, "whole frame")):
        ok = s2[(s2[:, b] >= s2[:, a]) & (s2[:, a] > 0)]
        if len(ok):
            d = ok[:, b] - ok[:, a]
            print(f"  {what:64s} median {np.median(d):6.0f}  p90 {np.percentile(d, 90):6.0f}")
