"""dev: A/B timing of the fused entry points across library builds, on one box, interleaved.

    python tools_dev/ab_bench.py [--shape F,L,H,W] [--iters 20] [--rounds 3] lib1.so lib2.so ...

Calls waldo_warp_composite_fwd / _bwd through ctypes directly (libraries older than version 1002
have no `delta` argument), events on the current stream around each call."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, '.')
from waldo_amd.tools.utils import get_grid  # noqa: E402
import waldo_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='112,8,256,512')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--sigma', type=float, default=0.05)
ap.add_argument('libs', nargs='+')
args = ap.parse_args()
F, L, H, W = (int(x) for x in args.shape.split(','))
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
layers = torch.rand(F, L, 4, H, W, generator=g, device=dev) * 2 - 1
ctrl = get_grid(4, 4).view(1, 16, 2).to(dev)
pts = ctrl + args.sigma * torch.randn(F * L, 16, 2, generator=g, device=dev)
score = torch.randn(F, L - 1, generator=g, device=dev)
s = torch.exp(-score ** 2) + 1e-6
occ = torch.zeros(F, L, L, device=dev)
occ[:, 1:, 1:] = s[:, :, None] / (s[:, :, None] + s[:, None, :]) - 0.5 * torch.eye(L - 1, device=dev)
occ[:, 1:, 0] = 1.0
tps = waldo_amd.TPSWarp(H, W, get_grid(4, 4).view(-1, 2)).to(dev)
from waldo_amd import functional as WF  # noqa: E402
mapping = WF.tps_mapping(tps.inverse_kernel, pts).contiguous()
basis_t = tps.basis_t.contiguous()
rgb = torch.empty(F, 3, H, W, device=dev)
grad_rgb = torch.randn(F, 3, H, W, generator=g, device=dev) * 1e-6
gl = torch.empty_like(layers)
gm = torch.zeros_like(mapping)
P = ctypes.c_void_p
i64, i32 = ctypes.c_int64, ctypes.c_int


def ptr(t):
    return P(t.data_ptr()) if t is not None else None


libs = []
for path in args.libs:
    lib = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
    lib.waldo_warp_composite_bwd_workspace_bytes.restype = i64
    lib.waldo_warp_composite_bwd_workspace_bytes.argtypes = [i64, i32, i32, i32, i32]
    lib.waldo_version.restype = i32
    lib.extra = (ctypes.c_float(0.0),) if lib.waldo_version() >= 1002 else ()
    tail = [ctypes.c_float] * len(lib.extra) + [P]
    lib.waldo_warp_composite_fwd.argtypes = [P] * 6 + [i64, i32, i32, i32, i32] + tail
    lib.waldo_warp_composite_bwd.argtypes = [P] * 10 + [i64, i64, i32, i32, i32, i32] + tail
    wsb = lib.waldo_warp_composite_bwd_workspace_bytes(F, L, H, W, 19)
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.int32, device=dev)
    libs.append((os.path.basename(path), lib, ws, wsb))
st = P(torch.cuda.current_stream().cuda_stream)


def fwd(lib):
    rc = lib.waldo_warp_composite_fwd(ptr(layers), ptr(basis_t), ptr(mapping), ptr(occ), ptr(rgb), None, F, L, H, W, 19, *lib.extra, st)
    assert rc == 0


def bwd(lib, ws, wsb):
    if wsb == 0:
        gl.zero_()
    rc = lib.waldo_warp_composite_bwd(ptr(layers), ptr(basis_t), ptr(mapping), ptr(occ), ptr(grad_rgb), None, ptr(gl),
                                      ptr(gm), None, ptr(ws) if wsb else None, wsb, F, L, H, W, 19, *lib.extra, st)
    assert rc == 0


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for r in range(args.rounds):
    for name, lib, ws, wsb in libs:
        tf = timeit(lambda: fwd(lib), args.iters)
        tb = timeit(lambda: bwd(lib, ws, wsb), args.iters)
        chk = (rgb.double().sum().item(), gl.double().abs().sum().item())
        print(f'round {r} {name:28s} fwd {tf:.4f} ms  bwd {tb:.4f} ms  sum {tf + tb:.4f}  chk rgb {chk[0]:.6e} gl {chk[1]:.6e}', flush=True)
