"""dev: A/B timing of the full-resolution forward entry points (frame_warp_fuse, flow_ctx_warp, flow_ctx_alpha)
across library builds on one box, interleaved, at the C5 recipe's size (512 x 1024, L = 12, 23 input channels,
4 context frames).  Calls the C ABI through ctypes directly.

    python tools_dev/ab_hd.py [--clips 2 --tp 10 --rounds 3] lib1.so lib2.so ..."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, '.')
from waldo_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, default=2)
ap.add_argument('--tp', type=int, default=10)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--dim', type=int, default=128)
ap.add_argument('--scale', type=int, default=4)
ap.add_argument('--layers', type=int, default=12)
ap.add_argument('--amp', type=float, default=0.02, help='flow amplitude in grid units (0.02 = 10 px at 1024)')
ap.add_argument('--coarse', type=int, default=32, help='the flow is smooth over this many pixels')
ap.add_argument('libs', nargs='+')
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
b, t, tc, tp, nl, ncls = a.clips, 14, 4, a.tp, a.layers, 20
c = 3 + ncls
h, w, s = a.dim, 2 * a.dim, a.scale
hd, wd = h * s, w * s
inp = torch.randn(b, t, c, hd, wd, generator=g, device=dev)
# smooth flows of a few pixels (grid units), like the composited TPS flows of the chain
lo = a.amp * torch.randn(b * tc * tp, 2, hd // a.coarse, wd // a.coarse, generator=g, device=dev)
flow = torch.nn.functional.interpolate(lo, size=(hd, wd), mode='bilinear').view(b, tc, tp, 2, hd, wd).contiguous()
alpha = torch.rand(b, tc, tp, nl, hd, wd, generator=g, device=dev) * 2 - 1
ctx_ts = torch.arange(tc, device=dev).view(1, tc, 1).expand(b, tc, tp).contiguous()
pred_ts = torch.arange(tc, tc + tp, device=dev)
out = torch.empty(b, tp, c + 1, hd, wd, device=dev)
raw = torch.empty(b, tp, tc, c + nl, hd, wd, device=dev)
# flow_ctx_warp inputs
m = b * tc * tp
flow_lr = 0.02 * torch.randn(m, nl, 2, h, w, generator=g, device=dev)
a01 = torch.rand(b * tc, nl, hd, wd, generator=g, device=dev)
occ = torch.rand(b, t, nl, nl, generator=g, device=dev) * 0.5
oflow = torch.empty(m, 2, hd, wd, device=dev)
oactx = torch.empty(m, nl, hd, wd, device=dev)
odis = torch.empty(m, hd, wd, device=dev)
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)


def ptr(x):
    return P(x.data_ptr()) if x is not None else None


libs = []
for path in a.libs:
    lib = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
    for name in ("waldo_frame_warp_fuse_fwd", "waldo_flow_ctx_warp_fwd"):
        getattr(lib, name).argtypes = _lib.SIGNATURES[name]
    libs.append((os.path.basename(path), lib))


def fwf(lib):
    assert lib.waldo_frame_warp_fuse_fwd(ptr(inp), ptr(flow), ptr(alpha), ptr(ctx_ts), ptr(out), ptr(raw), b, t, tc, tp,
                                         c, nl, hd, wd, 0, 1e-6, st) == 0


def fcw(lib):
    assert lib.waldo_flow_ctx_warp_fwd(ptr(flow_lr), None, ptr(a01), ptr(ctx_ts), ptr(pred_ts), ptr(occ), ptr(oflow),
                                       ptr(oactx), ptr(odis), None, b, t, tc, tc, tp, nl, h, w, s, st) == 0


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


px = b * tp * hd * wd
fwf_bytes = 4 * px * (tc * (c + 2 + nl) + (c + 1) + tc * (c + nl))
fcw_bytes = 4 * (m * nl * 2 * h * w + m * nl * hd * wd + m * (2 + nl + 1) * hd * wd)
for r in range(a.rounds):
    for name, lib in libs:
        t1 = timeit(lambda: fwf(lib), a.iters)
        t2 = timeit(lambda: fcw(lib), a.iters)
        print(f'round {r} {name:24s} frame_warp_fuse {t1:8.3f} ms ({fwf_bytes / t1 / 1e9:6.2f} TB/s alg)  flow_ctx_warp {t2:8.3f} ms '
              f'({fcw_bytes / t2 / 1e9:6.2f} TB/s alg)  chk {out.double().sum().item():.6e} {raw.double().sum().item():.6e} '
              f'{oactx.double().sum().item():.6e}', flush=True)
