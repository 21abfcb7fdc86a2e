"""dev: two-kernel backward vs the generic per-tap-atomics backward, with a breakdown of where they
differ.  usage: python tools_dev/dbg_bwd.py "nl,h,w,f" ..."""
import collections
import sys

import torch

sys.path.insert(0, '.')
from oracle import wif_oracle as O  # noqa: E402
import waldo_amd  # noqa: E402
from waldo_amd import _lib, functional as WF  # noqa: E402

dev = torch.device('cuda:0')


def run(nl, h, w, f=2, generic=False, seed=3, gocc=True):
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=seed, sigma=0.1)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    _lib.load().waldo_set_debug_option(_lib.DEBUG_BWD_GENERIC, int(generic))
    l2 = layers.to(dev).requires_grad_()
    p2 = pts.to(dev).requires_grad_()
    o2 = occ.to(dev).requires_grad_(gocc)
    torch.manual_seed(0)
    w1 = torch.randn(f, 3, h, w, device=dev)
    w2 = torch.randn(f, nl, h, w, device=dev)
    rgb, alpha = WF.warp_composite(l2, p2, o2, tps.inverse_kernel, tps.basis_t, return_alpha=True)
    ((rgb * w1).sum() + (alpha * w2).sum()).backward()
    _lib.load().waldo_set_debug_option(_lib.DEBUG_BWD_GENERIC, 0)
    return l2.grad.cpu(), p2.grad.cpu(), (o2.grad.cpu() if gocc else None)


cases = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(17, 16, 32, 1), (17, 32, 64, 2), (12, 32, 64, 2), (8, 32, 64, 2)]
for (nl, h, w, f) in cases:
    for gocc in (True, False):
        a, pa, oa = run(nl, h, w, f, gocc=gocc)
        b, pb, ob = run(nl, h, w, f, generic=True, gocc=gocc)
        d = (a - b).abs()
        print(nl, h, w, f, 'gocc', gocc, 'max err', d.max().item(), 'gpts err', (pa - pb).abs().max().item(),
              'gocc err', (oa - ob).abs().max().item() if gocc else None, 'scale', b.abs().max().item())
        if d.max() > 1e-3 * b.abs().max():
            idx = (d > 1e-3 * b.abs().max()).nonzero()
            print('  bad count', idx.shape[0], 'of', d.numel(), 'nan', torch.isnan(a).sum().item())
            print('  by frame', sorted(collections.Counter(idx[:, 0].tolist()).items()))
            print('  by layer', sorted(collections.Counter(idx[:, 1].tolist()).items()))
            print('  by chan', sorted(collections.Counter(idx[:, 2].tolist()).items()))
            print('  rows', sorted(collections.Counter(idx[:, 3].tolist()).items())[:40])
            print('  cols', sorted(collections.Counter(idx[:, 4].tolist()).items())[:70])
            print('  first', idx[:5].tolist(), a[tuple(idx[0])].item(), b[tuple(idx[0])].item())
