import sys, torch
sys.path.insert(0, '.')
from oracle import wif_oracle as O
import waldo_amd
from waldo_amd import functional as WF
dev = torch.device('cuda:0')
def run(nl, h, w, f=2, generic=False, seed=3):
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=seed, sigma=0.1)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    WF._FORCE_GENERIC_BWD = generic
    l2 = layers.to(dev).requires_grad_(); p2 = pts.to(dev).requires_grad_()
    torch.manual_seed(0)
    w1 = torch.randn(f, 3, h, w, device=dev)
    rgb = WF.warp_composite(l2, p2, occ.to(dev), tps.inverse_kernel, tps.basis_t)
    (rgb * w1).sum().backward()
    return l2.grad.cpu(), p2.grad.cpu()
for (nl, h, w) in [(8, 32, 48), (8, 64, 128), (6, 32, 48), (8, 8, 64), (8, 16, 64), (8, 32, 64)]:
    a, pa = run(nl, h, w)
    b, pb = run(nl, h, w, generic=True)
    d = (a - b).abs()
    print(nl, h, w, 'max err', d.max().item(), 'gpts err', (pa - pb).abs().max().item(), 'scale', b.abs().max().item())
    if d.max() > 1e-3:
        idx = (d > 1e-3).nonzero()
        print('  bad count', idx.shape[0], 'of', d.numel())
        import collections
        print('  by layer', collections.Counter(idx[:, 1].tolist()))
        print('  by chan', collections.Counter(idx[:, 2].tolist()))
        print('  rows', sorted(collections.Counter(idx[:, 3].tolist()).items())[:40])
        print('  first', idx[:5].tolist(), a[tuple(idx[0])].item(), b[tuple(idx[0])].item())
