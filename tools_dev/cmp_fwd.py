# dev: LDS-staged forward vs plain forward (must be bit-identical); run as two processes via env
import os, sys, torch
sys.path.insert(0, '.')
from oracle import wif_oracle as O
import waldo_amd
from waldo_amd import functional as WF
dev = torch.device('cuda:0')
outs = {}
for (f, nl, h, w, sig) in [(3, 8, 64, 128, 0.05), (2, 8, 256, 512, 0.05), (2, 5, 32, 64, 0.3), (2, 8, 16, 16, 0.1), (1, 3, 8, 12, 0.6),
                          (1, 20, 12, 12, 0.1), (1, 24, 12, 12, 0.1), (1, 20, 16, 16, 0.1), (1, 32, 8, 16, 0.1), (1, 18, 16, 16, 0.1), (2, 12, 32, 32, 0.1), (1, 17, 12, 12, 0.1), (1, 9, 12, 12, 0.1)]:
    layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=1, sigma=sig)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    rgb, alpha = WF.warp_composite(layers.to(dev), pts.to(dev), occ.to(dev), tps.inverse_kernel, tps.basis_t, return_alpha=True)
    outs[(f, nl, h, w, sig)] = (rgb.cpu(), alpha.cpu())
torch.save(outs, sys.argv[1])
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    for k in outs:
        d = max((outs[k][0] - other[k][0]).abs().max().item(), (outs[k][1] - other[k][1]).abs().max().item())
        print(k, 'max diff vs plain', d)
