# dev: where does the staged forward differ from the plain one?  (two processes via env)
import os, sys, torch
sys.path.insert(0, '.')
from oracle import wif_oracle as O
import waldo_amd
from waldo_amd import functional as WF
dev = torch.device('cuda:0')
f, nl, h, w, sig = 1, 24, 16, 16, 0.1
layers, pts, occ, inv, rep = O.make_synthetic(f, nl, h, w, seed=1, sigma=sig)
tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
occ = torch.zeros_like(occ)   # alpha_l' = alpha_l: isolates sampling per layer
rgb, alpha = WF.warp_composite(layers.to(dev), pts.to(dev), occ.to(dev), tps.inverse_kernel, tps.basis_t, return_alpha=True)
out = alpha.cpu()
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    d = (out - other).abs()[0]
    print('per-layer max diff', [round(x, 3) for x in d.flatten(1).max(1).values.tolist()])
    l = int(d.flatten(1).max(1).values.argmax())
    print('layer', l); print((d[l] > 1e-6).int())
torch.save(out, sys.argv[1])
