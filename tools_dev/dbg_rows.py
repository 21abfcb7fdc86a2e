"""dev: the lane-layer backward kernels (csrc/flow_ctx_bwd_rows.hip.h) against the per-pixel ones, entry point by entry point."""
import sys
import torch
sys.path.insert(0, '.')
from waldo_amd import _lib, functional as WF

dev = torch.device("cuda:0")
torch.manual_seed(0)
b, t, tw, nl, ncls, h, w, s = 2, 3, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 17, 6, 8, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 4
hd, wd = h * s, w * s
alpha_lr = torch.rand(b * tw, nl, h, w, device=dev)
inp = torch.randn(b, t, 3 + ncls, hd, wd, device=dev)
dist = torch.softmax(torch.randn(b, nl - 1, ncls, device=dev), -1)
occ = torch.rand(b, t, nl, nl, device=dev) * 0.8
res = {}
for px in (0, 1):
    pass  # (round 6: the lane-layer kernels are a VARIANT build, tools_dev/dropped/README.md; run this once per library with --lib)
    a, d, o = alpha_lr.clone().requires_grad_(), dist.clone().requires_grad_(), occ.clone().requires_grad_()
    a01, alpha = WF.flow_ctx_alpha(a, inp, d, o, tw, 3, s)
    torch.manual_seed(1)
    (a01 * torch.randn_like(a01)).sum().backward()
    res[px] = (a.grad, d.grad, o.grad)
for name, x, y in zip(("g_alpha_lr", "g_dist", "g_occ"), res[0], res[1]):
    print(name, "max |rows - pixel|", (x - y).abs().max().item(), "scale", y.abs().max().item())
print("g_dist rows\n", res[0][1][0, :3], "\npixel\n", res[1][1][0, :3])
