import os, sys, torch
sys.path.insert(0, os.getcwd())
from waldo_amd import functional as WF
from waldo_amd.tools.wif_step import WifStep
dev = torch.device("cuda:0")
step = WifStep(2, dev)
cap = {}
orig = WF.flow_ctx_alpha
def spy(*a, **k):
    r = orig(*a, **k); cap["bits"] = r[2] if len(r) > 2 else None; cap["a01"] = r[0]; return r
WF.flow_ctx_alpha = spy
step.decode(); torch.cuda.synchronize()
bits = cap["bits"]; a01 = cap["a01"]
n, hd, ns = bits.shape
pc = torch.zeros_like(bits)
for l in range(17): pc += (bits >> l) & 1
print("layers non-zero per (row, segment) word: mean %.2f" % pc.float().mean().item(), "frac of a01 != 0: %.3f" % (a01 != 0).float().mean().item())
# OR over a neighbourhood of +-40 rows and +-1 segment (flows of ~30 px): what a tile would see
b = torch.stack([((bits >> l) & 1).float() for l in range(17)], 1)        # n 17 hd ns
nb = torch.nn.functional.max_pool2d(b, (81, 3), 1, (40, 1))
print("layers present in a (81 rows x 3 segments) neighbourhood: mean %.2f" % nb.sum(1).mean().item())
nb = torch.nn.functional.max_pool2d(b, (33, 3), 1, (16, 1))
print("layers present in a (33 rows x 3 segments) neighbourhood: mean %.2f" % nb.sum(1).mean().item())
nb = torch.nn.functional.max_pool2d(b, (17, 1), 1, (8, 0))
print("layers present in a (17 rows x 1 segment) neighbourhood: mean %.2f" % nb.sum(1).mean().item())
