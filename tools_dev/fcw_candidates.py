"""dev: how many layers can be present in a flow_ctx_warp tile under the pipeline's real masks: per 16 x 64 HD tile
(x 4: 6 x 18 low-resolution cells) the count of layers with ANY object-mask cell above 0.5 (+ the background)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from waldo_amd import functional as WF  # noqa: E402
from waldo_amd.tools import pipeline  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
dev = torch.device("cuda:0")
seen = []
orig = WF.flow_ctx_warp_into_raw


def spy(flow_lr, isobj_lr, *a, **k):
    seen.append((isobj_lr.detach(), a[5]))   # masks, scale
    return orig(flow_lr, isobj_lr, *a, **k)


WF.flow_ctx_warp_into_raw = spy
import waldo_amd.nets.lvd as lvd  # noqa: E402
if hasattr(lvd.WF, "flow_ctx_warp_into_raw"):
    lvd.WF.flow_ctx_warp_into_raw = spy
pipe = pipeline.Pipeline(cfg, 2, dev)
with torch.no_grad():
    pipe()
torch.cuda.synchronize()
for isobj, scale in seen:
    m, no, h, w = isobj.shape
    rows = 16 // scale if scale >= 4 else 8 // scale   # tall tile rows in cells (x4: 16 px; x2: 8 px)
    cols = 64 // scale
    hit = (isobj > 0.5).float()
    # any cell of the (rows + 2) x (cols + 2) window around a tile
    pooled = F.max_pool2d(F.pad(hit, (1, 1, 1, 1)), kernel_size=(rows + 2, cols + 2), stride=(rows, cols))
    kc = pooled.sum(dim=1) + 1   # + background
    q = torch.bincount(kc.flatten().long(), minlength=no + 2).float()
    q = q / q.sum()
    print(f"{cfg}: masks {tuple(isobj.shape)} x{scale}: share of tiles by candidate count 1.. : " +
          " ".join(f"{v:.3f}" for v in q[1:].tolist()) + f" | <=4: {q[:5].sum():.3f}  <=8: {q[:9].sum():.3f}  mean {kc.mean():.2f}")
