"""dev: which host lines of the C5 / C4 pipeline launch the framework's copy / fill / reduce kernels?
(torch profiler, grouped by the innermost waldo_amd source line on the stack)"""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from waldo_amd.tools import pipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
pipe = pipeline.Pipeline(name, 4 if name == "C5" else 8, torch.device("cuda:0"))
with torch.no_grad():
    pipe()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        pipe()
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::copy_", "aten::fill_", "aten::amax", "aten::amin", "aten::cat", "aten::clone",
                       "aten::add", "aten::add_", "aten::masked_fill_", "aten::index_put_", "aten::gt", "aten::sub"):
        continue
    where = "?"
    for fr in ev.stack:
        if "waldo_amd" in fr:
            where = fr.split("waldo_amd/")[-1]
            break
    k = (ev.name, where)
    agg[k][0] += 1
    agg[k][1] += ev.device_time_total / 1e3
for (n, w), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{ms:8.3f} ms  x{c:<3d} {n:18s} {w}")
