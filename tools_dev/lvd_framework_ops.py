"""dev: the framework operators of one LVD-recipe step (fills, copies, adds ...) with their shapes and the autograd
node that ran them (torch profiler): what the step pays outside the library."""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from waldo_amd.tools.lvd_step import LvdStep  # noqa: E402

step = LvdStep(2, torch.device("cuda:0"))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = sorted(prof.events(), key=lambda e: e.time_range.start)
# parent autograd node of every op: the innermost enclosing event whose name mentions Backward / a Function
stack = []
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in evs:
    while stack and stack[-1].time_range.end < ev.time_range.start:
        stack.pop()
    parent = next((s.name for s in reversed(stack) if "Backward" in s.name or s.name.startswith("autograd::")), "forward")
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::sum",
                   "aten::mean", "aten::pow", "aten::clone") and ev.device_time_total > 0:
        k = (ev.name, parent, str(ev.input_shapes)[:60])
        agg[k][0] += 1
        agg[k][1] += ev.device_time_total
    stack.append(ev)
for (n, par, shp), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{us:8.1f} us x{c:<2d} {n:12s} {par[:48]:48s} {shp}")
