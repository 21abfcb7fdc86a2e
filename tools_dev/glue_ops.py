#!/usr/bin/env python3
"""dev: which framework (non-library) GPU kernels one step launches, grouped by the aten op and its input shapes
(torch.profiler, record_shapes), so the glue around the library calls can be priced op by op.
    python tools_dev/glue_ops.py LVD [clips]          # the LVD-recipe training step
    python tools_dev/glue_ops.py C5 [clips] [motion]  # the C4 / C5 predict pipeline"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "LVD"
if what == "LVD":
    from waldo_amd.tools.lvd_step import LvdStep
    step = LvdStep(int(sys.argv[2]) if len(sys.argv) > 2 else 2, dev)
else:
    from waldo_amd.tools.pipeline import Pipeline
    pipe = Pipeline(what, int(sys.argv[2]) if len(sys.argv) > 2 else 4, dev, motion=sys.argv[3] if len(sys.argv) > 3 else "calibrated")

    def step():
        with torch.no_grad():
            return pipe()
for _ in range(3):
    step()
torch.cuda.synchronize()
n = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(n):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / n, e.count / n, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
print("# self device time per step, us; count per step; op or kernel; input shapes (an op and its kernel both appear)")
for dt, c, k, s in rows:
    print(f"{dt:9.1f} {c:6.1f}  {k[:90]:90s} {s}")
