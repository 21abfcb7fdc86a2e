import collections, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, '.')
from waldo_amd.tools import pipeline
pipe = pipeline.Pipeline("C5", 4, torch.device("cuda:0"))
with torch.no_grad():
    pipe(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        pipe(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total > 0 and ev.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::clone"):
        k = (ev.name, str(ev.input_shapes)[:90]); agg[k][0] += 1; agg[k][1] += ev.device_time_total / 1e3
for (n, shp), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{ms:8.3f} ms x{c:<2d} {n:12s} {shp}")
