"""dev: what the layers of the C5 pipeline cost flow_ctx_warp: the same recipe with 3 / 7 / 11 objects (L = 4 / 8 / 12),
kernel time from the torch profiler.  Objects are small: the additional layers are absent from most wavefronts, so the
difference is (mostly) what an ABSENT layer costs -- its mask test, its bookkeeping and its constant output plane."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from waldo_amd.tools import pipeline  # noqa: E402

dev = torch.device("cuda:0")
base = pipeline.RECIPES["C5"]
for no in (3, 7, 11):
    pipeline.RECIPES["C5"] = base[:4] + (no,) + base[5:]
    pipe = pipeline.Pipeline("C5", 4, dev)
    with torch.no_grad():
        pipe(); pipe(); torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            pipe(); torch.cuda.synchronize()
    tot = {}
    for ev in prof.events():
        if ev.device_time_total > 0:
            k = ev.name.split('(')[0][-60:]
            tot[k] = tot.get(k, 0.0) + ev.device_time_total / 1e3
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:4]
    print(f"objects {no:2d}: " + "  ".join(f"{k[-34:]} {v:6.2f} ms" for k, v in top))
    del pipe
    torch.cuda.empty_cache()
