"""dev: per-step wall time of the LVD step (are there periodic stalls: allocator, garbage collector, index checks?)"""
import gc
import sys
import time

import torch

sys.path.insert(0, '.')
from waldo_amd.tools.lvd_step import LvdStep  # noqa: E402

device = torch.device("cuda:0")
step = LvdStep(2, device, seed=0)
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.disable()
ts = []
for i in range(120):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "num_alloc_retries", torch.cuda.memory_stats().get("num_alloc_retries"),
      "segments", torch.cuda.memory_stats().get("segment.all.current"))
