#!/usr/bin/env python3
"""dev: where the HOST time of one LVD-recipe step goes (cProfile over 100 steps; the GPU is not waited for until the
end, so the wall time of the loop is the host's launch time when the host is the bottleneck).
    python tools_dev/lvd_host_profile.py > gpurun_out/lvd_host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd.tools.lvd_step import LvdStep  # noqa: E402

dev = torch.device("cuda:0")
step = LvdStep(2, dev)
for _ in range(5):
    step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host loop {t_host / n * 1e3:.3f} ms per step, with the GPU drained {t_all / n * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    print(buf.getvalue())
