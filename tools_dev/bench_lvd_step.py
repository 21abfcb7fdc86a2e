#!/usr/bin/env python3
"""dev: the warp-path part of one LVD training step at the reference's recipe
(scripts/cityscapes/train_lvd.sh: 128x256, no HD raster, 16 objects, 5 frames, ctx_mode "prev",
include_self, layout filtering with weighted classes; models/synthesizer.py:815-841) under
autograd: estimate_alpha_grid_occ -> decode_output -> loss -> backward.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = torch.device("cuda:0")
if "--lib" in sys.argv:  # A/B runs: another build of the library (tools_dev/run_lvd_ab.sh)
    from waldo_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.use_library(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
b = int(_pos[0]) if len(_pos) > 0 else 2
iters = int(_pos[1]) if len(_pos) > 1 else 10
from waldo_amd.tools.lvd_step import LvdStep  # noqa: E402  (the workload itself: what bench.py --config LVD times)
step = LvdStep(b, dev)
leaves, no, nl, t = step.leaves, step.opt.num_obj, step.num_lyt, step.frames


graph = "--graph" in sys.argv
for _ in range(3):
    step()
torch.cuda.synchronize()
if graph:
    # the whole step -- forward, loss, backward -- captured once and replayed (the library launches on the
    # current stream and never synchronises; its host-side index checks are skipped during capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        loss = step()
    run = g_.replay
else:
    run = step
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out_ = run()
    loss = out_ if out_ is not None else loss
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
print(json.dumps({"config": f"LVD recipe step: B={b} T={t} L={no + 1} Nl={nl} 128x256, ctx prev + include_self" + (", one HIP graph" if graph else ""),
                  "ms_per_step": round(ms, 3), "loss": float(loss),
                  "grads_finite": all(bool(torch.isfinite(x.grad).all()) for x in leaves)}))
