"""dev: what strong scaling of the C4 / C5 pipeline would be on W ranks, projected on ONE GPU.

    python tools_dev/strong_projection.py C5 [clips]

Every rank's share of the job (demo.predict_sharded, shard = (r, W)) is timed one after the other on this GPU, eagerly
and replayed from one HIP graph; the projection for W ranks is T(1) / max_r T(r, W) -- the all-gather of the predicted
frames (overlapped with the next step in bench.py) and the xGMI fabric are NOT in it.  Prints one JSON object."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
from waldo_amd.tools import pipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
clips = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if name == "C5" else 8)
dev = torch.device("cuda:0")
pipe = pipeline.Pipeline(name, clips, dev, shard=(0, 1))


def timeit(fn, n=10):
    """The median of five blocks of n calls after five warm-up calls (the first block of a new shard shape has shown
    one-off allocator work -- rank 0 of 2: 15 - 18 ms against 11 -- that is no part of a steady step)."""
    for _ in range(5):
        fn()
    ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ms)[2]


res = {"config": name, "clips": clips, "frames": clips * pipe.frames, "worlds": {}}
with torch.no_grad():
    for world in (1, 2, 4, 8):
        eager, graph = [None] * world, [None] * world
        for sweep in range(2):  # every rank's share twice, the sweeps apart in time: the smaller median counts (a rank's
            for r in range(world):  # first visit has shown one-off allocator work and clock ramps)
                pipe.shard = (r, world)
                e = round(timeit(lambda: pipe()), 3)
                g = pipe.graphed()
                gr = round(timeit(lambda: g(*g.inputs)), 3)
                del g
                torch.cuda.empty_cache()
                eager[r] = e if eager[r] is None else min(eager[r], e)
                graph[r] = gr if graph[r] is None else min(graph[r], gr)
        res["worlds"][world] = {"eager_ms_per_rank": eager, "graph_ms_per_rank": graph}
t1e, t1g = res["worlds"][1]["eager_ms_per_rank"][0], res["worlds"][1]["graph_ms_per_rank"][0]
for world, row in res["worlds"].items():
    row["rank_max_over_min_eager"] = round(max(row["eager_ms_per_rank"]) / min(row["eager_ms_per_rank"]), 3)
    row["projected_speedup_eager"] = round(t1e / max(row["eager_ms_per_rank"]), 2)
    row["projected_speedup_graph"] = round(t1g / max(row["graph_ms_per_rank"]), 2)
    row["projected_speedup_graph_over_eager_1gpu"] = round(t1e / max(row["graph_ms_per_rank"]), 2)
print(json.dumps(res))
