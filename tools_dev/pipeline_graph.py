"""dev: the C4 / C5 pipeline eagerly against ONE HIP graph of the whole predict() (are the ~200 launches of a step
queued fast enough, or does the GPU wait for the host between kernels?)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from waldo_amd.graphs import GraphedCall  # noqa: E402
from waldo_amd.tools import demo, pipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
clips = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if name == "C5" else 8)
pipe = pipeline.Pipeline(name, clips, torch.device("cuda:0"))


def run(vid, lyt):
    out = demo.predict(pipe.opt, pipe.warper, pipe.wif, vid, lyt, pipe.net, pipe.ctx_len)
    return out["inp_pred_vid"], out["inp_rec_vid"], out["pred_flow"]


def timeit(fn, n=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    eager = [x.clone() for x in run(pipe.vid, pipe.lyt)]
    t_eager = timeit(lambda: run(pipe.vid, pipe.lyt))
    g = GraphedCall(run, pipe.vid, pipe.lyt)
    outs = g(*g.inputs)
    same = all(torch.equal(a, b) for a, b in zip(eager, outs))
    t_graph = timeit(lambda: g(*g.inputs))
print(f"{name} x{clips}: eager {t_eager:.3f} ms, one HIP graph {t_graph:.3f} ms, same bits {same}, "
      f"reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB")
