"""HIP-graph replay for launch-bound calls of the path (inference).

At small shapes (BASELINE config C2: 8 frames of 128x128, L = 8) one forward of the fused path is
two ~10 us kernels behind ~30 us of Python / ctypes / launch overhead.  The C-ABI library launches
on whatever stream the caller passes and never synchronises, so a call sequence can be captured
once into a HIP graph (``torch.cuda.CUDAGraph``, which is hipGraph on ROCm) and replayed with a
single launch: 54.6 us -> 25.9 us per C2 forward on MI355X (tools_dev/bench_graph.py).
"""
import torch

from . import _lib


class GraphedCall:
    """Captures ``fn(*inputs)`` (forward only, fixed shapes) and replays it.

    ``inputs`` are copied into static buffers on every call (``self.inputs``: pass those themselves
    to skip the copy); the returned tensors are the graph's static outputs (clone them if they must
    survive the next call).

    Frame indices (``ctx_ts`` / ``pred_ts``) are validated by the kernels on the device, in a replay as in an eager
    call: what the replays so far have reported is raised after each replay (no synchronisation; ``check()`` waits
    for the stream first)."""

    def __init__(self, fn, *example_inputs, warmup=3):
        self._static = [x.clone() if torch.is_tensor(x) else x for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # allocator / lazy-init work must not happen during capture
                fn(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        self.inputs = tuple(self._static)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph), torch.no_grad():
            self._out = fn(*self._static)

    def __call__(self, *inputs):
        if len(inputs) != len(self._static):
            raise ValueError(f"expected {len(self._static)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static, inputs):
            if torch.is_tensor(dst):
                if dst.shape != src.shape or dst.dtype != src.dtype:
                    raise ValueError(f"graphed call was captured for {tuple(dst.shape)} {dst.dtype}, "
                                     f"got {tuple(src.shape)} {src.dtype}")
                if src is not dst:  # identity only: a view or a recycled pointer may alias other contents
                    dst.copy_(src)
        self._graph.replay()
        _lib.IndexStatus.check_all()
        return self._out

    def check(self):
        """Wait for the replays queued so far and raise if a kernel met a frame index outside its time axis."""
        _lib.IndexStatus.check_all(sync=True)
