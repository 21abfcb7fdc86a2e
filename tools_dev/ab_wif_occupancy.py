"""dev: the WIF training step's no-grad decode with and without the layer-occupancy skip of the flow pass (round 6),
interleaved in one process; per-entry-point ms from event pairs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd import _lib  # noqa: E402
from waldo_amd.tools.wif_step import WifStep  # noqa: E402

dev = torch.device("cuda:0")
step = WifStep(2, dev)
for _ in range(10):
    step.decode()
for rnd in range(3):
    for occ in (False, True):
        step.warper.layer_occupancy = occ
        for _ in range(5):
            step.decode()
        torch.cuda.synchronize()
        with _lib.KernelTimer() as kt:
            for _ in range(30):
                step.decode()
            torch.cuda.synchronize()
        t = kt.summary()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            step.decode()
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} occupancy {occ!s:5}  decode {e0.elapsed_time(e1) / 30:.4f} ms   alpha {t['waldo_flow_ctx_alpha_fwd'][1]:.4f}"
              f"  warp {t['waldo_flow_ctx_warp_raw_fwd'][1]:.4f}  frame warp {t['waldo_frame_warp_fuse_raw_fwd'][1]:.4f}")
