"""dev: statistics of optical-flow files (.flo, Middlebury format: data/base_dataset.py:185-208) -- the yardstick
for the stand-in motion of waldo_amd/tools/demo.py (BG_MOTION).  Per file: mean / max |flow| in pixels and the
quantiles of the local stretch |d ix / dx|, |d iy / dy| of the sample position ix = x + flow_x a backward warp by
this flow would use.

    python tools_dev/flo_stats.py /root/reference/datasets/demo_cityscapes/leftImg8bit_sequence_raft_128/val/munster"""
import glob
import os
import sys

import numpy as np

rows = []
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "*.flo"))):
        with open(f, "rb") as fh:
            fh.read(4)
            w, h = np.frombuffer(fh.read(8), np.int32)
            fl = np.frombuffer(fh.read(), np.float32).reshape(h, w, 2)
        dx = np.abs(1 + np.diff(fl[..., 0], axis=1))
        dy = np.abs(1 + np.diff(fl[..., 1], axis=0))
        rows.append((w, h, np.abs(fl[..., 0]).mean(), np.abs(fl[..., 0]).max(), np.abs(fl[..., 1]).mean(),
                     np.abs(fl[..., 1]).max(), *np.quantile(dx, [.1, .5, .9, .99]), *np.quantile(dy, [.1, .5, .9, .99])))
r = np.array(rows)
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print(f"{len(r)} files of {int(r[0, 0])} x {int(r[0, 1])}")
print("|fx| mean / max px:", r[:, 2].mean().round(2), r[:, 3].max().round(2), " |fy|:", r[:, 4].mean().round(2), r[:, 5].max().round(2))
print("|d ix/dx| quantiles 10 / 50 / 90 / 99 %: mean over files", r[:, 6:10].mean(0), " worst file", r[:, 6:10].max(0))
print("|d iy/dy|                              : mean over files", r[:, 10:14].mean(0), " worst file", r[:, 10:14].max(0))
