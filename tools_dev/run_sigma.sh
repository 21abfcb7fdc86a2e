#!/bin/bash
# dev: fused fwd / bwd time against the control-point noise (footprint boxes past the LDS cap fall
# back to per-tap gathers)
for s in 0.02 0.05 0.1 0.15 0.2 0.3; do
  echo "== sigma=$s"
  python tools_dev/ab_bench.py --sigma $s --iters 10 --rounds 1 waldo_amd/lib/libwaldo_hip.so 2>&1 | tail -1
done
