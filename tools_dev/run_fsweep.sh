#!/bin/bash
# dev: per-frame time of the fused entry points against the number of frames per call
# (does a record working set that fits the infinity cache run faster?)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_warper.py -q -k inverse_warp 2>&1 | tail -5
for f in 4 8 16 28 56 112; do
  echo "== F=$f"
  python tools_dev/ab_bench.py --shape $f,8,256,512 --iters 20 --rounds 2 tools_dev/_variants/cur.so 2>&1 | tail -3
done
