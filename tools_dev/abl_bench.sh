#!/bin/bash
# dev helper: time bench.py with each ablation library (waldo_amd/lib/abl/*.so)
for so in "" waldo_amd/lib/abl/*.so; do
  if [ -n "$so" ]; then export WALDO_HIP_LIB=$PWD/$so; else unset WALDO_HIP_LIB; fi
  echo "== ${so:-baseline}"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('step %.3f ms  fwd %.3f ms  bwd %.3f ms' % (d['ms_per_step'], k['waldo_warp_composite_fwd']['ms'], k['waldo_warp_composite_bwd']['ms']))"
done
