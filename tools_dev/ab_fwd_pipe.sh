set -e
mkdir -p gpurun_out/r05c
for i in 1 2 3; do
  for opt in "" "--lib tools_dev/_variants/pipe.so" "--lib tools_dev/_variants/nomap.so" "--lib tools_dev/_variants/ahead3.so" "--lib tools_dev/_variants/ahead3.so --lib tools_dev/_variants/pipe.so"; do
    python bench.py --no-cpu-baseline --steps 100 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('C3 [$opt]', d['ms_per_step'], 'fwd', k['waldo_warp_composite_fwd']['ms'], 'bwd', k['waldo_warp_composite_bwd']['ms'])" | tee -a gpurun_out/r05c/ab2.txt
  done
done
