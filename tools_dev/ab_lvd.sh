# dev: LVD-recipe step, product kernels against a debug option (or a variant library: tools_dev/dropped/README.md), interleaved on one box
set -e
OPT=${1:---lib tools_dev/_variants/rows.so}
mkdir -p gpurun_out/ab_lvd
for i in 1 2 3; do
  for opt in "" "$OPT"; do
    python bench.py --config LVD --steps 200 --warmup 20 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
print('LVD [$opt]', d['ms_per_step'], 'warp_bwd', e['waldo_flow_ctx_warp_bwd']['ms_per_step'], 'alpha_bwd', e['waldo_flow_ctx_alpha_bwd']['ms_per_step'], 'gs_bwd', e['waldo_grid_sample2d_bwd']['ms_per_step'])" | tee -a gpurun_out/ab_lvd/ab.txt
  done
done
