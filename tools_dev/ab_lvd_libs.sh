# dev: LVD-recipe step with the product library and with each variant library given, interleaved on one box
set -e
mkdir -p gpurun_out/ab_lvd
for i in 1 2; do
  for lib in "" "$@"; do
    opt=""; [ -n "$lib" ] && opt="--lib $lib"
    python bench.py --config LVD --steps 200 --warmup 20 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
print('LVD [$lib]', d['ms_per_step'], 'warp_bwd', e['waldo_flow_ctx_warp_bwd']['ms_per_step'], 'alpha_bwd', e['waldo_flow_ctx_alpha_bwd']['ms_per_step'], 'gs_bwd', e['waldo_grid_sample2d_bwd']['ms_per_step'])" | tee -a gpurun_out/ab_lvd/ab_libs.txt
  done
done
