# dev: LVD-recipe step with the product library and with each variant library given, interleaved on one box
#   bash tools_dev/ab_lvd_libs.sh tools_dev/_variants/a.so tools_dev/_variants/b.so
set -e
mkdir -p gpurun_out/ab_lvd
for i in 1 2; do
  for lib in "" "$@"; do
    opt=""; [ -n "$lib" ] && opt="--lib $lib"
    python bench.py --config LVD --steps 200 --warmup 20 --no-cpu-baseline $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
g=lambda k: e.get(k,{}).get('ms_per_step')
print('LVD [$lib] eager', d['ms_per_step_eager'], 'graph', d['ms_per_step_graph_replay'], 'warp_bwd', g('waldo_flow_ctx_warp_bwd'), 'alpha_bwd', g('waldo_flow_ctx_alpha_bwd'), 'gs_bwd', g('waldo_grid_sample2d_bwd'), 'gs_ex_bwd', g('waldo_grid_sample2d_ex_bwd'), 'iw_fwd', g('waldo_inverse_warp_fwd'),  'iw_bwd', g('waldo_inverse_warp_bwd'), 'fwf_bwd', g('waldo_frame_warp_fuse_bwd'), 'fwf', g('waldo_frame_warp_fuse_fwd'))" | tee -a gpurun_out/ab_lvd/ab_libs.txt
  done
done
