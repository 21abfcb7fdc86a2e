# dev: LVD-recipe step, the product library with / without a debug option and variant libraries, interleaved on one box
#   bash tools_dev/ab_lvd_opts.sh "--lib tools_dev/_variants/rows.so" "--lib tools_dev/_variants/a.so" ...
set -e
mkdir -p gpurun_out/ab_lvd
for i in 1 2; do
  for opt in "" "$@"; do
    python bench.py --config LVD --steps 200 --warmup 20 --no-cpu-baseline $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
g=lambda k: e.get(k,{}).get('ms_per_step')
print('LVD [$opt] eager', d['ms_per_step_eager'], 'graph', d['ms_per_step_graph_replay'], 'warp_bwd', g('waldo_flow_ctx_warp_bwd'), 'alpha_bwd', g('waldo_flow_ctx_alpha_bwd'))" | tee -a gpurun_out/ab_lvd/ab_opts.txt
  done
done
