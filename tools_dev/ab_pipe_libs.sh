# dev: C5 and C4 pipelines with the product library and each variant library given, interleaved on one box
#   bash tools_dev/ab_pipe_libs.sh tools_dev/_variants/a.so ...
set -e
mkdir -p gpurun_out/ab_pipe
for i in 1 2; do
  for lib in "" "$@"; do
    opt=""; [ -n "$lib" ] && opt="--lib $lib"
    for cfg in C5 C4; do
    python bench.py --config $cfg --pipeline --steps 10 --warmup 2 --no-cpu-baseline $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
g=lambda k: e.get(k,{}).get('ms_per_step')
print('$cfg [$lib]', d['ms_per_step'], 'fwf', g('waldo_frame_warp_fuse_raw_fwd'), 'fcw', g('waldo_flow_ctx_warp_raw_fwd'), 'fca', g('waldo_flow_ctx_alpha_fwd'), 'iw', g('waldo_inverse_warp_fwd'), 'gs', g('waldo_grid_sample2d_ex_fwd'))" | tee -a gpurun_out/ab_pipe/ab.txt
    done
  done
done
