# dev: C5 pipeline, product library against variant libraries, interleaved on one box:  bash tools_dev/ab_c5_libs.sh a.so b.so ...
# prints ms_per_step and the flow_ctx_warp / frame_warp_fuse entry points per run
for round in 1 2 3; do
  for lib in "" "$@"; do
    if [ -z "$lib" ]; then arg=""; name=product; else arg="--lib $lib"; name=$(basename $lib); fi
    python bench.py --config C5 --pipeline --steps 10 --warmup 3 ${MOTION:+--motion $MOTION} $arg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['pipeline']['entry_points']
print('round $round %-14s step %7.3f ms  flow_ctx_warp %.4f  frame_warp_fuse %.4f  flow_ctx_alpha %.4f' % ('$name', d['ms_per_step'], t['waldo_flow_ctx_warp_raw_fwd']['ms_per_step'], t['waldo_frame_warp_fuse_raw_fwd']['ms_per_step'], t['waldo_flow_ctx_alpha_fwd']['ms_per_step']))"
  done
done
