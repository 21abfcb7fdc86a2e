# dev: C5 pipeline (calibrated and wild) with the product library and each variant library given, interleaved on one box
set -e
mkdir -p gpurun_out/ab_c5
for i in 1 2; do
  for lib in "" "$@"; do
    opt=""; [ -n "$lib" ] && opt="--lib $lib"
    for motion in calibrated wild; do
    python bench.py --config C5 --pipeline --motion $motion --steps 10 --warmup 2 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['pipeline']['entry_points']
print('C5 $motion [$lib]', d['ms_per_step'], 'fwf', e['waldo_frame_warp_fuse_raw_fwd']['ms_per_step'], 'fcw', e['waldo_flow_ctx_warp_raw_fwd']['ms_per_step'])" | tee -a gpurun_out/ab_c5/ab.txt
    done
  done
done
