#!/bin/bash
# dev helper: rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/prof_latest/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_latest
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_latest'
for f in glob.glob(out + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.3:
            print('%-70s calls %4s avg %10.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
tail -1 $OUT/log.txt | cut -c1-200
