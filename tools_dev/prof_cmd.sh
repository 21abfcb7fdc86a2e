#!/bin/bash
# dev helper: rocprofv3 kernel stats of an arbitrary python script:  tools_dev/prof_cmd.sh NAME script.py [args]
cd /tmp && export TMPDIR=/tmp
name=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
rm -rf $OUT; mkdir -p $OUT
script=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $script "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:${TOPN:-16}]:
        print('%-86s calls %4s avg %9.1f us  %5.1f%%' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
