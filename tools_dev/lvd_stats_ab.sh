#!/bin/bash
# dev helper: per-kernel times of the LVD-recipe step with each variant library under tools_dev/_variants/
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
for so in $GRAFT_REPO_ROOT/tools_dev/_variants/*.so; do
  n=$(basename $so .so)
  rm -rf /tmp/st_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$n -- python3 $GRAFT_REPO_ROOT/tools_dev/bench_lvd_step.py --lib $so 2 10 > /tmp/st_$n.log 2>&1
  echo "== $n"
  python3 - <<PY
import csv, glob
for f in glob.glob('/tmp/st_$n/*/*kernel_stats.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'waldo' in r['Name']]
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:14]:
        print(f"{r['Name'].split('(')[0][-50:]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
