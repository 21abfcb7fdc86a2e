#!/bin/bash
# dev helper: the three profiles a round commits under profiles/ -- kernel stats, HBM traffic (PMC),
# issue-side counters -- of the default bench command.   usage: tools_dev/profile_round.sh TAG
TAG=${1:-latest}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT gpurun_out/prof_latest gpurun_out/traffic gpurun_out/pmc_issue; mkdir -p $OUT
[ -x tools_dev/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o tools_dev/fetch_calib tools_dev/fetch_calib.hip
python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
bash tools_dev/prof.sh > $OUT/kernel_stats.txt 2>&1
cp gpurun_out/prof_latest/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv
tail -1 gpurun_out/prof_latest/log.txt > $OUT/bench_under_rocprof.json
grep waldo $OUT/kernel_stats.txt
python3 tools_dev/traffic.py > $OUT/traffic.log 2>&1
cp gpurun_out/traffic/traffic.json $OUT/traffic.json
python3 -c "import json; d=json.load(open('$OUT/traffic.json')); print(d['bytes_per_launch']); print(json.dumps(d['bytes_per_dispatch_by_kernel'], indent=1))"
bash tools_dev/pmc_issue.sh > $OUT/pmc_issue_counters.txt 2>&1
grep -A25 "px16\|fwd_lds\|splat" $OUT/pmc_issue_counters.txt | head -90
