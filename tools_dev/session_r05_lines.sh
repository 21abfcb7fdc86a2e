#!/bin/bash
# dev: the bench lines of tools_dev/session_r05.sh alone (no profiler passes): a second sample on another box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lines2
python bench.py > gpurun_out/lines2/r05_bench.json 2> gpurun_out/lines2/err.txt
for c in C2 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/lines2/r05_bench_$c.json 2>> gpurun_out/lines2/err.txt; done
python bench.py --config C5 --pipeline --steps 8 --warmup 2 > gpurun_out/lines2/r05_bench_C5_pipeline.json 2>> gpurun_out/lines2/err.txt
python bench.py --config C5 --pipeline --steps 8 --warmup 2 --motion wild > gpurun_out/lines2/r05_bench_C5_pipeline_wild.json 2>> gpurun_out/lines2/err.txt
python bench.py --config C4 --pipeline --steps 8 --warmup 2 > gpurun_out/lines2/r05_bench_C4_pipeline.json 2>> gpurun_out/lines2/err.txt
python bench.py --config C4 --pipeline --steps 8 --warmup 2 --motion wild > gpurun_out/lines2/r05_bench_C4_pipeline_wild.json 2>> gpurun_out/lines2/err.txt
python bench.py --config LVD --steps 40 > gpurun_out/lines2/r05_bench_LVD.json 2>> gpurun_out/lines2/err.txt
python tools_dev/strong_projection.py C5 > gpurun_out/lines2/r05_strong_projection_C5.json 2>> gpurun_out/lines2/err.txt
python tools_dev/strong_projection.py C4 > gpurun_out/lines2/r05_strong_projection_C4.json 2>> gpurun_out/lines2/err.txt
echo done
