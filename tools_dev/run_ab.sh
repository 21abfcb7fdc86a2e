#!/bin/bash
# dev helper: A/B the variant libraries under waldo_amd/lib/abl/ on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1200 python tools_dev/ab_bench.py --rounds ${ROUNDS:-2} "$@" waldo_amd/lib/abl/*.so 2>&1 | tee gpurun_out/ab/ab.log | grep round
