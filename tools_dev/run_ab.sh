#!/bin/bash
# dev helper: A/B the variant libraries under tools_dev/_variants/ on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1200 python tools_dev/ab_bench.py --rounds ${ROUNDS:-2} "$@" tools_dev/_variants/*.so 2>&1 | tee gpurun_out/ab/ab.log | grep round
