#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 300 python tools_dev/dbg_bwd.py > gpurun_out/s2/dbg.log 2>&1
cat gpurun_out/s2/dbg.log | tail -60
timeout 600 bash tools_dev/prof.sh 2>&1 | grep waldo
