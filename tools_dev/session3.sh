#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
timeout 300 python tools_dev/dbg_bwd.py 17,16,32,1 17,32,64,2 9,32,64,2 > gpurun_out/s3/dbg.log 2>&1
grep -v "^  " gpurun_out/s3/dbg.log | tail
timeout 600 python tools_dev/ab_bench.py waldo_amd/lib/abl/r01.so waldo_amd/lib/abl/cur.so 2>&1 | tee gpurun_out/s3/ab.log | grep round
