#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
timeout 600 python tools_dev/ab_bench.py waldo_amd/lib/abl/r01.so waldo_amd/lib/abl/cur.so 2>&1 | tee gpurun_out/s4/ab.log | grep round
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warp_composite or baseline or two_kernel or staged or non_finite or fixed_point or c3 or backward_repro or long_batches" > gpurun_out/s4/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s4/pytest.log
