#!/bin/bash
# dev helper: fused-path parity tests, then a short bench and its kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "warp_composite or baseline or two_kernel or staged or non_finite or fixed_point or c3 or backward_repro or long_batches" > gpurun_out/s1/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/s1/pytest.log
tail -15 gpurun_out/s1/pytest.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/s1/bench.log 2>&1
tail -2 gpurun_out/s1/bench.log | cut -c1-1500
timeout 600 bash tools_dev/prof.sh 2>&1 | tail -12
