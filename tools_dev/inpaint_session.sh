#!/bin/bash
# dev: WIF.inpaint at the recipe size -- the tests, the timing line, the kernel statistics of the timing test
set -e
mkdir -p gpurun_out
python -m pytest tests/test_inpaint.py -x -q -m gpu -s > gpurun_out/inpaint_tests.txt 2>&1 || { tail -40 gpurun_out/inpaint_tests.txt; exit 1; }
tail -5 gpurun_out/inpaint_tests.txt
cp gpurun_out/inpaint_R_timing.json gpurun_out/inpaint_R_timing_first.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_inpaint -- python3 -m pytest $GRAFT_REPO_ROOT/tests/test_inpaint.py -x -q -m gpu -k timing -p no:cacheprovider > $GRAFT_REPO_ROOT/gpurun_out/inpaint_prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_inpaint/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/inpaint_kernel_stats.csv
rm -rf gpurun_out/prof_inpaint
head -40 gpurun_out/inpaint_kernel_stats.csv | cut -c1-160
