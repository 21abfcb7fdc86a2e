#!/bin/bash
# dev helper: the LVD-recipe step with each variant library under waldo_amd/lib/abl/
cd $GRAFT_REPO_ROOT
for so in "$@"; do
  echo "== $so"
  WALDO_HIP_LIB=$PWD/$so timeout 300 python tools_dev/bench_lvd_step.py 2 20 2>&1 | tail -1 | cut -c1-140
done
