#!/bin/bash
# dev helper: the LVD-recipe step and Warper.forward with each variant library under tools_dev/_variants/
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for so in tools_dev/_variants/*.so; do
  echo "$(basename $so): $(timeout 300 python tools_dev/bench_lvd_step.py --lib $PWD/$so 2>/dev/null | tail -1 | cut -c1-160)"
  echo "$(basename $so): $(timeout 300 python tools_dev/bench_warper_fwd.py --lib $PWD/$so 2>/dev/null | tail -1 | cut -c1-200)"
done; done
