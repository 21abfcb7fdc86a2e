#!/bin/bash
for so in "" waldo_amd/lib/abl/*.so; do
  if [ -n "$so" ]; then export WALDO_HIP_LIB=$GRAFT_REPO_ROOT/$so; else unset WALDO_HIP_LIB; fi
  echo "== ${so:-baseline}"
  bash tools_dev/prof.sh 2>&1 | grep -E "waldo::warp" | cut -c1-60,70-120
done
