#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/s5/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/s5/pytest.log
