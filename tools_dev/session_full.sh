#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/full/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/full/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
