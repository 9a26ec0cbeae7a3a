#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
