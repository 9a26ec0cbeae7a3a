#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for c in 5 2 4; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame')"; done
bash profiles/c5_quick.sh | grep false
