#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
YCGE_PATH=megakernel timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_mega.log 2>&1; echo "pytest(megakernel) rc=$?"; tail -2 gpurun_out/pytest_gpu_mega.log
for i in 1 2 3; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MEGA', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline']['mean_launch_ms'])"; done
