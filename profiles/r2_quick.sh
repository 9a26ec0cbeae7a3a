#!/bin/bash
# quick iteration on the GPU box: parity suite, benches of configs 4 / 3 / 5, optional micro benchmarks
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-q}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
for c in 4 3 5 2; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-post 2>> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_${TAG}_cfg$c.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'])"; done
if [ -n "$2" ]; then bash -c "$2"; fi
