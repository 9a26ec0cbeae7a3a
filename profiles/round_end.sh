#!/bin/bash
# round-end measurement on the GPU box: full GPU test suite, default bench (with CPU baseline), secondary configs,
# rocprofv3 kernel trace + PMC passes of the same bench command.  Everything lands in gpurun_out/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r01c}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu_$TAG.log
python bench.py --post > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json
for c in 2 3 5; do python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_${TAG}_cfg$c.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame', d['roofline'])"; done
bash profiles/run_profiles.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; echo "profiles rc=$?"; head -12 gpurun_out/prof_$TAG/summary.txt
