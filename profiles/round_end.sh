#!/bin/bash
# round-end measurement on the GPU box: full GPU test suite, default bench (with CPU baseline and the SDR frame), secondary configs,
# rocprofv3 kernel trace + PMC passes of configs 4 and 5 (every profiler run under its own timeout).  Everything lands in gpurun_out/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r02}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json
for c in 2 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_${TAG}_cfg$c.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame', d['roofline']['mean_launch_ms'], d.get('post_stage'))"; done
bash profiles/run_profiles.sh ${TAG}_c4 > gpurun_out/prof_${TAG}_c4.log 2>&1; echo "profiles c4 rc=$?"; head -12 gpurun_out/prof_${TAG}_c4/summary.txt
bash profiles/run_profiles.sh ${TAG}_c5 --config 5 > gpurun_out/prof_${TAG}_c5.log 2>&1; echo "profiles c5 rc=$?"
bash profiles/post_quick.sh 2>&1 | head -24
