#!/bin/bash
# round 2, first GPU call: GPU test suite (incl. the timed-variant tests), default bench, PMC passes of configs 4 and 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r02a}
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
grep -E "timed variants|sdr frame" gpurun_out/pytest_gpu_$TAG.log | head -40
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json
bash profiles/run_profiles.sh ${TAG}_c4 > gpurun_out/prof_${TAG}_c4.log 2>&1; echo "profiles c4 rc=$?"; head -14 gpurun_out/prof_${TAG}_c4/summary.txt; cat gpurun_out/prof_${TAG}_c4/pmc_config4.json
bash profiles/run_profiles.sh ${TAG}_c5 --config 5 > gpurun_out/prof_${TAG}_c5.log 2>&1; echo "profiles c5 rc=$?"; cat gpurun_out/prof_${TAG}_c5/pmc_config5.json
