#!/bin/bash
# round 3, first GPU call: full GPU suite, bench of every GPU config WITH the CPU baseline beside it, kernel trace + PMC passes of configs 4 and 5.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r03a}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
for c in 2 3 5; do timeout 400 python bench.py --config $c --cpu-seconds 10 2>> gpurun_out/bench_$TAG.err > gpurun_out/bench_${TAG}_cfg$c.json; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_cfg$c.json"))
    print("config $c", d["value"], "Mrays/s", d["ms_per_step"], "ms/frame; trace", d.get("trace_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "moving", (d.get("moving_camera") or {}).get("trace_ms"), "post", d.get("post_stage"))
except Exception as e:
    print("config $c failed", e)
PY
done
timeout 300 python bench.py --camera orbit --no-cpu-baseline --no-post > gpurun_out/bench_${TAG}_orbit.json 2>> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_${TAG}_orbit.json | head -c 1500; echo
bash profiles/run_profiles.sh ${TAG}_c4 > gpurun_out/prof_${TAG}_c4.log 2>&1; echo "profiles c4 rc=$?"; head -14 gpurun_out/prof_${TAG}_c4/summary.txt
bash profiles/run_profiles.sh ${TAG}_c5 --config 5 > gpurun_out/prof_${TAG}_c5.log 2>&1; echo "profiles c5 rc=$?"; head -14 gpurun_out/prof_${TAG}_c5/summary.txt
