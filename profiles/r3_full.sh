#!/bin/bash
# full GPU suite + default bench + rank emulation
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r03b}
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print(d['value'], d['ms_per_step'], d['trace_ms'], d['roofline']['frac'], d['roofline'].get('timed_work'), d['cpu_baseline']['value'], d.get('moving_camera',{}).get('trace_ms'), d.get('post_stage'))"
for fan in "-" "YCGE_FAN=0" "YCGE_FAN=5,YCGE_FAN_CAP=512"; do echo "== rank emulation $fan"; ( if [ "$fan" != "-" ]; then for kv in ${fan//,/ }; do export "$kv"; done; fi; timeout 600 python profiles/rank_times.py 4 2>&1 | grep -E "world" ); done
