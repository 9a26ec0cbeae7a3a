#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-r03c}
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_$TAG.log
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'))"; }
for v in "" cr2 cr3; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  echo "== bench variant '${v:-default}'"
  for i in 1 2; do timeout 200 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4; done
  timeout 200 python bench.py --config 3 --steps 200 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
done
