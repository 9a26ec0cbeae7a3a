#!/bin/bash
# A/B of library variants (default first and last), config 4 and 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'))"; }
for v in "" "$@" ""; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  echo "== variant '${v:-default}'"
  timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving --no-flight 2>> gpurun_out/bench.err | one cfg4
  timeout 200 python bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving --no-flight 2>> gpurun_out/bench.err | one cfg3
done
