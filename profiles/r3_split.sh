#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
YCGE_SPLIT_TOP=48 YCGE_SPLIT_TOP_LG=4 timeout 300 python profiles/r3_bisect.py 4 4 2>&1 | grep -v amdgpu.ids | tail -2
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'))"; }
for e in "YCGE_SPLIT_TOP=32,YCGE_SPLIT_TOP_LG=2" "YCGE_SPLIT_TOP=32,YCGE_SPLIT_TOP_LG=3" "YCGE_SPLIT_TOP=32,YCGE_SPLIT_TOP_LG=4" "YCGE_SPLIT_TOP=64,YCGE_SPLIT_TOP_LG=4" "YCGE_SPLIT_TOP=128,YCGE_SPLIT_TOP_LG=4" "YCGE_SPLIT_TOP=64,YCGE_SPLIT_TOP_LG=3" "YCGE_SPLIT_TOP=32,YCGE_SPLIT_TOP_LG=5"; do
  echo "== $e"
  ( for kv in ${e//,/ }; do export "$kv"; done
    timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4
    timeout 200 python bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3 )
done
