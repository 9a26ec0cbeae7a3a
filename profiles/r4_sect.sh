#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_coopstat.so timeout 200 python profiles/coop_stats.py 4 2>&1 | grep -v amdgpu.ids
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'))"; }
for v in "32 2" "64 2" "128 2" "48 3" "16 2"; do set -- $v; echo "== YCGE_SPLIT_TOP=$1 LG=$2"; YCGE_SPLIT_TOP=$1 YCGE_SPLIT_TOP_LG=$2 timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving --no-flight 2>> gpurun_out/bench.err | one cfg4; done
