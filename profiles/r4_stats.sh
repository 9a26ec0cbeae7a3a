#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for c in 4 3; do YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_batchstat.so timeout 300 python profiles/batch_stats.py $c 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4_batch_stats.txt
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'lane steps', (d['roofline'].get('timed_work') or {}).get('lane_steps_per_launch'))"; }
for v in 0 4 8 16; do echo "== YCGE_BFS=$v"; YCGE_BFS=$v timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving --no-flight 2>> gpurun_out/bench.err | one cfg4; done
