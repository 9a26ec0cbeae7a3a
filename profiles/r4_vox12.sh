#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "-" "YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_tracep6.so" "YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_tracep6.so;YCGE_PW_PER_CU=48" "YCGE_PW_PER_CU=40" "YCGE_PW_PER_CU=20" "-"; do
  echo "-- $v"
  ( if [ "$v" != "-" ]; then for kv in ${v//;/ }; do export "$kv"; done; fi
    python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'moving', d['moving_camera']['trace_ms']['median'])" )
done
