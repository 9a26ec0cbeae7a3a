#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "-" "YCGE_ROUND=6,10,4" "YCGE_ROUND=6,10,8" "YCGE_ROUND=6,10,16" "YCGE_ROUND=6,10,24" "YCGE_ROUND=6,10,32" "YCGE_ROUND=8,12,16" "YCGE_ROUND=4,8,16" "-"; do
  echo "-- $v"
  ( if [ "$v" != "-" ]; then export "$v"; fi
    python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'moving', d['moving_camera']['trace_ms']['median'])" )
done
