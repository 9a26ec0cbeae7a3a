#!/bin/bash
# fan-out of the heaviest blocks (k_trace_fan): how many blocks, from which schedule class - config 4 trace
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for fc in "0 0" "5 100" "5 200" "5 400" "5 800" "4 400" "4 800" "4 1600" "3 1600" "6 200"; do
  set -- $fc
  echo -n "YCGE_FAN=$1 YCGE_FAN_CAP=$2: "
  YCGE_FAN=$1 YCGE_FAN_CAP=$2 timeout 180 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms, trace', d['roofline']['mean_launch_ms'])"
done
