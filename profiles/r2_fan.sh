#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for kv in "YCGE_FAN=5 YCGE_FAN_CAP=200" "YCGE_FAN=5 YCGE_FAN_CAP=400" "YCGE_FAN=4 YCGE_FAN_CAP=400" "YCGE_FAN=4 YCGE_FAN_CAP=800" "YCGE_FAN=3 YCGE_FAN_CAP=1200" "YCGE_FAN=0" "YCGE_FAN=6 YCGE_FAN_CAP=100"; do
  for i in 1 2; do
  env $kv timeout 180 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-post 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['value'],'Mrays/s', d['ms_per_step'],'ms trace', d['roofline']['mean_launch_ms'])"
  done
done
