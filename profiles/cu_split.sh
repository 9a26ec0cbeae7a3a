#!/bin/bash
# (the knob this drove - YCGE_CU_SPLIT, CU-masked streams in ycge_host.cpp - was measured and taken out again: DESIGN section 5, experiment table)
# k_trace_fan on n CUs of its own, k_trace on the rest (YCGE_CU_SPLIT, CU-masked streams): config 4 trace
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for cfgs in "0 200" "32 200" "48 200" "64 200" "96 200" "64 400" "96 400" "96 800" "128 800" "0 200"; do
  set -- $cfgs
  echo -n "YCGE_CU_SPLIT=$1 YCGE_FAN_CAP=$2: "
  YCGE_CU_SPLIT=$1 YCGE_FAN=5 YCGE_FAN_CAP=$2 timeout 180 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-post 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms, trace', d['roofline']['mean_launch_ms'])"
done
