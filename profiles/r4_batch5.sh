#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_variants.py -m gpu -q -x -k "resident or flight or pipelined or tiled or two_rank" 2>&1 | grep -v amdgpu.ids | tail -3
try() { local bad=0; for i in $(seq 1 $1); do out=$(timeout 120 python profiles/rank_flight.py 4 8 residentc $3 2>&1 | tail -1 | cut -c1-230); case "$out" in *period*) last="$out";; *) bad=$((bad+1)); echo "   $out";; esac; done; echo "$2: $bad bad of $1; $last"; }
YCGE_RES_LOOP_BATCH=3 try 20 "batch 3 ring 9" 9
YCGE_RES_LOOP_BATCH=2 try 6 "batch 2 ring 6" 6
YCGE_RES_LOOP_BATCH=4 try 4 "batch 4 ring 12" 12
YCGE_RES_LOOP_BATCH=5 try 3 "batch 5 ring 15" 15
