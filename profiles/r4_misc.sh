#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
echo "== resident loop from C: exchange/resolve stream at high priority, then plain"
for p in 1 0; do for k in 2 4 6; do YCGE_RES_LOOP_PRIO=$p timeout 300 python profiles/rank_flight.py 4 8 residentc $k 0,3,5 2>&1 | tail -1; done; done
echo "== per-rank trace alone (rank_times)"; timeout 300 python profiles/rank_times.py 4 8 2>&1 | tail -1
