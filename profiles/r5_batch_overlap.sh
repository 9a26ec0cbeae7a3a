#!/bin/bash
# do consecutive k_trace_batch launches (two streams) overlap?  Event marks around the last batches of the loop, unprofiled
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for spec in "12 4" "15 3" "8 2"; do set -- $spec; echo "== ring $1 batch $2"; YCGE_RES_LOOP_TIMELINE=1 YCGE_RES_LOOP_BATCH=$2 timeout 300 python profiles/rank_flight.py 4 8 residentc $1 1 2>&1 | grep -E "batch|period" | cut -c1-160; done
echo "== frame by frame K=4 and K=8"; for k in 4 8; do YCGE_RES_LOOP_TIMELINE=1 timeout 300 python profiles/rank_flight.py 4 8 residentc $k 1 2>&1 | grep -E "trace|period" | cut -c1-160; done
