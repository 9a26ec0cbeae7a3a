#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for k in 2 4; do for pol in 33220000 0; do echo "== timeline K=$k YCGE_SPLIT=$pol (rank 5 of 8, config 4)"; YCGE_RES_LOOP_TIMELINE=1 YCGE_SPLIT=$pol timeout 300 python profiles/rank_flight.py 4 8 residentc $k 3,5 2>&1 | grep -E "trace|period" | tail -17 | cut -c1-200; done; done
