#!/bin/bash
# round 5, late: the round sizes of k_wf_trace_p again, with walk_phase's second form (YCGE_ROUND=tree,cell,refill,mode; YCGE_PW_PER_CU)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for t in 0.5 0.25; do for v in "-" "YCGE_ROUND=4,20,16,1" "YCGE_ROUND=8,20,16,1" "YCGE_ROUND=12,20,16,1" "YCGE_ROUND=6,16,16,1" "YCGE_ROUND=6,32,16,1" "YCGE_ROUND=8,32,16,1" "YCGE_ROUND=12,32,16,1" "YCGE_ROUND=6,20,8,1" "YCGE_ROUND=6,20,24,1" "YCGE_ROUND=8,24,16,0" "YCGE_PW_PER_CU=24" "YCGE_PW_PER_CU=40" "-"; do
  ( if [ "$v" != "-" ]; then export $v; fi; echo -n "t01 $t $v: "; timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1 )
done; done
