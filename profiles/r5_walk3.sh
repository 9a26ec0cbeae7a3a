#!/bin/bash
# round 5, late: a phase of k_wf_trace_p's round ends for every lane once fewer than E eighths of the lanes that entered it are left (YCGE_ROUND 5th field)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2; do for t in 0.5 0.25; do for E in 0 2 3 4 5 6; do
  echo -n "t01 $t exit_eighths $E: "; YCGE_ROUND=6,20,16,1,$E timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1
done; done; done
for t in 0.5 0.25; do for v in "8,20,16,1,3" "8,32,16,1,3" "10,32,16,1,4" "6,32,16,1,4" "8,32,16,1,2"; do
  echo -n "t01 $t YCGE_ROUND=$v: "; YCGE_ROUND=$v timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1
done; done
