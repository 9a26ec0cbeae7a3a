#!/bin/bash
# round sizes of k_wf_trace_p with the phase gating on (mode 1), config 5 at full size, lit and dark; then bench.py's launcher forms (eager resolves)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for r in "6,10,16,1" "8,12,16,1" "8,16,16,1" "4,8,16,1" "10,20,16,1" "6,16,16,1" "6,10,8,1" "6,10,24,1" "12,10,16,1" "6,10,16,1"; do for t in 0.5 0.25; do
  echo -n "YCGE_ROUND=$r t01 $t: "; YCGE_ROUND=$r timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1 | cut -c30-140; done; done
bash profiles/forms.sh 2>&1 | grep -v "^\[" | cut -c1-700
