#!/bin/bash
# round 5: is a rank's share throughput-bound once launches overlap?  Two overlapping batches of 4 shares = a whole frame's blocks take 0.84 ms against the whole
# frame's 0.48: the per-class split of the heavy blocks (policy 033220000: 8 / 4 parts of a block's 64 pixels) costs slot time.  Policies by batch / ring.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for pol in 033220000 0 011110000 022110000 022220000 000110000 002210000; do
  echo -n "YCGE_SPLIT=$pol ring 12 batch 4: "; YCGE_SPLIT=$pol YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 4 8 residentc 12 1,2,5 2>&1 | tail -1 | cut -c75-150
  echo -n "YCGE_SPLIT=$pol K=4: "; YCGE_SPLIT=$pol timeout 300 python profiles/rank_flight.py 4 8 residentc 4 1,2,5 2>&1 | tail -1 | cut -c75-150
  echo -n "YCGE_SPLIT=$pol K=8 queues 8: "; GPU_MAX_HW_QUEUES=8 YCGE_SPLIT=$pol timeout 300 python profiles/rank_flight.py 4 8 residentc 8 1,2,5 2>&1 | tail -1 | cut -c75-150
done
echo "== split_top on an unsplit policy (the whole-frame rule: the N heaviest blocks in 4 parts)"
for st in 8 32; do echo -n "YCGE_SPLIT_TOP=$st (no class split) ring 12 batch 4: "; YCGE_SPLIT_TOP=$st YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 4 8 residentc 12 1,2,5 2>&1 | tail -1 | cut -c75-150; done
