#!/bin/bash
# round 6: the new split default (two parts, count by block count) against round 5's (32 blocks in four), the spread of frame times with the SAME frame
# number every frame (same rays) against consecutive frames, the GPU suite on this build
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6_split2_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6_split2_pytest.log
for cfg in 4 3; do
  echo "== config $cfg"
  YCGE_SPLIT_TOP=32 YCGE_SPLIT_TOP_LG=2 timeout 200 python profiles/sync_ms.py $cfg 300 "round 5: 32 x 4 parts" 2>&1 | tail -1
  timeout 200 python profiles/sync_ms.py $cfg 300 "default" 2>&1 | tail -1
  YCGE_SPLIT_TOP=32 YCGE_SPLIT_TOP_LG=2 timeout 200 python profiles/sync_ms.py $cfg 300 "round 5: 32 x 4 parts" 2>&1 | tail -1
  timeout 200 python profiles/sync_ms.py $cfg 300 "default" 2>&1 | tail -1
  SAME_FRAME=40 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 40 every time" 2>&1 | tail -1
  SAME_FRAME=41 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 41 every time" 2>&1 | tail -1
  SAME_FRAME=42 timeout 200 python profiles/sync_ms.py $cfg 300 "default, frame 42 every time" 2>&1 | tail -1
done
