#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for r in "6,16,16,1" "6,20,16,1" "6,24,16,1" "6,32,16,1" "5,16,16,1" "7,16,16,1" "6,16,12,1" "6,16,16,1"; do for t in 0.5 0.25; do
  echo -n "YCGE_ROUND=$r t01 $t: "; YCGE_ROUND=$r timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1 | cut -c30-140; done; done
echo "moving camera (bench leg) with 6,10 and 6,16"; for r in "6,10,16,1" "6,16,16,1" "6,24,16,1"; do echo -n "$r: "; YCGE_ROUND=$r timeout 300 python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'moving', d['moving_camera']['frame_ms'])"; done
