#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for r in 6,8 3,8 12,8 6,4 6,16 12,16 24,32 3,4 2,2; do
  YCGE_ROUND=$r python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame')"
done
