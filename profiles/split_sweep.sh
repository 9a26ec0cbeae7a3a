#!/bin/bash
# schedule policies of k_trace: octal digits = log2(parts) per class, most significant = class 7 (>= 1024 steps) ... class 0
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for pol in 0 6000000 4000000 3000000 2000000 6100000 6200000 4100000; do
  YCGE_SPLIT=$pol python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('policy $pol', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline']['mean_launch_ms'])"
done
