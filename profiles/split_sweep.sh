#!/bin/bash
# schedule policies of k_trace: octal digits, most significant = class 7 (>= 1024 steps) ... class 0
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for pol in 0 20000000 22000000 22200000 42000000 62000000 64200000 11000000 11100000; do
  YCGE_SPLIT=$pol YCGE_PATH=megakernel python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('policy $pol', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline']['mean_launch_ms'])"
done
