#!/bin/bash
# compare environment knobs on config 4: each argument is a comma-separated list of NAME=value settings ("-" = none)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  ( if [ "$v" != "-" ]; then for kv in ${v//,/ }; do export "$kv"; done; fi
    for i in 1 2; do
    YCGE_PATH=megakernel python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MEGA', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline']['mean_launch_ms'])"
    done )
done
