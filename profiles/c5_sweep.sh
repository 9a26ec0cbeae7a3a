#!/bin/bash
# config 5 (voxel world, wavefront path): persistent wavefronts per CU of k_wf_trace_p and its round sizes
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in "$@"; do
  echo "== $v"
  ( if [ "$v" != "-" ]; then for kv in ${v//;/ }; do export "$kv"; done; fi
    python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame', d['roofline']['mean_launch_ms'])" )
done
