#!/bin/bash
# A/B of library variants on config 4 inside one call: bench x2 + per-wavefront profile (slot time), default first
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; fi
  echo "== variant '${v:-default}'"
  for i in 1 2; do
    timeout 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-post 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg4', d['value'],'Mrays/s', d['ms_per_step'],'ms trace', d['roofline']['mean_launch_ms'])"
  done
  timeout 180 python bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --no-post 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg3', d['value'],'Mrays/s', d['ms_per_step'],'ms trace', d['roofline']['mean_launch_ms'])"
  YCGE_FAN=0 timeout 180 python profiles/mega_prof.py 4 2>&1 | grep -E "trace_ms|slot time|kernel span|in flight at 0.(5|7|9)|>= 256"
done
