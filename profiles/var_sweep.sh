#!/bin/bash
# compare builds of the library (yetanotherconsolegameengine_amd/lib/var_*.so, made with -D experiment macros) on config 4
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "" $@; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; fi
  echo "== variant '${v:-default}'"
  for i in 1 2; do
  YCGE_PATH=megakernel python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MEGA', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline']['mean_launch_ms'])"
  done
  python profiles/mega_prof.py 4 2>&1 | grep -E "trace_ms|ns per longest|kernel span"
done
