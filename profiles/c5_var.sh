#!/bin/bash
# config 5: builds of the library with other occupancy budgets for the voxel-world stage kernels (lib/var_<name>.so, -DYCGE_LIGHTS_WAVES / -DYCGE_TRACEP_WAVES)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; fi
  echo "== variant '${v:-default}'"
  for i in 1 2; do python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame')"; done
done
