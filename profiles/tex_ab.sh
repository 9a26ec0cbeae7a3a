#!/bin/bash
# cost of carrying the texture branch in untextured scenes: default build against lib/var_notex.so (-DYCGE_TEXTURES=0), alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for round in 1 2 3; do
  for v in default notex; do
    if [ $v = notex ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_notex.so; else unset YCGE_LIB; fi
    echo -n "$v: "
    for c in 4 3 2; do timeout 200 python bench.py --config $c --steps 40 --warmup 6 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c', d['value'], d['roofline']['mean_launch_ms'], end='  ')"; done; echo
  done
done
