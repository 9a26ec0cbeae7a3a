#!/bin/bash
# (-DYCGE_PRIO_BLOCKS - s_setprio for the first N schedule entries in k_trace - was measured and taken out again: DESIGN section 5, experiment table)
# wavefront priority for the head of the longest-first schedule (lib/var_prio<N>.so, -DYCGE_PRIO_BLOCKS=N): config 4 / 3 trace, fan on and off
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for round in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset YCGE_LIB; else export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; fi
    echo -n "$v: "
    for fan in "" 0; do
      for c in 4 3; do YCGE_FAN=$fan timeout 200 python bench.py --config $c --steps 40 --warmup 6 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c fan[$fan]', d['value'], d['roofline']['mean_launch_ms'], end='  ')"; done
    done; echo
  done
done
