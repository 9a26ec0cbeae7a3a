#!/bin/bash
# per-wavefront profile (slot time, span, tail) of k_trace alone (YCGE_FAN=0) for library variants, each with and without the cooperative walk
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  for nc in 0 1; do
    if [ $nc = 1 ]; then export YCGE_NO_COOP=1; else unset YCGE_NO_COOP; fi
    echo "== variant '${v:-default}' no_coop=$nc"
    YCGE_FAN=0 timeout 180 python profiles/mega_prof.py ${CFG:-4} 2>&1 | grep -E "trace_ms|slot time|kernel span|in flight at 0.(5|7|9)|>= 256|wave duration"
  done
done
