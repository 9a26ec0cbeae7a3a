#!/bin/bash
# builds of the library with other occupancy targets (var_<name>.so) under the fan-out knobs
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in w34 w44; do
export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so
echo "#### $v (wavefronts per SIMD: k_trace, k_trace_fan)"
bash profiles/env_sweep.sh "$@"
done
