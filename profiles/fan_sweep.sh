#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
bash profiles/env_sweep.sh YCGE_FAN=0 YCGE_FAN=6 YCGE_FAN=5 YCGE_FAN=4 YCGE_FAN=3
export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_w4.so
echo "#### 4 wavefronts per SIMD"
bash profiles/env_sweep.sh YCGE_FAN=0 YCGE_FAN=6 YCGE_FAN=5 YCGE_FAN=4 YCGE_FAN=3
