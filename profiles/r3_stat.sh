#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in ${VARS:-w3stat}; do for fan in 0; do echo "== $v YCGE_FAN=$fan"; for c in 4 3; do YCGE_FAN=$fan YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so timeout 200 python profiles/coop_stats.py $c 2>&1 | grep -v amdgpu.ids; done; done; done
