#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_w3.so
for fan in 0 5; do echo "== w3 coop YCGE_FAN=$fan"; YCGE_FAN=$fan timeout 180 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids; done
echo "== w3 NO_COOP FAN=0"; YCGE_NO_COOP=1 YCGE_FAN=0 timeout 180 python profiles/mega_prof.py 4 2>&1 | grep -E "longest|wave [0-9]+ tile" | head -14
