#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
for v in g8 g8r6; do echo "== parity $v"; YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so timeout 200 python profiles/r3_bisect.py 3 3 2>&1 | grep -v amdgpu.ids | tail -2; YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so timeout 300 python profiles/r3_bisect.py 4 3 2>&1 | grep -v amdgpu.ids | tail -2; done
bash profiles/r3_ab.sh g8 g8r6
