#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
python -m pytest tests/test_gpu_timed_variants.py -x -q -m gpu -k "voxel" 2>&1 | tail -2
for rep in 1 2; do
timeout 300 python profiles/flight_ab.py 5 100 2>&1 | tail -1
echo "== ordinary stores"
YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/libycge_hip_nt0.so timeout 300 python profiles/flight_ab.py 5 100 2>&1 | tail -1
done
