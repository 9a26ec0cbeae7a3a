#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_scene_bvh_device_build.py -m gpu -x -q -s ${F2_K:+-k "$F2_K"} > gpurun_out/pytest_f2.log 2>&1; echo "pytest(f2) rc=$?"; grep "ycge\]" gpurun_out/pytest_f2.log | sort | uniq -c | head; tail -${F2_TAIL:-30} gpurun_out/pytest_f2.log
