#!/bin/bash
# persistent forms of the in-place A-trous iteration (YCGE_POST_MODE=0 level hand-over / 3 same, block order / 4 group hand-over) against the launch form (=2)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "denoise or post_stage or sdr" > gpurun_out/pytest_post.log 2>&1; echo "pytest(post, default mode) rc=$?"; tail -3 gpurun_out/pytest_post.log
for mode in 2 0 3 4; do echo "== hash of frames, YCGE_POST_MODE=$mode"; YCGE_POST_MODE=$mode timeout 300 python profiles/post_ab.py 4 ${FRAMES:-8} 2>&1 | tail -${TAILN:-4}; done
