#!/bin/bash
# window form of k_atrous_band against the hash form: parity tests on both, then timing + frame hashes
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "denoise or post_stage or sdr" > gpurun_out/pytest_post.log 2>&1; echo "pytest(post, window) rc=$?"; tail -3 gpurun_out/pytest_post.log
YCGE_POST_HASH_FORM=1 timeout 900 python -m pytest tests -m gpu -x -q -k "denoise or post_stage or sdr" > gpurun_out/pytest_post_hash.log 2>&1; echo "pytest(post, hash) rc=$?"; tail -3 gpurun_out/pytest_post_hash.log
for form in 0 1; do echo "== YCGE_POST_HASH_FORM=$form"; YCGE_POST_HASH_FORM=$form timeout 300 python profiles/post_prof.py 4; done
for form in 0 1; do echo "== hash of frames, YCGE_POST_HASH_FORM=$form"; YCGE_POST_HASH_FORM=$form timeout 300 python profiles/post_ab.py 2>&1 | tail -4; done
