#!/bin/bash
# levels per launch of the banded in-place A-trous iteration (config 4, 1080p)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for k in "$@"; do
  echo "== YCGE_POST_K=$k"
  YCGE_POST_K=$k python profiles/post_prof.py 4 2>&1 | grep "^frame" | tail -2
done
