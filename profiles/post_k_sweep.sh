#!/bin/bash
# banded in-place A-trous iteration (config 4, 1080p): arguments are "rows,K" pairs
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rk in "$@"; do
  r=${rk%,*}; k=${rk#*,}
  echo "== band rows $r, levels per launch $k"
  YCGE_POST_BAND_ROWS=$r YCGE_POST_K=$k python profiles/post_prof.py 4 2>&1 | grep "^frame" | tail -1
done
