#!/bin/bash
# banded in-place A-trous iteration (config 4, 1080p): arguments are "rows,K,G" triples
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rk in "$@"; do
  IFS=, read r k g <<< "$rk"
  echo "== band rows $r, levels per launch $k, pixels per pass ${g:-16}"
  YCGE_POST_BAND_ROWS=$r YCGE_POST_K=$k YCGE_POST_GROUPS=${g:-16} timeout 120 python profiles/post_prof.py 4 2>&1 | grep "^frame" | tail -1
done
