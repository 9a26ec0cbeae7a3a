#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for e in "-" "YCGE_SPLIT=66666666" "YCGE_SPLIT=44444444" "YCGE_SPLIT=0" "YCGE_SPLIT=0,YCGE_NO_COOP=1"; do
  echo "== $e"
  ( if [ "$e" != "-" ]; then for kv in ${e//,/ }; do export "$kv"; done; fi; timeout 300 python profiles/rank_times.py 4 256 16 2>&1 | grep -E "world" )
done
