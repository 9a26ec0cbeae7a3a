#!/bin/bash
# per-rank trace time of the tiled frame (config 4) under the fan-out / split knobs: slots are plentiful per rank
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in "$@"; do
  echo "== $v"
  ( if [ "$v" != "-" ]; then for kv in ${v//,/ }; do export "$kv"; done; fi; python profiles/rank_times.py 4 2>&1 | grep -E "world (4|8)" )
done
