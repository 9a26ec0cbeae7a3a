#!/bin/bash
# per-rank trace time of the tiled frame (config 4) under the fan-out / split / refill knobs: slots are plentiful per rank
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in "-" YCGE_FAN=5 YCGE_FAN=4 YCGE_FAN=3 YCGE_FAN=2 YCGE_FAN=1 YCGE_REFILL=16; do
  echo "== $v"
  ( if [ "$v" != "-" ]; then export "$v"; fi; python profiles/rank_times.py 4 2>&1 | grep -E "world (1|4|8)" )
done
