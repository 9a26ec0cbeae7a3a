#!/bin/bash
# round 6: the tile-resident resolve as ONE launch (k_resolve_tiles) against round 5's two; hardware queues per stream; a rank's period frame by frame
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_resident" 2>&1 | tail -3
run() { echo -n "$1 | world $2 K=$3: "; ( for kv in $1; do export "$kv"; done; timeout 300 python profiles/rank_flight.py 4 $2 residentc $3 2>&1 | tail -1 | cut -c75-210 ); }
for w in 8 4; do
  run "A=one_launch" $w 4
  run "YCGE_RES_SPLIT_RESOLVE=1" $w 4
  run "GPU_MAX_HW_QUEUES=8" $w 4
  run "GPU_MAX_HW_QUEUES=8 YCGE_RES_SPLIT_RESOLVE=1" $w 4
  run "GPU_MAX_HW_QUEUES=8" $w 6
  run "GPU_MAX_HW_QUEUES=8" $w 8
  run "GPU_MAX_HW_QUEUES=16" $w 8
  run "A=one_launch" $w 6
  run "YCGE_RES_LOOP_COMM=slot" $w 4
  run "GPU_MAX_HW_QUEUES=8 YCGE_RES_LOOP_COMM=slot" $w 4
  run "GPU_MAX_HW_QUEUES=8 YCGE_RES_LOOP_COMM=slot" $w 8
done
echo "== ring 1 (latency 1: the synchronous form of a rank)"; run "A=one_launch" 8 2
