#!/bin/bash
# the tiled (multi-GPU) frame on ONE rank under torchrun: sequential vs pipelined loop of bench.py, plus the tile-split parity tests
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 600 python -m pytest tests -m gpu -x -q -k "tile_split" 2>&1 | tail -3
for pl in 0 1; do
  for c in 4 5; do
  YCGE_BENCH_PIPELINE=$pl YCGE_BENCH_FORCE_TILED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline=$pl config $c', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame', d['roofline']['mean_launch_ms'])"
  done
done
