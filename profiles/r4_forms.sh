#!/bin/bash
# bench.py's one-process-per-GPU forms on ONE rank (torchrun, world 1, YCGE_BENCH_FORCE_TILED): the RCCL code paths of the slab form and of the tile-resident form
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for spec in "rccl" "resident --batch 0" "resident" "resident --batch 3 --steps 50" "auto"; do
  echo "== --form $spec"
  YCGE_BENCH_FORCE_TILED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 52 --warmup 5 --form $spec --no-cpu-baseline --no-post 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame;', d['config']['form'], '|', d['config']['parallelism'][-150:])
except Exception as e:
    print('FAILED', e, l[-600:])"
done
