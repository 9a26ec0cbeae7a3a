#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q -k "zero_intensity or one_call or pipelined or tiled" 2>&1 | tail -4
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'post', d.get('post_stage'), 'n_gpus', d['n_gpus'], d['config'].get('device_tiles'))"; }
for v in cr2 cr3; do
  export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so
  echo "== bench variant '$v'"
  timeout 200 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4
  timeout 200 python bench.py --config 3 --steps 200 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
done
unset YCGE_LIB
echo "== SDR frame with the pinned, reused buffer"
timeout 200 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-moving 2>> gpurun_out/bench.err | one cfg4
echo "== tiled path under torchrun with one rank"
YCGE_BENCH_FORCE_TILED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --form rccl --steps 100 --warmup 8 2>> gpurun_out/bench.err | one tiled
echo "== plain --gpus 2 on a one-GPU box must fail loudly"
timeout 100 python bench.py --gpus 2 --steps 5 2>&1 | tail -1
tail -3 gpurun_out/bench.err
