#!/bin/bash
# config 5 (voxel world): parity of the timed stage kernels at full size, then bench A/B of variants
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q -k "voxel or mesh_kernels_full" 2>&1 | tail -4
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'post', (d.get('post_stage') or {}).get('post_ms'))"; }
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  echo "== bench variant '${v:-default}'"
  timeout 300 python bench.py --config 5 --steps 40 --warmup 5 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg5
done
