#!/bin/bash
# quick loop: parity of the timed mesh kernels (config 3 full size, 4 frames) + coop statistics + short bench A/B, for the variants named
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
echo "== parity (default lib)"; timeout 200 python profiles/r3_bisect.py 3 4 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python profiles/r3_bisect.py 4 3 2>&1 | grep -v amdgpu.ids | tail -4
for v in ${STAT:-w3stat}; do echo "== stats $v"; for c in 4 3; do YCGE_FAN=0 YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so timeout 200 python profiles/coop_stats.py $c 2>&1 | grep -v amdgpu.ids; done; done
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'))"; }
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  for fan in 5 0; do
    echo "== bench variant '${v:-default}' YCGE_FAN=$fan"
    YCGE_FAN=$fan timeout 200 python bench.py --steps 150 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4
    YCGE_FAN=$fan timeout 200 python bench.py --config 3 --steps 150 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
  done
done
