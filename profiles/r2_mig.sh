#!/bin/bash
# path migration: parity on the timed variants, then library variants and round / shade knobs on configs 4 and 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q 2>&1 | tail -5
bash profiles/r2_var.sh base mig3 mig2
unset YCGE_LIB
for kv in "YCGE_MIG_ROUND=6" "YCGE_MIG_ROUND=24" "YCGE_MIG_SHADE=4" "YCGE_MIG_SHADE=32" "YCGE_MIG_ROUND=24 YCGE_MIG_SHADE=32"; do
  echo "== default lib, $kv"
  env $kv timeout 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-post 2>> gpurun_out/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg4', d['value'],'Mrays/s', d['ms_per_step'],'ms trace', d['roofline']['mean_launch_ms'])"
done
