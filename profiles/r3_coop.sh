#!/bin/bash
# round 3: parity of the cooperative walk (timed-variant + parity suites), then A/B inside one call: variants x {coop, YCGE_NO_COOP=1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
TESTS=${TESTS:-"tests/test_gpu_timed_variants.py tests/test_gpu_parity.py"}
if [ "$TESTS" != "none" ]; then
  timeout 1500 python -m pytest $TESTS -m gpu -x -q > gpurun_out/pytest_coop.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_coop.log
fi
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'timed steps', d['roofline']['timed_work']['lane_steps_per_launch'])"; }
for v in "" "$@"; do
  if [ -n "$v" ]; then export YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_$v.so; else unset YCGE_LIB; fi
  for nc in 0 1; do
    if [ $nc = 1 ]; then export YCGE_NO_COOP=1; else unset YCGE_NO_COOP; fi
    echo "== variant '${v:-default}' no_coop=$nc"
    for i in 1 2; do timeout 200 python bench.py --steps 150 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4; done
    timeout 200 python bench.py --config 3 --steps 150 --warmup 8 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
  done
done
unset YCGE_LIB YCGE_NO_COOP
tail -5 gpurun_out/bench.err
