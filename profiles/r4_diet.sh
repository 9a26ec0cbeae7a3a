#!/bin/bash
# round 4: instruction diet of the cooperative walk - parity, bench (two runs), clocks per cooperative iteration (-DYCGE_DBG_COOPSTAT build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "timed_mesh_kernels or zero_intensity or primitive or glass or obj_text or fan_out" > gpurun_out/r4_diet_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_diet_pytest.log
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'lane steps', (d['roofline'].get('timed_work') or {}).get('lane_steps_per_launch'), 'flight', (d.get('frames_in_flight') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4
  timeout 200 python bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
done
YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_coopstat.so timeout 200 python profiles/coop_stats.py 4 2>&1 | grep -v amdgpu.ids
timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_mega_prof_diet.txt; grep -E "trace_ms|span|slot time|>= 256" gpurun_out/r4_mega_prof_diet.txt
