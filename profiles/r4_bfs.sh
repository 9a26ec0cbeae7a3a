#!/bin/bash
# round 4: order-free occlusion queries (mesh_anyhit_bfs) - parity first, then A/B against the ordered walk (YCGE_NO_BFS=1) inside one call
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "order_free or timed_mesh_kernels or zero_intensity or primitive or glass or obj_text or flight_are or texture" --durations=5 > gpurun_out/r4_bfs_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r4_bfs_pytest.log
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'],'Mrays/s', d['ms_per_step'],'ms/frame; trace', d.get('trace_ms'), 'lane steps', (d['roofline'].get('timed_work') or {}).get('lane_steps_per_launch'), 'flight', (d.get('frames_in_flight') or {}).get('ms_per_step'))"; }
for v in bfs nobfs bfs nobfs; do
  if [ $v = nobfs ]; then export YCGE_NO_BFS=1; else unset YCGE_NO_BFS; fi
  echo "== $v"
  timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg4
  timeout 200 python bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline --no-post --no-moving 2>> gpurun_out/bench.err | one cfg3
done
unset YCGE_NO_BFS
timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_mega_prof_bfs.txt; grep -E "trace_ms|span|slot time|>= 256" gpurun_out/r4_mega_prof_bfs.txt; grep -A8 "longest waves" gpurun_out/r4_mega_prof_bfs.txt | head -10
YCGE_NO_BFS=1 timeout 300 python profiles/mega_prof.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_mega_prof_nobfs.txt; grep -E "trace_ms|span|slot time|>= 256" gpurun_out/r4_mega_prof_nobfs.txt
