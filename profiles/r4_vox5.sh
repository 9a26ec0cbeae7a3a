#!/bin/bash
# the light loop beside the next round's trace (stage pipeline): voxel parity tests, A/B against YCGE_NO_LIGHTS_BESIDE
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "voxel or volume or grid or lit or cull or config5 or world or graze or walk_tree or analytic or several_devices or update_lights" 2>&1 | tail -4
for t in 0.5 0.25; do for v in "-" "YCGE_NO_LIGHTS_BESIDE=1" "-" "YCGE_NO_LIGHTS_BESIDE=1"; do
  echo "-- t01 $t $v"
  ( if [ "$v" != "-" ]; then export "$v"; fi
    python bench.py --config 5 --t01 $t --steps 20 --warmup 3 --no-cpu-baseline --no-post 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'moving', d['moving_camera']['trace_ms']['median'], 'in flight', d['frames_in_flight']['ms_per_step'])" )
done; done
for c in 1 2; do for v in "-" "YCGE_NO_LIGHTS_BESIDE=1"; do echo "-- config $c $v"; ( if [ "$v" != "-" ]; then export "$v"; fi
    python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-post --no-moving 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'in flight', d['frames_in_flight']['ms_per_step'])" ); done; done
