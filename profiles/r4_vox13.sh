#!/bin/bash
# the empty-brick run of the cell walk: voxel parity tests, A/B against lib/var_norun.so (-DYCGE_EMPTY_BRICK_RUN=0)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "voxel or volume or grid or lit or cull or config5 or world or graze or walk_tree" 2>&1 | tail -4
for t in 0.5 0.25; do for v in "-" "YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_norun.so" "YCGE_ROUND=6,4,16" "YCGE_ROUND=6,2,16" "YCGE_ROUND=8,3,16" "-" "YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_norun.so"; do
  echo "-- t01 $t $v"
  ( if [ "$v" != "-" ]; then export "$v"; fi
    python bench.py --config 5 --t01 $t --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'moving', d['moving_camera']['trace_ms']['median'])" )
done; done
YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_voxstat.so timeout 300 python profiles/vox_stats.py 0.5 2>&1 | grep -v amdgpu.ids
