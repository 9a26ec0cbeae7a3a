#!/bin/bash
# two voxel-world frames in flight at a time (second set of stage queues): in-flight tests, A/B against YCGE_NO_FLIGHT_STAGE_OVERLAP
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_scene_bvh_device_build.py -m gpu -q -x -k "flight" 2>&1 | tail -4
for t in 0.5 0.25; do for v in "-" "YCGE_NO_FLIGHT_STAGE_OVERLAP=1" "-" "-"; do
  echo "-- t01 $t $v"
  ( if [ "$v" != "-" ]; then export "$v"; fi
    python bench.py --config 5 --t01 $t --steps 20 --warmup 3 --no-cpu-baseline --no-moving 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); fl=d['frames_in_flight']; print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame trace', d['roofline']['mean_launch_ms'], 'in flight', fl['ms_per_step'], 'with sdr', fl['with_sdr']['ms_per_step'], 'sync sdr', d['post_stage']['frame_ms_with_sdr_readback'], fl.get('two_trace_streams'))" )
done; done
