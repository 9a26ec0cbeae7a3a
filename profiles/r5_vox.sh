#!/bin/bash
# round 5: the voxel stages - the persistent light stage (k_wf_lights_p, YCGE_LIGHTS_P) and the phase gating of a round (YCGE_ROUND=tree,cell,refill,mode)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
echo "== parity of the timed voxel kernels with the new forms on (lit worlds, full size and the cull poses)"
for v in "YCGE_LIGHTS_P=1" "YCGE_ROUND=6,10,16,1" "YCGE_ROUND=6,10,16,2" "YCGE_LIGHTS_P=1 YCGE_ROUND=6,10,16,1"; do
  ( export $v; echo -n "$v: "; timeout 900 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "voxel or volume or world or grid" 2>&1 | tail -1 )
done
echo "== config 5 at full size: lit (t01 0.5), dark (0.25), night (0.8)"
for t in 0.5 0.25 0.8; do for v in "-" "YCGE_LIGHTS_P=1" "YCGE_ROUND=6,10,16,1" "YCGE_ROUND=6,10,16,2" "YCGE_LIGHTS_P=1 YCGE_ROUND=6,10,16,1" "YCGE_LIGHTS_P=1 YCGE_ROUND=6,10,16,2" "YCGE_LIGHTS_P=1 YCGE_ROUND=6,10,32,0" "-"; do
  ( if [ "$v" != "-" ]; then export $v; fi; echo -n "t01 $t $v: "; timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1 ); done; done
echo "== the same at 960x270 (8 100 tiles)"
for v in "-" "YCGE_LIGHTS_P=1" "YCGE_ROUND=6,10,16,1" "YCGE_LIGHTS_P=1 YCGE_ROUND=6,10,16,1"; do ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$v: "; timeout 300 python profiles/small_frames.py 5 960x270 60 0.5 2>&1 | tail -1 ); done
