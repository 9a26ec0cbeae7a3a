#!/bin/bash
# round 5, late: walk_phase's second form (YCGE_WALK_PHASE 2: useless stack entries discarded in a loop of their own, packed slab products, no
# result code in the loop) against round 4's (lib/var_walk1.so = -DYCGE_WALK_PHASE=1), same call
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
echo "== parity of the voxel kernels (new form = the product build)"
timeout 1500 python -m pytest tests/test_gpu_timed_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "voxel or volume or world or grid or chunk" 2>&1 | tail -3
echo "== config 5 at full size: lit (t01 0.5), dark (0.25), night (0.8); A = product, B = var_walk1"
for rep in 1 2; do for t in 0.5 0.25 0.8; do
  echo -n "t01 $t A: "; timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1
  echo -n "t01 $t B: "; YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_walk1.so timeout 300 python profiles/small_frames.py 5 - 40 $t 2>&1 | tail -1
done; done
echo "== 960x270"
for t in 0.5 0.25; do
  echo -n "t01 $t A: "; timeout 300 python profiles/small_frames.py 5 960x270 60 $t 2>&1 | tail -1
  echo -n "t01 $t B: "; YCGE_LIB=$REPO/yetanotherconsolegameengine_amd/lib/var_walk1.so timeout 300 python profiles/small_frames.py 5 960x270 60 $t 2>&1 | tail -1
done
