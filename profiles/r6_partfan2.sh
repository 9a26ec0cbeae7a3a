#!/bin/bash
# round 6: (1) the fault of the first partfan run, alone, under each library; (2) the exception barrier under allocation failure; (3) MODE 3 for parts only
# (var_partfan2.so: whole blocks through round 5's loop) against MODE 3 for everything (product) against round 5 (var_nopartfan.so); rank emulation of each.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
L=$REPO/yetanotherconsolegameengine_amd/lib
for lib in var_nopartfan.so libycge_hip.so var_partfan2.so; do
  echo "== $lib: test_one_call_drives_several_devices"
  YCGE_LIB=$L/$lib timeout 300 python -m pytest tests/test_gpu_timed_variants.py -m gpu -x -q -k "one_call_drives_several_devices" 2>&1 | grep -E "passed|failed|fault|Error" | head -5
done
echo "== abi barrier"; timeout 900 python -m pytest tests/test_gpu_abi_barrier.py -m gpu -x -q 2>&1 | tail -8
for cfg in 4 3; do
  echo "== config $cfg"
  for spec in "var_nopartfan.so 32" "libycge_hip.so 32" "var_partfan2.so 32" "var_partfan2.so 64" "var_partfan2.so 128" "var_partfan2.so 256" "var_nopartfan.so 32"; do set -- $spec
    YCGE_LIB=$L/$1 YCGE_SPLIT_TOP=$2 timeout 200 python profiles/sync_ms.py $cfg 200 "$1 split_top=$2" 2>&1 | tail -1; done
done
echo "== a rank of 8 / 4, tile-resident ring of 4 (frame by frame) and batches of 4"
for lib in var_nopartfan.so libycge_hip.so var_partfan2.so; do
  for w in 8 4; do echo -n "$lib world $w K=4: "; YCGE_LIB=$L/$lib timeout 300 python profiles/rank_flight.py 4 $w residentc 4 2>&1 | tail -1 | cut -c1-200; done
  echo -n "$lib world 8 slab two: "; YCGE_LIB=$L/$lib timeout 300 python profiles/rank_flight.py 4 8 two 2>&1 | tail -1 | cut -c1-200
done
