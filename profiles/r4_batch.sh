#!/bin/bash
# ycge_trace_tiles_resident_batch: the resident parity tests (batched cases included), then a rank's period per frame with n frames a launch
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
if [ "$1" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resident" 2>&1 | grep -v amdgpu.ids | tail -6; fi
for spec in "8 4 0" "8 12 4" "8 15 5" "8 8 4" "8 15 7" "8 9 3" "8 6 2" "4 12 4" "4 6 2" "2 12 4" "2 6 2"; do set -- $spec
  echo -n "world $1 ring $2 batch $3: "; YCGE_RES_LOOP_BATCH=$3 timeout 300 python profiles/rank_flight.py 4 $1 residentc $2 2>&1 | tail -1 | cut -c1-260
done
echo -n "config 3 world 8 ring 12 batch 4: "; YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 3 8 residentc 12 2>&1 | tail -1 | cut -c1-260
