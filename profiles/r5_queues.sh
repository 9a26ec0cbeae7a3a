#!/bin/bash
# round 5: hardware queues.  The runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round robin: two of a rank's K trace
# streams share a queue and their launches follow each other (profiles/r05: tl_res4 - queue 5 carries every second trace).  More queues?
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out; OUT=$REPO/gpurun_out
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  for k in 4 6 8; do echo -n "K=$k: "; GPU_MAX_HW_QUEUES=$q timeout 300 python profiles/rank_flight.py 4 8 residentc $k 1,2,5 2>&1 | tail -1 | cut -c60-260; done
  echo -n "ring 12 batch 4: "; GPU_MAX_HW_QUEUES=$q YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 4 8 residentc 12 1,2,5 2>&1 | tail -1 | cut -c60-260
  echo -n "ring 8 batch 2: "; GPU_MAX_HW_QUEUES=$q YCGE_RES_LOOP_BATCH=2 timeout 300 python profiles/rank_flight.py 4 8 residentc 8 1,2,5 2>&1 | tail -1 | cut -c60-260
  echo -n "single GPU, config 4, in flight: "; GPU_MAX_HW_QUEUES=$q timeout 300 python profiles/flight_ab.py 4 300 2>&1 | tail -2 | tr '\n' ' '; echo
  echo -n "world 4 K=6: "; GPU_MAX_HW_QUEUES=$q timeout 300 python profiles/rank_flight.py 4 4 residentc 6 1,2 2>&1 | tail -1 | cut -c60-260
  echo -n "world 2 K=6: "; GPU_MAX_HW_QUEUES=$q timeout 300 python profiles/rank_flight.py 4 2 residentc 6 1 2>&1 | tail -1 | cut -c60-260
done
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_res6_q8 -o t -- python3 $REPO/profiles/rank_flight.py 4 8 residentc 6 1 > $OUT/tl_res6_q8.log 2>&1; python3 $REPO/profiles/timeline.py $OUT/tl_res6_q8 800 70 | grep -v "gather_halo\|pack_history\|copyBuffer\|scatter_halo"
cd $REPO/profiles/micro && ./atrous_chain
