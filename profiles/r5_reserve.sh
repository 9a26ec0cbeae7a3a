#!/bin/bash
# round 5: compute units kept free of the traces for the exchange + resolve stream (YCGE_RES_LOOP_CU_RESERVE), a rank of 8 on config 4
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for r in 0 8 16 32; do for k in 4 6 8; do echo -n "reserve $r K=$k: "; YCGE_RES_LOOP_CU_RESERVE=$r timeout 300 python profiles/rank_flight.py 4 8 residentc $k 1,2,5 2>&1 | tail -1 | cut -c60-200; done
  echo -n "reserve $r ring 12 batch 4: "; YCGE_RES_LOOP_CU_RESERVE=$r YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 4 8 residentc 12 1,2,5 2>&1 | tail -1 | cut -c60-200; done
echo "== more queues with the reserve"; for k in 6 8; do echo -n "queues 8 reserve 16 K=$k: "; GPU_MAX_HW_QUEUES=8 YCGE_RES_LOOP_CU_RESERVE=16 timeout 300 python profiles/rank_flight.py 4 8 residentc $k 1,2,5 2>&1 | tail -1 | cut -c60-200; done
echo "== the suite's analytic / in-flight / texture tests on the build that sends analytic scenes down the single launch"
timeout 900 python -m pytest tests -m gpu -q -x -k "analytic or zoo or textur or in_flight or empty or primitive or refus or page_locked or objects" 2>&1 | tail -4
for c in 1 2; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; trace', d['trace_ms'], 'in flight', (d.get('frames_in_flight') or {}).get('ms_per_step'), 'sdr', (d.get('post_stage') or {}).get('frame_ms_with_sdr_readback'))"; done
