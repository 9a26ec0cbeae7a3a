#!/bin/bash
# round 5, third call: suite; a rank's period with eager resolves + one-wavefront TAA workgroups; small analytic frames by path; headline
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out; OUT=$REPO/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r5_pytest_3.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/r5_pytest_3.log
echo "== a rank's period at 8 ranks (config 4), tile-resident, loop driven from C: ring K frame by frame, eager resolves (YCGE_RES_LOOP_EAGER=0: as round 4)"
for k in 4 6 8; do for e in 1 0; do echo -n "K=$k eager=$e: "; YCGE_RES_LOOP_EAGER=$e timeout 300 python profiles/rank_flight.py 4 8 residentc $k 2>&1 | tail -1; done; done
echo "== ... batches"
for spec in "12 4" "8 4" "9 3" "15 5"; do set -- $spec; for e in 1 0; do echo -n "ring $1 batch $2 eager=$e: "; YCGE_RES_LOOP_EAGER=$e YCGE_RES_LOOP_BATCH=$2 timeout 300 python profiles/rank_flight.py 4 8 residentc $1 2>&1 | tail -1; done; done
echo "== 4 and 2 ranks"
for w in 4 2; do echo -n "world $w K=4: "; timeout 300 python profiles/rank_flight.py 4 $w residentc 4 2>&1 | tail -1; echo -n "world $w K=6: "; timeout 300 python profiles/rank_flight.py 4 $w residentc 6 2>&1 | tail -1; echo -n "world $w ring 12 batch 4: "; YCGE_RES_LOOP_BATCH=4 timeout 300 python profiles/rank_flight.py 4 $w residentc 12 2>&1 | tail -1; done
echo "== timeline, 8 ranks, batches of 4 (rank 0)"
cd /tmp && export TMPDIR=/tmp
YCGE_RES_LOOP_BATCH=4 timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_res_b4e -o t -- python3 $REPO/profiles/rank_flight.py 4 8 residentc 12 0 > $OUT/tl_res_b4e.log 2>&1; python3 $REPO/profiles/timeline.py $OUT/tl_res_b4e 600 60
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_res6e -o t -- python3 $REPO/profiles/rank_flight.py 4 8 residentc 6 0 > $OUT/tl_res6e.log 2>&1; python3 $REPO/profiles/timeline.py $OUT/tl_res6e 800 50
cd $REPO
echo "== config 1's scene by console size: stage pipeline (default) against single launch (YCGE_PATH=m)"
for sz in 80x45 160x45 320x90 640x180 960x270; do for v in "-" "YCGE_PATH=m"; do ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$sz $v: "; timeout 200 python profiles/small_frames.py 1 $sz 200 2>&1 | tail -1 ); done; done
echo "== config 2: default against the schedule built anyway"; for v in "-" "YCGE_LPT_ALWAYS=1" "-"; do ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$v: "; timeout 120 python profiles/small_frames.py 2 - 300 2>&1 | tail -1 ); done
echo "== headline"; timeout 600 python bench.py --steps 100 --cpu-seconds 5 > $OUT/r5_third_bench.json 2> $OUT/r5_third_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/r5_third_bench.json')); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms; trace', d['trace_ms'], 'post', d.get('post_stage'), 'flight', (d.get('frames_in_flight') or {}).get('ms_per_step'))"
for c in 1 2 3; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; trace', d['trace_ms'], 'in flight', (d.get('frames_in_flight') or {}).get('ms_per_step'))"; done
