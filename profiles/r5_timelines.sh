#!/bin/bash
# round 5: kernel timelines (rocprofv3 --kernel-trace) of the small frames and of a rank's tile-resident loop; suite durations
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out; OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
  timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_cfg$c -o t -- python3 $REPO/bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-post --no-flight --no-moving > $OUT/tl_cfg$c.log 2>&1
  echo "== config $c: the last synchronous frames"; python3 $REPO/profiles/timeline.py $OUT/tl_cfg$c -30 30
  tail -1 $OUT/tl_cfg$c.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', d['ms_per_step'], 'ms/frame trace', d.get('trace_ms'))"
done
echo "== a rank of 8, tile-resident ring of 4, frame by frame (loop driven from C)"
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_res4 -o t -- python3 $REPO/profiles/rank_flight.py 4 8 residentc 4 0 > $OUT/tl_res4.log 2>&1; tail -1 $OUT/tl_res4.log
python3 $REPO/profiles/timeline.py $OUT/tl_res4 -90 90
echo "== ... batches of 4 (ring 12)"
YCGE_RES_LOOP_BATCH=4 timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/tl_res_b4 -o t -- python3 $REPO/profiles/rank_flight.py 4 8 residentc 12 0 > $OUT/tl_res_b4.log 2>&1; tail -1 $OUT/tl_res_b4.log
python3 $REPO/profiles/timeline.py $OUT/tl_res_b4 -90 90
cd $REPO
echo "== config 5 with the counting twin"; timeout 400 python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline'], indent=1)[:3000])"
timeout 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dark: frac', r['frac'], 'reference walk', r.get('reference_walk',{}).get('frac'), r.get('timed_work',{}).get('bytes_by_kind'))"
echo "== GPU suite with durations"; timeout 1500 python -m pytest tests -m gpu -q -x --durations=30 > $OUT/r5_pytest_durations.log 2>&1; echo "pytest rc=$?"; tail -45 $OUT/r5_pytest_durations.log
