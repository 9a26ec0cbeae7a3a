#!/bin/bash
# round 5, second call: the suite on the build that never lets the GPU write caller heap memory; small-frame A/Bs; config 5's counting twin
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out; OUT=$REPO/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 > $OUT/r5_pytest_2.log 2>&1; echo "pytest rc=$?"; tail -34 $OUT/r5_pytest_2.log
echo "== config 1 (80x45): default | persistent extend stage as before | tree_phase instead of analytic_walk | single launch | single launch + tree_phase"
for v in "-" "YCGE_PERSIST_MIN_TILES=0" "YCGE_NO_ANALYTIC_WALK=1" "YCGE_PATH=m" "YCGE_PATH=m YCGE_NO_ANALYTIC_WALK=1" "-"; do
  ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$v: "; timeout 120 python profiles/small_frames.py 1 - 300 2>&1 | tail -1 )
done
echo "== config 2"; for v in "-" "YCGE_PATH=w" "YCGE_PATH=w YCGE_NO_ANALYTIC_WALK=1" "-"; do ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$v: "; timeout 120 python profiles/small_frames.py 2 - 300 2>&1 | tail -1 ); done
echo "== voxel world, lit, by console size: extend stage (MIN_TILES=1000000) against persistent (MIN_TILES=0)"
for sz in 128x36 256x72 384x108 512x144 768x216 960x270; do for v in "YCGE_PERSIST_MIN_TILES=1000000" "YCGE_PERSIST_MIN_TILES=0"; do
  ( export $v; echo -n "$sz $v: "; timeout 200 python profiles/small_frames.py 5 $sz 60 2>&1 | tail -1 ); done; done
echo "== analytic zoo-like scene (config 1 scene) at larger consoles: analytic_walk against tree_phase, stage pipeline"
for sz in 320x90 960x270; do for v in "-" "YCGE_NO_ANALYTIC_WALK=1"; do ( if [ "$v" != "-" ]; then export $v; fi; echo -n "$sz $v: "; timeout 200 python profiles/small_frames.py 1 $sz 100 2>&1 | tail -1 ); done; done
echo "== config 5 with the counting twin"
timeout 400 python bench.py --config 5 --t01 0.5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lit', d['ms_per_step'], 'ms; frac', r['frac'], 'achieved', r['achieved'], 'reference walk', r.get('reference_walk',{}).get('frac')); print(json.dumps(r.get('timed_work'), indent=1))"
timeout 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dark', d['ms_per_step'], 'ms; frac', r['frac'], 'reference walk', r.get('reference_walk',{}).get('frac'), r.get('timed_work',{}).get('bytes_by_kind'))"
echo "== soak: the test that died in the last call, 12 times; then the in-flight tests 3 times"
for i in $(seq 1 12); do timeout 300 python -m pytest tests/test_gpu_timed_variants.py -m gpu -q -x -k "frames_in_flight_are_the_frames" 2>&1 | tail -1; done
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_timed_variants.py -m gpu -q -x -k "in_flight" 2>&1 | tail -1; done
