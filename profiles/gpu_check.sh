#!/bin/bash
# quick GPU iteration: parity tests on both paths + bench + per-kernel timing
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
YCGE_PATH=wavefront timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest(wavefront) rc=$?"; tail -3 gpurun_out/pytest_gpu.log
YCGE_PATH=megakernel timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_mega.log 2>&1; echo "pytest(megakernel) rc=$?"; tail -3 gpurun_out/pytest_gpu_mega.log
YCGE_PATH=wavefront python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/bench.err | tee gpurun_out/bench_wf.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WF  ', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline'])"
YCGE_PATH=megakernel python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> gpurun_out/bench.err | tee gpurun_out/bench_mega.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MEGA', d['value'],'Mrays/s', d['ms_per_step'],'ms', d['roofline'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_quick -o t -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv
for row in csv.DictReader(open("$REPO/gpurun_out/prof_quick/t_kernel_stats.csv")):
    n=row['Name'].replace('ycge::','').replace('void ','').split('(')[0]
    print(f"{n:40s} calls={row['Calls']:>4s} avg_us={float(row['AverageNs'])/1e3:9.1f} min={float(row['MinNs'])/1e3:9.1f} max={float(row['MaxNs'])/1e3:9.1f}")
PY
