#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_c5q -o t -- python3 $REPO/bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv
for row in csv.DictReader(open("$REPO/gpurun_out/prof_c5q/t_kernel_stats.csv")):
    n=row['Name'].replace('ycge::','').replace('void ','').split('(')[0]
    print(f"{n:40s} calls={row['Calls']:>4s} avg_us={float(row['AverageNs'])/1e3:9.1f} min={float(row['MinNs'])/1e3:9.1f} max={float(row['MaxNs'])/1e3:9.1f}")
PY
