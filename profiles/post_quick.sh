#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_post -o t -- python3 $REPO/profiles/post_prof.py 4 2>&1 | grep "^frame"
python3 - <<PY
import csv
for row in csv.DictReader(open("$REPO/gpurun_out/prof_post/t_kernel_stats.csv")):
    n=row['Name'].replace('ycge::','').replace('void ','').split('(')[0]
    print(f"{n:40s} calls={row['Calls']:>6s} total_ms={float(row['TotalDurationNs'])/1e6:9.3f} avg_us={float(row['AverageNs'])/1e3:9.2f} max={float(row['MaxNs'])/1e3:9.1f}")
PY
