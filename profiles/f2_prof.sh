#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 600 python profiles/f2_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/f2_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_f2 -o t -- python3 $REPO/profiles/f2_time.py > /dev/null 2>&1
python3 - <<PY
import csv
for row in csv.DictReader(open("$REPO/gpurun_out/prof_f2/t_kernel_stats.csv")):
    n=row['Name']
    if 'bvh' in n: print(n.split('(')[0], 'calls', row['Calls'], 'avg_us', float(row['AverageNs'])/1e3, 'min', float(row['MinNs'])/1e3, 'max', float(row['MaxNs'])/1e3)
PY
