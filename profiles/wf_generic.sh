#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
export YCGE_PATH=wavefront YCGE_GENERIC_WALK=1
python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('wavefront generic', d['value'],'Mrays/s', d['ms_per_step'],'ms')"
rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_wfg -o t -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv
for row in csv.DictReader(open("$REPO/gpurun_out/prof_wfg/t_kernel_stats.csv")):
    n=row['Name'].replace('ycge::','').replace('void ','').split('(')[0]
    if 'true' in n.split('<')[-1].split(',')[0]: continue
    print(f"{n:40s} calls={row['Calls']:>4s} avg_us={float(row['AverageNs'])/1e3:9.1f} min={float(row['MinNs'])/1e3:9.1f} max={float(row['MaxNs'])/1e3:9.1f}")
PY
