#!/bin/bash
# kernel-trace statistics (rocprofv3 --kernel-trace --stats) of a short bench run; usage: r3_kstats.sh <config> [env...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}; CFG=${1:-5}; shift
OUT=$REPO/gpurun_out/kstats_c$CFG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-post --no-moving > $OUT/trace.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"].replace("ycge::","").replace("void ","").split("(")[0]
        print(f"{n:44s} calls={row['Calls']:>5s} avg_us={float(row['AverageNs'])/1e3:10.2f} total%={row['Percentage']}")
PY
