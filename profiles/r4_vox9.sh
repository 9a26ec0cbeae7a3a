#!/bin/bash
# lit config 5 at the static bench pose: kernel durations of the stage pipeline (one frame in launch order)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out/vox
cd /tmp && export TMPDIR=/tmp
for t in ${1:-0.5}; do
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving --config 5 --t01 $t"
rm -rf $REPO/gpurun_out/vox/t$t
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/vox/t$t -o trace -- $BENCH > $REPO/gpurun_out/vox/trace_$t.log 2>&1
echo "== t01 $t trace rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$REPO/gpurun_out/vox/t$t/**/trace_kernel_trace.csv", recursive=True)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", "")) for r in csv.DictReader(open(f[0])))
idx = [i for i, r in enumerate(rows) if "k_wf_primary<false" in r[2]]
for i0 in idx[-2:]:
    t0 = rows[i0][0]
    for s, e, k, st in rows[i0:i0 + 9]:
        print(f"   {k[:50]:50s} begin {(s - t0)/1e3:9.1f} end {(e - t0)/1e3:9.1f}  dur {(e - s)/1e3:8.1f}")
    print()
PY
done
