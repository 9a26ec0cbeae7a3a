#!/bin/bash
# lit config 5 at the static bench pose: kernel durations of the stage pipeline, then the round / persistent-wave knobs
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out/vox
cd /tmp && export TMPDIR=/tmp
for t in 0.5; do
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving --config 5 --t01 $t"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/vox/t$t -o trace -- $BENCH > $REPO/gpurun_out/vox/trace_$t.log 2>&1
echo "== t01 $t trace rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$REPO/gpurun_out/vox/t$t/**/trace_kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in sorted(d.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    du = sorted((e - s) / 1e3 for s, e in v)
    print(f"{k[:60]:60s} n={len(du):4d} median {du[len(du)//2]:9.1f} us  min {du[0]:9.1f} max {du[-1]:9.1f}")
# a frame's kernels in launch order (the last non-counting frame)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f[0])))
idx = [i for i, r in enumerate(rows) if r[2].startswith("void ycge::k_wf_primary<false") or r[2].startswith("ycge::k_wf_primary<false")]
if idx:
    i0 = idx[-1]; t0 = rows[i0][0]
    for s, e, k in rows[i0:i0 + 9]:
        print(f"   {k[:50]:50s} begin {(s - t0)/1e3:9.1f} end {(e - t0)/1e3:9.1f}  dur {(e - s)/1e3:8.1f}")
PY
done
cd $REPO
echo "== knobs, lit"
for v in "-" "YCGE_ROUND=4,8" "YCGE_ROUND=4,12" "YCGE_ROUND=6,16" "YCGE_ROUND=8,10" "YCGE_ROUND=8,16" "YCGE_ROUND=10,20" "YCGE_ROUND=3,6" "YCGE_PW_PER_CU=24" "YCGE_PW_PER_CU=40"; do
  echo "-- $v"
  ( if [ "$v" != "-" ]; then for kv in ${v//;/ }; do export "$kv"; done; fi
    python bench.py --config 5 --t01 0.5 --steps 10 --warmup 3 --no-cpu-baseline --no-post --no-flight --no-moving 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms/frame', d['roofline']['mean_launch_ms'])" )
done
