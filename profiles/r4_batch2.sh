#!/bin/bash
# kernel timeline of the batched resident loop (rank 5 of 8, config 4): how long a batch's launch takes and how batches lie to each other
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp; mkdir -p $REPO/gpurun_out/batch
for spec in "12 4" "6 2"; do set -- $spec
rm -rf $REPO/gpurun_out/batch/t
YCGE_RES_LOOP_BATCH=$2 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $REPO/gpurun_out/batch/t -o trace -- python3 $REPO/profiles/rank_flight.py 4 8 residentc $1 5 > $REPO/gpurun_out/batch/log_$1_$2.txt 2>&1
echo "== ring $1 batch $2: $(grep period $REPO/gpurun_out/batch/log_$1_$2.txt | tail -1 | cut -c1-160)"
python3 - <<PY
import csv, glob
f = glob.glob("$REPO/gpurun_out/batch/t/**/trace_kernel_trace.csv", recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], "q" + r.get("Queue_Id", "") + " s" + r.get("Stream_Id", "")) for r in csv.DictReader(open(f[0]))]
for g in glob.glob("$REPO/gpurun_out/batch/t/**/trace_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20] + " " + r.get("Bytes", r.get("Size", "")), "s" + r.get("Stream_Id", "")))
rows.sort()
tr = [i for i, r in enumerate(rows) if "k_trace" in r[2]]
i0 = tr[-5]; t0 = rows[i0][0]
for s, e, k, q in rows[i0 - 6:i0 + 60]:
    print(f"   {k:42s} {q:8s} begin {(s - t0)/1e3:8.1f} end {(e - t0)/1e3:8.1f} dur {(e - s)/1e3:7.1f}")
PY
done
