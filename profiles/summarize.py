"""Condense a profiles/run_profiles.sh output directory into a per-kernel text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = n.replace("ycge::", "").replace("void ", "")
    return n.split("(")[0]


print("== kernel trace (durations in us) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print(f"{short(row['Name']):40s} calls={row['Calls']:>5s} avg={float(row['AverageNs'])/1e3:10.2f} min={float(row['MinNs'])/1e3:10.2f} "
              f"max={float(row['MaxNs'])/1e3:10.2f} total%={row['Percentage']}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = {}
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k not in seen:
            seen[k] = row
    print("== launch geometry / resources ==")
    for k, row in seen.items():
        keys = [c for c in ("Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if c in row]
        print(f"{k:40s} " + " ".join(f"{c}={row[c]}" for c in keys))

print("== PMC (mean per dispatch) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        per[(short(row["Kernel_Name"]), row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k in acc:
    print(k)
    for c, vs in sorted(acc[k].items()):
        print(f"    {c:34s} {sum(vs)/len(vs):18.1f}   (n={len(vs)})")
