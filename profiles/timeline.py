"""Kernel timeline of one frame from a rocprofv3 --kernel-trace csv (argument: the *_kernel_trace.csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('ycge::', '').replace('void ', '').split('(')[0] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_taa')]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]['End_Timestamp'])
for i in range(a, b + 1):
    r = rows[i]; s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{names[i][:34]:36s} start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}")
