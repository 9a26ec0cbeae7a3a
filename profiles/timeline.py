"""A rocprofv3 --kernel-trace CSV as a timeline: per launch its queue, start (us from the first launch of the window), duration and the gap
to the end of the launch before it ON THE DEVICE (any queue).  What a loop of small launches is made of.
    python profiles/timeline.py <dir with *kernel_trace.csv> [first launch of the window = -120] [count = 120] [--by-queue]"""
import csv, glob, os, re, sys
d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -120
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), re.sub(r"\(.*", "", r["Kernel_Name"])[:44]))
rows.sort()
win = rows[first:][:count] if first < 0 else rows[first:first + count]
if not win:
    sys.exit("no launches")
t0 = win[0][0]
end = win[0][0]
busy = 0
print(f"{'start us':>10} {'dur us':>8} {'gap us':>8}  queue  kernel        ({len(rows)} launches in the trace; window of {len(win)})")
for s, e, q, k in win:
    gap = (s - end) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {q:>5}  {k}")
    if e > end:
        busy += (e - max(s, end)); end = e
span = (end - t0) / 1e3
print(f"window: {span:.1f} us, device busy {busy / 1e3:.1f} us ({100 * busy / 1e3 / span:.0f} %), idle {span - busy / 1e3:.1f} us")
