"""Condense a profiles/run_profiles.sh output directory into a per-kernel text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# summarize.py <dir> [--json <file> --config N --source "<text>"]: also write the per-frame counter summary bench.py reads
json_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
cfg_n = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 4
source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else out


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


# ---------------------------------------------------------------------------------------------------------------------
# per-FRAME counter summary of the trace kernels (the launches stats.trace_ms brackets), for bench.py's roofline object
if json_path:
    import json
    TRACE = ("k_trace", "k_trace_nomesh", "k_trace_fan", "k_trace_refill", "k_wf_primary", "k_wf_extend", "k_wf_trace_p", "k_wf_shade", "k_wf_lights")
    def is_trace(k):      # the kernels the benchmark times: the NON-counting instances (bench.py's counting replay launches the <true, ...> ones)
        return k.split("<")[0] in TRACE and ("<" not in k or k.split("<")[1].split(",")[0].strip() in ("false", "false>") or k.split("<")[0] == "k_wf_shade")
    # calibration: bytes a copy of known size moves / what the counters say (copycal under the same two PMC passes)
    cal = {"fetch": 2.0, "write": 1.0, "measured": False}      # the guide's gfx950 note for reads; writes as reported
    cal_bytes = None
    for f in glob.glob(os.path.join(out, "copycal*.log")):
        for line in open(f):
            if line.startswith("copycal bytes_read_per_launch="):
                cal_bytes = float(line.split("=")[1].split()[0])
    if cal_bytes and "k_copy16" in "".join(acc.keys()):
        for k in acc:
            if k.startswith("k_copy16"):
                if "FETCH_SIZE" in acc[k]: cal["fetch"] = cal_bytes / (sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024.0); cal["measured"] = True
                if "WRITE_SIZE" in acc[k]: cal["write"] = cal_bytes / (sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024.0)
            if k.startswith("k_copy4"):
                if "FETCH_SIZE" in acc[k]: cal["fetch_4B_lanes"] = cal_bytes / (sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024.0)
                if "WRITE_SIZE" in acc[k]: cal["write_4B_lanes"] = cal_bytes / (sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024.0)
    n_frames = {}
    tot = defaultdict(float)
    lead = [k for k in acc if is_trace(k) and k.split("<")[0] in ("k_trace", "k_trace_nomesh", "k_wf_primary")]      # one launch of these per frame
    for k in lead:
        for c, vs in acc[k].items():
            n_frames[c] = max(n_frames.get(c, 0), len(vs))
    for k in acc:
        if is_trace(k):
            for c, vs in acc[k].items():
                tot[c] += sum(vs)
    per_frame = {c: v / n_frames[c] for c, v in tot.items() if n_frames.get(c)}
    dur = {}
    scratch = {}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if is_trace(short(row["Name"])):
                dur[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) / 1e3, 2)}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if is_trace(k) and "Scratch_Size" in row:
                scratch[k] = int(row["Scratch_Size"])
    g = lambda c: per_frame.get(c)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from yetanotherconsolegameengine_amd.build import source_hash
    d = {"config": cfg_n, "source": source, "source_hash": source_hash(), "kernels": dur, "calibration": cal, "counters_per_frame": {c: round(v, 1) for c, v in sorted(per_frame.items())}}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["fetch_bytes_x2"] = int(g("FETCH_SIZE") * 1024.0 * cal["fetch"])
        d["write_bytes_calibrated"] = int(g("WRITE_SIZE") * 1024.0 * cal["write"])
        d["traffic_bytes_per_launch"] = d["fetch_bytes_x2"] + d["write_bytes_calibrated"]
    if scratch:
        d["scratch_bytes_per_lane"] = scratch
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        d["lanes_active"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0), 3)
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
        d["wait_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; SQ_BUSY_CYCLES is summed over the shader engines' SQs:
        # busy fraction of the VALU pipes while the kernels are resident = 4 * active / (SIMD count * launch cycles); the launch
        # cycles come from GRBM_GUI_ACTIVE when that pass exists
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; concurrent kernels (k_trace_fan beside k_trace) overlap, so the launch's
        # span is the lead kernel's; the stage kernels of the wavefront pipeline run one after the other, so theirs add up
        span = 0.0
        for k in acc:
            if is_trace(k) and "GRBM_GUI_ACTIVE" in acc[k] and (k.split("<")[0] != "k_trace_fan"):
                span += sum(acc[k]["GRBM_GUI_ACTIVE"]) / n_frames["GRBM_GUI_ACTIVE"] / 8.0
        if span > 0:
            d["launch_cycles"] = round(span)
            d["valu_busy"] = round(4.0 * g("SQ_ACTIVE_INST_VALU") / (span * 1024.0), 3)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        d["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3)
    ev = []
    if "wait_frac" in d: ev.append(f"SQ_WAIT_ANY/SQ_WAVE_CYCLES = {d['wait_frac']}")
    if "lanes_active" in d: ev.append(f"lanes active per VALU instruction = {d['lanes_active']}")
    if "valu_busy" in d: ev.append(f"VALU pipes busy = {d['valu_busy']}")
    d["bound"] = "latency"
    d["bound_evidence"] = "; ".join(ev) + ": wavefronts parked on dependent fetches and divergent lanes, not fabric bandwidth"
    open(json_path, "w").write(json.dumps(d, indent=1) + "\n")
    print("wrote", json_path)
