"""BASELINE.md's per-config table from a round's bench lines: python profiles/baseline_table.py profiles/r04 r04z"""
import json, sys
from pathlib import Path
d, tag = Path(sys.argv[1]), sys.argv[2]
names = {1: "1 Cornell Box, 80×45 console, 1 spp (CPU plumbing)", 2: "2 Mirror spheres on checker, 640×360, 1 spp", 3: "3 Bunny 69,451 tris, 1280×720, 1 spp",
         4: "4 Dragon-class 871,200-tri stand-in, 1920×1080, 1 spp", 5: "5 Voxel grid 544×256×544, 1920×1080 out, ss=2 (4 spp) + TAA, dark (t01 0.25: both lights at intensity 0)",
         "5lit": "5 … at noon (t01 0.5: sun 282 353)"}
files = {1: f"bench_{tag}_cfg1.json", 2: f"bench_{tag}_cfg2.json", 3: f"bench_{tag}_cfg3.json", 4: f"bench_{tag}.json", 5: f"bench_{tag}_cfg5.json", "5lit": f"bench_{tag}_cfg5_t050.json"}
sp = lambda v: f"{v:,.0f}".replace(",", " ")
print("| Config (from BASELINE.json) | Trace grid | Rays / frame (traced by the timed kernels) | CPU Mrays/s trace only (per-frame min / median / max) | CPU ms/frame: trace + serial TAA (parallel TAA) | GPU ×1 Mrays/s over traced rays (by the reference's count; in flight) | GPU ×1 ms/frame (trace median / min; in flight) | CPU whole frame ÷ GPU frame |")
print("|---|---|---|---|---|---|---|---|")
for k, f in files.items():
    try:
        j = json.load(open(d / f))
    except Exception as e:
        print(f"| {names[k]} | missing: {e} |"); continue
    c = j.get("cpu_baseline") or {}
    pf = c.get("per_frame_mrays", {})
    fl = j.get("frames_in_flight") or {}
    grid = j["config"].get("trace_grid", "?")
    traced = j.get("rays_traced_per_frame", j["rays_per_frame"])
    rays = sp(j["rays_per_frame"]) + ("" if traced == j["rays_per_frame"] else f" ({sp(traced)})")
    cpu = f"{c.get('value', float('nan')):.1f} ({pf.get('min', 0):.1f} / {pf.get('median', 0):.1f} / {pf.get('max', 0):.1f})" if c else "—"
    cpums = f"{c.get('trace_ms_per_frame', 0):.1f} + {c.get('taa_serial_ms', 0):.2f} ({c.get('taa_parallel_ms', 0):.1f})" if c else "—"
    ref = j.get("value_reference_ray_count")
    gpu = f"{sp(j['value'])}" + (f" ({sp(ref)}" if ref and abs(ref - j['value']) > 1 else " (=") + f"; {sp(fl.get('value', 0))})"
    ms = f"{j['ms_per_step']:.4g} ({j['trace_ms']['median']:.4g} / {j['trace_ms']['min']:.4g}; {fl.get('ms_per_step', 0):.4g})"
    print(f"| {names[k]} | {grid} | {rays} | {cpu} | {cpums} | {gpu} | {ms} | {j.get('gpu_over_cpu', '—')}× |")
