"""Per-wavefront start/duration of the single-launch kernel k_trace (YCGE_WAVE_PROF=mega)."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["YCGE_WAVE_PROF"] = "mega"
os.environ["YCGE_PATH"] = "megakernel"
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
sc, w, h, ss, pose = scenes.config_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for _ in range(3):
    r.TryFlipAndBlit()
print("trace_ms", r.stats.trace_ms)
n_tiles = ((r.hiW + 31) // 32) * ((r.hiH + 7) // 8)
buf = np.zeros(n_tiles * 16, dtype=np.uint64)
r.L.ycge_debug_read_wave_prof.restype = C.c_int
r.L.ycge_debug_read_wave_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert r.L.ycge_debug_read_wave_prof(r.ctx, buf.ctypes.data, buf.size) == 0
p = buf.reshape(-1, 4).astype(np.int64)
dur = p[:, 1] - p[:, 0]
xcc = p[:, 3] & 0xf
lg = (p[:, 3] >> 32) & 0xf          # log2(parts) of the block's schedule entry (part 0 reports)
steps = p[:, 2] >> 32               # traversal steps of the wavefront's longest lane
p[:, 2] &= 0xffffffff
print("xcc ids seen", np.unique(xcc))
t0 = p[:, 0].min()
start = p[:, 0] - t0
end = start + dur
print("timestamps are 100 MHz ticks (10 ns). kernel span us:", end.max() / 100.0)
valid = dur > 0
print("slot time: sum of wavefront durations %.1f us = %.3f ms on 4096 slots (4 per SIMD) / %.3f ms on 3072" % (dur[valid].sum() / 100.0, dur[valid].sum() / 100.0 / 4096 / 1e3, dur[valid].sum() / 100.0 / 3072 / 1e3))
print("wave duration us pcts 50/90/99/max", np.percentile(dur, [50, 90, 99]) / 100.0, dur.max() / 100.0)
print("blocks split into 1/4/16/64 parts:", [int((lg == v).sum()) for v in (0, 2, 4, 6)])
print("last waves to finish:")
for i in np.argsort(-end)[:8]:
    print(f"  wave {i} tile {i//4} block {p[i,2]} xcc {xcc[i]} start_us {start[i]/100:.1f} dur_us {dur[i]/100:.1f} end_us {end[i]/100:.1f}")
print("longest waves (part 0 of each block reports):")
for i in np.argsort(-dur)[:12]:
    print(f"  wave {i} tile {i//4} block {p[i,2]} xcc {xcc[i]} lg_parts {lg[i]} start_us {start[i]/100:.1f} dur_us {dur[i]/100:.1f} end_us {end[i]/100:.1f} max_lane_steps {steps[i]} -> {dur[i]*10.0/max(1,steps[i]):.0f} ns per step of the longest lane")
ev = np.concatenate([np.stack([start, np.ones(len(p))], 1), np.stack([end, -np.ones(len(p))], 1)]).astype(np.float64)
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
for f in (0.05, 0.1, 0.2, 0.3, 0.5, 0.7, 0.9):
    i = np.searchsorted(ev[:, 0], end.max() * f)
    print(f"waves in flight at {f:.2f} of span: {int(conc[min(i, len(conc) - 1)])}")

heavy = steps >= 256
print("blocks with >= 256 steps:", int(heavy.sum()), " ns per longest-lane step: median", np.median(dur[heavy] * 10.0 / steps[heavy]), "p10", np.percentile(dur[heavy] * 10.0 / steps[heavy], 10), "p90", np.percentile(dur[heavy] * 10.0 / steps[heavy], 90))
late = heavy & (start > np.percentile(start[heavy], 90))
print("  the 10% that start last:", np.median(dur[late] * 10.0 / steps[late]), "ns/step; the 10% that start first:", np.median(dur[heavy & (start <= np.percentile(start[heavy], 10))] * 10.0 / steps[heavy & (start <= np.percentile(start[heavy], 10))]))
