"""Per-wavefront duration / iteration histogram of k_wf_primary (YCGE_WAVE_PROF=1, counting variant)."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["YCGE_WAVE_PROF"] = os.environ.get("PROF_STAGE", "primary")
os.environ.setdefault("YCGE_PATH", "wavefront")
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
sc, w, h, ss, pose = scenes.config_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
COUNT = os.environ.get("PROF_COUNT", "0") == "1"
r = RaytraceRenderer(sc, w, h, pose["fov"], ss, count_work=COUNT)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for _ in range(3):
    r.TryFlipAndBlit()
n_tiles = ((r.hiW + 31) // 32) * ((r.hiH + 7) // 8)
buf = np.zeros(n_tiles * 16, dtype=np.uint64)
r.L.ycge_debug_read_wave_prof.restype = C.c_int
r.L.ycge_debug_read_wave_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = r.L.ycge_debug_read_wave_prof(r.ctx, buf.ctypes.data, buf.size)
assert rc == 0, rc
p = buf.reshape(-1, 4).astype(np.int64)
dur = p[:, 1] - p[:, 0]
t0 = p[:, 0].min()
print("waves", len(p), "kernel span (cycles)", p[:, 1].max() - t0)
print("duration cycles pcts 50/90/99/99.9/max", np.percentile(dur, [50, 90, 99, 99.9]), dur.max())
c0, c1, c2, c3 = p[:, 2] & 0xffffffff, p[:, 2] >> 32, p[:, 3] & 0xffffffff, p[:, 3] >> 32
for name, c in (("node/unified iterations", c0), ("leaf phases", c1), ("grid setups", c2), ("dda steps", c3)):
    print(f"wave-level {name}: pcts 50/90/99/max", np.percentile(c, [50, 90, 99]), c.max(), "mean", c.mean())
p[:, 2], p[:, 3] = c0 + c2, c1 + c3
heavy = np.argsort(-dur)[:10]
for i in heavy:
    print("wave", i, "start", p[i, 0] - t0, "dur", dur[i], "node_iters", p[i, 2], "leaf_phases", p[i, 3], "cycles/iter", dur[i] / max(1, p[i, 2] + p[i, 3]))
# concurrency over time
ev = np.concatenate([np.stack([p[:, 0] - t0, np.ones(len(p))], 1), np.stack([p[:, 1] - t0, -np.ones(len(p))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
span = ev[-1, 0]
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    idx = np.searchsorted(ev[:, 0], span * frac)
    print(f"waves in flight at {frac:.2f} of span: {int(conc[min(idx, len(conc)-1)])}")
print("mean waves in flight", dur.sum() / span)
