"""Frame time of a configuration at a console size, synchronous calls, under the current environment (knobs): median / min over N frames.
    python profiles/small_frames.py CONFIG [WxH] [N] [t01]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]); size = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
t01 = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
sc, w, h, ss, pose = scenes.config_scene(cfg, t01=t01)
if size:
    w, h = (int(v) for v in size.split("x"))
r = RaytraceRenderer(flatten(sc), w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for _ in range(10): r.TryFlipAndBlit()
tr, tot = [], []
t0 = time.perf_counter()
for _ in range(n):
    r.TryFlipAndBlit(); tr.append(r.stats.trace_ms); tot.append(r.stats.total_ms)
wall = (time.perf_counter() - t0) / n * 1e3
print(f"config {cfg} {w}x{h} ss {ss}: frame {np.median(tot):.4f} ms (min {min(tot):.4f}, wall {wall:.4f}), trace {np.median(tr):.4f} (min {min(tr):.4f})")
r.close()
