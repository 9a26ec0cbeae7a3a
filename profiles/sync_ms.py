"""The synchronous call on one config: ms per frame (wall) and the trace launch's HIP-event time, median / min / p95 / p99 / max over N frames.
    python profiles/sync_ms.py <config> [frames] [label]      (YCGE_LIB / YCGE_* knobs select what is measured; A/B inside one gpurun call)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
label = sys.argv[3] if len(sys.argv) > 3 else ""
sc, w, h, ss, pose = scenes.config_scene(cfg)
r = RaytraceRenderer(flatten(sc), w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
import os
same = int(os.environ.get("SAME_FRAME", "0"))          # SAME_FRAME=k: every timed frame is frame number k again - the same rays, the same chains: what is left of the spread is the machine's
for _ in range(12):
    r.TryFlipAndBlit()
tr, fr = [], []
for _ in range(N):
    if same: r.set_frame_counter(same - 1)
    if os.environ.get("NO_STATS"):          # the call the C# wrapper makes: no statistics asked for, no timing events recorded
        t0 = time.perf_counter(); rc = r.L.ycge_render_frame(r.ctx, None, None); fr.append((time.perf_counter() - t0) * 1e3); tr.append(0.0); assert rc == 0
        continue
    t0 = time.perf_counter(); r.TryFlipAndBlit(); fr.append((time.perf_counter() - t0) * 1e3); tr.append(r.stats.trace_ms)
tr, fr = np.array(tr), np.array(fr)
q = lambda a: "median %.4f min %.4f mean %.4f p95 %.4f p99 %.4f max %.4f" % (np.median(a), a.min(), a.mean(), np.percentile(a, 95), np.percentile(a, 99), a.max())
print(f"{label} config {cfg}: frame ms {q(fr)} | trace ms {q(tr)}")
