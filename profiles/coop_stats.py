"""Profiling build (-DYCGE_DBG_COOPSTAT): how often the cooperative walk runs, how many steps, how long one takes."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc, w, h, ss, pose = scenes.config_scene(cfg)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
f = r.L.ycge_debug_read_coop_stats; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p]
def read():
    a = (C.c_uint64 * 16)(); assert f(r.ctx, a) == 0; return np.array(list(a), dtype=np.float64)
for _ in range(6): r.TryFlipAndBlit()
a = read(); s0 = r.timed_steps()
n = 10; ms = []
for _ in range(n): r.TryFlipAndBlit(); ms.append(r.stats.trace_ms)
d = (read() - a) / n; steps = (r.timed_steps() - s0) / n
print(f"trace {np.median(ms):.3f} ms; per frame: coop invocations {d[0]:.0f}, rays handed over {d[5]:.0f}, loop iterations {d[1]:.0f} ({d[1]/max(1,d[0]):.1f} per invocation), "
      f"group node steps {d[2]:.0f}, group leaf steps {d[3]:.0f}; time inside {d[4]/100:.0f} us summed over wavefronts = {d[4]*10/max(1,d[1]):.0f} ns per loop iteration; "
      f"lane steps of the frame {steps:.0f}; shader clocks per iteration {d[7]/max(1,d[1]):.0f}, of which the fetch (issue to data) {d[6]/max(1,d[1]):.0f}; "
      f"by section: head {d[8]/max(1,d[1]):.0f}, node step {d[9]/max(1,d[1]):.0f}, leaf step {d[10]/max(1,d[1]):.0f}, stack {d[11]/max(1,d[1]):.0f} "
      f"(a node step runs in {100*d[2]/max(1,d[2]+d[3]):.0f} % of the group steps)")
