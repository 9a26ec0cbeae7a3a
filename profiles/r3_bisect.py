"""Debug aid: N frames of a config through the timed kernels against the oracle, printing per-frame mismatch counts (no asserts)."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_binding as ob
import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]); frames = int(sys.argv[2]); div = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc, w, h, ss, pose = scenes.config_scene(cfg)
w //= div; h //= div
flat = flatten(sc)
o = ob.OracleRenderer(sc, w, h, ss, pose, flat=flat)
g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True)
g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for f in range(frames):
    o.render(stages=1, threads=32); g.TryFlipAndBlit()
    st = pu.compare_frame(o, g, check_counters=False)
    bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
    print("frame", f + 1, "fan", g.stats.fan_blocks, "trace_ms %.3f" % g.stats.trace_ms, bad or "bit-exact", flush=True)
    if bad and "current_hdr_mismatch" in bad and f == 0:
        a, b = o.read(abi.BUF_CURRENT_HDR), g.read(abi.BUF_CURRENT_HDR)
        ys, xs = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(-1))
        print("  first mismatching pixels:", list(zip(xs[:8].tolist(), ys[:8].tolist())), "oracle", a[ys[0], xs[0]], "gpu", b[ys[0], xs[0]])
        print("  mismatches by 8x8 block (top):", np.unique((ys // 8) * 10000 + xs // 8, return_counts=True)[1][:20])
