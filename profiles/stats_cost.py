"""What the per-frame statistics cost a synchronous frame: ycge_render_frame with a stats record (HIP timing events around trace and TAA, two
elapsed-time queries) against stats = NULL (what the C# wrapper passes).  Wall time per frame over N frames."""
import sys, time, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
for cfg in (4, 2, 1):
    sc, w, h, ss, pose = scenes.config_scene(cfg)
    r = RaytraceRenderer(flatten(sc), w, h, pose["fov"], ss)
    r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for _ in range(20): r.TryFlipAndBlit()
    out = []
    for with_stats in (True, False, True, False):
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            r.L.ycge_render_frame(r.ctx, None, C.byref(r.stats) if with_stats else None)
        out.append(((time.perf_counter() - t0) / n * 1e3, with_stats))
    print(f"config {cfg}: " + "  ".join(f"{'stats' if s else 'NULL '} {t:.4f} ms" for t, s in out))
    r.close()
