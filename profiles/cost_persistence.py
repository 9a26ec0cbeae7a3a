"""How well does a block's traversal cost in one frame predict the next frame's?  (schedule feedback, fan-out candidates)"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["YCGE_WAVE_PROF"] = "mega"
os.environ["YCGE_PATH"] = "megakernel"
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
sc, w, h, ss, pose = scenes.config_scene(4)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
n_tiles = ((r.hiW + 31) // 32) * ((r.hiH + 7) // 8)
r.L.ycge_debug_read_wave_prof.restype = C.c_int
r.L.ycge_debug_read_wave_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
frames = []
for f in range(10):
    r.TryFlipAndBlit()
    buf = np.zeros(n_tiles * 16, dtype=np.uint64)
    assert r.L.ycge_debug_read_wave_prof(r.ctx, buf.ctypes.data, buf.size) == 0
    p = buf.reshape(-1, 4).astype(np.int64)
    frames.append(((p[:, 2] >> 32).copy(), (p[:, 1] - p[:, 0]).copy()))      # iterations, duration (10 ns ticks)
it = np.stack([f[0] for f in frames[2:]]).astype(np.float64)       # frames x blocks
du = np.stack([f[1] for f in frames[2:]]).astype(np.float64)
print("blocks", it.shape[1], "frames", it.shape[0])
for name, m in (("iterations", it), ("duration", du)):
    cur, prev = m[1:], m[:-1]
    print(name, "correlation frame f-1 -> f:", np.mean([np.corrcoef(prev[i], cur[i])[0, 1] for i in range(len(cur))]).round(3))
for top in (20, 50, 100, 200, 400):
    hits_prev, hits_max4 = [], []
    for f in range(4, it.shape[0]):
        heavy = set(np.argsort(-du[f])[:top])
        pred_prev = set(np.argsort(-it[f - 1])[:top])
        pred_max4 = set(np.argsort(-it[f - 4:f].max(0))[:top])
        hits_prev.append(len(heavy & pred_prev) / top); hits_max4.append(len(heavy & pred_max4) / top)
    print(f"top {top} longest-running blocks of a frame found among the top {top} of: previous frame {np.mean(hits_prev):.2f}, max of previous four {np.mean(hits_max4):.2f}")
for cover in (100, 200, 400, 800, 1600):
    got = []
    for f in range(4, it.shape[0]):
        heavy = set(np.argsort(-du[f])[:40])
        got.append(len(heavy & set(np.argsort(-it[f - 4:f].max(0))[:cover])) / 40)
    print(f"the 40 longest-running blocks covered by the top {cover} of max-of-four: {np.mean(got):.2f}")
