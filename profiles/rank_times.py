"""Per-rank trace time of a frame tiled over `world` GPUs, emulated on one GPU (one context per rank, run one after the
other): the multi-GPU frame cannot be faster than its slowest rank."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc, w, h, ss, pose = scenes.config_scene(cfg)
flat = flatten(sc)
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
        r = RaytraceRenderer(flat, w, h, pose["fov"], ss, rank=rank, world_size=world)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        slab = torch.empty(r.tile_slab_bytes() // 4, dtype=torch.float32, device="cuda")
        ms = []
        for i in range(8):
            r.trace_tiles(slab.data_ptr(), 0, want_stats=True)
            ms.append(r.stats.trace_ms)
        times.append(float(np.mean(ms[3:])))
        r.close()
    print(f"config {cfg} world {world}: per-rank trace ms {[round(t, 3) for t in times]}  max {max(times):.3f}")
