"""Per-rank trace time of a frame tiled over `world` GPUs, emulated on one GPU (one context per rank, run one after the
other): the multi-GPU frame cannot be faster than its slowest rank.

    rank_times.py CONFIG [WORLDS [RANK_STRIDE]]   e.g. "4 8,4,2"; "4 256 16" runs every 16th rank of 256 - a nearly empty machine,
    i.e. what the heaviest blocks take with nothing beside them (the floor no schedule goes under)."""
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
worlds = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for world in worlds:
    times = []
    for rank in range(0, world, stride):
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
