"""A rank's share of a tiled frame, pipelined as bench.py's RCCL form does it, emulated on one GPU: frames queued on alternating trace
streams, a copy standing in for the all-gather, ycge_resolve_gathered on a third stream.  Prints the period per frame of each rank.
    rank_flight.py CONFIG WORLD [one]      ("one": a single trace stream, the form before two traces could overlap)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
one = len(sys.argv) > 3 and sys.argv[3] == "one"
sc, w, h, ss, pose = scenes.config_scene(cfg)
flat = flatten(sc)
N = 200
periods = []
for rank in range(world):
    r = RaytraceRenderer(flat, w, h, pose["fov"], ss, rank=rank, world_size=world, slab_albedo=False)
    r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    nb = r.tile_slab_bytes() // 4
    slabs = [torch.empty(nb, dtype=torch.float32, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(world * nb, dtype=torch.float32, device="cuda") for _ in range(2)]
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    if one: st[1] = st[0]
    sc_ = torch.cuda.Stream()
    evt = [torch.cuda.Event() for _ in range(2)]; evr = [torch.cuda.Event() for _ in range(2)]
    def frame(i):
        k = i & 1
        with torch.cuda.stream(st[k]):
            st[k].wait_event(evr[k])
            r.trace_tiles(slabs[k].data_ptr(), st[k].cuda_stream)
            evt[k].record(st[k])
        with torch.cuda.stream(sc_):
            sc_.wait_event(evt[k])
            gathered[k][rank * nb:(rank + 1) * nb].copy_(slabs[k], non_blocking=True)      # stands in for the all-gather (this rank's part)
            r.resolve_gathered(gathered[k].data_ptr(), sc_.cuda_stream)
            evr[k].record(sc_)
    for i in range(8): frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 8 + N): frame(i)
    torch.cuda.synchronize()
    periods.append((time.perf_counter() - t0) / N * 1e3)
    r.close()
print(f"config {cfg} world {world} {'one trace stream' if one else 'two trace streams'}: per-rank period ms/frame {[round(p, 3) for p in periods]}  max {max(periods):.3f}")
