"""A rank's share of a tiled frame, pipelined as bench.py's RCCL forms do it, EMULATED on one GPU (each rank alone on the machine, one
after the other): prints the period per frame of each rank.

    rank_flight.py CONFIG WORLD [one]           the slab form: frames on alternating trace streams, a copy standing in for the all-gather,
                                                ycge_resolve_gathered on a third stream ("one": a single trace stream)
    rank_flight.py CONFIG WORLD residentc K [RANKS]   the same loop driven from C inside the library (no scripting-language overhead per frame)
    rank_flight.py CONFIG WORLD resident K      the tile-resident form: K traces in flight over K streams (config.tile_ring = K), a copy of
                                                the rank's halo records standing in for the all-to-all, ycge_resolve_tiles_resident (TAA on
                                                the rank's own tiles + the 12-byte history slab) on another stream
A rank's trace is its longest chains with most of the machine idle around them; K traces in flight make its period per frame
max(slot time, chain / K) - what the aggregate frame rate of N GPUs is made of.  Link time is not in these numbers (DESIGN section 7 adds it)."""
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
mode = sys.argv[3] if len(sys.argv) > 3 else "two"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sc, w, h, ss, pose = scenes.config_scene(cfg)
flat = flatten(sc)
N = 200
periods = []
issue = []
ranks = range(world) if len(sys.argv) <= 5 else [int(x) for x in sys.argv[5].split(',')]
for rank in ranks:
    if mode == "residentc":
        # the same loop driven from C inside the library (ycge_debug_resident_loop): the host cost per frame is the library's and the driver's,
        # not Python's - what a C# / C++ host would see
        import ctypes as C
        r = RaytraceRenderer(flat, w, h, pose["fov"], ss, rank=rank, world_size=world, tile_ring=K)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        fn = r.L.ycge_debug_resident_loop; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        per, iss = C.c_double(), C.c_double()
        rc = fn(r.ctx, N, C.byref(per), C.byref(iss))
        assert rc == 0, (rc, r.L.ycge_last_error(r.ctx))
        periods.append(per.value); issue.append(iss.value)
        r.close()
        continue
    if mode == "resident":
        r = RaytraceRenderer(flat, w, h, pose["fov"], ss, rank=rank, world_size=world, tile_ring=K)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        s_cnt, r_cnt = r.halo_counts()
        ns, nr = max(1, sum(s_cnt)), max(1, sum(r_cnt))
        send = [torch.zeros(ns * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
        recv = [torch.zeros(nr * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
        hist = [torch.zeros(r.history_slab_bytes() // 4, dtype=torch.float32, device="cuda") for _ in range(K)]
        st = [torch.cuda.Stream() for _ in range(K)]
        sc_ = torch.cuda.Stream()
        evt = [torch.cuda.Event() for _ in range(K)]; evr = [torch.cuda.Event() for _ in range(K)]
        issued = []
        def resolve(k):
            with torch.cuda.stream(sc_):
                sc_.wait_event(evt[k])
                n = min(ns, nr) * 4
                recv[k][:n].copy_(send[k][:n], non_blocking=True)      # stands in for the all-to-all of halo records (as many bytes as this rank receives)
                r.resolve_tiles_resident(recv[k].data_ptr(), hist[k].data_ptr(), sc_.cuda_stream)
                evr[k].record(sc_)
        def frame(i):
            k = i % K
            if len(issued) == K - 1 + 1:          # the ring is full: the oldest frame is resolved before the next trace is queued (as a rank's loop does)
                resolve(issued.pop(0))
            with torch.cuda.stream(st[k]):
                st[k].wait_event(evr[k])            # the send / history buffers of slot k are free again
                r.trace_tiles_resident(send[k].data_ptr(), st[k].cuda_stream)
                evt[k].record(st[k])
            issued.append(k)
        def drain():
            while issued: resolve(issued.pop(0))
    else:
        one = mode == "one"
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
        def drain(): pass
    for i in range(12): frame(i)
    drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12, 12 + N): frame(i)
    drain()
    t_issue = time.perf_counter() - t0          # what the HOST needed to queue the frames (the loop is host-bound where this is the period)
    torch.cuda.synchronize()
    periods.append((time.perf_counter() - t0) / N * 1e3)
    issue.append(t_issue / N * 1e3)
    r.close()
label = f"tile-resident form, {K} traces in flight" + (" (loop driven from C)" if mode == "residentc" else "") if mode.startswith("resident") else ("one trace stream" if mode == "one" else "two trace streams")
print(f"config {cfg} world {world} {label}: per-rank period ms/frame {[round(p, 3) for p in periods]}  max {max(periods):.3f}; host issue time per frame {[round(p, 3) for p in issue]}")
