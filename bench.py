"""bench.py — headline benchmark of the MI355X ray-trace core (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 4]

A "step" is one frame of the hot path (ray-gen + per-pixel trace + TAA; with N > 1 also the
RCCL all-gather of the tile slabs and the un-permute) over the configuration BASELINE.json quotes
the metric on: config 4, the Dragon-class mesh (871,200-triangle procedural stand-in for the
missing xyzrgb_dragon.obj) at a 1920x1080 trace grid, 1 spp.  Scene, BVH and all per-pixel buffers
are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

value   = Mrays/s over ALL rays: calls to Scene.Hit + Scene.Occluded (primary, shadow, bounce) per
          frame / frame time, whole job (all ranks).  Ray counts are exact: the timed frames are
          re-run with the counting kernel variant afterwards (same frame numbers, untimed).
roofline= ALGORITHMIC bytes of the trace per launch (32*N_box + 48*N_tri + 64*N_prim +
          1*N_vox + 118*pixels, counters from the counting replay) / its mean launch duration from
          HIP events recorded around the kernel on its own stream inside the timed region (N > 1: in a few
          extra frames right after it, so that the timed region has no per-step host synchronisation):
          `achieved` / `frac`.  That figure prices work, not memory traffic - most of those bytes are cache
          hits.  What the hardware counters of the same build say is reported beside it, from the committed
          rocprofv3 PMC passes (profiles/r02/pmc_config<N>.json, written by profiles/run_profiles.sh +
          summarize.py): `traffic` = fabric bytes per launch (FETCH_SIZE doubled per the gfx950 note of
          MI355X_MICROARCH.md, WRITE_SIZE scaled by the copy calibration), `hbm_counter_gbs`,
          `frac_hbm_counter`, `lanes_active`, `valu_busy`, `wait_frac`; `bound` says what those show.
cpu_baseline = the oracle (scalar C++ restatement of the reference, all host threads) on a bounded
          sample of the same workload, rank 0, N = 1 only: `value` = trace only (what `value` of the GPU line
          counts rays over is the whole frame, so `whole_frame_serial_taa` is the like-for-like figure); the
          reference's TAA is one serial loop (`taa_serial_ms`), `taa_parallel_ms` is the same loop in row bands
          (SURVEY 8d asks for both).  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes(st, pixels):
    return 32 * st["n_box"] + 48 * st["n_tri"] + 64 * st["n_prim"] + 1 * st["n_vox"] + 118 * pixels


def stats_dict(s):
    return {k: int(getattr(s, k)) for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox")}


WORKLOADS = {1: "Cornell box", 2: "mirror spheres on checker", 3: "Stanford bunny 69,451 tris",
             4: "Dragon-class stand-in mesh 871,200 tris (seeded torus-knot, dragon OBJ is a missing blob)",
             5: "voxel world 544x256x544"}
METRIC_SHAPES = {1: "Cornell box 80x90 1spp", 2: "mirror spheres 640x360 1spp", 3: "Bunny BVH 1280x720 1spp",
                 4: "Dragon-class BVH 1920x1080 1spp", 5: "voxel volume grid 1920x1080 4spp + TAA"}


def load_pmc(config):
    """Counter summary of the trace kernels of THIS config from the committed rocprofv3 PMC passes (profiles/r02/), or None.
    Produced on the GPU box by profiles/run_profiles.sh -> summarize.py --json; bench.py itself never runs a profiler."""
    p = ROOT / "profiles" / "r02" / f"pmc_config{config}.json"
    try:
        return json.loads(p.read_text()) if p.exists() else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-post", action="store_true", help="skip the frames WITH the denoise/exposure/tonemap stage (reported apart as post_stage)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    multi = world > 1 or bool(os.environ.get("YCGE_BENCH_FORCE_TILED"))     # the env knob runs the tiled path (slab + all-gather + resolve) with one rank

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-trace path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from yetanotherconsolegameengine_amd import abi, build, scenes
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    from yetanotherconsolegameengine_amd.scene import flatten
    if rank == 0:
        build.build_library()
    if multi:
        dist.barrier()

    scene, fbw, fbh, ss, pose = scenes.config_scene(args.config)
    flat = flatten(scene)
    hiW, hiH = fbw * ss, fbh * 2 * ss
    pixels = hiW * hiH

    def make(count):
        # the multi-GPU frame ends with TAA (the metric's frame): lean slabs, no albedo plane in the all-gather (32 instead of 44 B per pixel)
        r = RaytraceRenderer(flat, fbw, fbh, pose["fov"], ss, count_work=count, device=local_rank, rank=rank, world_size=world, slab_albedo=not multi)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        return r

    r = make(False)
    stream = torch.cuda.current_stream()
    slab = all_slabs = None
    # Several GPUs: the trace of frame N+1 does not depend on the all-gather / resolve (TAA) of frame N - the library keeps the
    # trace's outputs apart from the resolved frame - so the two run on two streams with double-buffered slabs: a frame
    # costs max(trace, gather + resolve) instead of their sum.  Every frame is still traced, gathered and resolved inside
    # the timed region (both streams are drained before the clock stops).  YCGE_BENCH_PIPELINE=0: one stream, in sequence.
    pipelined = multi and os.environ.get("YCGE_BENCH_PIPELINE", "1") != "0"
    if multi:
        nb = r.tile_slab_bytes()
        slabs = [torch.empty(nb // 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.empty(world * (nb // 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        slab, all_slabs = slabs[0], gathered[0]
        s_trace, s_comm = torch.cuda.Stream(), torch.cuda.Stream()
        ev_traced = [torch.cuda.Event() for _ in range(2)]
        ev_resolved = [torch.cuda.Event() for _ in range(2)]
        n_issued = [0]

    def step(rr, want_stats=False):
        if not multi:
            rr.TryFlipAndBlit()
            return rr.stats.trace_ms
        if not pipelined or want_stats:
            rr.trace_tiles(slab.data_ptr(), stream.cuda_stream, want_stats=want_stats)
            t = rr.stats.trace_ms if want_stats else 0.0
            dist.all_gather_into_tensor(all_slabs, slab)
            rr.resolve_gathered(all_slabs.data_ptr(), stream.cuda_stream)
            return t
        k = n_issued[0] & 1
        n_issued[0] += 1
        with torch.cuda.stream(s_trace):
            s_trace.wait_event(ev_resolved[k])           # slab k was last read by the gather of two frames ago
            rr.trace_tiles(slabs[k].data_ptr(), s_trace.cuda_stream)
            ev_traced[k].record(s_trace)
        with torch.cuda.stream(s_comm):
            s_comm.wait_event(ev_traced[k])
            dist.all_gather_into_tensor(gathered[k], slabs[k])
            rr.resolve_gathered(gathered[k].data_ptr(), s_comm.cuda_stream)
            ev_resolved[k].record(s_comm)
        return 0.0

    def fence():
        if multi:
            s_trace.synchronize(); s_comm.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(r)
    fence()
    first_frame = args.warmup + 1
    trace_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # one GPU: ycge_render_frame times k_trace with HIP events on its own stream as part of the call.  Several GPUs:
        # no per-step host synchronisation inside the timed region (trace on one stream, all-gather -> resolve on another, see above)
        trace_ms.append(step(r, want_stats=not multi))
    fence()
    elapsed = time.perf_counter() - t0
    if multi:       # kernel duration for the roofline line: a few extra, untimed frames with event timing
        trace_ms = [step(r, want_stats=True) for _ in range(4)][1:]
        fence()
    if multi:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- exact work of the timed frames: counting replay (untimed)
    rc = make(True)
    rc.set_frame_counter(first_frame - 1)
    tot = {k: 0 for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox")}
    for _ in range(args.steps):
        if multi:
            rc.trace_tiles(slab.data_ptr(), stream.cuda_stream, want_stats=True)
        else:
            rc.TryFlipAndBlit()
        for k, v in stats_dict(rc.stats).items():
            tot[k] += v
    if multi:
        tt = torch.tensor([tot[k] for k in sorted(tot)], dtype=torch.int64, device="cuda")
        dist.all_reduce(tt)
        tot = dict(zip(sorted(tot), [int(x) for x in tt.tolist()]))
    rc.close()

    per_frame = {k: v / args.steps for k, v in tot.items()}
    mrays = tot["n_rays"] / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    mean_trace_ms = float(np.mean(trace_ms)) if trace_ms and trace_ms[0] > 0 else None
    # roofline of the dominant kernel (k_trace) on THIS rank's share of the frame
    my_alg = algorithmic_bytes({k: v / world for k, v in per_frame.items()}, pixels / world)
    roof = None
    if mean_trace_ms:
        ach = my_alg / (mean_trace_ms * 1e-3) / 1e9
        # which launches trace_ms brackets: the single-launch kernel k_trace with k_trace_fan beside it for the head of the schedule
        # (two streams forked from and joined to the frame's stream: the duration is their makespan; scenes without heavy blocks
        # run k_trace alone), or the stage pipeline of scenes with a real top-level tree (config 5)
        kernel = "k_wf_* stages" if args.config == 5 else "k_trace" if args.config in (1, 2) else "k_trace + k_trace_fan (concurrent)"
        roof = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "achieved_is": "algorithmic bytes (SURVEY 8d) / launch time - mostly cache hits, NOT memory traffic",
                "algorithmic_bytes_per_launch": int(my_alg), "mean_launch_ms": round(mean_trace_ms, 4)}
        pmc = load_pmc(args.config) if world == 1 else None
        if pmc:
            # counters of the same kernels from the committed PMC passes; the rate uses THIS run's launch time
            t = pmc.get("traffic_bytes_per_launch")
            roof["traffic"] = int(t) if t else None
            if t:
                roof["hbm_counter_gbs"] = round(t / (mean_trace_ms * 1e-3) / 1e9, 1)
                roof["frac_hbm_counter"] = round(t / (mean_trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            for k in ("fetch_bytes_x2", "write_bytes_calibrated", "scratch_bytes_per_lane", "lanes_active", "valu_busy", "wait_frac", "l2_hit_rate"):
                if pmc.get(k) is not None:
                    roof[k] = pmc[k]
            roof["bound"] = pmc.get("bound", "latency")
            roof["bound_evidence"] = pmc.get("bound_evidence")
            roof["pmc_source"] = pmc.get("source")

    post = None
    if not args.no_post and world == 1:        # the frame the C# wrapper asks for (SDR out); outside the headline metric, which SURVEY 8d times through TAA
        ms = []
        for _ in range(5):
            r.TryFlipAndBlit(want_sdr=True)
            ms.append((float(r.stats.trace_ms), float(r.stats.taa_ms), float(r.stats.post_ms), float(r.stats.total_ms)))
        ms = np.array(ms[1:])            # the first frame builds the in-place level schedule
        post = {"trace_ms": round(float(ms[:, 0].mean()), 4), "taa_ms": round(float(ms[:, 1].mean()), 4),
                "post_ms": round(float(ms[:, 2].mean()), 4), "frame_ms_with_sdr_readback": round(float(ms[:, 3].mean()), 4),
                "what": "ycge_render_frame with an SDR buffer: + A-trous denoise, auto-exposure, tonemap/downsample, read-back (4 frames)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_binding as ob
        threads = os.cpu_count() or 1
        o = ob.OracleRenderer(scene, fbw, fbh, ss, pose, flat=flat)
        o.set_frame_counter(first_frame - 1)
        rays = 0; t_trace = 0.0; t_taa = 0.0; frames = 0
        while t_trace + t_taa < args.cpu_seconds and frames < args.steps:
            o.render(stages=1, threads=threads)
            rays += int(o.stats.n_rays); t_trace += o.stats.trace_ms * 1e-3; t_taa += o.stats.taa_ms * 1e-3; frames += 1
        o.set_taa_threads(threads)           # the same TAA loop in row bands: identical result, "also reported parallel" 
        t_taa_par = 0.0; n_par = min(frames, 3)
        for _ in range(n_par):
            o.render(stages=1, threads=threads)
            t_taa_par += o.stats.taa_ms * 1e-3
        o.close()
        cpu = {"value": round(rays / t_trace / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"{frames} frame(s) of the same workload (frame numbers {first_frame}..{first_frame + frames - 1}): ray-gen + trace on {threads} threads "
                         f"({t_trace:.1f} s of host time), then the reference's serial TAA ({t_taa:.1f} s); {n_par} more frame(s) with the TAA in {threads} row bands",
               "value_is": "trace only (ray-gen + TraceFull)",
               "trace_ms_per_frame": round(t_trace / frames * 1e3, 2),
               "taa_serial_ms": round(t_taa / frames * 1e3, 2),
               "taa_parallel_ms": round(t_taa_par / max(1, n_par) * 1e3, 2),
               "whole_frame_serial_taa": {"value": round(rays / (t_trace + t_taa) / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round((t_trace + t_taa) / frames * 1e3, 2)}}

    if rank == 0:
        name, cus = r.device_info()
        out = {
            "metric": f"Mrays/s (all rays: Scene.Hit + Scene.Occluded calls) and ms/frame, {METRIC_SHAPES[args.config]}",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config {args.config}: " + WORKLOADS[args.config],
                       "trace_grid": f"{hiW}x{hiH}", "spp": ss * ss, "triangles": flat.n_triangles,
                       "frame": "ray-gen + trace + TAA" + (" + RCCL all-gather of tile slabs + un-permute" if multi else ""),
                       "parallelism": f"framebuffer tiles 32x8 round-robin over {world} GPU(s)" + (", one all-gather per frame; trace of frame N+1 beside gather + resolve of frame N (two streams)" if pipelined else ", one all-gather per frame" if multi else ""), "device": name, "compute_units": cus},
            "primary_mrays_per_s": round(pixels * args.steps / elapsed / 1e6, 2),
            "rays_per_frame": round(per_frame["n_rays"], 1),
            "work_per_frame": {k: round(v, 1) for k, v in per_frame.items()},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if cpu:       # like for like: whole frames (trace + TAA) on both sides
            out["gpu_over_cpu"] = round(mrays / cpu["whole_frame_serial_taa"]["value"], 2)
        if post:
            out["post_stage"] = post
        print(json.dumps(out))
    r.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
