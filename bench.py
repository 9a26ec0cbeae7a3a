"""bench.py — headline benchmark of the MI355X ray-trace core (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 4] [--camera static|orbit] [--form auto|onecall|rccl] [--t01 0.25]

A "step" is one frame of the hot path (ray-gen + per-pixel trace + TAA; with N > 1 also the reassembly of the tiles) over the
configuration BASELINE.json quotes the metric on: config 4, the Dragon-class mesh (871,200-triangle procedural stand-in for the
missing xyzrgb_dragon.obj) at a 1920x1080 trace grid, 1 spp.  Scene, BVH and all per-pixel buffers are resident in HBM before the
timed region.  Prints ONE JSON line (rank 0).

How N GPUs are driven (`--form`):
  resident one process per GPU (RANK / WORLD_SIZE in the environment), the tile-RESIDENT form - the north_star's partition: 32x8 tiles dealt
           round-robin, ycge_trace_tiles_resident -> one RCCL all-to-all of the one-pixel halo records (1.4 MB per rank at 8 ranks) ->
           ycge_resolve_tiles_resident (TAA on the rank's own tiles, history resident) -> RCCL all-gather of the resolved history (12 B per
           pixel: the frame reassembled on every rank) -> ycge_unpack_history.  --ring frame sets = traces in flight.  --batch n > 1: n
           frames of the rank's tiles per launch (ycge_trace_tiles_resident_batch) - needs the next n poses, so it is opt-in.
  rccl     one process per GPU: ycge_trace_tiles -> one RCCL all-gather of the tile slabs -> ycge_resolve_gathered on every rank.
  onecall  ONE process, ONE ycge_render_frame call per frame - the reference's shape (RaytraceEntity.cs:230): config.devices =
           0..N-1, the peers write their tiles into device 0's frame buffers over xGMI, TAA on device 0.  No RCCL.
  auto     under a launcher: resident, frame by frame.  WITHOUT a launcher and --gpus N > 1 (the plain `python bench.py --gpus N`):
           this process touches no GPU; it starts N one-process-per-GPU children itself (`python -m torch.distributed.run
           --nproc-per-node N bench.py ... --form resident`, a child process, never an exec), relays rank 0's line, and when the
           children exit non-zero starts the next form in FRESH children: resident -> rccl -> onecall.  `forms_tried` in the line says
           what ran; `rccl_world` is dist.get_world_size() after a checked all-reduce.  After a resident headline the batched
           launches (--batch 4) run as a second job whose result is attached as `batched` - never the headline: `latency_frames`
           says how many frames lie between a pose and its image.
`n_gpus` in the line is the number of devices that traced tiles this run (`device_tiles` lists their tile counts), never the flag.

value   = Mrays/s over the rays the timed kernels TRACE: calls to Scene.Hit + Scene.Occluded (primary, shadow, bounce) per frame /
          frame time, whole job.  Ray counts are exact: the timed frames are re-run with the counting kernel variant afterwards
          (same frame numbers and poses, untimed).  Shadow rays towards a light of intensity 0 - which the reference traces and
          the timed kernels do not (their contribution is a zero whatever they find) - are NOT in `value`; the rate with them
          counted is printed beside it as `value_reference_ray_count` (config 5 at the survey's day phase --t01 0.25 has both
          lights at 0; --t01 0.5 is noon, every ray traced).  `frame_ms` / `trace_ms`: median, min and mean over the timed steps.
roofline= ALGORITHMIC bytes of the trace per launch (32*N_box + 48*N_tri + 64*N_prim + 1*N_vox + 118*pixels, counters
          of the REFERENCE's traversal from the counting replay) / its mean launch duration from HIP events on the kernel's own
          stream inside the timed region: `achieved` / `frac`.  That prices work, not memory traffic - most of those bytes are
          cache hits - and it prices the reference's walk: the timed kernels stop shadow queries at the first hit and skip culled
          voxel grids.  `timed_work` is what the timed kernels themselves walked (lane steps, 72 bytes fetched per step), with
          its own rate.  The hardware counters (`traffic`, `hbm_counter_gbs`, `lanes_active`, ...) come from the committed
          rocprofv3 PMC passes and are printed only when that summary was taken from THIS build (source hash), else `pmc_stale`.
cpu_baseline = the oracle (scalar C++ restatement of the reference, all host threads) on a bounded sample of the same workload,
          rank 0, N = 1 only; per-frame spread reported.  A reported baseline, not the target.
moving_camera / post_stage = the frame the host really drives: the pose changes every frame (below and above the TAA reset
          thresholds, TemporalAA.cs:58-67) and the SDR frame (denoise, exposure, tonemap, read-back), reported beside the headline.
"""
from __future__ import annotations

import argparse
import json
import math
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
STEP_FETCH_BYTES = 72       # what one traversal step of the timed kernels fetches per lane (4 x 16 + 8 bytes: a node or a triangle-pair record)


def algorithmic_bytes(st, pixels):
    return 32 * st["n_box"] + 48 * st["n_tri"] + 64 * st["n_prim"] + 1 * st["n_vox"] + 118 * pixels


def stats_dict(s):
    return {k: int(getattr(s, k)) for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox", "n_rays_dark")}


def dist3(a):
    a = np.asarray(a, dtype=np.float64)
    d = {"median": round(float(np.median(a)), 4), "min": round(float(a.min()), 4), "mean": round(float(a.mean()), 4)}
    if a.size >= 20:        # the tail: a frame the host drives is as late as its slowest (config 3's mean sat 6 % above its median)
        d.update({"p95": round(float(np.percentile(a, 95)), 4), "p99": round(float(np.percentile(a, 99)), 4), "max": round(float(a.max()), 4)})
    return d


WORKLOADS = {1: "Cornell box", 2: "mirror spheres on checker", 3: "Stanford bunny 69,451 tris",
             4: "Dragon-class stand-in mesh 871,200 tris (seeded torus-knot, dragon OBJ is a missing blob)",
             5: "voxel world 544x256x544"}
METRIC_SHAPES = {1: "Cornell box 80x90 1spp", 2: "mirror spheres 640x360 1spp", 3: "Bunny BVH 1280x720 1spp",
                 4: "Dragon-class BVH 1920x1080 1spp", 5: "voxel volume grid 1920x1080 4spp + TAA"}


def load_pmc(config, build_hash, tag=""):
    """Counter summary of the trace kernels of THIS config from the committed rocprofv3 PMC passes, newest round first.
    Returns (summary or None, stale): stale = a summary exists but was taken from another build of the kernels.
    Produced on the GPU box by profiles/run_profiles.sh -> summarize.py --json; bench.py itself never runs a profiler."""
    stale = False
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        p = ROOT / "profiles" / rnd / f"pmc_config{config}{tag}.json"
        try:
            if p.exists():
                d = json.loads(p.read_text())
                if d.get("source_hash") == build_hash:
                    return d, False
                stale = True
        except Exception:
            pass
    return None, stale


def timed_twin_work(flat, fbw, fbh, ss, pose, first_frame, frames, device, moving, warmup, set_pose):
    """What the TIMED stage kernels of a voxel world walk, per frame, from their counting twin (build.VARIANTS['voxstat']: the non-counting
    instances + per-lane counters; never timed).  None when the variant has not been built."""
    import ctypes as C
    from yetanotherconsolegameengine_amd import abi, build
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    path = build.variant_path("voxstat")
    if not path.exists() or build.is_stale(path, build.VARIANTS["voxstat"]):
        return None
    Lt = abi.load_library(path)
    rt = RaytraceRenderer(flat, fbw, fbh, pose["fov"], ss, device=device, lib=Lt)
    rt.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    f = Lt.ycge_debug_read_batch_stats
    f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p]

    def read():
        a = (C.c_uint64 * 64)()
        if f(rt.ctx, a) != 0:
            return None
        return np.array(list(a), dtype=np.float64).reshape(8, 8)

    rt.set_frame_counter(first_frame - 1)
    a = read()
    for k in range(frames):
        set_pose(rt, warmup + k, moving)
        rt.TryFlipAndBlit()
    b = read()
    rt.close()
    if a is None or b is None:
        return None
    d = (b - a) / frames
    pixels = fbw * ss * fbh * 2 * ss
    tree = d[4, 0] + d[4, 1] + d[4, 2]
    lrec, cont = d[5, 0], d[5, 1]
    parts = {"scene_tree_steps_x64": 64 * tree, "cell_fetches_x1": d[4, 7], "light_records_64_w_r": 128 * lrec,
             "continuation_rays_48_16_w_r": 128 * cont, "primary_hits_16_w_r": 32 * pixels, "per_pixel_118": 118 * pixels}
    return {"bytes_per_launch": int(sum(parts.values())), "bytes_by_kind": {k: int(v) for k, v in parts.items()},
            "scene_tree_steps": round(tree, 1), "objects_culled_by_their_solid_box": round(d[4, 3], 1), "grids_asked": round(d[4, 4], 1), "grids_entered": round(d[4, 5], 1),
            "cell_steps": round(d[4, 6], 1), "cell_fetches": round(d[4, 7], 1), "light_records": round(lrec, 1), "continuation_rays": round(cont, 1),
            "frames": frames, "what": "per frame, every stage kernel (k_wf_primary, k_wf_trace_p, k_wf_lights, k_wf_shade) of the timed build's counting twin on the timed frames"}


def orbit_pose(pose, k):
    """Frame k of the moving camera: the eye circles the point it looks at (2 units ahead), the yaw follows.  Three steps in four
    are SMALL (0.0008 rad: 0.0016 units of translation - below MotionTransReset = MotionRotReset = 0.0025, TemporalAA.cs:58-67,
    the history is kept and the image region under each 8x8 block drifts), every fourth is LARGE (0.004 rad / 0.008 units: reset)."""
    px, py, pz = pose["pos"]
    yaw0, pitch = pose["yaw"], pose["pitch"]
    # forward of RaytraceRenderer.cs:413-417: (sin(yaw) cos(pitch), sin(pitch), -cos(yaw) cos(pitch))
    r = 2.0
    cx, cz = px + r * math.sin(yaw0) * math.cos(pitch), pz - r * math.cos(yaw0) * math.cos(pitch)
    a = sum(0.004 if (i % 4) == 3 else 0.0008 for i in range(k))
    yaw = yaw0 + a
    return (cx - r * math.sin(yaw) * math.cos(pitch), py, cz + r * math.cos(yaw) * math.cos(pitch)), yaw, pitch


# ---------------------------------------------------------------------------------------------------------------------------------
# Self-launch: `python bench.py --gpus N` with no launcher around it.  Nothing below imports torch, loads the library or makes a HIP
# call: the parent only starts child processes and reads their output.

FORM_CHAIN = (            # tried in this order, each in fresh children; the first that prints a line is the headline
    ("resident", ["--form", "resident", "--batch", "0"]),
    ("rccl", ["--form", "rccl"]),
    ("onecall", ["--form", "onecall"]),
)
BATCHED_LEG = ("resident_batch4", ["--form", "resident", "--batch", "4"])
LAUNCHER_FORMS = ("resident", "rccl", "resident_batch4")        # forms that need one process per GPU


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def strip_flag(argv, names):
    """argv without the flags in `names` (each takes one value; `--flag v` and `--flag=v`)."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in names:
            skip = True
            continue
        if any(a.startswith(n + "=") for n in names):
            continue
        out.append(a)
    return out


def last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                d = json.loads(line)
                if isinstance(d, dict) and "metric" in d:
                    return d
            except ValueError:
                pass
    return None


def ranks_last_words(err, n=1200):
    """the end of what the RANKS wrote: torch.distributed.run appends its own log lines and a long traceback after a rank fails"""
    import re
    m = re.search(r"^[EW]\d{4} [\d:.]+ +\d+ torch/distributed/", err, flags=re.M)
    return err[:m.start()][-n:] if m and err[:m.start()].strip() else err[-n:]


def descendants(pid):
    """pids of every process below `pid` (children, their children, ...), from /proc: what a job this process started consists of"""
    kids = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open(f"/proc/{d}/stat") as f:
                    st = f.read()
                kids.setdefault(int(st[st.rindex(")") + 2:].split()[1]), []).append(int(d))
            except (OSError, ValueError, IndexError):
                pass
    out, todo = [], [pid]
    while todo:
        for k in kids.get(todo.pop(), []):
            out.append(k); todo.append(k)
    return out


def alive(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            st = f.read()
        return st[st.rindex(")") + 2:].split()[0] != "Z"
    except (OSError, ValueError, IndexError):
        return False


def child_command(form_name, n, port, script, python=None):
    python = python or sys.executable
    if form_name in LAUNCHER_FORMS:
        return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                "--master-port", str(port), str(script)]
    return [python, str(script)]


FORM_TIMEOUT_S = 420


def launch_ranks(n, argv, chain=FORM_CHAIN, extra=BATCHED_LEG, script=None, timeout=None, env=None, log=sys.stderr):
    """Runs the forms of `chain` one after the other, each as FRESH child processes, until one prints the bench line; returns (that line
    as a dict with `forms_tried` added - or None -, the list of attempts).  `argv` = this invocation's own arguments (forwarded; --form /
    --batch are replaced).  After a headline from the tile-resident form, `extra` runs as one more job and its figures are attached as
    `batched`.  A job that outlives `timeout` is ended as a whole: the launcher AND its ranks are one process group of their own.
    The default, FORM_TIMEOUT_S, is a hang detector and not a budget: a form's job takes 7-10 s on a warm one-GPU box
    (profiles/r05/a_forms_self_launch.txt) and a cold box adds a minute or two of first `import torch`; a form that deadlocks on its first
    real node must leave the forms after it the time to produce the line (YCGE_BENCH_FORM_TIMEOUT overrides, seconds)."""
    import signal
    import subprocess
    if timeout is None:
        timeout = float(os.environ.get("YCGE_BENCH_FORM_TIMEOUT", FORM_TIMEOUT_S))
    script = script or Path(__file__).resolve()
    base = strip_flag(list(argv), ("--form", "--batch"))
    env = dict(os.environ if env is None else env)
    env["YCGE_BENCH_CHILD"] = "1"               # a child never launches children of its own
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    tried, headline = [], None

    def run(name, form_args):
        cmd = child_command(name, n, free_port(), script) + base + form_args
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            # end the job this call started - the launcher and every process below it (torch.distributed.run gives its ranks sessions of their
            # own, so the launcher's process group alone would leave them running, holding their GPUs): SIGTERM to the launcher first (it takes
            # its ranks down itself), then SIGKILL to whatever of ITS descendants is left
            below = descendants(proc.pid)
            proc.send_signal(signal.SIGTERM)
            t_end = time.perf_counter() + 15.0
            while time.perf_counter() < t_end and (proc.poll() is None or any(alive(q) for q in below)):
                time.sleep(0.2)
                below = sorted(set(below) | set(descendants(proc.pid)))
            for q in below + [proc.pid]:
                if alive(q):
                    try:
                        os.kill(q, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
            try:
                out, err = proc.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                out, err = "", "bench.py: the ended job's pipes stayed open"
            rc, err = -9, (err or "") + f"\nbench.py: job ended after {timeout} s"
        line = last_json_line(out) if rc == 0 else None
        rec = {"form": name, "rc": rc, "seconds": round(time.perf_counter() - t0, 1), "ok": line is not None}
        if line is None:
            rec["stderr_tail"] = ranks_last_words(err or "")
            print(f"bench.py: form {name} failed (rc {rc}); " + (err or "")[-800:], file=log)
        tried.append(rec)
        return line

    for name, form_args in chain:
        headline = run(name, form_args)
        if headline is not None:
            break
    if headline is None:
        return None, tried
    if extra and tried[-1]["form"] == "resident":
        b = run(*extra)
        headline["batched"] = ({k: b.get(k) for k in ("value", "unit", "ms_per_step", "n_gpus", "rccl_world", "latency_frames")} | {"parallelism": b["config"]["parallelism"]}) if b else {"failed": tried[-1]["rc"]}
    headline["forms_tried"] = tried
    headline["launched_by"] = "bench.py itself: " + " ".join(child_command(tried[0]["form"], n, "<port>", "bench.py", python="python")) + " ..."
    return headline, tried


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--camera", choices=("static", "orbit"), default="static", help="orbit: the pose changes every frame of the timed region (the headline then is the moving-camera frame)")
    ap.add_argument("--form", choices=("auto", "onecall", "rccl", "resident"), default="auto", help="how N > 1 GPUs are driven (see the module docstring)")
    ap.add_argument("--ring", type=int, default=4, help="--form resident: frame sets in the ring = tiled traces in flight (config.tile_ring)")
    ap.add_argument("--batch", type=int, default=0, help="--form resident: frames of a rank's tiles traced in ONE launch (ycge_trace_tiles_resident_batch; 0 or 1 = frame by frame, the default: a batch needs the next n poses); the ring is then at least three batches deep")
    ap.add_argument("--t01", type=float, default=0.25, help="config 5: day phase of the sun and moon (DayNightCycle.cs:48-82); 0.25 = SURVEY 8(d): sun on the horizon, BOTH lights at intensity 0; 0.5 = noon, 0.8 = night")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-post", action="store_true", help="skip the frames WITH the denoise/exposure/tonemap stage (reported apart as post_stage)")
    ap.add_argument("--no-flight", action="store_true", help="skip the frames-in-flight leg (ycge_render_frame_async; reported apart as frames_in_flight)")
    ap.add_argument("--no-moving", action="store_true", help="skip the moving-camera leg (reported apart as moving_camera)")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    force_tiled = bool(os.environ.get("YCGE_BENCH_FORCE_TILED"))        # runs the one-process-per-GPU forms with ONE rank (a one-GPU box)
    if not under_launcher and not os.environ.get("YCGE_BENCH_CHILD") and (args.gpus > 1 or force_tiled) and args.form != "onecall":
        # the plain `python bench.py --gpus N`: start the ranks ourselves (module docstring, `auto`); this process never touches a GPU
        from yetanotherconsolegameengine_amd import build as _build           # (imports no torch; hipcc cross-compiles without a GPU)
        _build.build_library()
        chain = FORM_CHAIN if args.form == "auto" else tuple(f for f in FORM_CHAIN if f[0] == args.form)
        if args.form == "resident" and args.batch > 1:
            chain = (("resident", ["--form", "resident", "--batch", str(args.batch)]),)
        if force_tiled and args.gpus == 1:
            chain = tuple(f for f in chain if f[0] != "onecall")
        line, tried = launch_ranks(args.gpus, sys.argv[1:], chain=chain, extra=BATCHED_LEG if args.form == "auto" else None)
        if line is None:
            raise SystemExit("bench.py --gpus %d: no form produced a line: %s" % (args.gpus, json.dumps(tried)))
        print(json.dumps(line))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    form = args.form
    if form == "auto":
        # under a launcher: the tile-resident form, frame by frame (a batch of n frames a launch needs the next n poses: opt-in, --batch n)
        form = "resident" if (world > 1 or (under_launcher and force_tiled)) else "onecall"
    resident = form == "resident"
    if resident:
        form = "rccl"          # (the same launcher, ranks and tile partition; only the per-frame exchange differs)
        if args.batch > 1:
            args.batch = min(args.batch, 5)
            args.ring = min(15, max(args.ring, 3 * args.batch))
    if form == "rccl" and world != args.gpus and not (world == 1 and os.environ.get("YCGE_BENCH_FORCE_TILED")):
        raise SystemExit(f"--form rccl --gpus {args.gpus} needs one process per GPU (torchrun --nproc-per-node {args.gpus}); WORLD_SIZE={world}")
    if form == "onecall" and world > 1:
        raise SystemExit(f"--form onecall is ONE process driving --gpus {args.gpus} devices; it was started under a launcher with WORLD_SIZE={world}")
    multi = form == "rccl" and (world > 1 or bool(os.environ.get("YCGE_BENCH_FORCE_TILED")))     # the env knob runs the tiled path (slab + all-gather + resolve) with one rank
    n_dev = args.gpus if form == "onecall" else 1          # devices THIS process drives

    import torch          # (first: the library below must bind to the HIP runtime torch has loaded, not a second copy)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-trace path has no CPU fallback")
    from yetanotherconsolegameengine_amd import abi, build, scenes
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py --gpus {args.gpus}: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible) - the ray-trace path has no CPU fallback and will not run on fewer GPUs than asked for")
    torch.cuda.set_device(local_rank)
    rccl_world = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # a CHECKED all-reduce before anything is timed: every rank contributes rank + 1, every rank must see world (world + 1) / 2
        chk = torch.tensor([rank + 1], dtype=torch.int64, device="cuda")
        dist.all_reduce(chk)
        if int(chk.item()) != world * (world + 1) // 2:
            raise SystemExit(f"RCCL all-reduce over {world} rank(s) returned {int(chk.item())}, expected {world * (world + 1) // 2}")
        rccl_world = dist.get_world_size()
    # rank 0 builds (a no-op when the parent of a self-launch or __graft_entry__.build() already did), the others load only behind the barrier
    if rank == 0:
        build.build_library()
    if multi:
        dist.barrier()
    L = abi.load_library()
    visible = L.ycge_device_count()
    if visible < (n_dev if form == "onecall" else local_rank + 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: {visible} HIP device(s) visible - the ray-trace path has no CPU fallback and will not run on fewer GPUs than asked for")

    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    from yetanotherconsolegameengine_amd.scene import flatten

    scene, fbw, fbh, ss, pose = scenes.config_scene(args.config, t01=args.t01)
    lit_tag = "" if (args.config != 5 or args.t01 == 0.25) else "_t%03d" % round(args.t01 * 100)
    flat = flatten(scene)
    hiW, hiH = fbw * ss, fbh * 2 * ss
    pixels = hiW * hiH
    build_hash = build.source_hash()

    def make(count):
        # the multi-GPU frame ends with TAA (the metric's frame): lean slabs, no albedo plane in the all-gather (32 instead of 44 B per pixel)
        r = RaytraceRenderer(flat, fbw, fbh, pose["fov"], ss, count_work=count, device=local_rank, rank=rank, world_size=world if multi else 1,
                             slab_albedo=not multi, devices=list(range(n_dev)) if n_dev > 1 else None, tile_ring=args.ring if (resident and not count) else 0)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        return r

    def set_pose(rr, k, moving):
        if moving:
            pos, yaw, pitch = orbit_pose(pose, k)
            rr.SetCamera(pos, yaw, pitch)

    r = make(False)
    stream = torch.cuda.current_stream()
    slab = all_slabs = None
    # Several processes: the trace of frame N+1 does not depend on the all-gather / resolve (TAA) of frame N - the library keeps the
    # trace's outputs apart from the resolved frame - so the two run on two streams with double-buffered slabs: a frame
    # costs max(trace, gather + resolve) instead of their sum.  Every frame is still traced, gathered and resolved inside
    # the timed region (both streams are drained before the clock stops).  YCGE_BENCH_PIPELINE=0: one stream, in sequence.
    pipelined = multi and os.environ.get("YCGE_BENCH_PIPELINE", "1") != "0"
    if multi:
        nb = r.tile_slab_bytes()
        slabs = [torch.empty(nb // 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.empty(world * (nb // 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        slab, all_slabs = slabs[0], gathered[0]
        # two trace streams, taken in turn: consecutive frames' traces may overlap (the bulk of one in the wavefront slots the tail of the
        # other leaves empty - on a rank's share of a frame that is most of the machine); the library keeps what a trace writes per frame parity
        s_traces, s_comm = [torch.cuda.Stream(), torch.cuda.Stream()], torch.cuda.Stream()
        if os.environ.get("YCGE_BENCH_ONE_TRACE_STREAM"):
            s_traces[1] = s_traces[0]
        ev_traced = [torch.cuda.Event() for _ in range(2)]
        ev_resolved = [torch.cuda.Event() for _ in range(2)]
        n_issued = [0]
    if multi and resident:
        # the tile-resident form: K frame slots (send / receive halo records, history slab, gathered history), K trace streams taken in turn
        K = args.ring
        s_cnt, r_cnt = r.halo_counts()
        hb = r.history_slab_bytes() // 4
        h_send = [torch.zeros(max(1, sum(s_cnt)) * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
        h_recv = [torch.zeros(max(1, sum(r_cnt)) * 4, dtype=torch.float32, device="cuda") for _ in range(K)]
        h_hist = [torch.zeros(hb, dtype=torch.float32, device="cuda") for _ in range(K)]
        h_all = [torch.zeros(world * hb, dtype=torch.float32, device="cuda") for _ in range(K)]
        rs_traces = [torch.cuda.Stream() for _ in range(K)]
        rs_ev_t = [torch.cuda.Event() for _ in range(K)]; rs_ev_r = [torch.cuda.Event() for _ in range(K)]
        rs_issued = []
        s_traces = rs_traces[:2] if K >= 2 else [rs_traces[0], rs_traces[0]]
        B = args.batch if args.batch > 1 else 0
        rs_poses, rs_batches = [], [0]
        # the launches of the TIMED region, bracketed by HIP events on the stream each one is launched on (the roofline's launch duration)
        rs_timing, rs_launches = [False], []

        def rs_bracket(st, n_frames):
            if not rs_timing[0]:
                return None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rs_launches.append((e0, e1, n_frames))
            return e1

        # everything a frame's exchange needs, made ONCE: the host's cost per frame is part of a rank's period (0.12 ms a frame with views, split
        # lists and stream contexts rebuilt per call - more than the 0.09 ms the GPU needs in the batched form; `host_issue_ms_per_frame`)
        rs_recv_v = [t[:sum(r_cnt) * 4] for t in h_recv]; rs_send_v = [t[:sum(s_cnt) * 4] for t in h_send]
        rs_out_split = [c * 4 for c in r_cnt]; rs_in_split = [c * 4 for c in s_cnt]
        rs_recv_p = [t.data_ptr() for t in h_recv]; rs_send_p = [t.data_ptr() for t in h_send]
        rs_hist_p = [t.data_ptr() for t in h_hist]; rs_all_p = [t.data_ptr() for t in h_all]
        rs_trace_sp = [st.cuda_stream for st in rs_traces]; s_comm_p = s_comm.cuda_stream
        torch.cuda.set_stream(s_comm)          # the collectives below run on torch's CURRENT stream: the exchange stream for the life of the loop (the library's calls take their stream as an argument)

        def rs_resolve(k):
            s_comm.wait_event(rs_ev_t[k])
            dist.all_to_all_single(rs_recv_v[k], rs_send_v[k], output_split_sizes=rs_out_split, input_split_sizes=rs_in_split)
            r.resolve_tiles_resident(rs_recv_p[k], rs_hist_p[k], s_comm_p)
            dist.all_gather_into_tensor(h_all[k], h_hist[k])          # whoever shows the frame: here every rank (a gather to rank 0 moves an eighth of it)
            r.unpack_history(rs_all_p[k], s_comm_p)
            rs_ev_r[k].record(s_comm)

    def rs_issue_batch(rr):
        """the frames whose poses wait in rs_poses, in ONE launch; consecutive batches on two streams; resolves issued as soon as a batch is two batches old"""
        n = len(rs_poses)
        while len(rs_issued) + n > K:
            rs_resolve(rs_issued.pop(0))
        slots = [(n_issued[0] + j) % K for j in range(n)]
        n_issued[0] += n
        st = rs_traces[(rs_batches[0] & 1) * (2 if K >= 3 else 1)]          # (neighbouring streams tend to share a hardware queue: 4 queues, round robin)
        rs_batches[0] += 1
        for k in slots:
            st.wait_event(rs_ev_r[k])
        e1 = rs_bracket(st, n)
        rr.trace_tiles_resident_batch(list(rs_poses), [rs_send_p[k] for k in slots], st.cuda_stream)
        if e1 is not None:
            e1.record(st)
        for k in slots:
            rs_ev_t[k].record(st)
        rs_issued.extend(slots)
        rs_poses.clear()
        # exchange + resolve of the batch's frames are queued NOW, behind its launch (they wait for its event on the exchange stream): the
        # launch that takes these ring slots three batches later finds them done (queued only when the slots were needed, they ran starved
        # beside the launches in flight and the next launch waited for them: profiles/r05, the gaps between k_trace_batch launches)
        while rs_issued:
            rs_resolve(rs_issued.pop(0))

    def step(rr, want_stats=False):
        """One frame; returns (trace_ms, frame_ms) as the library measured them (0 where the pipelined form takes no per-step timing)."""
        if not multi:
            rr.TryFlipAndBlit()
            return float(rr.stats.trace_ms), float(rr.stats.total_ms)
        if resident and rr is r and not want_stats and B:
            rs_poses.append((tuple(rr._pos), rr._yaw, rr._pitch, rr._fov))
            if len(rs_poses) == B:
                rs_issue_batch(rr)
            return 0.0, 0.0
        if resident and rr is r and not want_stats:
            k = n_issued[0] % K
            n_issued[0] += 1
            st = rs_traces[k]
            st.wait_event(rs_ev_r[k])          # slot k's buffers were last read by the exchange of K frames ago
            e1 = rs_bracket(st, 1)
            rr.trace_tiles_resident(rs_send_p[k], rs_trace_sp[k])
            if e1 is not None:
                e1.record(st)
            rs_ev_t[k].record(st)
            rs_resolve(k)          # queued right behind its trace (waits for rs_ev_t[k] on the exchange stream): done long before slot k is taken again
            return 0.0, 0.0
        if not pipelined or want_stats:
            rr.trace_tiles(slab.data_ptr(), stream.cuda_stream, want_stats=want_stats)
            t = float(rr.stats.trace_ms) if want_stats else 0.0
            dist.all_gather_into_tensor(all_slabs, slab)
            rr.resolve_gathered(all_slabs.data_ptr(), stream.cuda_stream)
            return t, 0.0
        k = n_issued[0] & 1
        n_issued[0] += 1
        s_trace = s_traces[k]
        with torch.cuda.stream(s_trace):
            s_trace.wait_event(ev_resolved[k])           # slab k was last read by the gather of two frames ago
            rr.trace_tiles(slabs[k].data_ptr(), s_trace.cuda_stream)
            ev_traced[k].record(s_trace)
        with torch.cuda.stream(s_comm):
            s_comm.wait_event(ev_traced[k])
            dist.all_gather_into_tensor(gathered[k], slabs[k])
            rr.resolve_gathered(gathered[k].data_ptr(), s_comm.cuda_stream)
            ev_resolved[k].record(s_comm)
        return 0.0, 0.0

    def fence():
        if multi and resident:
            if B and rs_poses:
                rs_issue_batch(r)
            while rs_issued:
                rs_resolve(rs_issued.pop(0))
            for s_ in rs_traces:
                s_.synchronize()
        if multi:
            s_traces[0].synchronize(); s_traces[1].synchronize(); s_comm.synchronize()
            dist.barrier()
        if n_dev > 1:           # one process, several devices: drain them all (a rank of the RCCL form only ever touches its own device)
            for d in range(n_dev):
                torch.cuda.synchronize(d)
        torch.cuda.synchronize()

    moving = args.camera == "orbit"
    for k in range(args.warmup):
        set_pose(r, k, moving)
        step(r)
    fence()
    first_frame = args.warmup + 1
    steps0 = r.timed_steps() if not multi else 0
    per_step = []
    if multi and resident:
        rs_timing[0] = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        # one process: ycge_render_frame times the trace with HIP events on its own stream as part of the (synchronous) call.  Several
        # processes: no per-step host synchronisation inside the timed region (trace on one stream, all-gather -> resolve on another)
        set_pose(r, args.warmup + k, moving)
        per_step.append(step(r, want_stats=not multi))
    issue_s = time.perf_counter() - t0          # host time to QUEUE the timed frames (several processes: nothing waits inside the loop)
    fence()
    elapsed = time.perf_counter() - t0
    timed_lane_steps = (r.timed_steps() - steps0) / args.steps if not multi else None
    device_tiles = [int(r.stats.device_tiles[i]) for i in range(int(r.stats.n_devices_traced))] if not multi else None
    trace_ms = [p[0] for p in per_step]
    frame_ms = [p[1] for p in per_step]
    launch_frames = 1          # frames of this rank's tiles one launch of the dominant kernel traces
    if multi:
        if resident:    # the launches of the timed region themselves (HIP events on their own streams; traces in flight overlap, so a launch's
            # duration includes what it shares the machine with)
            rs_timing[0] = False
            launch_frames = B or 1
            trace_ms = [e0.elapsed_time(e1) for e0, e1, _ in rs_launches]
        else:           # slab form: a few extra, untimed frames with the library's event timing
            trace_ms = [step(r, want_stats=True)[0] for _ in range(4)][1:]
            fence()
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        nt = torch.tensor([r.stats.device_tiles[0] if r.stats.n_devices_traced else 0], dtype=torch.int64, device="cuda")
        allt = [torch.zeros_like(nt) for _ in range(world)]
        dist.all_gather(allt, nt)
        device_tiles = [int(t.item()) for t in allt]
    n_gpus_used = sum(1 for t in device_tiles if t > 0)

    # ---- exact work of the timed frames: counting replay (untimed; same frame numbers, same poses)
    rc = make(True)
    rc.set_frame_counter(first_frame - 1)
    tot = {k: 0 for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox", "n_rays_dark")}
    replay = args.steps
    for k in range(replay):
        set_pose(rc, args.warmup + k, moving)
        if multi:
            rc.trace_tiles(slab.data_ptr(), stream.cuda_stream, want_stats=True)
        else:
            rc.TryFlipAndBlit()
        for kk, v in stats_dict(rc.stats).items():
            tot[kk] += v
    if multi:
        tt = torch.tensor([tot[k] for k in sorted(tot)], dtype=torch.int64, device="cuda")
        dist.all_reduce(tt)
        tot = dict(zip(sorted(tot), [int(x) for x in tt.tolist()]))
    rc.close()

    per_frame = {k: v / replay for k, v in tot.items()}
    traced_per_frame = per_frame["n_rays"] - per_frame["n_rays_dark"]     # what the timed kernels trace; n_rays is the reference's call count
    mrays = traced_per_frame * args.steps / elapsed / 1e6
    mrays_ref_count = per_frame["n_rays"] * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    mean_trace_ms = float(np.mean(trace_ms)) if trace_ms and trace_ms[0] > 0 else None
    # roofline of the dominant kernel (k_trace) on THIS process's share of the frame
    share = world if multi else 1
    my_alg = algorithmic_bytes({k: v / share for k, v in per_frame.items()}, pixels / share) * launch_frames
    roof = None
    if mean_trace_ms:
        ach = my_alg / (mean_trace_ms * 1e-3) / 1e9
        # which launches trace_ms brackets: the single-launch kernel k_trace with k_trace_fan beside it for the head of the schedule
        # (two streams forked from and joined to the frame's stream: the duration is their makespan; scenes without heavy blocks
        # run k_trace alone), or the stage pipeline of scenes with a real top-level tree (config 5)
        fanned = int(r.stats.fan_blocks) > 0
        kernel = "k_wf_* stages" if args.config == 5 else "k_trace + k_trace_fan (concurrent)" if fanned else "k_trace"
        if multi and resident:
            kernel = (f"k_trace_batch ({launch_frames} frames of the rank's tiles per launch)" if launch_frames > 1 else "k_trace on the rank's tiles") + f", up to {args.ring} traces in flight (durations overlap)"
        roof = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "achieved_is": "algorithmic bytes of the REFERENCE's traversal (SURVEY 8d counters) / launch time - mostly cache hits, NOT memory traffic",
                "algorithmic_bytes_per_launch": int(my_alg), "mean_launch_ms": round(mean_trace_ms, 4)}
        if timed_lane_steps is not None and args.config != 5:
            tb = STEP_FETCH_BYTES * timed_lane_steps + 118 * pixels
            roof["timed_work"] = {"lane_steps_per_launch": round(timed_lane_steps, 1), "bytes_per_launch": int(tb),
                                  "achieved_gbs": round(tb / (mean_trace_ms * 1e-3) / 1e9, 2), "frac": round(tb / (mean_trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                  "what": f"what the TIMED kernels walked: {STEP_FETCH_BYTES} B fetched per lane step (node / triangle-pair record; a cooperative step - a treelet or a whole leaf - counts as one) + 118 B per pixel"}
        elif timed_lane_steps is not None:
            # A voxel world: the figure above prices the REFERENCE's walk - every grid entered, every shadow ray traced - which the timed stage
            # kernels do not make (solid-voxel cull, walk tree, no ray towards a dark light), so its `frac` can exceed 1 and says nothing about
            # them.  The roofline of the timed kernels comes from their counting TWIN (lib/var_voxstat.so: the same non-counting instances with
            # per-lane counters), replaying the timed frames: scene-tree steps x 64 B (one node / object record each), cell fetches x 1 B, the
            # stage-queue records written and read back (light record 64 B, continuation ray 48 + 16 B, primary hit 16 B), 118 B per pixel.
            tw = timed_twin_work(flat, fbw, fbh, ss, pose, first_frame, min(args.steps, 6), local_rank, moving, args.warmup, set_pose)
            roof["reference_walk"] = {"algorithmic_bytes_per_launch": int(my_alg), "achieved": roof["achieved"], "frac": roof["frac"],
                                      "what": "SURVEY 8d counters of the REFERENCE's traversal / the timed kernels' launch time: NOT the work they do (frac may exceed 1)"}
            if tw:
                tb = tw["bytes_per_launch"]
                roof["achieved"] = round(tb / (mean_trace_ms * 1e-3) / 1e9, 2)
                roof["frac"] = round(tb / (mean_trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                roof["algorithmic_bytes_per_launch"] = int(tb)
                roof["achieved_is"] = "bytes the TIMED stage kernels fetch and store by their own counters (counting twin of the timed instances, same frames) / launch time"
                roof["timed_work"] = {"lane_steps_per_launch": round(timed_lane_steps, 1), **tw}
            else:
                roof["timed_work"] = {"lane_steps_per_launch": round(timed_lane_steps, 1), "bytes_per_launch": None,
                                      "what": "lib/var_voxstat.so (the counting twin, built by __graft_entry__.build()) is missing: no byte figure for the timed kernels; `frac` above prices the reference's walk"}
        pmc, stale = load_pmc(args.config, build_hash, lit_tag) if (world == 1 and n_dev == 1) else (None, False)
        if pmc:
            # counters of the same kernels, same build, from the committed PMC passes; the rate uses THIS run's launch time
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
        elif stale:
            roof["pmc_stale"] = True       # a committed counter summary exists but belongs to another build of the kernels: not printed
            roof["bound"] = "latency"
        roof["build"] = build_hash

    single = world == 1 and n_dev == 1
    # The same frames as a host drives them that asks for NO statistics (bindings/csharp/HipRaytraceWrapper.cs passes null): the library then
    # records no timing events - the event between the trace and TAA is a packet of its own that the TAA launch waits behind.  The headline
    # `value` stays the call WITH statistics: its trace duration is what the roofline is priced on.
    no_stats = None
    if single and not moving:
        n = min(args.steps, 200)
        ts = []
        for k in range(n):
            t1 = time.perf_counter()
            rc = r.L.ycge_render_frame(r.ctx, None, None)
            ts.append((time.perf_counter() - t1) * 1e3)
            if rc != 0:
                raise abi.YcgeError(rc, "ycge_render_frame without statistics")
        no_stats = {"frames": n, "frame_ms": dist3(ts), "what": "ycge_render_frame(ctx, NULL, NULL): the frame through TAA with no statistics asked for, synchronous"}
    mov = None
    if not args.no_moving and not moving and not multi:      # the frame the host really drives: the pose changes every frame (RaytraceEntity.cs:221-232)
        n = 64
        rows = []
        resets = 0
        for k in range(n):
            set_pose(r, k, True)
            r.TryFlipAndBlit()
            rows.append((float(r.stats.trace_ms), float(r.stats.taa_ms), float(r.stats.total_ms)))
            resets += int(r.stats.history_reset)
        a = np.array(rows[8:])
        mov = {"frames": n - 8, "history_resets": resets, "trace_ms": dist3(a[:, 0]), "taa_ms": dist3(a[:, 1]), "frame_ms": dist3(a[:, 2]),
               "what": "pose changes every frame: 3 small steps (below the TAA reset thresholds) then 1 large (above), orbiting the look-at point; first 8 frames dropped"}
        if not args.no_post:
            rows = []
            for k in range(n, n + 24):
                set_pose(r, k, True)
                r.TryFlipAndBlit(want_sdr=True, copy=False)
                rows.append((float(r.stats.trace_ms), float(r.stats.post_ms), float(r.stats.total_ms)))
            a = np.array(rows[4:])
            mov["with_sdr"] = {"frames": len(a), "trace_ms": dist3(a[:, 0]), "post_ms": dist3(a[:, 1]), "frame_ms_with_sdr_readback": dist3(a[:, 2])}
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])

    flight = None
    if single and not moving and not args.no_flight:        # frames in flight (ycge_render_frame_async; no counterpart in the reference): the same frames, queued without waiting
        n = min(args.steps, 300)
        for _ in range(args.warmup):
            r.RenderAsync()
        r.async_trace_ms()
        tf0 = time.perf_counter()
        for _ in range(n):
            r.RenderAsync()
        r.Wait()
        tf = time.perf_counter() - tf0
        ft = r.async_trace_ms()
        fsdr = None
        if not args.no_post:      # ... and with the post stage + read-back (ycge_render_frame_async_sdr), three page-locked SDR arrays taken in turn
            m = 60
            for i in range(6):
                r.RenderAsync(sdr_slot=i % 3)
            r.Wait()
            ts0 = time.perf_counter()
            for i in range(m):
                r.RenderAsync(sdr_slot=i % 3)
            r.Wait()
            fsdr = {"frames": m, "ms_per_step": round((time.perf_counter() - ts0) / m * 1e3, 4),
                    "what": "ycge_render_frame_async_sdr: the post stage and read-back of frame N beside the traces and TAA of the frames after it; compare post_stage.frame_ms_with_sdr_readback"}
        fi = r.flight_info()
        flight = {"frames": n, "ms_per_step": round(tf / n * 1e3, 4),
                  "gate": "on" if fi["placed_gate"] else "off", "two_trace_streams": bool(fi["two_trace_streams"]), "placed_waits": fi["placed_waits"], "value": round(traced_per_frame * n / tf / 1e6, 2), "unit": "Mrays/s",
                  "trace_ms": dist3([float(x) for x in ft]) if len(ft) else None, "with_sdr": fsdr,
                  "what": "the same frames queued with ycge_render_frame_async: no host wait between frames, two traces at a time on two streams, TAA of frame N and the "
                          "schedule of frame N + 3 on a third stream, three sets of trace outputs taken in turn; bit-identical frames (tests/test_gpu_timed_variants.py). "
                          "The headline `value` stays the synchronous call, which is the reference's TryFlipAndBlit; trace_ms here is the launch with TAA running beside it"}

    post = None
    if not args.no_post and not multi:        # the frame the C# wrapper asks for (SDR out); outside the headline metric, which SURVEY 8d times through TAA
        ms = []
        for _ in range(12):
            r.TryFlipAndBlit(want_sdr=True, copy=False)
            ms.append((float(r.stats.trace_ms), float(r.stats.taa_ms), float(r.stats.post_ms), float(r.stats.total_ms)))
        ms = np.array(ms[2:])            # the first frame builds the in-place level schedule
        post = {"trace_ms": round(float(np.median(ms[:, 0])), 4), "taa_ms": round(float(np.median(ms[:, 1])), 4),
                "post_ms": round(float(np.median(ms[:, 2])), 4), "frame_ms_with_sdr_readback": round(float(np.median(ms[:, 3])), 4),
                "frame_ms_min": round(float(ms[:, 3].min()), 4),
                "what": "ycge_render_frame with an SDR buffer: + A-trous denoise, auto-exposure, tonemap/downsample, read-back (medians of 10 frames, static camera)"}
        # ... and the same frame with the in-place quirk of the denoiser's second iteration WAIVED (config.atrous_inplace_exact = 0, SURVEY 8-f1
        # "reproduce or explicitly waive"; INTEGRATION.md section 2 states what a host gives up): the delivered frame, both ways, side by side
        cfgw = abi.default_config(); cfgw.atrous_inplace_exact = 0
        rw = RaytraceRenderer(flat, fbw, fbh, pose["fov"], ss, cfg=cfgw, device=local_rank)
        rw.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        msw = []
        for _ in range(12):
            rw.TryFlipAndBlit(want_sdr=True, copy=False)
            msw.append((float(rw.stats.post_ms), float(rw.stats.total_ms)))
        rw.close()
        msw = np.array(msw[2:])
        post["waived"] = {"post_ms": round(float(np.median(msw[:, 0])), 4), "frame_ms_with_sdr_readback": round(float(np.median(msw[:, 1])), 4)}

    cpu = None
    if rank == 0 and single and not args.no_cpu_baseline:
        import oracle_binding as ob
        threads = os.cpu_count() or 1
        o = ob.OracleRenderer(scene, fbw, fbh, ss, pose, flat=flat)
        o.set_frame_counter(first_frame - 1)
        rays = 0; t_trace = 0.0; t_taa = 0.0; frames = 0
        per = []
        while t_trace + t_taa < args.cpu_seconds and frames < args.steps:
            if moving:
                pos, yaw, pitch = orbit_pose(pose, args.warmup + frames)
                o.set_camera(pos, yaw, pitch)
            o.render(stages=1, threads=threads)
            rays += int(o.stats.n_rays); t_trace += o.stats.trace_ms * 1e-3; t_taa += o.stats.taa_ms * 1e-3; frames += 1
            per.append(int(o.stats.n_rays) / (o.stats.trace_ms * 1e-3) / 1e6)
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
               "per_frame_mrays": {"min": round(min(per), 2), "median": round(float(np.median(per)), 2), "max": round(max(per), 2), "frames": frames},
               "trace_ms_per_frame": round(t_trace / frames * 1e3, 2),
               "taa_serial_ms": round(t_taa / frames * 1e3, 2),
               "taa_parallel_ms": round(t_taa_par / max(1, n_par) * 1e3, 2),
               "whole_frame_serial_taa": {"value": round(rays / (t_trace + t_taa) / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round((t_trace + t_taa) / frames * 1e3, 2)}}

    if rank == 0:
        name, cus = r.device_info()
        how = ("one process, one ycge_render_frame call per frame drives all devices; peers push their tiles into device 0 over xGMI" if (form == "onecall" and n_dev > 1)
               else f"one process per GPU, tile-resident TAA; per frame one RCCL all-to-all of the one-pixel halo records and one all-gather of the resolved history (12 B per pixel); a ring of {args.ring} frame sets" + (f", {args.batch} frames of the rank's tiles per launch (ycge_trace_tiles_resident_batch)" if args.batch > 1 else ", frame by frame") if (multi and resident)
               else "one process per GPU; one RCCL all-gather of the tile slabs per frame" + ("; traces of consecutive frames on two streams (they may overlap), gather + resolve of frame N on a third" if pipelined else "") if multi else "single GPU")
        out = {
            "metric": f"Mrays/s (all rays: Scene.Hit + Scene.Occluded calls) and ms/frame, {METRIC_SHAPES[args.config]}",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_gpus_used, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config {args.config}: " + WORKLOADS[args.config],
                       "trace_grid": f"{hiW}x{hiH}", "spp": ss * ss, "triangles": flat.n_triangles, "camera": args.camera, **({"t01": args.t01} if args.config == 5 else {}),
                       "lights": {"intensity": [float(l.Intensity) for l in scene.Lights],
                                  "note": "the timed kernels trace no shadow ray towards a light of intensity 0 (its contribution is a zero whatever the ray finds: bit-identical pixels); "
                                          "`value` counts traced rays only, `value_reference_ray_count` the reference's Scene.Hit / Scene.Occluded calls, which include those rays" if any(float(l.Intensity) == 0.0 for l in scene.Lights) else None},
                       "frame": "ray-gen + trace + TAA" + (" on the rank's own tiles + RCCL all-to-all of halo records + all-gather of the history" if (multi and resident) else " + RCCL all-gather of tile slabs + un-permute" if multi else " + peer tile push" if n_dev > 1 else ""),
                       "parallelism": f"framebuffer tiles 32x8 round-robin over {n_gpus_used} GPU(s): " + how, "form": "resident" if (multi and resident) else form,
                       "gpus_requested": args.gpus, "device_tiles": device_tiles, "device": name, "compute_units": cus},
            # frames between a pose and its image: 1 = the synchronous call (TryFlipAndBlit); the one-process-per-GPU forms keep frames in
            # flight (the ring of the tile-resident form: K traces, or three batches of n frames - a batch also NEEDS its n poses up front)
            "latency_frames": (args.ring if resident else 2 if pipelined else 1) if multi else 1,
            "rccl_world": rccl_world,
            # one process per GPU: what this rank's host thread spent queueing a frame (Python + torch.distributed + the library's calls); a rank's
            # period cannot be shorter, whatever its GPU does - compare DESIGN section 8's emulated periods, whose loop is driven from C
            "host_issue_ms_per_frame": round(issue_s / args.steps * 1e3, 4) if multi else None,
            "primary_mrays_per_s": round(pixels * args.steps / elapsed / 1e6, 2),
            "rays_per_frame": round(per_frame["n_rays"], 1),
            # the reference's call count includes shadow rays towards lights of intensity 0, which the timed kernels never trace (bit-identical
            # pixels): `value` is over traced rays, the rate with the reference's count is beside it
            "rays_traced_per_frame": round(traced_per_frame, 1),
            "rays_to_dark_lights_per_frame": round(per_frame["n_rays_dark"], 1),
            "value_reference_ray_count": round(mrays_ref_count, 2),
            "work_per_frame": {k: round(v, 1) for k, v in per_frame.items()},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if post:        # the frame TryFlipAndBlit delivers (SDR chexels in host memory), beside the headline's trace + TAA
            out["sdr_frame_ms"] = {"exact": post["frame_ms_with_sdr_readback"], "waived": post["waived"]["frame_ms_with_sdr_readback"],
                                   "what": "ycge_render_frame with an SDR buffer, synchronous, median: exact = the reference's in-place A-trous iteration bit for bit; waived = config.atrous_inplace_exact 0"}
        if flight:
            out["frames_in_flight"] = flight
        if frame_ms and frame_ms[0] > 0:
            out["frame_ms"] = dist3(frame_ms)
        if trace_ms and trace_ms[0] > 0:
            out["trace_ms"] = dist3(trace_ms)
        if n_gpus_used != args.gpus:
            out["warning"] = f"--gpus {args.gpus} asked for, {n_gpus_used} device(s) traced tiles"
        if cpu:       # like for like: whole frames (trace + TAA) on both sides, as a ratio of FRAME TIMES (the CPU side traces every ray of the
            # reference, dark lights included, so a ratio of ray rates would credit the GPU with rays it never traces); 3 significant digits:
            # the CPU side varies by a factor of two from frame to frame on a shared host (cpu_baseline.per_frame_mrays)
            ratio = cpu["whole_frame_serial_taa"]["ms_per_frame"] / ms_per_step
            out["gpu_over_cpu"] = float(f"{ratio:.3g}")
            out["gpu_over_cpu_is"] = "cpu whole-frame ms (trace on all host threads + the reference's serial TAA) / gpu ms_per_step"
        if no_stats:
            out["without_statistics"] = no_stats
        if mov:
            out["moving_camera"] = mov
        if post:
            out["post_stage"] = post
        print(json.dumps(out))
    r.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
