"""Soak of tests/test_gpu_random_scenes.py: seeds lo .. hi - 1 of its two scene generators (plain, and pushed one way: glass-heavy, many lights,
thousands of objects, camera inside an object, degenerate objects, scaled by 1e-2 .. 1e3, textured, voxel chunks, a material per triangle) on both device paths, two frames each, every buffer
and counter against the oracle bit for bit; then its drawn CALL SEQUENCES (16 steps each; synchronous, then with frames in flight) its drawn renderer constants, its mesh viewers and its draws on 2 - 8 emulated ranks (both tiled forms).  Prints the frames that differ and a total; exit status 1 if there is one.

    gpurun -- python profiles/fuzz_scenes.py 100 500        (round 6: 194 704 frames over seeds 0 .. 2759, none differs - profiles/r06/g_fuzz_scenes.txt)
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT))
import torch                         # noqa: E402  (first, as tests/conftest.py and bench.py do: torch and the library then share ONE libamdhip64)
torch.zeros(1, device="cuda")
import oracle_binding as ob          # noqa: E402  (the checker; this script is a test driver, not a product path)
import parity_util as pu             # noqa: E402
import test_gpu_random_scenes as T   # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
SIZES = [(160, 45, 1), (97, 31, 1), (64, 20, 2), (200, 60, 1)]
if len(sys.argv) > 3 and sys.argv[3] == "poison":          # values no scene should hold: what equals the oracle's and what merely survives (tests/test_gpu_random_scenes.py)
    for values, what in (("all", "lights"), ("all", "materials"), ("tame", "geometry"), (1e9, "geometry"), (1e17, "geometry"), (1e19, "geometry"), (1e30, "geometry"),
                         (3.4028234663852886e38, "geometry"), (float("inf"), "geometry"), (float("nan"), "geometry")):
        n_frames = n_bad = 0
        for path in ("wavefront", "megakernel"):
            os.environ["YCGE_PATH"] = path
            for seed in range(lo, hi):
                found, n = T.run_poisoned(ob, seed, values, what, log=lambda *a: None)
                n_frames += n; n_bad += len(found)
        print(f"poisoned {what}, values {values}: {n_bad} of {n_frames} frames differ from the oracle's somewhere (every one was rendered)", flush=True)
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "large":          # the drawn scenes (plain and pushed) at BASELINE's frame sizes, two frames each, then stop: a few seeds take minutes of oracle time
    import time
    n_bad = 0
    for pushed in (False, True):
        for path in ("wavefront", "megakernel"):
            os.environ["YCGE_PATH"] = path
            for seed in range(lo, hi):
                s, pose = T.random_scene(seed)
                tag = T.harden(s, pose, seed) if pushed else "plain"
                w, h, ss = [(1920, 540, 1), (960, 270, 2)][seed % 2]
                t0 = time.time()
                o, g = pu.run_pair(ob, s, w, h, ss, pose, frames=1)
                for f in range(2):
                    if f:
                        o.render(stages=1, threads=32); g.TryFlipAndBlit()
                    st = pu.compare_frame(o, g)
                    bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
                    bad.update({k: st[k] for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if st[k][0] != st[k][1]})
                    n_bad += bool(bad)
                    print(path, "seed", seed, tag, f"{w}x{h} ss {ss}", "frame", f, len(s.Objects), "objects", int(g.stats.n_rays), "rays", f"trace {float(g.stats.trace_ms):.3f} ms",
                          "DIFFERS " + repr(bad) if bad else "equal", f"({time.time() - t0:.0f} s)", flush=True)
                o.close(); g.close()
    sys.exit(1 if n_bad else 0)
n_frames = n_bad = 0
for pushed in (False, True):
    for path in ("wavefront", "megakernel"):
        os.environ["YCGE_PATH"] = path
        for seed in range(lo, hi):
            s, pose = T.random_scene(seed)
            tag = T.harden(s, pose, seed) if pushed else "plain"
            try:
                o, g = pu.run_pair(ob, s, *SIZES[seed % 4], pose, frames=1)
            except Exception as e:      # a scene the library refuses the oracle must refuse too: print, go on
                print("REFUSED", path, seed, tag, repr(e)[:200], flush=True)
                continue
            for f in range(2):
                if f:
                    o.render(stages=1, threads=8); g.TryFlipAndBlit()
                st = pu.compare_frame(o, g)
                bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
                cnt = {k: st[k] for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if st[k][0] != st[k][1]}
                n_frames += 1
                if bad or cnt:
                    n_bad += 1
                    print("MISMATCH", path, seed, tag, "frame", f, len(s.Objects), "objects", bad, cnt, flush=True)
            o.close(); g.close()
        print(f"{'pushed' if pushed else 'plain'} scenes, {path}: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
# call sequences (camera moves around the TAA thresholds, lights, moved objects, new scenes, resizes, frame-counter jumps, SDR frames in between)
for flight in (False, True):          # (True: through ycge_render_frame_async / _async_sdr, one to three frames queued before ycge_wait)
    for path in ("wavefront", "megakernel"):
        os.environ["YCGE_PATH"] = path
        for seed in range(lo, hi):
            found = T.run_sequence(ob, seed, steps=16, log=lambda *a: None, flight=flight)
            n_frames += 16; n_bad += len(found)
            for label, bad in found:
                print("MISMATCH", path, "in flight" if flight else "", label, bad, flush=True)
        print(f"call sequences{' with frames in flight' if flight else ''}, {path}: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
for path in ("wavefront", "megakernel"):          # drawn renderer constants (ycge_config), five frames each, every other one through the post stage
    os.environ["YCGE_PATH"] = path
    for seed in range(lo, hi):
        found = T.run_drawn_config(ob, seed, log=lambda *a: None)
        n_frames += 5; n_bad += len(found)
        for label, bad in found:
            print("MISMATCH", path, label, bad, flush=True)
    print(f"drawn renderer constants, {path}: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
for path in ("wavefront", "megakernel"):          # mesh viewers (the flat single-launch kernels' scenes), counting and timed kernel instances
    os.environ["YCGE_PATH"] = path
    for seed in range(lo, hi):
        found = T.run_mesh_viewer(ob, seed, log=lambda *a: None)
        n_frames += 6; n_bad += len(found)
        for label, bad in found:
            print("MISMATCH", path, label, bad, flush=True)
    print(f"mesh viewers, {path}: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
os.environ.pop("YCGE_PATH", None)
for form, kw in (("three contexts on the one GPU (peer push)", dict(devices=[0, 0, 0])), ("the all-gather form, a world of one", dict(rccl=True))):          # the one-process multi-GPU context of the C# host
    for seed in range(lo, hi):
        found = T.run_sequence(ob, seed, steps=16, log=lambda *a: None, **kw)
        n_frames += 16; n_bad += len(found)
        for label, bad in found:
            print("MISMATCH", form, label, bad, flush=True)
    print(f"call sequences, {form}: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
for seed in range(lo, hi):          # the tiled forms on 2 - 8 emulated ranks against one context
    found = T.run_tile_split(seed, log=lambda *a: None)
    n_frames += 6; n_bad += len(found)
    for label, bad in found:
        print("MISMATCH", label, bad, flush=True)
print(f"tiled forms on emulated ranks: seeds {lo}..{hi - 1} done; {n_bad} of {n_frames} frames differ so far", flush=True)
sys.exit(1 if n_bad else 0)
