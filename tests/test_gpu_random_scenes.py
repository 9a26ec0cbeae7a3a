"""-m gpu: seeded RANDOM scenes, the HIP path against the oracle through the C-ABI.

The configurations and the hand-made scenes of test_gpu_parity.py each exercise what they were written for.  These scenes are drawn:
30 - 260 objects of every Hittable class (BoundedObjects.cs, Surfaces.cs, Triangle.cs, Mesh.cs, VolumeGrid.cs) with materials of every
branch of TraceFull (RaytraceRenderer.cs:448-620: diffuse, specular, mirror >= 0.9, partial reflectivity, glass of several indices, emissive,
checker), 0 - 4 lights (zero intensity included), a real top-level tree of some depth (BVH.cs:258-459) with leaves of mixed kinds - and the
accidents a drawn scene has that a designed one avoids: objects inside one another, coincident surfaces (exact ties in t: a duplicated sphere,
boxes sharing a face), slivers, objects behind the camera and around it.  Bar: every buffer bit for bit, counters equal, on both device paths,
over a reset frame and two blended ones.
"""
import copy
import ctypes as C
import os
import dataclasses

import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import (AmbientLight, Box, Checker, CylinderY, Disk, Material, Mesh, Plane, PointLight, Scene, Solid,
                                                   Sphere, Texture, Triangle, VolumeGrid, XYRect, XZRect, YZRect, flatten, vec3, ZERO)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["wavefront", "megakernel"])
def path(request, monkeypatch):
    monkeypatch.setenv("YCGE_PATH", request.param)
    return request.param


from random_scenes import HARD_MODES, _f, harden, poison, random_scene      # noqa: E402,F401  (the generators: tests/random_scenes.py)


def _frames_equal(oracle, s, pose, size, label, frames=3):
    o, g = pu.run_pair(oracle, s, *size, pose, frames=1)
    for f in range(frames):
        if f:
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
        st = pu.compare_frame(o, g)
        print(f"{label} frame {f}: {len(s.Objects)} objects, {len(s.Lights)} lights", st)
        for k in ("rays", "prim_id", "sub_id", "hit_t", "rng_state", "sky", "g_depth", "current_hdr", "taa_history", "g_albedo", "g_normal"):
            assert st[k + "_mismatch"] == 0, f"{label} frame {f}: {k} differs in {st[k + '_mismatch']} elements"
        for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
            assert st[k][0] == st[k][1], f"{label} frame {f}: counter {k} oracle {st[k][0]} != hip {st[k][1]}"
    return o, g


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_scenes_bit_exact(product_lib, oracle, path, seed):
    s, pose = random_scene(seed)
    o, g = _frames_equal(oracle, s, pose, (160, 45, 1), f"seed {seed}")
    hit = g.read(abi.BUF_PRIM_ID).reshape(-1)
    assert len(np.unique(hit[hit >= 0])) >= 5, "the camera sees next to nothing: the draw needs another look"
    o.close(); g.close()


@pytest.mark.parametrize("seed", list(range(27, 45)))          # (seed % 9 = the mode: every mode twice)
def test_random_scenes_pushed_one_way_bit_exact(product_lib, oracle, path, seed):
    s, pose = random_scene(seed)
    tag = harden(s, pose, seed)
    size = [(160, 45, 1), (97, 31, 1), (64, 20, 2), (200, 60, 1)][seed % 4]
    o, g = _frames_equal(oracle, s, pose, size, f"seed {seed} ({tag})", frames=2)
    o.close(); g.close()


def test_random_scene_with_supersampling_and_post_stage(product_lib, oracle, path):
    """one of the draws at ss 2 through the delivered frame (A-trous with its in-place iteration, the serial exposure sum, tone map and
    downsample: SURVEY 8-f1): history, denoised buffer, exposure and the SDR cells equal the oracle's"""
    s, pose = random_scene(11)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, 96, 27, 2, pose, flat=flat)
    g = RaytraceRenderer(flat, 96, 27, pose["fov"], 2)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(3):
        so = o.render(stages=2, threads=8, want_sdr=True)
        sg = g.TryFlipAndBlit(want_sdr=True)
        assert pu.mismatch_count(o.read(abi.BUF_TAA_HISTORY), g.read(abi.BUF_TAA_HISTORY)) == 0, f"frame {f}: history"
        assert pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED)) == 0, f"frame {f}: denoised"
        assert np.float32(o.stats.exposure).view(np.uint32) == np.float32(g.stats.exposure).view(np.uint32), f"frame {f}: exposure"
        assert pu.mismatch_count(so, sg) == 0, f"frame {f}: SDR"
    o.close(); g.close()


# ---- call sequences ---------------------------------------------------------------------------------------------------------------------------
# The frames above start from a fresh context.  A caller does not: RaytraceEntity moves the camera every frame (RaytraceEntity.cs:221-232), entities
# move objects (Scene.cs:122-127) and lights (DayNightCycle.cs:80-89), the terminal is resized (RaytraceEntity.cs:289), scenes are switched.  A
# drawn SEQUENCE of those calls goes to the oracle and to the library, with a frame - sometimes the delivered SDR frame - compared after each.
SEQ_SIZES = [(160, 45, 1), (97, 31, 1), (64, 20, 2), (33, 9, 3), (120, 40, 1)]
FRAME_COUNTERS = [0, 5, (1 << 31) - 3, (1 << 31) - 1, (1 << 32) - 2, 1 << 40, (1 << 62) + 12345]      # frameIdx = frame & 0x7fffffff (RaytraceRenderer.cs:175)


def _moved(s, rng):
    """a few objects of the scene somewhere else (geometry only: materials, meshes and grids stay what the upload made them)"""
    d = lambda: _f(rng.uniform(-0.6, 0.6))
    sh = lambda v: vec3(_f(v[0] + dx), _f(v[1] + dy), _f(v[2] + dz))
    cand = [o for o in s.Objects if isinstance(o, (Sphere, Box, CylinderY, Triangle, Disk))]
    for o in rng.choice(np.array(cand, dtype=object), size=min(len(cand), int(rng.integers(1, 9))), replace=False):
        dx, dy, dz = d(), d(), d()
        if isinstance(o, (Sphere, Disk)): o.Center = sh(o.Center)
        elif isinstance(o, Box): o.Min, o.Max = sh(o.Min), sh(o.Max)
        elif isinstance(o, CylinderY): o.Center = sh(o.Center); o.YMin = _f(o.YMin + dy); o.YMax = _f(o.YMax + dy)
        else: o.A, o.B, o.C = sh(o.A), sh(o.B), sh(o.C)


def run_sequence(oracle, seed, steps=12, log=print, flight=False, devices=None, rccl=False):
    """Returns the list of differences found (empty = the library followed the oracle through the whole sequence).
    flight: the library renders through ycge_render_frame_async / _async_sdr, one to three frames queued before ycge_wait - the calls between them
    (camera, lights, moved objects, resize ..) arrive while frames are in flight; the last frame of a burst and every SDR array held are compared.
    devices: the one-process multi-GPU context of the C# host (config.n_devices; [0, 0, 0] = three contexts sharing the one GPU of the box: the root
    and two peers with their threads, tiles pushed into the root's buffers) - every call has to reach all of them; rccl: its all-gather form
    (config.multi_device_exchange = YCGE_EXCHANGE_RCCL) as a world of one."""
    rng = np.random.default_rng(77_000 + seed)
    u = lambda lo, hi: _f(rng.uniform(lo, hi))
    s, pose = random_scene(seed)
    w, h, ss = SEQ_SIZES[seed % len(SEQ_SIZES)]
    flat = uploaded = flatten(s)
    gone = []
    keep = [flat]                    # (the oracle and the library copy during the call; kept anyway until the contexts are gone)
    o = oracle.OracleRenderer(s, w, h, ss, pose, flat=flat)
    cfg = abi.default_config()
    if rccl: cfg.multi_device_exchange = abi.EXCHANGE_RCCL
    debug = not flight and not rccl          # frames in flight keep neither debug captures nor counters; the all-gather form gathers the frame's planes, NOT the debug capture's (rays, hit ids, RNG state stay unwritten on devices[0]: a known gap of a debug feature)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=cfg, capture_debug=debug, count_work=not flight, devices=[0] if rccl else devices)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    found, held, burst_left, lights_now = [], [], 0, None
    for step in range(steps):
        ops = []
        if rng.random() < 0.45:
            k = [1e-4, 1e-3, 0.02, 0.5][int(rng.integers(0, 4))]          # below and above TemporalAA's reset thresholds (TemporalAA.cs:58-67)
            pose = dict(pose, pos=tuple(_f(c + rng.uniform(-k, k)) for c in pose["pos"]), yaw=_f(pose["yaw"] + rng.uniform(-k, k) * 0.3), pitch=_f(pose["pitch"] + rng.uniform(-k, k) * 0.1))
            ops.append(f"camera {k}")
        if rng.random() < 0.1:
            pose = dict(pose, fov=u(30, 90)); ops.append("fov")
        if ops:
            o.set_camera(pose["pos"], pose["yaw"], pose["pitch"], pose["fov"])
            g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"]); g.SetFov(pose["fov"])
        if rng.random() < 0.2:
            lights = [PointLight(vec3(u(-8, 8), u(0.5, 9), u(-20, 4)), vec3(u(0.5, 1), u(0.5, 1), u(0.5, 1)), 0.0 if rng.random() < 0.2 else u(5, 120)) for _ in range(int(rng.integers(0, 5)))]
            amb, top, bot = AmbientLight(vec3(u(0.5, 1), u(0.5, 1), u(0.5, 1)), u(0, 0.2)), vec3(u(0, 1), u(0, 1), u(0, 1)), vec3(u(0, 1), u(0, 1), u(0, 1))
            g.UpdateLights(lights, amb, top, bot)
            s.Lights, s.Ambient, s.BackgroundTop, s.BackgroundBottom = lights, amb, top, bot      # (the oracle takes moved objects as a full upload: the scene it is flattened from carries the lights of the moment)
            arr = (abi.Light * max(1, len(lights)))()
            for i, l in enumerate(lights):
                arr[i].position, arr[i].color, arr[i].intensity = abi.Vec3(*l.Position), abi.Vec3(*l.Color), float(l.Intensity)
            a_, t_, b_ = abi.Vec3(*amb.Color), abi.Vec3(*top), abi.Vec3(*bot)
            lights_now = (arr, len(lights), C.byref(a_), float(amb.Intensity), C.byref(t_), C.byref(b_)); keep.append((arr, a_, t_, b_))
            assert o.L.orc_scene_update_lights(o.ctx, *lights_now) == 0
            ops.append(f"{len(lights)} lights")
        if rng.random() < 0.3:
            _moved(s, rng)
            k = int(rng.integers(0, 5))          # ... and sometimes some leave, some come back, copies appear, the order changes (Scene.Objects is a list entities add to and remove from)
            if k == 1 and len(s.Objects) > 4: gone += [s.Objects.pop(int(i)) for i in sorted(rng.choice(len(s.Objects), size=len(s.Objects) // 3, replace=False).tolist(), reverse=True)]
            elif k == 2 and gone: s.Objects += gone; gone = []
            elif k == 3: s.Objects += [copy.copy(s.Objects[int(i)]) for i in rng.integers(0, len(s.Objects), 20) if not isinstance(s.Objects[int(i)], (Mesh, VolumeGrid))]
            elif k == 4: s.Objects = [s.Objects[int(i)] for i in rng.permutation(len(s.Objects))]
            flat = flatten(s, against=uploaded); keep.append(flat)          # the moved objects against the tables of the last UPLOAD: what the library holds
            assert o.L.orc_scene_upload(o.ctx, flat.byref()) == 0
            if lights_now is not None: assert o.L.orc_scene_update_lights(o.ctx, *lights_now) == 0          # (the oracle took a full upload: the uploaded lights came back with it)
            g.UpdateObjects(flat)
            ops.append(["objects moved", "objects moved, a third left", "objects moved, the leavers back", "objects moved, 20 copies", "objects moved, another order"][k])
        if rng.random() < 0.08:
            s, _ = random_scene(1000 + 50 * seed + step)
            flat = uploaded = flatten(s); keep.append(flat); gone = []; lights_now = None
            assert o.L.orc_scene_upload(o.ctx, flat.byref()) == 0
            g.UploadScene(flat)
            ops.append(f"new scene ({len(s.Objects)} objects)")
        if rng.random() < 0.12:
            w, h, ss = SEQ_SIZES[int(rng.integers(0, len(SEQ_SIZES)))]
            o.resize(w, h, ss); g.Resize(w, h, ss)
            ops.append(f"resize {w}x{h} ss {ss}")
        if rng.random() < 0.1:
            n = FRAME_COUNTERS[int(rng.integers(0, len(FRAME_COUNTERS)))]
            o.set_frame_counter(n); g.set_frame_counter(n)
            ops.append(f"frame counter {n}")
        with_sdr = rng.random() < 0.35
        label = f"sequence {seed} step {step} [{', '.join(ops) or 'nothing'}{', SDR' if with_sdr else ''}]"
        if os.environ.get("YCGE_SEQ_VERBOSE"): log("next:", label, f"({len(s.Objects)} objects, {w}x{h} ss {ss})")
        if flight:
            if not burst_left: burst_left = int(rng.integers(1, 4))
            so = o.render(stages=2, threads=8, want_sdr=True) if with_sdr else o.render(stages=1, threads=8)
            if held and held[0][2].shape != (h, w, 2, 3): held = []          # (a resize joined the frames and retired the arrays of the old size; their frames were compared below or are skipped)
            arr = g.RenderAsync(sdr_slot=len(held) if with_sdr else None)
            if with_sdr: held.append((label, so, arr))
            burst_left -= 1
            if burst_left and step + 1 < steps:
                log(label, "in flight")
                continue
            g.Wait()
            bad = {}
            for name, which in (("current_hdr", abi.BUF_CURRENT_HDR), ("g_albedo", abi.BUF_G_ALBEDO), ("g_normal", abi.BUF_G_NORMAL), ("g_depth", abi.BUF_G_DEPTH),
                                ("sky", abi.BUF_SKY_MASK), ("taa_history", abi.BUF_TAA_HISTORY)):
                n = pu.mismatch_count(o.read(which), g.read(which))
                if n: bad[name] = n
            for lab, want, got in held:
                if pu.mismatch_count(want, got): bad["sdr of " + lab] = pu.mismatch_count(want, got)
            held = []
        else:
            if with_sdr:
                so = o.render(stages=2, threads=8, want_sdr=True); sg = g.TryFlipAndBlit(want_sdr=True)
            else:
                o.render(stages=1, threads=8); g.TryFlipAndBlit()
            if debug:
                st = pu.compare_frame(o, g)
                bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
                bad.update({k: st[k] for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if st[k][0] != st[k][1]})
            else:
                bad = {}
                for name, which in (("current_hdr", abi.BUF_CURRENT_HDR), ("g_albedo", abi.BUF_G_ALBEDO), ("g_normal", abi.BUF_G_NORMAL), ("g_depth", abi.BUF_G_DEPTH),
                                    ("sky", abi.BUF_SKY_MASK), ("taa_history", abi.BUF_TAA_HISTORY)):
                    n = pu.mismatch_count(o.read(which), g.read(which))
                    if n: bad[name] = n
                bad.update({k: (int(getattr(o.stats, k)), int(getattr(g.stats, k))) for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if int(getattr(o.stats, k)) != int(getattr(g.stats, k))})
            if int(o.stats.history_reset) != int(g.stats.history_reset): bad["history_reset"] = (int(o.stats.history_reset), int(g.stats.history_reset))
            if with_sdr:
                if pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED)): bad["denoised"] = True
                if np.float32(o.stats.exposure).view(np.uint32) != np.float32(g.stats.exposure).view(np.uint32): bad["exposure"] = (float(o.stats.exposure), float(g.stats.exposure))
                if pu.mismatch_count(so, sg): bad["sdr"] = pu.mismatch_count(so, sg)
        log(label, "DIFFERS " + repr(bad) if bad else "equal")
        if bad: found.append((label, bad))
    o.close(); g.close()
    return found


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_call_sequences_follow_the_oracle(product_lib, oracle, path, seed):
    found = run_sequence(oracle, seed, steps=14)
    assert not found, found


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
@pytest.mark.parametrize("form", ["three contexts on the one GPU", "all-gather, a world of one"])
def test_random_call_sequences_on_the_one_process_multi_gpu_context(product_lib, oracle, seed, form, monkeypatch):
    monkeypatch.delenv("YCGE_PATH", raising=False)
    found = run_sequence(oracle, seed, steps=14, devices=[0, 0, 0] if form.startswith("three") else None, rccl=form.startswith("all-gather"))
    assert not found, found


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_call_sequences_with_frames_in_flight(product_lib, oracle, path, seed):
    found = run_sequence(oracle, seed, steps=16, flight=True)
    assert not found, found


# ---- the multi-GPU entry points on drawn scenes ----------------------------------------------------------------------------------------------------
def run_tile_split(seed, log=print):
    """`world` ranks emulated on one GPU (2 - 8, drawn), both tiled forms of include/ycge.h against ONE context rendering whole frames, over four
    frames of a camera that moves below and above the TAA thresholds: (a) ycge_trace_tiles -> slabs side by side as an all-gather leaves them ->
    ycge_resolve_gathered on every rank: radiance, G-buffer and history of every rank equal the single context's; (b) the tile-resident form:
    ycge_trace_tiles_resident -> the halo exchange by device copies (ycge_halo_counts' split sizes) -> ycge_resolve_tiles_resident -> the gathered
    history slabs un-permuted equal the single context's history.  Returns the differences found."""
    import torch
    from test_gpu_parity import _exchange_halos
    from yetanotherconsolegameengine_amd import tiles
    rng = np.random.default_rng(55_000 + seed)
    s, pose = random_scene(seed)
    tag = harden(s, pose, seed) if seed % 3 == 2 else "plain"
    w, h, ss = SEQ_SIZES[int(rng.integers(0, len(SEQ_SIZES)))]
    world = int(rng.integers(2, 9))
    lean = bool(rng.random() < 0.3)
    flat = flatten(s)
    mk = lambda rank, n, **kw: RaytraceRenderer(flat, w, h, pose["fov"], ss, rank=rank, world_size=n, **kw)
    single = mk(0, 1)
    gath = [mk(i, world, slab_albedo=not lean) for i in range(world)]
    resi = [mk(i, world) for i in range(world)]
    found = []
    keep, gone, what = [flat], [], ""
    for frame in range(6):
        k = [0.0, 1e-3, 0.02, 0.3][int(rng.integers(0, 4))]
        pose = dict(pose, pos=tuple(_f(c + rng.uniform(-k, k)) for c in pose["pos"]), yaw=_f(pose["yaw"] + rng.uniform(-k, k) * 0.3))
        everyone = [single] + gath + resi
        if frame >= 2:          # between frames, on EVERY context alike: entities move / leave / come back (ycge_scene_update_objects), lights change, the console is resized
            op = int(rng.integers(0, 5))
            if op == 4:
                w, h, ss = SEQ_SIZES[int(rng.integers(0, len(SEQ_SIZES)))]
                for r in everyone: r.Resize(w, h, ss)
                what = f", resized to {w}x{h} ss {ss}"
                op = 9
            if op == 0:
                _moved(s, rng); what = ", objects moved"
            elif op == 1 and len(s.Objects) > 6:
                gone += [s.Objects.pop(int(i)) for i in sorted(rng.choice(len(s.Objects), size=len(s.Objects) // 3, replace=False).tolist(), reverse=True)]; what = ", a third left"
            elif op == 2 and gone:
                s.Objects += gone; gone = []; what = ", the leavers back"
            elif op != 9:
                lights = [PointLight(vec3(_f(rng.uniform(-8, 8)), _f(rng.uniform(0.5, 9)), _f(rng.uniform(-20, 4))), vec3(1, 1, 1), _f(rng.uniform(0, 90))) for _ in range(int(rng.integers(0, 4)))]
                for r in everyone: r.UpdateLights(lights, s.Ambient, s.BackgroundTop, s.BackgroundBottom)
                what = f", {len(lights)} lights"
            if op <= 2:
                f2 = flatten(s, against=flat); keep.append(f2)
                for r in everyone: r.UpdateObjects(f2)
        nb, hb = gath[0].tile_slab_bytes(), resi[0].history_slab_bytes()          # (of the size of the moment)
        n_send = [max(1, sum(r.halo_counts()[0])) for r in resi]
        for r in everyone:
            r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        single.TryFlipAndBlit()
        label = f"tiles {seed} ({tag}, {w}x{h} ss {ss}, world {world}{', lean slabs' if lean else ''}) frame {frame} move {k}{what}"
        bad = {}
        gathered = torch.zeros(world * nb // 4, dtype=torch.float32, device="cuda")
        for i, r in enumerate(gath):
            r.trace_tiles(gathered[i * nb // 4:].data_ptr(), 0, want_stats=True)
        torch.cuda.synchronize()
        for i, r in enumerate(gath):
            r.resolve_gathered(gathered.data_ptr(), 0, want_stats=True)
            if int(r.stats.history_reset) != int(single.stats.history_reset): bad[f"gathered form, rank {i}: history_reset"] = int(r.stats.history_reset)
            for name, which in (("current_hdr", abi.BUF_CURRENT_HDR), ("g_albedo", abi.BUF_G_ALBEDO), ("g_normal", abi.BUF_G_NORMAL), ("g_depth", abi.BUF_G_DEPTH),
                                ("sky", abi.BUF_SKY_MASK), ("taa_history", abi.BUF_TAA_HISTORY)):
                if lean and which == abi.BUF_G_ALBEDO: continue
                n = pu.mismatch_count(single.read(which), r.read(which))
                if n: bad[f"gathered form, rank {i}: {name}"] = n
        send = [torch.zeros(m * 4, dtype=torch.float32, device="cuda") for m in n_send]
        torch.cuda.synchronize()
        for i, r in enumerate(resi):
            r.trace_tiles_resident(send[i].data_ptr(), 0)
        torch.cuda.synchronize()
        recv = _exchange_halos(resi, send, torch)
        hist = torch.zeros(world * hb // 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for i, r in enumerate(resi):
            r.resolve_tiles_resident(recv[i].data_ptr(), hist[i * hb // 4:].data_ptr(), 0, want_stats=True)
            if int(r.stats.history_reset) != int(single.stats.history_reset): bad[f"resident form, rank {i}: history_reset"] = int(r.stats.history_reset)
        torch.cuda.synchronize()
        n = pu.mismatch_count(tiles.unpermute(hist.cpu().numpy(), single.hiW, single.hiH, world, 3), single.read(abi.BUF_TAA_HISTORY))
        if n: bad["resident form: gathered history"] = n
        log(label, "DIFFERS " + repr(bad) if bad else "equal")
        if bad: found.append((label, bad))
    for r in [single] + gath + resi:
        r.close()
    return found


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_scenes_on_emulated_ranks(product_lib, seed, monkeypatch):
    monkeypatch.delenv("YCGE_PATH", raising=False)
    found = run_tile_split(seed)
    assert not found, found


# ---- mesh viewers: the scenes the timed single-launch kernels are for ------------------------------------------------------------------------------
def run_mesh_viewer(oracle, seed, log=print):
    """MeshScenes.BuildBunnyScene's pattern (MeshScenes.cs:117-124: floor, two lights, ONE auto-grounded mesh - the scenes that reach the flat
    single-launch kernels, their longest-first schedule, split blocks and cooperative walk) with a drawn mesh of 60 - 80 000 triangles, a drawn
    material (matte, partial mirror, true mirror, glass), lights moved / dimmed / switched off, a drawn camera around the mesh (sometimes inside its
    bounds, sometimes far off), a drawn size.  Three frames of a slightly moving camera, once with debug capture and counters (the counting kernel
    instances) and once without (the instances the benchmark times), on whatever path YCGE_PATH selects.  Returns the differences found."""
    rng = np.random.default_rng(33_000 + seed)
    u = lambda lo, hi: _f(rng.uniform(lo, hi))
    nu, nv = int(rng.integers(6, 400)), int(rng.integers(5, 100))
    pos, faces = scenes.make_torus_knot(nu, nv, seed=int(rng.integers(1, 1 << 20)))
    k = int(rng.integers(0, 4))
    mat = [scenes.Matte(vec3(u(0, 1), u(0, 1), u(0, 1)), u(0, 0.5), 0.0), scenes.MirrorMat(vec3(u(0.5, 1), u(0.5, 1), u(0.5, 1)), u(0.3, 0.89)),
           Material(vec3(0.95, 0.95, 0.95), 0.0, u(0.9, 1.0)), Material(vec3(1, 1, 1), 0.05, u(0, 0.1), ZERO, u(0.4, 0.95), u(1.1, 2.0), vec3(u(0.6, 1), u(0.6, 1), u(0.6, 1)))][k]
    target = (u(-0.5, 0.5), u(0.3, 0.8), u(0.5, 1.5))
    s = scenes.BuildMeshScene(pos, faces, mat, target)
    if rng.random() < 0.35:          # a material per triangle (ycge_mesh.tri_material): every kind side by side on one mesh, through the flat kernels' records
        mesh = s.Objects[-1]
        n = len(np.asarray(mesh.Triangles).reshape(-1, 9))
        mesh.TriMaterials = [scenes.Matte(vec3(0.9, 0.2, 0.2), 0.1, 0.0), Material(vec3(0.95, 0.95, 0.95), 0.0, 0.95), scenes.MirrorMat(vec3(0.7, 0.8, 0.9), 0.6),
                             Material(vec3(1, 1, 1), 0.0, 0.05, ZERO, 0.85, 1.45, vec3(0.8, 1.0, 0.85)), Material(vec3(0.1, 0.1, 0.1), 0.0, 0.0, vec3(1.5, 1.2, 0.6))]
        mesh.TriMaterialIndex = rng.integers(0, 5, n) if rng.random() < 0.5 else (np.arange(n) // max(1, n // 10)) % 5
        k = 9
    if rng.random() < 0.3: s.Lights[int(rng.integers(0, 2))].Intensity = 0.0
    if rng.random() < 0.3: s.Lights[0].Position = vec3(u(-3, 3), u(1, 6), u(-3, 3))
    if rng.random() < 0.2: s.Lights.pop()
    if rng.random() < 0.2: s.BackgroundTop, s.BackgroundBottom = vec3(u(0, 1), u(0, 1), u(0, 1)), vec3(u(0, 1), u(0, 1), u(0, 1))
    dist = [0.2, 0.6, 1.2, 2.0, 8.0][int(rng.integers(0, 5))]
    ang = u(0, 6.2831)
    cam = (_f(target[0] + dist * np.sin(ang)), _f(target[1] + u(0.0, 0.9)), _f(target[2] + dist * np.cos(ang)))
    d = np.array([target[0] - cam[0], target[1] + 0.4 - cam[1], target[2] - cam[2]], np.float64)          # towards the mesh, give or take: fwd = (sin yaw cos pitch, sin pitch, -cos yaw cos pitch)
    pose = dict(pos=cam, yaw=_f(np.arctan2(d[0], -d[2]) + rng.uniform(-0.2, 0.2)), pitch=_f(np.arcsin(d[1] / np.linalg.norm(d)) + rng.uniform(-0.1, 0.1)), fov=u(22, 60))
    w, h, ss = [(320, 90, 1), (640, 180, 1), (200, 60, 1), (480, 135, 1), (160, 45, 2)][int(rng.integers(0, 5))]
    flat = flatten(s)
    o = oracle.OracleRenderer(s, w, h, ss, pose, flat=flat)
    gs = [RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, count_work=True), RaytraceRenderer(flat, w, h, pose["fov"], ss)]
    found = []
    for f in range(3):
        p = tuple(_f(c + f * 0.0007) for c in pose["pos"])
        o.set_camera(p, pose["yaw"], pose["pitch"], pose["fov"]); o.render(stages=1, threads=8)
        label = f"mesh viewer {seed} ({2 * nu * nv} triangles, material {k}, distance {dist}, {w}x{h} ss {ss}) frame {f}"
        bad = {}
        for gi, g in enumerate(gs):
            g.SetCamera(p, pose["yaw"], pose["pitch"]); g.TryFlipAndBlit()
            if gi == 0:
                st = pu.compare_frame(o, g)
                bad.update({k_: v for k_, v in st.items() if k_.endswith("_mismatch") and v})
                bad.update({k_: st[k_] for k_ in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if st[k_][0] != st[k_][1]})
            else:
                for name, which in (("current_hdr", abi.BUF_CURRENT_HDR), ("g_albedo", abi.BUF_G_ALBEDO), ("g_normal", abi.BUF_G_NORMAL), ("g_depth", abi.BUF_G_DEPTH),
                                    ("sky", abi.BUF_SKY_MASK), ("taa_history", abi.BUF_TAA_HISTORY)):
                    n = pu.mismatch_count(o.read(which), g.read(which))
                    if n: bad["timed instances: " + name] = n
        seen = float((o.read(abi.BUF_PRIM_ID) == 1).mean())
        log(label, f"mesh in {seen:.0%} of the pixels", "DIFFERS " + repr(bad) if bad else "equal")
        if bad: found.append((label, bad))
    o.close()
    for g in gs: g.close()
    return found


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_random_mesh_viewers(product_lib, oracle, path, seed):
    found = run_mesh_viewer(oracle, seed)
    assert not found, found


def test_voxel_lookup_misses_take_the_default_material(product_lib, oracle, path):
    """ycge_grid.default_material (include/ycge.h: the material of a (matId, metaId) pair the lookup does not list - the `default:` arm of the
    reference's palette switch, VoxelMaterialPalette.cs:35-98): half of every grid's lookup entries are dropped and their voxels fall to a drawn
    default; and with default_material < 0 such a voxel is an error from ycge_scene_upload, not a guess."""
    s, pose = random_scene(7)
    harden(s, pose, 7)          # voxel chunks
    flat = flatten(s)
    assert flat.struct.n_grids >= 3
    dropped = 0
    for i in range(flat.struct.n_grids):
        g = flat.grids[i]
        dropped += g.n_lookup - g.n_lookup // 2
        g.n_lookup //= 2
        g.default_material = i % flat.struct.n_materials
    assert dropped >= 6
    o = oracle.OracleRenderer(s, 160, 45, 1, pose, flat=flat)
    g_ = RaytraceRenderer(flat, 160, 45, pose["fov"], 1, capture_debug=True, count_work=True)
    g_.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(2):
        o.render(stages=1, threads=8); g_.TryFlipAndBlit()
        st = pu.compare_frame(o, g_)
        bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
        assert not bad and int(g_.stats.n_vox) == int(o.stats.n_vox) > 1000, (f, bad)
    flat.grids[0].default_material = -1
    with pytest.raises(abi.YcgeError, match="no material for"):
        g_.UploadScene(flat)
    o.close(); g_.close()


@pytest.mark.parametrize("size", [(24, 7, 4), (17, 5, 5), (9, 4, 7), (8, 3, 8), (5, 2, 12), (1, 1, 16), (640, 2, 1), (2, 300, 1), (1, 1, 1), (3, 1, 2)])
def test_unusual_sizes_and_supersampling_through_the_post_stage(product_lib, oracle, path, size):
    """Super-sampling factors up to 16 (ToneMapper's box downsample over ss x 2 ss samples a half-cell, RaytraceRenderer.cs:229-264), consoles of one row,
    one column, one cell: three frames of a drawn scene through TAA, the A-trous iterations (the in-place schedule of a 2-pixel-high grid), exposure
    and tone map - every buffer, the exposure and the SDR cells equal the oracle's."""
    w, h, ss = size
    s, pose = random_scene(50 + w + ss)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, count_work=True)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(3):
        so = o.render(stages=2, threads=8, want_sdr=True); sg = g.TryFlipAndBlit(want_sdr=True)
        st = pu.compare_frame(o, g)
        bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
        assert not bad, (size, f, bad)
        assert pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED)) == 0, (size, f)
        assert np.float32(o.stats.exposure).view(np.uint32) == np.float32(g.stats.exposure).view(np.uint32), (size, f)
        assert pu.mismatch_count(so, sg) == 0, (size, f)
    o.close(); g.close()


def test_the_exact_post_stage_says_where_it_ends(product_lib, oracle):
    """The in-place A-trous iteration addresses its 300 bytes of static weights a pixel by 32-bit offsets: above 14.3 M pixels (or 65 535 a side) the
    EXACT post stage refuses the frame - YCGE_ERR_UNSUPPORTED, a message that says why - instead of wrapping around; nothing else ends there.  On a
    5120 x 2880 trace grid (14.7 M pixels; BASELINE's largest is 8.3 M): frames up to TAA equal the oracle's, the refusal leaves the context
    usable, and the waived form (config.atrous_inplace_exact = 0) delivers its SDR frame."""
    s = Scene()
    s.Add(Sphere(vec3(0.0, 0.5, -3.0), 0.8, Material(vec3(0.8, 0.3, 0.2), 0.1, 0.0)))
    s.Add(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.8, 0.8, 0.8), vec3(0.2, 0.2, 0.2), 1.0), 0.0, 0.0))
    s.Lights.append(PointLight(vec3(2.0, 4.0, 0.0), vec3(1, 1, 1), 40.0))
    pose = dict(pos=(0.0, 1.0, 1.0), yaw=0.0, pitch=-0.1, fov=50.0)
    w, h, ss = 2560, 720, 2
    flat = flatten(s)
    o = oracle.OracleRenderer(s, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    o.render(stages=1, threads=32); g.TryFlipAndBlit()
    assert pu.mismatch_count(o.read(abi.BUF_TAA_HISTORY), g.read(abi.BUF_TAA_HISTORY)) == 0
    with pytest.raises(abi.YcgeError, match="14.3 M pixels") as e:
        g.TryFlipAndBlit(want_sdr=True)
    assert e.value.status == abi.YCGE_ERR_UNSUPPORTED
    g.set_frame_counter(1); o.set_frame_counter(1)                          # (the refused call may have counted a frame: both sides start the next one from the same number)
    o.render(stages=1, threads=32); g.TryFlipAndBlit()                      # the context is none the worse for it
    assert pu.mismatch_count(o.read(abi.BUF_CURRENT_HDR), g.read(abi.BUF_CURRENT_HDR)) == 0
    o.close(); g.close()
    cfg = abi.default_config(); cfg.atrous_inplace_exact = 0
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=cfg)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    sdr = g.TryFlipAndBlit(want_sdr=True)
    assert sdr.shape == (h, w, 2, 3) and np.isfinite(sdr).all() and float(sdr.max()) > 0.2
    g.close()


@pytest.mark.parametrize("what", ["lights", "meshes", "textures", "objects"])
def test_counts_far_beyond_the_configurations(product_lib, oracle, path, what):
    """300 lights (every diffuse vertex walks them all, RaytraceRenderer.cs:560-590), 600 meshes, 400 textures, 40 000 objects with a material each:
    the arrays behind the scene records have no small fixed sizes.  Two frames, every buffer and counter against the oracle."""
    rng = np.random.default_rng(5)
    u = lambda lo, hi: _f(rng.uniform(lo, hi))
    s, pose = random_scene(6)
    size = (120, 34, 1)
    if what == "lights":
        for _ in range(300): s.Lights.append(PointLight(vec3(u(-8, 8), u(0.5, 9), u(-20, 4)), vec3(1, 1, 1), u(0, 5)))
        size = (64, 20, 1)
    elif what == "meshes":
        for k in range(600):
            pos, faces = scenes.make_torus_knot(int(rng.integers(6, 14)), int(rng.integers(4, 7)), seed=k + 1)
            c = np.array([rng.uniform(-9, 9), rng.uniform(0, 6), rng.uniform(-24, -2)], np.float32)
            s.Add(Mesh((pos[faces] * np.float32(0.3) + c).astype(np.float32), Material(vec3(u(0, 1), u(0, 1), u(0, 1)), 0.1, 0.95 if k % 7 == 0 else 0.0)))
    elif what == "textures":
        texs = [Texture(rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4), dtype=np.uint8)) for _ in range(400)]
        for k in range(400):
            c = vec3(u(-9, 9), u(0, 6), u(-24, -2))
            s.Add(XYRect(c[0], _f(c[0] + 0.8), c[1], _f(c[1] + 0.8), c[2], Material(vec3(1, 1, 1), DiffuseTexture=texs[k], UVScale=1.0), 0.0, 0.0))
    else:
        for k in range(40000):
            s.Add(Sphere(vec3(u(-30, 30), u(0, 20), u(-80, -2)), u(0.05, 0.4), Material(vec3(u(0, 1), u(0, 1), u(0, 1)), 0.1, 0.0)))
    o, g = _frames_equal(oracle, s, pose, size, f"{what} beyond the configurations", frames=2)
    o.close(); g.close()


@pytest.mark.parametrize("debug", [False, True])
def test_tables_that_no_object_uses(product_lib, oracle, path, debug):
    """Found by the drawn call sequences (round 6, seed 2204: `Memory access fault by GPU` in the timed k_trace): every mesh and voxel entity LEFT
    Scene.Objects (ycge_scene_update_objects with analytic objects only) while the scene's tables - the last upload's - still list three meshes and
    a grid.  The objects were then `analytic_only`, the context still `has_grid`, and a walk tree installed beside that flag sent the non-counting
    kernel out of bounds (install_walk_tree, csrc/ycge_host.cpp: now only with a grid OBJECT).  Here: the entities leave, come back, leave again;
    and a scene is UPLOADED whose tables list a mesh and a grid nothing uses - on the timed instances (debug False) and the counting ones."""
    s, pose = random_scene(3)          # three meshes, a voxel volume, 230 analytic objects
    assert sum(isinstance(o_, Mesh) for o_ in s.Objects) == 3 and sum(isinstance(o_, VolumeGrid) for o_ in s.Objects) == 1
    everything = list(s.Objects)
    analytic = [o_ for o_ in everything if not isinstance(o_, (Mesh, VolumeGrid))]
    uploaded = flatten(s)
    o = oracle.OracleRenderer(s, 120, 40, 1, pose, flat=uploaded)
    g = RaytraceRenderer(uploaded, 120, 40, pose["fov"], 1, capture_debug=debug, count_work=debug)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    keep = []

    def frame(label):
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
            assert pu.mismatch_count(o.read(which), g.read(which)) == 0, (label, which)

    frame("as uploaded")
    no_grid = [o_ for o_ in everything if not isinstance(o_, VolumeGrid)]
    no_mesh = [o_ for o_ in everything if not isinstance(o_, Mesh)]
    for label, objs in (("entities left", analytic), ("entities back", everything), ("left again, another order", analytic[::-1]), ("only the voxel entity left", no_grid),
                        ("only the meshes left", no_mesh), ("one sphere", analytic[:1]), ("nobody", []), ("everybody back", everything)):
        s.Objects = objs
        f = flatten(s, against=uploaded); keep.append(f)
        assert o.L.orc_scene_upload(o.ctx, f.byref()) == 0
        g.UpdateObjects(f)
        frame(label); frame(label)
    s.Objects = analytic
    unused = flatten(s, against=uploaded); keep.append(unused)          # a scene whose tables list three meshes and a grid that NO object uses, uploaded whole
    assert unused.struct.n_meshes == 3 and unused.struct.n_grids == 1
    assert o.L.orc_scene_upload(o.ctx, unused.byref()) == 0
    g.UploadScene(unused)
    frame("uploaded with unused tables"); frame("uploaded with unused tables")
    o.close(); g.close()


@pytest.mark.parametrize("debug", [False, True])
def test_object_lists_around_a_mesh_viewer(product_lib, oracle, path, debug):
    """MeshScenes' pattern - floor and ONE mesh: the scenes of the flat single-launch kernels - taken through ycge_scene_update_objects: ten more
    objects (a real tree), back to two, the mesh alone, the floor alone, the order swapped, the mesh TWICE, five objects, four (the flat loop's
    limit), two again.  Whatever decides between the flat and the generic kernels has to be decided again at every update: two frames after each,
    against the oracle (the timed instances with debug False, the counting ones with counters)."""
    pos, faces = scenes.make_torus_knot(120, 30)
    s = scenes.BuildMeshScene(pos, faces, scenes.Matte(scenes.Emerald, 0.12, 0.0))
    pose = scenes.MESH_BENCH_POSE
    floor, mesh = s.Objects[0], s.Objects[1]

    def planes(n):
        out = []
        for k in range(n):
            q = copy.copy(floor); q.Point = vec3(0.0, 0.3 + 0.2 * k, 2.0 + k); q.Normal = vec3(0.0, 0.3, -1.0); out.append(q)
        return out

    uploaded = flatten(s)
    o = oracle.OracleRenderer(s, 320, 90, 1, pose, flat=uploaded)
    g = RaytraceRenderer(uploaded, 320, 90, pose["fov"], 1, capture_debug=debug, count_work=debug)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    keep = []
    for label, objs in (("as uploaded", None), ("+ 10 planes", [floor, mesh] + planes(10)), ("floor + mesh again", [floor, mesh]), ("mesh only", [mesh]), ("floor only", [floor]),
                        ("order swapped", [mesh, floor]), ("the mesh twice", [mesh, floor, mesh]), ("5 objects", [floor, mesh] + planes(3)), ("4 objects", [floor, mesh] + planes(2)),
                        ("floor + mesh", [floor, mesh])):
        if objs is not None:
            s.Objects = objs
            f = flatten(s, against=uploaded); keep.append(f)
            assert o.L.orc_scene_upload(o.ctx, f.byref()) == 0
            g.UpdateObjects(f)
        for fr in range(2):
            o.render(stages=1, threads=16); g.TryFlipAndBlit()
            for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
                assert pu.mismatch_count(o.read(which), g.read(which)) == 0, (label, fr, which)
            if debug:
                for k in ("n_rays", "n_box", "n_tri", "n_prim"):
                    assert int(getattr(o.stats, k)) == int(getattr(g.stats, k)), (label, fr, k)
    o.close(); g.close()


@pytest.mark.parametrize("debug", [False, True])
def test_voxel_chunks_streaming_in_and_out(product_lib, oracle, path, debug):
    """What WorldManager does to a VolumeScene while the player walks (WorldManager.cs:372-397, 696-731: chunks attach and detach): the reduced voxel
    world's chunk entities leave and return through ycge_scene_update_objects - a drawn third gone, back, all but one gone, NONE left (the state that
    faulted: tables of grids, no grid object), everybody back, a drawn half - at noon, with the sun's shadow rays through the culled grids and the walk
    tree rebuilt at every step.  Two frames after each, against the oracle."""
    sc, _, _, _, pose = scenes.config_scene(5, small=True, t01=0.5)
    rng = np.random.default_rng(9)
    chunks = list(sc.Objects)
    assert len(chunks) > 20 and all(isinstance(c_, VolumeGrid) for c_ in chunks)
    uploaded = flatten(sc)
    o = oracle.OracleRenderer(sc, 128, 36, 1, pose, flat=uploaded)
    g = RaytraceRenderer(uploaded, 128, 36, pose["fov"], 1, capture_debug=debug, count_work=debug)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    keep = []
    third = [c_ for c_ in chunks if rng.random() < 0.67]
    half = [c_ for c_ in chunks if rng.random() < 0.5]
    for label, objs in (("as uploaded", None), ("a third detached", third), ("all attached", chunks), ("one chunk", chunks[len(chunks) // 2:len(chunks) // 2 + 1]), ("no chunk", []),
                        ("all attached again", chunks), ("half, another order", half[::-1])):
        if objs is not None:
            sc.Objects = objs
            f = flatten(sc, against=uploaded); keep.append(f)
            assert o.L.orc_scene_upload(o.ctx, f.byref()) == 0
            g.UpdateObjects(f)
        for fr in range(2):
            o.render(stages=1, threads=16); g.TryFlipAndBlit()
            for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
                assert pu.mismatch_count(o.read(which), g.read(which)) == 0, (label, fr, which)
            if debug:
                for k in ("n_rays", "n_box", "n_vox"):
                    assert int(getattr(o.stats, k)) == int(getattr(g.stats, k)), (label, fr, k)
    o.close(); g.close()


def test_destroy_with_frames_in_flight(product_lib):
    """ycge_destroy while frames are queued (ycge_render_frame_async / _async_sdr, after a resize or an upload or neither): it joins them, nothing
    faults or hangs, and the page-locked SDR arrays handed out outlive the context (they own their pages) with the frames they were promised."""
    import gc
    for rep in range(9):
        s, pose = random_scene(rep)
        flat = flatten(s)
        g = RaytraceRenderer(flat, 320, 90, pose["fov"], 1)
        g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        held = [g.RenderAsync(sdr_slot=k) if k % 2 == 0 else g.RenderAsync() for k in range(3)]
        if rep % 3 == 0: g.Resize(200, 60, 1)
        elif rep % 3 == 1: g.UploadScene(flat)
        g.RenderAsync(); last = g.RenderAsync(sdr_slot=0)
        g.close()
        arrays = [a for a in held + [last] if a is not None]
        assert all(np.isfinite(a).all() for a in arrays) and float(last.max()) > 0.0
        del held, arrays, last; gc.collect()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_refused_updates_change_nothing(product_lib, oracle, path, seed):
    """ycge_scene_update_objects and ycge_scene_upload given a record the validator refuses (a material index out of range, an unknown type, a mesh
    or grid reference that does not exist): an error code and a message - and the context renders on with what it held, frame for frame the
    oracle's (which was never told)."""
    rng = np.random.default_rng(seed)
    s, pose = random_scene(seed)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, 97, 31, 1, pose, flat=flat)
    g = RaytraceRenderer(flat, 97, 31, pose["fov"], 1, capture_debug=True, count_work=True)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    o.render(stages=1, threads=8); g.TryFlipAndBlit()
    for step in range(5):
        broken = flatten(s, against=flat) if step < 4 else flatten(s)
        i = int(rng.integers(0, broken.struct.n_prims))
        if step % 4 == 0: broken.prims[i].material = 10_000 if broken.prims[i].type < 9 else broken.prims[i].material; broken.prims[0].type, broken.prims[0].material = 0, 10_000
        elif step % 4 == 1: broken.prims[i].type = 77
        elif step % 4 == 2: broken.prims[i].type, broken.prims[i].ref = 9, 999
        else: broken.prims[i].type, broken.prims[i].ref = 10, -3
        with pytest.raises(abi.YcgeError):
            g.UpdateObjects(broken) if step < 4 else g.UploadScene(broken)
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        st = pu.compare_frame(o, g)
        bad = {k: v for k, v in st.items() if k.endswith("_mismatch") and v}
        bad.update({k: st[k] for k in ("n_rays", "n_box", "n_prim") if st[k][0] != st[k][1]})
        assert not bad, (step, bad)
    o.close(); g.close()


@pytest.mark.parametrize("debug", [False, True])
def test_light_counts_through_update_lights(product_lib, oracle, path, debug):
    """ycge_scene_update_lights with light counts far from the upload's (a scene uploaded with two): 500, none, 3, 1 200, 1, 65 - the device array
    grows and shrinks with them.  Two frames after each, against the oracle."""
    rng = np.random.default_rng(1)
    s, pose = random_scene(5)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, 64, 20, 1, pose, flat=flat)
    g = RaytraceRenderer(flat, 64, 20, pose["fov"], 1, capture_debug=debug, count_work=debug)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    o.render(stages=1, threads=8); g.TryFlipAndBlit()
    for n in (500, 0, 3, 1200, 1, 65):
        lights = [PointLight(vec3(_f(rng.uniform(-8, 8)), _f(rng.uniform(0.5, 9)), _f(rng.uniform(-20, 4))), vec3(1, 1, 1), _f(rng.uniform(0, 3))) for _ in range(n)]
        g.UpdateLights(lights, s.Ambient, s.BackgroundTop, s.BackgroundBottom)
        arr = (abi.Light * max(1, n))()
        for i, l in enumerate(lights):
            arr[i].position, arr[i].color, arr[i].intensity = abi.Vec3(*l.Position), abi.Vec3(*l.Color), float(l.Intensity)
        a_, t_, b_ = abi.Vec3(*s.Ambient.Color), abi.Vec3(*s.BackgroundTop), abi.Vec3(*s.BackgroundBottom)
        assert o.L.orc_scene_update_lights(o.ctx, arr, n, C.byref(a_), float(s.Ambient.Intensity), C.byref(t_), C.byref(b_)) == 0
        for fr in range(2):
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
            for which in (abi.BUF_CURRENT_HDR, abi.BUF_TAA_HISTORY):
                assert pu.mismatch_count(o.read(which), g.read(which)) == 0, (n, fr, which)
            if debug: assert int(o.stats.n_rays) == int(g.stats.n_rays), (n, fr)
    o.close(); g.close()


@pytest.mark.parametrize("debug", [False, True])
def test_every_short_object_list_over_a_small_universe(product_lib, oracle, path, debug):
    """DESIGN section 11.6(c), as far as a test can take it: the host-side decisions ycge_scene_update_objects makes again at every call - objects
    analytic_only or not, a grid OBJECT or only a grid table, a walk tree or none (one owner per grid), the flat loop of up to four objects or a real
    tree, the mirror rounds - depend on WHICH objects are in the list.  Universe: a floor, a mirror sphere, a mesh, two voxel grids.  EVERY list of
    0 - 4 of them (repeats allowed: 781 lists) and a few longer ones is installed in turn on one context - each state reached from the one before - and
    a frame compared with the oracle's."""
    import itertools
    rng = np.random.default_rng(2)
    floor = Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.8, 0.8, 0.8), vec3(0.2, 0.2, 0.2), 1.0), 0.0, 0.0)
    ball = Sphere(vec3(-1.2, 0.7, -3.0), 0.7, Material(vec3(0.95, 0.95, 0.95), 0.0, 0.95))
    pos, faces = scenes.make_torus_knot(24, 8)
    mesh = Mesh((pos[faces] * np.float32(0.35) + np.float32([0.8, 0.9, -3.2])).astype(np.float32), Material(vec3(0.2, 0.7, 0.3), 0.1, 0.0))

    def grid(corner, seed):
        r_ = np.random.default_rng(seed)
        cells = np.zeros((6, 6, 6, 2), np.int32)
        cells[..., 0] = np.where(r_.random((6, 6, 6)) < 0.5, r_.integers(1, 12, (6, 6, 6)), 0)
        return VolumeGrid(cells, corner, vec3(0.3, 0.3, 0.3), scenes.VoxelMaterialLookup, True, 0.06, 16.0)

    g1, g2 = grid(vec3(-0.4, 0.0, -2.2), 1), grid(vec3(1.6, 0.0, -4.5), 2)
    universe = [floor, ball, mesh, g1, g2]
    s = Scene()
    s.Objects = list(universe)
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.1)
    s.Lights.append(PointLight(vec3(1.0, 4.0, 0.0), vec3(1, 1, 1), 50.0))
    pose = dict(pos=(0.2, 1.1, 0.8), yaw=0.05, pitch=-0.15, fov=55.0)
    uploaded = flatten(s)
    o = oracle.OracleRenderer(s, 64, 18, 1, pose, flat=uploaded)
    g = RaytraceRenderer(uploaded, 64, 18, pose["fov"], 1, capture_debug=debug, count_work=debug)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    lists = [list(c_) for n in range(5) for c_ in itertools.product(universe, repeat=n)]
    lists += [universe, universe[::-1], universe + universe, [floor, ball, mesh, g1], [g1, g2, g1, g2, mesh], [floor] * 5, [mesh] * 5, [g1] * 5]
    order = rng.permutation(len(lists))
    keep = []
    for n_done, li in enumerate(order):
        s.Objects = lists[int(li)]
        f = flatten(s, against=uploaded); keep = [f]
        assert o.L.orc_scene_upload(o.ctx, f.byref()) == 0
        g.UpdateObjects(f)
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        names = [type(x).__name__[0] + ("1" if x is g1 else "2" if x is g2 else "") for x in s.Objects]
        for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
            assert pu.mismatch_count(o.read(which), g.read(which)) == 0, (n_done, names, which)
        if debug:
            for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
                assert int(getattr(o.stats, k)) == int(getattr(g.stats, k)), (n_done, names, k)
    o.close(); g.close()


# ---- the renderer's constants ------------------------------------------------------------------------------------------------------------------------
def drawn_config(seed):
    """ycge_config with the constants of RaytraceRenderer.cs:31-43, 65, 218, 221-227 DRAWN instead of defaulted (the three the library fixes -
    diffuse bounces 1, mirror bounces 2, refractions 2: they size TraceFull's path stack - stay)."""
    rng = np.random.default_rng(91_000 + seed)
    u = lambda lo, hi: _f(rng.uniform(lo, hi))
    c = abi.default_config()
    c.mirror_threshold = u(0.3, 0.99)
    c.eps = _f(10.0 ** rng.uniform(-5.5, -2.5))
    c.seed_salt = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
    c.taa_alpha = u(0.005, 1.0)
    c.motion_trans_reset = _f(10.0 ** rng.uniform(-4, -1))
    c.motion_rot_reset = _f(10.0 ** rng.uniform(-4, -1))
    c.diffuse_sigma_deg = u(0.0, 80.0)
    c.taa_clamp_radius = int(rng.integers(0, 4))
    c.taa_luminance_pad = u(0.0, 0.6)
    c.atrous_iterations = int(rng.integers(0, 6))
    c.atrous_c_phi, c.atrous_n_phi, c.atrous_z_phi, c.atrous_a_phi = u(0.2, 8.0), u(0.05, 1.5), u(0.2, 6.0), u(0.02, 1.0)
    return c


def _copy_config(c):
    d = abi.Config()
    C.memmove(C.byref(d), C.byref(c), C.sizeof(c))
    return d


def run_drawn_config(oracle, seed, log=print):
    """A drawn scene under a drawn ycge_config, five frames of a camera that moves by amounts around the DRAWN reset thresholds, every other frame
    through the post stage (the drawn iteration count and phis, iteration 1 in place).  Returns the differences found."""
    rng = np.random.default_rng(92_000 + seed)
    s, pose = random_scene(seed)
    w, h, ss = SEQ_SIZES[seed % len(SEQ_SIZES)]
    cfg = drawn_config(seed)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, w, h, ss, pose, cfg=_copy_config(cfg), flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=_copy_config(cfg), capture_debug=True, count_work=True)
    found = []
    for f in range(5):
        k = float(cfg.motion_trans_reset) * [0.0, 0.5, 1.5, 0.9, 3.0][f]
        pose = dict(pose, pos=(_f(pose["pos"][0] + k), pose["pos"][1], pose["pos"][2]), yaw=_f(pose["yaw"] + float(cfg.motion_rot_reset) * [0.0, 0.5, 0.0, 1.2, 0.3][f]))
        o.set_camera(pose["pos"], pose["yaw"], pose["pitch"], pose["fov"]); g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        with_sdr = f % 2 == 1
        label = f"drawn config {seed} frame {f}{' SDR' if with_sdr else ''} (threshold {float(cfg.mirror_threshold):.2f}, eps {float(cfg.eps):.1e}, alpha {float(cfg.taa_alpha):.3f}, sigma {float(cfg.diffuse_sigma_deg):.0f}, clamp radius {cfg.taa_clamp_radius}, {cfg.atrous_iterations} iterations)"
        if with_sdr:
            so = o.render(stages=2, threads=8, want_sdr=True); sg = g.TryFlipAndBlit(want_sdr=True)
        else:
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
        st = pu.compare_frame(o, g)
        bad = {k_: v for k_, v in st.items() if k_.endswith("_mismatch") and v}
        bad.update({k_: st[k_] for k_ in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox") if st[k_][0] != st[k_][1]})
        if int(o.stats.history_reset) != int(g.stats.history_reset): bad["history_reset"] = (int(o.stats.history_reset), int(g.stats.history_reset))
        if with_sdr:
            if pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED)): bad["denoised"] = pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED))
            if np.float32(o.stats.exposure).view(np.uint32) != np.float32(g.stats.exposure).view(np.uint32): bad["exposure"] = (float(o.stats.exposure), float(g.stats.exposure))
            if pu.mismatch_count(so, sg): bad["sdr"] = pu.mismatch_count(so, sg)
        log(label, "reset", int(g.stats.history_reset), "DIFFERS " + repr(bad) if bad else "equal")
        if bad: found.append((label, bad))
    o.close(); g.close()
    return found


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_drawn_renderer_constants(product_lib, oracle, path, seed):
    found = run_drawn_config(oracle, seed)
    assert not found, found


# ---- the one call that may come from another thread ------------------------------------------------------------------------------------------------
def test_set_camera_from_another_thread_never_tears_a_frame(product_lib, oracle, path):
    """include/ycge.h: calls on a context come from one thread EXCEPT ycge_set_camera, which is safe against a concurrent ycge_render_frame - the
    reference takes `lock (camLock)` on both sides (RaytraceRenderer.cs:142-147, 159-166: the input thread moves the camera while the terminal
    loop renders).  A second thread flips the camera between two poses that differ in every component as fast as it can while 150 frames are
    rendered: every frame's rays are - bit for bit - the oracle's rays of that frame number for ONE of the two poses, never a mixture, and both
    poses turn up."""
    import threading
    s, pose_a = random_scene(4)
    pose_b = dict(pos=tuple(_f(c + 1.25) for c in pose_a["pos"]), yaw=_f(pose_a["yaw"] + 0.7), pitch=_f(pose_a["pitch"] - 0.2), fov=_f(pose_a["fov"] + 11.0))
    flat = flatten(s)
    o = oracle.OracleRenderer(s, 16, 4, 1, pose_a, flat=flat)
    g = RaytraceRenderer(flat, 16, 4, pose_a["fov"], 1, capture_debug=True)
    g.SetCamera(pose_a["pos"], pose_a["yaw"], pose_a["pitch"])
    stop = threading.Event()
    L, ctx = g.L, g.ctx
    pa, pb = (C.c_float * 3)(*pose_a["pos"]), (C.c_float * 3)(*pose_b["pos"])

    def flip():          # (ctypes drops the GIL for the length of a foreign call: the two threads really are inside the library at once)
        while not stop.is_set():
            L.ycge_set_camera(ctx, pa, pose_a["yaw"], pose_a["pitch"], pose_a["fov"])
            L.ycge_set_camera(ctx, pb, pose_b["yaw"], pose_b["pitch"], pose_b["fov"])

    th = threading.Thread(target=flip, daemon=True)
    th.start()
    seen = [0, 0]
    try:
        for frame in range(1, 151):
            g._check(L.ycge_render_frame(ctx, None, C.byref(g.stats)))
            rays = g.read(abi.BUF_RAYS)
            want = []
            for pz in (pose_a, pose_b):
                o.set_frame_counter(frame - 1); o.set_camera(pz["pos"], pz["yaw"], pz["pitch"], pz["fov"]); o.render(stages=0)
                want.append(o.read(abi.BUF_RAYS))
            which = [pu.bits_equal(rays, w) for w in want]
            assert any(which), f"frame {frame}: the rays are neither pose's - a torn camera snapshot (origin {rays[0, 0, :3]})"
            seen[which.index(True)] += 1
    finally:
        stop.set(); th.join()
    assert min(seen) > 0, f"one pose never turned up in 150 frames ({seen}): the second thread did not run beside the frames"
    o.close(); g.close()


# ---- values no scene should hold --------------------------------------------------------------------------------------------------------------------
def nan_aware_mismatches(a, b):
    """elements that differ, every NaN counting as equal to every other NaN (x86 and the GPU hand out different default NaNs)"""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype.kind != "f": return int(np.count_nonzero(a != b))
    na, nb = np.isnan(a), np.isnan(b)
    return int(np.count_nonzero(na != nb) + np.count_nonzero((a.view(np.uint32) != b.view(np.uint32)) & ~na & ~nb))


def run_poisoned(oracle, seed, values, what, log=print):
    """A drawn scene with a few poisoned values (random_scenes.poison), two frames.  Returns (differences, frames rendered): the library must take
    the scene and finish its frames whatever the values; what of it must EQUAL the oracle's is the callers' business."""
    s, pose = random_scene(seed, n_range=(20, 120))
    tags = poison(s, pose, seed, values, what)
    flat = flatten(s)
    o = oracle.OracleRenderer(s, 97, 31, 1, pose, flat=flat)
    g = RaytraceRenderer(flat, 97, 31, pose["fov"], 1, capture_debug=True, count_work=True)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    found = []
    for f in range(2):
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        bad = {}
        for name, which in (("rays", abi.BUF_RAYS), ("prim_id", abi.BUF_PRIM_ID), ("sub_id", abi.BUF_SUB_ID), ("hit_t", abi.BUF_HIT_T), ("rng_state", abi.BUF_RNG_STATE),
                            ("current_hdr", abi.BUF_CURRENT_HDR), ("g_albedo", abi.BUF_G_ALBEDO), ("g_normal", abi.BUF_G_NORMAL), ("g_depth", abi.BUF_G_DEPTH),
                            ("sky", abi.BUF_SKY_MASK), ("taa_history", abi.BUF_TAA_HISTORY)):
            n = nan_aware_mismatches(o.read(which), g.read(which))
            if n: bad[name] = n
        for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
            if int(getattr(o.stats, k)) != int(getattr(g.stats, k)): bad[k] = (int(getattr(o.stats, k)), int(getattr(g.stats, k)))
        label = f"poisoned {seed} ({values}, {what}: {tags}) frame {f}, NaN in {int(np.isnan(o.read(abi.BUF_CURRENT_HDR)).any(-1).sum())} pixels"
        log(label, "DIFFERS " + repr(bad) if bad else "equal")
        if bad: found.append((label, bad))
    o.close(); g.close()
    return found, 2


@pytest.mark.parametrize("values,what", [("all", "lights"), ("all", "materials"), ("tame", "geometry")])
def test_poisoned_lights_materials_and_tame_geometry_equal_the_oracle(product_lib, oracle, path, values, what):
    """What IS held bit for bit (NaN for NaN) beyond sane scenes: lights and materials with NaN, +-inf, 1e30, FLT_MAX, denormals, -0.0 in them -
    an intensity, a position, an albedo, an index of refraction of 0 or below, transparency / reflectivity out of range - and GEOMETRY with
    denormal, signed-zero and 1e9 coordinates."""
    for seed in range(12):
        found, _ = run_poisoned(oracle, seed, values, what)
        assert not found, found


def test_wild_geometry_is_survived(product_lib, oracle, path):
    """What is NOT held, and says so: geometry with NaN, +-inf or coordinates whose squares overflow binary32 (>= 1e17 was seen
    to matter).  The trees are still the reference's (the builders follow it through inf and NaN, bit for bit) - but a NaN discriminant makes a NaN
    `t` that the reference's comparisons then ACCEPT as the closest hit, and from there every `t < closest` goes the way the C# happens to be
    written; the kernels' fast forms (hardware min / max in the slab test of a ray with finite reciprocals, ycge_rt.hip.h: box_scene) assume finite
    boxes.  A measured 25-70 % of such frames differ somewhere (profiles/r06/g_fuzz_scenes.txt).  What the library owes such a scene: it takes
    it, renders, returns - no fault, no hang, no error - and rays and RNG state, which no geometry touches, still equal the oracle's."""
    n_diff = 0
    for seed in range(24):
        s, pose = random_scene(seed, n_range=(20, 120))
        poison(s, pose, seed, "wild", "geometry")
        flat = flatten(s)
        o = oracle.OracleRenderer(s, 97, 31, 1, pose, flat=flat)
        g = RaytraceRenderer(flat, 97, 31, pose["fov"], 1, capture_debug=True)
        g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        for f in range(2):
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
            assert pu.bits_equal(o.read(abi.BUF_RAYS), g.read(abi.BUF_RAYS))
            n_diff += bool(nan_aware_mismatches(o.read(abi.BUF_CURRENT_HDR), g.read(abi.BUF_CURRENT_HDR)))
        sdr = g.TryFlipAndBlit(want_sdr=True)          # ... and the post stage finishes on whatever TAA left (NaN radiance included)
        assert sdr.shape == (31, 97, 2, 3)
        o.close(); g.close()
    print(f"wild geometry: radiance differs somewhere in {n_diff} of 48 frames")
