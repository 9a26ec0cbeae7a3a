"""-m gpu: the kernel variants bench.py TIMES, held to the oracle.

Every other oracle comparison runs with config.count_work = 1, i.e. the counting template instances (k_trace<true, .>, the
COUNT stage kernels).  The benchmark runs the non-counting ones - k_trace<false,true> and k_trace_fan<false,true> with their
4-wavefronts-per-SIMD register budget (spills), the occupancy-budgeted stage kernels of the voxel world - in their steady state:
longest-first schedule, cost history over four frames, the head of the schedule fanned out.  Here those binaries run the
BASELINE configs at full size for several frames and every buffer of every frame is compared with the oracle
(reference order and semantics: Objects/MeshBVH.cs:132-236, RaytraceRenderer.cs:448-620, 274-398).
"""
import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten

pytestmark = pytest.mark.gpu

BUFFERS = ("rays", "prim_id", "sub_id", "hit_t", "rng_state", "current_hdr", "g_albedo", "g_normal", "g_depth", "sky", "taa_history")


def _assert_frame(o, g, label):
    st = pu.compare_frame(o, g, check_counters=False)
    bad = {k: st[k + "_mismatch"] for k in BUFFERS if st[k + "_mismatch"]}
    print(label, "fan_blocks", g.stats.fan_blocks, "trace_ms %.3f" % g.stats.trace_ms, bad or "bit-exact")
    assert not bad, f"{label}: {bad}"
    for k in ("current_hdr", "taa_history"):
        assert st[k + "_rms"] <= pu.RMS_TOL


@pytest.mark.parametrize("cfg_n,frames,fan", [(3, 5, None), (4, 6, None), (4, 6, "5")])
def test_timed_mesh_kernels_full_size_steady_state(product_lib, oracle, monkeypatch, cfg_n, frames, fan):
    """Configs 3 and 4 at full size, default path (single launch for mesh viewers), NON-counting kernels, capture on: what the
    benchmark launches - k_trace<false,true> with the cooperative walk of its sparse wavefronts (ycge_coop.hip.h), longest first
    from frame 2 on.  Round 3: a whole frame runs WITHOUT the query fan-out by default (asserted: stats.fan_blocks == 0); the
    third case switches it on (YCGE_FAN=5, as a rank's share of a tiled frame has it) so that k_trace_fan<false,true>, its three
    wavefronts walking cooperatively too, meets the oracle at full size as well - asserted through stats.fan_blocks > 0."""
    monkeypatch.delenv("YCGE_PATH", raising=False)
    if fan is not None:
        monkeypatch.setenv("YCGE_FAN", fan)
    sc, w, h, ss, pose = scenes.config_scene(cfg_n)
    o, g = pu.run_pair(oracle, sc, w, h, ss, pose, frames=1, oracle_threads=64, count=False)
    _assert_frame(o, g, f"cfg{cfg_n} timed variants frame 1")
    fanned = []
    for f in range(2, frames + 1):
        o.render(stages=1, threads=64); g.TryFlipAndBlit()
        _assert_frame(o, g, f"cfg{cfg_n} timed variants frame {f}")
        fanned.append(int(g.stats.fan_blocks))
    if fan is None:
        assert fanned == [0] * len(fanned), f"fan-out on a whole frame: {fanned}"
    else:
        assert fanned[-1] > 0 and fanned[-2] > 0, f"k_trace_fan never launched: {fanned}"
        assert fanned[-1] <= 200
    assert g.timed_steps() > 0
    o.close(); g.close()


def _two_mesh_scene():
    """A plane, a bumpy sphere mesh and a one-leaf mesh (a floating quad: Mesh.Hit on a root that is a leaf) under two lights: shadow rays
    of the plane cross both meshes, shadow rays of the sphere cross the quad."""
    from yetanotherconsolegameengine_amd.scene import AmbientLight, Checker, Material, Mesh, Plane, PointLight, Scene, vec3
    rng = np.random.default_rng(7)
    th = np.linspace(0.05, np.pi - 0.05, 18); ph = np.linspace(0, 2 * np.pi, 28, endpoint=False)
    P = np.array([[np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)] for t in th for p in ph], np.float32)
    P = (P * (1.0 + rng.normal(0, 0.03, (len(P), 1)))).astype(np.float32) * np.float32(0.8) + np.array([0.0, 0.9, -3.0], np.float32)
    tris = []
    for i in range(len(th) - 1):
        for j in range(len(ph)):
            a, b = i * len(ph) + j, i * len(ph) + (j + 1) % len(ph)
            c, d = a + len(ph), b + len(ph)
            tris += [[P[a], P[b], P[c]], [P[b], P[d], P[c]]]
    quad = np.array([[[-0.9, 2.3, -3.6], [0.7, 2.3, -3.6], [0.7, 2.5, -2.2]], [[-0.9, 2.3, -3.6], [0.7, 2.5, -2.2], [-0.9, 2.5, -2.2]]], np.float32)
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.05)
    s.Objects.append(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.8, 0.8, 0.8), vec3(0.3, 0.3, 0.3), 0.5), 0.0, 0.0))
    s.Objects.append(Mesh(np.array(tris, np.float32), Material(vec3(0.8, 0.45, 0.25))))
    s.Objects.append(Mesh(quad, Material(vec3(0.3, 0.5, 0.9))))
    s.Lights.append(PointLight(vec3(1.5, 6.0, -1.0), vec3(1, 1, 1), 80.0))
    s.Lights.append(PointLight(vec3(-3.0, 3.0, -2.0), vec3(0.9, 0.95, 1.0), 40.0))
    return s, dict(pos=(0.2, 1.3, 0.4), yaw=0.03, pitch=-0.12, fov=55.0)


@pytest.mark.parametrize("variant", ["bfs64", "bfs112", "bfs8", "ordered"])
def test_order_free_occlusion_queries_against_the_oracle(product_lib, oracle, monkeypatch, variant):
    """mesh_anyhit_bfs (csrc/ycge_anyhit.hip.h): the shadow rays of a wavefront against a mesh, breadth-first from one shared work list - in
    scenes without transparent materials an occlusion query is `OR over reachable leaves of OR over triangles` (RaytraceRenderer.cs:757-781,
    MeshBVH.cs:132-304), whatever the order.  The product build, a build whose list holds 112 entries (wide rounds are cut back, one-item
    dives happen: every mode of the list on ordinary scenes) and the ordered walk (YCGE_NO_BFS=1) against the oracle: the bunny under its
    two lights at a quarter of config 3's size, three frames, and a scene with two meshes, one of them a single leaf.  (YCGE_BFS = the
    number of asking lanes up to which a batch goes to the list: 64 = every batch.)"""
    from yetanotherconsolegameengine_amd import build
    monkeypatch.delenv("YCGE_PATH", raising=False)
    lib = product_lib
    monkeypatch.setenv("YCGE_BFS", {"bfs64": "64", "bfs112": "64", "bfs8": "8", "ordered": "0"}[variant])          # rays a batch may have for the list to take it
    if variant == "bfs112":
        lib = abi.load_library(build.build_variant("bfs112"))
    steps = 0
    for label, (sc, w, h, ss, pose) in (("bunny", scenes.config_scene(3)), ("two meshes", _two_mesh_scene()[:1] + (160, 45, 1) + _two_mesh_scene()[1:])):
        if label == "bunny":
            w, h = 320, 90
        flat = flatten(sc)
        o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
        g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, lib=lib)
        g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        for f in range(3):
            o.render(stages=1, threads=16); g.TryFlipAndBlit()
            _assert_frame(o, g, f"{variant}: {label} frame {f + 1}")
        lit = g.read(abi.BUF_CURRENT_HDR).reshape(-1, 3).max(axis=1)
        assert (lit > 0).mean() > 0.3
        steps += g.timed_steps()
        o.close(); g.close()
    print(variant, "lane steps", steps)


def test_order_free_occlusion_queries_full_size_small_list(product_lib, oracle, monkeypatch):
    """Config 3 at full size through the 112-entry build: 14 400 wavefronts' shadow batches, two frames (the second in schedule order, its
    heaviest blocks in parts)."""
    from yetanotherconsolegameengine_amd import build
    monkeypatch.delenv("YCGE_PATH", raising=False)
    monkeypatch.setenv("YCGE_BFS", "64")
    lib = abi.load_library(build.build_variant("bfs112"))
    sc, w, h, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, lib=lib)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(2):
        o.render(stages=1, threads=64); g.TryFlipAndBlit()
        _assert_frame(o, g, f"bfs112 cfg3 frame {f + 1}")
    o.close(); g.close()


def test_timed_mesh_kernels_sdr_frame(product_lib, oracle, monkeypatch):
    """The frame the C# wrapper asks for (SDR out): non-counting trace kernels + TAA + post stage on config 3 at full size,
    three frames, against the oracle's stages=2."""
    monkeypatch.delenv("YCGE_PATH", raising=False)
    sc, w, h, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(3):
        so = o.render(stages=2, threads=64, want_sdr=True)
        sg = g.TryFlipAndBlit(want_sdr=True)
        assert pu.mismatch_count(o.read(abi.BUF_TAA_HISTORY), g.read(abi.BUF_TAA_HISTORY)) == 0, f
        assert pu.mismatch_count(o.read(abi.BUF_DENOISED), g.read(abi.BUF_DENOISED)) == 0, f
        assert np.float32(o.stats.exposure).view(np.uint32) == np.float32(g.stats.exposure).view(np.uint32), f
        assert pu.mismatch_count(so, sg) == 0, f
        print(f"cfg3 sdr frame {f + 1}: post_ms {g.stats.post_ms:.3f}")
    o.close(); g.close()


def test_timed_voxel_stage_kernels_full_size(product_lib, oracle, monkeypatch):
    """Config 5 at full size (3840x2160 trace grid), default path (stage pipeline), NON-counting stage kernels: k_wf_trace_p at 6
    wavefronts per SIMD and k_wf_lights at 5 (16 spilled registers) exist only in this form.  Three frames: the 3-frame
    taaHistory gate of SURVEY 8(d)."""
    monkeypatch.delenv("YCGE_PATH", raising=False)
    sc, w, h, ss, pose = scenes.config_scene(5)
    o, g = pu.run_pair(oracle, sc, w, h, ss, pose, frames=1, oracle_threads=64, count=False)
    _assert_frame(o, g, "cfg5 timed variants frame 1")
    for f in (2, 3):
        o.render(stages=1, threads=64); g.TryFlipAndBlit()
        _assert_frame(o, g, f"cfg5 timed variants frame {f}")
    o.close(); g.close()


@pytest.mark.parametrize("t01,ypath", [(0.5, None), (0.5, "megakernel"), (0.8, None), (0.8, "megakernel")])
def test_timed_voxel_kernels_full_size_with_a_light_on(product_lib, oracle, monkeypatch, t01, ypath):
    """Config 5 at full size with a light that SHINES.  At SURVEY 8(d)'s day phase t01 = 0.25 the sun sits on the horizon and
    DayNightCycle.cs:48-82 gives sun AND moon intensity 0, so the timed kernels elide every shadow query there (light_is_dark) and
    the VolumeScene shadow rule - RaytraceRenderer.cs:757-765: a VolumeScene asks the binary Scene.Occluded, any-hit through the
    chunk grids, with the solid-voxel cull of the timed kernels on the way - never met the oracle in the kernels the benchmark
    times.  Noon (t01 = 0.5: sun 300000 * sy^2 = 282 353, moon 0) and night (0.8: moon 438, sun 0), NON-counting kernels, the
    default stage pipeline and the single-launch kernel, three frames (the taaHistory gate of SURVEY 8(d)), every buffer."""
    if ypath is None:
        monkeypatch.delenv("YCGE_PATH", raising=False)
    else:
        monkeypatch.setenv("YCGE_PATH", ypath)
    sc, w, h, ss, pose = scenes.config_scene(5, t01=t01)
    lit = [l.Intensity for l in sc.Lights]
    assert sum(1 for i in lit if i > 0) == 1, lit
    o, g = pu.run_pair(oracle, sc, w, h, ss, pose, frames=1, oracle_threads=64, count=False)
    _assert_frame(o, g, f"cfg5 t01={t01} {ypath or 'default'} frame 1")
    for f in (2, 3):
        o.render(stages=1, threads=64); g.TryFlipAndBlit()
        _assert_frame(o, g, f"cfg5 t01={t01} {ypath or 'default'} frame {f}")
    # a lit frame is not a dark one: some pixel's radiance is above what the sky and the (zero) ambient give
    hdr = g.read(abi.BUF_CURRENT_HDR); sky = g.read(abi.BUF_SKY_MASK)
    assert float(hdr.reshape(-1, 3)[sky.reshape(-1) == 0].max()) > 0.0
    assert g.timed_steps() > 0
    o.close(); g.close()


@pytest.mark.parametrize("ypath", ["wavefront", "megakernel"])
@pytest.mark.parametrize("t01", [0.31, 0.5, 0.8])
def test_timed_voxel_kernels_shadow_rays_graze_culled_grids(product_lib, oracle, monkeypatch, ypath, t01):
    """The cull poses of the test below with the sun (low: 0.31, high: 0.5) or the moon (0.8) ON: shadow rays from the terrain
    towards a light 2 000 units away leave through chunk after chunk of air above the solid voxels - exactly the rays the
    solid-voxel cull of the timed kernels turns away from grids, now with an any-hit answer that decides a pixel's radiance
    (RaytraceRenderer.cs:757-765, 586-602).  Small world, both device paths, two frames per pose."""
    monkeypatch.setenv("YCGE_PATH", ypath)
    sc, w, h, ss, pose = scenes.config_scene(5, small=True, t01=t01)
    assert any(l.Intensity > 0 for l in sc.Lights)
    poses = [pose,
             dict(pose, pitch=0.0), dict(pose, yaw=0.0, pitch=-1.5707964), dict(pose, yaw=1.5707964, pitch=0.0),
             dict(pose, pos=(pose["pos"][0], pose["pos"][1] + 40.0, pose["pos"][2]), pitch=-0.05),
             dict(pose, pos=(pose["pos"][0] - 300.0, pose["pos"][1] + 10.0, pose["pos"][2]), yaw=1.5707964, pitch=-0.02)]
    lit_px = 0
    for k, ps in enumerate(poses):
        o, g = pu.run_pair(oracle, sc, 96, 27, 2, ps, frames=1, oracle_threads=16, count=False)
        _assert_frame(o, g, f"{ypath} t01={t01} pose {k} frame 1")
        o.render(stages=1, threads=16); g.TryFlipAndBlit()
        _assert_frame(o, g, f"{ypath} t01={t01} pose {k} frame 2")
        hdr = g.read(abi.BUF_CURRENT_HDR).reshape(-1, 3); sky = g.read(abi.BUF_SKY_MASK).reshape(-1)
        lit_px += int(((hdr.max(axis=1) > 0) & (sky == 0)).sum())
        o.close(); g.close()
    assert lit_px > 0, "no shaded pixel received light: the test does not exercise the shadow path"


@pytest.mark.parametrize("ypath", ["wavefront", "megakernel"])
def test_timed_voxel_kernels_skip_grids_the_ray_cannot_hit(product_lib, oracle, monkeypatch, ypath):
    """The non-counting kernels do not enter a grid whose solid voxels the ray misses (grid_cull: the box of the solid voxels, one
    voxel of margin); the counting ones walk it as VolumeGrid.Hit does.  Same pixels either way, from poses that put the decision on
    the spot: above the terrain looking along the chunk tops (rays graze the boxes), straight down and straight along an axis
    (direction components of exactly 0: 0 x inf in the slab test), inside a chunk's air, far outside the world."""
    monkeypatch.setenv("YCGE_PATH", ypath)
    sc, w, h, ss, pose = scenes.config_scene(5, small=True)
    poses = [pose,
             dict(pose, pitch=0.0), dict(pose, yaw=0.0, pitch=-1.5707964), dict(pose, yaw=1.5707964, pitch=0.0),
             dict(pose, pos=(pose["pos"][0], pose["pos"][1] + 40.0, pose["pos"][2]), pitch=-0.05),
             dict(pose, pos=(pose["pos"][0] - 300.0, pose["pos"][1] + 10.0, pose["pos"][2]), yaw=1.5707964, pitch=-0.02)]
    for k, ps in enumerate(poses):
        o, g = pu.run_pair(oracle, sc, 96, 27, 2, ps, frames=1, oracle_threads=16, count=False)
        _assert_frame(o, g, f"{ypath} pose {k} frame 1")
        o.render(stages=1, threads=16); g.TryFlipAndBlit()
        _assert_frame(o, g, f"{ypath} pose {k} frame 2")
        o.close(); g.close()


@pytest.mark.parametrize("volume_scene", [False, True])
@pytest.mark.parametrize("ypath", ["wavefront", "megakernel"])
def test_timed_kernels_with_an_all_air_grid_and_a_one_voxel_grid(product_lib, oracle, monkeypatch, ypath, volume_scene):
    """The extremes of the solid-voxel box: a grid without a solid voxel (the box is marked empty: never entered by the timed
    kernels, walked as air by the counting ones and by the oracle), a grid with ONE solid voxel in a corner (the box is that voxel
    plus a voxel of margin, partly outside the grid), a grid that is solid throughout - in front of, beside and behind each other.
    Under a lit light, as a plain Scene (transmittance loop, RaytraceRenderer.cs:767-798) and as a VolumeScene (binary
    Scene.Occluded, :761-765)."""
    from yetanotherconsolegameengine_amd.scene import Material, Plane, PointLight, Scene, Solid, VolumeGrid, vec3
    monkeypatch.setenv("YCGE_PATH", ypath)
    look = lambda m, meta: Material(vec3(0.2 + 0.1 * m, 0.5, 0.3))
    air = np.zeros((8, 8, 8, 2), np.int32)
    one = np.zeros((8, 8, 8, 2), np.int32); one[7, 0, 7] = (2, 0)
    full = np.zeros((4, 4, 4, 2), np.int32); full[..., 0] = 3
    # ... at the origin and 65 536 units away from it, where a binary32 coordinate resolves 1/128 of a voxel: the walk's cells are
    # then whatever its own arithmetic says, and the voxel of margin has to cover the difference to the slab test of the cull
    for far in (0.0, 65536.0):
        s = Scene()
        s.IsVolumeScene = volume_scene
        s.Add(Plane(vec3(0.0, -0.5, 0.0), vec3(0.0, 1.0, 0.0), Solid(vec3(0.6, 0.6, 0.55)), 0.05, 0.0))
        s.Add(VolumeGrid(air, vec3(far - 4.0, 0.0, -far - 10.0), vec3(1.0, 1.0, 1.0), look))          # in front of the others, empty
        s.Add(VolumeGrid(one, vec3(far - 4.0, 0.0, -far - 20.0), vec3(1.0, 1.0, 1.0), look))
        s.Add(VolumeGrid(full, vec3(far + 3.0, 0.0, -far - 14.0), vec3(0.5, 0.5, 0.5), look))
        s.Lights.append(PointLight(vec3(far, 12.0, -far - 6.0), vec3(1, 1, 1), 300.0))
        for k, pose in enumerate((dict(pos=(far, 3.0, -far), yaw=0.0, pitch=-0.1, fov=60.0), dict(pos=(far + 0.5, 4.0, -far - 14.0), yaw=0.0, pitch=-0.4, fov=70.0))):
            o, g = pu.run_pair(oracle, s, 96, 27, 2, pose, frames=1, count=False)
            _assert_frame(o, g, f"{ypath} extremes at {far:g} pose {k} frame 1")
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
            _assert_frame(o, g, f"{ypath} extremes at {far:g} pose {k} frame 2")
            o.close(); g.close()


def test_one_call_drives_several_devices(product_lib, oracle, monkeypatch):
    """config.n_devices = 2: ONE ycge_render_frame call traces the frame on two device contexts (here both on GPU 0 - the
    driver's box has one), the peer pushes its tiles straight into rank 0's frame buffers, TAA and the post stage run on
    rank 0.  Every buffer equals the oracle's over three frames (counters are the sum over the devices), and the SDR frame
    equals the single-device one."""
    monkeypatch.delenv("YCGE_PATH", raising=False)
    sc, _, _, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    w, h = 320, 90
    o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, count_work=True, devices=[0, 0])
    one = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    for r in (g, one):
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(3):
        so = o.render(stages=2, threads=8, want_sdr=True)
        sg = g.TryFlipAndBlit(want_sdr=True)
        s1 = one.TryFlipAndBlit(want_sdr=True)
        st = pu.compare_frame(o, g)
        bad = {k: st[k + "_mismatch"] for k in BUFFERS if st[k + "_mismatch"]}
        assert not bad, (f, bad)
        for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
            assert st[k][0] == st[k][1], (f, k, st[k])
        assert pu.bits_equal(sg, s1) and pu.mismatch_count(so, sg) == 0, f
    # moving an entity and the lights reaches every device
    g.Resize(96, 27, 1); one.Resize(96, 27, 1)
    lights, top, bottom = scenes.sun_moon_lights(0.3)
    for r in (g, one):
        r.UpdateLights(lights, sc.Ambient, top, bottom)
        r.TryFlipAndBlit()
    for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
        assert pu.bits_equal(g.read(which), one.read(which)), which
    # the one-process-per-GPU halves are refused on such a context, and so is a plain world_size > 1 context in render_frame
    with pytest.raises(abi.YcgeError):
        g.trace_tiles(0, 0)
    o.close(); g.close(); one.close()
    with RaytraceRenderer(flat, 96, 27, pose["fov"], 1, rank=0, world_size=2) as half:
        with pytest.raises(abi.YcgeError, match="n_devices"):
            half.TryFlipAndBlit()


def test_one_call_drives_two_real_devices(product_lib, oracle, monkeypatch):
    """The same one-call frame on TWO DIFFERENT GPUs (hipDeviceEnablePeerAccess + k_push_tiles across xGMI): skipped on a box with one
    device, so that the peer path runs the first time a multi-GPU node exists.  Timed (non-counting) instances at full config-3 size:
    what the benchmark launches."""
    n = product_lib.ycge_device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 HIP devices for real peer access; this box has {n}")
    monkeypatch.delenv("YCGE_PATH", raising=False)
    sc, w, h, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, devices=list(range(min(n, 8))))
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(4):
        so = o.render(stages=2, threads=16, want_sdr=True)
        sg = g.TryFlipAndBlit(want_sdr=True)
        st = pu.compare_frame(o, g)
        bad = {k: st[k + "_mismatch"] for k in BUFFERS if st[k + "_mismatch"]}
        assert not bad, (f, bad)
        assert pu.mismatch_count(so, sg) == 0, f
        tiles = [g.stats.device_tiles[i] for i in range(g.stats.n_devices_traced)]
        assert g.stats.n_devices_traced == min(n, 8) and all(t > 0 for t in tiles) and sum(tiles) == ((w * ss + 31) // 32) * ((h * 2 * ss + 7) // 8), tiles
    o.close(); g.close()


@pytest.mark.parametrize("path", ["megakernel", "wavefront"])
def test_zero_intensity_lights_get_no_shadow_query_in_the_timed_kernels(product_lib, oracle, monkeypatch, path):
    """A light with Intensity == 0 adds `throughput * (f * nDotL * (Color * 0 / dist2) * tr)` = a zero to the radiance whatever its
    shadow ray finds (RaytraceRenderer.cs:592-602; the reference traces the ray all the same).  The timed
    kernels skip the query (GLight::dark); the pixels stay the oracle's, bit for bit, on both device paths, with one light dark, with
    all lights dark (no light record is even written), and the timed instances walk fewer steps than with the light on."""
    monkeypatch.setenv("YCGE_PATH", path)
    steps = {}
    for label, dark in (("lit", ()), ("one dark", (1,)), ("all dark", (0, 1))):
        sc, _, _, ss, pose = scenes.config_scene(3)        # the bunny under its two lights (MeshScenes.cs:160-171): shadow rays walk the mesh
        for li in dark:
            sc.Lights[li].Intensity = 0.0
        o, g = pu.run_pair(oracle, sc, 320, 90, ss, pose, frames=0, count=False)
        for f in range(3):
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
            _assert_frame(o, g, f"{path}, {label}, frame {f + 1}")
        steps[label] = g.timed_steps()
        o.close(); g.close()
        # the counting instances trace those rays like the reference and say how many there are (stats.n_rays_dark = the oracle's count)
        oc, gc = pu.run_pair(oracle, sc, 320, 90, ss, pose, frames=1, count=True)
        st = pu.compare_frame(oc, gc)
        assert st["n_rays"][0] == st["n_rays"][1] and st["n_rays_dark"][0] == st["n_rays_dark"][1], (label, st["n_rays"], st["n_rays_dark"])
        assert (st["n_rays_dark"][1] > 0) == bool(dark), (label, st["n_rays_dark"])
        oc.close(); gc.close()
    assert steps["all dark"] < steps["one dark"] < steps["lit"], steps


def test_dynamic_texture_scenes_restart_the_history_every_frame(product_lib, oracle):
    """Scene.HasDynamicTextures (Scenes/Scene.cs:30): `resetHistory = taa.ShouldResetHistory(...) || scene.HasDynamicTextures`
    (RaytraceRenderer.cs:171) - with a static camera every frame still copies current -> history."""
    sc, _, _, ss, pose = scenes.config_scene(2)
    sc.HasDynamicTextures = True
    o, g = pu.run_pair(oracle, sc, 160, 45, ss, pose, frames=0)
    for f in range(3):
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        st = pu.compare_frame(o, g)
        assert st["taa_history_mismatch"] == 0 and st["current_hdr_mismatch"] == 0, f
        assert g.stats.history_reset == 1 and pu.bits_equal(g.read(abi.BUF_TAA_HISTORY), g.read(abi.BUF_CURRENT_HDR)), f
    o.close(); g.close()


def test_timed_step_counter_reports_the_timed_kernels_own_work(product_lib):
    """ycge_read_timed_steps: lane steps of the NON-counting instances, cumulative.  A frame of config 3 walks a positive number of
    steps, about the same number when the same frame is traced again (frame counter rewound: the second time the blocks run in schedule
    order and the heaviest in parts, so other lanes share a wavefront and the cooperative walk takes over at other points - a few
    per cent), and fewer than the reference's box + triangle tests would need one by one (two boxes per node step, two triangles per
    leaf step, a treelet or a whole leaf per cooperative step, shadow queries stop at the first hit)."""
    sc, w, h, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    g = RaytraceRenderer(flat, w // 4, h // 4, pose["fov"], ss)
    c = RaytraceRenderer(flat, w // 4, h // 4, pose["fov"], ss, count_work=True)
    for r in (g, c):
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    assert g.timed_steps() == 0
    g.TryFlipAndBlit(); c.TryFlipAndBlit()
    s1 = g.timed_steps()
    g.set_frame_counter(0); g.TryFlipAndBlit()
    s2 = g.timed_steps() - s1
    assert s1 > 0 and abs(s2 - s1) <= 0.1 * s1, (s1, s2)
    assert s1 <= c.stats.n_box // 2 + c.stats.n_tri + c.stats.n_prim, (s1, c.stats.n_box, c.stats.n_tri)
    assert c.timed_steps() == 0          # the counting instances report the reference's counters instead
    g.close(); c.close()


def test_pipelined_tiled_frames_with_a_moving_camera(product_lib):
    """The trace of frame N+1 is issued BEFORE frame N is resolved (bench.py's two-stream loop), and the camera moves between
    them - sometimes below the TAA reset threshold, sometimes above (TemporalAA.cs:58-67).  Every ycge_resolve_gathered must
    resolve ITS frame: the pose it was traced with decides the history reset and is what CommitCamera stores
    (RaytraceRenderer.cs:159-176, 266).  Reference behaviour = the same frames issued strictly in sequence."""
    import torch
    sc, _, _, ss, pose = scenes.config_scene(2)
    flat = flatten(sc)
    w, h = 160, 45
    moves = [0.0, 0.001, 0.0012, 0.02, 0.0201, 0.0201, 0.05, 0.0505]          # x offsets: small, small, big, small, none, big, small
    yaws = [0.0, 0.0, 0.001, 0.001, 0.01, 0.01, 0.01, 0.0101]

    def mk():
        return RaytraceRenderer(flat, w, h, pose["fov"], ss)

    def cam(r, i):
        r.SetCamera((pose["pos"][0] + moves[i], pose["pos"][1], pose["pos"][2]), pose["yaw"] + yaws[i], pose["pitch"])

    seq, pip = mk(), mk()
    n = seq.tile_slab_bytes() // 4
    slab = torch.empty(n, dtype=torch.float32, device="cuda")
    seq_resets, seq_hist = [], []
    for i in range(len(moves)):
        cam(seq, i)
        seq.trace_tiles(slab.data_ptr(), 0, want_stats=True)
        seq.resolve_gathered(slab.data_ptr(), 0, want_stats=True)
        seq_resets.append(int(seq.stats.history_reset)); seq_hist.append(seq.read(abi.BUF_TAA_HISTORY))
    assert seq_resets == [1, 0, 0, 1, 1, 0, 1, 0], seq_resets
    slabs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(2)]
    pip_resets = []
    # software pipeline of depth 2: trace(i + 1) is issued - with the camera already moved - before resolve(i)
    cam(pip, 0)
    pip.trace_tiles(slabs[0].data_ptr(), 0)
    for i in range(len(moves)):
        if i + 1 < len(moves):
            cam(pip, i + 1)
            pip.trace_tiles(slabs[(i + 1) & 1].data_ptr(), 0)
        pip.resolve_gathered(slabs[i & 1].data_ptr(), 0, want_stats=True)
        pip_resets.append(int(pip.stats.history_reset))
        assert int(pip.stats.frame) == i + 1
        assert pu.bits_equal(pip.read(abi.BUF_TAA_HISTORY), seq_hist[i]), i
    assert pip_resets == seq_resets
    with pytest.raises(abi.YcgeError, match="no traced frame"):
        pip.resolve_gathered(slabs[0].data_ptr(), 0)
    seq.close(); pip.close()


@pytest.mark.parametrize("cfg_n,w,h", [(3, None, None), (5, 96, 27), (2, 160, 45), (5, 1024, 288)])
def test_frames_in_flight_are_the_frames_of_the_synchronous_calls(product_lib, cfg_n, w, h):
    """ycge_render_frame_async: the trace of frame N + 1 beside the TAA of frame N, trace outputs alternating between two sets of
    buffers.  (The voxel world at 1024 x 288: from 4096 tiles on two frames of the stage pipeline are traced at a time - a second set
    of stage queues - and their light loops run beside the next round's trace.)  Reference behaviour = the same frames by TryFlipAndBlit one after the other (RaytraceRenderer.cs:157-267): every
    buffer the newest frame left and the history, bit for bit - after a burst, after single frames, and with synchronous SDR
    frames in between (their post stage reads whichever set is current); the camera moves, sometimes past the TAA reset threshold."""
    sc, w0, h0, ss, pose = scenes.config_scene(cfg_n)
    flat = flatten(sc)
    w, h = w or w0, h or h0
    moves = [0.0, 0.001, 0.0012, 0.02, 0.0201, 0.0201, 0.05, 0.0505, 0.0505, 0.051]
    plan = "aaasaasaaa"            # a: in flight, s: a synchronous frame with SDR output
    watch = (abi.BUF_CURRENT_HDR, abi.BUF_G_ALBEDO, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY, abi.BUF_PREV_DEPTH)

    def cam(r, i):
        r.SetCamera((pose["pos"][0] + moves[i], pose["pos"][1], pose["pos"][2]), pose["yaw"] + 0.3 * moves[i], pose["pitch"])

    seq = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    want, want_sdr = [], {}
    for i, kind in enumerate(plan):
        cam(seq, i)
        if kind == "s":
            want_sdr[i] = seq.TryFlipAndBlit(want_sdr=True)
        else:
            seq.TryFlipAndBlit()
        want.append([seq.read(b) for b in watch])
    seq.close()
    # (1) everything queued back to back, looked at only where the plan has a synchronous frame and at the end
    burst = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    for i, kind in enumerate(plan):
        cam(burst, i)
        if kind == "s":
            sdr = burst.TryFlipAndBlit(want_sdr=True)
            assert pu.bits_equal(sdr, want_sdr[i]), f"SDR of frame {i + 1} after frames in flight"
            for b, a in zip(watch, want[i]):
                assert pu.bits_equal(burst.read(b), a), (i, b)
        else:
            burst.RenderAsync()
    for b, a in zip(watch, want[-1]):
        assert pu.bits_equal(burst.read(b), a), ("burst end", b)
    burst.Wait()
    burst.close()
    # (2) a read-back after every frame: each one joins, the next frame starts from a quiet context again
    single = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    for i, kind in enumerate(plan):
        cam(single, i)
        if kind == "s":
            single.TryFlipAndBlit(want_sdr=True)
        else:
            single.RenderAsync()
        for b, a in zip(watch, want[i]):
            assert pu.bits_equal(single.read(b), a), (i, b)
    single.close()


@pytest.mark.parametrize("cfg_n,w,h", [(3, 320, 90), (5, 96, 27), (3, None, None), (5, 1024, 288)])
def test_frames_in_flight_with_the_post_stage(product_lib, cfg_n, w, h):
    """ycge_render_frame_async_sdr: denoise, exposure and tonemap of frame N beside the traces and TAA of the frames after it (the
    denoiser reads frame N's G-buffer set and, in its first iteration, the TAA history; the exposure state passes from frame to frame).
    Reference behaviour = TryFlipAndBlit(fb) frame after frame (RaytraceRenderer.cs:157-267): the same SDR chexels, exposure included,
    bit for bit, three frames in flight, plain frames and synchronous ones in between, a moving camera."""
    sc, w0, h0, ss, pose = scenes.config_scene(cfg_n)
    flat = flatten(sc)
    w, h = w or w0, h or h0
    moves = [0.0, 0.001, 0.0012, 0.02, 0.0201, 0.0201, 0.05, 0.0505, 0.0505, 0.051, 0.0511, 0.0512]
    plan = "pppappspppap"          # p: in flight with the post stage, a: in flight without, s: synchronous SDR frame

    def cam(r, i):
        r.SetCamera((pose["pos"][0] + moves[i], pose["pos"][1], pose["pos"][2]), pose["yaw"] + 0.3 * moves[i], pose["pitch"])

    seq = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    want = {}
    for i, kind in enumerate(plan):
        cam(seq, i)
        if kind == "a":
            seq.TryFlipAndBlit()
        else:
            want[i] = seq.TryFlipAndBlit(want_sdr=True)
    want_hist = seq.read(abi.BUF_TAA_HISTORY)
    seq.close()
    fl = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    held = {}
    for i, kind in enumerate(plan):
        cam(fl, i)
        if kind == "p":
            held[i] = fl.RenderAsync(sdr_slot=i % 3)
            if len(held) == 3:                      # three frames in flight: look at them, then go on
                fl.Wait()
                for j, a in held.items():
                    assert pu.bits_equal(a, want[j]), f"SDR of frame {j + 1} (in flight)"
                held = {}
        elif kind == "a":
            fl.RenderAsync()
        else:
            sdr = fl.TryFlipAndBlit(want_sdr=True)          # joins: the frames queued before it are complete
            for j, a in held.items():
                assert pu.bits_equal(a, want[j]), f"SDR of frame {j + 1} (in flight, before a synchronous frame)"
            held = {}
            assert pu.bits_equal(sdr, want[i]), f"SDR of the synchronous frame {i + 1}"
    fl.Wait()
    for j, a in held.items():
        assert pu.bits_equal(a, want[j]), f"SDR of frame {j + 1} (in flight, at the end)"
    assert pu.bits_equal(fl.read(abi.BUF_TAA_HISTORY), want_hist)
    fl.close()


@pytest.mark.parametrize("gate", ["1", "0"])
def test_frames_in_flight_say_what_they_do_and_do_not_depend_on_the_placed_gate(product_lib, monkeypatch, gate):
    """The placed-value gate ("a trace starts when the trace before it has placed its last workgroup": the last block INDEX of a launch stores
    a value, the next trace's stream waits for it) rests on an observed dispatcher property and decides timing only.  Frames in flight
    must be the synchronous frames, bit for bit, with the gate on and with YCGE_FLIGHT_PLACED_GATE=0 - where the order of the first frames
    of a burst (synchronous schedule -> cost-slot clears -> fork of the second trace stream) must hold by itself - and on a frame that is
    ONE 8x8 block (the last workgroup is the first: the value is stored before any pixel is traced).  ycge_flight_query reports the state."""
    monkeypatch.setenv("YCGE_FLIGHT_PLACED_GATE", gate)
    monkeypatch.delenv("YCGE_PATH", raising=False)
    sc, _, _, ss, pose = scenes.config_scene(3)
    flat = flatten(sc)
    watch = (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY)
    for (w, h) in ((320, 90), (8, 4)):
        seq = RaytraceRenderer(flat, w, h, pose["fov"], ss)
        fl = RaytraceRenderer(flat, w, h, pose["fov"], ss)
        for r in (seq, fl):
            r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        info = fl.flight_info()
        assert info["two_trace_streams"] == 1 and info["placed_gate"] == int(gate) and info["frames_outstanding"] == 0, info
        n = 0
        for burst in (1, 2, 5, 9):            # bursts of frames in flight between synchronous frames (each burst starts from a quiet context)
            seq.TryFlipAndBlit(); fl.TryFlipAndBlit()
            for _ in range(burst):
                seq.TryFlipAndBlit(); fl.RenderAsync(); n += 1
            assert fl.flight_info()["frames_outstanding"] == 1
            for b in watch:
                assert pu.bits_equal(fl.read(b), seq.read(b)), (w, h, burst, b)
        info = fl.flight_info()
        assert info["frames_outstanding"] == 0          # (a read-back joins)
        if gate == "1":
            assert 0 < info["placed_waits"] <= n, info          # every frame of a burst but its first waits for a placed value
        else:
            assert info["placed_waits"] == 0, info
        seq.close(); fl.close()


def test_frames_in_flight_refuse_what_they_cannot_keep(product_lib):
    sc, _, _, ss, pose = scenes.config_scene(2)
    flat = flatten(sc)
    r = RaytraceRenderer(flat, 64, 18, pose["fov"], ss, count_work=True)
    with pytest.raises(abi.YcgeError, match="ycge_render_frame"):
        r.RenderAsync()
    r.close()
    r = RaytraceRenderer(flat, 64, 18, pose["fov"], ss, rank=0, world_size=2)
    with pytest.raises(abi.YcgeError, match="single-device"):
        r.RenderAsync()
    r.close()
    r = RaytraceRenderer(None, 64, 18, pose["fov"], ss)          # no scene yet: the reference throws (Scene.cs:73), so do both entry points
    with pytest.raises(abi.YcgeError, match="Scene BVH not built"):
        r.RenderAsync()
    with pytest.raises(abi.YcgeError, match="Scene BVH not built"):
        r.RenderAsync(sdr_slot=0)
    r.UploadScene(flat)
    r.RenderAsync(); r.RenderAsync(sdr_slot=0); r.Wait()          # ... and the context is none the worse for it
    # a frame in flight fills its SDR array while the caller runs on: a PAGEABLE array is refused (the device never writes memory whose
    # mapping the library does not control) - the synchronous call takes one and stages it
    import ctypes as C
    plain = np.zeros((18, 64, 2, 3), dtype=np.float32)
    rc = r.L.ycge_render_frame_async_sdr(r.ctx, plain.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == abi.YCGE_ERR_INVALID_ARG and b"page-locked" in r.L.ycge_last_error(r.ctx)
    held = r.RenderAsync(sdr_slot=1); r.Wait()
    r._check(r.L.ycge_render_frame(r.ctx, plain.ctypes.data_as(C.POINTER(C.c_float)), None))
    assert plain.any() and held.any() and plain.shape == held.shape
    r.close()


@pytest.mark.parametrize("cfg_n,w,h", [(3, 320, 90), (5, 96, 27)])
def test_tiled_traces_of_consecutive_frames_on_two_streams(product_lib, cfg_n, w, h):
    """bench.py's RCCL form queues the traces of consecutive frames on two streams taken in turn, so that they may overlap; the
    library keeps a trace's outputs, stack spill area and schedule per frame parity, and makes the stage pipeline of a voxel world
    (shared queues) wait for the frame before.  Reference behaviour = the same frames issued strictly in sequence on one stream."""
    import torch
    sc, _, _, ss, pose = scenes.config_scene(cfg_n)
    flat = flatten(sc)
    moves = [0.0, 0.001, 0.0012, 0.02, 0.0201, 0.0201, 0.05, 0.0505, 0.0506, 0.0507]

    def cam(r, i):
        r.SetCamera((pose["pos"][0] + moves[i], pose["pos"][1], pose["pos"][2]), pose["yaw"] + 0.2 * moves[i], pose["pitch"])

    seq = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    n = seq.tile_slab_bytes() // 4
    slab = torch.empty(n, dtype=torch.float32, device="cuda")
    want = []
    for i in range(len(moves)):
        cam(seq, i)
        seq.trace_tiles(slab.data_ptr(), 0)
        seq.resolve_gathered(slab.data_ptr(), 0)
        want.append((seq.read(abi.BUF_TAA_HISTORY), seq.read(abi.BUF_CURRENT_HDR), seq.read(abi.BUF_G_DEPTH)))
    seq.close()
    two = RaytraceRenderer(flat, w, h, pose["fov"], ss)
    slabs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(2)]
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    s_res = torch.cuda.Stream()
    ev_t = [torch.cuda.Event() for _ in range(2)]; ev_r = [torch.cuda.Event() for _ in range(2)]
    got = []
    for i in range(len(moves)):
        k = i & 1
        cam(two, i)
        with torch.cuda.stream(st[k]):
            st[k].wait_event(ev_r[k])
            two.trace_tiles(slabs[k].data_ptr(), st[k].cuda_stream)
            ev_t[k].record(st[k])
        with torch.cuda.stream(s_res):
            s_res.wait_event(ev_t[k])
            two.resolve_gathered(slabs[k].data_ptr(), s_res.cuda_stream)
            ev_r[k].record(s_res)
        if i in (3, 7, len(moves) - 1):          # a look in between drains everything; the pipeline starts again behind it
            torch.cuda.synchronize()
            got.append((i, two.read(abi.BUF_TAA_HISTORY), two.read(abi.BUF_CURRENT_HDR), two.read(abi.BUF_G_DEPTH)))
    for i, hist, hdr, dep in got:
        assert pu.bits_equal(hist, want[i][0]), ("history", i)
        assert pu.bits_equal(hdr, want[i][1]) and pu.bits_equal(dep, want[i][2]), ("frame", i)
    two.close()


def test_textured_material_and_bad_indices_are_refused(product_lib):
    """A textured material whose texture index points nowhere, a texture without pixels, an object with a material index out of
    range: refused loudly, with the reason."""
    from yetanotherconsolegameengine_amd.scene import Material, Scene, Sphere, vec3
    s = Scene()
    s.Add(Sphere(vec3(0, 1, -3), 1.0, Material(vec3(0.7, 0.3, 0.3), Kind=abi.MAT_TEXTURED)))
    with pytest.raises(abi.YcgeError) as e:
        RaytraceRenderer(s, 32, 9)
    assert e.value.status == abi.YCGE_ERR_INVALID_ARG and "texture index" in str(e.value)
    ok = Scene()
    ok.Add(Sphere(vec3(0, 1, -3), 1.0, Material(vec3(0.7, 0.3, 0.3))))
    f = flatten(ok)
    f.prims[0].material = 5
    with pytest.raises(abi.YcgeError) as e:
        RaytraceRenderer(f, 32, 9)
    assert e.value.status == abi.YCGE_ERR_INVALID_ARG
    with RaytraceRenderer(ok, 32, 9) as r:
        with pytest.raises(abi.YcgeError):
            r.set_frame_counter(-5)
        r.set_frame_counter(2 ** 40)
        r.TryFlipAndBlit()
        assert r.stats.frame == 2 ** 40 + 1


OBJ_TEXT = """# a small hull with every index form MeshLoader.cs:23-55 accepts
v -0.6 0.0 -0.4
v 0.6 0.0 -0.4
v 0.6 0.0 0.4
v -0.6 0.0 0.4
v 0.0 0.9 0.0
vt 0 0
vn 0 1 0
f 1 2 3 4
f 1/1/1 2/1/1 5/1/1
f 2//1 3//1 5//1
f -3 -2 -1
f 4 1 5
v 0.0 -0.5 0.0
f 1 2 3 4 6
f -1 -3 -5 -6
"""


def test_scene_loaded_from_obj_text_against_oracle(product_lib, oracle, tmp_path):
    """SURVEY 8-f4 through the path: OBJ TEXT -> parse_obj (quads / pentagons fan out, `v/vt/vn`, `v//vn` and negative indices) ->
    AddMeshAutoGround placement -> Scene -> both renderers.  Every buffer of three frames is bit-identical."""
    from yetanotherconsolegameengine_amd import mesh_loader
    from yetanotherconsolegameengine_amd.scene import AmbientLight, Checker, Material, Mesh, Plane, PointLight, Scene, vec3
    path = tmp_path / "hull.obj"
    path.write_text(OBJ_TEXT)
    pos, faces = mesh_loader.load_obj(path)
    assert pos.shape == (6, 3) and faces.shape == (2 + 1 + 1 + 1 + 1 + 3 + 2, 3)
    tris = mesh_loader.add_mesh_auto_ground(pos, faces, 1.4, (0.0, 0.0, -3.0))
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.1)
    s.Objects.append(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.8, 0.8, 0.8), vec3(0.25, 0.25, 0.25), 0.7), 0.0, 0.0))
    s.Objects.append(Mesh(tris, Material(vec3(0.85, 0.4, 0.2))))
    s.Lights.append(PointLight(vec3(2.0, 4.0, 0.0), vec3(1, 1, 1), 60.0))
    s.Lights.append(PointLight(vec3(-3.0, 2.5, -1.0), vec3(0.8, 0.9, 1.0), 30.0))
    pose = dict(pos=(0.3, 1.1, 0.0), yaw=0.05, pitch=-0.15, fov=50.0)
    for count in (True, False):
        o, g = pu.run_pair(oracle, s, 160, 45, 1, pose, frames=1, count=count)
        _assert_frame(o, g, f"obj text scene count={count} frame 1")
        assert (g.read(abi.BUF_PRIM_ID) == 1).any()          # the mesh is in view
        for f in (2, 3):
            o.render(stages=1, threads=8); g.TryFlipAndBlit()
            _assert_frame(o, g, f"obj text scene count={count} frame {f}")
        o.close(); g.close()


def test_world_file_to_attached_chunks_against_oracle(product_lib, oracle, tmp_path):
    """SURVEY 8-f4 through the path: a VG01 world file (WorldManager.cs:612-629) is written, read back, cut into chunk grids by the
    streaming-set logic (:372-397, :696-731) and rendered as a VolumeScene by both renderers."""
    from yetanotherconsolegameengine_amd import world_file as wf
    from yetanotherconsolegameengine_amd.scene import AmbientLight, PointLight, Scene, vec3
    rng = np.random.default_rng(3)
    nx, ny, nz = 48, 24, 48
    world = np.zeros((nx, ny, nz, 2), np.int32)
    h = (6 + 5 * np.sin(np.arange(nx)[:, None] * 0.3) + 4 * np.cos(np.arange(nz)[None, :] * 0.25)).astype(int)
    for x in range(nx):
        for z in range(nz):
            world[x, :max(1, h[x, z]), z, 0] = (scenes.STONE, scenes.DIRT, scenes.GRASS)[(x + z) % 3]
    world[20:23, 10:14, 20:23, 0] = scenes.SAND            # a floating block
    world[..., 1] = rng.integers(0, 3, size=(nx, ny, nz)) * (world[..., 0] == scenes.STONE)
    p = tmp_path / "world.vg"
    wf.write_vg01(p, world)
    back = wf.read_vg01(p)
    assert np.array_equal(back, world)
    palette = scenes.VoxelMaterialLookup
    s = Scene()
    s.IsVolumeScene = True
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.0)
    lights, top, bottom = scenes.sun_moon_lights(0.25)
    s.Lights.extend(lights)
    s.BackgroundTop, s.BackgroundBottom = top, bottom
    cam = (24.0, 14.0, 24.0)
    added = wf.attach_view(s, back, cam, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 16, 1, palette)
    assert len(added) >= 9 and len(s.Objects) == len(added)
    pose = dict(pos=cam, yaw=0.6, pitch=-0.45, fov=60.0)
    o, g = pu.run_pair(oracle, s, 128, 36, 1, pose, frames=1, count=False)
    _assert_frame(o, g, "vg01 world frame 1")
    for f in (2, 3):
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        _assert_frame(o, g, f"vg01 world frame {f}")
    assert (g.read(abi.BUF_SKY_MASK) == 0).mean() > 0.3
    o.close(); g.close()


def _solid_box_of(obj):
    """the box of an object's solid voxels as ycge_scene_upload derives it (GGrid::solid_lo / solid_hi: one voxel of margin); None for a
    grid without a solid voxel, everything for any other object"""
    from yetanotherconsolegameengine_amd.scene import VolumeGrid
    f32 = np.float32
    if not isinstance(obj, VolumeGrid):
        return np.full(3, -np.inf, f32), np.full(3, np.inf, f32)
    solid = np.argwhere(np.asarray(obj.Cells)[..., 0] != 0)
    if solid.size == 0:
        return None
    mn, vs = np.array(obj.MinCorner, f32), np.array(obj.VoxelSize, f32)
    return mn + (solid.min(0) - 1).astype(f32) * vs, mn + (solid.max(0) + 2).astype(f32) * vs


WALK_LEAF_NODES = 6          # YCGE_WALK_LEAF_NODES
IN_ORDER = 0x10000000        # YCGE_WALK_IN_ORDER


def _check_walk_tree(g, sc):
    """SceneDev::walk_nodes against the device's own scene nodes: the copies, the rewritten references, every leaf opened into nodes that
    hold its objects in index order under the unions of their solid-voxel boxes, grid_owner"""
    import ctypes as C
    from yetanotherconsolegameengine_amd.scene import VolumeGrid
    f = g.L.ycge_debug_read_walk_tree; f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    grids = [k for k, o in enumerate(sc.Objects) if isinstance(o, VolumeGrid)]          # flatten() numbers grids in object order
    cap = max(1, len(sc.Objects))
    nodes = np.zeros((cap, 16), np.uint32); walk = np.zeros((cap * (1 + 2 * WALK_LEAF_NODES), 16), np.uint32)
    owner = np.zeros(max(1, len(grids)), np.int32); root = np.zeros(2, np.uint32)
    n = f(g.ctx, nodes.ctypes.data, walk.ctypes.data, cap, owner.ctypes.data, len(grids), root.ctypes.data)
    assert n >= 0, n
    if n == 0:
        return 0
    assert list(owner[:len(grids)]) == grids
    assert root[0] == (5 << 29) | 0
    leaf = np.asarray(g.accel(abi.ACCEL_SCENE_LEAF_INDEX)).view(np.int32).ravel()
    boxes = [_solid_box_of(o) for o in sc.Objects]
    grid_no = {k: i for i, k in enumerate(grids)}
    wf = walk.view(np.float32)
    PLANES = {0: ([0, 1, 2], [4, 5, 3]), 1: ([6, 7, 8], [10, 11, 9])}          # GNode: (x y)(z Z)(X Y) per child

    def ref_of(obj):
        return (6 << 29) | grid_no[obj] if obj in grid_no else (4 << 29) | obj

    def objects_below(ref):
        """objects under a walk reference in visit order (left first), checking every leaf node on the way"""
        kind, pay = ref >> 29, ref & 0x1FFFFFFF
        if kind == 6: return [grids[pay]]
        if kind == 4: return [pay]
        assert kind == 5 and pay & IN_ORDER, ref          # left child first
        pay &= ~IN_ORDER
        assert pay >= n and walk[pay, 14] == 0
        out = []
        for side in range(2):
            below = objects_below(int(walk[pay, 12 + side]))
            parts = [boxes[o] for o in below if boxes[o] is not None]
            lo_i, hi_i = PLANES[side]
            assert parts, (pay, side)
            assert (wf[pay, lo_i] == np.min([p[0] for p in parts], 0)).all() and (wf[pay, hi_i] == np.max([p[1] for p in parts], 0)).all(), (pay, side)
            out += below
        return out

    opened = 0
    for i in range(n):
        assert (walk[i, :12] == nodes[i, :12]).all() and walk[i, 14] == 0
        for side in range(2):
            ref, wref = int(nodes[i, 12 + side]), int(walk[i, 12 + side])
            kind, pay = ref >> 29, ref & 0x1FFFFFFF
            if kind == 0:
                assert wref == (5 << 29) | pay
                continue
            assert kind == 1
            objs = [int(leaf[(pay >> 3) + k]) for k in range(pay & 7)]
            keep = [o for o in objs if boxes[o] is not None] or objs[:1]
            if len(keep) == 1:
                assert wref == ref_of(keep[0]), (i, side)
            else:
                assert wref == (5 << 29) | IN_ORDER | (n + (2 * i + side) * WALK_LEAF_NODES), (i, side)
                assert objects_below(wref) == keep, (i, side)
                opened += 1
    return n, opened


def test_the_walk_tree_of_a_voxel_world(product_lib, oracle, monkeypatch):
    """SceneDev::walk_nodes (k_scene_walk, ycge_bvh_build.hip): the scene tree with every leaf opened into nodes over the solid-voxel boxes
    of its objects, built on the device from whichever tree is installed (host builder at upload; device builder after
    ycge_scene_update_objects on a world of >= 1400 chunks) - and what it is for: frames walked down it and down the scene tree
    (YCGE_NO_WALK_TREE) are the same bits, which the oracle parity of every voxel test then covers.  A world with other objects among
    the grids - spheres and meshes: their boxes are everything, they keep the object step and `tree_phase` -, grids of air only (left
    out of a leaf that has others), a fractional lattice."""
    from yetanotherconsolegameengine_amd.scene import Material, PointLight, Scene, Solid, Sphere, VolumeGrid, vec3
    monkeypatch.delenv("YCGE_PATH", raising=False)
    # (1) the small voxel world, host-built tree
    sc, w, h, ss, pose = scenes.config_scene(5, small=True, t01=0.5)
    g = RaytraceRenderer(sc, 96, 27, pose["fov"], 2)
    n, opened = _check_walk_tree(g, sc)
    assert n > 0 and opened > 0
    g.close()
    # (2) grids among other objects, all-air grids, voxels of 0.3 on an uneven corner
    look = lambda m, meta: Material(vec3(0.2 + 0.1 * m, 0.5, 0.3))
    rng = np.random.default_rng(5)
    s = Scene()
    s.IsVolumeScene = True
    for k in range(40):
        cells = np.zeros((8, 8, 8, 2), np.int32)
        if k % 7 != 3 and k not in (16, 17, 18, 19):
            m = rng.integers(0, 8, (rng.integers(1, 6), 3)); cells[m[:, 0], m[:, 1], m[:, 2], 0] = 1 + k % 3
        s.Add(VolumeGrid(cells, vec3(0.37 + 2.4 * (k % 8), 0.11 + 2.4 * (k // 8), -20.0 + 0.3 * (k % 5)), vec3(0.3, 0.3, 0.3), look))
    for k in range(6):
        s.Add(Sphere(vec3(1.0 + 3.0 * k, 5.0, -14.0), 0.6, Solid(vec3(0.7, 0.3, 0.3))))
    # ... and two meshes (one with a tree of its own, one whose root is a leaf): a lane leaves walk_phase for them and comes back to it
    two, _ = _two_mesh_scene()
    from yetanotherconsolegameengine_amd.scene import Mesh
    for k, m in enumerate(o for o in two.Objects if isinstance(o, Mesh)):
        s.Add(Mesh((np.asarray(m.Triangles, np.float32) * np.float32(1.5) + np.array([6.0 + 5.0 * k, 2.0, -12.0], np.float32)).astype(np.float32), m.Mat))
    s.Lights.append(PointLight(vec3(8.0, 30.0, 0.0), vec3(1, 1, 1), 3000.0))
    for ps in (dict(pos=(9.0, 6.0, 6.0), yaw=0.0, pitch=-0.1, fov=60.0), dict(pos=(9.0, 5.0, -19.0), yaw=1.5707964, pitch=0.0, fov=70.0)):
        o, g = pu.run_pair(oracle, s, 96, 27, 2, ps, frames=2, count=False)
        _assert_frame(o, g, "grids among spheres")
        n, opened = _check_walk_tree(g, s)
        assert n > 0 and opened > 0
        monkeypatch.setenv("YCGE_NO_WALK_TREE", "1")
        g2 = RaytraceRenderer(s, 96, 27, ps["fov"], 2)
        g2.SetCamera(ps["pos"], ps["yaw"], ps["pitch"])
        assert _check_walk_tree(g2, s) == 0
        for _ in range(2): g2.TryFlipAndBlit()
        for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_ALBEDO, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY):
            assert pu.bits_equal(g.read(which), g2.read(which)), which
        monkeypatch.delenv("YCGE_NO_WALK_TREE")
        o.close(); g.close(); g2.close()
    # (3) the full world: >= 1400 chunks, so ycge_scene_update_objects builds the tree on the device
    sc, w, h, ss, pose = scenes.config_scene(5, t01=0.5)
    flat = flatten(sc)
    o, g = pu.run_pair(oracle, sc, 96, 27, 2, pose, frames=1, oracle_threads=16, count=False)
    _assert_frame(o, g, "full world, host-built tree")
    n_host, opened_host = _check_walk_tree(g, sc)
    g.UpdateObjects(flat)
    assert _check_walk_tree(g, sc) == (n_host, opened_host) and n_host > 300
    o.render(stages=1, threads=16); g.TryFlipAndBlit()          # (the same objects: the oracle's tree stands) - the frame walked down the device-built trees
    _assert_frame(o, g, "full world, device-built tree and its walk tree")
    o.close(); g.close()


@pytest.mark.parametrize("burst", [3, 4, 7])
@pytest.mark.parametrize("path", ["auto", "wavefront"])
def test_live_texture_updates_between_frames_in_flight(product_lib, burst, path, monkeypatch):
    """ycge_scene_update_texture between ycge_render_frame_async calls on a frame of >= 4 096 tiles, on BOTH device paths: the scene is
    analytic objects only, so `auto` traces it with the single launch (k_trace on two streams, frame_is_single_launch since round 5), and
    YCGE_PATH=wavefront sends the same frames through the stage pipeline's TWO trace streams and second set of queues (odd frames on the
    second) - ycge_flight_query says which ran.  The copy of a live texture's next frame
    must wait for the trace that still samples the old one - whichever stream holds it - and the next trace - on whichever stream - must
    wait for the copy (Renderer/Texture.cs:113-116: a frame samples what GetCurrentFramePtr() showed when it was traced).  Reference
    behaviour = the same updates and frames made synchronously.  An update AFTER the last frame must not reach it; bursts of 3, 4 and 7
    frames end on either stream.  HasDynamicTextures restarts the history every frame, so the history IS the last frame."""
    from yetanotherconsolegameengine_amd.scene import (AmbientLight, Box, LiveTexture, Material, PointLight, Scene, XYRect, XZRect, vec3)
    rng = np.random.default_rng(5)
    cam = LiveTexture(rng.integers(0, 256, (512, 512, 4), dtype=np.uint8))          # 1 MB a frame: a copy long enough to be caught half-way
    video = LiveTexture(rng.integers(0, 256, (256, 384, 3), dtype=np.uint8), flipU=True)
    s = Scene()
    s.HasDynamicTextures = True
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.1)
    s.Add(XZRect(-8.0, 8.0, -14.0, 3.0, 0.0, Material(vec3(0.8, 0.8, 0.8), DiffuseTexture=cam, UVScale=0.1), 0.05, 0.0))
    s.Add(XYRect(-8.0, 8.0, 0.0, 6.0, -12.0, Material(vec3(0.9, 0.9, 0.9), DiffuseTexture=video, UVScale=0.08), 0.0, 0.0))
    s.Add(Box(vec3(1.5, 0.0, -5.0), vec3(3.0, 1.5, -3.5), Material(vec3(1, 1, 1), DiffuseTexture=cam, TextureWeight=0.8), 0.1, 0.0))
    s.Lights.append(PointLight(vec3(-2.0, 5.0, -1.0), vec3(1.0, 0.95, 0.9), 90.0))
    s.BackgroundTop, s.BackgroundBottom = vec3(0.5, 0.7, 1.0), vec3(0.9, 0.95, 1.0)
    flat = flatten(s)
    if path == "wavefront": monkeypatch.setenv("YCGE_PATH", "wavefront")
    else: monkeypatch.delenv("YCGE_PATH", raising=False)
    w, h, fov = 1536, 432, 55.0           # 1536 x 864 trace grid = 5 184 tiles of 32 x 8
    frames = [[rng.integers(0, 256, t.frame.shape, dtype=np.uint8) for t in (cam, video)] for _ in range(burst + 1)]
    watch = (abi.BUF_CURRENT_HDR, abi.BUF_G_ALBEDO, abi.BUF_TAA_HISTORY)

    def drive(r, in_flight):
        r.SetCamera((0.2, 1.7, 2.2), 0.04, -0.22)
        for i in range(burst):
            for t, f in zip((cam, video), frames[i]):
                t.set_frame(f); r.UpdateTexture(t)
            r.RenderAsync() if in_flight else r.TryFlipAndBlit()
        if in_flight:           # the frames of the NEXT burst arrive while the last trace may still be running
            for t, f in zip((cam, video), frames[burst]):
                t.set_frame(f); r.UpdateTexture(t)
        return [r.read(b) for b in watch]

    seq = RaytraceRenderer(flat, w, h, fov, 1)
    want = drive(seq, False)
    seq.close()
    for rep in range(3):
        fl = RaytraceRenderer(flat, w, h, fov, 1)
        got = drive(fl, True)
        info = fl.flight_info()
        fl.close()
        for b, a, g_ in zip(watch, want, got):
            assert pu.bits_equal(a, g_), (rep, b, int((a.view(np.uint32) != g_.view(np.uint32)).sum()))
    assert info["two_trace_streams"], "this scene was meant to take both trace streams"
    assert info["stage_pipeline"] == (1 if path == "wavefront" else 0), (path, info)
    assert len(np.unique(want[1].reshape(-1, 3).round(3), axis=0)) > 1000, "the albedo is not textured"
