"""-m gpu: SURVEY 8-f2 - the scene-level BVH rebuilt ON THE DEVICE when entities move (ycge_scene_update_objects ->
csrc/ycge_bvh_build.hip; reference Scene.RebuildBVH, Scenes/Scene.cs:122-127 -> Objects/BVH.cs:258-459).  The tree the kernel
leaves - node boxes, pre-order numbering, leaf order - must equal the oracle's builder bit for bit, move after move, and the
frames traced through it must stay identical."""
import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import (AmbientLight, Box, Material, PointLight, Scene, Solid, Sphere, flatten, vec3, ZERO)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_builder_at_every_size(monkeypatch):
    """The library picks the builder by object count (host below YCGE_SCENE_BVH_DEVICE_MIN, default 1 400: the measured crossover);
    these tests are about the DEVICE builder, so they ask for it at every size.  test_builder_is_picked_by_object_count checks the default."""
    monkeypatch.setenv("YCGE_SCENE_BVH_DEVICE_MIN", "1")


POSE = dict(pos=(0.0, 6.0, 14.0), yaw=0.0, pitch=-0.35, fov=55.0)


def _crowd(n, seed):
    """n small spheres and boxes over a 24 x 6 x 24 region, materials shared."""
    rng = np.random.default_rng(seed)
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.2)
    mats = [Material(vec3(*rng.uniform(0.2, 0.9, 3)), 0.1, 0.0, ZERO) for _ in range(6)]
    tex = [Solid(vec3(*rng.uniform(0.2, 0.9, 3))) for _ in range(3)]
    for i in range(n):
        c = rng.uniform((-12, 0, -12), (12, 6, 12)).astype(np.float32)
        r = np.float32(rng.uniform(0.05, 0.3))
        if i % 3 == 2:
            s.Add(Box(vec3(*(c - r)), vec3(*(c + r)), tex[i % 3], 0.1, 0.0))
        else:
            s.Add(Sphere(vec3(*c), float(r), mats[i % 6]))
    s.Lights.append(PointLight(vec3(0, 12, 6), vec3(1, 1, 1), 400.0))
    s.BackgroundTop, s.BackgroundBottom = vec3(0.5, 0.7, 1.0), vec3(0.9, 0.95, 1.0)
    return s


def _move(scene, rng, fraction):
    for o in scene.Objects:
        if rng.random() >= fraction:
            continue
        d = rng.uniform(-0.8, 0.8, 3).astype(np.float32)
        if isinstance(o, Sphere):
            o.Center = vec3(*(np.asarray(o.Center, np.float32) + d))
        else:
            o.Min = vec3(*(np.asarray(o.Min, np.float32) + d)); o.Max = vec3(*(np.asarray(o.Max, np.float32) + d))


def _same_tree(o, g, label):
    assert pu.bits_equal(o.accel(abi.ACCEL_SCENE_NODES), g.accel(abi.ACCEL_SCENE_NODES)), f"{label}: scene nodes differ"
    assert pu.bits_equal(o.accel(abi.ACCEL_SCENE_LEAF_INDEX), g.accel(abi.ACCEL_SCENE_LEAF_INDEX)), f"{label}: leaf order differs"


def _frame_parity(o, g, label):
    o.render(stages=1, threads=16); g.TryFlipAndBlit()
    st = pu.compare_frame(o, g)
    for k in ("rays", "prim_id", "sub_id", "hit_t", "rng_state", "sky", "g_depth", "current_hdr", "taa_history", "g_albedo", "g_normal"):
        assert st[k + "_mismatch"] == 0, f"{label}: {k} differs in {st[k + '_mismatch']} elements"
    for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
        assert st[k][0] == st[k][1], f"{label}: counter {k}"


def test_hundred_entity_moves_rebuild_the_tree_on_the_device(product_lib, oracle):
    n = 2300
    scene = _crowd(n, 11)
    o, g = pu.run_pair(oracle, scene, 160, 90, 1, POSE, frames=1)
    _same_tree(o, g, "upload")
    rng = np.random.default_rng(5)
    for step in range(100):
        _move(scene, rng, 0.02 if step % 10 else 1.0)         # a few entities most steps, everything every tenth
        moved = flatten(scene)
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"move {step}")
        if step % 25 == 24:
            _frame_parity(o, g, f"frame after move {step}")
    st = g.scene_bvh_stats()
    print("scene BVH builds:", st)
    assert st["device_builds"] == 100 and st["host_fallbacks"] == 0 and st["host_builds"] == 1       # the one host build is the upload
    o.close(); g.close()


@pytest.mark.parametrize("n", [1, 2, 4, 5, 9, 64, 65, 257, 2560])
def test_device_build_at_the_sizes_that_change_its_code_path(product_lib, oracle, n):
    """A root that is a leaf (n <= 4), one split, a node of exactly one / just over one 64-item chunk, the kernel's capacity."""
    scene = _crowd(n, 100 + n)
    o, g = pu.run_pair(oracle, scene, 96, 54, 1, POSE, frames=1)
    rng = np.random.default_rng(n)
    for step in range(3):
        _move(scene, rng, 1.0)
        moved = flatten(scene)
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"n={n} move {step}")
    _frame_parity(o, g, f"n={n}")
    assert g.scene_bvh_stats()["device_builds"] == 3
    o.close(); g.close()


def test_the_array_sort_case_is_built_on_the_device_too(product_lib, oracle):
    """Ten spheres on one centre: no bin split exists, the reference sorts and splits at the median (BVH.cs:386-391) - the kernel runs
    the same introsort restatement as the host builder; the oracle counts the same number of sorts."""
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.3)
    m = Material(vec3(0.8, 0.3, 0.3), 0.1, 0.0, ZERO)
    for i in range(10):
        s.Add(Sphere(vec3(0.0, 1.0, -4.0), 0.3 + 0.05 * i, m))
    s.Lights.append(PointLight(vec3(0, 5, 0), vec3(1, 1, 1), 60.0))
    pose = dict(pos=(0.0, 1.0, 1.0), yaw=0.0, pitch=0.0, fov=50.0)
    o, g = pu.run_pair(oracle, s, 96, 54, 1, pose, frames=1)
    moved = flatten(s)
    assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
    g.UpdateObjects(moved)
    _same_tree(o, g, "coincident centres")
    st = g.scene_bvh_stats()
    assert st["device_builds"] == 1 and st["host_fallbacks"] == 0
    assert st["sort_fallbacks"] == o.build_stats()["scene_sort_fallbacks"] >= 1
    _frame_parity(o, g, "coincident centres")
    o.close(); g.close()


def test_more_items_than_the_kernel_takes_and_the_host_knob(product_lib, oracle, monkeypatch):
    scene = _crowd(2561, 3)
    o, g = pu.run_pair(oracle, scene, 96, 54, 1, POSE, frames=1)
    g.UpdateObjects(flatten(scene))
    _same_tree(o, g, "2561 items")
    assert g.scene_bvh_stats()["device_builds"] == 0 and g.scene_bvh_stats()["host_builds"] == 2
    o.close(); g.close()
    monkeypatch.setenv("YCGE_SCENE_BVH_HOST", "1")
    scene = _crowd(300, 4)
    o, g = pu.run_pair(oracle, scene, 96, 54, 1, POSE, frames=1)
    g.UpdateObjects(flatten(scene))
    _same_tree(o, g, "host knob")
    assert g.scene_bvh_stats()["device_builds"] == 0
    o.close(); g.close()


def _permute_prims(flat, perm):
    """The same scene with Scene.Objects in another order: object records permuted, everything they refer to kept."""
    import ctypes as C
    n = flat.struct.n_prims
    new = (abi.Prim * n)()
    for dst, src in enumerate(perm):
        C.memmove(C.byref(new[dst]), C.byref(flat.prims[int(src)]), C.sizeof(abi.Prim))
    flat.prims = new
    flat.struct.prims = C.cast(new, C.POINTER(abi.Prim))
    return flat


def test_voxel_world_objects_reinstalled_by_the_device_builder(product_lib, oracle):
    """Config 5's chunk grids (the case SURVEY names): the same objects again, then in two other orders - chunk streaming re-orders
    Scene.Objects, and the order decides the tree - with no host build, although the chunk lattice sends every one of these builds
    through the reference's Array.Sort case; frames through the rebuilt trees equal the oracle's."""
    sc, w, h, ss, pose = scenes.config_scene(5)
    flat = flatten(sc)
    o, g = pu.run_pair(oracle, sc, 160, 90, 1, pose, frames=1, oracle_threads=16)
    g.UpdateObjects(flat)
    _same_tree(o, g, "config 5, same order")
    rng = np.random.default_rng(9)
    for k in range(2):
        moved = _permute_prims(flatten(sc), rng.permutation(flat.struct.n_prims))
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"config 5, order {k}")
        _frame_parity(o, g, f"config 5 after the device-side rebuild, order {k}")
    st = g.scene_bvh_stats()
    assert st["device_builds"] == 3 and st["host_fallbacks"] == 0 and st["host_builds"] == 1          # the upload
    assert st["sort_fallbacks"] == o.build_stats()["scene_sort_fallbacks"]
    print("config 5:", flat.struct.n_prims, "objects, device-side rebuild + install", st["last_build_us"], "us,", st["sort_fallbacks"], "Array.Sort cases, depth", st["max_depth"])
    o.close(); g.close()


def _device_and_host_tree(L, b, c):
    import ctypes as C
    n = len(b)
    L.ycge_host_build_tree.restype = C.c_int
    L.ycge_host_build_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ycge_debug_device_bvh.restype = C.c_int
    L.ycge_debug_device_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    hn = np.zeros((2 * n, 10), np.float32); hl = np.zeros(n, np.int32); st = np.zeros(3, np.int32)
    k = L.ycge_host_build_tree(b.ctypes.data, c.ctypes.data, n, 0, hn.ctypes.data, hl.ctypes.data, st.ctypes.data)
    dn = np.zeros((2 * n, 10), np.float32); dl = np.zeros(n, np.int32); res = np.zeros(16, np.uint32)
    kd = L.ycge_debug_device_bvh(b.ctypes.data, c.ctypes.data, n, dn.ctypes.data, dl.ctypes.data, res.ctypes.data, None)
    return (k, hn[:k], hl, int(st[1]), int(st[2])), (kd, dn[:max(kd, 0)], dl, int(res[9]), int(res[11]))


@pytest.mark.parametrize("waves", ["1", "3", "16"])
def test_device_builder_against_the_host_builder_on_hard_inputs(product_lib, monkeypatch, waves):
    """The kernel alone (ycge_debug_device_bvh) on item sets chosen to hit its ties and edge cases, with 1, 3 and 16 of its wavefronts
    taking nodes (the order nodes are split in must not matter): every tree equals the host builder's, which the CPU suite holds
    against the oracle's; the item sets that send the reference through Array.Sort included."""
    monkeypatch.setenv("YCGE_BVH_WAVES", waves)
    rng = np.random.default_rng(77)
    f32 = np.float32

    def boxes(cen, half):
        cen = np.asarray(cen, f32); half = np.broadcast_to(np.asarray(half, f32), cen.shape)
        b = np.concatenate([cen - half, cen + half], 1).astype(f32)
        return b, (f32(0.5) * (b[:, :3] + b[:, 3:])).astype(f32)

    cases = {}
    for n in (5, 6, 63, 64, 65, 127, 128, 129, 700, 2559, 2560):
        cases[f"uniform {n}"] = boxes(rng.uniform(-50, 50, (n, 3)), rng.uniform(0.01, 2.0, (n, 1)))
    g = np.stack(np.meshgrid(np.arange(17), np.arange(8), np.arange(17), indexing="ij"), -1).reshape(-1, 3)
    cases["chunk lattice 17x8x17 (equal costs, equal keys)"] = boxes(g * 16.0 + 8.0, 8.0)
    cases["lattice, shuffled"] = boxes(rng.permutation(g) * 16.0 + 8.0, 8.0)
    cases["flat in y (one axis without extent)"] = boxes(np.c_[rng.uniform(-9, 9, 900), np.full(900, 2.5), rng.uniform(-9, 9, 900)], 0.25)
    cases["a line (two axes without extent)"] = boxes(np.c_[np.linspace(-40, 40, 333), np.zeros(333), np.zeros(333)], 0.5)
    cases["clusters"] = boxes(np.repeat(rng.uniform(-30, 30, (12, 3)), 150, 0) + rng.normal(0, 0.05, (1800, 3)), 0.02)
    cases["signed zeros and huge boxes"] = boxes(np.where(rng.random((400, 3)) < 0.3, -0.0, rng.uniform(-1, 1, (400, 3))), rng.choice([1e-3, 1.0, 1e6], (400, 1)))
    cases["few distinct keys (quantised)"] = boxes(np.round(rng.uniform(-4, 4, (1500, 3))), 0.4)
    cases["all on one point (Array.Sort case)"] = boxes(np.tile(f32([[1.0, 2.0, 3.0]]), (40, 1)), rng.uniform(0.1, 1.0, (40, 1)))
    sorted_sets = 0
    for label, (b, c) in cases.items():
        (k, hn, hl, depth, sorts), (kd, dn, dl, ddepth, dsorts) = _device_and_host_tree(product_lib, b, c)
        assert kd == k, f"{label}: {kd} nodes, host {k}"
        assert hn.tobytes() == dn.tobytes(), f"{label}: nodes differ"
        assert (hl == dl).all(), f"{label}: leaf order differs"
        assert ddepth == depth and dsorts == sorts, f"{label}: depth {ddepth} / sorts {dsorts}, host {depth} / {sorts}"
        sorted_sets += sorts > 0
    print(f"{len(cases)} item sets, {sorted_sets} of them with the reference's Array.Sort case")
    assert sorted_sets >= 2


def test_moved_objects_reach_every_device_of_a_multi_device_context(product_lib, oracle):
    """ycge_scene_update_objects on a context that drives two devices (both GPU 0 here): the tree is built on each of them by the
    same kernel, and the frames stay the oracle's."""
    scene = _crowd(700, 21)
    flat = flatten(scene)
    o = oracle.OracleRenderer(scene, 160, 90, 1, POSE, flat=flat)
    g = RaytraceRenderer(flat, 160, 90, POSE["fov"], 1, capture_debug=True, count_work=True, devices=[0, 0])
    g.SetCamera(POSE["pos"], POSE["yaw"], POSE["pitch"])
    o.render(stages=1, threads=16); g.TryFlipAndBlit()
    rng = np.random.default_rng(2)
    for step in range(3):
        _move(scene, rng, 0.5)
        moved = flatten(scene)
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"two devices, move {step}")
        _frame_parity(o, g, f"two devices, frame after move {step}")
    assert g.scene_bvh_stats()["device_builds"] == 3
    o.close(); g.close()


def test_builder_is_picked_by_object_count(product_lib, oracle, monkeypatch):
    """Default policy: fewer objects than the measured crossover are rebuilt by the host builder, more by the kernel; same trees either way."""
    monkeypatch.delenv("YCGE_SCENE_BVH_DEVICE_MIN")
    for n, dev in ((300, 0), (976, 0), (1400, 1), (2300, 1)):
        scene = _crowd(n, 40 + n)
        o, g = pu.run_pair(oracle, scene, 96, 54, 1, POSE, frames=1)
        _move(scene, np.random.default_rng(n), 1.0)
        moved = flatten(scene)
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"{n} objects")
        st = g.scene_bvh_stats()
        assert st["device_builds"] == dev and st["host_builds"] == 2 - dev, (n, st)
        o.close(); g.close()


def test_moved_objects_of_a_deep_mesh_scene_on_two_devices(product_lib, oracle):
    """A context driving two devices whose scene holds a DEEP mesh (config 3's bunny: its tree needs stack levels beyond the 12 kept in LDS)
    plus entities that move: the peer's spill area must be sized from the ROOT's mesh depth after ycge_scene_update_objects (it was
    sized from the peer's own, never-set value: out-of-bounds stack writes on the peer)."""
    sc, w, h, ss, pose = scenes.config_scene(3)
    rng = np.random.default_rng(8)
    m = Material(vec3(0.7, 0.4, 0.2), 0.1, 0.0, ZERO)
    for i in range(40):
        c = rng.uniform((-1.5, 0.2, -0.5), (1.5, 2.0, 2.5)).astype(np.float32)
        sc.Add(Sphere(vec3(*c), 0.06, m))
    flat = flatten(sc)
    o = oracle.OracleRenderer(sc, 160, 90, 1, pose, flat=flat)
    g = RaytraceRenderer(flat, 160, 90, pose["fov"], 1, capture_debug=True, count_work=True, devices=[0, 0])
    g.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    o.render(stages=1, threads=16); g.TryFlipAndBlit()
    for step in range(2):
        for ob in sc.Objects:
            if isinstance(ob, Sphere):
                ob.Center = vec3(*(np.asarray(ob.Center, np.float32) + rng.uniform(-0.2, 0.2, 3).astype(np.float32)))
        moved = flatten(sc)
        assert o.L.orc_scene_upload(o.ctx, moved.byref()) == 0
        g.UpdateObjects(moved)
        _same_tree(o, g, f"deep mesh, two devices, move {step}")
        _frame_parity(o, g, f"deep mesh, two devices, frame after move {step}")
    assert g.scene_bvh_stats()["device_builds"] == 2
    o.close(); g.close()


def test_frames_in_flight_with_entities_that_move_lights_that_change_and_a_resize(product_lib):
    """RaytraceEntity.Update moves things between frames (RaytraceEntity.cs:221-232): every scene update waits for the frames in flight
    (the uploads rewrite live device arrays) and the next queued frame sees the new scene.  Same frames as the synchronous calls, bit
    for bit, through object moves (device-side rebuild), a light change and a Resize in the middle."""
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    watch = (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY)

    def run(in_flight):
        scene = _crowd(300, 3)
        rng = np.random.default_rng(9)
        r = RaytraceRenderer(flatten(scene), 160, 90, POSE["fov"], 1)
        r.SetCamera(POSE["pos"], POSE["yaw"], POSE["pitch"])
        out = []
        for step in range(14):
            if step in (3, 4, 9):
                _move(scene, rng, 0.3); r.UpdateObjects(flatten(scene))
            if step == 6:
                scene.Lights[0].Intensity = 250.0; r.UpdateLights(scene.Lights)
            if step == 8:
                r.Resize(128, 72, 1); r.SetCamera(POSE["pos"], POSE["yaw"], POSE["pitch"])
            if in_flight:
                r.RenderAsync()
            else:
                r.TryFlipAndBlit()
            if step in (2, 5, 7, 10, 13):
                out.append([r.read(b) for b in watch])
        r.close()
        return out

    want, got = run(False), run(True)
    for i, (a, b) in enumerate(zip(want, got)):
        for wch, x, y in zip(watch, a, b):
            assert x.shape == y.shape and pu.bits_equal(x, y), (i, wch)
