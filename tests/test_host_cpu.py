"""CPU-only tests of the product's host side: ABI surface, builders (vs the oracle), loaders, tiling.
No compute entry point is called here (there is no GPU in the authoring container)."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as ob
from yetanotherconsolegameengine_amd import abi, mesh_loader, scenes, tiles
from yetanotherconsolegameengine_amd.scene import flatten

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol(product_lib):
    header = (ROOT / "include" / "ycge.h").read_text()
    declared = sorted(set(re.findall(r"\b(ycge_[a-z_]+)\s*\(", header)))
    assert set(declared) == set(abi.EXPORTED_SYMBOLS), (declared, abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(product_lib, name), name


def test_every_other_export_is_listed_as_a_hook(product_lib):
    """include/ycge_hooks.h: what the library exports BESIDE the boundary - test and profiling hooks, host-side builders, the functions that
    cross from the host translation units into the kernel ones - is written down, prototype by prototype, and is all of it: `nm -D` of the
    library = include/ycge.h + include/ycge_hooks.h + the kernel launchers (ycge_launch_*).  (ycge_debug_fail_allocation exists in the
    fault-injection variant only.)"""
    import subprocess
    from yetanotherconsolegameengine_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", str(build.LIB)], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("ycge_")}
    boundary = set(re.findall(r"\b(ycge_[a-z0-9_]+)\s*\(", (ROOT / "include" / "ycge.h").read_text()))
    hooks = set(re.findall(r"\b(ycge_[a-z0-9_]+)\s*\(", (ROOT / "include" / "ycge_hooks.h").read_text())) - {"ycge_debug_fail_allocation"}
    launchers = {s for s in exported if s.startswith("ycge_launch_")}
    assert not (boundary & hooks), boundary & hooks
    assert exported - boundary - launchers == hooks, (sorted(exported - boundary - launchers - hooks), sorted(hooks - exported))
    for name in hooks:
        assert hasattr(product_lib, name), name


def test_ctypes_mirror_matches_header_layout(product_lib):
    product_lib.ycge_abi_sizeof.restype = C.c_size_t
    product_lib.ycge_abi_sizeof.argtypes = [C.c_int32]
    for which, t in enumerate([abi.Vec3, abi.Material, abi.Prim, abi.Mesh, abi.VoxelLookup, abi.Grid, abi.Light, abi.Scene, abi.Config, abi.FrameStats, abi.FlightInfo]):
        assert product_lib.ycge_abi_sizeof(which) == C.sizeof(t), t.__name__
    c = abi.Config()
    assert product_lib.ycge_config_default(C.byref(c)) == 0
    d = abi.default_config()
    for f, _ in abi.Config._fields_:
        a, b = getattr(c, f), getattr(d, f)
        if hasattr(a, "__len__"): a, b = list(a), list(b)
        assert (a == b) or (isinstance(a, float) and np.float32(a) == np.float32(b)), f


def test_create_fails_loudly_without_a_gpu(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = product_lib.ycge_create(C.byref(abi.default_config()), C.byref(ctx))
    assert rc == abi.YCGE_ERR_NO_DEVICE_CODE and not ctx.value
    assert b"no CPU fallback" in product_lib.ycge_last_error(None)
    bad = abi.default_config(); bad.abi_version = 99
    assert product_lib.ycge_create(C.byref(bad), C.byref(ctx)) == abi.YCGE_ERR_INVALID_ARG
    bad = abi.default_config(); bad.diffuse_bounces = 3
    assert product_lib.ycge_create(C.byref(bad), C.byref(ctx)) == abi.YCGE_ERR_UNSUPPORTED
    assert product_lib.ycge_create(None, C.byref(ctx)) == abi.YCGE_ERR_INVALID_ARG
    # MaxMirrorBounces sizes TraceFull's path stack (RaytraceRenderer.cs:450: at most 3 live items with the reference's 2):
    # any other value is refused, never silently truncated
    for v in (0, 1, 3, 7):
        bad = abi.default_config(); bad.max_mirror_bounces = v
        assert product_lib.ycge_create(C.byref(bad), C.byref(ctx)) == abi.YCGE_ERR_UNSUPPORTED
    bad = abi.default_config(); bad.n_devices = 9
    assert product_lib.ycge_create(C.byref(bad), C.byref(ctx)) == abi.YCGE_ERR_INVALID_ARG
    bad = abi.default_config(); bad.n_devices = 2; bad.world_size = 2
    assert product_lib.ycge_create(C.byref(bad), C.byref(ctx)) == abi.YCGE_ERR_INVALID_ARG


def test_scene_upload_argument_checks_fuzzed(product_lib):
    """ycge_validate_scene = the argument checks of ycge_scene_upload (host only): every index a scene carries is poked out of
    range, one at a time and at random, and must be refused with INVALID_ARG (UNSUPPORTED for textured / unknown materials) - never
    accepted, never a crash.  The untouched scene passes."""
    from yetanotherconsolegameengine_amd.scene import Material, Mesh, PointLight, Scene, Sphere, VolumeGrid, vec3
    rng = np.random.default_rng(11)

    def build():
        s = Scene()
        s.Add(Sphere(vec3(0, 1, -3), 1.0, Material(vec3(0.7, 0.3, 0.3))))
        s.Add(Mesh(rng.random((12, 3, 3), dtype=np.float32), Material(vec3(0.2, 0.8, 0.2))))
        cells = np.zeros((4, 4, 4, 2), np.int32); cells[1, 1, 1] = (3, 0); cells[2, 1, 1] = (5, 1)
        s.Add(VolumeGrid(cells, vec3(2, 0, -4), vec3(1, 1, 1), lambda m, meta: Material(vec3(0.5, 0.5, 0.1 * m))))
        s.Lights.append(PointLight(vec3(0, 4, 0), vec3(1, 1, 1), 30.0))
        return flatten(s)

    msg = C.create_string_buffer(256)
    ok = build()
    assert product_lib.ycge_validate_scene(ok.byref(), msg, 256) == abi.YCGE_OK, msg.value
    assert product_lib.ycge_validate_scene(None, msg, 256) == abi.YCGE_ERR_INVALID_ARG
    n_mat = ok.struct.n_materials
    out_of_range = [-1, -7, n_mat, n_mat + 1, 1 << 20, -(1 << 31)]
    pokes = 0
    for trial in range(200):
        f = build()
        kind = trial % 10
        v = int(out_of_range[rng.integers(len(out_of_range))])
        expect = abi.YCGE_ERR_INVALID_ARG
        if kind == 0: f.prims[0].material = v
        elif kind == 1: f.prims[1].ref = int(rng.choice([-1, 1, 99]))
        elif kind == 2: f.prims[2].ref = int(rng.choice([-1, 1, 99]))
        elif kind == 3: f.prims[int(rng.integers(3))].type = int(rng.choice([-1, 11, 1000]))
        elif kind == 4: f.meshes[0].material = v
        elif kind == 5: f.grids[0].lookup[int(rng.integers(2))].material = v
        elif kind == 6: f.struct.n_prims = -1
        elif kind == 7: f.grids[0].nx = int(rng.choice([0, -3]))
        elif kind == 8: f.materials[int(rng.integers(n_mat))].kind = abi.MAT_TEXTURED      # a textured material without a texture to index
        else: f.materials[int(rng.integers(n_mat))].kind = int(rng.choice([3, -1, 77])); expect = abi.YCGE_ERR_UNSUPPORTED
        rc = product_lib.ycge_validate_scene(f.byref(), msg, 256)
        assert rc == expect, (trial, kind, v, rc, msg.value)
        assert msg.value, (trial, kind)
        pokes += 1
    # per-triangle materials are checked triangle by triangle
    f = build()
    tm = np.zeros(12, np.int32); tm[7] = n_mat
    f.meshes[0].tri_material = tm.ctypes.data_as(C.POINTER(C.c_int32))
    assert product_lib.ycge_validate_scene(f.byref(), msg, 256) == abi.YCGE_ERR_INVALID_ARG and b"triangle material" in msg.value
    assert pokes == 200
    # the limits of a voxel grid (include/ycge.h): 2^30 cells, 2^23 bricks across a face - checked before a cell is read
    for dims, what in (((1024, 1024, 1024), b"2^30 cells"), ((1, 8, (1 << 26) + 8), b"2^23 bricks")):
        f = build()
        f.grids[0].nx, f.grids[0].ny, f.grids[0].nz = dims
        rc = product_lib.ycge_validate_scene(f.byref(), msg, 256)
        assert rc == abi.YCGE_ERR_UNSUPPORTED and what in msg.value, (dims, rc, msg.value)


# ---- builders: product (ycge_accel.cpp) vs oracle (orc_scene.cpp), node for node ----------------------------
def _product_tree(lib, bounds, cents, flavour):
    n = len(bounds)
    lib.ycge_host_build_tree.restype = C.c_int
    lib.ycge_host_build_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    nodes = np.zeros(max(1, 2 * n), dtype=ob.NODE_DTYPE); leaf = np.zeros(max(1, n), dtype=np.int32); st = np.zeros(3, dtype=np.int32)
    b = np.ascontiguousarray(bounds, dtype=np.float32); c = np.ascontiguousarray(cents, dtype=np.float32)
    nn = lib.ycge_host_build_tree(b.ctypes.data, c.ctypes.data, n, flavour, nodes.ctypes.data, leaf.ctypes.data, st.ctypes.data)
    return nodes[:nn], leaf[:n], st


def _product_mesh(lib, tris):
    n = len(tris)
    lib.ycge_host_build_mesh.restype = C.c_int
    lib.ycge_host_build_mesh.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    nodes = np.zeros(max(1, 2 * n), dtype=ob.NODE_DTYPE); leaf = np.zeros(max(1, n), dtype=np.int32); st = np.zeros(3, dtype=np.int32)
    nn = lib.ycge_host_build_mesh(t.ctypes.data, n, nodes.ctypes.data, leaf.ctypes.data, st.ctypes.data)
    return nodes[:nn], leaf[:n], st


def _oracle_mesh(tris):
    from yetanotherconsolegameengine_amd.scene import Material, Mesh, Scene, vec3
    s = Scene(); s.Objects.append(Mesh(np.asarray(tris, dtype=np.float32), Material(vec3(1, 1, 1))))
    with ob.OracleRenderer(s, 8, 4) as r:
        return r.accel(abi.ACCEL_MESH_NODES), r.accel(abi.ACCEL_MESH_LEAF_INDEX), r.build_stats(0)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("n_tris", [1, 2, 9, 700])
def test_mesh_arena_is_the_reference_tree_record_for_record(product_lib, n_tris):
    """The device layout of a mesh (emit_mesh_records: GNode / GTriPair records in depth-first order, references in
    32-byte units) against the reference-format tree it is made from: same shape, a node's two boxes are its
    children's boxes, a leaf's records hold its triangles in leaf order (A, e1 = B - A, e2 = C - A, original index),
    an odd leaf ends with an all-zero slot, no byte of the arena is unused."""
    rng = np.random.RandomState(5)
    c = rng.uniform(-1, 1, (n_tris, 1, 3)); tris = (c + rng.normal(scale=0.05, size=(n_tris, 3, 3))).astype(np.float32)
    nodes, leaf, st = _product_mesh(product_lib, tris)
    L = product_lib
    L.ycge_host_mesh_arena.restype = C.c_int
    L.ycge_host_mesh_arena.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    root = np.zeros(1, dtype=np.uint32)
    t9 = np.ascontiguousarray(tris.reshape(-1, 9))
    nbytes = L.ycge_host_mesh_arena(t9.ctypes.data, n_tris, None, 0, root.ctypes.data)
    assert nbytes > 0 and nbytes % 32 == 0
    arena = np.zeros(nbytes, dtype=np.uint8)
    assert L.ycge_host_mesh_arena(t9.ctypes.data, n_tris, arena.ctypes.data, nbytes, root.ctypes.data) == nbytes
    f32, u32, i32 = arena.view(np.float32), arena.view(np.uint32), arena.view(np.int32)
    used = np.zeros(nbytes // 32, dtype=bool)
    seen = []
    KIND_NODE, KIND_LEAF = 2, 3

    def walk(ref, ni):
        kind, pay = int(ref) >> 29, int(ref) & 0x1fffffff
        nd = nodes[ni]
        unit = pay >> 4
        if nd["count"] > 0:
            assert kind == KIND_LEAF and (pay & 15) == nd["count"]
            n_rec = (int(nd["count"]) + 1) // 2
            assert not used[unit:unit + 3 * n_rec].any(); used[unit:unit + 3 * n_rec] = True
            for k in range(int(nd["count"])):
                base = (unit + 3 * (k // 2)) * 8          # float index of the record
                sl = k & 1
                comp = f32[base + sl: base + 18: 2]       # ax ay az e1x e1y e1z e2x e2y e2z
                ti = int(leaf[nd["start"] + k])
                A, B, Cc = tris[ti]
                assert np.array_equal(comp, np.concatenate([A, B - A, Cc - A]).astype(np.float32))
                assert i32[base + 18 + sl] == ti and i32[base + 20 + sl] == 0
                seen.append(ti)
            if nd["count"] & 1:                           # the spare slot: zeros (det = 0 rejects it, the walk masks it anyway)
                base = (unit + 3 * (n_rec - 1)) * 8
                assert not f32[base + 1: base + 18: 2].any() and i32[base + 19] == 0
            return
        assert kind == KIND_NODE and (pay & 15) == 0
        assert not used[unit:unit + 2].any(); used[unit:unit + 2] = True
        g = f32[unit * 8: unit * 8 + 16]
        lch, rch = nodes[nd["left"]], nodes[nd["right"]]
        # plane order (x y)(z Z)(X Y) per child, lower case = min
        assert np.array_equal(g[:12], np.float32([lch["min"][0], lch["min"][1], lch["min"][2], lch["max"][2], lch["max"][0], lch["max"][1],
                                                 rch["min"][0], rch["min"][1], rch["min"][2], rch["max"][2], rch["max"][0], rch["max"][1]]))
        lref, rref = u32[unit * 8 + 12], u32[unit * 8 + 13]
        assert (int(lref) & 0x1fffffff) >> 4 == unit + 2  # depth-first: the left child's records follow the node
        walk(lref, nd["left"]); walk(rref, nd["right"])

    walk(root[0], int(st[0]))
    assert used.all() and sorted(seen) == list(range(n_tris))


@pytest.mark.parametrize("case", ["random", "grid_ties", "all_equal", "collinear", "tiny"])
def test_mesh_builder_matches_oracle(product_lib, case):
    rng = np.random.RandomState(11)
    if case == "random":
        c = rng.uniform(-1, 1, (3000, 1, 3)); tris = (c + rng.normal(scale=0.02, size=(3000, 3, 3))).astype(np.float32)
    elif case == "grid_ties":         # many identical centroids per axis -> degenerate bins and Array.Sort fallbacks
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3), indexing="ij"), -1).reshape(-1, 1, 3).astype(np.float32)
        tris = (g + np.float32([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0]])).astype(np.float32)
    elif case == "all_equal":         # every triangle identical: zero centroid extent on all axes
        tris = np.tile(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), (100, 1, 1))
    elif case == "collinear":         # centroids on one axis only
        x = np.linspace(0, 10, 257, dtype=np.float32)
        tris = np.stack([np.stack([x, 0 * x, 0 * x], -1), np.stack([x + 0.01, 0 * x, 0 * x], -1), np.stack([x, 0 * x + 0.01, 0 * x], -1)], 1)
    else:
        tris = rng.uniform(-1, 1, (9, 3, 3)).astype(np.float32)
    pn, pl, ps = _product_mesh(product_lib, tris)
    on, ol, os_ = _oracle_mesh(tris)
    assert _same(pn, on) and np.array_equal(pl, ol)
    assert ps[2] == os_["mesh_sort_fallbacks"] and ps[1] == os_["mesh_max_depth"]
    if case == "all_equal":
        assert ps[2] > 0                      # the Array.Sort path really ran (zero centroid extent)


def test_mesh_builder_matches_oracle_on_bunny_and_knot(product_lib):
    for tris in (scenes.BuildBunnyScene().Objects[1].Triangles, scenes.BuildDragonStandInScene(132, 33).Objects[1].Triangles):
        pn, pl, ps = _product_mesh(product_lib, tris)
        on, ol, os_ = _oracle_mesh(tris)
        assert _same(pn, on) and np.array_equal(pl, ol) and ps[1] <= 64


def test_scene_builder_matches_oracle_including_partition_quirk(product_lib):
    """Scene-level BVH: partition origin/extent come from the first/last ITEM (BVH.cs:394-396)."""
    from yetanotherconsolegameengine_amd.scene import Material, Scene, Sphere, vec3
    rng = np.random.RandomState(5)
    for n in (1, 4, 5, 9, 40, 333, 2000):
        s = Scene()
        for _ in range(n):
            s.Add(Sphere(vec3(*rng.uniform(-20, 20, 3)), float(rng.uniform(0.1, 2.0)), Material(vec3(1, 1, 1))))
        with ob.OracleRenderer(s, 8, 4) as r:
            on, ol = r.accel(abi.ACCEL_SCENE_NODES), r.accel(abi.ACCEL_SCENE_LEAF_INDEX)
        cen = np.float32([o.Center for o in s.Objects]); rad = np.float32([[o.Radius] for o in s.Objects])
        mn, mx = cen - rad, cen + rad
        bounds = np.concatenate([mn, mx], 1); cents = np.float32(0.5) * (mn + mx)
        pn, pl, st = _product_tree(product_lib, bounds, cents, 0)
        assert _same(pn, on) and np.array_equal(pl, ol), n
    # chunk-lattice boxes (the voxel world's scene BVH): the quirk pushes nodes onto the Array.Sort path
    from yetanotherconsolegameengine_amd.scene import Box, Solid
    s = Scene(); white = Solid(vec3(1, 1, 1))
    for cx in range(5):
        for cy in range(3):
            for cz in range(5):
                if (cx + cy + cz) % 7 != 3:
                    s.Add(Box(vec3(32 * cx, 32 * cy, 32 * cz), vec3(32 * cx + 32, 32 * cy + 32, 32 * cz + 32), white, 0.0, 0.0))
    with ob.OracleRenderer(s, 8, 4) as r:
        on, ol, ost = r.accel(abi.ACCEL_SCENE_NODES), r.accel(abi.ACCEL_SCENE_LEAF_INDEX), r.build_stats()
    mn = np.float32([o.Min for o in s.Objects]); mx = np.float32([o.Max for o in s.Objects])
    pn, pl, st = _product_tree(product_lib, np.concatenate([mn, mx], 1), np.float32(0.5) * (mn + mx), 0)
    assert _same(pn, on) and np.array_equal(pl, ol)
    assert st[2] == ost["scene_sort_fallbacks"] and st[2] > 0


# ---- loaders ------------------------------------------------------------------------------------------------------
def test_obj_subset_parser():
    text = """# comment
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0

f 1/1/1 2/2/2 3/3/3 4/4/4
v 0 0 1
f -1 -2 -3
f 1// 2// 5//
""".splitlines()
    pos, faces = mesh_loader.parse_obj(text)
    assert pos.shape == (5, 3) and pos.dtype == np.float32
    assert faces.tolist() == [[0, 1, 2], [0, 2, 3], [4, 3, 2], [0, 1, 4]]           # fan triangulation; negative = from the end
    tris = mesh_loader.from_obj_arrays(pos, faces, scale=2.0, translate=(1, 0, 0), normalize=True, target_size=1.0)
    assert tris.shape == (4, 3, 3)
    p = tris.reshape(-1, 3)
    assert np.isclose(p[:, 0].min(), 0.0) and np.isclose(p[:, 0].max(), 2.0)       # (x - 0.5) * 1 * 2 + 1


def test_bunny_fixture_and_auto_ground():
    pos, faces = scenes.load_bunny_arrays()
    assert pos.shape == (35947, 3) and faces.shape == (69451, 3)
    tris = mesh_loader.add_mesh_auto_ground(pos, faces, 1.0, (0.0, 0.5, 1.0))
    p = tris.reshape(-1, 3)
    ext = p.max(0) - p.min(0)
    assert abs(float(ext.max()) - 1.0) < 1e-5                                        # normalised to max extent 1
    # (the two normalisations of the reference differ - bbox centre vs centroid - so the mesh floats a little off y = 0.51)
    assert 0.3 < float(p[:, 1].min()) < 0.6 and abs(float(p[:, 2].mean()) - 1.0) < 0.2


def test_config_scenes_flatten():
    for n in (1, 2):
        sc, w, h, ss, pose = scenes.config_scene(n)
        f = flatten(sc)
        assert f.struct.n_prims == len(sc.Objects) and f.struct.n_lights == len(sc.Lights)
    sc, w, h, ss, pose = scenes.config_scene(5, small=True)
    f = flatten(sc)
    assert f.struct.n_grids == len(sc.Objects) > 4 and f.struct.is_volume_scene == 1
    g = f.grids[0]
    assert (g.nx, g.ny, g.nz) == (32, 32, 32) and g.n_lookup >= 1 and g.wireframe == 1


# ---- multi-GPU tile layout: 2 processes, gloo ----------------------------------------------------------------------------
def test_tile_layout_roundtrip_single_process():
    rng = np.random.RandomState(0)
    for (W, H, world) in ((80, 90, 1), (80, 90, 3), (1920, 1080, 8), (70, 50, 4)):
        frame = rng.rand(H, W, 11).astype(np.float32)
        slabs = np.concatenate([tiles.pack_slab(frame, r, world) for r in range(world)])
        assert slabs.size == world * tiles.slab_floats(world, tiles.tile_grid(W, H)[2])
        assert np.array_equal(tiles.unpermute(slabs, W, H, world), frame)


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from yetanotherconsolegameengine_amd import tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 200, 90
yy, xx = np.mgrid[0:H, 0:W]
FLOATS = int(sys.argv[2])                       # 11, or 8 for lean slabs (config.slab_albedo = 0)
frame = np.stack([(xx * 1000 + yy + c * 0.125).astype(np.float32) for c in range(FLOATS)], -1)  # what the full frame must be
mine = np.full_like(frame, np.nan)                                                              # this rank only "traces" its tiles
tx, ty, n = tiles.tile_grid(W, H)
for tid in tiles.owned_tiles(rank, world, n):
    x0, y0 = (tid % tx) * 32, (tid // tx) * 8
    mine[y0:y0 + 8, x0:x0 + 32] = frame[y0:y0 + 8, x0:x0 + 32]
slab = torch.from_numpy(tiles.pack_slab(mine, rank, world))
gathered = torch.empty(world * slab.numel(), dtype=torch.float32)
dist.all_gather_into_tensor(gathered, slab)              # the one collective of the path
full = tiles.unpermute(gathered.numpy(), W, H, world, FLOATS)
assert np.array_equal(full, frame), "rank %d: reassembled frame differs" % rank
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("floats", [11, 8])
def test_two_process_all_gather_reassembles_the_frame(tmp_path, floats):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29533 + floats), str(script), str(ROOT), str(floats)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


# ---- tile-resident form: halo lists (TAA's 3x3 window across tile borders), host C++ against tiles.py, then 2 processes over gloo -----
@pytest.mark.parametrize("W,H,world", [(200, 90, 2), (200, 90, 3), (1920, 1080, 8), (70, 50, 4), (33, 9, 2), (64, 16, 5)])
def test_halo_lists_of_the_library_are_the_layout_module_s(product_lib, W, H, world):
    """ycge_host_halo_layout (csrc/ycge_host.cpp: halo_layout, pure host code) and tiles.halo_lists state the same exchange: for every
    rank the same counts per peer and the same pixel lists; what rank q sends to rank r is what r expects from q, record for record;
    and the ring is complete - every pixel of a rank's tiles finds its whole (image-clamped) 3x3 window among the rank's own pixels and
    the pixels it receives (TemporalBlendWithClamp, RaytraceRenderer.cs:218, 341-352)."""
    fn = product_lib.ycge_host_halo_layout
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    tx, ty, n = tiles.tile_grid(W, H)
    lists = []
    for rank in range(world):
        sc = np.zeros(world, np.int64); rc = np.zeros(world, np.int64)
        cap = n * 84 + 16
        spx = np.zeros(cap, np.uint32); rpx = np.zeros(cap, np.uint32)
        assert fn(W, H, rank, world, sc.ctypes.data, rc.ctypes.data, spx.ctypes.data, rpx.ctypes.data, cap) == 0
        send, recv = tiles.halo_lists(rank, world, W, H)
        assert [len(v) for v in send] == sc.tolist() and [len(v) for v in recv] == rc.tolist(), rank
        assert np.array_equal(np.concatenate(send) if sc.sum() else np.zeros(0, np.int64), spx[:sc.sum()].astype(np.int64))
        assert np.array_equal(np.concatenate(recv) if rc.sum() else np.zeros(0, np.int64), rpx[:rc.sum()].astype(np.int64))
        lists.append((send, recv))
    for q in range(world):
        for r in range(world):
            if q != r:
                assert np.array_equal(lists[q][0][r], lists[r][1][q]), (q, r)          # q's segment for r IS r's segment from q
    # completeness of the ring, rank 0 (and the last rank): own pixels + received pixels cover every 3x3 window
    for rank in {0, world - 1}:
        have = np.zeros(W * H, bool)
        for t in tiles.owned_tiles(rank, world, n):
            x0, y0 = (t % tx) * 32, (t // tx) * 8
            yy, xx = np.mgrid[y0:min(y0 + 8, H), x0:min(x0 + 32, W)]
            have[(xx + yy * W).ravel()] = True
        own = have.copy()
        for v in lists[rank][1]:
            have[v] = True
        ys, xs = np.nonzero(own.reshape(H, W))
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                assert have[np.clip(xs + dx, 0, W - 1) + np.clip(ys + dy, 0, H - 1) * W].all(), (rank, dx, dy)


_HALO_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from yetanotherconsolegameengine_amd import tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 200, 90
tx, ty, n = tiles.tile_grid(W, H)
yy, xx = np.mgrid[0:H, 0:W]
frame = np.stack([(xx * 1000 + yy + c * 0.125).astype(np.float32) for c in range(4)], -1).reshape(-1, 4)     # {hdr rgb, sky} of the whole frame
mine = np.full_like(frame, np.nan)                       # this rank "traces" its tiles only
for tid in tiles.owned_tiles(rank, world, n):
    x0, y0 = (tid % tx) * 32, (tid // tx) * 8
    ys, xs = np.mgrid[y0:min(y0 + 8, H), x0:min(x0 + 32, W)]
    mine[(xs + ys * W).ravel()] = frame[(xs + ys * W).ravel()]
send, recv = tiles.halo_lists(rank, world, W, H)
send_buf = torch.from_numpy(np.concatenate([mine[v] for v in send]).astype(np.float32).reshape(-1))
recv_buf = torch.empty(sum(len(v) for v in recv) * 4, dtype=torch.float32)
# the one exchange of the tile-resident form: all_to_all_single with ycge_halo_counts' split sizes (records of 4 floats)
dist.all_to_all_single(recv_buf, send_buf, output_split_sizes=[len(v) * 4 for v in recv], input_split_sizes=[len(v) * 4 for v in send])
got = recv_buf.numpy().reshape(-1, 4)
mine[np.concatenate(recv)] = got                         # k_scatter_halo
# every own pixel now finds its (image-clamped) 3x3 window, with the values the full frame holds
for tid in tiles.owned_tiles(rank, world, n):
    x0, y0 = (tid % tx) * 32, (tid // tx) * 8
    ys, xs = np.mgrid[y0:min(y0 + 8, H), x0:min(x0 + 32, W)]
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            j = (np.clip(xs + dx, 0, W - 1) + np.clip(ys + dy, 0, H - 1) * W).ravel()
            assert np.array_equal(mine[j], frame[j]), "rank %d tile %d tap (%d, %d)" % (rank, tid, dx, dy)
# the history slabs: 12 bytes per pixel of the own tiles, gathered to whoever shows the frame
hist = frame[:, :3].reshape(H, W, 3)
slab = torch.from_numpy(tiles.pack_slab(np.where(np.isnan(mine[:, :3]), 0, mine[:, :3]).reshape(H, W, 3).astype(np.float32), rank, world))
allh = torch.empty(world * slab.numel(), dtype=torch.float32)
dist.all_gather_into_tensor(allh, slab)
assert np.array_equal(tiles.unpermute(allh.numpy(), W, H, world, 3), hist)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_and_history_gather_over_gloo(tmp_path, world):
    """The tile-resident form's two exchanges between `world` processes (gloo here, RCCL on the GPUs): every rank ends up with the {hdr, sky}
    ring its TAA window needs, and the gathered history slabs reassemble the frame."""
    script = tmp_path / "halo_worker.py"
    script.write_text(_HALO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29571 + world), str(script), str(ROOT)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("w,h,step", [(37, 23, 2), (64, 48, 2), (80, 90, 2), (9, 7, 2), (33, 40, 8), (5, 5, 2), (1, 13, 2)])
def test_inplace_atrous_schedule_respects_scan_order(product_lib, w, h, step):
    """Levels of the in-place A-trous iteration (RaytraceRenderer.cs:648-650,718): for every stencil pair the
    pixel that comes first in scan order must sit on a strictly earlier level, whichever of the two reads the
    other - then a level-by-level evaluation sees NEW values of earlier pixels and OLD values of later ones,
    exactly like the reference's serial scan."""
    L = product_lib
    L.ycge_host_inplace_schedule.restype = C.c_int
    L.ycge_host_inplace_schedule.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    px = np.zeros(w * h, np.uint32); off = np.zeros(w * h + 2, np.uint32)
    n = L.ycge_host_inplace_schedule(w, h, step, px.ctypes.data, off.ctypes.data, off.size)
    assert n > 0
    off = off[:n + 1]
    assert off[0] == 0 and off[-1] == w * h and np.all(np.diff(off.astype(np.int64)) >= 0)
    assert np.array_equal(np.sort(px), np.arange(w * h, dtype=np.uint32))          # every pixel exactly once
    level = np.zeros(w * h, np.int64)
    for l in range(n):
        level[px[off[l]:off[l + 1]]] = l
    ys, xs = np.mgrid[0:h, 0:w]
    p = (xs + ys * w).ravel()
    for ky in range(-2, 3):
        for kx in range(-2, 3):
            sx = np.clip(xs + kx * step, 0, w - 1); sy = np.clip(ys + ky * step, 0, h - 1)
            q = (sx + sy * w).ravel()
            earlier = q < p
            later = q > p
            assert np.all(level[q[earlier]] < level[p[earlier]])
            assert np.all(level[q[later]] > level[p[later]])
            # the banded pipeline of k_atrous_band: launch = band + level // K; a band runs its levels in order inside one
            # workgroup, different bands only meet across kernel boundaries
            for rows_per_band, K in ((16, 8), (4, 3)):
                band = (np.arange(w * h) // w) // rows_per_band
                launch = band + level // K
                same = band[q] == band[p]
                e_same, e_diff = earlier & same, earlier & ~same
                assert np.all((launch[q[e_same]] < launch[p[e_same]]) | ((launch[q[e_same]] == launch[p[e_same]]) & (level[q[e_same]] < level[p[e_same]])))
                assert np.all(launch[q[e_diff]] < launch[p[e_diff]])


@pytest.mark.parametrize("w,h,step,rows", [(37, 23, 2, 16), (80, 90, 2, 16), (64, 48, 2, 8), (33, 40, 8, 16), (5, 5, 2, 4)])
def test_inplace_atrous_band_pass_lists(product_lib, w, h, step, rows):
    """What k_atrous_band reads: per band, the level schedule restricted to the band's rows, every level padded to whole
    passes of 32 entries (x | y << 16, 0xffffffff = no pixel), pass offsets per (band, level)."""
    L = product_lib
    L.ycge_host_inplace_schedule.restype = C.c_int
    L.ycge_host_inplace_schedule.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    px = np.zeros(w * h, np.uint32); off = np.zeros(w * h + 2, np.uint32)
    n_levels = L.ycge_host_inplace_schedule(w, h, step, px.ctypes.data, off.ctypes.data, off.size)
    level = np.zeros(w * h, np.int64)
    for l in range(n_levels):
        level[px[off[l]:off[l + 1]]] = l
    L.ycge_host_inplace_bands.restype = C.c_int
    L.ycge_host_inplace_bands.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    info = np.zeros(3, np.int32)
    n_pass = L.ycge_host_inplace_bands(w, h, step, rows, None, 0, None, 0, info.ctypes.data)
    levels, n_bands, max_level_pixels = (int(v) for v in info)
    assert levels == n_levels and n_bands == (h + rows - 1) // rows and n_pass > 0
    ent = np.zeros(n_pass * 32, np.uint32); boff = np.zeros(n_bands * (levels + 1), np.uint32)
    assert L.ycge_host_inplace_bands(w, h, step, rows, ent.ctypes.data, ent.size, boff.ctypes.data, boff.size, info.ctypes.data) == n_pass
    boff = boff.reshape(n_bands, levels + 1).astype(np.int64)
    assert boff[0, 0] == 0 and boff[-1, -1] == n_pass and np.all(np.diff(boff.ravel()) >= 0)        # bands follow one another, levels in order
    seen = np.zeros(w * h, bool)
    widest = 0
    for b in range(n_bands):
        for t in range(levels):
            e = ent[boff[b, t] * 32: boff[b, t + 1] * 32]
            real = e[e != 0xffffffff]
            x, y = (real & 0xffff).astype(np.int64), (real >> 16).astype(np.int64)
            assert np.all(y // rows == b) and np.all(x < w) and np.all(y < h)
            p = x + y * w
            assert np.all(level[p] == t) and not seen[p].any()
            seen[p] = True
            assert len(e) - len(real) < 32 and (len(real) > 0) == (len(e) > 0)      # padding only completes the last pass of a level
            assert np.all(e[:len(real)] != 0xffffffff)                               # ... and sits at its end
            widest = max(widest, len(e))
    assert seen.all() and widest == max_level_pixels


@pytest.mark.parametrize("w,h,step,rows,K", [(1920, 1080, 2, 8, 8), (640, 360, 2, 8, 8), (257, 131, 2, 8, 8), (320, 200, 4, 8, 8), (97, 61, 2, 16, 4), (33, 40, 8, 16, 8)])
def test_inplace_atrous_window_is_collision_free(product_lib, w, h, step, rows, K):
    """k_atrous_band's window form keeps what one launch writes at (row in band) * WX + (x mod WX): the width the host picks
    must separate every two pixels of one launch (levels [K g, K g + K) of one band), re-checked here from the pass lists."""
    L = product_lib
    L.ycge_host_band_window_width.restype = C.c_int
    L.ycge_host_band_window_width.argtypes = [C.c_int32] * 6
    rows_eff = max(rows, 2 * step)
    wx = L.ycge_host_band_window_width(w, h, step, rows_eff, K, 32)
    assert wx >= 64 and wx & (wx - 1) == 0 and wx * rows_eff <= 2048, wx
    if (w, h) == (1920, 1080):
        assert wx == 64 and L.ycge_host_band_window_width(w, h, step, rows_eff, K, 16) == 64       # the shipped defaults
    L.ycge_host_inplace_bands.restype = C.c_int
    L.ycge_host_inplace_bands.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    info = np.zeros(3, np.int32)
    n_pass = L.ycge_host_inplace_bands(w, h, step, rows_eff, None, 0, None, 0, info.ctypes.data)
    levels, n_bands, _ = (int(v) for v in info)
    ent = np.zeros(n_pass * 32, np.uint32); boff = np.zeros(n_bands * (levels + 1), np.uint32)
    L.ycge_host_inplace_bands(w, h, step, rows_eff, ent.ctypes.data, ent.size, boff.ctypes.data, boff.size, info.ctypes.data)
    boff = boff.reshape(n_bands, levels + 1).astype(np.int64)
    for b in range(n_bands):
        for t0 in range(0, levels, K):
            e = ent[boff[b, t0] * 32: boff[b, min(t0 + K, levels)] * 32]
            e = e[e != 0xffffffff].astype(np.int64)
            slot = ((e >> 16) - b * rows_eff) * wx + ((e & 0xffff) & (wx - 1))
            assert len(np.unique(slot)) == len(slot) and (slot >= 0).all() and (slot < 2048).all()
    assert L.ycge_host_band_window_width(0, h, step, rows_eff, K, 32) < 0 and L.ycge_host_band_window_width(w, h, step, rows_eff, K, 7) < 0


@pytest.mark.parametrize("w,h", [(1920, 1080), (640, 360), (131, 90), (64, 27), (40, 24)])
def test_row_parity_band_layout_of_the_persistent_atrous(product_lib, w, h):
    """The persistent in-place A-trous launch cuts the grid (step 2) into the first four rows, two chains of half-bands (the even / the
    odd rows of every 8-row stretch) and the last four rows.  Checked here from the stencil itself: every row is in exactly one band;
    whatever row a pixel's taps land on (clamped) is in its own band, in a band it waits for (above) or in a band that waits for it
    (below); nobody waits for more than two bands; no band has more than 8 pixels in a level (its workgroup keeps 8 groups a set)."""
    L = product_lib
    L.ycge_host_split_bands.restype = C.c_int
    L.ycge_host_split_bands.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    row_band = np.full(h, -1, np.int32); desc = np.full(8 * (h + 4), -7, np.int32); max_px = np.zeros(h + 4, np.int32)
    nb = L.ycge_host_split_bands(w, h, 2, row_band.ctypes.data, desc.ctypes.data, desc.size, max_px.ctypes.data)
    assert nb == 2 + 2 * ((h - 8 + 7) // 8)
    d = desc[:8 * nb].reshape(nb, 8)
    rows_of = [[int(y0) + k * int(st) for k in range(int(n))] for y0, n, st in d[:, :3]]
    assert sorted(y for r in rows_of for y in r) == list(range(h))                      # a partition of the rows
    assert all(row_band[y] == b for b, r in enumerate(rows_of) for y in r)
    assert rows_of[0] == [0, 1, 2, 3] and rows_of[-1] == list(range(h - 4, h))
    for b in range(1, nb - 1):
        assert d[b, 2] == 2 and len({y & 1 for y in rows_of[b]}) <= 1                    # a half-band: one row parity
    for y in range(h):
        a = int(row_band[y])
        ups, dns = {int(v) for v in d[a, 4:6] if v >= 0}, {int(v) for v in d[a, 6:8] if v >= 0}
        for k in (-2, -1, 0, 1, 2):
            q = int(row_band[min(max(y + 2 * k, 0), h - 1)])
            if q != a:
                assert q in (ups if min(max(y + 2 * k, 0), h - 1) < y else dns), (y, k, a, q)
    for b in range(nb):                                                                  # the two lists agree
        for u in d[b, 4:6]:
            if u >= 0: assert b in d[u, 6:8]
        for v in d[b, 6:8]:
            if v >= 0: assert b in d[v, 4:6]
    assert (d[:, 3] == 8).all() and max_px[:nb].max() <= 8, max_px[:nb].max()
    assert L.ycge_host_split_bands(w, h, 4, row_band.ctypes.data, desc.ctypes.data, desc.size, None) == 0       # the layout is step 2's
    assert L.ycge_host_split_bands(w, 16, 2, row_band.ctypes.data, desc.ctypes.data, desc.size, None) == 0      # too few rows


def test_vg01_world_file_roundtrip_and_errors(tmp_path):
    """the VG01 world file (WorldManager.cs:612-629 writer, :399-441 reader) and its error behaviour."""
    import struct
    from yetanotherconsolegameengine_amd import world_file as wf
    rng = np.random.default_rng(7)
    cells = rng.integers(0, 9, size=(5, 7, 3, 2), dtype=np.int32)
    p = tmp_path / "w.vg"
    wf.write_vg01(p, cells)
    raw = p.read_bytes()
    assert raw[:4] == b"VG01" and struct.unpack("<iii", raw[4:16]) == (5, 7, 3) and len(raw) == 16 + 5 * 7 * 3 * 8
    # z fastest, then y, then x; matId before metaId
    assert struct.unpack("<ii", raw[16:24]) == (int(cells[0, 0, 0, 0]), int(cells[0, 0, 0, 1]))
    assert struct.unpack("<ii", raw[24:32]) == (int(cells[0, 0, 1, 0]), int(cells[0, 0, 1, 1]))
    assert np.array_equal(wf.read_vg01(p), cells)
    (tmp_path / "bad.vg").write_bytes(b"VG02" + raw[4:])
    with pytest.raises(ValueError, match="VG01"):
        wf.read_vg01(tmp_path / "bad.vg")
    (tmp_path / "dims.vg").write_bytes(b"VG01" + struct.pack("<iii", 0, 1, 1))
    with pytest.raises(ValueError, match="dimensions"):
        wf.read_vg01(tmp_path / "dims.vg")
    (tmp_path / "short.vg").write_bytes(raw[:-4])
    with pytest.raises(EOFError):
        wf.read_vg01(tmp_path / "short.vg")
    with pytest.raises(FileNotFoundError):
        wf.read_vg01(tmp_path / "nope.vg")


def test_chunk_streaming_set_and_attach_order():
    """BuildDesiredSet (:372-397) + AttachChunkFromPreloaded (:696-731): key order, clipping, air skipping."""
    from yetanotherconsolegameengine_amd import world_file as wf
    from yetanotherconsolegameengine_amd.scene import Scene
    keys = wf.build_desired_set((5.0, 0.0, -40.0), (-64.0, 0.0, -64.0), (1.0, 1.0, 1.0), 32, 1, 2)
    # centre column: floor((5+64)/32) = 2, floor((-40+64)/32) = 0 -> cx 1..3, cz -1..1, every cy; cx outermost, cy innermost
    assert keys[0] == (1, 0, -1) and keys[1] == (1, 1, -1) and keys[2] == (1, 0, 0) and keys[-1] == (3, 1, 1) and len(keys) == 18
    world = np.zeros((70, 40, 64, 2), np.int32)
    world[:, :20, :, 0] = 1                       # ground slab: chunks with cy = 1 are air
    world[69, 39, 63, 0] = 2                      # one voxel in the clipped far corner chunk (2, 1, 1)
    sc = Scene()
    loaded = {}
    added = wf.attach_view(sc, world, (5.0, 0.0, -40.0), (-64.0, 0.0, -64.0), (1.0, 1.0, 1.0), 32, 1, lambda m, t: None, loaded=loaded)
    assert added == [(1, 0, 0), (1, 0, 1), (2, 0, 0), (2, 0, 1), (2, 1, 1)]      # negative cz and cx = 3 lie outside the 70-wide world
    assert sc.Objects[4].Cells.shape == (6, 8, 32, 2)                            # clipped: 70 - 64, 40 - 32
    assert sc.Objects[2].MinCorner == (0.0, 0.0, -64.0)
    assert wf.attach_view(sc, world, (5.0, 0.0, -40.0), (-64.0, 0.0, -64.0), (1.0, 1.0, 1.0), 32, 1, lambda m, t: None, loaded=loaded) == []


def test_oracle_parallel_taa_equals_the_serial_loops():
    """bench.py's cpu_baseline reports the reference's serial TAA and the same loop in row bands: identical pixels."""
    sc, w, h, ss, pose = scenes.config_scene(2)
    a = ob.OracleRenderer(sc, 96, 27, 1, pose)
    b = ob.OracleRenderer(sc, 96, 27, 1, pose)
    b.set_taa_threads(5)
    for f in range(3):
        a.render(stages=1, threads=2); b.render(stages=1, threads=2)
        for which in (abi.BUF_TAA_HISTORY, abi.BUF_PREV_NORMAL, abi.BUF_PREV_DEPTH, abi.BUF_PREV_SKY):
            assert np.array_equal(a.read(which).view(np.uint8), b.read(which).view(np.uint8)), (f, which)
    a.close(); b.close()


def test_reference_obj_assets_parse_to_their_triangle_counts():
    """MeshLoader.cs:23-55 on the reference's own assets (CPU container only: /root/reference does not travel): cow.obj has
    n-gons that fan out to 5 804 triangles, teapot.obj 6 320, the bunny 69 451 = the committed fixture, vertex for vertex."""
    assets = Path("/root/reference/ConsoleGame/assets")
    if not assets.is_dir():
        pytest.skip("reference assets not present (GPU box)")
    for name, n_tri in (("cow.obj", 5804), ("teapot.obj", 6320)):
        pos, faces = mesh_loader.load_obj(assets / name)
        assert faces.shape == (n_tri, 3), (name, faces.shape)
        assert faces.min() >= 0 and faces.max() < len(pos)
        tris = mesh_loader.from_obj_arrays(pos, faces)
        ext = tris.reshape(-1, 3).max(0) - tris.reshape(-1, 3).min(0)
        assert abs(float(ext.max()) - 1.0) < 1e-5                       # normalize: max extent = targetSize
    pos, faces = mesh_loader.load_obj(assets / "stanford-bunny.obj")
    fpos, ffaces = scenes.load_bunny_arrays()
    assert np.array_equal(pos, fpos) and np.array_equal(faces, ffaces)


def test_desired_chunk_set_known_answer():
    """BuildDesiredSet (WorldManager.cs:372-397) against a count and membership worked out by hand, not by the function's own
    logic: view distance 8 around any column is 17 x 17 columns, times 8 vertical chunks = 2 312 keys (BASELINE config 5);
    the column of a camera at x = -0.5 with WorldMin.x = 0 is floor(-0.5 / 32) = -1, not 0 (floor, not truncation)."""
    from yetanotherconsolegameengine_amd import world_file as wf
    keys = wf.build_desired_set((100.0, 50.0, 100.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 32, 8, 8)
    assert len(keys) == 17 * 17 * 8 == 2312 and len(set(keys)) == 2312
    xs = sorted({k[0] for k in keys}); ys = sorted({k[1] for k in keys}); zs = sorted({k[2] for k in keys})
    assert xs == list(range(3 - 8, 3 + 9)) and zs == xs and ys == list(range(8))          # 100 / 32 = 3.125 -> column 3
    neg = wf.build_desired_set((-0.5, 0.0, -0.5), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 32, 0, 1)
    assert neg == [(-1, 0, -1)]
    half = wf.build_desired_set((10.0, 0.0, 10.0), (0.0, 0.0, 0.0), (0.5, 0.5, 0.5), 16, 1, 2)       # 8-unit chunks: column 1
    assert len(half) == 3 * 3 * 2 and half[0] == (0, 0, 0) and half[-1] == (2, 1, 2)


def test_sdr_arrays_of_the_python_mirror_live_in_the_librarys_page_locked_memory():
    """renderer.py keeps its SDR frames - the wrapper's one buffer and the ring of the frames in flight - in memory ycge_alloc_host_buffer
    hands out (hipHostMalloc) and gives every block back exactly once; it never registers numpy memory (hipHostRegister on process heap
    was behind the GPU memory faults of round 4: csrc/ycge_host.cpp, copy_out).  Checked with a stand-in library: no GPU needed."""
    import ctypes as C
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer

    live, freed, blocks = {}, [], []

    class _Lib:
        @staticmethod
        def ycge_alloc_host_buffer(n, out):
            buf = (C.c_uint8 * n)()          # zeroed, like the library's
            blocks.append(buf)
            out._obj.value = C.addressof(buf)
            live[C.addressof(buf)] = n
            return 0

        @staticmethod
        def ycge_free_host_buffer(p):
            assert p.value in live, "freed twice or never allocated"
            freed.append(p.value); del live[p.value]
            return 0

        @staticmethod
        def ycge_wait(ctx):
            return 0

        def __getattr__(self, name):
            raise AssertionError("the mirror called " + name + ": SDR arrays must not be registered / pinned by hand")

    class _Self:
        L = _Lib(); ctx = None; fbH, fbW = 27, 96

    me = _Self()
    for f in ("_page_locked_zeros", "_free_page_locked", "_sdr_buffer", "_drop_sdr_buffer", "_drop_sdr_ring"):
        setattr(_Self, f, getattr(RaytraceRenderer, f))
    a = me._sdr_buffer()
    assert a.shape == (27, 96, 2, 3) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and not a.any() and a.ctypes.data in live
    a[...] = 1.0
    assert me._sdr_buffer() is a and len(live) == 1                     # one buffer for the life of the renderer
    me.fbH, me.fbW = 45, 160
    b = me._sdr_buffer()                                                # another console size: a new block comes; the old one goes back ...
    assert b.shape == (45, 160, 2, 3) and len(live) == 2 and not freed and float(a[0, 0, 0, 0]) == 1.0     # ... not under a caller who still holds the array
    import gc
    del a; gc.collect()                                                 # ... but when its last view dies (round 6: the array owns its pages)
    assert len(live) == 1 and len(freed) == 1 and b.ctypes.data in live
    ring = me.__dict__.setdefault("_sdr_ring", {})
    for k in range(3):
        ring[k] = me._page_locked_zeros((45, 160, 2, 3))
    assert len(live) == 4
    me._drop_sdr_ring(keep_shape=(45, 160, 2, 3))
    assert len(live) == 4 and len(ring) == 3                            # same size: kept
    me._drop_sdr_ring(keep_shape=(1, 1, 2, 3)); gc.collect()
    assert len(live) == 1 and not ring
    me._drop_sdr_buffer(); me._drop_sdr_buffer()                        # idempotent
    del b; gc.collect()
    assert not live and len(freed) == 5 and len(set(freed)) == 5


def test_pin_refuses_what_is_not_whole_pages_of_its_own():
    """ycge_pin_host_buffer (hipHostRegister is page-granular; ABI 8): a range that does not start on a page boundary or is not a whole number
    of pages is refused with YCGE_ERR_INVALID_ARG before the runtime is asked anything - so this runs without a GPU.  (The accepted case, and
    ycge_alloc_host_buffer, need the runtime: tests/test_gpu_parity.py.)"""
    import ctypes as C
    L = abi.load_library()
    page = L.ycge_host_page_size()
    assert page >= 4096 and page & (page - 1) == 0
    raw = np.zeros(4 * page, dtype=np.uint8)
    base = raw.ctypes.data + (-raw.ctypes.data) % page
    assert L.ycge_pin_host_buffer(C.c_void_p(base + 16), page) == abi.YCGE_ERR_INVALID_ARG            # not on a page boundary
    assert L.ycge_pin_host_buffer(C.c_void_p(base), page + 24) == abi.YCGE_ERR_INVALID_ARG             # not whole pages
    assert L.ycge_pin_host_buffer(C.c_void_p(base), 0) == abi.YCGE_ERR_INVALID_ARG
    assert L.ycge_pin_host_buffer(None, page) == abi.YCGE_ERR_INVALID_ARG
    assert L.ycge_unpin_host_buffer(C.c_void_p(base + 8)) == abi.YCGE_ERR_INVALID_ARG
    assert L.ycge_unpin_host_buffer(None) == abi.YCGE_ERR_INVALID_ARG
    out = C.c_void_p(123)
    assert L.ycge_alloc_host_buffer(0, C.byref(out)) == abi.YCGE_ERR_INVALID_ARG and not out.value
    assert L.ycge_alloc_host_buffer(64, None) == abi.YCGE_ERR_INVALID_ARG
    assert L.ycge_free_host_buffer(None) == abi.YCGE_OK


def test_no_exception_crosses_the_c_abi(product_lib):
    """"no exceptions/longjmp across the ABI".  Every export's body is a function-try-block that ends in abi_catch
    (csrc/ycge_ctx.h): std::bad_alloc -> YCGE_ERR_OUT_OF_MEMORY, anything else -> YCGE_ERR_INTERNAL, the text in ycge_last_error.  Raised
    here by the test hook from inside such a body, with no context (the message then goes where ycge_create's failures go).  The n-th
    allocation failing inside ycge_scene_upload / ycge_create is the GPU test tests/test_gpu_abi_barrier.py (they need a context)."""
    L = product_lib
    L.ycge_debug_throw.restype = C.c_int
    L.ycge_debug_throw.argtypes = [C.c_void_p, C.c_int32]
    assert L.ycge_debug_throw(None, 0) == abi.YCGE_OK
    want = {1: (abi.YCGE_ERR_OUT_OF_MEMORY, b"std::bad_alloc"), 2: (abi.YCGE_ERR_INTERNAL, b"requested by ycge_debug_throw"),
            3: (abi.YCGE_ERR_INTERNAL, b"unknown C++ exception"), 4: (abi.YCGE_ERR_INTERNAL, b"internal error"), 5: (abi.YCGE_ERR_INTERNAL, b"thread")}
    for kind, (code, text) in want.items():
        assert L.ycge_debug_throw(None, kind) == code, kind
        msg = L.ycge_last_error(None)
        assert text in msg and b"C-ABI" in msg, (kind, msg)
    assert abi.STATUS_NAMES[abi.YCGE_ERR_INTERNAL] == "YCGE_ERR_INTERNAL"


def test_every_export_of_the_host_sources_is_guarded():
    """The barrier is mechanical: in csrc/ycge_host.cpp, ycge_frame.cpp, ycge_post_host.cpp and ycge_resident.cpp (the translation units with std containers, threads and
    `new`) every function defined inside an extern "C" block is a function-try-block whose handler calls abi_catch - except the one that
    cannot throw (ycge_last_error returns a pointer)."""
    import re
    csrc = Path(abi.__file__).resolve().parent / "csrc"
    for name in ("ycge_host.cpp", "ycge_frame.cpp", "ycge_post_host.cpp", "ycge_resident.cpp"):
        lines = (csrc / name).read_text().split("\n")
        in_c, n = False, 0
        for i, line in enumerate(lines):
            if line.startswith('extern "C" {'): in_c = True
            if line.startswith('} // extern "C"'): in_c = False
            m = re.match(r"^(int|size_t|void|const char \*)\s*(ycge_\w+)\(", line) if in_c else None
            if not m or line.rstrip().endswith(";"):
                continue
            if m.group(2) in ("ycge_last_error", "ycge_debug_fail_allocation", "ycge_peer_worker_main"):      # (the last: a thread's main, not an entry point - its body catches for itself)
                continue
            j = i
            while lines[j] not in ("try {", "{") and j < i + 6: j += 1
            assert lines[j] == "try {", f"{name}:{i + 1} {m.group(2)} is not a function-try-block"
            k = j + 1
            while lines[k] != "}": k += 1
            assert lines[k + 1].startswith("catch (...) {") and "abi_catch(" in lines[k + 1], f"{name}:{k + 2} {m.group(2)}"
            n += 1
        assert n >= (4 if name == "ycge_post_host.cpp" else 10), (name, n)


def test_the_nth_allocation_fails_inside_host_side_exports():
    """lib/var_faultinject.so (-DYCGE_FAULT_INJECTION=1: the library's own operator new throws std::bad_alloc on the n-th call after
    ycge_debug_fail_allocation(n)): n walks through the mesh-BVH builder and the scene validator - the host-side code ycge_scene_upload runs
    before it touches a device, reachable without one - and every call returns YCGE_ERR_OUT_OF_MEMORY or the full answer, bit-equal to the
    answer without a failure.  (With a context: tests/test_gpu_abi_barrier.py.)"""
    from yetanotherconsolegameengine_amd import build
    L = abi.load_library(build.build_variant("faultinject"))
    L.ycge_debug_fail_allocation.restype = C.c_int
    L.ycge_debug_fail_allocation.argtypes = [C.c_int64]
    rng = np.random.default_rng(5)
    tris = rng.uniform(-1, 1, size=(300, 9)).astype(np.float32)
    want_nodes, want_leaf, _ = _product_mesh(L, tris)
    L.ycge_host_build_mesh.restype = C.c_int
    failed, n = 0, 0
    while True:
        nodes = np.zeros(600, dtype=ob.NODE_DTYPE); leaf = np.zeros(300, dtype=np.int32); st = np.zeros(3, dtype=np.int32)
        L.ycge_debug_fail_allocation(n)
        rc = L.ycge_host_build_mesh(tris.ctypes.data, 300, nodes.ctypes.data, leaf.ctypes.data, st.ctypes.data)
        left = L.ycge_debug_fail_allocation(-1)
        if rc == abi.YCGE_ERR_OUT_OF_MEMORY:
            failed += 1
            assert b"bad_alloc" in L.ycge_last_error(None)
        else:
            assert rc == len(want_nodes) and _same(nodes[:rc], want_nodes) and _same(leaf, want_leaf), n
            if left >= 0: break
        n += 1 if n < 64 else n // 4
    assert failed >= 5, failed
    # the validator (std::string messages, index tables) on a real scene
    from yetanotherconsolegameengine_amd import scenes
    from yetanotherconsolegameengine_amd.scene import flatten
    flat = flatten(scenes.config_scene(1)[0])
    msg = C.create_string_buffer(256)
    codes = set()
    for n in range(24):
        L.ycge_debug_fail_allocation(n)
        codes.add(L.ycge_validate_scene(flat.byref(), msg, 256))
        L.ycge_debug_fail_allocation(-1)
    assert codes <= {abi.YCGE_OK, abi.YCGE_ERR_OUT_OF_MEMORY} and abi.YCGE_OK in codes, codes


def test_page_locked_sdr_arrays_own_their_pages():
    """TryFlipAndBlit(copy=False) and RenderAsync(sdr_slot=k) hand out numpy views over pages of the library; close(),
    Resize() and a new console size used to free them under a caller who still held the array.  Now the allocation's life is the array's:
    the pages go back (ycge_free_host_buffer, once) when the LAST view dies, whatever the renderer did meanwhile.  Stub allocator, no GPU."""
    import gc
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]

    class Stub:
        freed = []
        def ycge_alloc_host_buffer(self, n, out):
            out._obj.value = libc.malloc(n); C.memset(out._obj.value, 0, n); return 0
        def ycge_free_host_buffer(self, p):
            Stub.freed.append(p.value); libc.free(p); return 0
        def ycge_wait(self, ctx): return 0
        def ycge_destroy(self, ctx): return None

    r = RaytraceRenderer.__new__(RaytraceRenderer)
    r.L, r.ctx, r.fbW, r.fbH = Stub(), C.c_void_p(), 5, 3
    a = r._sdr_buffer()
    addr = a.ctypes.data
    a[...] = 7.0
    view = a[1:, :, 0]
    r.fbW = 6                       # a new console size: the renderer lets go of the old buffer ...
    b = r._sdr_buffer()
    del a; gc.collect()
    assert Stub.freed == [] and float(view.sum()) == 7.0 * view.size          # ... which the caller's view keeps alive and readable
    r.close(); gc.collect()
    assert Stub.freed == [] and b.shape == (3, 6, 2, 3) and float(b.sum()) == 0.0   # close() frees nothing a caller still holds
    del view; gc.collect()
    assert Stub.freed == [addr]
    del b; gc.collect()
    assert len(Stub.freed) == 2 and len(set(Stub.freed)) == 2
