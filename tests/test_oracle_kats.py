"""Known-answer tests pinning the ORACLE (CPU only).

The reference has no tests or golden vectors for this path ("parity unpinned"), so the oracle is pinned by
(1) KATs derived independently here from the cited formulas with plain Python integer / numpy-float32
arithmetic, (2) analytic cases, (3) structural invariants, (4) committed golden buffers (regression).
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob
from pathlib import Path as _Path
GOLDEN_DIR = _Path(__file__).resolve().parent / "golden"          # committed golden buffers (tests/golden/make_fixtures.py)
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.scene import (Box, CylinderY, Disk, Material, Plane, PointLight, Scene, Solid, Sphere,
                                                   Triangle, XYRect, XZRect, YZRect, vec3, ZERO)

M64 = (1 << 64) - 1
SALT = 0x9E3779B97F4A7C15
f32 = np.float32


# ---- independent restatement of RaytraceSampler.cs:43-80 in Python integers ------------------------
def py_splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def py_seed(x, y, frame, jx=0, jy=0, salt=SALT):
    h = 1469598103934665603
    h ^= (x * 0x9E3779B97F4A7C15) & M64; h = py_splitmix64(h)
    h ^= (y * 0xC2B2AE3D27D4EB4F) & M64; h = py_splitmix64(h)
    h ^= (frame * 0x165667B19E3779F9) & M64; h = py_splitmix64(h)
    h ^= ((jx & 0xff) << 8) ^ (jy & 0xff); h = py_splitmix64(h)
    h ^= salt; h = py_splitmix64(h)
    return h


def py_next_unit(state):
    state = py_splitmix64(state)
    m24 = state >> 40
    return state, f32(f32(m24) + f32(0.5)) * f32(1.0 / 16777216.0)


def bits(x):
    return int(np.float32(x).view(np.uint32))


def test_splitmix_and_seed_kats():
    L = ob.lib()
    assert L.orc_splitmix64(0) == 0xE220A8397B1DCDAF == py_splitmix64(0)
    for (x, y, fr), want in {(0, 0, 1): 0x17EF7D0094EB2C76, (1, 0, 1): 0xE30B62FE1AC2EDC5, (0, 1, 1): 0x7EDFB4004F82140E,
                             (79, 89, 1): 0x011D69A47BC4BB30, (1919, 1079, 7): 0x34D62C289D4318E2}.items():
        assert py_seed(x, y, fr) == want
        assert L.orc_per_frame_seed(x, y, fr, 0, 0, SALT) == want
    rng = np.random.RandomState(1)
    for _ in range(200):
        x, y, fr = int(rng.randint(0, 4000)), int(rng.randint(0, 2200)), int(rng.randint(1, 1 << 40))
        assert L.orc_per_frame_seed(x, y, fr, 0, 0, SALT) == py_seed(x, y, fr)


def test_next_unit_kats():
    L = ob.lib()
    st = C.c_uint64(L.orc_rng_init(0x17EF7D0094EB2C76))
    a = L.orc_rng_next_unit(C.byref(st)); b = L.orc_rng_next_unit(C.byref(st))
    assert bits(a) == 0x3EE1315F and bits(b) == 0x3DEF486C
    s = 0x17EF7D0094EB2C76
    s, pa = py_next_unit(s); s, pb = py_next_unit(s)
    assert bits(pa) == 0x3EE1315F and bits(pb) == 0x3DEF486C and s == st.value
    assert L.orc_rng_init(0) == SALT                       # Rng ctor: seed 0 -> golden-ratio constant


def test_blue_noise_and_jitter_kats():
    L = ob.lib()
    assert bits(L.orc_blue_noise_sample(0, 0, 1, 0)) == 0x3F047F54
    assert bits(L.orc_blue_noise_sample(0, 0, 1, 1)) == 0x3E170870
    frac = lambda v: f32(v) - f32(math.floor(float(v)))
    assert bits(frac(f32(2) * f32(0.61803398875))) == bits(L.orc_frac(f32(2) * f32(0.61803398875)))
    assert abs(float(frac(f32(2) * f32(0.61803398875))) - 0.2360680103) < 1e-9
    assert abs(float(frac(f32(2) * f32(0.38196601125))) - 0.7639320493) < 1e-9
    tile = np.array([[0, 32, 8, 40, 2, 34, 10, 42], [48, 16, 56, 24, 50, 18, 58, 26], [12, 44, 4, 36, 14, 46, 6, 38], [60, 28, 52, 20, 62, 30, 54, 22],
                     [3, 35, 11, 43, 1, 33, 9, 41], [51, 19, 59, 27, 49, 17, 57, 25], [15, 47, 7, 39, 13, 45, 5, 37], [63, 31, 55, 23, 61, 29, 53, 21]])
    assert sorted(tile.reshape(-1)) == list(range(64))      # a permutation of 0..63 (RaytraceSampler.cs:9-19)
    for y in range(8):
        for x in range(8):
            want = frac(f32(f32(tile[y, x]) + f32(0.5)) * f32(1 / 64) + frac(f32(1) * f32(0.7548776662466927)))
            assert bits(L.orc_blue_noise_sample(x + 8, y + 16, 0, 0)) == bits(want)


def test_csharp_numeric_semantics():
    L = ob.lib()
    nan, inf = float("nan"), float("inf")
    assert math.isnan(L.orc_max(nan, 1.0)) and math.isnan(L.orc_max(1.0, nan))       # MathF.Max propagates NaN
    assert math.isnan(L.orc_min(nan, 1.0)) and math.isnan(L.orc_min(1.0, nan))
    assert math.copysign(1, L.orc_max(-0.0, 0.0)) == 1 and math.copysign(1, L.orc_min(0.0, -0.0)) == -1
    assert L.orc_f2i(3.9) == 3 and L.orc_f2i(-3.9) == -3
    for v in (nan, inf, -inf, 3e9, -3e9, 2147483648.0):
        assert L.orc_f2i(v) == -2147483648                                            # cvttss2si "integer indefinite"
    assert L.orc_f2i(-2147483648.0) == -2147483648 and L.orc_f2i(2147483520.0) == 2147483520


def test_transcendental_kernels_against_libm():
    L = ob.lib()
    s, c = C.c_float(), C.c_float()
    xs = np.concatenate([np.linspace(0, 6.2831855, 4001, dtype=np.float32), np.float32([0.0, 1e-8, 1.5707964, 3.1415927, 6.2831855])])
    worst = 0
    for x in xs:
        L.orc_sincos(x, C.byref(s), C.byref(c))
        rs, rc = np.float32(math.sin(float(x))), np.float32(math.cos(float(x)))
        worst = max(worst, abs(int(np.float32(s.value).view(np.int32)) - int(rs.view(np.int32))) if abs(rs) > 1e-6 else 0,
                    abs(int(np.float32(c.value).view(np.int32)) - int(rc.view(np.int32))) if abs(rc) > 1e-6 else 0)
        assert abs(s.value - float(rs)) <= 1.2e-7 and abs(c.value - float(rc)) <= 1.2e-7
    assert worst <= 1                                                                  # within 1 ulp of correctly rounded
    for x in np.linspace(0, 1, 1001, dtype=np.float32):
        assert bits(L.orc_pow5(x)) == bits(np.float32(float(x) ** 5))                  # exact product, rounded once
    for x in np.linspace(-20, 5, 501, dtype=np.float32):
        assert abs(L.orc_exp(x) / math.exp(float(x)) - 1) < 1.3e-7
    for x in np.geomspace(1e-6, 50, 501).astype(np.float32):
        assert abs(L.orc_log(x) - math.log(float(x))) <= 1.3e-7 * max(1.0, abs(math.log(float(x))))
    for x in np.linspace(0.001, 1, 300, dtype=np.float32):
        assert abs(L.orc_pow(x, np.float32(1 / 2.2)) / (float(x) ** float(np.float32(1 / 2.2))) - 1) < 2e-7


def test_morton_index_table():
    L = ob.lib()
    seen = set()
    for z in range(8):
        for y in range(8):
            for x in range(8):
                m = L.orc_morton3(x, y, z)
                want = sum((((x >> b) & 1) << (3 * b)) | (((y >> b) & 1) << (3 * b + 1)) | (((z >> b) & 1) << (3 * b + 2)) for b in range(3))
                assert m == want
                seen.add(m)
    assert seen == set(range(512))


# ---- analytic intersection KATs -----------------------------------------------------------------------
def _scene(objs, lights=()):
    s = Scene()
    for o in objs:
        s.Add(o)
    s.Lights.extend(lights)
    return s


def test_axis_ray_vs_sphere_and_rect_edges():
    white = Solid(vec3(0.8, 0.8, 0.8))
    sc = _scene([Sphere(vec3(0, 0, -5), 1.0, Material(vec3(1, 0, 0))), XZRect(-1.0, 1.0, -1.0, 1.0, -2.0, white, 0.0, 0.0)])
    with ob.OracleRenderer(sc, 8, 4) as r:
        h = r.scene_hit((0, 0, 0), (0, 0, -1))
        assert h[0] == 1 and h[1] == 0 and h[3] == 4.0                              # t = d - r
        assert tuple(h[7:10]) == (0.0, 0.0, 1.0)                                    # outward normal
        h = r.scene_hit((0, 0, -5), (0, 0, -1))                                     # from the centre: far root, outward N
        assert h[0] == 1 and h[3] == 1.0 and tuple(h[7:10]) == (0.0, 0.0, -1.0)
        # rect edge inclusivity (px >= X0 & px <= X1), Surfaces.cs:269
        assert r.scene_hit((1.0, 0, 0), (0, -1, 0), brute=True)[0] == 1 and r.scene_hit((-1.0, 0, 1.0), (0, -1, 0), brute=True)[0] == 1
        assert r.scene_hit((np.nextafter(f32(1.0), f32(2.0)), 0, 0), (0, -1, 0), brute=True)[0] == 0
        # ... but through the BVH the same edge ray is LOST: dir.x == 0 gives invD = inf and the slab that the
        # origin sits on evaluates (1 - 1) * inf = NaN, which MathF.Max/Min propagate (BVH.cs:228-235) -> miss
        assert r.scene_hit((1.0, 0, 0), (0, -1, 0))[0] == 0
        assert r.scene_hit((0.5, 0, 0), (0, -1, 0))[0] == 1
        # tMin / tMax are inclusive for rects and spheres
        assert r.scene_hit((0, 0, 0), (0, -1, 0), t_min=2.0, t_max=2.0)[0] == 1


def test_disk_ignores_y_in_radius_test_and_cylinder_ignores_center_y():
    sc = _scene([Disk(vec3(0, 1, -3), vec3(0, 0, 1), 0.5, Solid(vec3(1, 1, 1)), 0.0, 0.0),           # vertical disk: quirk 4
                 CylinderY(vec3(5, 100.0, 0), 0.5, 0.0, 1.0, True, Material(vec3(1, 1, 1)))])          # quirk 6
    with ob.OracleRenderer(sc, 8, 4) as r:
        h = r.scene_hit((0.2, 3.0, 0), (0, 0, -1), brute=True)   # 2 units above the disk centre in y: the x,z-only test accepts
        assert h[0] == 1 and h[1] == 0 and h[3] == 3.0
        assert r.scene_hit((0.2, 3.0, 0), (0, 0, -1))[0] == 0     # (its BVH box, centre +- radius, still culls that ray)
        h = r.scene_hit((5, 0.5, 5), (0, 0, -1))         # cylinder sits at absolute y in [0,1] although Center.Y = 100
        assert h[0] == 1 and h[1] == 1 and abs(h[3] - 4.5) < 1e-6


def test_box_face_order_and_tie_rule():
    sc = _scene([Box(vec3(-1, -1, -1), vec3(1, 1, 1), Solid(vec3(1, 1, 1)), 0.0, 0.0)])
    with ob.OracleRenderer(sc, 8, 4) as r:
        assert r.scene_hit((0, 0, 5), (0, 0, -1))[2] == 0        # +Z face first (BoundedObjects.cs:84)
        assert r.scene_hit((0, 0, -5), (0, 0, 1))[2] == 1
        assert r.scene_hit((0, 5, 0), (0, -1, 0))[2] == 2 and r.scene_hit((0, -5, 0), (0, 1, 0))[2] == 3
        assert r.scene_hit((5, 0, 0), (-1, 0, 0))[2] == 4 and r.scene_hit((-5, 0, 0), (1, 0, 0))[2] == 5
        # a ray through the +Z/+X edge hits both faces at the same t: the LATER face (index 4) wins the tie
        h = r.scene_hit((2, 0, 2), (-1, 0, -1))
        assert h[0] == 1 and h[2] == 4


# ---- structural invariants of the builders ---------------------------------------------------------------
def _check_tree(nodes, leaf, n_items):
    assert sorted(leaf.tolist()) == list(range(n_items))                               # every item in exactly one leaf
    for i, nd in enumerate(nodes):
        if nd["count"] > 0:
            assert nd["left"] == -1 and nd["right"] == -1
            continue
        l, r = nodes[nd["left"]], nodes[nd["right"]]
        assert nd["left"] == i + 1                                                     # pre-order: left child follows its parent
        for c in (l, r):
            assert (c["min"] >= nd["min"]).all() and (c["max"] <= nd["max"]).all()    # child boxes inside the parent


def test_builder_invariants_and_bvh_equals_bruteforce():
    rng = np.random.RandomState(7)
    objs = [Sphere(vec3(*rng.uniform(-4, 4, 3)), float(rng.uniform(0.2, 0.8)), Material(vec3(1, 1, 1))) for _ in range(40)]
    objs += [Box(vec3(-6, -1, -6), vec3(-5, 1, -5), Solid(vec3(1, 1, 1)), 0.0, 0.0), XZRect(-8.0, 8.0, -8.0, 8.0, -5.0, Solid(vec3(1, 1, 1)), 0.0, 0.0)]
    pos, faces = scenes.make_torus_knot(40, 12)
    from yetanotherconsolegameengine_amd.scene import Mesh
    objs.append(Mesh(pos[faces], Material(vec3(0.5, 0.5, 0.5))))
    sc = _scene(objs)
    with ob.OracleRenderer(sc, 8, 4) as r:
        _check_tree(r.accel(abi.ACCEL_SCENE_NODES), r.accel(abi.ACCEL_SCENE_LEAF_INDEX), len(objs))
        _check_tree(r.accel(abi.ACCEL_MESH_NODES), r.accel(abi.ACCEL_MESH_LEAF_INDEX), faces.shape[0])
        for _ in range(400):
            o = rng.uniform(-9, 9, 3); d = rng.normal(size=3)
            a, b = r.scene_hit(o, d), r.scene_hit(o, d, brute=True)
            assert a[0] == b[0]
            if a[0]:
                assert a[3] == b[3]                                                   # same closest t, bit for bit


def test_dotnet_introsort_properties():
    L = ob.lib()
    rng = np.random.RandomState(3)
    for n in (2, 3, 5, 16, 17, 33, 200, 1000):
        for dup in (False, True):
            keys = (rng.randint(0, 7, n) if dup else rng.uniform(-1, 1, n)).astype(np.float32)
            idx = np.arange(n, dtype=np.int32)
            k2, i2 = keys.copy(), idx.copy()
            L.orc_introsort(k2.ctypes.data_as(C.POINTER(C.c_float)), i2.ctypes.data_as(C.POINTER(C.c_int32)), n)
            assert (np.diff(k2) >= 0).all() and sorted(i2.tolist()) == list(range(n)) and (keys[i2] == k2).all()
    # regression vector for the unstable order of equal keys (heapsort is not reached; insertion + partition order)
    keys = np.float32([1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0])
    idx = np.arange(20, dtype=np.int32)
    L.orc_introsort(keys.ctypes.data_as(C.POINTER(C.c_float)), idx.ctypes.data_as(C.POINTER(C.c_int32)), 20)
    assert idx.tolist() == GOLDEN_EQUAL_KEY_ORDER, idx.tolist()


GOLDEN_EQUAL_KEY_ORDER = [9, 17, 15, 13, 11, 7, 19, 5, 3, 1, 8, 18, 10, 4, 12, 14, 2, 16, 6, 0]   # frozen regression vector


def test_committed_golden_buffers_cornell():
    z = np.load(GOLDEN_DIR / "cornell_80x45_frames123.npz")
    sc, w, h, ss, pose = scenes.config_scene(1)
    with ob.OracleRenderer(sc, w, h, ss, pose) as r:
        for frame in (1, 2, 3):
            r.render(stages=1)
            if frame == 1:
                for name, which in (("rays", abi.BUF_RAYS), ("prim_id", abi.BUF_PRIM_ID), ("sub_id", abi.BUF_SUB_ID), ("hit_t", abi.BUF_HIT_T),
                                    ("current_hdr", abi.BUF_CURRENT_HDR), ("rng_state", abi.BUF_RNG_STATE)):
                    assert np.array_equal(r.read(which).view(np.uint8), z[f"f1_{name}"].view(np.uint8)), name
            assert np.array_equal(r.read(abi.BUF_TAA_HISTORY).view(np.uint8), z[f"f{frame}_taa_history"].view(np.uint8))
        # a Cornell primary ray through the image centre hits the back wall (object 4) at z = -5 from (0,1,0)
        pid = r.read(abi.BUF_PRIM_ID)
        assert pid[45, 40] in (4, 6, 7) and (pid >= 0).all()        # closed box: no primary ray escapes


# ---- independent restatement of steps 6-8 of TryFlipAndBlit in numpy float32 scalars ------------------------------
# ApplyAtrousDenoise (RaytraceRenderer.cs:622-722) incl. the buffer walk of :648-650 / :718 (iteration 1 in place),
# ToneMapper.UpdateExposure (ToneMapper.cs:49-91), the box average of :229-264 and ToneMapAndEncode (:204-260).
# exp / log / pow are the oracle's own scalar kernels (pinned against libm above): everything ELSE - loop order,
# clamping, buffer aliasing, the serial sum, the tone curve - is written again here from the C# text.
def _py_post(L, fbw, fbh, ss, hdr, alb, nrm, dep, sky, iters, phi, ae0):
    ex = lambda v: f32(L.orc_exp(float(v)))
    W, H = fbw * ss, fbh * 2 * ss
    k = [f32(1) / f32(16), f32(1) / f32(4), f32(3) / f32(8), f32(1) / f32(4), f32(1) / f32(16)]

    def normalized(v):
        ls = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
        if ls <= 0:
            return v.copy()
        inv = f32(1) / f32(np.sqrt(ls))
        return np.array([v[0] * inv, v[1] * inv, v[2] * inv], f32)

    def luma(c):
        return f32(f32(f32(f32(0.2126) * c[0]) + f32(f32(0.7152) * c[1])) + f32(f32(0.0722) * c[2]))

    src = hdr.copy(); A = np.zeros_like(hdr); B = np.zeros_like(hdr)
    cur, dst = src, A
    phis = [max(f32(1e-6), f32(p)) for p in phi]
    for it in range(max(1, iters)):
        step = 1 << it
        for y in range(H):
            for x in range(W):
                if sky[y, x]:
                    dst[y, x] = cur[y, x]; continue
                c0 = cur[y, x].copy(); a0 = alb[y, x]; n0 = normalized(nrm[y, x]); z0 = dep[y, x]
                wsum = f32(0); acc = np.zeros(3, f32)
                for ky in range(-2, 3):
                    sy = min(max(y + ky * step, 0), H - 1)
                    for kx in range(-2, 3):
                        sx = min(max(x + kx * step, 0), W - 1)
                        if sky[sy, sx] != sky[y, x]:
                            continue
                        wb = f32(k[kx + 2] * k[ky + 2])
                        c = cur[sy, sx]; a = alb[sy, sx]; n = normalized(nrm[sy, sx]); z = dep[sy, sx]
                        dl = abs(f32(luma(c) - luma(c0)))
                        dot = f32(f32(f32(n0[0] * n[0]) + f32(n0[1] * n[1])) + f32(n0[2] * n[2]))
                        dn = max(f32(0), f32(f32(1) - dot))
                        dz = abs(f32(z - z0))
                        da = f32(f32(abs(f32(a[0] - a0[0])) + abs(f32(a[1] - a0[1]))) + abs(f32(a[2] - a0[2])))
                        w = f32(f32(f32(f32(wb * ex(f32(-dl / phis[0]))) * ex(f32(-dn / phis[1]))) * ex(f32(-dz / phis[2]))) * ex(f32(-da / phis[3])))
                        acc = np.array([f32(acc[0] + f32(c[0] * w)), f32(acc[1] + f32(c[1] * w)), f32(acc[2] + f32(c[2] * w))], f32)
                        wsum = f32(wsum + w)
                if wsum > f32(1e-8):
                    inv = f32(1) / wsum
                    dst[y, x] = [acc[0] * inv, acc[1] * inv, acc[2] * inv]
                else:
                    dst[y, x] = c0
        tmp = cur; cur = dst; dst = B if tmp is A else A            # `var tmp = cur; cur = dst; dst = (tmp == scratchA) ? scratchB : scratchA`
    den = cur
    # UpdateExposure (serial overload)
    st = max(2, ss * 2)
    log_sum = f32(0); cnt = 0
    for py in range(0, H, st):
        for px in range(0, W, st):
            if sky[py, px]:
                continue
            lum = luma(den[py, px])
            if lum > 0:
                log_sum = f32(log_sum + f32(L.orc_log(float(f32(f32(1e-6) + lum))))); cnt += 1
    avg_log = f32(log_sum / f32(max(1, cnt))) if cnt > 0 else f32(0)
    avg_lum = ex(avg_log)
    target = f32(f32(0.18) / max(f32(1e-6), avg_lum)) if cnt > 0 else f32(ae0)
    target = min(max(target, f32(0.10)), f32(1.50))
    s = f32(f32(1) - ex(f32(-0.2)))
    ae = f32(f32(ae0) + f32(f32(target - f32(ae0)) * s))
    eff = f32(f32(1.0) * ae)

    def aces(xv):
        num = f32(xv * f32(f32(f32(2.51) * xv) + f32(0.03)))
        dn_ = f32(f32(xv * f32(f32(f32(2.43) * xv) + f32(0.59))) + f32(0.14))
        yv = f32(num / dn_) if dn_ > 0 else f32(0)
        return min(max(yv, f32(0)), f32(1))

    def map_pixel(c):
        inv_g = f32(f32(1) / max(f32(0.1), f32(2.2)))
        v = []
        for ch in range(3):
            t = aces(f32(max(f32(0), c[ch]) * eff))
            t = min(max(t, f32(0)), f32(1))
            v.append(min(max(f32(L.orc_pow(float(t), float(inv_g))), f32(0)), f32(1)))
        r, g, b = v
        yv = luma(np.array([r, g, b], f32))
        chroma = f32(max(r, max(g, b)) - min(r, min(g, b)))
        fsat = f32(f32(2.0) * f32(f32(1) + f32(f32(0.0) * f32(f32(1) - chroma))))
        return [min(max(f32(yv + f32(f32(ch_ - yv) * fsat)), f32(0)), f32(1)) for ch_ in (r, g, b)]

    sdr = np.zeros((fbh, fbw, 2, 3), f32)
    inv = f32(f32(1) / f32(ss * ss))
    for cy in range(fbh):
        for cx in range(fbw):
            for half, y0 in ((0, cy * 2 * ss), (1, (cy * 2 + 1) * ss)):
                sm = np.zeros(3, f32)
                for sy in range(ss):
                    for sx in range(ss):
                        sm = (sm + den[y0 + sy, cx * ss + sx]).astype(f32)
                sdr[cy, cx, half] = map_pixel((sm * inv).astype(f32))
    return den, ae, eff, sdr


@pytest.mark.parametrize("fbw,fbh,ss,iters", [(7, 4, 1, 3), (5, 3, 2, 3), (6, 3, 1, 2)])
def test_post_stage_against_python_restatement(fbw, fbh, ss, iters):
    L = ob.lib()
    L.orc_post_probe.restype = C.c_int
    L.orc_post_probe.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    W, H = fbw * ss, fbh * 2 * ss
    rng = np.random.default_rng(fbw * 100 + fbh * 10 + ss)
    hdr = rng.uniform(0, 2.5, (H, W, 3)).astype(f32)
    alb = rng.uniform(0, 1, (H, W, 3)).astype(f32)
    nrm = rng.normal(size=(H, W, 3)).astype(f32); nrm[0, 0] = 0           # one zero normal: Normalized() returns it unchanged
    dep = rng.uniform(1, 9, (H, W)).astype(f32)
    sky = (rng.uniform(size=(H, W)) < 0.2).astype(np.uint8)
    hdr[sky == 1] = f32(0.7)
    phi = np.array([3.0, 0.35, 2.0, 0.20], f32)
    den_o = np.zeros_like(hdr); sdr_o = np.zeros((fbh, fbw, 2, 3), f32); expo = np.array([1.0, 0.0], f32)
    assert L.orc_post_probe(fbw, fbh, ss, hdr.ctypes.data, alb.ctypes.data, nrm.ctypes.data, dep.ctypes.data, sky.ctypes.data, iters,
                            phi.ctypes.data, den_o.ctypes.data, expo.ctypes.data, sdr_o.ctypes.data) == 0
    den_p, ae, eff, sdr_p = _py_post(L, fbw, fbh, ss, hdr, alb, nrm, dep, sky, iters, phi, 1.0)
    assert np.array_equal(den_o.view(np.uint32), np.ascontiguousarray(den_p, f32).view(np.uint32))
    assert bits(expo[0]) == bits(ae) and bits(expo[1]) == bits(eff)
    assert np.array_equal(sdr_o.view(np.uint32), sdr_p.view(np.uint32))
    # the in-place iteration is not a ping-pong: redoing iteration 1 out of place must give a different image
    if iters >= 2:
        assert not np.array_equal(den_p, hdr)


def test_committed_golden_post_stage_cornell():
    """regression fixture of steps 6-8 (tests/golden/make_fixtures.py post)"""
    z = np.load(GOLDEN_DIR / "cornell_80x45_post.npz")
    sc, w, h, ss, pose = scenes.config_scene(1)
    with ob.OracleRenderer(sc, w, h, ss, pose) as r:
        for frame in (1, 2, 3):
            sdr = r.render(stages=2, want_sdr=True)
            assert np.array_equal(sdr.view(np.uint32), z[f"f{frame}_sdr"].view(np.uint32)), frame
            assert bits(r.stats.exposure) == bits(z[f"f{frame}_exposure"])
        assert np.array_equal(r.read(abi.BUF_DENOISED).view(np.uint32), z["f3_denoised"].view(np.uint32))
        assert 0.1 <= float(r.stats.exposure) <= 1.5 and (sdr >= 0).all() and (sdr <= 1).all()


def test_whole_pixel_path_against_python_restatement():
    """The oracle against tests/py_restatement.py (a second, independent restatement written from the C# text): camera
    rays, primary hits, radiance incl. checker floor, true mirror, glass (reflect + refract split, transmittance of the
    shadow rays), emissive panel, two lights, diffuse bounce; RNG state; two frame numbers.  Bit for bit."""
    import py_restatement as pr
    s = Scene()
    s.Ambient.Color, s.Ambient.Intensity = vec3(1, 1, 1), 0.05
    from yetanotherconsolegameengine_amd.scene import Checker
    s.Add(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.75, 0.75, 0.75), vec3(0.2, 0.2, 0.2), 0.8), 0.05, 0.0))
    s.Add(Sphere(vec3(-1.1, 0.7, -3.2), 0.7, Material(vec3(0.8, 0.3, 0.2), 0.1, 0.0, ZERO)))
    s.Add(Sphere(vec3(0.6, 0.55, -2.6), 0.55, Material(vec3(0.97, 0.97, 0.97), 0.0, 0.95, ZERO)))                       # mirror branch
    s.Add(Sphere(vec3(-0.2, 0.4, -1.7), 0.4, Material(vec3(1, 1, 1), 0.0, 0.05, ZERO, 0.9, 1.5, vec3(0.9, 1.0, 0.9))))   # glass
    s.Add(XZRect(-0.6, 0.6, -2.9, -2.2, 2.2, Material(vec3(0, 0, 0), 0.0, 0.0, vec3(2.0, 1.8, 1.5)), 0.0, 0.0))          # emissive panel
    s.Lights.append(PointLight(vec3(-2.0, 3.5, -1.0), vec3(1.0, 0.95, 0.9), 60.0))
    s.Lights.append(PointLight(vec3(2.5, 2.0, -4.0), vec3(0.9, 0.95, 1.0), 35.0))
    pose = dict(pos=(0.05, 1.0, 0.4), yaw=0.04, pitch=-0.15, fov=55.0)
    fbw, fbh = 14, 6
    with ob.OracleRenderer(s, fbw, fbh, 1, pose) as o:
        for frame in (1, 2):
            o.render(stages=0)
            p = pr.render_frame(o.L, s, o.hiW, o.hiH, pose, frame)
            for name, which, key in (("rays", abi.BUF_RAYS, "rays"), ("radiance", abi.BUF_CURRENT_HDR, "hdr"), ("albedo", abi.BUF_G_ALBEDO, "albedo"),
                                     ("normal", abi.BUF_G_NORMAL, "normal"), ("depth", abi.BUF_G_DEPTH, "depth")):
                a, b = o.read(which), p[key]
                assert np.array_equal(a.view(np.uint32), np.ascontiguousarray(b, f32).view(np.uint32)), (frame, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
            assert np.array_equal(o.read(abi.BUF_SKY_MASK), p["sky"]) and np.array_equal(o.read(abi.BUF_PRIM_ID), p["prim"])
            assert np.array_equal(o.read(abi.BUF_RNG_STATE), p["rng"])
            assert set(np.unique(p["prim"])) >= {0, 1, 2, 3}                         # floor, diffuse, mirror and glass spheres are all seen


def _check_frames_against_restatement(scene, fbw, fbh, pose, frames, need_objects, bvh=False):
    """bvh: the restatement answers Scene.Hit through ITS OWN restated tree (builder + walk) instead of the brute-force loop - needed wherever
    an object's bounds do not cover what its Hit accepts (a tilted Disk: the reference bounds it by a cube of its radius, Surfaces.cs:97-105,
    and tests the radius in x and z only, :108-142 - the tree then never reaches hits a loop over all objects finds) or two hits tie."""
    import py_restatement as pr
    taa = {}
    scene_for_pr = pr.BvhScene(scene) if bvh else scene
    with ob.OracleRenderer(scene, fbw, fbh, 1, pose) as o:
        seen = set()
        for frame in range(1, frames + 1):
            o.render(stages=1)
            p = pr.render_frame(o.L, scene_for_pr, o.hiW, o.hiH, pose, frame)
            for name, which, key in (("rays", abi.BUF_RAYS, "rays"), ("radiance", abi.BUF_CURRENT_HDR, "hdr"), ("albedo", abi.BUF_G_ALBEDO, "albedo"),
                                     ("normal", abi.BUF_G_NORMAL, "normal"), ("depth", abi.BUF_G_DEPTH, "depth")):
                a, b = o.read(which), np.ascontiguousarray(p[key], f32)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (frame, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
            assert np.array_equal(o.read(abi.BUF_SKY_MASK), p["sky"]) and np.array_equal(o.read(abi.BUF_PRIM_ID), p["prim"])
            assert np.array_equal(o.read(abi.BUF_RNG_STATE), p["rng"])
            hist = pr.temporal_blend(taa, p["hdr"], p["normal"], p["depth"], p["sky"], force_reset=False)
            assert np.array_equal(o.read(abi.BUF_TAA_HISTORY).view(np.uint32), np.ascontiguousarray(hist, f32).view(np.uint32)), (frame, "taa")
            seen |= set(np.unique(p["prim"]).tolist())
        assert seen >= need_objects, seen


def test_every_analytic_primitive_and_taa_against_python_restatement():
    """Disk, the three rects, Box (six faces), CylinderY with caps, Triangle (scalar path) and three frames of
    TemporalBlendWithClamp, oracle vs tests/py_restatement.py, bit for bit."""
    from yetanotherconsolegameengine_amd.scene import Checker
    s = Scene()
    s.Ambient.Color, s.Ambient.Intensity = vec3(1, 1, 1), 0.03
    s.Add(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Checker(vec3(0.7, 0.7, 0.7), vec3(0.25, 0.25, 0.25), 0.6), 0.0, 0.0))
    s.Add(CylinderY(vec3(-1.3, 0.0, -3.0), 0.5, 0.0, 1.4, True, Material(vec3(0.2, 0.35, 0.9), 0.1, 0.0, ZERO)))
    s.Add(Disk(vec3(1.5, 0.02, -2.0), vec3(0, 1, 0), 0.7, Solid(vec3(0.8, 0.8, 0.1)), 0.0, 0.0))
    s.Add(Triangle(vec3(0.1, 0.0, -3.4), vec3(1.2, 1.3, -3.0), vec3(-0.7, 0.8, -2.7), Material(vec3(0.9, 0.25, 0.25), 0.1, 0.0, ZERO)))
    s.Add(Box(vec3(-2.6, 0.0, -5.0), vec3(-1.7, 1.1, -4.1), Solid(vec3(0.8, 0.8, 0.8)), 0.1, 0.92))                      # mirror via the ctor override
    s.Add(XYRect(0.8, 2.2, 0.0, 1.2, -4.6, Solid(vec3(0.3, 0.8, 0.4)), 0.0, 0.0))
    s.Add(YZRect(0.0, 1.0, -3.6, -2.8, 2.4, Solid(vec3(0.6, 0.3, 0.8)), 0.0, 0.0))
    s.Add(XZRect(-0.5, 0.5, -2.6, -2.0, 2.0, Material(vec3(0, 0, 0), 0.0, 0.0, vec3(1.8, 1.7, 1.5)), 0.0, 0.0))
    s.Lights.append(PointLight(vec3(-1.5, 3.0, -0.5), vec3(1.0, 0.95, 0.9), 50.0))
    pose = dict(pos=(0.0, 1.1, 0.8), yaw=-0.03, pitch=-0.12, fov=60.0)
    _check_frames_against_restatement(s, 16, 6, pose, 3, {0, 1, 2, 3, 4, 5, 6})


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_drawn_scenes_against_python_restatement(seed):
    """tests/random_scenes.py (the draws the HIP path is held to in tests/test_gpu_random_scenes.py), shrunk to what pure Python walks in a
    second: 12 - 40 objects of every Hittable class incl. one small mesh and sometimes a voxel volume, every material branch, duplicated spheres
    and boxes sharing a face (ties), 0 - 4 lights.  The oracle against the second restatement - its own tree builder, walk and per-class bounds -
    over two frames incl. TemporalBlendWithClamp: rays, primary hit, radiance, G-buffer, sky mask, RNG state, history, bit for bit."""
    import random_scenes
    s, pose = random_scenes.random_scene(seed, n_range=(12, 40), mesh_nu=(5, 9), mesh_nv=(3, 5), max_meshes=1)
    _check_frames_against_restatement(s, 12, 5, pose, 2, set(), bvh=True)


@pytest.mark.parametrize("seed", [0, 1, 3, 4, 5, 7, 9, 10, 12, 13, 14, 16])
def test_pushed_drawn_scenes_against_python_restatement(seed):
    """... and the same draws pushed one way (random_scenes.harden: mostly glass, a dozen lights, the camera inside an object, degenerate objects,
    everything scaled by 1e-2 .. 1e3, a VolumeScene of voxel chunks with lit lights) - the modes the restatement covers (it has no textures, no
    per-triangle materials, and thousands of objects are beyond pure Python).  120 such draws were soaked: none differs."""
    import random_scenes
    s, pose = random_scenes.random_scene(seed, n_range=(12, 40), mesh_nu=(5, 9), mesh_nv=(3, 5), max_meshes=1)
    random_scenes.harden(s, pose, seed)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        _check_frames_against_restatement(s, 12, 5, pose, 2, set(), bvh=True)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_frame_to_frame_state_against_python_restatement(seed):
    """The renderer's state BETWEEN frames, oracle against the restatement (py_restatement.FrameState + temporal_blend): a drawn camera walk whose steps
    straddle the two reset thresholds (TemporalAA.cs:58-67: translation, yaw, pitch, each in binary32), a Resize in the middle (history invalid, the
    committed camera forgotten: TemporalAA.cs:34-45, RaytraceRenderer.cs:128-137 - so the frame after it is a reset by VALIDITY, and the one after that
    compares with a camera committed since), frame-counter jumps.  Every frame: the oracle's `history_reset`, its radiance and its history, bit for bit."""
    import py_restatement as pr
    import random_scenes
    rng = np.random.default_rng(40 + seed)
    s, pose = random_scenes.random_scene(seed, n_range=(10, 24), mesh_nu=(5, 8), mesh_nv=(3, 5), max_meshes=1)
    w, h = 10, 4
    st, taa = pr.FrameState(), {}
    ps = pr.BvhScene(s)
    with ob.OracleRenderer(s, w, h, 1, pose) as o:
        resets = []
        for step in range(9):
            k = [0.0, 0.0017, 0.0024, 0.0026, 0.004, 0.3][int(rng.integers(0, 6))]         # around TemporalAA's 0.0025
            which = int(rng.integers(0, 3))
            if which == 0: pose = dict(pose, pos=(f32(f32(pose["pos"][0]) + f32(k)), pose["pos"][1], pose["pos"][2]))
            elif which == 1: pose = dict(pose, yaw=float(f32(f32(pose["yaw"]) + f32(k))))
            else: pose = dict(pose, pitch=float(f32(f32(pose["pitch"]) - f32(k))))
            if step == 4:
                w, h = 8, 5
                o.resize(w, h, 1); taa.clear(); st.forget_camera()
            if step == 6:
                o.set_frame_counter((1 << 31) - 2); st.frame = (1 << 31) - 2
            o.set_camera(pose["pos"], pose["yaw"], pose["pitch"], pose["fov"])
            o.render(stages=1)
            st.frame += 1
            want_reset = st.should_reset(pose["pos"], pose["yaw"], pose["pitch"])
            p = pr.render_frame(o.L, ps, o.hiW, o.hiH, pose, st.frame)
            first = not taa
            hist = pr.temporal_blend(taa, p["hdr"], p["normal"], p["depth"], p["sky"], force_reset=want_reset)
            st.commit(pose["pos"], pose["yaw"], pose["pitch"])
            assert int(o.stats.history_reset) == int(want_reset or first), (step, k, which, want_reset, first)
            assert np.array_equal(o.read(abi.BUF_CURRENT_HDR).view(np.uint32), np.ascontiguousarray(p["hdr"], f32).view(np.uint32)), (step, "radiance")
            assert np.array_equal(o.read(abi.BUF_TAA_HISTORY).view(np.uint32), np.ascontiguousarray(hist, f32).view(np.uint32)), (step, "history")
            resets.append(int(want_reset or first))
        assert 0 < sum(resets) < len(resets), resets          # the walk did both: frames that blend and frames that start over


def test_a_tilted_disk_is_missed_by_the_tree_as_in_the_reference():
    """A quirk the drawn scenes found (it is the REFERENCE's, and the oracle keeps it): Disk.TryGetBounds is a cube of the radius around the
    centre (Surfaces.cs:97-105) while Disk.Hit tests the radius in x and z only (:119-121) - on a steep disk the accepted
    points reach far above and below that cube, and Scene.Hit, which only ever goes through the tree (Scene.cs:71-76), never tries them.
    The oracle's tree walk misses where its own brute-force loop (and a loop over Scene.Objects in any restatement) hits."""
    s = Scene()
    s.Add(Sphere(vec3(6.0, 0.0, -6.0), 0.5, Material(vec3(0.5, 0.5, 0.5), 0.0, 0.0, ZERO)))             # (two more objects: a tree with an inner node)
    s.Add(Sphere(vec3(-6.0, 0.0, -6.0), 0.5, Material(vec3(0.5, 0.5, 0.5), 0.0, 0.0, ZERO)))
    s.Add(Disk(vec3(0.0, 0.0, -5.0), vec3(0.0, 0.05, 1.0), 1.0, Solid(vec3(0.8, 0.2, 0.2)), 0.0, 0.0))   # nearly vertical: y = -20 (z + 5) on the plane
    pose = dict(pos=(0.0, 0.0, 0.0), yaw=0.0, pitch=0.0, fov=45.0)
    with ob.OracleRenderer(s, 4, 2, 1, pose) as o:
        inside = o.scene_hit((0.0, 0.0, 0.0), (0.0, 0.0, -1.0))             # through the centre: inside the cube, found by both
        assert inside[0] == 1.0 and inside[1] == 2.0 and o.scene_hit((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), brute=True)[1] == 2.0
        d = np.float32([0.0, 4.0, -4.8]); d = d / np.float32(np.sqrt(np.float32(d @ d)))
        tree, loop = o.scene_hit((0.0, 0.0, 0.0), d), o.scene_hit((0.0, 0.0, 0.0), d, brute=True)
        assert loop[0] == 1.0 and loop[1] == 2.0, "x, z within the radius: Disk.Hit accepts the point 4 above the centre"
        assert tree[0] == 0.0, "... and the tree, whose box for the disk ends 1 above the centre, never asks"


def test_voxel_grid_walk_against_python_restatement():
    """VolumeGrid.Hit (Amanatides-Woo walk, accumulated t, entry nudge, face normals, binary64 wire test) and the
    VolumeScene shadow rule, oracle vs tests/py_restatement.py, bit for bit."""
    from yetanotherconsolegameengine_amd.scene import VolumeGrid
    rng = np.random.default_rng(5)
    cells = np.zeros((12, 9, 10, 2), np.int32)
    cells[:, :3, :, 0] = 1                                   # ground
    cells[..., 0] = np.where(rng.uniform(size=cells.shape[:3]) < 0.06, 2, cells[..., 0])          # floating blocks
    cells[2:5, 3:7, 4:6, 0] = 3                              # a wall
    palette = {1: Material(vec3(0.35, 0.6, 0.25), 0.0, 0.0, ZERO), 2: Material(vec3(0.7, 0.5, 0.3), 0.0, 0.0, ZERO),
               3: Material(vec3(0.6, 0.6, 0.65), 0.0, 0.0, ZERO)}
    s = Scene()
    s.IsVolumeScene = True
    s.Ambient.Color, s.Ambient.Intensity = vec3(1, 1, 1), 0.0
    s.Objects.append(VolumeGrid(cells, vec3(-6.0, 0.0, -12.0), vec3(1, 1, 1), lambda m, t: palette[m], True, 0.06, 9.0))
    s.Objects.append(VolumeGrid(cells[::-1].copy(), vec3(6.0, 0.0, -12.0), vec3(0.5, 0.5, 0.5), lambda m, t: palette[m], False))
    s.Lights.append(PointLight(vec3(40.0, 90.0, 30.0), vec3(1.0, 0.96, 0.88), 9000.0))
    pose = dict(pos=(0.3, 4.6, 1.5), yaw=0.1, pitch=-0.2, fov=60.0)
    _check_frames_against_restatement(s, 16, 6, pose, 2, {0, 1})


def test_mesh_triangle_test_against_python_restatement():
    """MeshBVH.TriHit (sign-normalised, |det|-scaled ranges, division on accept) and the mesh hit attributes through the
    oracle's BVH vs a brute-force loop over the triangles in tests/py_restatement.py, bit for bit, incl. the bounce and
    shadow rays that start ON the mesh."""
    from yetanotherconsolegameengine_amd.scene import Mesh
    pos, faces = scenes.make_torus_knot(22, 9)
    pos = (pos * f32(0.35) + np.array([0.0, 1.2, -3.0], f32)).astype(f32)
    tri = pos[faces].astype(f32)                              # [n, 3, 3]
    s = Scene()
    s.Ambient.Color, s.Ambient.Intensity = vec3(1, 1, 1), 0.1
    s.Add(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Solid(vec3(0.6, 0.6, 0.6)), 0.0, 0.0))
    s.Objects.append(Mesh(tri, Material(vec3(0.1, 0.2, 0.85), 0.0, 0.7, ZERO)))
    s.Lights.append(PointLight(vec3(0.5, 6.0, -1.0), vec3(1, 1, 1), 80.0))
    pose = dict(pos=(0.0, 1.3, 0.0), yaw=0.0, pitch=-0.05, fov=50.0)
    _check_frames_against_restatement(s, 16, 6, pose, 1, {0, 1})


def _oracle_sort(keys, idx):
    L = ob.lib()
    k = np.ascontiguousarray(keys, f32).copy(); i = np.ascontiguousarray(idx, np.int32).copy()
    L.orc_introsort(k.ctypes.data_as(C.POINTER(C.c_float)), i.ctypes.data_as(C.POINTER(C.c_int32)), len(k))
    return k, i


def test_introsort_against_python_restatement():
    """The oracle's Array.Sort (.NET introspective sort) vs the restatement in tests/py_restatement.py: same permutation,
    ties included, on random, heavily tied, descending and median-of-three-killer inputs (the last reach the heapsort)."""
    import py_restatement as pr
    rng = np.random.RandomState(1)
    cases = []
    for trial in range(120):
        n = int(rng.choice([2, 3, 5, 16, 17, 18, 33, 64, 100, 257, 1000]))
        mode = trial % 4
        k = (rng.uniform(-1, 1, n) if mode == 0 else rng.randint(0, 4, n).astype(float) if mode == 1 else
             np.sort(rng.uniform(-1, 1, n))[::-1] if mode == 2 else np.where(rng.uniform(size=n) < 0.5, 0.0, rng.randint(0, 3, n)))
        cases.append(np.asarray(k, f32))
    for n in (64, 200, 1000):                  # Musser's median-of-3 killer
        h = n // 2; a = [0] * n
        for i in range(1, h + 1):
            a[i - 1] = i if i % 2 == 1 else h + i - 1
            a[h + i - 1] = 2 * i
        cases.append(np.asarray(a, f32))
    for k in cases:
        i = np.arange(len(k), dtype=np.int32)
        ko, io = _oracle_sort(k, i)
        kp, ip = pr.dotnet_introsort(k, i)
        assert np.array_equal(ko, kp) and np.array_equal(io, ip), len(k)


def _rows(nodes):
    return np.array([(tuple(n[:3]), tuple(n[3:6]), n[6], n[7], n[8], n[9]) for n in nodes], dtype=ob.NODE_DTYPE)


@pytest.mark.parametrize("case", ["random", "lattice", "equal", "clustered"])
def test_both_bvh_builders_against_python_restatement(case):
    """Binned-SAH builders of BVH.cs (leaf 4, partition binned with the first / last item) and MeshBVH.cs (leaf 8),
    restated in tests/py_restatement.py: node boxes, child links, leaf ranges and the leaf index order must equal the
    oracle's.  Array.Sort on the fallback path is the restatement's own introsort (test_introsort_against_python_restatement)."""
    import py_restatement as pr
    from yetanotherconsolegameengine_amd.scene import Mesh
    rng = np.random.RandomState(3)
    if case == "random":
        c = rng.uniform(-4, 4, (300, 1, 3)); tris = (c + rng.normal(scale=0.15, size=(300, 3, 3))).astype(f32)
    elif case == "lattice":         # equal centroids per axis: degenerate bins, Array.Sort fallbacks
        g = np.stack(np.meshgrid(np.arange(7), np.arange(6), np.arange(3), indexing="ij"), -1).reshape(-1, 1, 3).astype(f32)
        tris = (g + f32([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0]])).astype(f32)
    elif case == "equal":
        tris = np.tile(f32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), (37, 1, 1))
    else:
        c = np.concatenate([rng.normal(scale=0.05, size=(120, 1, 3)), rng.normal(loc=5.0, scale=2.0, size=(60, 1, 3))])
        tris = (c + rng.normal(scale=0.02, size=(180, 3, 3))).astype(f32)
    # mesh flavour
    s = Scene(); s.Objects.append(Mesh(tris, Material(vec3(1, 1, 1))))
    with ob.OracleRenderer(s, 8, 4) as r:
        on, ol = r.accel(abi.ACCEL_MESH_NODES), r.accel(abi.ACCEL_MESH_LEAF_INDEX)
    b, c = pr.triangle_items(tris)
    root, nodes, leaves = pr.build_bvh(b, c, True, pr.dotnet_introsort)
    assert np.array_equal(_rows(nodes).view(np.uint8), on.view(np.uint8)) and np.array_equal(np.int32(leaves), ol)
    # scene flavour over spheres with the same centres
    s = Scene()
    cen = tris.mean(axis=1).astype(f32)
    for k in range(len(cen)):
        s.Add(Sphere(vec3(*cen[k]), 0.1 + 0.01 * (k % 7), Material(vec3(1, 1, 1))))
    with ob.OracleRenderer(s, 8, 4) as r:
        on, ol = r.accel(abi.ACCEL_SCENE_NODES), r.accel(abi.ACCEL_SCENE_LEAF_INDEX)
    cc = np.float32([o.Center for o in s.Objects]); rr = np.float32([[o.Radius] for o in s.Objects])
    mn, mx = (cc - rr).astype(f32), (cc + rr).astype(f32)
    root, nodes, leaves = pr.build_bvh(np.concatenate([mn, mx], 1), (f32(0.5) * (mn + mx)).astype(f32), False, pr.dotnet_introsort)
    assert np.array_equal(_rows(nodes).view(np.uint8), on.view(np.uint8)) and np.array_equal(np.int32(leaves), ol)


def test_bvh_traversal_order_and_work_counters_against_python_restatement():
    """BVH.Hit / MeshBVH.Hit (near child first, far child stacked, re-test on pop, leaf objects in order) over trees built
    by the restated builders: same pixels AND the same work counters (Scene.Hit calls, AABB evaluations, triangle tests,
    primitive tests) as the oracle for a whole frame - the counters are what SURVEY 8(d)'s algorithmic bytes are made of."""
    import py_restatement as pr
    from yetanotherconsolegameengine_amd.scene import Mesh
    rng = np.random.RandomState(9)
    pos, faces = scenes.make_torus_knot(18, 7)
    tri = (pos * f32(0.3) + np.array([0.4, 1.0, -2.8], f32)).astype(f32)[faces].astype(f32)
    s = Scene()
    s.Ambient.Color, s.Ambient.Intensity = vec3(1, 1, 1), 0.05
    s.Add(Plane(vec3(0, 0, 0), vec3(0, 1, 0), Solid(vec3(0.6, 0.6, 0.6)), 0.0, 0.0))
    for k in range(11):                                      # enough objects for a real scene-level tree (leaf size 4)
        s.Add(Sphere(vec3(*(rng.uniform(-2.5, 2.5), rng.uniform(0.2, 1.6), rng.uniform(-6.0, -2.0))), float(rng.uniform(0.15, 0.4)),
                     Material(vec3(*rng.uniform(0.2, 0.9, 3)), 0.0, 0.95 if k == 3 else 0.0, ZERO)))
    s.Add(Box(vec3(-2.4, 0.0, -3.4), vec3(-1.8, 0.7, -2.8), Solid(vec3(0.8, 0.5, 0.3)), 0.0, 0.0))
    s.Objects.append(Mesh(tri, Material(vec3(0.1, 0.2, 0.85), 0.0, 0.7, ZERO)))
    s.Add(XZRect(-0.5, 0.5, -3.0, -2.4, 2.4, Material(vec3(0, 0, 0), 0.0, 0.0, vec3(1.5, 1.5, 1.4)), 0.0, 0.0))
    s.Lights.append(PointLight(vec3(1.0, 5.0, 0.0), vec3(1, 1, 1), 70.0))
    pose = dict(pos=(0.0, 1.2, 0.5), yaw=0.0, pitch=-0.08, fov=55.0)
    with ob.OracleRenderer(s, 14, 5, 1, pose) as o:
        o.render(stages=0)
        bs = pr.BvhScene(s)
        assert np.array_equal(pr_rows(bs.nodes).view(np.uint8), o.accel(abi.ACCEL_SCENE_NODES).view(np.uint8))
        p = pr.render_frame(o.L, bs, o.hiW, o.hiH, pose, 1)
        for which, key in ((abi.BUF_CURRENT_HDR, "hdr"), (abi.BUF_G_NORMAL, "normal"), (abi.BUF_G_DEPTH, "depth")):
            assert np.array_equal(o.read(which).view(np.uint32), np.ascontiguousarray(p[key], f32).view(np.uint32)), key
        assert np.array_equal(o.read(abi.BUF_PRIM_ID), p["prim"])
        st = o.stats
        assert (int(st.n_rays), int(st.n_box), int(st.n_tri), int(st.n_prim)) == (bs.cnt.rays, bs.cnt.box, bs.cnt.tri, bs.cnt.prim)
        assert bs.cnt.tri > 0 and len(bs.nodes) > 3


def pr_rows(nodes):
    return _rows(nodes)


def test_center_block_highlight_is_unreachable():
    """VolumeGrid's "center block" highlight (VolumeGrid.cs:176-187) needs a query with |screenU - 0.5| <= 1e-6 AND |screenV - 0.5| <= 1e-6
    (IsCenterUV, :286-289, binary32).  The tracer passes uCenter = (px + 0.5f) / hiW, vCenter = (py + 0.5f) / hiH (RaytraceRenderer.cs:204-205)
    with hiH = fbH * 2 * ss (:86-87, :119): EVEN for every framebuffer height and supersampling factor, so no row has vCenter within 1e-6 of
    0.5 - checked here in binary32 for every even hiH up to 8 192 rows and a sample beyond; every other caller of Scene.Hit passes (0, 0)
    (VolumeScenes.cs:288, 313, 404, 411, 446, 525).  Hence neither the oracle nor the kernels model the (racy) shared state: it is never set."""
    f32 = np.float32
    half, win = f32(0.5), f32(0.000001)
    for hiH in list(range(2, 8194, 2)) + [20000, 65536, 131072, 400000]:
        py = np.arange(max(0, hiH // 2 - 2), min(hiH, hiH // 2 + 2), dtype=np.float32)          # the rows next to the middle: the only candidates
        v = ((py + half) / f32(hiH)).astype(np.float32)
        assert not (np.abs(v - half) <= win).any(), hiH
    # (an odd row count WOULD have such a row - which is why the parity of hiH is the whole argument)
    for hiH in (9, 45, 1079):
        py = np.arange(hiH, dtype=np.float32)
        assert (np.abs(((py + half) / f32(hiH)).astype(np.float32) - half) <= win).sum() == 1
