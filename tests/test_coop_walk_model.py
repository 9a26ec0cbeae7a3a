"""CPU (no GPU): the wave-cooperative walk of csrc/ycge_coop.hip.h, emulated step for step in numpy binary32 over the REAL device
records (mesh arena + treelet region as the host lays them out, ycge_host_mesh_arena_treelets), against the lane-serial walk of
mesh_walk emulated the same way - and that one against the oracle's MeshBVH.Hit (reference: Objects/MeshBVH.cs:132-236).

What must hold for a ray to change from the serial form to the cooperative form at ANY step boundary: same next reference, same
stack contents (reference, entry distance), same closest / hit after every step group.  The emulation runs both forms on random and
adversarial rays (grazing, axis-parallel with exact-zero direction components, inside the mesh, bounded tmax, any-hit), switching
at a random step, and compares the final (t bits, triangle) and the stack after every cooperative step with the serial walk's stack
at the same point of the visit sequence."""
import ctypes as C

import numpy as np
import pytest

from yetanotherconsolegameengine_amd import abi, build

f32 = np.float32
NONE = 0xFFFFFFFF
MESH_NODE, MESH_LEAF = 2, 3
TL_BYTES_PER_UNIT = 256


def kind(r): return r >> 29


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return abi.load_library()


def make_arena(lib, tris):
    t9 = np.ascontiguousarray(tris.reshape(-1, 9), dtype=np.float32)
    n = len(t9)
    fn = lib.ycge_host_mesh_arena_treelets
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    root = np.zeros(1, np.uint32); tl = np.zeros(1, np.uint32)
    nbytes = fn(t9.ctypes.data, n, None, 0, root.ctypes.data, tl.ctypes.data)
    assert nbytes > 0
    arena = np.zeros(nbytes + 128, np.uint8)
    assert fn(t9.ctypes.data, n, arena.ctypes.data, nbytes, root.ctypes.data, tl.ctypes.data) == nbytes
    return arena, int(root[0]), int(tl[0])


class Ray:
    def __init__(self, o, d, tmin, tmax, anyhit=False):
        self.o = np.asarray(o, f32); self.d = np.asarray(d, f32)
        with np.errstate(divide="ignore"):
            self.inv = (f32(1.0) / self.d).astype(f32)
        self.s = self.inv < 0
        self.tmin = f32(tmin); self.tmax = f32(tmax); self.anyhit = anyhit


def fmax(a, b):     # IEEE maxNum (v_max_f32): a NaN operand loses
    a, b = f32(a), f32(b)
    if np.isnan(a): return b
    if np.isnan(b): return a
    return a if a > b else b


def fmin(a, b):
    a, b = f32(a), f32(b)
    if np.isnan(a): return b
    if np.isnan(b): return a
    return a if a < b else b


def slab(mn, mx, r, closest):
    """MeshBVH.BoxHitFast as box_mesh / coop_walk evaluate it: (hit, entry distance)."""
    with np.errstate(invalid="ignore", over="ignore"):
        en = [f32(f32((mx[a] if r.s[a] else mn[a]) - r.o[a]) * r.inv[a]) for a in range(3)]
        ex = [f32(f32((mn[a] if r.s[a] else mx[a]) - r.o[a]) * r.inv[a]) for a in range(3)]
    tn = fmax(fmax(fmax(r.tmin, en[0]), en[1]), en[2])
    tx = fmin(fmin(fmin(closest, ex[0]), ex[1]), ex[2])
    return bool(tx >= tn), tn


def tri_parts(T, sl, r):
    """TriHit's closest-independent part for slot sl of a pair record (17 floats view): (ok, t_num_s, det_abs, t)."""
    with np.errstate(all="ignore"):
        ax, ay, az, e1x, e1y, e1z, e2x, e2y, e2z = [f32(T[2 * c + sl]) for c in range(9)]
        dx, dy, dz = r.d; ox, oy, oz = r.o
        px = f32(f32(dy * e2z) - f32(dz * e2y)); py = f32(f32(dz * e2x) - f32(dx * e2z)); pz = f32(f32(dx * e2y) - f32(dy * e2x))
        det = f32(f32(f32(e1x * px) + f32(e1y * py)) + f32(e1z * pz))
        if det > f32(-1e-8) and det < f32(1e-8): return False, f32(0), f32(0), f32(0)
        sx, sy, sz = f32(ox - ax), f32(oy - ay), f32(oz - az)
        u_num = f32(f32(f32(sx * px) + f32(sy * py)) + f32(sz * pz))
        sgn = f32(1.0) if det > 0 else f32(-1.0)
        det_abs = f32(det * sgn); u_s = f32(u_num * sgn)
        if u_s < 0 or u_s > det_abs: return False, f32(0), f32(0), f32(0)
        qx = f32(f32(sy * e1z) - f32(sz * e1y)); qy = f32(f32(sz * e1x) - f32(sx * e1z)); qz = f32(f32(sx * e1y) - f32(sy * e1x))
        v_num = f32(f32(f32(dx * qx) + f32(dy * qy)) + f32(dz * qz))
        v_s = f32(v_num * sgn)
        if v_s < 0 or f32(u_s + v_s) > det_abs: return False, f32(0), f32(0), f32(0)
        t_num = f32(f32(f32(e2x * qx) + f32(e2y * qy)) + f32(e2z * qz))
        t_s = f32(t_num * sgn)
        if t_s < f32(r.tmin * det_abs): return False, f32(0), f32(0), f32(0)
        return True, t_s, det_abs, f32(t_num * f32(f32(1.0) / det))


class Walk:
    """State shared by both forms: cur, stack of (ref, tnear), closest, hit."""
    def __init__(self, arena, tl, r, root):
        self.A = arena; self.F = arena.view(np.float32); self.U = arena.view(np.uint32); self.tl = tl; self.r = r
        self.cur = root; self.stack = []; self.closest = r.tmax; self.hit = -1; self.steps = 0; self.visits = []

    def pop(self):
        self.cur = NONE
        while self.stack:
            ref, tn = self.stack.pop()
            if self.closest >= tn:
                self.cur = ref
                return

    # ---- the lane-serial form: one node or one triangle pair per step (mesh_walk)
    def serial_step(self):
        r = self.r; unit = (self.cur & 0x1FFFFFF0) >> 4
        self.steps += 1
        if kind(self.cur) == MESH_NODE:
            g = self.F[unit * 8: unit * 8 + 16]; gu = self.U[unit * 8: unit * 8 + 16]
            # GNode: lmin xyz, lmax_z | lmax_x, lmax_y, rmin_x, rmin_y | rmin_z, rmax_z, rmax_x, rmax_y | lref rref
            hl, ln = slab((g[0], g[1], g[2]), (g[4], g[5], g[3]), r, self.closest)
            hr, rn = slab((g[6], g[7], g[8]), (g[10], g[11], g[9]), r, self.closest)
            lref, rref = int(gu[12]), int(gu[13])
            self.visits.append(("n", unit))
            if hl and hr:
                lf = ln < rn
                self.stack.append((rref, rn) if lf else (lref, ln))
                self.cur = lref if lf else rref
            elif hl: self.cur = lref
            elif hr: self.cur = rref
            else: self.cur = NONE
        else:
            left = self.cur & 15
            T = self.F[unit * 8: unit * 8 + 18]
            for sl in range(min(2, left)):
                ok, t_s, da, t = tri_parts(T, sl, r)
                if ok and not (t_s > f32(self.closest * da)):
                    self.closest = t; self.hit = unit * 2 + sl
            self.visits.append(("t", unit))
            self.cur = (self.cur + ((3 << 4) - 2)) if left > 2 else NONE
            if r.anyhit and self.hit >= 0:
                self.cur = NONE; self.stack.clear()
        if self.cur == NONE: self.pop()

    # ---- the cooperative form (coop_walk): a treelet or a whole leaf per step
    def coop_step(self):
        r = self.r; unit = (self.cur & 0x1FFFFFF0) >> 4
        self.steps += 1
        if kind(self.cur) == MESH_NODE:
            base = (self.tl + unit * TL_BYTES_PER_UNIT) // 4
            H = LF = LK = 0
            tn = [f32(0)] * 16; ref = [0] * 16
            for b in range(14):
                s = self.F[base + 8 * b: base + 8 * b + 8]; su = self.U[base + 8 * b: base + 8 * b + 8]
                h, tn[b] = slab((s[0], s[1], s[2]), (s[3], s[4], s[5]), r, self.closest)
                ref[b] = int(su[6])
                if h and su[7] != 0: H |= 1 << b
                if kind(ref[b]) != MESH_NODE: LK |= 1 << b
            for b in range(0, 14, 2):
                if tn[b] < tn[b + 1]: LF |= 1 << b

            def pick(p):
                hl, hr, lf = (H >> p) & 1, (H >> (p + 1)) & 1, (LF >> p) & 1
                near = (p if lf else p + 1) if (hl and hr) else p if hl else p + 1 if hr else -1
                far = (p + 1 if lf else p) if (hl and hr) else -1
                return near, far
            n1, f1 = pick(0); ex = n1; fars = [f1]
            if n1 >= 0 and not (LK >> n1) & 1:
                n2, f2 = pick(2 + 2 * n1); ex = n2; fars.append(f2)
                if n2 >= 0 and not (LK >> n2) & 1:
                    n3, f3 = pick(2 + 2 * n2); ex = n3; fars.append(f3)
            for f in fars:
                if f >= 0: self.stack.append((ref[f], tn[f]))
            self.cur = ref[ex] if ex >= 0 else NONE
        else:
            left = self.cur & 15
            cands = []
            for gl in range(8):
                if 2 * gl >= left: break
                T = self.F[(unit + 3 * gl) * 8: (unit + 3 * gl) * 8 + 18]
                for sl in range(2):
                    if 2 * gl + sl >= left: continue
                    ok, t_s, da, t = tri_parts(T, sl, r)
                    if ok and not (t_s > f32(self.closest * da)): cands.append((t_s, da, t, (unit + 3 * gl) * 2 + sl))      # against the ENTRY closest
            for t_s, da, t, sub in cands:         # replay in leaf order against the running closest
                if not (t_s > f32(self.closest * da)):
                    self.closest = t; self.hit = sub
            self.cur = NONE
            if r.anyhit and self.hit >= 0: self.stack.clear()
        if self.cur == NONE: self.pop()


def run(arena, tl, root, r, switch_at):
    """serial for `switch_at` steps, cooperative from there (switch_at < 0: serial all the way)"""
    w = Walk(arena, tl, r, root)
    # Mesh.Hit: the root's own box first (flat_begin / traverse)
    Ff = arena.view(np.float32)
    while w.cur != NONE:
        if switch_at < 0 or w.steps < switch_at: w.serial_step()
        else: w.coop_step()
    return w


def _sphere_mesh(n_lat, n_lon, rng, jitter=0.0):
    th = np.linspace(0.05, np.pi - 0.05, n_lat); ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    P = np.array([[np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)] for t in th for p in ph], np.float32)
    P += rng.normal(0, jitter, P.shape).astype(np.float32)
    tris = []
    for i in range(n_lat - 1):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            tris += [[P[a], P[b], P[c]], [P[b], P[d], P[c]]]
    return np.array(tris, np.float32)


def _rays(rng, n):
    out = []
    for i in range(n):
        m = i % 6
        if m == 0:      # from outside towards the mesh
            o = rng.normal(0, 1, 3); o = 3.0 * o / np.linalg.norm(o); tgt = rng.uniform(-0.9, 0.9, 3); d = tgt - o
        elif m == 1:    # grazing
            o = np.array([-3.0, rng.uniform(-1.05, 1.05), rng.uniform(-0.2, 0.2)]); d = np.array([1.0, rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02)])
        elif m == 2:    # axis-parallel: exact zeros in the direction (reciprocal +-inf, NaN slab products)
            ax = rng.integers(0, 3); o = rng.uniform(-0.5, 0.5, 3); o[ax] = -3.0; d = np.zeros(3); d[ax] = 1.0
        elif m == 3:    # from inside
            o = rng.uniform(-0.3, 0.3, 3); d = rng.normal(0, 1, 3)
        elif m == 4:    # short shadow-like segment with any-hit
            o = rng.normal(0, 1, 3); o = 1.01 * o / np.linalg.norm(o); d = rng.normal(0, 1, 3)
        else:           # away from the mesh
            o = np.array([0.0, 3.0, 0.0]); d = np.array([rng.uniform(-1, 1), 1.0, rng.uniform(-1, 1)])
        d = d / np.linalg.norm(d)
        tmax = f32(3.4028234663852886e38) if m != 4 else f32(rng.uniform(0.2, 3.0))
        out.append(Ray(o, d, 0.001 if m != 4 else 1e-4, tmax, anyhit=(m == 4 and i % 12 == 4)))
    return out


@pytest.mark.parametrize("seed,n_lat,n_lon,jitter", [(1, 14, 20, 0.0), (2, 24, 36, 0.01), (3, 9, 9, 0.0)])
def test_cooperative_walk_equals_the_serial_walk_at_any_switch_point(lib, seed, n_lat, n_lon, jitter):
    rng = np.random.default_rng(seed)
    tris = _sphere_mesh(n_lat, n_lon, rng, jitter)
    # duplicated triangles: exact ties in t (the later-visited one must win in both forms)
    tris = np.concatenate([tris, tris[:: 7]])
    arena, root, tl = make_arena(lib, tris)
    assert tl > 0 and kind(root) == MESH_NODE
    n_switched = n_hits = 0
    longest = 0
    for r in _rays(rng, 240):
        ref = run(arena, tl, root, r, -1)
        longest = max(longest, ref.steps)
        for sw in {0, 1, int(rng.integers(0, max(1, ref.steps)))}:
            w = run(arena, tl, root, r, sw)
            n_switched += 1
            if r.anyhit:
                assert (w.hit >= 0) == (ref.hit >= 0)
            else:
                assert w.hit == ref.hit and np.float32(w.closest).view(np.uint32) == np.float32(ref.closest).view(np.uint32), (sw, w.hit, ref.hit, w.closest, ref.closest)
            assert w.steps <= ref.steps
        n_hits += ref.hit >= 0
    assert n_hits > 40 and longest > 20
    print(f"{n_switched} walks, {n_hits} of 240 rays hit, longest serial walk {longest} steps")


def test_treelet_region_mirrors_the_node_records(lib):
    """Every treelet slot holds its descendant's box and reference exactly as the GNode records do; slots under a leaf are invalid."""
    rng = np.random.default_rng(5)
    tris = _sphere_mesh(12, 16, rng, 0.005)
    arena, root, tl = make_arena(lib, tris)
    F = arena.view(np.float32); U = arena.view(np.uint32)

    def children(ref):
        unit = (ref & 0x1FFFFFF0) >> 4
        g = F[unit * 8: unit * 8 + 16]; gu = U[unit * 8: unit * 8 + 16]
        return [((g[0], g[1], g[2]), (g[4], g[5], g[3]), int(gu[12])), ((g[6], g[7], g[8]), (g[10], g[11], g[9]), int(gu[13]))]
    todo = [root]; n_nodes = 0
    while todo:
        ref = todo.pop(); n_nodes += 1
        unit = (ref & 0x1FFFFFF0) >> 4
        base = (tl + unit * TL_BYTES_PER_UNIT) // 4
        expect = [None] * 14

        def fill(b, ref_b, depth):
            for side, (mn, mx, cref) in enumerate(children(ref_b)):
                c = 2 * b + 2 + side
                expect[c] = (mn, mx, cref)
                if depth < 3 and kind(cref) == MESH_NODE: fill(c, cref, depth + 1)
        fill(-1, ref, 1)
        for b in range(14):
            s = F[base + 8 * b: base + 8 * b + 8]; su = U[base + 8 * b: base + 8 * b + 8]
            if expect[b] is None:
                assert su[7] == 0
            else:
                mn, mx, cref = expect[b]
                assert su[7] == 1 and int(su[6]) == cref and tuple(s[0:3]) == tuple(mn) and tuple(s[3:6]) == tuple(mx), (unit, b)
        for _, _, cref in children(ref):
            if kind(cref) == MESH_NODE: todo.append(cref)
    assert n_nodes > 30
