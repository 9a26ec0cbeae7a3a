"""CPU (no GPU): the order-free occlusion query of csrc/ycge_anyhit.hip.h (mesh_anyhit_bfs), emulated round for round in numpy
binary32 over the REAL device records, against the ordered any-hit walk of mesh_walk (emulated in test_coop_walk_model.py) and
against brute force over every leaf whose ancestors' boxes are hit.

Reference: ComputeTransmittanceToLight -> Scene.Hit -> MeshBVH.Hit (RaytraceRenderer.cs:757-781, Objects/MeshBVH.cs:132-304): with no
transparent material the query returns "occluded" iff ANY triangle is accepted in [tmin, tmax].

What must hold: (1) the boolean equals the ordered walk's for every ray, whatever the list size - the wide rounds, the rounds cut back
to their top item and the one-item dives must agree; (2) the list never holds more than LIST entries (the kernel does not check:
the bound is an argument, HIGH + depth + MAXPUSH, and this test is its measurement); (3) every entry belongs to an asking ray."""
import numpy as np
import pytest

from test_coop_walk_model import (MESH_NODE, NONE, Ray, Walk, _rays, _sphere_mesh, f32, kind, lib, make_arena, slab, tri_parts)  # noqa: F401

MAXPUSH, DEPTH = 16, 64


def items_of(ref):
    return 1 if kind(ref) == MESH_NODE else ((ref & 15) + 1) >> 1


def bfs_batch(arena, root, rays, LIST):
    """mesh_anyhit_bfs for up to 64 rays that all ask; returns (occluded flags, rounds, peak occupancy, rounds by mode)."""
    HIGH = LIST - DEPTH - MAXPUSH
    assert HIGH >= 2 * MAXPUSH
    F = arena.view(np.float32); U = arena.view(np.uint32)
    lst = []
    for lane, r in enumerate(rays):          # seeds, in lane order
        unit = (root & 0x1FFFFFF0) >> 4
        for k in range(items_of(root)):
            lst.append((lane, kind(root) != MESH_NODE, unit + 3 * k))
    done = [False] * len(rays)
    rounds = peak = 0
    modes = {"wide": 0, "cut": 0, "dive": 0}
    while lst and not all(done):
        occ = len(lst)
        peak = max(peak, occ)
        wide = occ <= HIGH - MAXPUSH
        n = min(64, occ) if wide else 1
        popped = [lst[occ - 1 - lane] for lane in range(n)]
        pushes = []
        for ray, is_rec, unit in popped:
            mine = []
            if not done[ray]:
                r = rays[ray]
                if not is_rec:
                    g = F[unit * 8: unit * 8 + 16]; gu = U[unit * 8: unit * 8 + 16]
                    hl, ln = slab((g[0], g[1], g[2]), (g[4], g[5], g[3]), r, r.tmax)
                    hr, rn = slab((g[6], g[7], g[8]), (g[10], g[11], g[9]), r, r.tmax)
                    lref, rref = int(gu[12]), int(gu[13])
                    left_near = ln < rn
                    near, far = (lref, rref) if left_near else (rref, lref)
                    h_near, h_far = (hl, hr) if left_near else (hr, hl)
                    for ref, h in ((far, h_far), (near, h_near)):
                        if h:
                            u = (ref & 0x1FFFFFF0) >> 4
                            mine += [(ray, kind(ref) != MESH_NODE, u + 3 * k) for k in range(items_of(ref))]
                else:
                    T = F[unit * 8: unit * 8 + 18]
                    for sl in range(2):
                        ok, t_s, da, _ = tri_parts(T, sl, r)
                        if ok and not (t_s > f32(r.tmax * da)):
                            done[ray] = True
            assert len(mine) <= MAXPUSH
            pushes.append(mine)
        total = sum(len(m) for m in pushes)
        if wide and occ - n + total > HIGH:
            n = 1; pushes = pushes[:1]; modes["cut"] += 1
        elif wide:
            modes["wide"] += 1
        else:
            modes["dive"] += 1
        del lst[occ - n:]
        for m in pushes:
            lst += m
        assert len(lst) <= LIST, (len(lst), LIST)
        rounds += 1
    return done, rounds, max(peak, len(lst)), modes


def brute(arena, root, r):
    """OR over the leaves whose ancestors' boxes are all hit of OR over their triangles: the set expression the answer is."""
    F = arena.view(np.float32); U = arena.view(np.uint32)
    todo = [root]
    while todo:
        ref = todo.pop(); unit = (ref & 0x1FFFFFF0) >> 4
        if kind(ref) == MESH_NODE:
            g = F[unit * 8: unit * 8 + 16]; gu = U[unit * 8: unit * 8 + 16]
            if slab((g[0], g[1], g[2]), (g[4], g[5], g[3]), r, r.tmax)[0]: todo.append(int(gu[12]))
            if slab((g[6], g[7], g[8]), (g[10], g[11], g[9]), r, r.tmax)[0]: todo.append(int(gu[13]))
        else:
            for k in range(items_of(ref)):
                T = F[(unit + 3 * k) * 8: (unit + 3 * k) * 8 + 18]
                for sl in range(2):
                    ok, t_s, da, _ = tri_parts(T, sl, r)
                    if ok and not (t_s > f32(r.tmax * da)):
                        return True
    return False


def shadow_rays(rng, n):
    out = []
    for i in range(n):
        m = i % 5
        if m == 0:          # from the surface outwards, short and long segments
            o = rng.normal(0, 1, 3); o = 1.01 * o / np.linalg.norm(o); d = rng.normal(0, 1, 3); tmax = rng.uniform(0.2, 3.0)
        elif m == 1:        # grazing the whole mesh: many open subtrees, no or a late occluder
            o = np.array([-3.0, rng.uniform(-1.05, 1.05), rng.uniform(-0.3, 0.3)]); d = np.array([1.0, rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02)]); tmax = 6.0
        elif m == 2:        # axis-parallel (reciprocal +-inf, NaN slab products)
            ax = rng.integers(0, 3); o = rng.uniform(-0.9, 0.9, 3); o[ax] = -3.0; d = np.zeros(3); d[ax] = 1.0; tmax = rng.uniform(1.0, 6.0)
        elif m == 3:        # from inside, towards a far light
            o = rng.uniform(-0.4, 0.4, 3); d = rng.normal(0, 1, 3); tmax = 30.0
        else:               # misses everything
            o = np.array([0.0, 3.0, 0.0]); d = np.array([rng.uniform(-1, 1), 1.0, rng.uniform(-1, 1)]); tmax = 10.0
        d = d / np.linalg.norm(d)
        out.append(Ray(o, d, 1e-4, f32(tmax), anyhit=True))
    return out


@pytest.mark.parametrize("seed,n_lat,n_lon,jitter", [(11, 14, 20, 0.0), (12, 24, 36, 0.01), (13, 9, 9, 0.0)])
def test_order_free_occlusion_equals_the_ordered_walk_for_every_list_size(lib, seed, n_lat, n_lon, jitter):
    rng = np.random.default_rng(seed)
    tris = _sphere_mesh(n_lat, n_lon, rng, jitter)
    tris = np.concatenate([tris, tris[::7]])
    arena, root, tl = make_arena(lib, tris)
    rays = shadow_rays(rng, 192)
    want = []
    longest = 0
    for r in rays:
        w = Walk(arena, tl, r, root)
        while w.cur != NONE: w.serial_step()
        want.append(w.hit >= 0)
        longest = max(longest, w.steps)
        assert brute(arena, root, r) == want[-1]
    assert 30 < sum(want) < 170
    seen = {"wide": 0, "cut": 0, "dive": 0}
    for LIST in (112, 160, 768):
        for b in range(0, len(rays), 64):
            batch = rays[b:b + 64]
            got, rounds, peak, modes = bfs_batch(arena, root, batch, LIST)
            # an unanswered ray's flag is final only when the list ran empty; the loop ends early when EVERY ray is answered
            assert got == want[b:b + 64], (LIST, b)
            assert peak <= LIST
            for k in seen: seen[k] += modes[k]
            if LIST == 768:
                assert rounds <= longest + 8, (rounds, longest)          # a batch costs no more rounds than its longest ray's steps
    assert seen["wide"] > 0 and seen["cut"] > 0 and seen["dive"] > 0, seen          # every mode of the list ran
    # a lone grazing ray: all its open subtrees at once
    lone = [r for r, h in zip(rays, want) if not h][:8]
    for r in lone:
        w = Walk(arena, tl, r, root)
        while w.cur != NONE: w.serial_step()
        got, rounds, peak, _ = bfs_batch(arena, root, [r], 768)
        assert got == [False] and rounds <= w.steps


def test_a_mesh_that_is_one_leaf(lib):
    """Mesh.Hit on a mesh whose root IS a leaf (<= 8 triangles): the seeds are its pair records."""
    rng = np.random.default_rng(3)
    quad = np.array([[[-1, 0, -1], [1, 0, -1], [1, 0, 1]], [[-1, 0, -1], [1, 0, 1], [-1, 0, 1]], [[-1, 0.5, -1], [1, 0.5, 1], [-1, 0.5, 1]]], np.float32)
    arena, root, tl = make_arena(lib, quad)
    assert kind(root) != MESH_NODE and items_of(root) == 2
    rays = [Ray([x, 2.0, z], [0.0, -1.0, 0.0], 1e-4, f32(tm), anyhit=True) for x, z, tm in ((0.2, 0.1, 5.0), (0.2, 0.1, 1.0), (3.0, 0.0, 5.0), (-0.5, 0.5, 1.6), (0.5, -0.5, 1.6))]
    want = [brute(arena, root, r) for r in rays]
    assert want == [True, False, False, True, False]
    got, rounds, peak, _ = bfs_batch(arena, root, rays, 768)
    assert got == want and rounds == 1
