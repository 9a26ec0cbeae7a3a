// ycge_anyhit.hip.h — ORDER-FREE occlusion queries against a mesh: the shadow rays of a wavefront, breadth-first from ONE shared work list.
//
// A shadow query of a scene without transparent materials (and of a VolumeScene) only asks "is any triangle accepted in [tmin, tmax]?"
// (ComputeTransmittanceToLight, RaytraceRenderer.cs:757-781: every occluder has Transparency <= 0, the first hit returns zero).  Until the
// first accept `closest` IS tmax, so every slab test and every TriHit range test of MeshBVH.Hit (MeshBVH.cs:132-304) runs against the
// unchanged tmax, and the pop re-test `closest >= tNear` always passes (tNear <= the box's exit <= tmax).  The answer is therefore
//
//      OR over the leaves whose ancestors' boxes are all hit  ( OR over the leaf's triangles  TriHit(tri, tmin, tmax) )
//
// - a set expression: whatever order, grouping or lane evaluates the terms, the boolean is the reference's.  (The ordered walk of
// mesh_walk / coop_walk already stops at the first accept; this form also drops the ORDER.)  What the order cost: a wavefront's shadow
// batch took as many steps as its longest ray (round 2: 457 k wave-steps per frame on config 4 where perfectly packed lanes need 216 k),
// and a ray that grazes the mesh is a chain of 50-90 dependent fetches on one lane while the other 63 wait.
//
// Here the rays of a batch share one LIFO work list in LDS.  An item is (ray, node) or (ray, leaf pair-record); every round the top
// min(64, items) entries are taken, one per lane, whichever ray they belong to: a node lane fetches the GNode (both child boxes, one
// round trip), tests them with box_mesh's operations and pushes the hit children - a leaf child as one item per pair record - a record
// lane runs TriHit's tests for its two triangles against [tmin, tmax]; an accept marks the ray answered and its remaining items are
// dropped as they surface.  A batch then costs about max(items / 64, tree depth) rounds instead of its longest ray's steps, and a lone
// grazing ray advances on every open subtree at once.
//
// LDS: the per-lane traversal stacks are EMPTY while a query looks at a mesh object of a flat scene (traverse), so the list (768
// entries) and the rays' parameters (64 x 48 bytes) live in that area - no extra allocation.  The list cannot overflow:
//   * a popped node pushes at most MAXPUSH = 16 entries (two leaf children of 15 triangles: 8 records each; MeshBVH.cs:14 caps a leaf
//     at 8 triangles), a record none;
//   * a WIDE round (up to 64 pops) is taken only while items <= HIGH - MAXPUSH and is cut back to its top item when it would end
//     above HIGH;
//   * above that ONE item is popped per round - a depth-first dive, which holds at most depth + MAXPUSH entries above where it
//     started (<= HIGH), and the reference refuses meshes deeper than 64 (MeshBVH.cs:150; ycge_scene_upload) - so LIST = HIGH + 80.
// -DYCGE_BFS_LIST=<n> shrinks the list (tests: the narrow modes must give the same answers).
#pragma once

namespace ycge {

#ifndef YCGE_BFS_LIST
#define YCGE_BFS_LIST 768u
#endif
#define YCGE_BFS_MAXPUSH 16u
#define YCGE_BFS_DEPTH 64u
#define YCGE_BFS_HIGH (YCGE_BFS_LIST - YCGE_BFS_DEPTH - YCGE_BFS_MAXPUSH)
static_assert(YCGE_BFS_LIST * 4u <= YCGE_LDS_STACK_LEVELS * 64u * 8u / 2u, "the work list is the first half of the 64-lane stack area");
static_assert(64u * 48u <= YCGE_LDS_STACK_LEVELS * 64u * 8u / 2u, "the rays' parameters are its second half");
static_assert(YCGE_BFS_HIGH >= 2u * YCGE_BFS_MAXPUSH, "a wide round needs room for its top item's pushes");

// item: ray (6 bits) | leaf-record flag | 32-byte unit of the record in the mesh arena (25 bits: the reference encoding's own limit)
#define YCGE_BFS_ITEM(ray, is_rec, unit) (((uint32_t)(ray) << 26) | ((is_rec) ? 0x2000000u : 0u) | (uint32_t)(unit))

// inclusive prefix sum over the 64 lanes (values are small: no overflow); DPP row shifts + the two row broadcasts
__device__ __forceinline__ uint32_t wave_prefix_incl(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}

// entries a child reference stands for: a node is one item, a leaf one item per pair record
__device__ __forceinline__ uint32_t bfs_items_of(uint32_t ref) { return YCGE_REF_KIND(ref) == REF_MESH_NODE ? 1u : ((ref & 15u) + 1u) >> 1; }

// A lane's ray in its slot of the rays' area - (o, tmin)(1/d, tmax)(d, -) - and back.  Every lane parks its ray before the list is worked
// and takes it back afterwards: the slot IS where the work items read a ray from, and the twelve values are not live in registers across
// the rounds (explicit ds instructions: the compiler cannot forward the stored registers to the loads; the same device as g_shade_ctx).
__device__ __forceinline__ uint32_t bfs_ray_addr(uint32_t lane) { return (uint32_t)(uintptr_t)g_lds_stack64 + YCGE_LDS_STACK_LEVELS * 64u * 4u + lane * 48u; }
__device__ __forceinline__ void bfs_park_ray(F3 o, F3 inv, F3 d, float tmin, float tmax)
{
    const f32x4 v0 = {o.x, o.y, o.z, tmin}, v1 = {inv.x, inv.y, inv.z, tmax}, v2 = {d.x, d.y, d.z, 0.0f};
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32" : : "v"(bfs_ray_addr(threadIdx.x & 63u)), "v"(v0), "v"(v1), "v"(v2) : "memory");
}
__device__ __forceinline__ void bfs_unpark_ray(F3 &o, F3 &inv, F3 &d, float &tmin, float &tmax)
{
    f32x4 v0, v1, v2;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(bfs_ray_addr(threadIdx.x & 63u)) : "memory");
    o = f3(v0.x, v0.y, v0.z); tmin = v0.w; inv = f3(v1.x, v1.y, v1.z); tmax = v1.w; d = f3(v2.x, v2.y, v2.z);
}

// Every lane of the wavefront enters, its ray parked (bfs_park_ray); `mine`: this lane has an occlusion query whose mesh root box
// [tmin, tmax] is hit, `root_ref` its mesh's root.  Returns the mask of lanes whose query is answered "occluded".  The caller's stacks
// must be empty (they are the list).
__device__ __forceinline__ unsigned long long mesh_anyhit_bfs(const SceneDev &S, bool mine, uint32_t root_ref, Work &w)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *list = (uint32_t *)g_lds_stack64;
    const f32x4 *rays = (const f32x4 *)((uint8_t *)g_lds_stack64 + YCGE_LDS_STACK_LEVELS * 64u * 4u);
    const unsigned long long asking = __ballot(mine);
    // seeds: every asking ray's root (a root that is a leaf: its records) - at most 64 x 8 entries, below HIGH
    uint32_t occ;
    {
        const uint32_t c = mine ? bfs_items_of(root_ref) : 0u;
        const uint32_t incl = wave_prefix_incl(c);
        const uint32_t unit = (root_ref & 0x1ffffff0u) >> 4;
        const bool rec = YCGE_REF_KIND(root_ref) != REF_MESH_NODE;
        for (uint32_t k = 0; k < c; k++) list[incl - c + k] = YCGE_BFS_ITEM(lane, rec, unit + 3u * k);
        occ = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long done = 0ull;
    while (occ != 0u && (done & asking) != asking) {
        const bool wide = occ <= YCGE_BFS_HIGH - YCGE_BFS_MAXPUSH;
        uint32_t n = wide ? (occ < 64u ? occ : 64u) : 1u;
        const bool popped = lane < n;
        const uint32_t item = popped ? list[occ - 1u - lane] : 0u;
        const uint32_t ray = item >> 26;
        const bool act = popped && !((done >> ray) & 1ull);          // an answered ray's leftovers are dropped as they surface
        uint32_t c_far = 0u, c_near = 0u, r_far = 0u, r_near = 0u;
        bool hit = false;
        if (lane == 0u || act) w.steps++;                               // (lane 0 counts the rounds: the block's schedule cost)
        if (__any(act)) {
            const uint32_t unit = item & 0x1ffffffu;
            f32x4 a, b, c, e;
            f32x2 f;
            load_record72(S.mesh_arena, act ? unit << 5 : 0u, a, b, c, e, f);
            const f32x4 r0 = rays[ray * 3u + 0u];
            if (act && !(item & 0x2000000u)) {
                // MeshBVH.BoxHitFast on both child boxes, the operations of mesh_walk's node step with closest = tmax
                const f32x4 r1 = rays[ray * 3u + 1u];
                const f32x2 oxy = {r0.x, r0.y}, ozz = {r0.z, r0.z}, ixy = {r1.x, r1.y}, izz = {r1.z, r1.z};
                const bool sx = r1.x < 0.0f, sy = r1.y < 0.0f, sz = r1.z < 0.0f;
                const float q_tmin = r0.w, q_tmax = r1.w;
                const f32x2 t0 = (a.xy - oxy) * ixy, t1 = (a.zw - ozz) * izz, t2 = (b.xy - oxy) * ixy;
                const f32x2 t3 = (b.zw - oxy) * ixy, t4 = (c.xy - ozz) * izz, t5 = (c.zw - oxy) * ixy;
                const float ln = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(q_tmin, sx ? t2.x : t0.x), sy ? t2.y : t0.y), sz ? t1.y : t1.x);
                const float lx = __builtin_fminf(__builtin_fminf(__builtin_fminf(q_tmax, sx ? t0.x : t2.x), sy ? t0.y : t2.y), sz ? t1.x : t1.y);
                const float rn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(q_tmin, sx ? t5.x : t3.x), sy ? t5.y : t3.y), sz ? t4.y : t4.x);
                const float rx = __builtin_fminf(__builtin_fminf(__builtin_fminf(q_tmax, sx ? t3.x : t5.x), sy ? t3.y : t5.y), sz ? t4.x : t4.y);
                const bool hl = lx >= ln, hr = rx >= rn;
                const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
                // the nearer child on top (popped first): not needed for the answer, it finds an occluder sooner
                const bool left_near = ln < rn;
                r_near = left_near ? lref : rref; r_far = left_near ? rref : lref;
                c_near = (left_near ? hl : hr) ? bfs_items_of(r_near) : 0u;
                c_far = (left_near ? hr : hl) ? bfs_items_of(r_far) : 0u;
            } else if (act) {
                // MeshBVH.TriHit (MeshBVH.cs:239-304) for the record's two triangles against [tmin, tmax]: tri_pair_hit's operations up to
                // the range test; the division that yields t is not needed.  (An odd leaf's last record has an all-zero second slot: det = 0.)
                const f32x4 r2 = rays[ray * 3u + 2u];
                const f32x4 r1 = rays[ray * 3u + 1u];
                const float q_tmin = r0.w, q_tmax = r1.w;
                const f32x2 ax = a.xy, ay = a.zw, az = b.xy, e1x = b.zw, e1y = c.xy, e1z = c.zw, e2x = e.xy, e2y = e.zw, e2z = f;
                const f32x2 dx = {r2.x, r2.x}, dy = {r2.y, r2.y}, dz = {r2.z, r2.z};
                const f32x2 ox = {r0.x, r0.x}, oy = {r0.y, r0.y}, oz = {r0.z, r0.z};
                const f32x2 px = dy * e2z - dz * e2y;
                const f32x2 py = dz * e2x - dx * e2z;
                const f32x2 pz = dx * e2y - dy * e2x;
                const f32x2 det = e1x * px + e1y * py + e1z * pz;
                const f32x2 sxx = ox - ax, syy = oy - ay, szz = oz - az;
                const f32x2 u_num = sxx * px + syy * py + szz * pz;
                const f32x2 sgn = {det.x > 0.0f ? 1.0f : -1.0f, det.y > 0.0f ? 1.0f : -1.0f};
                const f32x2 det_abs = det * sgn;
                const f32x2 u_num_s = u_num * sgn;
                const f32x2 qx = syy * e1z - szz * e1y;
                const f32x2 qy = szz * e1x - sxx * e1z;
                const f32x2 qz = sxx * e1y - syy * e1x;
                const f32x2 v_num = dx * qx + dy * qy + dz * qz;
                const f32x2 v_num_s = v_num * sgn;
                const f32x2 uv_sum_s = u_num_s + v_num_s;
                const f32x2 t_num = e2x * qx + e2y * qy + e2z * qz;
                const f32x2 t_num_s = t_num * sgn;
                const f32x2 tmin2 = {q_tmin, q_tmin};
                const f32x2 t_min_scaled = tmin2 * det_abs;
                bool ok0 = !(det.x > -1e-8f && det.x < 1e-8f);
                ok0 &= !(u_num_s.x < 0.0f || u_num_s.x > det_abs.x);
                ok0 &= !(v_num_s.x < 0.0f || uv_sum_s.x > det_abs.x);
                ok0 &= !(t_num_s.x < t_min_scaled.x || t_num_s.x > q_tmax * det_abs.x);
                bool ok1 = !(det.y > -1e-8f && det.y < 1e-8f);
                ok1 &= !(u_num_s.y < 0.0f || u_num_s.y > det_abs.y);
                ok1 &= !(v_num_s.y < 0.0f || uv_sum_s.y > det_abs.y);
                ok1 &= !(t_num_s.y < t_min_scaled.y || t_num_s.y > q_tmax * det_abs.y);
                hit = ok0 || ok1;
            }
        }
        // answered rays (rare: once per ray)
        for (unsigned long long hm = __ballot(hit); hm != 0ull; hm &= hm - 1ull) {
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)ray, (int)__builtin_ctzll(hm));
            done |= 1ull << r;
        }
        // pushes: lane's entries are [far ... near], lanes in lane order above what is left of the list
        uint32_t cnt = c_far + c_near;
        uint32_t incl = wave_prefix_incl(cnt);
        uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (wide && occ - n + total > YCGE_BFS_HIGH) {
            // the round would end above HIGH: only its top item (lane 0) is taken, the other entries stay where they are
            n = 1u;
            if (lane != 0u) { cnt = 0u; c_far = c_near = 0u; }
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            incl = cnt;
        }
        const uint32_t at = occ - n + incl - cnt;
        {
            const uint32_t uf = (r_far & 0x1ffffff0u) >> 4, un = (r_near & 0x1ffffff0u) >> 4;
            const bool recf = YCGE_REF_KIND(r_far) != REF_MESH_NODE, recn = YCGE_REF_KIND(r_near) != REF_MESH_NODE;
            for (uint32_t k = 0; k < c_far; k++) list[at + k] = YCGE_BFS_ITEM(ray, recf, uf + 3u * k);
            for (uint32_t k = 0; k < c_near; k++) list[at + c_far + k] = YCGE_BFS_ITEM(ray, recn, un + 3u * k);
        }
        occ = occ - n + total;
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    return done;
}

} // namespace ycge
