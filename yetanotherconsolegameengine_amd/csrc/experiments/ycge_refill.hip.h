// experiments/ycge_refill.hip.h - EXPERIMENT (measured, rejected: DESIGN.md "measured and rejected"; compiled only with -DYCGE_EXPERIMENTS=1,
// i.e. into lib/var_experiments.so, which the parity test of k_trace_refill loads).
//
// In-wavefront query refill for the single-launch kernel (k_trace_refill, YCGE_REFILL=<steps>): a block's posted queries form a list, a
// lane that finishes one takes the next whichever pixel it belongs to, and the walk yields every `steps` mesh steps.  Bit-exact; slower
// (0.60 -> 0.69 ms on config 4): a step of a full, mixed wavefront costs 1.25 us against 0.93 for a sparse one.
#pragma once

namespace ycge {

// The FLAT query of traverse() as a resumable state machine (k_trace's refill mode): flat_begin is traverse's
// prologue, flat_advance runs at most `budget` mesh steps and returns true when the query is finished.  Object
// order, box tests, counters and every visit are traverse's; only WHEN a lane performs them differs.
struct FlatQuery {       // the walk's state; the ray itself (o, d, tmin) stays in the caller's RayQ
    F3 inv;
    float closest;
    int hit_prim, hit_sub, obj_i, n_top, mesh_prim;
    uint32_t cur;
    bool anyhit;
};
template <bool COUNT, class STK>
__device__ __forceinline__ void flat_begin(const SceneDev &S, const RayQ &q, STK &st, FlatQuery &fq, Work &w)
{
    fq.anyhit = !COUNT && q.anyhit;
    fq.closest = q.tmax;
    fq.hit_prim = -1; fq.hit_sub = 0;
    fq.cur = YCGE_REF_NONE_VALUE; fq.mesh_prim = -1;
    fq.obj_i = 0; fq.n_top = 0;
    fq.inv = f3(0.0f, 0.0f, 0.0f);
    st.reset();
    if (COUNT) w.rays++;
    if (S.scene_root_ref == YCGE_REF_NONE_VALUE) return;
    fq.inv = f3(1.0f / q.d.x, 1.0f / q.d.y, 1.0f / q.d.z);
    float tn;
    if (COUNT) w.box++;
    const bool root_hit = box_scene(S.scene_root_min[0], S.scene_root_min[1], S.scene_root_min[2], S.scene_root_max[0], S.scene_root_max[1],
                                    S.scene_root_max[2], q.o, fq.inv, q.tmin, fq.closest, tn);
    fq.n_top = root_hit ? (int)(YCGE_REF_PAYLOAD(S.scene_root_ref) & 7u) : 0;     // a missed root: no object is looked at (traverse)
}
template <bool COUNT, bool HAS_GRID, class STK>
__device__ __forceinline__ bool flat_advance(const SceneDev &S, const RayQ &q, STK &st, FlatQuery &fq, Work &w, int budget)
{
    const bool sx = fq.inv.x < 0.0f, sy = fq.inv.y < 0.0f, sz = fq.inv.z < 0.0f;
    const uint32_t leaf_start = YCGE_REF_PAYLOAD(S.scene_root_ref) >> 3;
    for (;;) {
        if (fq.cur == YCGE_REF_NONE_VALUE) {
            if (fq.obj_i >= fq.n_top || (fq.anyhit && fq.hit_prim >= 0)) return true;
            const int pi = (int)S.scene_leaf_prims[leaf_start + fq.obj_i];
            fq.obj_i++;
            const float4 *pp = (const float4 *)(S.prims + pi);
            const float4 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3];
            const int type = __float_as_int(q0.x);
            if (type == 9) {
                const uint32_t root_ref = __float_as_uint(q2.z);
                if (root_ref != YCGE_REF_NONE_VALUE) {
                    float tm;
                    if (COUNT) w.box++;
                    if (box_mesh(q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q.o, fq.inv, sx, sy, sz, q.tmin, fq.closest, tm)) { fq.cur = root_ref; fq.mesh_prim = pi; }
                }
            } else if (type == 10) {
                if (HAS_GRID) grid_dda<COUNT>(S, __float_as_int(q0.z), pi, q.o, q.d, fq.inv, q.tmin, fq.closest, fq.hit_prim, fq.hit_sub, w);
            } else {
                analytic_prim<COUNT>(q0, q1, q2, q3, type, pi, q.o, q.d, q.tmin, fq.closest, fq.hit_prim, fq.hit_sub, w);
            }
            continue;
        }
        mesh_walk<COUNT, true>(S, fq.cur, fq.mesh_prim, st, q.o, fq.inv, q.d, sx, sy, sz, q.tmin, fq.closest, fq.hit_prim, fq.hit_sub, w, budget, fq.anyhit);
        if (fq.cur != YCGE_REF_NONE_VALUE) return false;
    }
}

} // namespace ycge
