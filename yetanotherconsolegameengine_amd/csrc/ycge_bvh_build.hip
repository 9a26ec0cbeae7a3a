// ycge_bvh_build.hip - the scene-level BVH (Objects/BVH.cs:258-459) built ON THE DEVICE, for ycge_scene_update_objects
// (Scene.RebuildBVH when an entity moved, Scenes/Scene.cs:122-127): same topology, node numbering and leaf order as the
// reference's recursive builder - the host builder in ycge_accel.cpp is the sequential statement of the same thing and
// what the parity tests compare this against, node for node.
//
// One 1024-thread workgroup on one CU.  A node is split by ONE wavefront (items in chunks of 64); nodes wait in an LDS
// queue that the 16 wavefronts pop from, so subtrees are built side by side.  Everything order-dependent in the reference
// is reproduced exactly:
//   * binning (BVH.cs:322-352): counts are integer sums, bin boxes and centroid bounds are min / max - order-free (LDS
//     atomics on order-preserving integer keys).  The one thing an order could change is the sign of a zero bound, which
//     reaches only the SAH cost's sign of zero and no comparison;
//   * the SAH sweep (16 bins, strict '<' over axes then bins): running boxes and counts by scans over the bins (min / max / integer
//     sums), the cost of every split by one lane with the reference's operation order, first minimum in (axis, bin) order wins;
//   * the partition (BVH.cs:394-410) is the reference's two-pointer in-place loop, whose RESULT depends on the order the
//     items are visited in.  It has a closed form (tests/test_partition_closed_form.py checks it exhaustively): with
//     mid = #L, the front region = positions < mid plus position mid if it holds an R, the back region = the rest;
//       - a front L stays; the k-th front R (ascending) goes to slot e if k = 1, else to (position of the (k-1)-th back L,
//         counted from the end) - 1, and its slot - if below mid - receives the k-th back L;
//       - a back R moves down by one slot;
//     so one prefix count and one small index list give every item its slot in parallel;
//   * nodes are numbered in pre-order afterwards from subtree sizes (a node's children were created after it).
//   * the degenerate cases the reference hands to Array.Sort (no valid split, or a partition that leaves one side empty -
//     config 5's chunk lattice has them in every build) run the same introsort restatement as the host builder
//     (ycge_keysort.h), every lane of the wavefront in step on the one LDS slice, keys cached in LDS.
// `fallback` is left for a tree deeper than the reference's 128-entry stack: the host builder redoes it and reports the error.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ycge_device.h"
#include "ycge_keysort.h"
#include "ycge_math.h"

namespace ycge {

#define YCGE_BVH_DEV_BINS 16
#define YCGE_BVH_DEV_LEAF 4

struct BvhBuildNode {       // build-time record in global scratch; this workgroup is its only reader and writer
    int32_t start, count, depth, left;      // left < 0: leaf; right = left + 1
    int32_t inner, pre, ipre, pad;          // inner nodes in the subtree; pre-order index over all nodes / over inner nodes
    float mn[3], mx[3];
    int32_t pad2[2];
};
static_assert(sizeof(BvhBuildNode) == 64, "BvhBuildNode");

struct RefNodeDev { float mn[3], mx[3]; int32_t left, right, start, count; };     // = ycge::RefNode (ycge_accel.h)

__device__ __forceinline__ uint32_t fkey(float f) { const uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ float box_area(const float mn[3], const float mx[3])
{
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

struct BvhWaveBins { uint32_t cnt[3][YCGE_BVH_DEV_BINS]; uint32_t mn[3][YCGE_BVH_DEV_BINS][3], mx[3][YCGE_BVH_DEV_BINS][3]; };

struct BvhShared {
    uint16_t ord[YCGE_BVH_DEV_MAX_ITEMS], ord2[YCGE_BVH_DEV_MAX_ITEMS];
    uint16_t back_l[YCGE_BVH_DEV_MAX_ITEMS];                                        // per node range: the k-th back L, counted from the end (position in the range)
    unsigned long long queue[YCGE_BVH_DEV_MAX_ITEMS];                               // bit 63 valid | depth << 48 | count << 32 | start << 16 | node
    BvhWaveBins bins[16];
    float ikey[YCGE_BVH_DEV_MAX_ITEMS];                                             // Array.Sort case: the sort key of every item of the range, by item
    uint32_t q_head, q_tail, pending, n_nodes, fallback, max_depth, sorts;
};

// one wavefront splits node `id` = items ord[s .. s + cnt)
__device__ __forceinline__ bool bvh_split_node(BvhShared &sh, const float *__restrict__ items, const int n, BvhBuildNode *__restrict__ nodes, const int id,
                               const int s, const int cnt, const int depth)
{
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    BvhWaveBins &B = sh.bins[wave];
    const float *cpl[3] = {items + (size_t)6 * n, items + (size_t)7 * n, items + (size_t)8 * n};
    // centroid bounds (BVH.cs:300-312)
    float cmn[3] = {YCGE_INF, YCGE_INF, YCGE_INF}, cmx[3] = {-YCGE_INF, -YCGE_INF, -YCGE_INF};
    for (int i = lane; i < cnt; i += 64) {
        const int it = sh.ord[s + i];
        for (int a = 0; a < 3; a++) { const float c = cpl[a][it]; if (c < cmn[a]) cmn[a] = c; if (c > cmx[a]) cmx[a] = c; }
    }
    for (int a = 0; a < 3; a++)
        for (int o = 32; o > 0; o >>= 1) {
            const float lo = __shfl_xor(cmn[a], o, 64), hi = __shfl_xor(cmx[a], o, 64);
            if (lo < cmn[a]) cmn[a] = lo;
            if (hi > cmx[a]) cmx[a] = hi;
        }
    float ext[3], inv_ext[3];
    for (int a = 0; a < 3; a++) { ext[a] = cmx[a] - cmn[a]; inv_ext[a] = 1.0f / ext[a]; }
    // bins of all three axes in one pass over the items
    for (int w = lane; w < 3 * YCGE_BVH_DEV_BINS; w += 64) (&B.cnt[0][0])[w] = 0u;
    for (int w = lane; w < 9 * YCGE_BVH_DEV_BINS; w += 64) { (&B.mn[0][0][0])[w] = 0xffffffffu; (&B.mx[0][0][0])[w] = 0u; }      // min keys start at the top, max keys at the bottom
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < cnt; i += 64) {
        const int it = sh.ord[s + i];
        uint32_t kmn[3], kmx[3];
        for (int k = 0; k < 3; k++) { kmn[k] = fkey(items[(size_t)k * n + it]); kmx[k] = fkey(items[(size_t)(3 + k) * n + it]); }
        for (int a = 0; a < 3; a++) {
            if (!(ext[a] > 0.0f)) continue;
            int b = cs_f2i((cpl[a][it] - cmn[a]) * inv_ext[a] * (float)(YCGE_BVH_DEV_BINS - 1));
            if (b < 0) b = 0;
            if (b >= YCGE_BVH_DEV_BINS) b = YCGE_BVH_DEV_BINS - 1;
            atomicAdd(&B.cnt[a][b], 1u);
            for (int k = 0; k < 3; k++) { atomicMin(&B.mn[a][b][k], kmn[k]); atomicMax(&B.mx[a][b][k], kmx[k]); }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // SAH sweep (BVH.cs:354-383), one lane per (axis, bin): lanes 16 a + b.  The running boxes and counts of the reference's two
    // loops are inclusive scans over the 16 bins of an axis - min / max and integer sums, so their grouping is free - and every lane
    // then evaluates the ONE cost expression of its split, larea[b] * lc + rarea[b + 1] * rc, in the reference's operation order.
    // The reference keeps the first strict minimum in (axis, bin) order = the smallest lane among the lanes with the least cost.
    const int sa = lane >> 4, sb = lane & 15;
    const bool slot = lane < 48 && ext[sa < 3 ? sa : 0] > 0.0f;
    int c_pre = 0, c_suf = 0;
    float pmn[3] = {YCGE_INF, YCGE_INF, YCGE_INF}, pmx[3] = {-YCGE_INF, -YCGE_INF, -YCGE_INF};
    if (slot) {
        c_pre = (int)B.cnt[sa][sb];
        if (c_pre > 0)      // an empty bin joins no box (BVH.cs:358, 367)
            for (int k = 0; k < 3; k++) { pmn[k] = fkey_inv(B.mn[sa][sb][k]); pmx[k] = fkey_inv(B.mx[sa][sb][k]); }
    }
    c_suf = c_pre;
    float qmn[3] = {pmn[0], pmn[1], pmn[2]}, qmx[3] = {pmx[0], pmx[1], pmx[2]};
    for (int o = 1; o < 16; o <<= 1) {          // inclusive scans inside the 16-lane segment: prefix from the left, suffix from the right
        const int cu = __shfl_up(c_pre, o, 16), cd = __shfl_down(c_suf, o, 16);
        float umn[3], umx[3], dmn[3], dmx[3];
        for (int k = 0; k < 3; k++) { umn[k] = __shfl_up(pmn[k], o, 16); umx[k] = __shfl_up(pmx[k], o, 16); dmn[k] = __shfl_down(qmn[k], o, 16); dmx[k] = __shfl_down(qmx[k], o, 16); }
        if (sb >= o) { c_pre += cu; for (int k = 0; k < 3; k++) { if (umn[k] < pmn[k]) pmn[k] = umn[k]; if (umx[k] > pmx[k]) pmx[k] = umx[k]; } }
        if (sb + o < 16) { c_suf += cd; for (int k = 0; k < 3; k++) { if (dmn[k] < qmn[k]) qmn[k] = dmn[k]; if (dmx[k] > qmx[k]) qmx[k] = dmx[k]; } }
    }
    // split b: left = bins 0..b (this lane's prefix), right = bins b + 1..15 (the next lane's suffix)
    const int rc = __shfl_down(c_suf, 1, 16);
    float rmn[3], rmx[3];
    for (int k = 0; k < 3; k++) { rmn[k] = __shfl_down(qmn[k], 1, 16); rmx[k] = __shfl_down(qmx[k], 1, 16); }
    float my_cost = YCGE_INF;
    if (slot && sb < 15 && c_pre > 0 && rc > 0) {
        const float cost = box_area(pmn, pmx) * (float)c_pre + box_area(rmn, rmx) * (float)rc;
        if (cost < YCGE_INF) my_cost = cost;            // (+inf and NaN never beat the reference's initial +inf)
    }
    int my_lane = lane;
    for (int o = 32; o > 0; o >>= 1) {
        const float oc = __shfl_xor(my_cost, o, 64);
        const int ol = __shfl_xor(my_lane, o, 64);
        if (oc < my_cost || (oc == my_cost && ol < my_lane)) { my_cost = oc; my_lane = ol; }
    }
    const float best_cost = my_cost;
    int split_bin = -1, best_axis = 0;
    if (ext[1] > ext[0] && ext[1] >= ext[2]) best_axis = 1; else if (ext[2] > ext[0] && ext[2] >= ext[1]) best_axis = 2;      // BVH.cs:314-316
    if (best_cost < YCGE_INF) { split_bin = my_lane & 15; best_axis = my_lane >> 4; }
    const float *key = cpl[best_axis];
    int mid = 0;
    bool sort_it = split_bin < 0;
    if (!sort_it) {
        // partition (BVH.cs:394-410): bins re-derived from the first and the last item of the range
        const float origin = key[sh.ord[s]];
        const float extent = key[sh.ord[s + cnt - 1]] - origin;
        const float inv_extent = extent != 0.0f ? 1.0f / extent : 0.0f;
        const bool zero = !(inv_extent != 0.0f);
        auto is_left = [&](int pos) -> bool {
            const int b0 = zero ? 0 : cs_f2i((key[sh.ord[s + pos]] - origin) * inv_extent * (float)(YCGE_BVH_DEV_BINS - 1));
            return b0 <= split_bin;
        };
        int n_left = 0;
        for (int base = 0; base < cnt; base += 64) {
            const int pos = base + lane;
            const bool L = pos < cnt && is_left(pos);
            n_left += __popcll(__ballot(L));
        }
        mid = n_left;
        sort_it = n_left == 0 || n_left == cnt;          // BVH.cs:412-421: a side came out empty - the range is sorted AS THE LOOP LEFT IT
        if (n_left < cnt) {                                // (all left: the loop moved nothing)
            const int e = cnt - 1;
            const int front_hi = mid + (is_left(mid) ? 0 : 1);
            int run_l = 0;
            for (int base = 0; base < cnt; base += 64) {
                const int pos = base + lane;
                const bool in = pos < cnt, L = in && is_left(pos);
                const unsigned long long m = __ballot(L);
                const int pref_l = run_l + __popcll(m & ((1ull << lane) - 1ull));      // L's in [0, pos)
                if (in && pos >= front_hi && L) sh.back_l[s + (n_left - pref_l - 1)] = (uint16_t)pos;     // j - 1 = L's in (pos, e]
                run_l += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            run_l = 0;
            for (int base = 0; base < cnt; base += 64) {
                const int pos = base + lane;
                const bool in = pos < cnt, L = in && is_left(pos);
                const unsigned long long m = __ballot(L);
                const int pref_l = run_l + __popcll(m & ((1ull << lane) - 1ull));
                if (in) {
                    const uint16_t me = sh.ord[s + pos];
                    if (pos < front_hi) {
                        if (L) sh.ord2[s + pos] = me;
                        else {
                            const int k1 = pos - pref_l;                                    // k - 1 = R's in [0, pos)
                            const int dest = k1 == 0 ? e : (int)sh.back_l[s + k1 - 1] - 1;
                            sh.ord2[s + dest] = me;
                            if (pos < mid) sh.ord2[s + pos] = sh.ord[s + sh.back_l[s + k1]];
                        }
                    } else if (!L) sh.ord2[s + pos - 1] = me;
                }
                run_l += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < cnt; i += 64) sh.ord[s + i] = sh.ord2[s + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (sort_it) {
        // Array.Sort(items, start, count, axis comparer) + median split (BVH.cs:386-391, 414-421).  Every lane runs the same steps
        // on the same slice (one instruction at a time, so reads precede the writes of a step in every lane): no one-lane branch
        for (int i = lane; i < cnt; i += 64) { const int it = sh.ord[s + i]; sh.ikey[it] = key[it]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        KeySorter<uint16_t> sorter{sh.ord, sh.ikey};
        sorter.sort(s, cnt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        mid = cnt >> 1;
        if (lane == 0) atomicAdd(&sh.sorts, 1u);
    }
    // children
    int first = 0;
    if (lane == 0) {
        first = (int)atomicAdd(&sh.n_nodes, 2u);
        nodes[id].left = first;
        const int cs[2] = {s, s + mid}, cc[2] = {mid, cnt - mid};
        for (int k = 0; k < 2; k++) {
            BvhBuildNode &c = nodes[first + k];
            c.start = cs[k]; c.count = cc[k]; c.depth = depth + 1; c.left = -1; c.inner = 0; c.pre = 0; c.ipre = 0;
        }
        atomicMax(&sh.max_depth, (uint32_t)(depth + 1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the range's new order and the child records before the children are queued
    if (lane == 0) {
        const int cs[2] = {s, s + mid}, cc[2] = {mid, cnt - mid};
        for (int k = 0; k < 2; k++)
            if (cc[k] > YCGE_BVH_DEV_LEAF) {
                atomicAdd(&sh.pending, 1u);
                const uint32_t slot = atomicAdd(&sh.q_tail, 1u);
                const unsigned long long ent = (1ull << 63) | ((unsigned long long)(depth + 1) << 48) | ((unsigned long long)cc[k] << 32) |
                                               ((unsigned long long)cs[k] << 16) | (unsigned long long)(first + k);
                __hip_atomic_store(&sh.queue[slot], ent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    }
    return true;
}

// items: nine planes of n floats - box min x, y, z, box max x, y, z, centroid x, y, z (what Scene.RebuildBVH gets from
// Hittable.TryGetBounds, BVH.cs:32-53).  Outputs: ref_out = the reference-format nodes in pre-order (ycge_read_accel),
// gnodes_out = the traversal records of the inner nodes in their own pre-order, leaf_out = leafObjIndex, res.
__global__ __launch_bounds__(1024) void k_scene_bvh_build(const float *__restrict__ items, const int n, BvhBuildNode *__restrict__ nodes,
                                                          RefNodeDev *__restrict__ ref_out, GNode *__restrict__ gnodes_out,
                                                          uint32_t *__restrict__ leaf_out, BvhBuildResult *__restrict__ res, const int active_waves)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char bvh_lds[];      // above the 64 KB a static allocation may take
    BvhShared &sh = *reinterpret_cast<BvhShared *>(bvh_lds);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    for (int i = tid; i < n; i += 1024) sh.ord[i] = (uint16_t)i;
    for (int i = tid; i < YCGE_BVH_DEV_MAX_ITEMS; i += 1024) sh.queue[i] = 0ull;
    if (tid == 0) {
        sh.q_head = 0; sh.q_tail = 0; sh.pending = 0; sh.n_nodes = 1; sh.fallback = 0; sh.max_depth = 1; sh.sorts = 0;
        BvhBuildNode &r = nodes[0];
        r.start = 0; r.count = n; r.depth = 1; r.left = -1; r.inner = 0; r.pre = 0; r.ipre = 0;
        if (n > YCGE_BVH_DEV_LEAF) {
            sh.pending = 1; sh.q_tail = 1;
            sh.queue[0] = (1ull << 63) | (1ull << 48) | ((unsigned long long)n << 32);
        }
    }
    __syncthreads();
    // ---- the tree: every wavefront takes nodes off the queue until none is left or in flight
    // (lane 0 decides, the decision is broadcast: no loop and no wavefront-wide operation inside a one-lane branch)
    while ((tid >> 6) < active_waves) {
        __builtin_amdgcn_wave_barrier();
        uint32_t got = 0xffffffffu;                 // a queue slot, 0xfffffffe = all done, 0xffffffff = nothing yet
        if (lane == 0) {
            if (__hip_atomic_load(&sh.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) got = 0xfffffffeu;
            else {
                const uint32_t h = __hip_atomic_load(&sh.q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (h < __hip_atomic_load(&sh.q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    if (atomicCAS(&sh.q_head, h, h + 1u) == h) got = h;
                } else if (__hip_atomic_load(&sh.pending, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) got = 0xfffffffeu;
            }
        }
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (got == 0xfffffffeu) break;
        if (got == 0xffffffffu) { __builtin_amdgcn_s_sleep(2); continue; }
        unsigned long long ent;
        do ent = __hip_atomic_load(&sh.queue[got], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); while (!(ent >> 63));       // every lane, one address
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int id = (int)(ent & 0xffffu), s = (int)((ent >> 16) & 0xffffu), cnt = (int)((ent >> 32) & 0xffffu), depth = (int)((ent >> 48) & 0x7fffu);
        const bool ok = depth < 200 && bvh_split_node(sh, items, n, nodes, id, s, cnt, depth);
        if (lane == 0) {
            if (!ok) atomicExch(&sh.fallback, 1u);
            __hip_atomic_fetch_sub(&sh.pending, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int n_nodes = (int)sh.n_nodes, max_depth = (int)sh.max_depth;
    if (sh.fallback) {
        if (tid == 0) { res->fallback = 1; res->n_nodes = 0; res->n_inner = 0; res->max_depth = 0; res->root_ref = YCGE_REF_NONE_VALUE; }
        return;
    }
    // ---- leaf boxes (compare-assign in leaf order, BVH.cs:281-296), then inner boxes and subtree sizes bottom-up
    for (int i = tid; i < n_nodes; i += 1024) {
        BvhBuildNode &nd = nodes[i];
        if (nd.left >= 0) continue;
        float mn[3], mx[3];
        const int first = sh.ord[nd.start];
        for (int k = 0; k < 3; k++) { mn[k] = items[(size_t)k * n + first]; mx[k] = items[(size_t)(3 + k) * n + first]; }
        for (int j = 1; j < nd.count; j++) {
            const int it = sh.ord[nd.start + j];
            for (int k = 0; k < 3; k++) { const float v = items[(size_t)k * n + it]; if (v < mn[k]) mn[k] = v; }
            for (int k = 0; k < 3; k++) { const float v = items[(size_t)(3 + k) * n + it]; if (v > mx[k]) mx[k] = v; }
        }
        for (int k = 0; k < 3; k++) { nd.mn[k] = mn[k]; nd.mx[k] = mx[k]; }
        nd.inner = 0;
    }
    __syncthreads();
    for (int d = max_depth - 1; d >= 1; d--) {
        for (int i = tid; i < n_nodes; i += 1024) {
            BvhBuildNode &nd = nodes[i];
            if (nd.left < 0 || nd.depth != d) continue;
            const BvhBuildNode &L = nodes[nd.left], &R = nodes[nd.left + 1];
            for (int k = 0; k < 3; k++) { nd.mn[k] = cs_min(L.mn[k], R.mn[k]); nd.mx[k] = cs_max(L.mx[k], R.mx[k]); }      // MathF.Min / Max, BVH.cs:438-443
            nd.inner = 1 + L.inner + R.inner;
        }
        __syncthreads();
    }
    // ---- pre-order numbers top-down: left child next, right child after the left subtree (2 inner + 1 nodes)
    for (int d = 1; d < max_depth; d++) {
        for (int i = tid; i < n_nodes; i += 1024) {
            BvhBuildNode &nd = nodes[i];
            if (nd.left < 0 || nd.depth != d) continue;
            BvhBuildNode &L = nodes[nd.left], &R = nodes[nd.left + 1];
            L.pre = nd.pre + 1; L.ipre = nd.ipre + 1;
            R.pre = nd.pre + 1 + 2 * L.inner + 1; R.ipre = nd.ipre + 1 + L.inner;
        }
        __syncthreads();
    }
    for (int i = tid; i < n_nodes; i += 1024) {
        const BvhBuildNode &nd = nodes[i];
        RefNodeDev o;
        for (int k = 0; k < 3; k++) { o.mn[k] = nd.mn[k]; o.mx[k] = nd.mx[k]; }
        if (nd.left < 0) { o.left = o.right = -1; o.start = nd.start; o.count = nd.count; }
        else {
            const BvhBuildNode &L = nodes[nd.left], &R = nodes[nd.left + 1];
            o.left = L.pre; o.right = R.pre; o.start = 0; o.count = 0;
            GNode g;
            g.lmin_x = L.mn[0]; g.lmin_y = L.mn[1]; g.lmin_z = L.mn[2]; g.lmax_x = L.mx[0]; g.lmax_y = L.mx[1]; g.lmax_z = L.mx[2];
            g.rmin_x = R.mn[0]; g.rmin_y = R.mn[1]; g.rmin_z = R.mn[2]; g.rmax_x = R.mx[0]; g.rmax_y = R.mx[1]; g.rmax_z = R.mx[2];
            g.lref = L.left < 0 ? YCGE_REF(REF_SCENE_LEAF, ((uint32_t)L.start << 3) | (uint32_t)L.count) : YCGE_REF(REF_SCENE_NODE, (uint32_t)L.ipre);
            g.rref = R.left < 0 ? YCGE_REF(REF_SCENE_LEAF, ((uint32_t)R.start << 3) | (uint32_t)R.count) : YCGE_REF(REF_SCENE_NODE, (uint32_t)R.ipre);
            g.pad[0] = g.pad[1] = 0;
            gnodes_out[nd.ipre] = g;
        }
        ref_out[nd.pre] = o;
    }
    for (int i = tid; i < n; i += 1024) leaf_out[i] = sh.ord[i];
    if (tid == 0) {
        const BvhBuildNode &r = nodes[0];
        res->fallback = 0; res->n_nodes = n_nodes; res->n_inner = r.inner; res->max_depth = max_depth; res->sorts = sh.sorts;
        res->root_ref = r.left < 0 ? YCGE_REF(REF_SCENE_LEAF, ((uint32_t)r.start << 3) | (uint32_t)r.count) : YCGE_REF(REF_SCENE_NODE, 0u);
        for (int k = 0; k < 3; k++) { res->root_min[k] = r.mn[k]; res->root_max[k] = r.mx[k]; }
    }
}


// ---------------------------------------------------------------------------------- SceneDev::walk_nodes
// The walk tree of a world of voxel grids (ycge_device.h, SceneDev::walk_nodes) from whichever scene tree is installed: entry i < n is
// scene node i with its child references rewritten, and each leaf child of 2..7 objects gets the YCGE_WALK_LEAF_NODES entries from
// n + (2 i + side) * YCGE_WALK_LEAF_NODES on for its own nodes (a leaf of k objects uses k - 1).  One thread per (node, side): a leaf's
// nodes depend on its own objects only.
struct SolidBox { float lo[3], hi[3]; };
__device__ __forceinline__ SolidBox solid_of(const GPrim &g)
{
    const float inf = __builtin_huge_valf();
    SolidBox b;
    if (g.type == 10) { for (int a = 0; a < 3; a++) { b.lo[a] = g.p[a]; b.hi[a] = g.p[3 + a]; } }
    else { for (int a = 0; a < 3; a++) { b.lo[a] = -inf; b.hi[a] = inf; } }
    return b;
}
__device__ __forceinline__ uint32_t walk_ref_of(const GPrim &g, uint32_t prim) { return g.type == 10 ? YCGE_REF(REF_GRID, (uint32_t)g.ref) : YCGE_REF(REF_PRIM, prim); }
__global__ __launch_bounds__(256) void k_scene_walk(const GNode *__restrict__ nodes, int n_inner, const uint32_t *__restrict__ leaf_prims, const GPrim *__restrict__ prims,
                                                    GNode *__restrict__ walk)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= 2 * n_inner) return;
    const int i = t >> 1, side = t & 1;
    const GNode src = nodes[i];
    const uint32_t ref = side ? src.rref : src.lref;
    uint32_t out_ref;
    if (YCGE_REF_KIND(ref) == REF_SCENE_NODE) out_ref = YCGE_REF(REF_WALK_NODE, YCGE_REF_PAYLOAD(ref));
    else {
        const uint32_t pay = YCGE_REF_PAYLOAD(ref), start = pay >> 3, count = pay & 7u;
        // the leaf's objects in index order, grids without a solid voxel left out (the first one stays if nothing else does)
        uint32_t obj[7]; int m = 0;
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t pi = leaf_prims[start + k];
            const GPrim &g = prims[pi];
            if (g.type == 10 && g.p[3] < g.p[0]) continue;
            obj[m++] = pi;
        }
        if (m == 0) obj[m++] = leaf_prims[start];
        if (m == 1) out_ref = walk_ref_of(prims[obj[0]], obj[0]);
        else {
            // halve [a, b) until single: the node of a range is created before those of its halves, left half first
            const uint32_t base = (uint32_t)n_inner + (uint32_t)t * YCGE_WALK_LEAF_NODES;
            out_ref = YCGE_REF(REF_WALK_NODE, base | YCGE_WALK_IN_ORDER);
            int ra[8], rb[8], rslot[8], top = 0, next_slot = 1;
            ra[0] = 0; rb[0] = m; rslot[0] = 0; top = 1;
            while (top > 0) {
                top--;
                const int a = ra[top], b = rb[top], slot = rslot[top], mid = a + (b - a + 1) / 2;
                const float inf = __builtin_huge_valf();
                SolidBox L = {{inf, inf, inf}, {-inf, -inf, -inf}}, R = L;
                for (int k = a; k < b; k++) {
                    const SolidBox sb = solid_of(prims[obj[k]]);
                    SolidBox &u = k < mid ? L : R;
                    if (sb.hi[0] < sb.lo[0]) continue;
                    for (int ax = 0; ax < 3; ax++) { u.lo[ax] = fminf(u.lo[ax], sb.lo[ax]); u.hi[ax] = fmaxf(u.hi[ax], sb.hi[ax]); }
                }
                GNode g;
                g.lmin_x = L.lo[0]; g.lmin_y = L.lo[1]; g.lmin_z = L.lo[2]; g.lmax_x = L.hi[0]; g.lmax_y = L.hi[1]; g.lmax_z = L.hi[2];
                g.rmin_x = R.lo[0]; g.rmin_y = R.lo[1]; g.rmin_z = R.lo[2]; g.rmax_x = R.hi[0]; g.rmax_y = R.hi[1]; g.rmax_z = R.hi[2];
                if (mid - a == 1) g.lref = walk_ref_of(prims[obj[a]], obj[a]);
                else { g.lref = YCGE_REF(REF_WALK_NODE, (base + (uint32_t)next_slot) | YCGE_WALK_IN_ORDER); ra[top] = a; rb[top] = mid; rslot[top] = next_slot++; top++; }
                if (b - mid == 1) g.rref = walk_ref_of(prims[obj[mid]], obj[mid]);
                else { g.rref = YCGE_REF(REF_WALK_NODE, (base + (uint32_t)next_slot) | YCGE_WALK_IN_ORDER); ra[top] = mid; rb[top] = b; rslot[top] = next_slot++; top++; }
                g.pad[0] = 0u; g.pad[1] = 0u;
                walk[base + (uint32_t)slot] = g;
            }
        }
    }
    // this side of the scene node's copy: planes and the other reference are written by the two threads of the node between them
    GNode *dst = walk + i;
    if (side == 0) { dst->lmin_x = src.lmin_x; dst->lmin_y = src.lmin_y; dst->lmin_z = src.lmin_z; dst->lmax_z = src.lmax_z; dst->lmax_x = src.lmax_x; dst->lmax_y = src.lmax_y; dst->lref = out_ref; dst->pad[0] = 0u; }
    else { dst->rmin_x = src.rmin_x; dst->rmin_y = src.rmin_y; dst->rmin_z = src.rmin_z; dst->rmax_z = src.rmax_z; dst->rmax_x = src.rmax_x; dst->rmax_y = src.rmax_y; dst->rref = out_ref; dst->pad[1] = 0u; }
}

} // namespace ycge

extern "C" {

// walk: n_inner * (1 + 2 * YCGE_WALK_LEAF_NODES) records of 64 bytes (SceneDev::walk_nodes)
int ycge_launch_scene_walk(const void *nodes, int n_inner, const uint32_t *leaf_prims, const void *prims, void *walk, hipStream_t stream)
{
    if (n_inner <= 0) return 0;
    hipLaunchKernelGGL(ycge::k_scene_walk, dim3((unsigned)((2 * n_inner + 255) / 256)), dim3(256), 0, stream, (const ycge::GNode *)nodes, n_inner, leaf_prims, (const ycge::GPrim *)prims,
                       (ycge::GNode *)walk);
    return (int)hipGetLastError();
}

size_t ycge_bvh_build_scratch_bytes(int n) { return (size_t)(2 * n + 2) * sizeof(ycge::BvhBuildNode); }

// n items (1 .. YCGE_BVH_DEV_MAX_ITEMS); ref_out holds 2 n - 1 records of 40 bytes, gnodes_out n, leaf_out n
int ycge_launch_scene_bvh_build(const float *items, int n, void *scratch, void *ref_out, void *gnodes_out, uint32_t *leaf_out, void *result,
                                int active_waves, hipStream_t stream)
{
    if (n < 1 || n > YCGE_BVH_DEV_MAX_ITEMS) return (int)hipErrorInvalidValue;
    if (active_waves < 1 || active_waves > 16) active_waves = 16;
    static bool lds_set = false;
    if (!lds_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ycge::k_scene_bvh_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ycge::BvhShared));
        if (e != hipSuccess) return (int)e;
        lds_set = true;
    }
    hipLaunchKernelGGL(ycge::k_scene_bvh_build, dim3(1), dim3(1024), sizeof(ycge::BvhShared), stream, items, n, (ycge::BvhBuildNode *)scratch, (ycge::RefNodeDev *)ref_out,
                       (ycge::GNode *)gnodes_out, leaf_out, (ycge::BvhBuildResult *)result, active_waves);
    return (int)hipGetLastError();
}


static bool staged_copy_out(void *stage, void *dst, const void *src, size_t bytes)       // device -> page-locked staging -> caller memory (a CPU copy)
{
    if (hipMemcpy(stage, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (size_t i = 0; i < bytes; i++) ((unsigned char *)dst)[i] = ((const unsigned char *)stage)[i];
    return true;
}

// test hook (no context): the kernel on caller-supplied boxes - bounds = n x 6 (min xyz, max xyz), centroids = n x 3.  nodes_out
// holds 2 n records of 40 bytes (reference format, pre-order), leaf_out n, result_out 16 words (BvhBuildResult), build_out (optional)
// 2 n build records of 64 bytes in creation order.  Returns the node count, -1 on a device error, -2 when the kernel fell back.
int ycge_debug_device_bvh(const float *bounds, const float *centroids, int32_t n, void *nodes_out, int32_t *leaf_out, uint32_t *result_out, void *build_out)
{
    if (n < 1 || n > YCGE_BVH_DEV_MAX_ITEMS || !bounds || !centroids || !nodes_out || !leaf_out || !result_out) return -1;
    float *planes = (float *)malloc((size_t)9 * n * sizeof(float));
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { planes[(size_t)a * n + i] = bounds[6 * i + a]; planes[(size_t)(3 + a) * n + i] = bounds[6 * i + 3 + a]; planes[(size_t)(6 + a) * n + i] = centroids[3 * i + a]; }
    float *d_items = nullptr; void *d_scratch = nullptr, *d_ref = nullptr, *d_g = nullptr, *d_res = nullptr, *stage = nullptr; uint32_t *d_leaf = nullptr;
    int rc = -1;
    struct Out { void *stage; bool operator()(void *dst, const void *src, size_t bytes) const { return staged_copy_out(stage, dst, src, bytes); } } out{nullptr};
    if (hipMalloc((void **)&d_items, (size_t)9 * n * 4) == hipSuccess && hipMalloc(&d_scratch, ycge_bvh_build_scratch_bytes(n)) == hipSuccess &&
        hipMalloc(&d_ref, (size_t)2 * n * 40) == hipSuccess && hipMalloc(&d_g, (size_t)n * 64 + 64) == hipSuccess && hipMalloc(&d_res, 64) == hipSuccess &&
        hipMalloc((void **)&d_leaf, (size_t)n * 4) == hipSuccess && hipMemcpy(d_items, planes, (size_t)9 * n * 4, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemset(d_scratch, 0, ycge_bvh_build_scratch_bytes(n)) == hipSuccess &&
        ycge_launch_scene_bvh_build(d_items, n, d_scratch, d_ref, d_g, d_leaf, d_res, getenv("YCGE_BVH_WAVES") ? atoi(getenv("YCGE_BVH_WAVES")) : 16, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess &&
        hipHostMalloc(&stage, (size_t)2 * n * 64 + 64, hipHostMallocDefault) == hipSuccess && ((out.stage = stage), out(result_out, d_res, 64))) {
        // (every read-back through page-locked staging of this function's own and a CPU copy: the device writes no caller memory - DESIGN section 6)
        const ycge::BvhBuildResult *r = (const ycge::BvhBuildResult *)result_out;
        if (build_out) (void)out(build_out, d_scratch, (size_t)2 * n * 64);
        if (r->fallback) rc = -2;
        else if (out(nodes_out, d_ref, (size_t)r->n_nodes * 40) && out(leaf_out, d_leaf, (size_t)n * 4)) rc = r->n_nodes;
    }
    if (stage) (void)hipHostFree(stage);
    (void)hipFree(d_items); (void)hipFree(d_scratch); (void)hipFree(d_ref); (void)hipFree(d_g); (void)hipFree(d_res); (void)hipFree(d_leaf);
    free(planes);
    return rc;
}

}
