// ycge_coop.hip.h — the WAVE-COOPERATIVE walk of MeshBVH.Hit (MeshBVH.cs:132-236) for the sparse end of a wavefront's query batch.
//
// A query is a serial chain: one node visit or one triangle-pair test per memory round trip, hundreds of them for a ray that grazes
// the mesh, on ONE lane, while the other 63 lanes of its wavefront have finished and wait.  Those chains are the tail of every block
// and the end of the frame.  When at most YCGE_COOP_RAYS lanes of a wavefront are still walking, mesh_walk hands
// them over: every such ray gets a GROUP of 16 lanes, and a step of the group does what up to three steps of the lane did:
//
//  * NODE step: the 14 lanes of the group fetch the TREELET of the current node (GTreeSlot, ycge_device.h: the boxes and references
//    of its 2 children, 4 grandchildren, 8 great-grandchildren - one 32-byte slot per lane, ONE round trip) and test the fourteen
//    boxes side by side against [tmin, closest].  `closest` cannot change between the visits of internal nodes (only a triangle
//    lowers it), so these are exactly the slab tests the reference makes on its way down, and the near-first descent through up to
//    three levels is bit logic over two ballots: per pair (hit-left, hit-right, left-is-nearer) -> the child to visit, the child to
//    stack.  The far children are stacked by their own lanes in descent order (the reference's push order), the walk continues at
//    the node it left the treelet by, at a leaf, or - both children missed - at the stack.
//  * LEAF step: a leaf is at most 8 triangles (MeshBVH.cs:14) = 4 pair records (the format allows 15 = 8 records), fetched by as many lanes in one round trip.  Each lane runs
//    TriHit's closest-INDEPENDENT part for its two triangles (determinant, barycentric numerators, the tmin side of the range test);
//    a triangle that also passes the closest side against the leaf's ENTRY closest is a candidate - a superset of what the serial
//    loop accepts, because closest only shrinks.  Candidates are then replayed in leaf order against the running closest with
//    TriHit's own comparison `tNum * sgn <= closest * |det|` (MeshBVH.cs:291-297): one pass per candidate LANE, and almost always
//    there is at most one.
//
// Same visit order, same comparisons on the same operands, same stack contents (reference, entry distance) as the lane-serial walk:
// a ray may change from one form to the other at any step boundary, and does.  Per-lane state lives in LDS while the groups work
// (CoopSlots), the traversal stack stays the owner lane's own column.
#pragma once

namespace ycge {

#ifndef YCGE_COOP_GROUP
#define YCGE_COOP_GROUP 16          // lanes per ray: 16 (the three-level treelet, 14 box lanes) or 8 (its first two levels, 6 box lanes: twice the rays)
#endif
#define YCGE_COOP_LEVELS (YCGE_COOP_GROUP == 16 ? 3 : 2)
#define YCGE_COOP_BOX_LANES (YCGE_COOP_GROUP == 16 ? 14u : 6u)
#define YCGE_COOP_MASK ((1u << YCGE_COOP_BOX_LANES) - 1u)
#ifndef YCGE_COOP_RAYS
#define YCGE_COOP_RAYS (64 / YCGE_COOP_GROUP)      // rays a wavefront walks cooperatively = its groups; handed over when no more lanes than this still walk
#endif
static_assert(YCGE_COOP_GROUP == 16 || YCGE_COOP_GROUP == 8, "group size");
static_assert(YCGE_COOP_RAYS * YCGE_COOP_GROUP <= 64, "every handed-over ray needs a group");
// hand-over records, 16 words per ray, 4 rays per wavefront, up to 4 wavefronts per workgroup:
// [0..2] o  [3..5] d  [6..8] 1/d  [9] tmin  [10] closest  [11] cur  [12] sp  [13] hit_sub  [14] hit flag  [15] owner thread / steps back
static __shared__ __attribute__((aligned(16))) uint32_t g_coop_slots[4 * YCGE_COOP_RAYS * 16];

// the fetch of a cooperative step: node lanes take the 32 bytes of their treelet slot (a, b), leaf lanes the 72 bytes of their pair
// record (a, b, c, e, f); ONE wait.  `leaf_mask` = lanes that need the last three loads.
__device__ __forceinline__ void coop_fetch(const uint8_t *base, uint32_t byte_offset, unsigned long long leaf_mask, f32x4 &a, f32x4 &b, f32x4 &c,
                                           f32x4 &e, f32x2 &f)
{
    // (c, e, f of a lane outside leaf_mask keep whatever the registers held: every use of them is masked by the lane's leaf_lane flag)
    unsigned long long saved;
    asm volatile("global_load_dwordx4 %0, %6, %7\n\t"
                 "global_load_dwordx4 %1, %6, %7 offset:16\n\t"
                 "s_and_saveexec_b64 %5, %8\n\t"
                 "global_load_dwordx4 %2, %6, %7 offset:32\n\t"
                 "global_load_dwordx4 %3, %6, %7 offset:48\n\t"
                 "global_load_dwordx2 %4, %6, %7 offset:64\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e), "=&v"(f), "=&s"(saved)
                 : "v"(byte_offset), "s"(base), "s"(leaf_mask)
                 : "memory", "scc");
}

// Runs every lane's pending mesh walk (cur != none on at most YCGE_COOP_RAYS lanes) to its end.  Entered by the WHOLE wavefront.
template <class STK>
__device__ __forceinline__ void coop_walk(const SceneDev &S, uint32_t &cur, int mesh_prim, STK &st, F3 o, F3 inv, F3 d, float tmin, float &closest,
                                          int &hit_prim, int &hit_sub, Work &w, bool anyhit)
{
    const uint32_t lane = threadIdx.x & 63u, g = lane / YCGE_COOP_GROUP, gl = lane % YCGE_COOP_GROUP;
    uint32_t *slots = g_coop_slots + (threadIdx.x >> 6) * (YCGE_COOP_RAYS * 16);
    const bool mine = cur != YCGE_REF_NONE_VALUE;
    const unsigned long long live = __ballot(mine);
    const uint32_t n_live = (uint32_t)__popcll(live);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
    if (mine) {
        uint32_t *s = slots + rank * 16;
        s[0] = __float_as_uint(o.x); s[1] = __float_as_uint(o.y); s[2] = __float_as_uint(o.z);
        s[3] = __float_as_uint(d.x); s[4] = __float_as_uint(d.y); s[5] = __float_as_uint(d.z);
        s[6] = __float_as_uint(inv.x); s[7] = __float_as_uint(inv.y); s[8] = __float_as_uint(inv.z);
        s[9] = __float_as_uint(tmin); s[10] = __float_as_uint(closest); s[11] = cur; s[12] = (uint32_t)st.sp;
        s[13] = 0u; s[14] = anyhit ? 2u : 0u; s[15] = threadIdx.x;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the group's ray (replicated in its 16 lanes)
    const uint32_t *s = slots + (g < n_live ? g : 0u) * 16;
    const float ox = __uint_as_float(s[0]), oy = __uint_as_float(s[1]), oz = __uint_as_float(s[2]);
    const float dx = __uint_as_float(s[3]), dy = __uint_as_float(s[4]), dz = __uint_as_float(s[5]);
    const float ix = __uint_as_float(s[6]), iy = __uint_as_float(s[7]), iz = __uint_as_float(s[8]);
    const float c_tmin = __uint_as_float(s[9]);
    float c_closest = __uint_as_float(s[10]);
    uint32_t c_cur = g < n_live ? s[11] : YCGE_REF_NONE_VALUE;
    int c_sp = (int)s[12];
    uint32_t c_hit_sub = 0xffffffffu;             // triangle of the hit found here (record unit << 1 | slot), all ones = none yet
    const bool c_anyhit = (s[14] & 2u) != 0u;
    const uint32_t owner = s[15];
    uint32_t c_steps = 0u;
    const bool sx = ix < 0.0f, sy = iy < 0.0f, sz = iz < 0.0f;
    const uint32_t sh = g * YCGE_COOP_GROUP;
    // The near-first descent through the treelet as mask logic: the wavefront-wide 64-bit masks (hit, nearer, internal) live in scalar
    // registers and serve all four groups at once; a lane looks at ITS group through three constants - the slots above it on its path
    // (parent, grandparent), the slots below it in heap order, its group's 14 slots - in the half of the mask its group sits in.
    const uint32_t sh32 = sh & 31u;                                     // the group's place in its 32-bit half
    const bool hi_half = lane >= 32u;
    const uint32_t p2 = gl >= 2u ? (gl - 2u) >> 1 : 0u, p1 = gl >= 6u ? (p2 - 2u) >> 1 : 0u;
    const uint32_t anc32 = (gl >= 2u && gl < (uint32_t)YCGE_TL_SLOTS ? (1u << (sh32 + p2)) : 0u) | (gl >= 6u && gl < (uint32_t)YCGE_TL_SLOTS ? (1u << (sh32 + p1)) : 0u);
    const uint32_t grp32 = YCGE_COOP_MASK << sh32;
    const uint32_t below32 = ((1u << gl) - 1u) << sh32;
    // The entry a pop will take is asked for as soon as it is known (after the pushes of a node step, after a pop) and looked at
    // when the walk gets there: the LDS round trip is off the chain.  (LDS operations of a wavefront execute in order: a read
    // issued after the far children's writes sees them.)
    uint2 top = st.read_early(owner, c_sp - 1);
#if defined(YCGE_DBG_COOPSTAT)
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t dbg_iters = 0, dbg_node = 0, dbg_leaf = 0;
    unsigned long long dbg_fetch = 0, dbg_c0 = __builtin_amdgcn_s_memtime(), dbg_tnode = 0, dbg_tleaf = 0, dbg_tpop = 0, dbg_thead = 0, dbg_mark = dbg_c0;
#define YCGE_DBG_SECTION(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - dbg_mark; dbg_mark = t_; } while (0)
#else
#define YCGE_DBG_SECTION(acc) do { } while (0)
#endif

    for (;;) {
        const bool g_act = c_cur != YCGE_REF_NONE_VALUE;
        if (!__any(g_act)) break;
        const bool at_node = g_act && YCGE_REF_KIND(c_cur) == REF_MESH_NODE;
        const bool at_leaf = g_act && !at_node;
        const uint32_t unit = (c_cur & 0x1ffffff0u) >> 4;
        const uint32_t left = c_cur & 15u;                                    // triangles of the leaf
        const bool node_lane = at_node && gl < YCGE_COOP_BOX_LANES;
        const bool leaf_lane = at_leaf && gl < 8u && 2u * gl < left;          // (the record format allows leaves of up to 15 triangles = 8 records)
        const uint32_t off = node_lane ? S.tl_offset + unit * YCGE_TL_BYTES_PER_UNIT + gl * 32u : leaf_lane ? (unit + 3u * gl) * 32u : 0u;
        f32x4 a, b, c, e;
        f32x2 f;
        YCGE_DBG_SECTION(dbg_thead);
#if defined(YCGE_DBG_COOPSTAT)
        const unsigned long long dbg_f0 = __builtin_amdgcn_s_memtime();
#endif
        coop_fetch(S.mesh_arena, off, __ballot(leaf_lane), a, b, c, e, f);
#if defined(YCGE_DBG_COOPSTAT)
        dbg_fetch += __builtin_amdgcn_s_memtime() - dbg_f0;
#endif
        if (g_act) c_steps++;
#if defined(YCGE_DBG_COOPSTAT)
        dbg_iters++; dbg_node += (uint32_t)__popcll(__ballot(at_node && gl == 0u)); dbg_leaf += (uint32_t)__popcll(__ballot(at_leaf && gl == 0u));
#endif

        // ------------------------------------------------------------------ node step
#if defined(YCGE_DBG_COOPSTAT)
        dbg_mark = __builtin_amdgcn_s_memtime();
#endif
        if (__any(at_node)) {
            // slot: a = (min x, min y, min z, max x), b = (max y, max z, reference, valid) - MeshBVH.BoxHitFast, box_mesh()'s operations
            const float tx_en = ((sx ? a.w : a.x) - ox) * ix, tx_ex = ((sx ? a.x : a.w) - ox) * ix;
            const float ty_en = ((sy ? b.x : a.y) - oy) * iy, ty_ex = ((sy ? a.y : b.x) - oy) * iy;
            const float tz_en = ((sz ? b.y : a.z) - oz) * iz, tz_ex = ((sz ? a.z : b.y) - oz) * iz;
            const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(c_tmin, tx_en), ty_en), tz_en);
            const float tx = __builtin_fminf(__builtin_fminf(__builtin_fminf(c_closest, tx_ex), ty_ex), tz_ex);
            const uint32_t ref = __float_as_uint(b.z);
            const bool hit = node_lane && __float_as_uint(b.w) != 0u && tx >= tn;
            // The sibling's answer by one DPP swap each (quad_perm [1,0,3,2]); every lane then knows what its PAIR does once the walk
            // reaches it: which of the two is visited next (`near_me`: the only one hit, or of two the nearer - `if (lNear < rNear) left
            // first else right first`, MeshBVH.cs:213-223) and which is stacked (`far_me`).
            const float sib_tn = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(tn), 0xB1, 0xf, 0xf, true));
            const unsigned long long EVEN = 0x5555555555555555ull;            // left slots (a group starts on a multiple of 16)
            const unsigned long long H = __ballot(hit);
            const unsigned long long SH = ((H & EVEN) << 1) | ((H >> 1) & EVEN);          // the sibling's hit flag
            // `if (lNear < rNear) left first else right first` (MeshBVH.cs:213-223), seen from either slot of the pair
            const unsigned long long LF = (__ballot(tn < sib_tn) & EVEN) | (__ballot(sib_tn < tn) & ~EVEN);
            const unsigned long long MINE = ~(LF ^ EVEN);                     // this slot is the one the pair's order puts first
            const unsigned long long N = H & (~SH | MINE);                    // visited next once the walk reaches its pair: the only one hit, or of two the nearer
            const unsigned long long F = H & SH & ~MINE;                      // ... stacked then
            const unsigned long long I = __ballot(YCGE_REF_KIND(ref) == REF_MESH_NODE);
            const unsigned long long NI = N & I;
            // a pair is reached when every slot above it on its path was visited and is an internal node
            const uint32_t ni32 = hi_half ? (uint32_t)(NI >> 32) : (uint32_t)NI;
            const unsigned long long R = __ballot((ni32 & anc32) == anc32);
            // the far children, stacked by their own lanes: heap order is level order and a level stacks at most one, so a lane's
            // place is the number of stacking slots below it - the reference's push order
            const unsigned long long PUSHM = F & R;
            // the walk goes on at the one visited slot that is a leaf or sits on the last level; none: both children missed somewhere -> the stack
            const unsigned long long L3 = YCGE_COOP_LEVELS == 3 ? 0x3fc03fc03fc03fc0ull : 0x3c3c3c3c3c3c3c3cull;       // the slots of the last level
            const unsigned long long V = N & R;
            const unsigned long long EXM = (V & ~I) | (V & L3);
            const uint32_t push32 = (hi_half ? (uint32_t)(PUSHM >> 32) : (uint32_t)PUSHM) & grp32;
            const uint32_t ex32 = (hi_half ? (uint32_t)(EXM >> 32) : (uint32_t)EXM) & grp32;
            if (at_node) {
                if ((push32 >> (sh32 + gl)) & 1u) st.write_at(owner, c_sp + (int)__builtin_popcount(push32 & below32), ref, tn);
                c_sp += (int)__builtin_popcount(push32);
            }
            if (__any(push32 != 0u)) top = st.read_early(owner, c_sp - 1);
            const uint32_t next = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 32u) + (ex32 ? (uint32_t)__builtin_ctz(ex32) : 0u)) << 2), (int)ref);
            if (at_node) c_cur = ex32 ? next : YCGE_REF_NONE_VALUE;
        }

        YCGE_DBG_SECTION(dbg_tnode);
        // ------------------------------------------------------------------ leaf step
        if (__any(at_leaf)) {
            // TriHit for the lane's two triangles, every operation as in tri_pair_hit; the closest side of the range test against the
            // leaf's entry closest makes a CANDIDATE
            const f32x2 ax = a.xy, ay = a.zw, az = b.xy, e1x = b.zw, e1y = c.xy, e1z = c.zw, e2x = e.xy, e2y = e.zw, e2z = f;
            const f32x2 vdx = {dx, dx}, vdy = {dy, dy}, vdz = {dz, dz};
            const f32x2 vox = {ox, ox}, voy = {oy, oy}, voz = {oz, oz};
            const f32x2 px = vdy * e2z - vdz * e2y;
            const f32x2 py = vdz * e2x - vdx * e2z;
            const f32x2 pz = vdx * e2y - vdy * e2x;
            const f32x2 det = e1x * px + e1y * py + e1z * pz;
            const f32x2 sxx = vox - ax, syy = voy - ay, szz = voz - az;
            const f32x2 u_num = sxx * px + syy * py + szz * pz;
            const f32x2 sgn = {det.x > 0.0f ? 1.0f : -1.0f, det.y > 0.0f ? 1.0f : -1.0f};
            const f32x2 det_abs = det * sgn;
            const f32x2 u_num_s = u_num * sgn;
            const f32x2 qx = syy * e1z - szz * e1y;
            const f32x2 qy = szz * e1x - sxx * e1z;
            const f32x2 qz = sxx * e1y - syy * e1x;
            const f32x2 v_num = vdx * qx + vdy * qy + vdz * qz;
            const f32x2 v_num_s = v_num * sgn;
            const f32x2 uv_sum_s = u_num_s + v_num_s;
            const f32x2 t_num = e2x * qx + e2y * qy + e2z * qz;
            const f32x2 t_num_s = t_num * sgn;
            const f32x2 tmin2 = {c_tmin, c_tmin};
            const f32x2 t_min_scaled = tmin2 * det_abs;
            bool cand0 = leaf_lane;
            cand0 &= !(det.x > -1e-8f && det.x < 1e-8f);
            cand0 &= !(u_num_s.x < 0.0f || u_num_s.x > det_abs.x);
            cand0 &= !(v_num_s.x < 0.0f || uv_sum_s.x > det_abs.x);
            cand0 &= !(t_num_s.x < t_min_scaled.x || t_num_s.x > c_closest * det_abs.x);
            bool cand1 = leaf_lane && 2u * gl + 1u < left;
            cand1 &= !(det.y > -1e-8f && det.y < 1e-8f);
            cand1 &= !(u_num_s.y < 0.0f || u_num_s.y > det_abs.y);
            cand1 &= !(v_num_s.y < 0.0f || uv_sum_s.y > det_abs.y);
            cand1 &= !(t_num_s.y < t_min_scaled.y || t_num_s.y > c_closest * det_abs.y);
            const uint32_t first = (unit + 3u * gl) << 1;            // (record unit << 1) | slot: tri_pair_hit's hit_sub
            // replay in leaf order: the lowest candidate lane of each group accepts against the running closest, tells its group, next
            for (;;) {
                const unsigned long long CM = __ballot(cand0 || cand1);
                if (CM == 0ull) break;
                const uint32_t cm = (uint32_t)(CM >> sh) & 0xffu;
                const uint32_t pl = cm ? (uint32_t)__builtin_ctz(cm) : gl;           // groups without a candidate read themselves
                if (cm && gl == pl) {
                    // (TriHit's one division per accepted triangle, MeshBVH.cs:299: evaluated HERE, for the rare candidate only - the empty asm
                    // keeps the compiler from hoisting the two IEEE divisions, 22 dependent instructions, in front of every leaf step)
                    float d0 = det.x, d1 = det.y;
                    asm volatile("" : "+v"(d0), "+v"(d1));
                    if (cand0 && !(t_num_s.x > c_closest * det_abs.x)) { c_closest = t_num.x * (1.0f / d0); c_hit_sub = first; }
                    if (cand1 && !(t_num_s.y > c_closest * det_abs.y)) { c_closest = t_num.y * (1.0f / d1); c_hit_sub = first + 1u; }
                    cand0 = cand1 = false;
                }
                const int src = (int)((sh + pl) << 2);
                c_closest = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(c_closest)));
                c_hit_sub = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c_hit_sub);
            }
            if (at_leaf) {
                c_cur = YCGE_REF_NONE_VALUE;
                if (c_anyhit && c_hit_sub != 0xffffffffu) c_sp = 0;          // occlusion query answered: drop what is left
            }
        }

        YCGE_DBG_SECTION(dbg_tleaf);
        // ------------------------------------------------------------------ back to the stack (the reference's re-test on pop: closest >= tNear)
        const bool popping = g_act && c_cur == YCGE_REF_NONE_VALUE;
        if (popping) {
            bool first_pop = true;
            while (c_sp > 0) {
                c_sp--;
                uint32_t r; float tn;
                if (first_pop && c_sp < YCGE_LDS_STACK) { r = top.x; tn = __uint_as_float(top.y); }
                else st.read_at(owner, c_sp, r, tn);
                first_pop = false;
                if (c_closest >= tn) { c_cur = r; break; }
            }
            if (c_cur == YCGE_REF_NONE_VALUE && gl == 0u) {          // this ray is done: its answer goes back through its slot
                uint32_t *so = slots + g * 16;
                so[10] = __float_as_uint(c_closest); so[13] = c_hit_sub; so[14] = c_hit_sub != 0xffffffffu ? 1u : 0u; so[15] = c_steps;
            }
        }
        if (__any(popping)) top = st.read_early(owner, c_sp - 1);       // (a step without push or pop left the top where it was: already asked for)
        YCGE_DBG_SECTION(dbg_tpop);
    }
#if defined(YCGE_DBG_COOPSTAT)
    if (lane == 0u && S.dbg_counters) {       // [16 + 8 i]: invocations, loop iterations, group node steps, group leaf steps, 100 MHz ticks, rays
        unsigned long long *dc = S.dbg_counters + 16 + (size_t)((blockIdx.x * 2654435761u) >> 24) * 8;
        atomicAdd(dc + 0, 1ull); atomicAdd(dc + 1, (unsigned long long)dbg_iters); atomicAdd(dc + 2, (unsigned long long)dbg_node);
        atomicAdd(dc + 3, (unsigned long long)dbg_leaf); atomicAdd(dc + 4, __builtin_amdgcn_s_memrealtime() - dbg_t0); atomicAdd(dc + 5, (unsigned long long)n_live); atomicAdd(dc + 6, dbg_fetch); atomicAdd(dc + 7, __builtin_amdgcn_s_memtime() - dbg_c0);
        unsigned long long *d2 = dc + 8 * 256;      // second bank: shader clocks by section - head (decode, addresses), node step, leaf step, stack
        atomicAdd(d2 + 0, dbg_thead); atomicAdd(d2 + 1, dbg_tnode); atomicAdd(d2 + 2, dbg_tleaf); atomicAdd(d2 + 3, dbg_tpop);
    }
#endif
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        const uint32_t *so = slots + rank * 16;
        if (so[14] & 1u) { closest = __uint_as_float(so[10]); hit_prim = mesh_prim; hit_sub = (int)so[13]; }
        w.steps += so[15];
        cur = YCGE_REF_NONE_VALUE;
        st.reset();
    }
}

} // namespace ycge
