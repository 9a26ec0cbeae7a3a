// ycge_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ray-trace core.
//
// The per-pixel loop of RaytraceRenderer.TryFlipAndBlit (reference
// RayTracing/RaytraceRenderer.cs:183-218) runs as a WAVEFRONT pipeline of small stage kernels:
//
//   k_wf_primary   camera-ray generation (MakeJitteredRay) + closest hit of the primary ray
//   k_wf_shade     the body of TraceFull for one path vertex: miss -> sky; hit -> G-buffer,
//                  emission, mirror continuation, ambient, cosine-sampled bounce.  Emits the next
//                  path ray and one "light record" per diffuse vertex through wave-level
//                  ballot + prefix-sum compaction (one atomic per wavefront)
//   k_wf_lights    per light record: the light loop of TraceFull incl. ComputeTransmittanceToLight,
//                  accumulating into the pixel's radiance in light order
//   k_wf_extend    closest hit for the compacted continuation rays
//
// launched as primary, shade, lights, [extend, shade, lights] x (1..3).  Traversal kernels carry
// only a ray, a closest-hit triple and an LDS stack, so they run at high occupancy and every lane
// of a wavefront holds a live ray; all per-pixel floating-point sums happen in the order the
// reference performs them, so results are bit-identical to the single-thread-per-pixel C#.
//
// k_trace is the same computation as ONE launch (thread = pixel, in-kernel state machine).  It is
// the path for scenes with transparent materials (refraction splits need the per-pixel LIFO of
// TraceFull) and serves as an independent cross-check of the wavefront pipeline.
//
// K_taa = TemporalBlendWithClamp (RaytraceRenderer.cs:274-398); K_unpermute / K_pack_slab move tile
// slabs for the multi-GPU all-gather.  No MFMA anywhere: this is branchy pointer chasing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "ycge_rt.hip.h"

namespace ycge {

// ---------------------------------------------------------------------------------- ray generation
// MakeJitteredRay, RaytraceRenderer.cs:419-437 (+ Ray ctor renormalisation, Ray.cs:8-12)
__device__ __forceinline__ void make_primary_ray(const FrameParams &P, int px, int py, F3 &o, F3 &d)
{
    float jx = frac(blue_noise_sample(px, py, P.frame_idx, 0) + P.rot_x) - 0.5f;
    float jy = frac(blue_noise_sample(px, py, P.frame_idx, 1) + P.rot_y) - 0.5f;
    float u = (((float)px + 0.5f + jx) / (float)P.hiW) * 2.0f - 1.0f;
    float v = 1.0f - (((float)py + 0.5f + jy) / (float)P.hiH) * 2.0f;
    F3 fwd = f3(P.fwd), right = f3(P.right), up = f3(P.up);
    F3 dir = normalized(fwd + right * (u * P.half_w) + up * (v * P.half_h));
    o = f3(P.cam_pos);
    d = normalized(dir);
}

// block -> tile -> pixel: 32x8 tile per 256-thread block, 8x8 sub-tile per wavefront
__device__ __forceinline__ int block_tile(const FrameParams &P) { return P.tile_order ? (int)P.tile_order[blockIdx.x] : (int)blockIdx.x; }
__device__ __forceinline__ bool tile_pixel_wl(const FrameParams &P, int k, int wave, int lane, int &px, int &py, int &lx, int &ly)
{
    const int tile_id = P.rank + k * P.world_size;
    const int tx = tile_id % P.tiles_x, ty = tile_id / P.tiles_x;
    lx = wave * 8 + (lane & 7); ly = lane >> 3;
    px = tx * YCGE_TILE_W + lx; py = ty * YCGE_TILE_H + ly;
    return px < P.hiW && py < P.hiH;
}
__device__ __forceinline__ bool tile_pixel(const FrameParams &P, int k, int &px, int &py, int &lx, int &ly)
{
    return tile_pixel_wl(P, k, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), px, py, lx, ly);
}

template <bool COUNT>
__device__ __forceinline__ void flush_work(const Work &w, unsigned long long *counters)
{
    if (!counters) return;
    const int lane = threadIdx.x & 63;
    if (!COUNT) {       // the timed instances report what THEY walked: lane steps, one atomic per wavefront (ycge_read_timed_steps) -
        // spread over YCGE_TIMED_STEP_SLOTS cache lines: one hot address takes ~88 atomics per microsecond on this part, and a 4K frame's
        // primary stage alone retires 130 000 wavefronts (measured: 0.35 -> 1.58 ms with a single counter)
        unsigned long long x = w.steps;
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0 && x) atomicAdd(counters + 8 + (size_t)((blockIdx.x * 2654435761u) >> (32 - YCGE_TIMED_STEP_SLOTS_LG)) * 8, x);
        return;
    }
    const unsigned v[6] = {w.rays, w.box, w.tri, w.prim, w.vox, w.dark};
    for (int c = 0; c < 6; c++) {
        unsigned long long x = v[c];
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0 && x) atomicAdd(counters + c, x);
    }
}

#if defined(YCGE_DBG_VOXSTAT)
// The counting TWIN of the timed stage kernels (lib/var_voxstat.so; bench.py --config 5 replays its frames through it): what the TIMED
// instances walk, by kind, summed over every stage kernel - bank 4: scene-tree steps (node, leaf, object), objects culled, grids asked /
// entered, cell steps, cell fetches; bank 5: light records and continuation rays the shade stage wrote.
__device__ __forceinline__ void voxstat_flush_all(const SceneDev &S, const Work &w)
{
    if (!S.dbg_counters) return;
    unsigned long long *dc = S.dbg_counters + 16 + (size_t)8 * 256 * 4 + (size_t)((blockIdx.x * 2654435761u) >> 24) * 8;
    for (int i = 0; i < 8; i++) {
        unsigned long long x = w.dbg[i];
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(dc + i, x);
    }
}
#define YCGE_VOXSTAT_FLUSH(S, w) do { if (!COUNT) voxstat_flush_all(S, w); } while (0)
#else
#define YCGE_VOXSTAT_FLUSH(S, w) do { } while (0)
#endif

// ---------------------------------------------------------------------------------- wavefront records
struct alignas(16) QEntry {     // one live path ray between stages (48 B)
    float o[3];
    float d[3];
    float beta[3];
    uint32_t pixel;
    uint32_t flags;             // bits 0-3 mirror depth, 4-7 diffuse depth
    uint32_t pad;
};
struct alignas(16) HitRec {     // closest hit of a queued ray (16 B)
    float t;
    int32_t prim;
    int32_t sub;
    int32_t pad;
};
struct alignas(16) LEntry {     // shading context of one diffuse vertex for the light loop (64 B)
    float p[3];
    float n[3];
    float alb[3];
    float wo[3];
    float beta[3];
    uint32_t pixel;
};
static_assert(sizeof(QEntry) == 48 && sizeof(HitRec) == 16 && sizeof(LEntry) == 64, "wavefront record sizes");

struct WfBuffers {
    QEntry *q[2];               // ping-pong continuation-ray queues, segmented per tile: entry (k * 256 + slot)
    HitRec *hit;                // hit of queue entry (k * 256 + slot); round 0: slot = pixel's place in its tile
    LEntry *lq;                 // light records of the current round, same segmentation
    uint32_t *n_q;              // [round][owned tile] live entries of that tile's queue segment (round 0: implicit 256)
    uint32_t *n_lq;             // [owned tile] light records of the current round
    uint32_t *chunk_ctr;        // next tile segment of the persistent extend stage
    uint32_t tiles;             // owned tiles (= gridDim.x of every stage)
    const uint32_t *tile_order; // = FrameParams.tile_order
    uint32_t elide_dark;        // timed launches: lights of zero intensity get no shadow query (GLight::dark, light_is_dark)
};

// Compaction without global atomics: every tile (= workgroup) owns a 256-entry segment of each queue.
// Live lanes are packed to the front of the segment with wave ballots + a 4-entry prefix over the
// workgroup's wavefronts, and the segment's count is stored for the next stage.  A single hot counter
// would serialise at ~88 allocations/us (measured on this part: 65k wave allocations cost ~0.7 ms).
__device__ __forceinline__ uint32_t block_compact(bool want, uint32_t *s_cnt /* [4] in LDS */, uint32_t &total)
{
    const unsigned long long mask = __ballot(want);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    __syncthreads();
    total = c0 + c1 + c2 + c3;
    const uint32_t base = wave == 0 ? 0u : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return base + prefix;
}

// A frame's per-pixel results - G-buffer planes, the sky flag, the HDR colour of the single-launch kernel - are written once and read
// after the launch (TAA, the post stage): non-temporal stores, so that they do not push the tree out of the L2s while the trace runs
// (config 3: trace 0.291 -> 0.280 ms; config 4 0.492 -> 0.490, in flight 0.535 -> 0.530).
#ifndef YCGE_OUT_NT
#define YCGE_OUT_NT 1
#endif
template <class T> __device__ __forceinline__ void out_st(T *p, T v) { if (YCGE_OUT_NT) __builtin_nontemporal_store(v, p); else *p = v; }
// ... or written THROUGH (device-coherent, sc1) where the launch itself reads them again on another XCD: the trace kernel's own TAA (TaaFuse)
template <class T> __device__ __forceinline__ void out_st(T *p, T v, bool through) { if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else out_st(p, v); }

// ---------------------------------------------------------------------------------- k_wf_primary
// one workgroup per owned 32x8 tile: the hardware dispatcher balances the (very uneven) tiles
template <bool COUNT, bool HAS_GRID, bool FLAT>
__global__ __launch_bounds__(256) void k_wf_primary(const SceneDev S, const FrameParams P, const TraceOut O, const WfBuffers B)
{
    Work w = {0, 0, 0, 0, 0, 0, 0};
    const int k = block_tile(P);
    int px, py, lx, ly;
    unsigned long long t_start = 0;
    if (O.wave_prof && O.wave_prof_stage == 0) {
        if (COUNT) { if (threadIdx.x < 16) g_wave_iters[threadIdx.x] = 0; __syncthreads(); }
        t_start = __builtin_readcyclecounter();
    }
    if (tile_pixel(P, k, px, py, lx, ly)) {
        Stack st;
        st.init(O.stack_spill, O.stack_lanes);
        RayQ q;
        make_primary_ray(P, px, py, q.o, q.d);
        q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
        if (O.rays) {
            float *r = O.rays + ((size_t)px + (size_t)py * P.hiW) * 6;
            r[0] = q.o.x; r[1] = q.o.y; r[2] = q.o.z; r[3] = q.d.x; r[4] = q.d.y; r[5] = q.d.z;
        }
        float t; int prim, sub;
        traverse<COUNT, HAS_GRID, FLAT>(S, q, st, t, prim, sub, w);
        *(float4 *)(B.hit + ((size_t)k * 256 + threadIdx.x)) = make_float4(t, __int_as_float(prim), __int_as_float(sub), 0.0f);
    }
    if (O.wave_prof && O.wave_prof_stage == 0 && (threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        unsigned long long *dst = O.wave_prof + ((size_t)k * 4 + wv) * 4;
        dst[0] = t_start; dst[1] = __builtin_readcyclecounter();
        dst[2] = COUNT ? (((unsigned long long)g_wave_iters[wv * 4 + 1] << 32) | g_wave_iters[wv * 4]) : 0;
        dst[3] = COUNT ? (((unsigned long long)g_wave_iters[wv * 4 + 3] << 32) | g_wave_iters[wv * 4 + 2]) : 0;
    }
    flush_work<COUNT>(w, O.counters);
    YCGE_VOXSTAT_FLUSH(S, w);
}

// ---------------------------------------------------------------------------------- k_wf_extend
template <bool COUNT, bool HAS_GRID, bool FLAT>
__global__ __launch_bounds__(256) void k_wf_extend(const SceneDev S, const TraceOut O, const WfBuffers B, int round)
{
    const int k = B.tile_order ? (int)B.tile_order[blockIdx.x] : (int)blockIdx.x;
    const uint32_t n = B.n_q[(size_t)round * B.tiles + k];
    Work w = {0, 0, 0, 0, 0, 0, 0};
    const bool prof = O.wave_prof && O.wave_prof_stage == 1 && round == 1;
    unsigned long long t_start = 0;
    if (prof) {
        if (COUNT) { if (threadIdx.x < 16) g_wave_iters[threadIdx.x] = 0; __syncthreads(); }
        t_start = __builtin_readcyclecounter();
    }
    if (threadIdx.x < n) {
        const size_t i = (size_t)k * 256 + threadIdx.x;
        const QEntry *Q = B.q[round & 1];
        const float4 a = ((const float4 *)(Q + i))[0], b = ((const float4 *)(Q + i))[1];
        Stack st;
        st.init(O.stack_spill, O.stack_lanes);
        RayQ q;
        q.o = f3(a.x, a.y, a.z); q.d = f3(a.w, b.x, b.y);
        q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
        float t; int prim, sub;
        traverse<COUNT, HAS_GRID, FLAT>(S, q, st, t, prim, sub, w);
        *(float4 *)(B.hit + i) = make_float4(t, __int_as_float(prim), __int_as_float(sub), 0.0f);
    }
    if (prof && (threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        unsigned long long *dst = O.wave_prof + ((size_t)k * 4 + wv) * 4;
        dst[0] = t_start; dst[1] = __builtin_readcyclecounter();
        dst[2] = COUNT ? (((unsigned long long)g_wave_iters[wv * 4 + 1] << 32) | g_wave_iters[wv * 4]) : 0;
        dst[3] = COUNT ? (((unsigned long long)g_wave_iters[wv * 4 + 3] << 32) | g_wave_iters[wv * 4 + 2]) : 0;
    }
    flush_work<COUNT>(w, O.counters);
    YCGE_VOXSTAT_FLUSH(S, w);
}

// ---------------------------------------------------------------------------------- persistent trace stages (ray refill)
// Secondary rays of a voxel world differ in length by two orders of magnitude (a ray into the ground: a few
// steps; a ray along the horizon: hundreds of cells through a dozen chunks), so a wavefront that keeps its 64
// rays until the longest one ends runs almost empty: measured 8 % of VALU lanes active in k_wf_extend on
// config 5.  Here a lane that finishes its ray takes the next one: persistent wavefronts draw whole tile
// segments from one counter (one atomic per tile, no other atomics) and hand the segment's rays to their idle
// lanes; every round a lane does a bounded number of tree steps or of voxel steps (the voxel walk is resumable),
// so a short ray never waits for a long one.  Which lane traces a ray never changes the ray's result.
// Used for the continuation (bounce / mirror) rays.  The shadow rays of the light loop were tried the same way
// (one ray per (light record, light), contributions applied in light order afterwards): no gain - they all aim
// at the same light, their lengths are alike, and k_wf_lights already keeps 53 % of its lanes busy.  Round 5 tried again with the
// cheaper walk: a persistent light stage, a lane per VERTEX walking its light loop in these rounds (bit-exact on every voxel test):
// 4.10 -> 4.095 ms at noon, and a LOSS where few or no vertices have a light to ask (dark 3.16 -> 4.05: 8 192 wavefronts draw 32 400
// empty segments from one counter; night 3.70 -> 4.17).  Not kept (profiles/r05/j_voxel_stage_forms.txt).
#define YCGE_ROUND_TREE_STEPS 6
#define YCGE_ROUND_CELL_STEPS 20       // (round 5, with the phase gating: 10 / 16 / 20 / 24 / 32 cell steps a round: lit config 5 3.97 / 3.91 / 3.84 / 3.84 / 3.87 ms, dark 3.01 / 2.95 / 2.88 / 2.89 / 2.93)
#define YCGE_ROUND_PHASE_MODE 1        // 0: both phases every round; 1: only the phase most lanes wait for; 2: skip a phase with < 16 takers while the other has >= 16.  Config 5 at full size, mode 0 / 1 / 2: lit 4.10 / 3.99 / 4.04 ms, dark 3.16 / 3.04 / 3.09, night 3.70 / 3.56 / 3.65 (profiles/r05/j_voxel_stage_forms.txt)
#define YCGE_ROUND_REFILL_MIN 16       // lanes that wait for a ray before rays are handed out (or all of them): lit config 5 4.06 -> 4.02 ms, its moving-camera leg 2.22 -> 2.06 (profiles/r04/h_voxel_walk_tree.txt)
#ifndef YCGE_TRACEP_WAVES
#define YCGE_TRACEP_WAVES 5          // persistent extend stage: 5 wavefronts per SIMD (102 registers, no scratch; round 3: as fast as 6 with its 10 spilled registers - 7.04 against 7.00 ms - and 0.46 GB less written per 4K frame) and 32 persistent wavefronts per CU
#endif
template <bool COUNT, bool HAS_GRID>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((HAS_GRID && !COUNT) ? YCGE_TRACEP_WAVES : 1, 8))) void k_wf_trace_p(const SceneDev S, const FrameParams P, const TraceOut O, const WfBuffers B, int round,
                                                   uint32_t *__restrict__ chunk_ctr, int round_tree_steps, int round_cell_steps, int refill_min, int phase_mode)
{
    Work w = {0, 0, 0, 0, 0, 0, 0};
    StackT<64> st;
    st.init(O.stack_spill, O.stack_lanes);
    const int lane = (int)threadIdx.x;
    const uint32_t *counts = B.n_q + (size_t)round * B.tiles;
    const QEntry *Q = B.q[round & 1];
    const float tmin = 0.001f;
    bool have = false, more = false, in_dda = false;
    uint32_t src = 0, cur = YCGE_REF_NONE_VALUE;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1), inv = f3(0, 0, 0);
    bool sx = false, sy = false, sz = false, fast = false;
    float closest = YCGE_FLT_MAX;
    int hit_prim = -1, hit_sub = 0, mesh_prim = -1;
    DdaState D;
    D.ix = D.iy = D.iz = 0; D.t = D.t_max_x = D.t_max_y = D.t_max_z = D.t_delta_x = D.t_delta_y = D.t_delta_z = D.t_exit = D.tmax = 0.0f;
    D.step_x = D.step_y = D.step_z = D.last_axis = D.use_mask = 0; D.nx = D.ny = D.nz = D.nbx = D.nby = 1; D.cell_offset = D.mask_lo = D.mask_hi = 0; D.prim = -1;
    uint32_t tile = 0, next_ray = 0, end_ray = 0;          // wave-uniform: the segment being handed out
    bool exhausted = false;
#if defined(YCGE_DBG_VOXSTAT)
    unsigned long long vs_rounds = 0, vs_tree = 0, vs_dda = 0, vs_have = 0, vs_t_fill = 0, vs_t_tree = 0, vs_t_dda = 0, vs_mark = __builtin_amdgcn_s_memtime();
#define YCGE_VS_SECTION(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - vs_mark; vs_mark = t_; } while (0)
#else
#define YCGE_VS_SECTION(acc) do { } while (0)
#endif
    for (;;) {
        // ---- hand new rays to the idle lanes
        unsigned long long idle = __ballot(!have);
        // (new rays only once refill_min lanes wait for one - or all of them: handing out rays costs a pass over ~280 instructions)
        if ((int)__popcll(idle) < refill_min && idle != __ballot(true)) idle = 0ull;
        while (idle && !exhausted) {
            if (next_ray == end_ray) {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(chunk_ctr, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= B.tiles) { exhausted = true; break; }
                tile = B.tile_order ? B.tile_order[c] : c;
                next_ray = 0;
                end_ray = counts[tile];
                continue;
            }
            const uint32_t avail = end_ray - next_ray, n_idle = (uint32_t)__popcll(idle);
            const uint32_t take = avail < n_idle ? avail : n_idle;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!have && rank < take) {
                src = tile * 256u + next_ray + rank;
                const float4 *q = (const float4 *)(Q + src);
                const float4 a = q[0], b = q[1];
                o = f3(a.x, a.y, a.z); d = f3(a.w, b.x, b.y);
                // traverse(): Scene.Hit(r, 0.001f, float.MaxValue)
                closest = YCGE_FLT_MAX; hit_prim = -1; hit_sub = 0; mesh_prim = -1;
                st.reset();
                if (COUNT) w.rays++;
                cur = YCGE_REF_NONE_VALUE;
                if (S.scene_root_ref != YCGE_REF_NONE_VALUE) {
                    inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    sx = inv.x < 0.0f; sy = inv.y < 0.0f; sz = inv.z < 0.0f;
                    float tn, tf;
                    if (COUNT) w.box++;
                    if (box_scene(S.scene_root_min[0], S.scene_root_min[1], S.scene_root_min[2], S.scene_root_max[0], S.scene_root_max[1],
                                  S.scene_root_max[2], o, inv, tmin, closest, tn, tf))
                        cur = scene_entry<COUNT, HAS_GRID>(S, tf);
                    fast = YCGE_WALK_PHASE && YCGE_REF_KIND(cur) == REF_WALK_NODE && cs_abs(inv.x) < YCGE_INF && cs_abs(inv.y) < YCGE_INF && cs_abs(inv.z) < YCGE_INF;
                }
                more = cur != YCGE_REF_NONE_VALUE;
                in_dda = false;
                have = true;
            }
            next_ray += take;
            idle = __ballot(!have);
        }
        if (!__any(have)) break;
        YCGE_VS_SECTION(vs_t_fill);
#if defined(YCGE_DBG_VOXSTAT)
        vs_rounds++; vs_have += (unsigned long long)__popcll(__ballot(have)); vs_tree += (unsigned long long)__popcll(__ballot(have && more && !in_dda));
#endif
        // ---- one round: a bounded number of tree steps for the lanes in the tree, then a bounded number of cell
        // steps for the lanes inside a grid (their state persists in D); lanes that finish are refilled at the top.
        // Each phase is issued for the whole wavefront whatever the number of lanes in it (round 4's counters: 44 lanes in the tree phase, 28 in
        // the cell phase of an average round - half of the issue slots idle), so a round may run only the phase MOST of its lanes wait for
        // (phase_mode 1) or skip a phase that fewer than a quarter of the wavefront would take part in while the other has takers (2): the
        // others wait a round, their state keeps.  Which round a ray's steps fall in never changes the ray's result.
        bool run_tree = true, run_cell = true;
        if (phase_mode) {
            const int n_tree = (int)__popcll(__ballot(have && more && !in_dda)), n_dda = (int)__popcll(__ballot(in_dda));
            if (phase_mode == 1) { if (n_dda > n_tree) run_tree = false; else if (n_dda > 0 && n_tree > 0) run_cell = false; }
            else { if (n_tree < 16 && n_dda >= 16) run_tree = false; else if (n_dda < 16 && n_tree >= 16) run_cell = false; }
        }
        if (run_tree && have && more && !in_dda) {
            int parked_grid = -1, parked_prim = -1;
            float parked_tend = YCGE_INF;       // where the ray leaves the box of the grid's solid voxels: the walk ends there (timed kernels)
            // (a ray in the walk tree of a voxel world takes the loop that holds nothing else, and leaves it only for an object that is no grid)
            int r = TREE_OTHER;
            if (HAS_GRID && !COUNT && fast) r = walk_phase(S, cur, st, o, inv, tmin, closest, parked_grid, parked_prim, w, round_tree_steps);
            if (r == TREE_OTHER)
                r = tree_phase<COUNT, HAS_GRID>(S, cur, mesh_prim, st, o, d, inv, sx, sy, sz, tmin, closest, hit_prim, hit_sub, parked_grid,
                                                parked_prim, parked_tend, w, round_tree_steps);
            if (r == TREE_DONE) more = false;
            if (HAS_GRID && r == TREE_AT_GRID) in_dda = dda_begin<COUNT>(S, parked_grid, parked_prim, o, d, inv, tmin, COUNT ? closest : fminf(closest, parked_tend), D, w);
        }
        YCGE_VS_SECTION(vs_t_tree);
#if defined(YCGE_DBG_VOXSTAT)
        vs_dda += (unsigned long long)__popcll(__ballot(in_dda));
#endif
        if (HAS_GRID && run_cell && in_dda) {
            bool in = true;
#pragma unroll 1
            for (int k2 = 0; k2 < round_cell_steps && in; k2++) in = dda_step<COUNT>(S, D, tmin, closest, hit_prim, hit_sub, w);
            in_dda = in;
        }
        YCGE_VS_SECTION(vs_t_dda);
        if (have && !more && !in_dda) {
            *(float4 *)(B.hit + src) = make_float4(closest, __int_as_float(hit_prim), __int_as_float(hit_sub), 0.0f);
            have = false;
        }
    }
#if defined(YCGE_DBG_VOXSTAT)
    if (!COUNT && S.dbg_counters) {      // banks 2 (per-lane sums) and 3 (per-wavefront sums) of the profiling counters
        unsigned long long *dc = S.dbg_counters + 16 + (size_t)8 * 256 * 2 + (size_t)((blockIdx.x * 2654435761u) >> 24) * 8;
        for (int i = 0; i < 8; i++) atomicAdd(dc + i, (unsigned long long)w.dbg[i]);
        if (lane == 0) {
            dc += 8 * 256;
            atomicAdd(dc + 0, vs_rounds); atomicAdd(dc + 1, vs_have); atomicAdd(dc + 2, vs_tree); atomicAdd(dc + 3, vs_dda);
            atomicAdd(dc + 4, vs_t_fill); atomicAdd(dc + 5, vs_t_tree); atomicAdd(dc + 6, vs_t_dda); atomicAdd(dc + 7, 1ull);
        }
    }
#endif
    flush_work<COUNT>(w, O.counters);
    YCGE_VOXSTAT_FLUSH(S, w);
}

// ---------------------------------------------------------------------------------- k_wf_shade
// One path vertex: the part of TraceFull between `scene.Hit` (RaytraceRenderer.cs:472) and the next
// query, for scenes without transparent materials.  One workgroup per tile segment.
template <bool ROUND0, bool HAS_GRID>
__global__ __launch_bounds__(256) void k_wf_shade(const SceneDev S, const FrameParams P, const TraceOut O, const WfBuffers B, int round)
{
    __shared__ uint32_t s_cnt[4];
    const bool DEBUG = O.prim_id != nullptr;       // capture_debug (wave-uniform)
    const int k = block_tile(P);
    const size_t i = (size_t)k * 256 + threadIdx.x;
    const QEntry *Q = B.q[round & 1];
    QEntry *Qn = B.q[(round + 1) & 1];
    bool live;
    uint32_t pixel = 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1), beta = f3(1, 1, 1);
    int mirror_depth = 0, diffuse_depth = 0;
    int px = 0, py = 0;
    if (ROUND0) {
        int lx, ly;
        live = tile_pixel(P, k, px, py, lx, ly);
        pixel = (uint32_t)px + (uint32_t)py * (uint32_t)P.hiW;
        if (live) make_primary_ray(P, px, py, o, d);
    } else {
        live = threadIdx.x < B.n_q[(size_t)round * B.tiles + k];
        if (live) {
            const float4 a = ((const float4 *)(Q + i))[0], b = ((const float4 *)(Q + i))[1], c = ((const float4 *)(Q + i))[2];
            o = f3(a.x, a.y, a.z); d = f3(a.w, b.x, b.y); beta = f3(b.z, b.w, c.x);
            pixel = __float_as_uint(c.y);
            const uint32_t fl = __float_as_uint(c.z);
            mirror_depth = (int)(fl & 15u); diffuse_depth = (int)((fl >> 4) & 15u);
        }
    }
    bool emit_next = false, emit_light = false;
    F3 no = f3(0, 0, 0), nd = f3(0, 0, 1), nbeta = f3(0, 0, 0);
    int n_mirror = 0, n_diffuse = 0;
    F3 sh_p = f3(0, 0, 0), sh_n = f3(0, 0, 0), sh_alb = f3(0, 0, 0), sh_wo = f3(0, 0, 0);
    if (live) {
        const float4 hv = *(const float4 *)(B.hit + i);
        const float t_hit = hv.x;
        const int hit_prim = __float_as_int(hv.y), hit_sub = __float_as_int(hv.z);
        F3 rad = f3(0, 0, 0);
        if (!ROUND0) rad = f3(O.current_hdr[3 * (size_t)pixel], O.current_hdr[3 * (size_t)pixel + 1], O.current_hdr[3 * (size_t)pixel + 2]);
        uint64_t rng = 0;
        if (ROUND0 && DEBUG) { rng = per_frame_seed(px, py, P.frame, P.seed_salt); if (rng == 0) rng = 0x9E3779B97F4A7C15ULL; }
        if (hit_prim < 0) {
            float tbg = 0.5f * (d.y + 1.0f);
            F3 sky = lerp3(f3(S.bg_bottom), f3(S.bg_top), tbg);
            rad = rad + f3(beta.x * sky.x, beta.y * sky.y, beta.z * sky.z);
            if (ROUND0) {           // item.IsPrimary && !primaryHitSomething: sky G-buffer, :476-484
                out_st(O.g_albedo + 3 * (size_t)pixel, 0.0f); out_st(O.g_albedo + 3 * (size_t)pixel + 1, 0.0f); out_st(O.g_albedo + 3 * (size_t)pixel + 2, 0.0f);
                out_st(O.g_normal + 3 * (size_t)pixel, 0.0f); out_st(O.g_normal + 3 * (size_t)pixel + 1, 0.0f); out_st(O.g_normal + 3 * (size_t)pixel + 2, 0.0f);
                out_st(O.g_depth + pixel, YCGE_FLT_MAX);
                out_st(O.sky + pixel, (uint8_t)1);
                if (DEBUG) { if (O.prim_id) O.prim_id[pixel] = -1; if (O.sub_id) O.sub_id[pixel] = 0; if (O.hit_t) O.hit_t[pixel] = YCGE_FLT_MAX; }
            }
        } else {
            HitAttr h;
            resolve_hit<HAS_GRID>(S, hit_prim, hit_sub, t_hit, o, d, h);
            if (YCGE_TEXTURES && S.any_textured) apply_texture(S, hit_prim, hit_sub, o, d, h);
            if (ROUND0) {           // primary G-buffer, :488-499
                out_st(O.g_albedo + 3 * (size_t)pixel, h.m.albedo.x); out_st(O.g_albedo + 3 * (size_t)pixel + 1, h.m.albedo.y); out_st(O.g_albedo + 3 * (size_t)pixel + 2, h.m.albedo.z);
                out_st(O.g_normal + 3 * (size_t)pixel, h.n.x); out_st(O.g_normal + 3 * (size_t)pixel + 1, h.n.y); out_st(O.g_normal + 3 * (size_t)pixel + 2, h.n.z);
                out_st(O.g_depth + pixel, t_hit);
                out_st(O.sky + pixel, (uint8_t)0);
                if (DEBUG) { if (O.prim_id) O.prim_id[pixel] = hit_prim; if (O.sub_id) O.sub_id[pixel] = h.sub_public; if (O.hit_t) O.hit_t[pixel] = t_hit; }
            }
            if (h.m.emission.x != 0.0f || h.m.emission.y != 0.0f || h.m.emission.z != 0.0f) {
                F3 e = h.m.emission;
                rad = rad + f3(beta.x * e.x, beta.y * e.y, beta.z * e.z);
            }
            const F3 base_albedo = h.m.albedo;
            if (h.m.reflectivity >= P.mirror_threshold) {              // :559-570
                if (mirror_depth < P.max_mirror_bounces) {
                    F3 refl_dir = normalized(reflect(d, h.n));
                    no = h.p + h.n * P.eps;
                    nd = normalized(refl_dir);
                    nbeta = f3(beta.x * base_albedo.x, beta.y * base_albedo.y, beta.z * base_albedo.z);
                    n_mirror = mirror_depth + 1; n_diffuse = diffuse_depth;
                    emit_next = true;
                }
            } else {
                if (S.ambient_intensity > 0.0f) {                      // :571-576
                    F3 a = f3(S.ambient[0] * S.ambient_intensity, S.ambient[1] * S.ambient_intensity, S.ambient[2] * S.ambient_intensity);
                    F3 amb = f3(a.x * base_albedo.x, a.y * base_albedo.y, a.z * base_albedo.z);
                    rad = rad + f3(beta.x * amb.x, beta.y * amb.y, beta.z * amb.z);
                }
                sh_p = h.p; sh_n = h.n; sh_alb = base_albedo;
                sh_wo = normalized(d * -1.0f);
                // a light record only where some light will ask for a shadow query (the light loop head's own test, :578-591): vertices that
                // face away from every light, and - timed launches - whose lights are all dark, go without
                emit_light = false;
                for (int li = 0; li < S.n_lights && !emit_light; li++) {
                    const GLight &L = S.lights[li];
                    F3 to_l = f3(L.pos) - sh_p;
                    const float dist2 = dot(to_l, to_l);
                    const float dist = cs_sqrt(dist2);
                    const F3 ldir = vdiv(to_l, dist);
                    const float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                    emit_light = !(n_dot_l <= 0.0f) && !(B.elide_dark && L.dark != 0.0f && dist2 > 0.0f);
                }
                if (diffuse_depth < P.diffuse_bounces) {               // :604-615
                    // the only consumer of the pixel's RNG is this bounce, so its state is the fresh seed
                    if (!(ROUND0 && DEBUG)) {
                        const int ppx = (int)(pixel % (uint32_t)P.hiW), ppy = (int)(pixel / (uint32_t)P.hiW);
                        rng = per_frame_seed(ppx, ppy, P.frame, P.seed_salt);
                        if (rng == 0) rng = 0x9E3779B97F4A7C15ULL;
                    }
                    F3 bounce = cosine_sample_hemisphere(sh_n, rng);
                    F3 f_on = oren_nayar(sh_alb, sh_n, sh_wo, bounce, P.on_a, P.on_b);
                    const float factor = YCGE_PI;
                    F3 mult = f3(f_on.x * factor, f_on.y * factor, f_on.z * factor);
                    no = sh_p + sh_n * P.eps;
                    nd = normalized(bounce);
                    nbeta = f3(beta.x * mult.x, beta.y * mult.y, beta.z * mult.z);
                    n_mirror = mirror_depth; n_diffuse = diffuse_depth + 1;
                    emit_next = true;
                    if (DEBUG && O.rng_state && !ROUND0) O.rng_state[pixel] = rng;
                }
            }
        }
        O.current_hdr[3 * (size_t)pixel] = rad.x; O.current_hdr[3 * (size_t)pixel + 1] = rad.y; O.current_hdr[3 * (size_t)pixel + 2] = rad.z;
        if (ROUND0 && DEBUG && O.rng_state) O.rng_state[pixel] = rng;
    }
    // ---- compaction of the tile's live lanes into its queue segments (ballot + prefix, no atomics)
    uint32_t n_l, n_n;
    const uint32_t ls = block_compact(emit_light, s_cnt, n_l);
    if (emit_light) {
        float4 *dst = (float4 *)(B.lq + (size_t)k * 256 + ls);
        dst[0] = make_float4(sh_p.x, sh_p.y, sh_p.z, sh_n.x);
        dst[1] = make_float4(sh_n.y, sh_n.z, sh_alb.x, sh_alb.y);
        dst[2] = make_float4(sh_alb.z, sh_wo.x, sh_wo.y, sh_wo.z);
        dst[3] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(pixel));
    }
    const uint32_t qs = block_compact(emit_next, s_cnt, n_n);
    if (emit_next) {
        float4 *dst = (float4 *)(Qn + (size_t)k * 256 + qs);
        dst[0] = make_float4(no.x, no.y, no.z, nd.x);
        dst[1] = make_float4(nd.y, nd.z, nbeta.x, nbeta.y);
        dst[2] = make_float4(nbeta.z, __uint_as_float(pixel), __uint_as_float((uint32_t)n_mirror | ((uint32_t)n_diffuse << 4)), 0.0f);
    }
    if (threadIdx.x == 0) {
        B.n_lq[k] = n_l;
        B.n_q[(size_t)(round + 1) * B.tiles + k] = n_n;
#if defined(YCGE_DBG_VOXSTAT)
        if (S.dbg_counters) {       // bank 5 of the twin's counters: records this stage wrote
            unsigned long long *dc = S.dbg_counters + 16 + (size_t)8 * 256 * 5 + (size_t)((blockIdx.x * 2654435761u) >> 24) * 8;
            if (n_l) atomicAdd(dc + 0, (unsigned long long)n_l);
            if (n_n) atomicAdd(dc + 1, (unsigned long long)n_n);
        }
#endif
    }
}

// ---------------------------------------------------------------------------------- k_wf_lights
// The light loop of TraceFull (RaytraceRenderer.cs:578-603) with ComputeTransmittanceToLight
// (:757-798) for one diffuse vertex per thread; contributions are added in light order.
#ifndef YCGE_LIGHTS_WAVES
#define YCGE_LIGHTS_WAVES 5         // voxel worlds: 5 wavefronts per SIMD (a 102-VGPR budget, 16 spilled registers outside the walk): 12.63 -> 12.35 ms; 6: 13.0
#endif
template <bool COUNT, bool HAS_GRID, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((HAS_GRID && !FLAT && !COUNT) ? YCGE_LIGHTS_WAVES : 1, 8))) void k_wf_lights(const SceneDev S, const FrameParams P, const TraceOut O, const WfBuffers B, int round)
{
    const int k = block_tile(P);
    const uint32_t n = B.n_lq[k];
    Work w = {0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x < n) {
        // The vertex's shading context stays in its light record (64 bytes, just written by k_wf_shade: L2-resident) and is read again
        // where it is used - before a shadow query only position and normal, after it the rest - and the pixel's radiance is read and
        // written around each contribution: nothing but the loop state is live across a query.  (Kept in registers, the context cost the
        // 5-wavefront build of this kernel 21 spilled registers: 0.64 GB of scratch writes per launch on the 4K voxel frame.)
        const float4 *src = (const float4 *)(B.lq + (size_t)k * 256 + threadIdx.x);
        Stack st;
        st.init(O.stack_spill, O.stack_lanes);
        // per-lane loop state: one traversal call site serves every light and every transmittance segment
        int light = 0;
        bool in_query = false;
        RayQ q; q.o = f3(0, 0, 0); q.d = f3(0, 0, 1); q.tmin = 0.0f; q.tmax = 0.0f;
        float maxdist = 0.0f;
        float tr_r = 1.0f, tr_g = 1.0f, tr_b = 1.0f;
        int tr_counter = 0;
        for (;;) {
            if (!in_query) {
                // light loop head: find the next light that needs a shadow query
                const float4 a = src[0], b = src[1];
                const F3 sh_p = f3(a.x, a.y, a.z), sh_n = f3(a.w, b.x, b.y);
                for (; light < S.n_lights; light++) {
                    const GLight &L = S.lights[light];
                    F3 to_l = f3(L.pos) - sh_p;
                    const float dist2 = dot(to_l, to_l);
                    float dist = cs_sqrt(dist2);
                    const F3 ldir = vdiv(to_l, dist);
                    const float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                    if (n_dot_l <= 0.0f) continue;
                    if (light_is_dark<COUNT>(L, dist2, w)) continue;       // nothing to add whatever the query says
                    q.o = sh_p + sh_n * P.eps;
                    q.d = normalized(ldir);                     // new Ray(..., ldir)
                    maxdist = dist - P.eps;
                    q.tmax = maxdist;
                    q.tmin = S.is_volume_scene ? 0.001f : 0.0f + P.eps;
                    q.anyhit = S.is_volume_scene || !S.any_transparent;              // every occluder is opaque: the first hit answers the query
                    tr_r = tr_g = tr_b = 1.0f; tr_counter = 0;
                    in_query = true;
                    break;
                }
                if (!in_query) break;
            }
            float t_hit; int hit_prim, hit_sub;
            traverse<COUNT, HAS_GRID, FLAT>(S, q, st, t_hit, hit_prim, hit_sub, w);
            const bool hit = hit_prim >= 0;
            bool done = true;
            if (S.is_volume_scene) {                            // binary occlusion, :761-765
                if (hit) { tr_r = tr_g = tr_b = 0.0f; }
            } else if (hit && !S.any_transparent) {             // no material with Transparency > 0: `tr <= 0` for whatever was hit, :779-781
                tr_r = tr_g = tr_b = 0.0f;
            } else if (hit) {                                   // loop body, :775-795
                tr_counter++;
                HitAttr h;
                resolve_hit<HAS_GRID>(S, hit_prim, hit_sub, t_hit, q.o, q.d, h);
                const float tr = h.m.transparency;
                if (tr <= 0.0f) { tr_r = tr_g = tr_b = 0.0f; }
                else {
                    tr_r *= h.m.trans_color.x * tr; tr_g *= h.m.trans_color.y * tr; tr_b *= h.m.trans_color.z * tr;
                    if (tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f) { tr_r = tr_g = tr_b = 0.0f; }
                    else if (!(t_hit > maxdist)) {
                        q.tmin = t_hit + P.eps;
                        if (tr_counter < P.max_refractions) done = false;      // same ray, next segment
                    }
                }
            }
            if (done) {
                if (!(tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f)) {      // :592-602
                    // the light-loop head's values for this light again, by its own expressions on the same inputs: the same bits
                    const float4 a = src[0], b = src[1], c = src[2], e = src[3];
                    const F3 sh_p = f3(a.x, a.y, a.z), sh_n = f3(a.w, b.x, b.y), sh_alb = f3(b.z, b.w, c.x), sh_wo = f3(c.y, c.z, c.w), beta = f3(e.x, e.y, e.z);
                    const GLight &L = S.lights[light];
                    F3 to_l = f3(L.pos) - sh_p;
                    const float dist2 = dot(to_l, to_l);
                    float dist = cs_sqrt(dist2);
                    const F3 ldir = vdiv(to_l, dist);
                    const float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                    float atten = L.intensity / dist2;
                    F3 f_diffuse = oren_nayar(sh_alb, sh_n, sh_wo, ldir, P.on_a, P.on_b);
                    F3 Li = f3(L.color) * atten;
                    F3 contrib = (f_diffuse * n_dot_l) * Li;
                    contrib = f3(contrib.x * tr_r, contrib.y * tr_g, contrib.z * tr_b);
                    float *rp = O.current_hdr + 3 * (size_t)__float_as_uint(e.w);
                    rp[0] = rp[0] + beta.x * contrib.x; rp[1] = rp[1] + beta.y * contrib.y; rp[2] = rp[2] + beta.z * contrib.z;      // radiance += throughput * contrib, in light order
                }
                light++;
                in_query = false;
            }
        }
    }
    flush_work<COUNT>(w, O.counters);
    YCGE_VOXSTAT_FLUSH(S, w);
}

// ---------------------------------------------------------------------------------- K_trace (single launch)
enum Phase : int { PH_PATH = 0, PH_SHADOW_OCC = 1, PH_SHADOW_TR = 2, PH_DONE = 3 };

struct PathItem {       // PathWorkItem, RaytraceRenderer.cs:439-446 (IsPrimary is false for every pushed item)
    F3 o, d, beta;
    int mirror_depth, diffuse_depth;
};

// TraceFull's PathWorkItem stack (RaytraceRenderer.cs:450-453) lives in HBM, [slot][field][global lane]:
// it is touched only on refraction splits, and a private array would make the whole kernel scratch-backed.
struct PathStack {
    float *base; uint32_t lanes, lane_base;
    __device__ __forceinline__ float *at(int slot, int field) const { return base + ((size_t)(slot * 11 + field) * lanes + (lane_base + blockIdx.x * blockDim.x + threadIdx.x)); }
    __device__ __forceinline__ void store(int slot, const PathItem &it) const
    {
        *at(slot, 0) = it.o.x; *at(slot, 1) = it.o.y; *at(slot, 2) = it.o.z; *at(slot, 3) = it.d.x; *at(slot, 4) = it.d.y; *at(slot, 5) = it.d.z;
        *at(slot, 6) = it.beta.x; *at(slot, 7) = it.beta.y; *at(slot, 8) = it.beta.z;
        *at(slot, 9) = __int_as_float(it.mirror_depth); *at(slot, 10) = __int_as_float(it.diffuse_depth);
    }
    __device__ __forceinline__ PathItem load(int slot) const
    {
        PathItem it;
        it.o = f3(*at(slot, 0), *at(slot, 1), *at(slot, 2)); it.d = f3(*at(slot, 3), *at(slot, 4), *at(slot, 5));
        it.beta = f3(*at(slot, 6), *at(slot, 7), *at(slot, 8));
        it.mirror_depth = __float_as_int(*at(slot, 9)); it.diffuse_depth = __float_as_int(*at(slot, 10));
        return it;
    }
};

// ---------------------------------------------------------------------------------- TemporalBlendWithClamp, one pixel (k_taa, k_taa_tiles, trace_block)
__device__ __forceinline__ float luma(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

#ifndef YCGE_TAA_NT
#define YCGE_TAA_NT 1
#endif
// What TAA touches once per frame - the history and the guide planes it reads and rewrites, this frame's normal and depth - goes past
// the caches' keep lists (non-temporal): 230 MB a 1080p frame that would otherwise push the trace's tree out of the L2s just
// before the next trace (or while it runs: frames in flight).  The 3 x 3 colour taps stay ordinary loads, they are shared.
template <class T> __device__ __forceinline__ T taa_ld(const T *p) { return YCGE_TAA_NT ? __builtin_nontemporal_load(p) : *p; }
template <class T> __device__ __forceinline__ void taa_st(T *p, T v) { if (YCGE_TAA_NT) __builtin_nontemporal_store(v, p); else *p = v; }
// TemporalBlendWithClamp, RaytraceRenderer.cs:274-398.  One thread per pixel; the history and
// guide updates touch only the thread's own pixel, so the serial loops of the C# fuse into one pass.
// taa_blend: everything behind the pixel's own values of this frame and the luminance range of its window - the reset copy (:285-300), the
// per-pixel alpha (:318-340), the clamp of the history's luminance (:362-378), the blend and the guide copies (:380-396).
__device__ __forceinline__ void taa_blend(const TaaParams &T, const size_t i, const float cr, const float cg, const float cb, const float nx, const float ny, const float nz,
                                          const float z_now, const uint8_t sky_now, const float min_l, const float max_l, float *__restrict__ hist,
                                          float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    float pr = taa_ld(hist + 3 * i), pg = taa_ld(hist + 3 * i + 1), pb = taa_ld(hist + 3 * i + 2);
    float local_alpha = T.alpha;
    if ((sky_now != 0) != (taa_ld(prev_sky + i) != 0)) {
        local_alpha = 1.0f;
    } else {
        const float z_prev = taa_ld(prev_depth + i);
        F3 n_now = normalized(f3(nx, ny, nz));
        F3 n_prev = normalized(f3(taa_ld(prev_normal + 3 * i), taa_ld(prev_normal + 3 * i + 1), taa_ld(prev_normal + 3 * i + 2)));
        if (!cs_isfinite(z_now) || !cs_isfinite(z_prev)) {
            local_alpha = 1.0f;
        } else {
            float dz = cs_abs(z_now - z_prev);
            float rel = dz / cs_max(1e-4f, cs_min(z_now, z_prev));
            float ndot = dot(n_now, n_prev);
            if (rel > 0.05f || ndot < 0.8f) local_alpha = 1.0f;
        }
    }
    float range = max_l - min_l;
    float l_min = min_l - range * T.pad_lum;
    float l_max = max_l + range * T.pad_lum;
    float prev_l = luma(pr, pg, pb);
    if (prev_l > l_max) {
        float s = l_max / cs_max(1e-6f, prev_l);
        pr = pr * s; pg = pg * s; pb = pb * s;
    } else if (prev_l < l_min) {
        float s = l_min / cs_max(1e-6f, prev_l);
        pr = pr * s; pg = pg * s; pb = pb * s;
    }
    taa_st(hist + 3 * i, pr * (1.0f - local_alpha) + cr * local_alpha);
    taa_st(hist + 3 * i + 1, pg * (1.0f - local_alpha) + cg * local_alpha);
    taa_st(hist + 3 * i + 2, pb * (1.0f - local_alpha) + cb * local_alpha);
    taa_st(prev_normal + 3 * i, nx); taa_st(prev_normal + 3 * i + 1, ny); taa_st(prev_normal + 3 * i + 2, nz);
    taa_st(prev_depth + i, z_now);
    taa_st(prev_sky + i, sky_now);
}
__device__ __forceinline__ void taa_reset(const size_t i, const float cr, const float cg, const float cb, const float nx, const float ny, const float nz, const float z_now,
                                          const uint8_t sky_now, float *__restrict__ hist, float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    taa_st(hist + 3 * i, cr); taa_st(hist + 3 * i + 1, cg); taa_st(hist + 3 * i + 2, cb);
    taa_st(prev_normal + 3 * i, nx); taa_st(prev_normal + 3 * i + 1, ny); taa_st(prev_normal + 3 * i + 2, nz);
    taa_st(prev_depth + i, z_now);
    taa_st(prev_sky + i, sky_now);
}
// (TAP: where this frame's colour and sky flag of a window pixel come from - the frame's planes (PlaneTap: k_taa, k_taa_tiles) or, for the
// pixels other ranks own, the halo records as they arrived (HaloTap: k_resolve_tiles))
struct PlaneTap {
    const float *__restrict__ current; const uint8_t *__restrict__ sky; int w;
    __device__ __forceinline__ void operator()(int sx, int sy, float &r, float &g, float &b, uint8_t &s) const
    {
        const size_t j = (size_t)sx + (size_t)sy * w;
        s = sky[j]; r = current[3 * j]; g = current[3 * j + 1]; b = current[3 * j + 2];
    }
};
template <class TAP>
__device__ __forceinline__ void taa_pixel_t(const TaaParams &T, const int x, const int y, const TAP &tap, const float *__restrict__ normal,
                                            const float *__restrict__ depth, float *__restrict__ hist,
                                            float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    if (x >= T.w || y >= T.h) return;
    const size_t i = (size_t)x + (size_t)y * T.w;
    float cr, cg, cb;
    uint8_t sky_now;
    tap(x, y, cr, cg, cb, sky_now);
    const float nx = taa_ld(normal + 3 * i), ny = taa_ld(normal + 3 * i + 1), nz = taa_ld(normal + 3 * i + 2);
    const float z_now = taa_ld(depth + i);
    if (T.reset) { taa_reset(i, cr, cg, cb, nx, ny, nz, z_now, sky_now, hist, prev_normal, prev_depth, prev_sky); return; }
    float min_l = YCGE_INF, max_l = -YCGE_INF;
    const int r = T.radius;
    if (r == 1) {       // the default window: all nine taps fetched before any is looked at (the loop below waits for a tap's sky flag
                        // before it asks for the colour, nine times in a row); same comparisons in the same order
        uint8_t sk[9];
        float lr[9], lg[9], lb[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int sy = y + k / 3 - 1; if (sy < 0) sy = 0; else if (sy >= T.h) sy = T.h - 1;
            int sx = x + k % 3 - 1; if (sx < 0) sx = 0; else if (sx >= T.w) sx = T.w - 1;
            tap(sx, sy, lr[k], lg[k], lb[k], sk[k]);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float l = luma(lr[k], lg[k], lb[k]);
            const bool use = sk[k] == sky_now;
            if (use && l < min_l) min_l = l;
            if (use && l > max_l) max_l = l;
        }
    } else
    for (int oy = -r; oy <= r; oy++) {
        int sy = y + oy; if (sy < 0) sy = 0; else if (sy >= T.h) sy = T.h - 1;
        for (int ox = -r; ox <= r; ox++) {
            int sx = x + ox; if (sx < 0) sx = 0; else if (sx >= T.w) sx = T.w - 1;
            float tr, tg, tb; uint8_t ts;
            tap(sx, sy, tr, tg, tb, ts);
            if (ts != sky_now) continue;
            float l = luma(tr, tg, tb);
            if (l < min_l) min_l = l;
            if (l > max_l) max_l = l;
        }
    }
    taa_blend(T, i, cr, cg, cb, nx, ny, nz, z_now, sky_now, min_l, max_l, hist, prev_normal, prev_depth, prev_sky);
}
__device__ __forceinline__ void taa_pixel(const TaaParams &T, const int x, const int y, const float *__restrict__ current, const float *__restrict__ normal,
                                             const float *__restrict__ depth, const uint8_t *__restrict__ sky, float *__restrict__ hist,
                                             float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    const PlaneTap tap = {current, sky, T.w};
    taa_pixel_t(T, x, y, tap, normal, depth, hist, prev_normal, prev_depth, prev_sky);
}

// One 64-thread workgroup per schedule entry, listed longest first (k_cost_scatter).  An entry is an 8x8 pixel block
// (4 per tile) or - experiment knob YCGE_SPLIT, off by default - one PART of a block: a wavefront with few live lanes
// steps faster than a full one (one lane's path through the node / triangle / pop code instead of all of them,
// fewer divergent fetches per step), so a heavy block can be split over 4, 16 or 64 wavefronts of 16, 4 or 1 pixels.
// Which lane computes a pixel never changes the pixel.
//   entry = block (22 bits) | part << 22 (6 bits) | log2(parts) << 28
#define YCGE_ENT_BLOCK(e) ((e) & 0x3fffffu)
#define YCGE_ENT_PART(e) (((e) >> 22) & 63u)
#define YCGE_ENT_LG(e) ((e) >> 28)
// Query fan-out (k_trace_fan): the heaviest blocks of the previous frame get THREE wavefronts.  The queries a
// diffuse hit gives rise to - the first shadow segment towards each of the first two lit lights, and the bounce
// ray - depend on the hit but not on each other (TraceFull only consumes them one after the other,
// RaytraceRenderer.cs:578-616; the bounce direction is the next draw of the pixel's generator whatever the shadow
// rays return), so the three wavefronts trace them side by side and wavefront 0 then runs TraceFull's loop over
// the stored answers in the reference's order: same queries, same answers, same additions in the same order.  A
// block's chain becomes  primary + max(shadow, shadow, bounce) + max(shadow, shadow)  instead of the sum of the six.
// Every other query (path items of refractive hits, later transmittance segments, a third light) goes through
// slot 0 one at a time.  Slots live in LDS; a stage boundary is one workgroup barrier.
struct FanShared {
    float o[3][64];             // origin of the lane's posted queries (the queries of one hit share it), xyz, lane
    float q[3][6][64];          // slot, {d xyz, tmin (< 0: empty), tmax, any-hit flag}, lane
    float r[3][3][64];          // slot, {t, prim, sub}, lane
    uint32_t iters[3];          // traversal loop iterations per wavefront (summed into the block's cost)
    int alive;
    uint8_t list[192];          // refill mode: the posted queries, compacted (slot * 64 + lane)
};
// The same slots for ONE wavefront that traces a PART of a split block (trace_block MODE 3): at most 16 pixel lanes post, the wavefront's
// 64 lanes answer - lane = slot * (pixels of the part) + pixel.  1.9 KB beside the 10 KB a k_trace wavefront holds anyway.
struct FanPart {
    float o[3][16];
    float q[3][6][16];
    float r[4][3][16];          // (answer slot 3: the bounce ray's, kept while slot 0 serves the queries that go one at a time)
};
template <class FS>
__device__ __forceinline__ void fan_post(FS *F, int slot, int lane, F3 o, F3 d, float tmin, float tmax, bool anyhit)
{
    F->o[0][lane] = o.x; F->o[1][lane] = o.y; F->o[2][lane] = o.z;
    F->q[slot][0][lane] = d.x; F->q[slot][1][lane] = d.y; F->q[slot][2][lane] = d.z;
    F->q[slot][3][lane] = tmin; F->q[slot][4][lane] = tmax; F->q[slot][5][lane] = anyhit ? 1.0f : 0.0f;
}
template <class FS>
__device__ __forceinline__ RayQ fan_query(const FS *F, int slot, int lane)
{
    RayQ q;
    q.o = f3(F->o[0][lane], F->o[1][lane], F->o[2][lane]);
    q.d = f3(F->q[slot][0][lane], F->q[slot][1][lane], F->q[slot][2][lane]);
    q.tmin = F->q[slot][3][lane]; q.tmax = F->q[slot][4][lane];
    q.anyhit = F->q[slot][5][lane] != 0.0f;
    return q;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v)
{
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)v, off, 64); v = o2 > v ? o2 : v; }
    return v;
}

// The shading context of a diffuse hit (position, normal, albedo, direction to the eye) between the hit and the light-loop / bounce
// code that consumes it, per lane, 48 bytes: written once per hit, read once per shadow answer.  Explicit ds instructions (the
// compiler cannot forward through them): the twelve values are not live across the shadow queries, where they cost the 4-wavefront
// build of k_trace 15 spilled registers (135 MB of scratch writes per 1080p frame).  48-byte lane stride: b128 accesses of 16
// consecutive lanes cover every bank once.
static __shared__ __attribute__((aligned(16))) float g_shade_ctx[12 * 64];
__device__ __forceinline__ void shade_ctx_store(uint32_t addr, F3 p, F3 n, F3 alb, F3 wo)
{
    const f32x4 v0 = {p.x, p.y, p.z, n.x}, v1 = {n.y, n.z, alb.x, alb.y}, v2 = {alb.z, wo.x, wo.y, wo.z};
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32" : : "v"(addr), "v"(v0), "v"(v1), "v"(v2) : "memory");
}
__device__ __forceinline__ void shade_ctx_load(uint32_t addr, F3 &p, F3 &n, F3 &alb, F3 &wo)
{
    f32x4 v0, v1, v2;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(addr) : "memory");
    p = f3(v0.x, v0.y, v0.z); n = f3(v0.w, v1.x, v1.y); alb = f3(v1.z, v1.w, v2.x); wo = f3(v2.y, v2.z, v2.w);
}

#if YCGE_EXPERIMENTS
} // namespace ycge
#include "experiments/ycge_taa_in_trace.hip.h"
namespace ycge {
#endif
#ifndef YCGE_PARTFAN
// k_trace / k_trace_batch: 0 = round 5's loop (MODE 0), the product.  1 = MODE 3 for every entry, 2 = MODE 3 for the parts of split blocks and
// MODE 0 for whole blocks (two copies of the loop in one kernel).  Round 6 built MODE 3 - the parts of split blocks fan their queries out over
// their idle lanes - and measured it (profiles/r06/a_partfan.txt): a rank of 8 / 4 in the tile-resident ring 0.173 -> 0.157 / 0.267 -> 0.231 ms,
// config 3 at 256 split blocks 0.280 -> 0.275 ms; config 4's whole frame LOSES (0.480 -> 0.499 ms at 32 split blocks, more with more: the
// machine is within 15 % of full, every part is a wavefront slot, and MODE 3's loop runs whole blocks 4 % slower - 168 registers against 157).
// And the build with MODE 3 in it dies with a GPU memory fault where two contexts trace on one device (test_one_call_drives_several_devices,
// 9 of 9 runs; not with the register path alone, -DYCGE_PARTFAN_NOFAN=1, nor with the slots zeroed first, -DYCGE_PARTFAN_ZEROLDS=1): cause
// not found.  Not in the product; the variants stay buildable for whoever follows it up.
#define YCGE_PARTFAN 0
#endif
#ifndef YCGE_PARTFAN_COST_NUM
#define YCGE_PARTFAN_COST_NUM 6  // quarters: the iterations a fanned part reports are scaled by this / 4 so that its block keeps its schedule class
#endif
// MODE 0: one wavefront per block, queries traced where TraceFull asks for them.
// MODE 1: k_trace_fan, three wavefronts per block (see above).
// MODE 2: k_trace_refill, ONE wavefront per block and the same posting of a hit's queries, but stage B is a refill loop:
//         the block's posted queries (up to 192) form a list, a lane that finishes its query takes the next one whichever
//         pixel it belongs to, and the walk yields every `refill_steps` steps so that idle lanes can do so.  A block then
//         costs about max(its longest query, its steps / 64) per stage instead of the sum of the stage's longest lanes.
// MODE 3: k_trace since round 6.  ONE wavefront per schedule entry as in MODE 0, and the query fan-out of MODE 1 INSIDE that wavefront
//         where the entry is a PART of a split block (lg >= 2: at most 16 of the 64 lanes hold a pixel): the part's pixels post the
//         queries of a diffuse hit - first shadow segment towards each of the first two lit lights, the bounce ray - into three slots
//         and lane = slot * pixels + pixel answers them in ONE pass of the traversal loop, on lanes that idled before.  A part's chain
//         becomes primary + max(shadow, shadow, bounce) + max(shadow, shadow) with no wavefront slot more than the split already took
//         (k_trace_fan paid two helper wavefronts a block, which is why it lost on whole frames).  Entries that are whole blocks
//         (lg < 2) run the same loop with the pixel's own query in registers: stage B traces `q`, stage A consumes - MODE 0's order.
template <bool COUNT, bool FLAT, int MODE, bool FULLW = true, class FS = FanShared>
__device__ __forceinline__ void trace_block(const SceneDev &S, const FrameParams &P, const TraceOut &O, const uint32_t ent, const uint32_t sched_index,
                                            FS *F, const int refill_steps)
{
    constexpr bool FAN = MODE != 0;          // queries are posted to LDS slots and answered in stage B
    constexpr bool WAVES3 = MODE == 1;
    constexpr bool INW = MODE == 3;
    const bool DEBUG = O.prim_id != nullptr;
    Work w = {0, 0, 0, 0, 0, 0, 0};
    StackT<WAVES3 ? 192 : 64> st;
    st.init(O.stack_spill, O.stack_lanes, O.lane_base);
    const PathStack pstack = {O.path_stack, O.stack_lanes, O.lane_base};
    const int lane = (int)(threadIdx.x & 63u), wave = WAVES3 ? (int)(threadIdx.x >> 6) : 0;
    const uint32_t bid = YCGE_ENT_BLOCK(ent), lg = YCGE_ENT_LG(ent);
    const int k = (int)(bid >> 2), wave_in_tile = (int)(bid & 3);
    const int live_lanes = 64 >> lg;
#ifndef YCGE_PARTFAN_NOFAN
#define YCGE_PARTFAN_NOFAN 0            // debugging aid: MODE 3's loop with every entry on the register path
#endif
    const bool fanrt = !INW || (lg >= 2u && !YCGE_PARTFAN_NOFAN);        // (wave-uniform) MODE 3: this entry fans its queries out over its idle lanes
    const int pix_in_block = (int)YCGE_ENT_PART(ent) * live_lanes + lane;
    int px, py, lx, ly;
    const bool in_image = tile_pixel_wl(P, k, wave_in_tile, pix_in_block & 63, px, py, lx, ly) && lane < live_lanes;
    const bool prof = O.wave_prof && O.wave_prof_stage == 2;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();     // 100 MHz, chip-wide
    if (lg) __builtin_amdgcn_s_setprio(3);      // the frame's critical path: issue priority over the light blocks sharing the SIMD

    RayQ q;
    make_primary_ray(P, px, py, q.o, q.d);
    q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX; q.anyhit = false;
    if (in_image && O.rays && wave == 0) {
        float *r = O.rays + ((size_t)px + (size_t)py * P.hiW) * 6;
        r[0] = q.o.x; r[1] = q.o.y; r[2] = q.o.z; r[3] = q.d.x; r[4] = q.d.y; r[5] = q.d.z;
    }

    // ---- TraceFull state, RaytraceRenderer.cs:448-460
    uint64_t rng = per_frame_seed(px, py, P.frame, P.seed_salt);
    if (rng == 0) rng = 0x9E3779B97F4A7C15ULL;
    F3 radiance = f3(0, 0, 0), beta = f3(1, 1, 1);
    int mirror_depth = 0, diffuse_depth = 0;
    bool item_is_primary = true, primary_hit_something = false, gbuf_valid = false;
    // The G-buffer (PrimaryGBuffer, :400-405) and the sky flag are decided by the pixel's FIRST query - only that work item is primary
    // (:452, :462) - and are written out right there (write_gbuffer): nine values less to carry through every later query.
    // (the pixel index goes through an empty asm at every use: otherwise the compiler computes the six 64-bit store addresses of a
    // pixel once, in the prologue, keeps them live through the whole kernel - and spills them)
    auto pixel_index = [&]() { int x = px, y = py; asm volatile("" : "+v"(x), "+v"(y)); return (size_t)x + (size_t)y * P.hiW; };
#if YCGE_EXPERIMENTS
    const bool fuse = !FAN && O.taa.block_ctr != nullptr;          // (wave-uniform) experiments/ycge_taa_in_trace.hip.h: this launch resolves TAA itself, the planes TAA reads are written through
#else
    constexpr bool fuse = false;
#endif
    auto write_gbuffer = [&](F3 g_albedo, F3 g_normal, float g_depth, int g_prim, int g_sub, bool is_sky) {
        const size_t i = pixel_index();
        // (written once, read by TAA / the post stage after the launch: past the caches' keep lists, the tree stays in the L2s)
        out_st(O.g_albedo + 3 * i + 0, g_albedo.x); out_st(O.g_albedo + 3 * i + 1, g_albedo.y); out_st(O.g_albedo + 3 * i + 2, g_albedo.z);
        out_st(O.g_normal + 3 * i + 0, g_normal.x, fuse); out_st(O.g_normal + 3 * i + 1, g_normal.y, fuse); out_st(O.g_normal + 3 * i + 2, g_normal.z, fuse);
        out_st(O.g_depth + i, g_depth, fuse);
        out_st(O.sky + i, (uint8_t)(is_sky ? 1 : 0), fuse);
        if (DEBUG) {
            if (O.prim_id) O.prim_id[i] = g_prim;
            if (O.sub_id) O.sub_id[i] = g_sub;
            if (O.hit_t) O.hit_t[i] = g_depth;
        }
    };
    int psp = 0;            // a 16-entry stack in the C#; occupancy never exceeds 3 (mirrorDepth < 2 gates pushes)
    // shading context kept across the shadow queries of one hit
    // (position, normal, albedo, direction to the eye: parked in LDS between the hit and the light / bounce code that reads them,
    // g_shade_ctx - twelve registers that would otherwise be live across every shadow query)
    const uint32_t ctx_addr = (uint32_t)(uintptr_t)g_shade_ctx + (uint32_t)lane * 48u;
    int light = 0;
    float tr_r = 1.0f, tr_g = 1.0f, tr_b = 1.0f, sh_maxdist = 0.0f;
    int tr_counter = 0;
    int phase = (in_image && wave == 0) ? PH_PATH : PH_DONE;
    // fan-out state (FAN only): `want` = where the answer to the pending query q is (slot 0-2, 3 = the bounce answer
    // held in registers), parked = that answer is still to be traced by the next stage B
    int want = 0, pre_l1 = -1, pre_l2 = -1, pre_b_prim = -1, pre_b_sub = 0;
    float pre_b_t = 0.0f;
    bool parked = FAN, pre_b = false, bounce_in_flight = false;
    // the block's cost for the next frame's schedule: loop iterations its wavefront(s) spend in traversal = sum over the
    // query batches of the longest lane's steps.  The same scale whether the block is fanned or not.
    uint32_t wave_iters = 0;
    float ans_t = 0.0f;                     // MODE 3, whole-block entries: the answer to the lane's own query, from stage B to stage A
    int ans_prim = -1, ans_sub = 0;
#if defined(YCGE_PARTFAN_ZEROLDS)
    if constexpr (INW) { for (int z = lane; z < (int)(sizeof(FS) / 4); z += 64) ((float *)F)[z] = 0.0f; }
#endif
    if (FAN && fanrt) {
        if (WAVES3) F->q[wave][3][lane] = -1.0f;
        else if (!INW || lane < live_lanes) { F->q[0][3][lane] = -1.0f; F->q[1][3][lane] = -1.0f; F->q[2][3][lane] = -1.0f; }
        if (phase != PH_DONE) fan_post(F, 0, lane, q.o, q.d, q.tmin, q.tmax, false);      // the primary query
    }

    // what a block leaves behind when its last pixel is done (or handed over): the pixels, the schedule feedback, the profile record
    auto finish_block = [&]() {
    if (in_image) {                                     // :210-215
        const size_t i = pixel_index();
        out_st(O.current_hdr + 3 * i + 0, radiance.x, fuse); out_st(O.current_hdr + 3 * i + 1, radiance.y, fuse); out_st(O.current_hdr + 3 * i + 2, radiance.z, fuse);
        if (DEBUG && O.rng_state) O.rng_state[i] = rng;
    }
#if YCGE_EXPERIMENTS
    if (fuse) taa_in_trace(P, O, bid, lg, lane);
#endif
    // a part of a split block sees fewer lanes, hence fewer iterations than the whole block would: scaled so that the block
    // stays in its schedule class from frame to frame (x 1.5 for 4 parts, x 2 for 16, x 2.5 for 64: measured ratios are 1.3-2)
    uint32_t part_iters = wave_iters;
    if constexpr (WAVES3) part_iters = F->iters[0] + F->iters[1] + F->iters[2];
    // (MODE 3: a part that fans its queries out walks max(shadow, shadow, bounce) where the block walked their sum: x YCGE_PARTFAN_COST_NUM / 4 on top)
    const uint32_t part_scaled = part_iters + ((part_iters * lg) >> 2);
    const uint32_t wave_max_steps = (INW && fanrt) ? (part_scaled * YCGE_PARTFAN_COST_NUM) >> 2 : part_scaled;
    if (O.block_cost && lane == 0) atomicMax(O.block_cost + bid, wave_max_steps);      // feedback for the next frame's schedule
    if (prof && lane == 0 && YCGE_ENT_PART(ent) == 0) {
        unsigned long long *dst = O.wave_prof + ((size_t)k * 4 + wave_in_tile) * 4;
        dst[0] = t_start; dst[1] = __builtin_amdgcn_s_memrealtime(); dst[2] = sched_index | ((unsigned long long)wave_max_steps << 32);
        dst[3] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 20) | ((unsigned long long)(lg | (WAVES3 ? 8u : 0u)) << 32);   // XCC_ID; bit 3 = fanned block
    }
    };

  for (;;) {          // FAN: one round = stage A (wavefront 0 consumes answers, posts queries) + stage B (all trace)
   if (!FAN || wave == 0) {
    for (;;) {
        float t_hit = 0.0f;
        int hit_prim = -1, hit_sub = 0;
        if (!__any(phase != PH_DONE && !parked)) break;
        const uint32_t steps_before = w.steps;
        if (!FAN && !FULLW) {       // (the instance for scenes without a mesh: nothing to walk cooperatively, lanes without a query stay out)
            if (phase != PH_DONE) traverse<COUNT, true, FLAT, false>(S, q, st, t_hit, hit_prim, hit_sub, w);
        } else if (!FAN) {         // every lane of the wavefront enters (the cooperative walk needs them all): finished pixels carry live = false
            q.live = phase != PH_DONE;
            traverse<COUNT, true, FLAT, true>(S, q, st, t_hit, hit_prim, hit_sub, w);
        } else if (phase != PH_DONE && !parked) {
            if (INW && !fanrt) { t_hit = ans_t; hit_prim = ans_prim; hit_sub = ans_sub; }
            else if (want == 3 && !INW) { t_hit = pre_b_t; hit_prim = pre_b_prim; hit_sub = pre_b_sub; }
            else { t_hit = F->r[want][0][lane]; hit_prim = __float_as_int(F->r[want][1][lane]); hit_sub = __float_as_int(F->r[want][2][lane]); }
        }
        if (!FAN) wave_iters += wave_umax(w.steps - steps_before);
        if (phase == PH_DONE || parked) continue;
        const bool hit = hit_prim >= 0;
        F3 sh_p, sh_n, sh_alb, sh_wo;
        if (phase != PH_PATH) shade_ctx_load(ctx_addr, sh_p, sh_n, sh_alb, sh_wo);      // a shadow query's answer: back to the hit it belongs to
        else { sh_p = sh_n = sh_alb = sh_wo = f3(0, 0, 0); }
        int new_kind = 0;       // the next query: 1 = first shadow segment towards `light`, 2 = bounce, 0 = anything else
        bool fanned = false;

        bool go_lights = false, go_next_light = false, go_contrib = false, go_next_item = false;
        if (phase == PH_PATH) {
            if (!hit) {
                float tbg = 0.5f * (q.d.y + 1.0f);
                F3 sky = lerp3(f3(S.bg_bottom), f3(S.bg_top), tbg);
                if (item_is_primary && !primary_hit_something) {
                    if (!gbuf_valid) { write_gbuffer(f3(0, 0, 0), f3(0, 0, 0), YCGE_FLT_MAX, -1, 0, true); gbuf_valid = true; }
                }
                radiance = radiance + f3(beta.x * sky.x, beta.y * sky.y, beta.z * sky.z);
                go_next_item = true;
            } else {
                HitAttr h;
                resolve_hit<true>(S, hit_prim, hit_sub, t_hit, q.o, q.d, h);
                if (YCGE_TEXTURES && !FLAT && S.any_textured) apply_texture(S, hit_prim, hit_sub, q.o, q.d, h);      // (the host sends textured scenes to the generic kernels)
                if (item_is_primary) {
                    primary_hit_something = true;
                    if (!gbuf_valid) { write_gbuffer(h.m.albedo, h.n, t_hit, hit_prim, h.sub_public, false); gbuf_valid = true; }
                    item_is_primary = false;
                }
                if (h.m.emission.x != 0.0f || h.m.emission.y != 0.0f || h.m.emission.z != 0.0f) {
                    F3 e = h.m.emission;
                    radiance = radiance + f3(beta.x * e.x, beta.y * e.y, beta.z * e.z);
                }
                const F3 base_albedo = h.m.albedo;
                if (h.m.transparency > 0.0f) {          // :506-558
                    if (mirror_depth < P.max_mirror_bounces) {
                        F3 n = h.n, wo = q.d;
                        bool front = dot(n, wo) < 0.0f;
                        F3 nl = front ? n : n * -1.0f;
                        float eta_i = front ? 1.0f : h.m.ior;
                        float eta_t = front ? h.m.ior : 1.0f;
                        float eta = eta_i / eta_t;
                        F3 refl_dir = normalized(reflect(wo, nl));
                        F3 refr_dir;
                        bool has_refract = refract(wo, nl, eta, refr_dir);
                        float cos_theta = cs_abs(dot(nl, wo * -1.0f));
                        float R = fresnel_schlick(cos_theta, eta_i, eta_t);
                        float Tr = cs_clamp(h.m.transparency, 0.0f, 1.0f);
                        float T = has_refract ? (1.0f - R) * Tr : 0.0f;
                        R = cs_clamp(R + h.m.reflectivity * (1.0f - R), 0.0f, 1.0f);
                        if (R > 0.0f && psp < 3) {
                            PathItem it;
                            it.o = h.p + nl * P.eps;
                            it.d = normalized(refl_dir);
                            it.beta = f3(beta.x * base_albedo.x * R, beta.y * base_albedo.y * R, beta.z * base_albedo.z * R);
                            it.mirror_depth = mirror_depth + 1; it.diffuse_depth = diffuse_depth;
                            pstack.store(psp++, it);
                        }
                        if (T > 0.0f && psp < 3) {
                            PathItem it;
                            it.o = h.p - nl * P.eps;
                            it.d = normalized(normalized(refr_dir));
                            F3 tint = h.m.trans_color;
                            it.beta = f3(beta.x * tint.x * T, beta.y * tint.y * T, beta.z * tint.z * T);
                            it.mirror_depth = mirror_depth + 1; it.diffuse_depth = diffuse_depth;
                            pstack.store(psp++, it);
                        }
                    }
                    go_next_item = true;
                } else if (h.m.reflectivity >= P.mirror_threshold) {   // :559-570
                    if (mirror_depth >= P.max_mirror_bounces) {
                        go_next_item = true;
                    } else {
                        F3 refl_dir = normalized(reflect(q.d, h.n));
                        q.o = h.p + h.n * P.eps;
                        q.d = normalized(refl_dir);
                        q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
                        q.anyhit = false;
                        beta = f3(beta.x * base_albedo.x, beta.y * base_albedo.y, beta.z * base_albedo.z);
                        mirror_depth++;
                        phase = PH_PATH;
                    }
                } else {
                    if (S.ambient_intensity > 0.0f) {   // :571-576
                        float ai = S.ambient_intensity;
                        asm volatile("" : "+s"(ai));        // (evaluated here, per hit: hoisted out of the loop the three products sit in registers for the whole kernel)
                        F3 a = f3(S.ambient[0] * ai, S.ambient[1] * ai, S.ambient[2] * ai);
                        F3 amb = f3(a.x * base_albedo.x, a.y * base_albedo.y, a.z * base_albedo.z);
                        radiance = radiance + f3(beta.x * amb.x, beta.y * amb.y, beta.z * amb.z);
                    }
                    sh_p = h.p; sh_n = h.n; sh_alb = base_albedo;
                    sh_wo = normalized(q.d * -1.0f);
                    shade_ctx_store(ctx_addr, sh_p, sh_n, sh_alb, sh_wo);
                    light = 0;
                    go_lights = true;
                    if (FAN && fanrt) { // post this hit's independent queries: the expressions of the light loop head and the bounce below
                        int ns = 0;
                        pre_l1 = pre_l2 = -1;
                        pre_b = false;
                        for (int li = 0; li < S.n_lights && ns < 2; li++) {
                            const GLight &L = S.lights[li];
                            F3 to_l = f3(L.pos) - sh_p;
                            float dist2 = dot(to_l, to_l);
                            float dist = cs_sqrt(dist2);
                            F3 ldir = vdiv(to_l, dist);
                            float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                            if (n_dot_l <= 0.0f) continue;
                            if (light_is_dark<COUNT>(L, dist2, w)) continue;
                            fan_post(F, 1 + ns, lane, sh_p + sh_n * P.eps, normalized(ldir), S.is_volume_scene ? 0.001f : 0.0f + P.eps, dist - P.eps, S.is_volume_scene || !S.any_transparent);
                            if (ns == 0) pre_l1 = li; else pre_l2 = li;
                            ns++;
                        }
                        if (diffuse_depth < P.diffuse_bounces) {
                            uint64_t rng_peek = rng;        // the draw itself happens at the bounce, below
                            F3 bounce = cosine_sample_hemisphere(sh_n, rng_peek);
                            fan_post(F, 0, lane, sh_p + sh_n * P.eps, normalized(bounce), 0.001f, YCGE_FLT_MAX, false);
                            bounce_in_flight = true;
                        }
                        fanned = ns > 0 || bounce_in_flight;
                    }
                }
            }
        } else if (phase == PH_SHADOW_OCC) {            // VolumeScene: binary occlusion, :761-765
            if (hit) { tr_r = tr_g = tr_b = 0.0f; } else { tr_r = tr_g = tr_b = 1.0f; }
            go_contrib = true;
        } else {                                        // ComputeTransmittanceToLight loop body, :773-796
            if (!hit) {
                go_contrib = true;
            } else if (!S.any_transparent) {            // no material with Transparency > 0: `tr <= 0` for whatever was hit, :779-781
                tr_r = tr_g = tr_b = 0.0f; go_contrib = true;
            } else {
                tr_counter++;
                HitAttr h;
                resolve_hit<true>(S, hit_prim, hit_sub, t_hit, q.o, q.d, h);
                float tr = h.m.transparency;
                if (tr <= 0.0f) { tr_r = tr_g = tr_b = 0.0f; go_contrib = true; }
                else {
                    F3 tint = h.m.trans_color;
                    tr_r *= tint.x * tr; tr_g *= tint.y * tr; tr_b *= tint.z * tr;
                    if (tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f) { tr_r = tr_g = tr_b = 0.0f; go_contrib = true; }
                    else if (t_hit > sh_maxdist) go_contrib = true;
                    else {
                        q.tmin = t_hit + P.eps;
                        if (!(tr_counter < P.max_refractions)) go_contrib = true;
                    }
                }
            }
        }

        if (go_contrib) {                               // :592-602
            if (!(tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f)) {
                const GLight &L = S.lights[light];
                F3 to_l = f3(L.pos) - sh_p;
                float dist2 = dot(to_l, to_l);
                float dist = cs_sqrt(dist2);
                F3 ldir = vdiv(to_l, dist);
                float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                float atten = L.intensity / dist2;
                F3 f_diffuse = oren_nayar(sh_alb, sh_n, sh_wo, ldir, P.on_a, P.on_b);
                F3 Li = f3(L.color) * atten;
                F3 contrib = (f_diffuse * n_dot_l) * Li;
                contrib = f3(contrib.x * tr_r, contrib.y * tr_g, contrib.z * tr_b);
                radiance = radiance + f3(beta.x * contrib.x, beta.y * contrib.y, beta.z * contrib.z);
            }
            light++;
            go_next_light = true;
        }

        if (go_lights || go_next_light) {               // light loop head, :578-591
            bool queued = false;
            for (; light < S.n_lights; light++) {
                const GLight &L = S.lights[light];
                F3 to_l = f3(L.pos) - sh_p;
                float dist2 = dot(to_l, to_l);
                float dist = cs_sqrt(dist2);
                F3 ldir = vdiv(to_l, dist);
                float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                if (n_dot_l <= 0.0f) continue;
                if (light_is_dark<COUNT>(L, dist2, w)) continue;       // nothing to add whatever the query says (GLight::dark)
                q.o = sh_p + sh_n * P.eps;
                q.d = normalized(ldir);
                sh_maxdist = dist - P.eps;
                q.tmax = sh_maxdist;
                q.anyhit = S.is_volume_scene || !S.any_transparent;          // every occluder is opaque: the first hit answers the query
                if (S.is_volume_scene) { q.tmin = 0.001f; phase = PH_SHADOW_OCC; }
                else { q.tmin = 0.0f + P.eps; tr_r = tr_g = tr_b = 1.0f; tr_counter = 0; phase = PH_SHADOW_TR; }
                queued = true;
                new_kind = 1;
                break;
            }
            if (!queued) {                              // bounce, :604-616
                if (diffuse_depth < P.diffuse_bounces) {
                    F3 bounce = cosine_sample_hemisphere(sh_n, rng);
                    F3 f_on = oren_nayar(sh_alb, sh_n, sh_wo, bounce, P.on_a, P.on_b);
                    const float factor = YCGE_PI;
                    F3 mult = f3(f_on.x * factor, f_on.y * factor, f_on.z * factor);
                    q.o = sh_p + sh_n * P.eps;
                    q.d = normalized(bounce);
                    q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
                    q.anyhit = false;
                    beta = f3(beta.x * mult.x, beta.y * mult.y, beta.z * mult.z);
                    diffuse_depth++;
                    phase = PH_PATH;
                    new_kind = 2;
                } else {
                    go_next_item = true;
                }
            }
        }

        if (go_next_item) {                             // outer while (sp > 0), :461-468
            if (psp == 0) {
                phase = PH_DONE;
            } else {
                PathItem it = pstack.load(--psp);
                q.o = it.o; q.d = it.d; q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
                q.anyhit = false;
                beta = it.beta; mirror_depth = it.mirror_depth; diffuse_depth = it.diffuse_depth;
                item_is_primary = false;
                phase = PH_PATH;
            }
        }

        if (INW && !fanrt) parked = true;               // (a whole block: the query just set up waits in `q` for stage B)
        else if (FAN && phase != PH_DONE) {             // where is the answer to the query just set up?
            if (new_kind == 1 && light == pre_l1) { want = 1; pre_l1 = -1; }
            else if (new_kind == 1 && light == pre_l2) { want = 2; pre_l2 = -1; }
            else if (new_kind == 2 && (pre_b || bounce_in_flight)) { want = 3; pre_b = false; }
            else { fan_post(F, 0, lane, q.o, q.d, q.tmin, q.tmax, q.anyhit); want = 0; parked = true; }
            if (fanned) parked = true;
        }
    }
    if constexpr (WAVES3) { const bool alive = __any(phase != PH_DONE); if (lane == 0) F->alive = alive ? 1 : 0; }
   }
   if (!FAN) break;
   if constexpr (WAVES3) {
       __syncthreads();
       if (!F->alive) break;
       // ---- stage B: wavefront w answers slot w
       const float f_tmin = F->q[wave][3][lane];
       const uint32_t steps_before = w.steps;
       {
           RayQ fq = fan_query(F, wave, lane);
           fq.live = f_tmin >= 0.0f;
           float f_t; int f_prim, f_sub;
           traverse<COUNT, true, FLAT, true>(S, fq, st, f_t, f_prim, f_sub, w);
           if (fq.live) {
           F->r[wave][0][lane] = f_t; F->r[wave][1][lane] = __int_as_float(f_prim); F->r[wave][2][lane] = __int_as_float(f_sub);
           F->q[wave][3][lane] = -1.0f;
           }
       }
       wave_iters += wave_umax(w.steps - steps_before);
       __syncthreads();
   } else if constexpr (INW) {
       if (!__any(phase != PH_DONE)) break;
       // ---- stage B inside the one wavefront: lane = slot * pixels + pixel answers that pixel's slot (a part), or its own query (a whole block)
       __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
       __builtin_amdgcn_wave_barrier();
       const uint32_t steps_before = w.steps;
       const int b_slot = lane >> (6 - (int)lg), b_src = lane & (live_lanes - 1);
       const bool b_valid = fanrt && b_slot < 3;
       // (`q` itself carries the slot's query through the walk - a second ray beside it cost the kernel its last free registers - and the
       // pixel lanes take their pending query back from its slot afterwards: origin, direction, range, kind; tmin is set anew by whoever walks on)
       q.live = phase != PH_DONE;
       if (fanrt) {
           q = fan_query(F, b_valid ? b_slot : 0, b_src);
           q.live = b_valid && q.tmin >= 0.0f;
       }
       traverse<COUNT, true, FLAT, true>(S, q, st, ans_t, ans_prim, ans_sub, w);
       if (fanrt) {
           if (q.live) {
               F->r[b_slot][0][b_src] = ans_t; F->r[b_slot][1][b_src] = __int_as_float(ans_prim); F->r[b_slot][2][b_src] = __int_as_float(ans_sub);
               F->q[b_slot][3][b_src] = -1.0f;
           }
           __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
           __builtin_amdgcn_wave_barrier();
           if (phase != PH_DONE) q = fan_query(F, want == 3 ? 0 : want, lane);
       }
       wave_iters += wave_umax(w.steps - steps_before);
   } else {
#if YCGE_EXPERIMENTS
       if (!__any(phase != PH_DONE)) break;
       // ---- stage B: the posted queries as one list, lanes refill from it
       uint32_t n = 0;
       for (int sl = 0; sl < 3; sl++) {
           const bool v = F->q[sl][3][lane] >= 0.0f;
           const unsigned long long m = __ballot(v);
           if (v) F->list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint8_t)(sl * 64 + lane);
           n += (uint32_t)__popcll(m);
       }
       uint32_t next = 0;
       bool has = false;
       int id = 0;
       FlatQuery fq;
       RayQ rq;
       rq.o = f3(0, 0, 0); rq.d = f3(0, 0, 1); rq.tmin = 0.0f; rq.tmax = 0.0f;
       fq.cur = YCGE_REF_NONE_VALUE;
       for (;;) {
           const unsigned long long idle = __ballot(!has);
           if (next < n && idle != 0ull) {
               const uint32_t my = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
               if (!has && my < n) {
                   id = (int)F->list[my];
                   rq = fan_query(F, id >> 6, id & 63);
                   flat_begin<COUNT>(S, rq, st, fq, w);
                   has = true;
               }
               next += (uint32_t)__popcll(idle);
           }
           if (!__any(has)) break;
           wave_iters += (uint32_t)refill_steps;
           if (has && flat_advance<COUNT, true>(S, rq, st, fq, w, refill_steps)) {
               const int sl = id >> 6, ln = id & 63;
               F->r[sl][0][ln] = fq.closest; F->r[sl][1][ln] = __int_as_float(fq.hit_prim); F->r[sl][2][ln] = __int_as_float(fq.hit_sub);
               F->q[sl][3][ln] = -1.0f;
               has = false;
           }
       }
#else
       break;          // (MODE 2, k_trace_refill, exists in experiment builds only)
#endif
   }
   parked = false;
   if (bounce_in_flight) {      // slot 0 is needed for the queries that go one at a time: keep the bounce answer in registers (MODE 3: in a fourth answer slot)
        if constexpr (INW) { F->r[3][0][lane] = F->r[0][0][lane]; F->r[3][1][lane] = F->r[0][1][lane]; F->r[3][2][lane] = F->r[0][2][lane]; }
        else { pre_b_t = F->r[0][0][lane]; pre_b_prim = __float_as_int(F->r[0][1][lane]); pre_b_sub = __float_as_int(F->r[0][2][lane]); }
        bounce_in_flight = false; pre_b = true;
   }
  }

    if constexpr (WAVES3) {
        if (lane == 0) F->iters[wave] = wave_iters;
        __syncthreads();
        if (wave != 0) { flush_work<COUNT>(w, O.counters); return; }
    }
    finish_block();
    flush_work<COUNT>(w, O.counters);
}

// Occupancy targets of the flat, non-counting variants.  Round 2 ran k_trace at 4 wavefronts per SIMD (a 128-register budget, 19 spilled
// registers) because the fanned head of the schedule needed the extra slots.  Round 3: with the cooperative walk the tails are short,
// fan-out is off on whole frames, and 3 wavefronts per SIMD (168 registers: no scratch at all) are faster - each wavefront runs with
// less contention, and the frame is bound by its longest chains and the late starters, not by slots (config 4, same box: 0.514 ms
// at 3 wavefronts, 0.544 at 4; config 3: 0.306 / 0.331).
#ifndef YCGE_TRACE_WAVES
#define YCGE_TRACE_WAVES 3
#endif
#ifndef YCGE_FAN_WAVES
#define YCGE_FAN_WAVES 3        // k_trace_fan: no register cap below its natural ~150 (a fanned block's three wavefronts are latency chains, not throughput)
#endif
template <bool COUNT, bool FLAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((FLAT && !COUNT) ? YCGE_TRACE_WAVES : 2, 8))) void k_trace(const SceneDev S, const FrameParams P, const TraceOut O)
{
    if (O.placed_flag && blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) __hip_atomic_store(O.placed_flag, O.placed_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t idx = blockIdx.x, ent = blockIdx.x;
    if (O.block_order) {
        if (O.n_fan) idx += *O.n_fan;           // the first n_fan entries belong to k_trace_fan
        if (idx >= *O.n_order) return;
        ent = O.block_order[idx];
    } else if (blockIdx.x >= (uint32_t)P.n_owned_tiles * 4u) return;
#if YCGE_PARTFAN == 2      // parts through MODE 3, whole blocks through round 5's MODE 0: two copies of the loop in one kernel (A/B)
    __shared__ FanPart F;
    if (YCGE_ENT_LG(ent) >= 2u) trace_block<COUNT, FLAT, 3, true, FanPart>(S, P, O, ent, idx, &F, 0);
    else trace_block<COUNT, FLAT, 0>(S, P, O, ent, idx, (FanShared *)nullptr, 0);
#elif YCGE_PARTFAN
    __shared__ FanPart F;
    trace_block<COUNT, FLAT, 3, true, FanPart>(S, P, O, ent, idx, &F, 0);
#else
    trace_block<COUNT, FLAT, 0>(S, P, O, ent, idx, (FanShared *)nullptr, 0);
#endif
}
// SEVERAL frames' blocks in one launch (ycge_trace_tiles_resident_batch: a rank's tiles of n consecutive frames): workgroup b traces
// schedule entry b / n of frame b % n - the frames share one schedule, so the heaviest blocks of every frame go first - with that frame's
// parameters and outputs (records in the launch's own arguments, read with scalar loads: the index is wave-uniform).  What one frame's launch leaves
// idle around its longest chains, the other frames' blocks fill without any help from the queues.  The frames' TraceOut records share ONE
// spill area n times as wide (the column is the workgroup's index in the launch).
struct TraceBatch { FrameParams P[YCGE_TRACE_BATCH_MAX]; TraceOut O[YCGE_TRACE_BATCH_MAX]; };        // kernel ARGUMENT (2.7 KB of the 4 KB a launch may carry): nothing to stage, nothing a host that runs ahead could overwrite
template <bool COUNT, bool FLAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((FLAT && !COUNT) ? YCGE_TRACE_WAVES : 2, 8))) void k_trace_batch(const SceneDev S, const TraceBatch B, const uint32_t n)
{
    const uint32_t f = blockIdx.x % n, i = blockIdx.x / n;
    const FrameParams &P = B.P[f];
    const TraceOut &O = B.O[f];
    uint32_t idx = i, ent = i;
    if (O.block_order) {
        if (idx >= *O.n_order) return;
        ent = O.block_order[idx];
    } else if (i >= (uint32_t)P.n_owned_tiles * 4u) return;
#if YCGE_PARTFAN == 2      // parts through MODE 3, whole blocks through round 5's MODE 0: two copies of the loop in one kernel (A/B)
    __shared__ FanPart F;
    if (YCGE_ENT_LG(ent) >= 2u) trace_block<COUNT, FLAT, 3, true, FanPart>(S, P, O, ent, idx, &F, 0);
    else trace_block<COUNT, FLAT, 0>(S, P, O, ent, idx, (FanShared *)nullptr, 0);
#elif YCGE_PARTFAN
    __shared__ FanPart F;
    trace_block<COUNT, FLAT, 3, true, FanPart>(S, P, O, ent, idx, &F, 0);
#else
    trace_block<COUNT, FLAT, 0>(S, P, O, ent, idx, (FanShared *)nullptr, 0);
#endif
}
// The same kernel for scenes WITHOUT a mesh (nothing to walk cooperatively: the treelet code is compiled out and, with it, the register
// peak): 125 registers, 4 wavefronts per SIMD - analytic scenes are throughput, not chains (config 2: 0.082 ms at 3 wavefronts, 0.068 at 4).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_trace_nomesh(const SceneDev S, const FrameParams P, const TraceOut O)
{
    if (O.placed_flag && blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) __hip_atomic_store(O.placed_flag, O.placed_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t idx = blockIdx.x, ent = blockIdx.x;
    if (O.block_order) {
        if (O.n_fan) idx += *O.n_fan;
        if (idx >= *O.n_order) return;
        ent = O.block_order[idx];
    } else if (blockIdx.x >= (uint32_t)P.n_owned_tiles * 4u) return;
    trace_block<false, true, 0, false>(S, P, O, ent, idx, (FanShared *)nullptr, 0);
}
#if YCGE_EXPERIMENTS
template <bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(COUNT ? 2 : 3, 8))) void k_trace_refill(const SceneDev S, const FrameParams P, const TraceOut O, const int refill_steps)
{
    __shared__ FanShared F;
    uint32_t idx = blockIdx.x, ent = blockIdx.x;
    if (O.block_order) {
        if (idx >= *O.n_order) return;
        ent = O.block_order[idx];
    } else if (blockIdx.x >= (uint32_t)P.n_owned_tiles * 4u) return;
    trace_block<COUNT, true, 2>(S, P, O, ent, idx, &F, refill_steps);
}
#endif
template <bool COUNT, bool FLAT>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu((FLAT && !COUNT) ? YCGE_FAN_WAVES : 2, 8))) void k_trace_fan(const SceneDev S, const FrameParams P, const TraceOut O)
{
    __shared__ FanShared F;
    if (blockIdx.x >= *O.n_fan) return;
    trace_block<COUNT, FLAT, 1>(S, P, O, O.block_order[blockIdx.x], blockIdx.x, &F, 0);
}

// ---------------------------------------------------------------------------------- block schedule (feedback from the previous frame)
// cost[b] = traversal loop iterations of block b's wavefront(s) (trace_block).  Eight POLICY classes, finer towards the top
// (class 7 = longest): < 64, < 128, < 192, < 256, < 384, < 512, < 768, more - what the fan-out / split knobs speak of - and 32
// ORDER classes nested inside them (order_class), finer towards the bottom too: the schedule is written order class by order class,
// longest first.  (With eight classes everything below 64 iterations - nine blocks in ten, up to ~90 us each - came in index order,
// and the frame drained for 0.1 ms behind 100-us blocks that had started last: round 3.)  k_cost_hist counts entries per order class,
// k_cost_scatter writes the schedule and clears cost[] for the next frame's atomicMax; both batch their global atomics through LDS.
// ws: [16] total entries (read by k_trace), [18] = entries at the head of the schedule that go to k_trace_fan,
//     [32..63] entries per order class, [64..95] cursors
#define YCGE_ORDER_CLASSES 32
__device__ __forceinline__ int cost_class(uint32_t c)
{
    return c < 64u ? 0 : c < 128u ? 1 : c < 192u ? 2 : c < 256u ? 3 : c < 384u ? 4 : c < 512u ? 5 : c < 768u ? 6 : 7;
}
// 0..7: eight steps of 8 below 64; then four steps per octave-ish policy class pair: [64,128) by 16, [128,256) by 32, [256,512) by 64,
// [512,1024) by 128, [1024, ...) by 256 up to class 31.  Every policy-class boundary (64, 128, 192, 256, 384, 512, 768) is a boundary here.
// Round 6: class 0 is cost ZERO alone - blocks whose rays met nothing to walk (sky: a wavefront of 2-3 us, every frame) - and [1, 16) is class 1.
// The schedule ends on its lowest class, and with [0, 8) as one class in index order the LAST entries of config 4 were floor blocks whose four-frame
// maximum was a handful of iterations and whose bounce rays, this frame, found the mesh: 60-80 us wavefronts that started at 0.40 ms and ended the
// launch at 0.483 ms, 17 us behind the longest chain (profiles/r06/z_mega_prof_config4.txt).  With the sky last they start while half the
// frame's (trivial) entries are still to come.  -DYCGE_ORDER_ZERO_LAST=0: round 5's classes (A/B).
#ifndef YCGE_ORDER_ZERO_LAST
#define YCGE_ORDER_ZERO_LAST 1
#endif
__device__ __forceinline__ int order_class(uint32_t c)
{
    if (YCGE_ORDER_ZERO_LAST) { if (c == 0u) return 0; if (c < 16u) return 1; }
    if (c < 64u) return (int)(c >> 3);
    if (c < 128u) return 8 + (int)((c - 64u) >> 4);
    if (c < 256u) return 12 + (int)((c - 128u) >> 5);
    if (c < 512u) return 16 + (int)((c - 256u) >> 6);
    if (c < 1024u) return 20 + (int)((c - 512u) >> 7);
    const uint32_t k = 24u + ((c - 1024u) >> 8);
    return (int)(k < 31u ? k : 31u);
}
__device__ __forceinline__ int policy_class_of_order_class(int oc) { return oc < 8 ? 0 : oc < 12 ? 1 : oc < 14 ? 2 : oc < 16 ? 3 : oc < 18 ? 4 : oc < 20 ? 5 : oc < 22 ? 6 : 7; }
// split policy: log2(parts) of class c in bits [3c, 3c+3) of `policy`
__device__ __forceinline__ uint32_t class_lg_parts(uint32_t policy, int cls) { return (policy >> (3 * cls)) & 7u; }
// A block's schedule cost = the largest of its costs over the last YCGE_COST_FRAMES frames (ring of per-frame arrays, cost[f][b]):
// which blocks run long is a property of the image region (silhouettes seen at grazing angles), how long a given one runs in a
// given frame depends on that frame's random bounce directions - the 40 longest-running blocks of a frame are found among the
// top 200 of the previous frame's costs 58 % of the time, among the top 200 of max-over-four-frames 90 % (profiles/cost_persistence.py).
// (skip_mask: bit f set = ring slot f is being written by a trace that runs beside this schedule - frames in flight, tiled frames - and is left out)
// (n_frames: slots of the ring - YCGE_COST_FRAMES everywhere but on the K-deep ring of the tile-resident form, which keeps 8)
__device__ __forceinline__ uint32_t smoothed_cost(const uint32_t *cost, uint32_t n, uint32_t i, uint32_t skip_mask, uint32_t n_frames)
{
    uint32_t m = 0;
    if (n_frames == YCGE_COST_FRAMES) {
#pragma unroll
        for (int f = 0; f < YCGE_COST_FRAMES; f++) { const uint32_t v = ((skip_mask >> f) & 1u) ? 0u : cost[(size_t)f * n + i]; m = v > m ? v : m; }
    } else
        for (uint32_t f = 0; f < n_frames; f++) { const uint32_t v = ((skip_mask >> f) & 1u) ? 0u : cost[(size_t)f * n + i]; m = v > m ? v : m; }
    return m;
}
__global__ __launch_bounds__(1024) void k_cost_hist(const uint32_t *__restrict__ cost, uint32_t n, uint32_t skip_mask, uint32_t n_frames, uint32_t *__restrict__ ws)
{
    __shared__ uint32_t h[YCGE_ORDER_CLASSES];
    if (threadIdx.x < YCGE_ORDER_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int cls = order_class(smoothed_cost(cost, n, i, skip_mask, n_frames)); atomicAdd(&h[cls], 1u); }
    __syncthreads();
    if (threadIdx.x < YCGE_ORDER_CLASSES && h[threadIdx.x]) atomicAdd(&ws[32 + threadIdx.x], h[threadIdx.x]);
}
// Which blocks go in parts (a part = 64 >> lg pixels of the block, one wavefront each): either by policy class (`policy`, the YCGE_SPLIT
// knob and the 8-rank default) or - split_top > 0 - the split_top blocks at the HEAD of the schedule, in 4 parts of 16 pixels: a part's
// chain is the longest of 16 pixels' queries instead of 64, it reaches the cooperative walk's trigger (<= 4 live lanes) sooner, and a few
// dozen blocks are what the frame's end waits for (round 3: config 3 0.309 -> 0.275 ms with its ~100 blocks above 256 iterations
// split; by CLASS the same rule hands config 4 its 780 such blocks and loses).  Within an order class the first-come blocks are the split ones.
struct ClassLayout { uint32_t entries_before, blocks_before, n_split, lg; };
__device__ __forceinline__ ClassLayout class_layout(const uint32_t *ws, int cls, uint32_t policy, uint32_t split_top, bool split)
{
    ClassLayout L = {0u, 0u, 0u, 0u};
    for (int c = YCGE_ORDER_CLASSES - 1; c >= cls; c--) {
        const uint32_t cnt = ws[32 + c];
        const uint32_t lg = !split ? 0u : split_top ? (split_top >> 16) : class_lg_parts(policy, policy_class_of_order_class(c));
        const uint32_t top = split_top & 0xffffu;       // split_top = blocks | log2(parts) << 16
        const uint32_t ns = !split ? 0u : split_top ? (top > L.blocks_before ? (top - L.blocks_before < cnt ? top - L.blocks_before : cnt) : 0u) : (lg ? cnt : 0u);
        if (c == cls) { L.n_split = ns; L.lg = lg; return L; }
        L.entries_before += cnt + ns * ((1u << lg) - 1u);
        L.blocks_before += cnt;
    }
    return L;
}
// cost: what k_cost_hist read - the SAME values, or a block's class here and its class in the histogram differ and the entries land
// outside their class's range (holes of stale entries, writes past the list): where traces may still be writing costs while the schedule
// is built (traces in flight), both kernels read a snapshot of the ring (ycge_launch_order_blocks) and `clear` is the live ring.
__global__ __launch_bounds__(1024) void k_cost_scatter(const uint32_t *__restrict__ cost, uint32_t *__restrict__ clear, uint32_t n, uint32_t capacity, uint32_t policy, uint32_t split_top,
                                                       uint32_t fan_class, uint32_t fan_cap, uint32_t next_slot, uint32_t skip_mask, uint32_t n_frames, uint32_t *__restrict__ ws, uint32_t *__restrict__ order)
{
    __shared__ uint32_t h[YCGE_ORDER_CLASSES], rank0[YCGE_ORDER_CLASSES];
    __shared__ ClassLayout lay[YCGE_ORDER_CLASSES];
    __shared__ uint32_t s_split;
    if (threadIdx.x < YCGE_ORDER_CLASSES) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) {     // every workgroup derives the same decision from the finished histogram
        const ClassLayout all = class_layout(ws, -1, policy, split_top, true);         // (cls = -1: the sums over every class)
        s_split = all.entries_before <= capacity ? 1u : 0u;
        if (blockIdx.x == 0) {
            ws[16] = s_split ? all.entries_before : n;
            uint32_t n_fan = 0;                 // entries of the policy classes >= fan_class, at the head of the schedule
            if (fan_class > 0) {
                int lowest = YCGE_ORDER_CLASSES;
                while (lowest > 0 && policy_class_of_order_class(lowest - 1) >= (int)fan_class) lowest--;
                if (lowest < YCGE_ORDER_CLASSES) { const ClassLayout f = class_layout(ws, lowest - 1, policy, split_top, s_split != 0); n_fan = f.entries_before; }
            }
            ws[18] = n_fan < fan_cap ? n_fan : fan_cap;
        }
    }
    __syncthreads();
    const bool split = s_split != 0;
    if (threadIdx.x < YCGE_ORDER_CLASSES) lay[threadIdx.x] = class_layout(ws, (int)threadIdx.x, policy, split_top, split);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    uint32_t local = 0;
    if (i < n) {
        cls = order_class(smoothed_cost(cost, n, i, skip_mask, n_frames));
        clear[(size_t)next_slot * n + i] = 0;        // the ring slot the next frame's atomicMax goes to
        local = atomicAdd(&h[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < YCGE_ORDER_CLASSES) rank0[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&ws[64 + threadIdx.x], h[threadIdx.x]) : 0u;     // this workgroup's first rank in the class
    __syncthreads();
    if (cls >= 0) {
        const ClassLayout L = lay[cls];
        const uint32_t rank = rank0[cls] + local, parts = 1u << L.lg;
        const bool in_parts = rank < L.n_split;
        uint32_t *dst = order + L.entries_before + (in_parts ? rank * parts : L.n_split * parts + (rank - L.n_split));
        const uint32_t lgp = in_parts ? L.lg : 0u;
        for (uint32_t p = 0; p < (1u << lgp); p++) dst[p] = i | (p << 22) | (lgp << 28);
    }
}

// ---------------------------------------------------------------------------------- K_taa (taa_pixel: above, in front of trace_block)
__global__ __launch_bounds__(256) void k_taa(const TaaParams T, const float *__restrict__ current, const float *__restrict__ normal,
                                             const float *__restrict__ depth, const uint8_t *__restrict__ sky, float *__restrict__ hist,
                                             float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    // 32 x 8 pixels a workgroup; 32 x 2 (one wavefront) for the frames in flight: such a workgroup takes the place of ONE retiring
    // wavefront of the trace that runs beside it, a four-wavefront one waits until four places are free on one CU
    taa_pixel(T, blockIdx.x * 32 + (threadIdx.x & 31), blockIdx.y * (int)(blockDim.x >> 5) + (threadIdx.x >> 5), current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky);
}

// ---------------------------------------------------------------------------------- tile-resident TAA (multi-GPU, one process per GPU)
// TemporalBlendWithClamp on this rank's OWN tiles only: the same per-pixel code on the same operands as k_taa - the 3x3 neighbourhood of a
// tile's edge pixels reaches one pixel into tiles of other ranks, whose {hdr, sky} arrived as halo records (k_scatter_halo) - so the
// history, the guide copies and the reset rule never leave the rank that owns the tile.
// ONE wavefront per workgroup (an 8 x 8 sub-tile): this kernel runs beside the traces in flight, whose wavefronts hold 3 of a SIMD's register
// slots - a 256-thread workgroup needs room on all four SIMDs of ONE compute unit at once and waited for it (175-245 us for a 10 us kernel,
// profiles/r05: the resolve that the next launch on the same ring slots waits for); a single wavefront takes any slot that frees.
__global__ __launch_bounds__(64) void k_taa_tiles(const TaaParams T, const FrameParams P, const float *__restrict__ current, const float *__restrict__ normal,
                                                  const float *__restrict__ depth, const uint8_t *__restrict__ sky, float *__restrict__ hist,
                                                  float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky,
                                                  float *__restrict__ slab /* or null: the resolved history of the rank's tiles, k_pack_history's layout - one launch less in the resolve chain */)
{
    int px, py, lx, ly;
    const int k = (int)(blockIdx.x >> 2);
    if (!tile_pixel_wl(P, k, (int)(blockIdx.x & 3u), (int)threadIdx.x, px, py, lx, ly)) return;
    taa_pixel(T, px, py, current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky);
    if (slab) {         // (this thread's own stores, read back in program order)
        const size_t i = (size_t)px + (size_t)py * P.hiW;
        float *s = slab + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * 3;
        s[0] = hist[3 * i]; s[1] = hist[3 * i + 1]; s[2] = hist[3 * i + 2];
    }
}
// The resolve of a tile-resident frame as ONE launch (round 6): k_scatter_halo + k_taa_tiles + k_pack_history.  The halo
// records are not scattered into the frame's planes first - a tap that falls on a pixel another rank owns reads its record where the exchange
// left it (halo_index[pixel] = the record's place in the receive buffer; host: ensure_resident, from the same halo_layout the exchange follows).
// A record holds the four floats k_gather_halo made of the owner's pixel, sky as 1.0 / 0.0: the values k_scatter_halo would have stored.
struct HaloTap {
    const float *__restrict__ current; const uint8_t *__restrict__ sky; const float4 *__restrict__ records; const uint32_t *__restrict__ halo_index;
    int w, tiles_x, rank, world;
    __device__ __forceinline__ void operator()(int sx, int sy, float &r, float &g, float &b, uint8_t &s) const
    {
        const size_t j = (size_t)sx + (size_t)sy * w;
        const int owner = ((sy / YCGE_TILE_H) * tiles_x + sx / YCGE_TILE_W) % world;
        if (owner == rank) { s = sky[j]; r = current[3 * j]; g = current[3 * j + 1]; b = current[3 * j + 2]; }
        else { const float4 v = records[halo_index[j]]; r = v.x; g = v.y; b = v.z; s = v.w != 0.0f ? (uint8_t)1 : (uint8_t)0; }
    }
};
__global__ __launch_bounds__(64) void k_resolve_tiles(const TaaParams T, const FrameParams P, const float *__restrict__ current, const float *__restrict__ normal,
                                                      const float *__restrict__ depth, const uint8_t *__restrict__ sky, const float4 *__restrict__ records,
                                                      const uint32_t *__restrict__ halo_index, float *__restrict__ hist, float *__restrict__ prev_normal,
                                                      float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky, float *__restrict__ slab)
{
    int px, py, lx, ly;
    const int k = (int)(blockIdx.x >> 2);
    if (!tile_pixel_wl(P, k, (int)(blockIdx.x & 3u), (int)threadIdx.x, px, py, lx, ly)) return;
    const HaloTap tap = {current, sky, records, halo_index, P.hiW, P.tiles_x, P.rank, P.world_size};
    taa_pixel_t(T, px, py, tap, normal, depth, hist, prev_normal, prev_depth, prev_sky);
    if (slab) {
        const size_t i = (size_t)px + (size_t)py * P.hiW;
        float *s = slab + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * 3;
        s[0] = hist[3 * i]; s[1] = hist[3 * i + 1]; s[2] = hist[3 * i + 2];
    }
}
// halo records {hdr rgb, sky} of the listed pixels: out of this rank's frame buffers for the ranks that need them (gather), and the
// records received from the owners into this rank's frame buffers (scatter).  The lists are layout arithmetic (host: halo_layout; the
// same arithmetic in tiles.py): both sides enumerate the ring pixels of the receiver's tiles in one fixed order.
__global__ __launch_bounds__(256) void k_gather_halo(const float *__restrict__ hdr, const uint8_t *__restrict__ sky, const uint32_t *__restrict__ px, uint32_t n, float4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t p = px[i];
    out[i] = make_float4(hdr[3 * p], hdr[3 * p + 1], hdr[3 * p + 2], sky[p] ? 1.0f : 0.0f);
}
__global__ __launch_bounds__(256) void k_scatter_halo(const float4 *__restrict__ in, const uint32_t *__restrict__ px, uint32_t n, float *__restrict__ hdr, uint8_t *__restrict__ sky)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t p = px[i];
    const float4 v = in[i];
    hdr[3 * p] = v.x; hdr[3 * p + 1] = v.y; hdr[3 * p + 2] = v.z;
    sky[p] = v.w != 0.0f ? (uint8_t)1 : (uint8_t)0;
}
// the resolved history of this rank's tiles as a slab (3 floats per pixel, tile order as k_pack_slab) - what a consumer of the frame
// gathers: 12 bytes per pixel instead of the 32-44 of the tile slabs - and back into a full-frame history on whoever shows the frame
__global__ __launch_bounds__(256) void k_pack_history(const FrameParams P, const float *__restrict__ hist, float *__restrict__ slab)
{
    int px, py, lx, ly;
    const int k = blockIdx.x;
    if (!tile_pixel(P, k, px, py, lx, ly)) return;
    const size_t i = (size_t)px + (size_t)py * P.hiW;
    float *s = slab + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * 3;
    s[0] = hist[3 * i]; s[1] = hist[3 * i + 1]; s[2] = hist[3 * i + 2];
}
__global__ __launch_bounds__(256) void k_unpack_history(const float *__restrict__ all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size,
                                                       float *__restrict__ hist)
{
    const int tile_id = blockIdx.x;
    if (tile_id >= n_tiles) return;
    const int rank = tile_id % world_size, k = tile_id / world_size;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
    const int px = (tile_id % tiles_x) * YCGE_TILE_W + lx, py = (tile_id / tiles_x) * YCGE_TILE_H + ly;
    if (px >= hiW || py >= hiH) return;
    const float *s = all_slabs + (size_t)rank * slab_floats_per_rank + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * 3;
    const size_t i = (size_t)px + (size_t)py * hiW;
    hist[3 * i] = s[0]; hist[3 * i + 1] = s[1]; hist[3 * i + 2] = s[2];
}

// ---------------------------------------------------------------------------------- tile slabs (multi-GPU)
// pack this rank's tiles (tile_id % world == rank) from the full-frame buffers into its slab
// slab pixel = {hdr rgb, [albedo rgb,] normal xyz, depth, sky}: sf = 11 floats with the albedo plane, 8 without (lean)
__global__ __launch_bounds__(256) void k_pack_slab(const FrameParams P, const float *__restrict__ hdr, const float *__restrict__ albedo,
                                                   const float *__restrict__ normal, const float *__restrict__ depth,
                                                   const uint8_t *__restrict__ sky, float *__restrict__ slab, int sf)
{
    int px, py, lx, ly;
    const int k = blockIdx.x;
    if (!tile_pixel(P, k, px, py, lx, ly)) return;
    const size_t i = (size_t)px + (size_t)py * P.hiW;
    float *s = slab + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * (size_t)sf;
    s[0] = hdr[3 * i]; s[1] = hdr[3 * i + 1]; s[2] = hdr[3 * i + 2];
    if (sf == YCGE_SLAB_FLOATS) { s[3] = albedo[3 * i]; s[4] = albedo[3 * i + 1]; s[5] = albedo[3 * i + 2]; s += 3; }
    s[3] = normal[3 * i]; s[4] = normal[3 * i + 1]; s[5] = normal[3 * i + 2];
    s[6] = depth[i]; s[7] = sky[i] ? 1.0f : 0.0f;
}

// One process, several GPUs: a peer device copies ITS tiles of every listed plane from its own frame buffers straight into rank 0's
// (peer memory over xGMI; rows of 32 pixels are contiguous in both).  Replaces pack -> all-gather -> un-permute: no staging slabs.
__global__ __launch_bounds__(256) void k_push_tiles(const FrameParams P, const PushPlanes L)
{
    int px, py, lx, ly;
    const int k = blockIdx.x;
    if (!tile_pixel(P, k, px, py, lx, ly)) return;
    const size_t i = (size_t)px + (size_t)py * P.hiW;
    for (int a = 0; a < L.n; a++) {
        const int bpp = L.bytes_per_pixel[a];
        if (bpp == 1) L.dst[a][i] = L.src[a][i];
        else {
            const uint32_t *s = (const uint32_t *)L.src[a] + i * (size_t)(bpp >> 2);
            uint32_t *d = (uint32_t *)L.dst[a] + i * (size_t)(bpp >> 2);
            for (int w = 0; w < (bpp >> 2); w++) d[w] = s[w];
        }
    }
}

// all_slabs: world_size equal-sized slabs, rank-major, as an all-gather leaves them.
__global__ __launch_bounds__(256) void k_unpermute(const float *__restrict__ all_slabs, size_t slab_floats_per_rank, int hiW, int hiH,
                                                   int tiles_x, int n_tiles, int world_size, int sf, float *__restrict__ hdr,
                                                   float *__restrict__ albedo, float *__restrict__ normal, float *__restrict__ depth,
                                                   uint8_t *__restrict__ sky)
{
    const int tile_id = blockIdx.x;
    if (tile_id >= n_tiles) return;
    const int rank = tile_id % world_size, k = tile_id / world_size;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
    const int px = (tile_id % tiles_x) * YCGE_TILE_W + lx, py = (tile_id / tiles_x) * YCGE_TILE_H + ly;
    if (px >= hiW || py >= hiH) return;
    const float *s = all_slabs + (size_t)rank * slab_floats_per_rank + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * (size_t)sf;
    const size_t i = (size_t)px + (size_t)py * hiW;
    hdr[3 * i] = s[0]; hdr[3 * i + 1] = s[1]; hdr[3 * i + 2] = s[2];
    if (sf == YCGE_SLAB_FLOATS) { albedo[3 * i] = s[3]; albedo[3 * i + 1] = s[4]; albedo[3 * i + 2] = s[5]; s += 3; }
    normal[3 * i] = s[3]; normal[3 * i + 1] = s[4]; normal[3 * i + 2] = s[5];
    depth[i] = s[6];
    sky[i] = s[7] != 0.0f ? 1 : 0;
}

} // namespace ycge

// ---------------------------------------------------------------------------------- host-callable launchers
#include <type_traits>
#include <cstdio>
#include <cstdlib>
namespace {
// run-time flags -> compile-time kernel variants
template <class F> void sel3(bool a, bool b, bool c, F f)
{
    using T = std::true_type; using N = std::false_type;
    if (a) { if (b) { if (c) f(T{}, T{}, T{}); else f(T{}, T{}, N{}); } else { if (c) f(T{}, N{}, T{}); else f(T{}, N{}, N{}); } }
    else { if (b) { if (c) f(N{}, T{}, T{}); else f(N{}, T{}, N{}); } else { if (c) f(N{}, N{}, T{}); else f(N{}, N{}, N{}); } }
}
} // namespace

extern "C" {


size_t ycge_wf_sizes(int which)
{
    using namespace ycge;
    switch (which) { case 0: return sizeof(QEntry); case 1: return sizeof(HitRec); case 2: return sizeof(LEntry); }
    return 0;
}

// single-launch path
// (start / stop: events that take the KERNEL's own begin and end times - hipExtLaunchKernelGGL hangs it on the dispatch's completion signal, so nothing stands
// between this launch and the next one on the stream; an event recorded behind the launch is a packet of its own that the next launch waits
// behind: 8 us of every synchronous frame, round 6)
int ycge_launch_trace(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int count, int flat, int refill_steps,
                      hipStream_t stream, hipEvent_t start, hipEvent_t stop)
{
    using namespace ycge;
    if (P->n_owned_tiles <= 0) return 0;
    const dim3 grid((unsigned)P->n_owned_tiles * 4u * YCGE_SCHEDULE_SLACK), block(64);   // schedule capacity; idle entries exit at once
#if YCGE_EXPERIMENTS
    if (flat && refill_steps > 0) {
        if (count) hipLaunchKernelGGL((k_trace_refill<true>), grid, block, 0, stream, *S, *P, *O, refill_steps);
        else hipLaunchKernelGGL((k_trace_refill<false>), grid, block, 0, stream, *S, *P, *O, refill_steps);
        return (int)hipGetLastError();
    }
#endif
    static const unsigned lds_pad = getenv("YCGE_LDS_PAD") ? (unsigned)atoi(getenv("YCGE_LDS_PAD")) : 0u;   // experiment knob: fewer resident wavefronts
    if (!count && flat && S->tl_offset == 0u) {         // no mesh (or the cooperative walk switched off): the lean instance
        if (stop) hipExtLaunchKernelGGL(k_trace_nomesh, grid, block, lds_pad, stream, start, stop, 0, *S, *P, *O);
        else hipLaunchKernelGGL(k_trace_nomesh, grid, block, lds_pad, stream, *S, *P, *O);
        return (int)hipGetLastError();
    }
    sel3(count != 0, flat != 0, false, [&](auto C, auto F, auto) {
        if (stop) hipExtLaunchKernelGGL((k_trace<decltype(C)::value, decltype(F)::value>), grid, block, lds_pad, stream, start, stop, 0, *S, *P, *O);
        else hipLaunchKernelGGL((k_trace<decltype(C)::value, decltype(F)::value>), grid, block, lds_pad, stream, *S, *P, *O);
    });
    return (int)hipGetLastError();
}

// n frames' blocks in one launch (k_trace_batch); P / O: n records each (host memory: they travel as kernel arguments)
int ycge_launch_trace_batch(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int n, int count, int flat, hipStream_t stream)
{
    using namespace ycge;
    if (n <= 0 || n > YCGE_TRACE_BATCH_MAX || P[0].n_owned_tiles <= 0) return 0;
    TraceBatch B = {};
    for (int k = 0; k < n; k++) { B.P[k] = P[k]; B.O[k] = O[k]; }
    const dim3 grid((unsigned)P[0].n_owned_tiles * 4u * YCGE_SCHEDULE_SLACK * (unsigned)n), block(64);
    sel3(count != 0, flat != 0, false, [&](auto C, auto F, auto) {
        hipLaunchKernelGGL((k_trace_batch<decltype(C)::value, decltype(F)::value>), grid, block, 0, stream, *S, B, (uint32_t)n);
    });
    return (int)hipGetLastError();
}

// wavefront path.  bufs = {q0, q1, hit, lq, n_q, n_lq}: queues segmented per owned tile (256 entries each),
// n_q = (rounds + 1) x tiles counts, n_lq = tiles counts.  Every stage is one workgroup per tile.
// side / ev_fork / ev_join (optional): the light loop of round r - shadow rays from the vertices shade(r) found - and the trace of round
// r + 1 need nothing of each other (the one adds to current_hdr and reads the light queue, the other reads the ray queue and writes the
// hit records), so the light loop runs on `side` beside the trace and shade(r + 1) - which adds to current_hdr AFTER it, the reference's
// order, and refills the light queue - waits for both.
int ycge_launch_wavefront(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, void *const bufs[7], int rounds,
                          int has_grid, int flat, int count, int persistent_waves, hipStream_t stream, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join,
                          const ycge::TraceOut *O_side /* O with a traversal-stack spill area of its own: two kernels walk at the same time */)
{
    using namespace ycge;
    if (P->n_owned_tiles <= 0) return 0;
    WfBuffers B;
    B.q[0] = (QEntry *)bufs[0]; B.q[1] = (QEntry *)bufs[1]; B.hit = (HitRec *)bufs[2]; B.lq = (LEntry *)bufs[3];
    B.n_q = (uint32_t *)bufs[4]; B.n_lq = (uint32_t *)bufs[5];
    B.chunk_ctr = (uint32_t *)bufs[6];
    B.tiles = (uint32_t)P->n_owned_tiles;
    B.tile_order = P->tile_order;
    B.elide_dark = count ? 0u : 1u;
    const dim3 block(256), tiles((unsigned)P->n_owned_tiles);
    static int round_steps[4] = {0, 0, 0, 0};
    if (round_steps[0] == 0) {       // steps per round of k_wf_extend_p: YCGE_ROUND="tree,cell[,lanes waiting before a refill[,phase mode]]" overrides the tuned defaults
        round_steps[0] = YCGE_ROUND_TREE_STEPS; round_steps[1] = YCGE_ROUND_CELL_STEPS; round_steps[2] = YCGE_ROUND_REFILL_MIN; round_steps[3] = YCGE_ROUND_PHASE_MODE;
        if (const char *e = getenv("YCGE_ROUND")) { int a2 = 0, b2 = 0, c2 = YCGE_ROUND_REFILL_MIN, d2 = YCGE_ROUND_PHASE_MODE; if (sscanf(e, "%d,%d,%d,%d", &a2, &b2, &c2, &d2) >= 2 && a2 > 0 && b2 > 0 && c2 > 0 && d2 >= 0) { round_steps[0] = a2; round_steps[1] = b2; round_steps[2] = c2; round_steps[3] = d2; } }
    }
    sel3(count != 0, has_grid != 0, flat != 0, [&](auto C, auto G, auto F) {
        hipLaunchKernelGGL((k_wf_primary<decltype(C)::value, decltype(G)::value, decltype(F)::value>), tiles, block, 0, stream, *S, *P, *O, B);
    });
    bool forked = false;
    for (int r = 0; r < rounds; r++) {
        const bool persistent = !flat && persistent_waves > 0;
        if (r > 0 && persistent) {
            (void)hipMemsetAsync(B.chunk_ctr, 0, sizeof(uint32_t), stream);
            sel3(count != 0, has_grid != 0, false, [&](auto C, auto G, auto) {
                hipLaunchKernelGGL((k_wf_trace_p<decltype(C)::value, decltype(G)::value>), dim3((unsigned)persistent_waves), dim3(64), 0, stream,
                                   *S, *P, *O, B, r, B.chunk_ctr, round_steps[0], round_steps[1], round_steps[2], round_steps[3]);
            });
        } else if (r > 0)
            sel3(count != 0, has_grid != 0, flat != 0, [&](auto C, auto G, auto F) {
                hipLaunchKernelGGL((k_wf_extend<decltype(C)::value, decltype(G)::value, decltype(F)::value>), tiles, block, 0, stream, *S, *O, B, r);
            });
        if (forked) { (void)hipStreamWaitEvent(stream, ev_join, 0); forked = false; }
        sel3(r == 0, has_grid != 0, false, [&](auto R0, auto G, auto) {
            hipLaunchKernelGGL((k_wf_shade<decltype(R0)::value, decltype(G)::value>), tiles, block, 0, stream, *S, *P, *O, B, r);
        });
        const bool fork = side && ev_fork && ev_join && O_side && r + 1 < rounds;
        if (fork) { (void)hipEventRecord(ev_fork, stream); (void)hipStreamWaitEvent(side, ev_fork, 0); }
        sel3(count != 0, has_grid != 0, flat != 0, [&](auto C, auto G, auto F) {
            hipLaunchKernelGGL((k_wf_lights<decltype(C)::value, decltype(G)::value, decltype(F)::value>), tiles, block, 0, fork ? side : stream, *S, *P, fork ? *O_side : *O, B, r);
        });
        if (fork) { (void)hipEventRecord(ev_join, side); forked = true; }
    }
    if (forked) (void)hipStreamWaitEvent(stream, ev_join, 0);
    return (int)hipGetLastError();
}

// builds next frame's k_trace schedule from this frame's per-block step counts.  ws: 18 uint32 (see k_cost_hist)
int ycge_launch_order_blocks(uint32_t *cost, uint32_t n, uint32_t policy, uint32_t split_top, uint32_t fan_class, uint32_t fan_cap, uint32_t next_slot, uint32_t skip_mask,
                             uint32_t *ws, uint32_t *order, hipStream_t stream, int small_groups, uint32_t n_frames, uint32_t *snap)
{
    if (n == 0) return 0;
    if (n_frames == 0) n_frames = YCGE_COST_FRAMES;
    // snap (n x n_frames words of scratch, or null): the histogram and the scatter read a COPY of the cost ring taken first - wherever a
    // trace may still be writing costs while this schedule is built (frames in flight, tiled frames on two streams, the tile-resident
    // ring).  The two kernels must see the same class for every block (k_cost_scatter); a copy that is itself half-updated only costs
    // the schedule a little of its quality.
    const uint32_t *read = cost;
    if (snap) {
        const hipError_t e0 = hipMemcpyAsync(snap, cost, (size_t)n * n_frames * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e0 != hipSuccess) return (int)e0;
        read = snap;
    }
    // small_groups (frames in flight): a 1024-thread workgroup needs a quarter of a CU cleared before it starts and so waited for the
    // last third of the trace running beside it; four wavefronts find room far sooner (one-wavefront groups were no better)
    const unsigned threads = small_groups ? 256u : 1024u;
    hipError_t e = hipMemsetAsync(ws, 0, 96 * sizeof(uint32_t), stream);     // ws[16] / ws[18] (entries, n_fan) are rewritten by k_cost_scatter before anyone reads them
    if (e != hipSuccess) return (int)e;
    const dim3 grid((n + threads - 1u) / threads), block(threads);
    hipLaunchKernelGGL(ycge::k_cost_hist, grid, block, 0, stream, read, n, skip_mask, n_frames, ws);
    hipLaunchKernelGGL(ycge::k_cost_scatter, grid, block, 0, stream, read, cost, n, n * YCGE_SCHEDULE_SLACK, policy, split_top, fan_class, fan_cap, next_slot, skip_mask, n_frames, ws, order);
    return (int)hipGetLastError();
}

// the fanned blocks of the schedule (see k_trace_fan); grid = the cap the schedule was built with
int ycge_launch_trace_fan(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int count, int flat, uint32_t fan_cap,
                          hipStream_t stream)
{
    using namespace ycge;
    if (P->n_owned_tiles <= 0 || fan_cap == 0) return 0;
    sel3(count != 0, flat != 0, false, [&](auto C, auto F, auto) {
        hipLaunchKernelGGL((k_trace_fan<decltype(C)::value, decltype(F)::value>), dim3(fan_cap), dim3(192), 0, stream, *S, *P, *O);
    });
    return (int)hipGetLastError();
}

int ycge_launch_taa(const ycge::TaaParams *T, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                    float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, hipStream_t stream, int small_groups, hipEvent_t stop)
{
    // small_groups (frames in flight): one-wavefront workgroups take the place of a single retiring wavefront of the trace running beside them
    const unsigned rows = small_groups ? 2u : 8u;
    dim3 grid((unsigned)((T->w + 31) / 32), (unsigned)((T->h + (int)rows - 1) / (int)rows)), block(32u * rows);
    if (stop) hipExtLaunchKernelGGL(ycge::k_taa, grid, block, 0, stream, nullptr, stop, 0, *T, current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky);
    else hipLaunchKernelGGL(ycge::k_taa, grid, block, 0, stream, *T, current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky);
    return (int)hipGetLastError();
}

int ycge_launch_taa_tiles(const ycge::TaaParams *T, const ycge::FrameParams *P, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                          float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, float *slab, hipStream_t stream)
{
    if (P->n_owned_tiles <= 0) return 0;
    hipLaunchKernelGGL(ycge::k_taa_tiles, dim3((unsigned)P->n_owned_tiles * 4u), dim3(64), 0, stream, *T, *P, current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky, slab);
    return (int)hipGetLastError();
}
int ycge_launch_resolve_tiles(const ycge::TaaParams *T, const ycge::FrameParams *P, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                              const void *records, const uint32_t *halo_index, float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, float *slab, hipStream_t stream)
{
    if (P->n_owned_tiles <= 0) return 0;
    hipLaunchKernelGGL(ycge::k_resolve_tiles, dim3((unsigned)P->n_owned_tiles * 4u), dim3(64), 0, stream, *T, *P, current, normal, depth, sky, (const float4 *)records, halo_index,
                       hist, prev_normal, prev_depth, prev_sky, slab);
    return (int)hipGetLastError();
}
int ycge_launch_halo(int scatter, float *hdr, uint8_t *sky, const uint32_t *px, uint32_t n, void *records, hipStream_t stream)
{
    if (n == 0) return 0;
    const dim3 grid((n + 255u) / 256u), block(256);
    if (scatter) hipLaunchKernelGGL(ycge::k_scatter_halo, grid, block, 0, stream, (const float4 *)records, px, n, hdr, sky);
    else hipLaunchKernelGGL(ycge::k_gather_halo, grid, block, 0, stream, (const float *)hdr, (const uint8_t *)sky, px, n, (float4 *)records);
    return (int)hipGetLastError();
}
int ycge_launch_pack_history(const ycge::FrameParams *P, const float *hist, float *slab, hipStream_t stream)
{
    if (P->n_owned_tiles <= 0) return 0;
    hipLaunchKernelGGL(ycge::k_pack_history, dim3((unsigned)P->n_owned_tiles), dim3(256), 0, stream, *P, hist, slab);
    return (int)hipGetLastError();
}
int ycge_launch_unpack_history(const float *all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size, float *hist, hipStream_t stream)
{
    hipLaunchKernelGGL(ycge::k_unpack_history, dim3((unsigned)n_tiles), dim3(256), 0, stream, all_slabs, slab_floats_per_rank, hiW, hiH, tiles_x, n_tiles, world_size, hist);
    return (int)hipGetLastError();
}

int ycge_launch_pack_slab(const ycge::FrameParams *P, const float *hdr, const float *albedo, const float *normal, const float *depth,
                          const uint8_t *sky, float *slab, int slab_floats, hipStream_t stream)
{
    if (P->n_owned_tiles <= 0) return 0;
    dim3 grid((unsigned)P->n_owned_tiles), block(256);
    hipLaunchKernelGGL(ycge::k_pack_slab, grid, block, 0, stream, *P, hdr, albedo, normal, depth, sky, slab, slab_floats);
    return (int)hipGetLastError();
}

int ycge_launch_push_tiles(const ycge::FrameParams *P, const ycge::PushPlanes *planes, hipStream_t stream)
{
    if (P->n_owned_tiles <= 0 || planes->n <= 0) return 0;
    hipLaunchKernelGGL(ycge::k_push_tiles, dim3((unsigned)P->n_owned_tiles), dim3(256), 0, stream, *P, *planes);
    return (int)hipGetLastError();
}

int ycge_launch_unpermute(const float *all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size,
                          int slab_floats, float *hdr, float *albedo, float *normal, float *depth, uint8_t *sky, hipStream_t stream)
{
    dim3 grid((unsigned)n_tiles), block(256);
    hipLaunchKernelGGL(ycge::k_unpermute, grid, block, 0, stream, all_slabs, slab_floats_per_rank, hiW, hiH, tiles_x, n_tiles, world_size,
                       slab_floats, hdr, albedo, normal, depth, sky);
    return (int)hipGetLastError();
}

} // extern "C"
