// ycge_host.cpp - host side of the C-ABI in include/ycge.h: context, scene flattening and upload, frame orchestration (TryFlipAndBlit
// steps 1-9), frames in flight, the slab form of the tiled frame.  (The tile-resident multi-GPU form and the read-backs: ycge_resident.cpp.)
#include "ycge_ctx.h"
#include <dlfcn.h>

#ifndef YCGE_FAULT_INJECTION
#define YCGE_FAULT_INJECTION 0
#endif
#if YCGE_FAULT_INJECTION
#include <atomic>
#include <cstdlib>
#include <new>
static std::atomic<long long> g_fail_alloc_countdown{-1};      // < 0: off; n: the allocation n from now throws std::bad_alloc (ycge_debug_fail_allocation)
#endif
namespace ycge_host {

int alloc_frame_buffers(ycge_ctx *c)
{
    const size_t n = (size_t)c->hiW * c->hiH;
    HIP_TRY(c, c->current_hdr.alloc(3 * n)); HIP_TRY(c, c->g_albedo.alloc(3 * n)); HIP_TRY(c, c->g_normal.alloc(3 * n));
    HIP_TRY(c, c->g_depth.alloc(n)); HIP_TRY(c, c->sky.alloc(n));
    HIP_TRY(c, c->taa_hist.alloc(3 * n)); HIP_TRY(c, c->prev_normal.alloc(3 * n)); HIP_TRY(c, c->prev_depth.alloc(n));
    HIP_TRY(c, c->prev_sky.alloc(n));
    HIP_TRY(c, hipMemset(c->taa_hist.p, 0, 3 * n * sizeof(float)));
    if (c->cfg.capture_debug) {
        HIP_TRY(c, c->dbg_rays.alloc(6 * n)); HIP_TRY(c, c->dbg_prim.alloc(n)); HIP_TRY(c, c->dbg_sub.alloc(n));
        HIP_TRY(c, c->dbg_hit_t.alloc(n)); HIP_TRY(c, c->dbg_rng.alloc(n));
    }
    HIP_TRY(c, c->counters.alloc(YCGE_COUNTER_WORDS));            // [0..4] SURVEY 8(d) counters of the counting instances (zeroed per frame), [8 + 8 i] lane steps of the timed instances (cumulative, spread over cache lines)
    HIP_TRY(c, hipMemset(c->counters.p, 0, YCGE_COUNTER_WORDS * sizeof(unsigned long long)));
    return YCGE_OK;
}

// queue segments, hit records and stack-spill columns: one 256-entry segment per owned tile
int alloc_tile_buffers(ycge_ctx *c)
{
    if (c->fan_stream) { HIP_TRY(c, hipStreamSynchronize(c->fan_stream)); c->order_pending = false; }      // schedule kernels of the old size
    const size_t lanes = (size_t)(c->n_owned > 0 ? c->n_owned : 1) * 256;
    const size_t stack_lanes = lanes * YCGE_SCHEDULE_SLACK + (size_t)c->fan_cap * 192;      // k_trace's grid includes the schedule's slack entries; k_trace_fan's columns follow
    HIP_TRY(c, c->wf_q0.alloc(lanes * ycge_wf_sizes(0))); HIP_TRY(c, c->wf_q1.alloc(lanes * ycge_wf_sizes(0)));
    HIP_TRY(c, c->wf_hit.alloc(lanes * ycge_wf_sizes(1))); HIP_TRY(c, c->wf_lq.alloc(lanes * ycge_wf_sizes(2)));
    HIP_TRY(c, c->wf_seg.alloc(4));
    HIP_TRY(c, c->wf_counts.alloc((size_t)(c->n_owned > 0 ? c->n_owned : 1) * 8));
    HIP_TRY(c, c->stack_spill.alloc((size_t)(c->spill_levels > 0 ? c->spill_levels : 1) * stack_lanes));
    c->stack_spill2.release(); c->stack_spill_side.release(); c->stack_spill_side2.release();
    c->wf2_q0.release(); c->wf2_q1.release(); c->wf2_hit.release(); c->wf2_lq.release(); c->wf2_seg.release(); c->wf2_counts.release();
    c->path_stack.release();
   
    {
        const size_t nb = (size_t)(c->n_owned > 0 ? c->n_owned : 1) * 4;
        HIP_TRY(c, c->block_cost.alloc(nb * YCGE_COST_FRAMES)); HIP_TRY(c, c->block_order.alloc(nb * YCGE_SCHEDULE_SLACK)); HIP_TRY(c, c->order_ws.alloc(96));
        HIP_TRY(c, c->cost_snap.alloc(nb * (ycge_ctx::kResCostFrames > YCGE_COST_FRAMES ? ycge_ctx::kResCostFrames : YCGE_COST_FRAMES)));
        HIP_TRY(c, hipMemset(c->order_ws.p, 0, 96 * sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->block_cost.p, 0, nb * YCGE_COST_FRAMES * sizeof(uint32_t)));
        c->block_order_valid = false;
        for (int k = 0; k < 3; k++) { c->flight_order[k].release(); c->flight_ws[k].release(); c->flight_order_frame[k] = -1; }      // (allocated by the first frame in flight)
    }
    {   // XCD-aware block -> tile table: bucket the owned tiles by image strip (4 tiles = 128 px wide, strip s -> XCD s % 8),
        // then deal the buckets out round-robin so that block b (which lands on XCD b % 8) draws from bucket b % 8
        const int n = c->n_owned, world = c->cfg.world_size, rank = c->cfg.rank;
        std::vector<std::vector<uint32_t>> bucket(8);
        for (int k = 0; k < n; k++) {
            const int tile_id = rank + k * world;
            bucket[((tile_id % c->tiles_x) / 4) % 8].push_back((uint32_t)k);
        }
        std::vector<uint32_t> order;
        order.reserve(n > 0 ? n : 1);
        size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; (int)order.size() < n; b++) {
            int x = b % 8;
            if (pos[x] >= bucket[x].size()) {          // bucket exhausted: borrow from the fullest one
                size_t best = 0; int bi = -1;
                for (int y = 0; y < 8; y++) { const size_t left = bucket[y].size() - pos[y]; if (left > best) { best = left; bi = y; } }
                x = bi;
            }
            order.push_back(bucket[x][pos[x]++]);
        }
        if (order.empty()) order.push_back(0);
        HIP_TRY(c, c->tile_order.upload(order));
    }
    return YCGE_OK;
}

// ---- tile-resident form: halo lists.  Rank r needs {hdr, sky} of the one-pixel ring around each of its tiles (TAA's 3x3 window,
// TemporalBlendWithClamp clampRadius = 1, RaytraceRenderer.cs:218) from the ranks that own those pixels.  Both sides enumerate the ring
// pixels of the RECEIVER's tiles - tiles ascending; per tile the row above (x0 - 1 .. x0 + 32), the row below, the column left (y0 .. y0 + 7),
// the column right; pixels outside the image do not exist (TAA clamps its taps to the image) - and keep those the SENDER owns: the k-th such
// pixel is the k-th record of the (sender -> receiver) segment.  A pixel that borders two tiles of the receiver appears twice.  The same
// arithmetic in numpy: yetanotherconsolegameengine_amd/tiles.py (halo_lists); the CPU tests hold the two to each other.
static void halo_ring_of(int tile, int tiles_x, int hiW, int hiH, std::vector<uint32_t> &px)
{
    const int x0 = (tile % tiles_x) * YCGE_TILE_W, y0 = (tile / tiles_x) * YCGE_TILE_H;
    auto add = [&](int x, int y) { if (x >= 0 && y >= 0 && x < hiW && y < hiH) px.push_back((uint32_t)x + (uint32_t)y * (uint32_t)hiW); };
    for (int x = x0 - 1; x <= x0 + YCGE_TILE_W; x++) add(x, y0 - 1);
    for (int x = x0 - 1; x <= x0 + YCGE_TILE_W; x++) add(x, y0 + YCGE_TILE_H);
    for (int y = y0; y < y0 + YCGE_TILE_H; y++) add(x0 - 1, y);
    for (int y = y0; y < y0 + YCGE_TILE_H; y++) add(x0 + YCGE_TILE_W, y);
}
// send_px: pixels this rank gathers, segment by destination rank ascending; recv_px: pixels the received records scatter to, by source rank ascending
void halo_layout(int hiW, int hiH, int rank, int world, std::vector<int64_t> &send_counts, std::vector<int64_t> &recv_counts, std::vector<uint32_t> &send_px, std::vector<uint32_t> &recv_px)
{
    const int tiles_x = (hiW + YCGE_TILE_W - 1) / YCGE_TILE_W, tiles_y = (hiH + YCGE_TILE_H - 1) / YCGE_TILE_H, n_tiles = tiles_x * tiles_y;
    auto owner = [&](uint32_t p) { const int x = (int)(p % (uint32_t)hiW), y = (int)(p / (uint32_t)hiW); return ((y / YCGE_TILE_H) * tiles_x + x / YCGE_TILE_W) % world; };
    send_counts.assign((size_t)world, 0); recv_counts.assign((size_t)world, 0);
    send_px.clear(); recv_px.clear();
    std::vector<uint32_t> ring;
    std::vector<std::vector<uint32_t>> from((size_t)world);
    for (int t = rank; t < n_tiles; t += world) {          // what I receive: the ring of MY tiles, by owner
        ring.clear(); halo_ring_of(t, tiles_x, hiW, hiH, ring);
        for (uint32_t p : ring) { const int q = owner(p); if (q != rank) from[(size_t)q].push_back(p); }
    }
    for (int q = 0; q < world; q++) { recv_counts[(size_t)q] = (int64_t)from[(size_t)q].size(); recv_px.insert(recv_px.end(), from[(size_t)q].begin(), from[(size_t)q].end()); }
    for (int r = 0; r < world; r++) {                      // what I send: the ring of rank r's tiles, the pixels I own
        if (r == rank) continue;
        for (int t = r; t < n_tiles; t += world) {
            ring.clear(); halo_ring_of(t, tiles_x, hiW, hiH, ring);
            for (uint32_t p : ring) if (owner(p) == rank) { send_px.push_back(p); send_counts[(size_t)r]++; }
        }
    }
}
void release_resident(ycge_ctx *c)
{
    for (auto *rs : c->rsets) { if (rs->traced) (void)hipEventDestroy(rs->traced); if (rs->resolved) (void)hipEventDestroy(rs->resolved); delete rs; }
    c->rsets.clear();
    for (auto *b : c->res_order) delete b;
    for (auto *b : c->res_ws) delete b;
    c->res_order.clear(); c->res_ws.clear();
    for (hipEvent_t ev : c->res_order_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->res_order_read_ev) if (ev) (void)hipEventDestroy(ev);
    c->res_order_ev.clear(); c->res_order_read_ev.clear(); c->res_order_frame.clear();
    if (c->res_last_traced) { (void)hipEventDestroy(c->res_last_traced); c->res_last_traced = nullptr; }
    c->res_last_traced_used = false;
    c->res_cost.release(); c->d_halo_send_px.release(); c->d_halo_recv_px.release(); c->d_halo_index.release();
    c->halo_send_counts.clear(); c->halo_recv_counts.clear(); c->halo_ready = false;
}

// floats per slab pixel: hdr, [albedo,] normal, depth, sky
static size_t slab_floats(const ycge_ctx *c) { return c->cfg.slab_albedo ? (size_t)YCGE_SLAB_FLOATS : (size_t)YCGE_SLAB_FLOATS - 3; }

int set_geometry(ycge_ctx *c, int fbw, int fbh, int ss)
{
    if (fbw <= 0 || fbh <= 0) return c->fail(YCGE_ERR_INVALID_ARG, "framebuffer size must be positive");
    c->fbW = fbw; c->fbH = fbh; c->ss = ss < 1 ? 1 : ss;        // Math.Max(1, superSample), RaytraceRenderer.cs:81
    c->hiW = c->fbW * c->ss; c->hiH = c->fbH * 2 * c->ss;      // :86-87
    c->tiles_x = (c->hiW + YCGE_TILE_W - 1) / YCGE_TILE_W;
    c->tiles_y = (c->hiH + YCGE_TILE_H - 1) / YCGE_TILE_H;
    c->n_tiles = c->tiles_x * c->tiles_y;
    const int world = c->cfg.world_size, rank = c->cfg.rank;
    c->n_owned = rank < c->n_tiles ? (c->n_tiles - rank + world - 1) / world : 0;
    c->tiles_per_rank_padded = (c->n_tiles + world - 1) / world;
    c->taa_valid = false;                                       // Resize: taaHistoryValid = false (:137), taa.Resize (TemporalAA.cs:34-46)
    c->last_cam[0] = c->last_cam[1] = c->last_cam[2] = NAN; c->last_yaw = c->last_pitch = NAN;
    c->den_a.release(); c->den_b.release(); c->unit_n.release(); c->exp_terms.release(); c->exp_scratch.release(); c->d_sdr.release(); c->d_sdr2.release(); c->atrous_statw.release();     // spatialA / spatialB, :129-130
    c->denoised = nullptr;
    c->alt_post.release();
    c->wave_prof.release();                                     // sized for the tile grid
    c->pending.clear();
    c->t_hdr.release(); c->t_albedo.release(); c->t_normal.release(); c->t_depth.release(); c->t_sky.release();
    c->alt_hdr.release(); c->alt_albedo.release(); c->alt_normal.release(); c->alt_depth.release(); c->alt_sky.release();      // (callers have quiesced the device)
    c->alt2_hdr.release(); c->alt2_albedo.release(); c->alt2_normal.release(); c->alt2_depth.release(); c->alt2_sky.release();
    c->set_read[0] = c->set_read[1] = c->set_read[2] = false; c->out_set = 0; c->async_outstanding = false;
    c->tile_trace_used[0] = c->tile_trace_used[1] = false;
    release_resident(c);                                        // the ring of the tile-resident form and its halo lists are per size
    for (auto *sc : c->schedules) delete sc;      // level schedules are per size: rebuilt on demand (the destructor frees the device lists)
    c->schedules.clear();
    int rc = alloc_frame_buffers(c);
    if (rc != YCGE_OK) return rc;
    rc = alloc_tile_buffers(c);
    if (rc != YCGE_OK) return rc;
    const bool rccl_frame = c->cfg.multi_device_exchange == YCGE_EXCHANGE_RCCL;      // (ycge_create leaves the field set only on the contexts of a one-process multi-device frame)
    if (world > 1 || rccl_frame) HIP_TRY(c, c->own_slab.alloc((size_t)c->tiles_per_rank_padded * 256 * slab_floats(c)));
    if (rccl_frame) HIP_TRY(c, c->all_slabs.alloc((size_t)world * c->tiles_per_rank_padded * 256 * slab_floats(c))); else c->all_slabs.release();
    return YCGE_OK;
}

bool should_reset_history(const ycge_ctx *c, const float pos[3], float yaw, float pitch)   // TemporalAA.cs:58-67
{
    float dx = pos[0] - c->last_cam[0], dy = pos[1] - c->last_cam[1], dz = pos[2] - c->last_cam[2];
    float trans = is_nan(dx) ? 0.0f : cs_sqrt(dx * dx + dy * dy + dz * dz);
    float dyaw = is_nan(c->last_yaw) ? 0.0f : cs_abs(yaw - c->last_yaw);
    float dpitch = is_nan(c->last_pitch) ? 0.0f : cs_abs(pitch - c->last_pitch);
    return trans > c->cfg.motion_trans_reset || dyaw > c->cfg.motion_rot_reset || dpitch > c->cfg.motion_rot_reset;
}

struct H3 { float x, y, z; };
H3 h_norm(H3 a)
{
    float l = a.x * a.x + a.y * a.y + a.z * a.z;
    if (l <= 0.0f) return a;
    float inv = 1.0f / cs_sqrt(l);
    return H3{a.x * inv, a.y * inv, a.z * inv};
}
H3 h_cross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// per-frame constants of MakeJitteredRay (RaytraceRenderer.cs:413-434): host-side sin/cos/tan, as in the C#
void fill_frame_params(ycge_ctx *c, FrameParams &P, int64_t frame, const float pos[3], float yaw, float pitch, float fov_deg)
{
    std::memset(&P, 0, sizeof P);
    P.hiW = c->hiW; P.hiH = c->hiH;
    P.frame = frame;
    P.frame_idx = (int)(frame & 0x7fffffff);
    auto fracf = [](float v) { return v - cs_floor(v); };
    P.rot_x = fracf((float)(P.frame_idx + 1) * 0.61803398875f);
    P.rot_y = fracf((float)(P.frame_idx + 1) * 0.38196601125f);
    const float aspect = (float)c->hiW / (float)c->hiH;
    const float pi = 3.14159265358979323846f;
    float cp = cosf(pitch);
    H3 f = H3{sinf(yaw) * cp, sinf(pitch), -cosf(yaw) * cp};
    float fov_rad = fov_deg * (pi / 180.0f);
    P.half_h = tanf(0.5f * fov_rad);
    P.half_w = P.half_h * aspect;
    H3 fwd = h_norm(f);
    H3 right = h_norm(h_cross(fwd, H3{0.0f, 1.0f, 0.0f}));
    H3 up = h_norm(h_cross(right, fwd));
    P.cam_pos[0] = pos[0]; P.cam_pos[1] = pos[1]; P.cam_pos[2] = pos[2];
    P.fwd[0] = fwd.x; P.fwd[1] = fwd.y; P.fwd[2] = fwd.z;
    P.right[0] = right.x; P.right[1] = right.y; P.right[2] = right.z;
    P.up[0] = up.x; P.up[1] = up.y; P.up[2] = up.z;
    P.seed_salt = c->cfg.seed_salt;
    P.eps = c->cfg.eps;
    P.mirror_threshold = c->cfg.mirror_threshold;
    P.sigma_rad = c->cfg.diffuse_sigma_deg * (pi / 180.0f);
    {   // OrenNayarBRDF's sigma-only terms, same fp32 expressions as RaytraceRenderer.cs:823-825
        const float sigma2 = P.sigma_rad * P.sigma_rad;
        P.on_a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
        P.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
    }
    P.max_mirror_bounces = c->cfg.max_mirror_bounces;
    P.max_refractions = c->cfg.max_refractions;
    P.diffuse_bounces = c->cfg.diffuse_bounces;
    P.tiles_x = c->tiles_x; P.tiles_y = c->tiles_y;
    P.rank = c->cfg.rank; P.world_size = c->cfg.world_size;
    P.n_owned_tiles = c->n_owned;
    // measured on config 4: strip-per-XCD ordering is 2.1x SLOWER than plain round-robin (1.79 vs 0.84 ms): the
    // heavy tiles cluster in a few strips and the frame is bounded by its heaviest tiles, so spreading them over
    // all 8 XCDs beats L2 affinity.  Kept as an opt-in experiment knob only.
    P.tile_order = c->knobs.xcd_strips ? c->tile_order.p : nullptr;
}

// Hittable.TryGetBounds of each primitive class (see the citations in include/ycge.h)
bool prim_bounds(const ycge_prim &q, const std::vector<MeshHost> &meshes, const std::vector<std::array<float, 6>> &grid_bounds, float b[6], float cen[3])
{
    const float *p = q.p;
    const float eps = 1e-4f;
    bool from_box = true;
    switch (q.type) {
    case YCGE_PRIM_SPHERE:
        b[0] = p[0] - p[3]; b[1] = p[1] - p[3]; b[2] = p[2] - p[3]; b[3] = p[0] + p[3]; b[4] = p[1] + p[3]; b[5] = p[2] + p[3]; break;
    case YCGE_PRIM_PLANE:
        b[0] = b[1] = b[2] = -1e6f; b[3] = b[4] = b[5] = 1e6f; cen[0] = cen[1] = cen[2] = 0.0f; from_box = false; break;
    case YCGE_PRIM_DISK:
        b[0] = p[0] - p[6]; b[1] = p[1] - p[6]; b[2] = p[2] - p[6]; b[3] = p[0] + p[6]; b[4] = p[1] + p[6]; b[5] = p[2] + p[6]; break;
    case YCGE_PRIM_XYRECT: b[0] = p[0]; b[1] = p[2]; b[2] = p[4] - eps; b[3] = p[1]; b[4] = p[3]; b[5] = p[4] + eps; break;
    case YCGE_PRIM_XZRECT: b[0] = p[0]; b[1] = p[4] - eps; b[2] = p[2]; b[3] = p[1]; b[4] = p[4] + eps; b[5] = p[3]; break;
    case YCGE_PRIM_YZRECT: b[0] = p[4] - eps; b[1] = p[0]; b[2] = p[2]; b[3] = p[4] + eps; b[4] = p[1]; b[5] = p[3]; break;
    case YCGE_PRIM_BOX: for (int k = 0; k < 6; k++) b[k] = p[k]; break;
    case YCGE_PRIM_CYLINDER_Y:
        b[0] = p[0] - p[3]; b[1] = cs_min(p[4], p[5]); b[2] = p[2] - p[3]; b[3] = p[0] + p[3]; b[4] = cs_max(p[4], p[5]); b[5] = p[2] + p[3]; break;
    case YCGE_PRIM_TRIANGLE:
        for (int a = 0; a < 3; a++) {
            b[a] = cs_min(p[a], cs_min(p[3 + a], p[6 + a])) - eps;
            b[3 + a] = cs_max(p[a], cs_max(p[3 + a], p[6 + a])) + eps;
        }
        break;
    case YCGE_PRIM_MESH: {
        const BuiltTree &t = meshes[q.ref].tree;
        if (t.root < 0) return false;
        const RefNode &r = t.nodes[t.root];
        for (int a = 0; a < 3; a++) { b[a] = r.mn[a]; b[3 + a] = r.mx[a]; }
        break;
    }
    case YCGE_PRIM_VOLUME_GRID: {
        const std::array<float, 6> &gb = grid_bounds[q.ref];
        if (!(gb[3] >= gb[0])) return false;                 // empty grid (marked at upload)
        for (int k = 0; k < 6; k++) b[k] = gb[k];
        break;
    }
    default: return false;
    }
    if (from_box) for (int a = 0; a < 3; a++) cen[a] = 0.5f * (b[a] + b[3 + a]);
    return true;
}

int morton3(int x, int y, int z)
{
    return ((x & 1) << 0) | ((y & 1) << 1) | ((z & 1) << 2) | ((x & 2) << 2) | ((y & 2) << 3) | ((z & 2) << 4) | ((x & 4) << 4) | ((y & 4) << 5) | ((z & 4) << 6);
}

// librccl.so, dlopen'ed on first use (the library does not link it: a host without RCCL loses nothing but this option).  An instance the
// process already holds - bench.py's torch.distributed brings its own - is preferred over loading a second one.
struct RcclApi {
    void *h = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false, tried = false;
};
static RcclApi g_rccl;
static const RcclApi &load_rccl()
{
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (g_rccl.tried) return g_rccl;
    g_rccl.tried = true;
    const char *forced = getenv("YCGE_RCCL_LIB");             // (tests: a name that does not exist = "RCCL absent")
    const char *names[] = {forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !g_rccl.h; pass++)
        for (const char *n : names) {
            g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (g_rccl.h || forced) break;
        }
    if (!g_rccl.h) return g_rccl;
    auto sym = [&](const char *n) { return dlsym(g_rccl.h, n); };
    g_rccl.CommInitAll = (int (*)(void **, int, const int *))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
    g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
    g_rccl.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd;
    return g_rccl;
}

int abi_catch(const ycge_ctx *cc) noexcept
{
    ycge_ctx *c = const_cast<ycge_ctx *>(cc);
    int code = YCGE_ERR_INTERNAL;
    char text[320];
    std::snprintf(text, sizeof text, "internal error: unknown C++ exception stopped at the C-ABI");
    try { throw; }
    catch (const std::bad_alloc &) { code = YCGE_ERR_OUT_OF_MEMORY; std::snprintf(text, sizeof text, "out of host memory (std::bad_alloc stopped at the C-ABI)"); }
    catch (const std::exception &e) { std::snprintf(text, sizeof text, "internal error: %s (C++ exception stopped at the C-ABI)", e.what()); }
    catch (...) { }
    try { if (c) c->err = text; else g_create_error = text; } catch (...) { }      // (the message itself may not fit any more: the code still says what happened)
    return code;
}

} // namespace ycge_host

// =========================================================================== C-ABI
extern "C" {

int ycge_config_default(ycge_config *cfg)
try {
    if (!cfg) return YCGE_ERR_INVALID_ARG;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->abi_version = YCGE_ABI_VERSION;
    cfg->slab_albedo = 1; cfg->n_devices = 0;
    cfg->atrous_inplace_exact = 1;
    cfg->tile_ring = 0;
    cfg->multi_device_exchange = YCGE_EXCHANGE_PEER_PUSH;
    cfg->fb_width = 80; cfg->fb_height = 45; cfg->super_sample = 1;
    cfg->fov_deg = 45.0f;
    cfg->device = 0; cfg->rank = 0; cfg->world_size = 1;
    cfg->diffuse_bounces = 1; cfg->max_mirror_bounces = 2; cfg->max_refractions = 2;
    cfg->mirror_threshold = 0.9f; cfg->eps = 1e-4f;
    cfg->seed_salt = 0x9E3779B97F4A7C15ULL;
    cfg->taa_alpha = 0.01f; cfg->motion_trans_reset = 0.0025f; cfg->motion_rot_reset = 0.0025f;
    cfg->diffuse_sigma_deg = 25.0f;
    cfg->taa_clamp_radius = 1; cfg->taa_luminance_pad = 0.10f;
    cfg->atrous_iterations = 3; cfg->atrous_c_phi = 3.0f; cfg->atrous_n_phi = 0.35f; cfg->atrous_z_phi = 2.0f; cfg->atrous_a_phi = 0.20f;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

static void peer_worker_main(ycge_ctx *root, ycge_ctx *peer);
static int create_one(const ycge_config *cfg, ycge_ctx *parent, ycge_ctx **out)
{
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = "no HIP device: the ray-trace path has no CPU fallback";
        return YCGE_ERR_NO_DEVICE_CODE;
    }
    if (cfg->device < 0 || cfg->device >= n_dev) { g_create_error = "device ordinal out of range"; return YCGE_ERR_INVALID_ARG; }
    struct Owner { ycge_ctx *p; ~Owner() { if (p) ycge_destroy(p); } } owner{new ycge_ctx()};      // (an exception on the way out of this function must not leak the context and its streams)
    ycge_ctx *c = owner.p;
    c->err.reserve(320);
    c->cfg = *cfg;
    c->device = cfg->device;
    c->fov_deg = cfg->fov_deg;
    c->parent = parent;
    c->knobs.read();                // every YCGE_* knob is read here, once
    auto bail = [&](int code) { g_create_error = c->err; owner.p = nullptr; ycge_destroy(c); return code; };
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(YCGE_ERR_DEVICE); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { c->err = "hipGetDeviceProperties failed"; return bail(YCGE_ERR_DEVICE); }
    std::snprintf(c->device_name, sizeof c->device_name, "%s (%s)", prop.name, prop.gcnArchName);
    c->compute_units = prop.multiProcessorCount;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        c->err = std::string("device is ") + prop.gcnArchName + "; this library carries gfx950 code only";
        return bail(YCGE_ERR_NO_DEVICE_CODE);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(YCGE_ERR_DEVICE); }
    for (auto &ev : c->ev)
        if (hipEventCreate(&ev) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(YCGE_ERR_DEVICE); }
    {
        if (hipStreamCreateWithFlags(&c->fan_stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(YCGE_ERR_DEVICE); }
        if (c->cfg.world_size == 1 && c->cfg.n_devices <= 1) {
            // The second stream of the frames in flight (ycge_render_frame_async), created HERE, next to the other two: the runtime deals
            // streams onto a few hardware queues in order of creation, and a stream created later - after another context of the process
            // has come and gone - landed on the queue of this context's own stream: TAA and trace in ONE queue, 0.65 instead of 0.53 ms
            // a frame on config 4.  At the LOWEST priority: its kernels - TAA of the frame just traced, the schedule of the frame after
            // next - yield to the running trace's workgroups (measured neutral: 0.548 against 0.551 ms at normal priority)
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                hipStreamCreateWithPriority(&c->taa_stream, hipStreamNonBlocking, c->knobs.flight_priority > 0 ? hi : c->knobs.flight_priority < 0 ? lo : 0) != hipSuccess ||
                hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->flight_fork_ev, hipEventDisableTiming) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(YCGE_ERR_DEVICE); }
        }
        for (hipEvent_t *ev : {&c->fan_ev[0], &c->fan_ev[1], &c->traced_ev, &c->order_ev, &c->pushed_ev})
            if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(YCGE_ERR_DEVICE); }
        // Query fan-out (k_trace_fan).  On a rank's share of a tiled frame wavefront slots are plentiful and the rank's time is the
        // chain of its heaviest blocks: fan the classes >= 384 iterations (>= 256 from 4 ranks up), up to 2048 blocks (per rank on
        // config 4: 0.565 -> 0.418 ms at 8 ranks, 0.562 -> 0.433 at 4, 0.595 -> 0.493 at 2).  On a whole frame slots are what the
        // bulk is short of: only the 200 blocks at the head of the schedule (cost = max over four frames), with both kernels at 4
        // wavefronts per SIMD (0.590 -> 0.569 ms; 400+ blocks or 3 wavefronts per SIMD lose what the shorter chains gain).
        // Round 3: with the cooperative walk (ycge_coop.hip.h) a block's tail is short enough that on a WHOLE frame the helper wavefronts
        // cost the bulk more than the shorter chains gain (config 4: 0.514 ms without fan-out, 0.547 with the 200 blocks): off there.
        // The same holds on a rank's tiles once the heavy classes are split (trace_frame's default policy): per-rank trace on config 4
        // at 8 ranks 0.359 ms with fan-out + the round-2 split, 0.289 ms with the round-3 split alone; at 4 ranks 0.381 -> 0.351; at 2
        // ranks 0.485 -> 0.441 (profiles/r03/f_rank_emulation.txt).  k_trace_fan stays behind YCGE_FAN, parity-tested.
        const uint32_t fan_default = 0u;
        c->fan_class = c->knobs.fan_class >= 0 ? (uint32_t)c->knobs.fan_class : fan_default;
        if (hipHostMalloc((void **)&c->h_n_fan, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc failed"; return bail(YCGE_ERR_DEVICE); }
        *c->h_n_fan = 0;
        c->fan_cap = c->fan_class ? (c->knobs.fan_cap >= 0 ? (uint32_t)c->knobs.fan_cap : c->cfg.world_size >= 2 ? YCGE_FAN_CAP_DEFAULT : 200u) : 0u;
    }
    int rc = set_geometry(c, cfg->fb_width, cfg->fb_height, cfg->super_sample);
    if (rc != YCGE_OK) return bail(rc);
    owner.p = nullptr;
    *out = c;
    return YCGE_OK;
}

int ycge_create(const ycge_config *cfg, ycge_ctx **out)
try {
    if (!cfg || !out) { g_create_error = "null argument"; return YCGE_ERR_INVALID_ARG; }
    *out = nullptr;
    if (cfg->abi_version != YCGE_ABI_VERSION) { g_create_error = "abi_version mismatch"; return YCGE_ERR_INVALID_ARG; }
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) { g_create_error = "bad rank/world_size"; return YCGE_ERR_INVALID_ARG; }
    // the kernels implement the reference's compile-time constants (RaytraceRenderer.cs:31-36): in particular the path stack is
    // sized for MaxMirrorBounces = 2 (at most 3 live items), so any other value is refused here instead of dropping items
    if (cfg->diffuse_bounces != 1 || cfg->max_mirror_bounces != 2 || cfg->max_refractions != 2 || cfg->taa_clamp_radius < 0) {
        g_create_error = "DiffuseBounces/MaxMirrorBounces/MaxRefractions are compile-time constants in the reference (1/2/2)";
        return YCGE_ERR_UNSUPPORTED;
    }
    if (cfg->n_devices < 0 || cfg->n_devices > YCGE_MAX_DEVICES) { g_create_error = "n_devices out of range"; return YCGE_ERR_INVALID_ARG; }
    if (cfg->multi_device_exchange != YCGE_EXCHANGE_PEER_PUSH && cfg->multi_device_exchange != YCGE_EXCHANGE_RCCL) { g_create_error = "multi_device_exchange: unknown mode"; return YCGE_ERR_INVALID_ARG; }
    // the collective behind the one call (ABI 9): asked for, with a device list, and librccl.so there - else the peers push their tiles as before
    const bool want_rccl = cfg->multi_device_exchange == YCGE_EXCHANGE_RCCL && cfg->n_devices >= 1 && cfg->world_size == 1 && load_rccl().ok;
    if (cfg->n_devices <= 1 && !want_rccl) {
        ycge_config one = *cfg;
        if (cfg->n_devices == 1) one.device = cfg->devices[0];
        one.n_devices = 0;
        one.multi_device_exchange = YCGE_EXCHANGE_PEER_PUSH;
        return create_one(&one, nullptr, out);
    }
    // ---- one process, n_devices GPUs: this context is rank 0 on devices[0]; ranks 1.. live in peer contexts it owns
    if (cfg->world_size != 1 || cfg->rank != 0) { g_create_error = "n_devices > 1 excludes rank/world_size (that is the one-process-per-GPU form)"; return YCGE_ERR_INVALID_ARG; }
    ycge_config base = *cfg;
    base.world_size = cfg->n_devices; base.n_devices = 0;
    base.rank = 0; base.device = cfg->devices[0];
    base.multi_device_exchange = want_rccl ? YCGE_EXCHANGE_RCCL : YCGE_EXCHANGE_PEER_PUSH;
    if (want_rccl)
        for (int a = 0; a < cfg->n_devices; a++)
            for (int b = a + 1; b < cfg->n_devices; b++)
                if (cfg->devices[a] == cfg->devices[b]) { g_create_error = "multi_device_exchange = RCCL needs distinct devices (RCCL refuses two ranks on one GPU)"; return YCGE_ERR_INVALID_ARG; }
    ycge_ctx *root = nullptr;
    int rc = create_one(&base, nullptr, &root);
    if (rc != YCGE_OK) return rc;
    // (whatever throws below - a vector that grows, a thread that cannot start - takes the root, its peers and their threads with it)
    struct Owner { ycge_ctx *root, *peer; ~Owner() { if (peer) ycge_destroy(peer); if (root) ycge_destroy(root); } } owner{root, nullptr};
    root->cfg.n_devices = cfg->n_devices;
    root->peers.reserve((size_t)cfg->n_devices);
    for (int r = 1; r < cfg->n_devices; r++) {
        ycge_config pc = base;
        pc.rank = r; pc.device = cfg->devices[r];
        ycge_ctx *peer = nullptr;
        rc = create_one(&pc, root, &peer);
        if (rc != YCGE_OK) return rc;
        owner.peer = peer;
        root->peers.push_back(peer);
        owner.peer = nullptr;
        if (peer->device != root->device) {        // the peer's kernels write into the root's frame buffers over xGMI
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, peer->device, root->device) != hipSuccess || !can) {
                g_create_error = "devices cannot access each other's memory (no xGMI / PCIe peer path)";
                return YCGE_ERR_DEVICE;
            }
            (void)hipSetDevice(peer->device);
            const hipError_t pe = hipDeviceEnablePeerAccess(root->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { g_create_error = "hipDeviceEnablePeerAccess failed"; return YCGE_ERR_DEVICE; }
            (void)hipGetLastError();
        }
    }
    if (want_rccl) {            // one communicator per device, all in this process (ncclCommInitAll); rank r = devices[r]
        const RcclApi &R = load_rccl();
        root->nccl_comms.assign((size_t)cfg->n_devices, nullptr);
        const int nr = R.CommInitAll(root->nccl_comms.data(), cfg->n_devices, cfg->devices);
        if (nr != 0) {
            g_create_error = std::string("ncclCommInitAll failed: ") + (R.GetErrorString ? R.GetErrorString(nr) : "?");
            root->nccl_comms.clear();
            return YCGE_ERR_DEVICE;
        }
        root->exchange_mode = YCGE_EXCHANGE_RCCL;
        (void)hipSetDevice(root->device);
    }
    for (ycge_ctx *peer : root->peers) {        // each peer's share of a frame is issued by its own thread (trace_on_all_devices)
        peer->worker = new ycge_ctx::PeerWorker;
        peer->worker->th = std::thread(peer_worker_main, root, peer);
    }
    (void)hipSetDevice(root->device);
    owner.root = nullptr;
    *out = root;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

void ycge_destroy(ycge_ctx *c)
try {
    if (!c) return;
    if (c->worker) {
        { std::lock_guard<std::mutex> g(c->worker->m); c->worker->job = -1; }
        c->worker->cv.notify_all();
        if (c->worker->th.joinable()) c->worker->th.join();
        delete c->worker;
        c->worker = nullptr;
    }
    for (ycge_ctx *p : c->peers) ycge_destroy(p);
    c->peers.clear();
    if (!c->nccl_comms.empty()) {
        const RcclApi &R = load_rccl();
        for (void *cm : c->nccl_comms) if (cm && R.CommDestroy) (void)R.CommDestroy(cm);
        c->nccl_comms.clear();
    }
    c->all_slabs.release();
    (void)hipSetDevice(c->device);
    // frames in flight may still be on ANY of the context's streams (TAA, post stage and read-back on taa_stream / stream2): everything
    // is drained before the first buffer goes (not left to hipFree's implicit synchronisation)
    if (c->fan_stream) (void)hipStreamSynchronize(c->fan_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->taa_stream) (void)hipStreamSynchronize(c->taa_stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    c->current_hdr.release(); c->g_albedo.release(); c->g_normal.release(); c->g_depth.release(); c->taa_hist.release();
    c->prev_normal.release(); c->prev_depth.release(); c->sky.release(); c->prev_sky.release();
    c->dbg_rays.release(); c->dbg_hit_t.release(); c->dbg_prim.release(); c->dbg_sub.release(); c->dbg_rng.release();
    c->counters.release(); c->wave_prof.release(); c->own_slab.release(); c->dbg_counters.release(); c->taa_block_ctr.release(); c->taa_part_ctr.release();
    c->t_hdr.release(); c->t_albedo.release(); c->t_normal.release(); c->t_depth.release(); c->t_sky.release();
    if (c->taa_stream) { (void)hipStreamSynchronize(c->taa_stream); (void)hipStreamDestroy(c->taa_stream); c->taa_stream = nullptr; }
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); c->stream2 = nullptr; }
    if (c->flight_fork_ev) (void)hipEventDestroy(c->flight_fork_ev);
    if (c->placed_flag) (void)hipFree(c->placed_flag);
    for (int k = 0; k < 2; k++) if (c->tile_trace_ev[k]) (void)hipEventDestroy(c->tile_trace_ev[k]);
    for (hipEvent_t ev : {c->flight_taa_ev, c->post_hist_ev, c->post_done_ev, c->post_set_ev[0], c->post_set_ev[1], c->post_set_ev[2]}) if (ev) (void)hipEventDestroy(ev);
    c->stack_spill2.release(); c->stack_spill_side.release(); c->stack_spill_side2.release();
    c->wf2_q0.release(); c->wf2_q1.release(); c->wf2_hit.release(); c->wf2_lq.release(); c->wf2_seg.release(); c->wf2_counts.release();
    release_resident(c);
    c->batch_spill[0].release(); c->batch_spill[1].release();
    for (int k = 0; k < 2; k++) if (c->batch_done[k]) { (void)hipEventDestroy(c->batch_done[k]); c->batch_done[k] = nullptr; }
    for (int k = 0; k < 3; k++) if (c->set_resolved_ev[k]) (void)hipEventDestroy(c->set_resolved_ev[k]);
    for (int k = 0; k < 3; k++) { c->flight_order[k].release(); c->flight_ws[k].release(); if (c->flight_order_ev[k]) (void)hipEventDestroy(c->flight_order_ev[k]); }
    for (hipEvent_t ev : c->flight_ev) (void)hipEventDestroy(ev);
    c->flight_ev.clear();
    c->alt_hdr.release(); c->alt_albedo.release(); c->alt_normal.release(); c->alt_depth.release(); c->alt_sky.release();
    c->alt2_hdr.release(); c->alt2_albedo.release(); c->alt2_normal.release(); c->alt2_depth.release(); c->alt2_sky.release();
    c->den_a.release(); c->den_b.release(); c->unit_n.release(); c->exp_terms.release(); c->exp_scratch.release(); c->d_sdr.release(); c->d_sdr2.release(); c->atrous_statw.release(); c->tone_state.release();
    for (auto *sc : c->schedules) delete sc;
    c->schedules.clear();
    c->post_progress.release();
    c->alt_post.release();
    c->wf_q0.release(); c->wf_q1.release(); c->wf_hit.release(); c->wf_lq.release(); c->wf_seg.release(); c->wf_counts.release(); c->tile_order.release(); c->block_cost.release(); c->cost_snap.release(); c->block_order.release(); c->order_ws.release(); c->stack_spill.release(); c->path_stack.release();
    c->d_scene_nodes.release(); c->d_walk_nodes.release(); c->d_grid_owner.release(); c->d_mesh_arena.release(); c->d_scene_leaf.release(); c->d_prims.release();
    c->d_bvh_items.release(); c->d_bvh_scratch.release(); c->d_bvh_ref.release(); c->d_bvh_res.release();
    c->d_materials.release(); c->d_meshes.release(); c->d_grids.release(); c->d_cells.release(); c->d_lut.release(); c->d_lights.release(); c->d_tex_pixels.release(); c->d_tex_info.release();
    for (int k = 0; k < 2; k++) { if (c->tex_stage[k]) (void)hipHostFree(c->tex_stage[k]); if (c->tex_stage_ev[k]) (void)hipEventDestroy(c->tex_stage_ev[k]); }
    if (c->tex_order_ev) (void)hipEventDestroy(c->tex_order_ev);
    if (c->out_stage) { (void)hipHostFree(c->out_stage); c->out_stage = nullptr; c->out_stage_bytes = 0; }
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {c->fan_ev[0], c->fan_ev[1], c->traced_ev, c->order_ev, c->pushed_ev}) if (ev) (void)hipEventDestroy(ev);
    if (c->fan_stream) (void)hipStreamDestroy(c->fan_stream);
    if (c->h_n_fan) (void)hipHostFree(c->h_n_fan);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
catch (...) { (void)ycge_host::abi_catch(nullptr); }

const char *ycge_last_error(const ycge_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ycge_device_info(ycge_ctx *c, char *name, size_t name_bytes, int32_t *compute_units)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (name && name_bytes) { std::strncpy(name, c->device_name, name_bytes - 1); name[name_bytes - 1] = 0; }
    if (compute_units) *compute_units = c->compute_units;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

static int upload_lights(ycge_ctx *c, const ycge_light *lights, int n)
{
    std::vector<GLight> L(n);
    c->any_light_lit = false;
    for (int i = 0; i < n; i++) {
        L[i].pos[0] = lights[i].position.x; L[i].pos[1] = lights[i].position.y; L[i].pos[2] = lights[i].position.z;
        L[i].color[0] = lights[i].color.x; L[i].color[1] = lights[i].color.y; L[i].color[2] = lights[i].color.z;
        L[i].intensity = lights[i].intensity;
        L[i].dark = (lights[i].intensity == 0.0f && std::isfinite(lights[i].color.x) && std::isfinite(lights[i].color.y) && std::isfinite(lights[i].color.z)) ? 1.0f : 0.0f;
        if (L[i].dark == 0.0f) c->any_light_lit = true;
    }
    HIP_TRY(c, c->d_lights.upload(L));
    c->sd.lights = c->d_lights.p;
    c->sd.n_lights = n;
    return YCGE_OK;
}

}
namespace ycge_host {

// ---- device -> host.  The device writes host memory in two places only: page-locked memory (the library's own - hipHostMalloc - or whole
// pages the caller registered, ycge_pin_host_buffer) and the library's own staging buffer below; a PAGEABLE destination is filled by the CPU
// from that staging buffer.  Handing a pageable pointer to hipMemcpy lets the runtime choose how the bytes get there, and for transfers of
// a megabyte and more it page-locks the caller's pages on the fly and lets the copy engine write them: a mapping of process heap whose
// lifetime neither the caller nor the library controls (the allocator trims and regrows the heap, the runtime caches what it pinned).
// Twice in ~25 runs of the GPU suite in round 4 and once in the first full run of round 5 - AFTER the registered arrays of the Python
// mirror had been given pages of their own - a read-back died with "Memory access fault by GPU ... Write access" at a heap address.
// (the WHOLE range [p, p + bytes): a destination that starts inside a page-locked block and runs past its end - a registered sub-range, an
// array sized for an older console - would let the copy engine write pageable heap behind it, the very fault class this exists to exclude.
// The runtime is asked for the allocation the first byte lies in (hipMemGetAddressRange on its device alias: base and size of the block
// hipHostMalloc / hipHostRegister made) and the range must end inside it; a runtime that cannot say is asked about the last byte too, and
// the two ends must be page-locked host memory whose device aliases lie exactly as far apart as the host addresses.  Anything else is
// answered "no": the staging copy is always right.)
bool host_memory_is_page_locked(const void *p, size_t bytes)
{
    auto locked = [](const void *q, hipPointerAttribute_t &a) {
        std::memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }      // (older runtimes: an error for plain heap memory)
        return a.type == hipMemoryTypeHost;
    };
    hipPointerAttribute_t a0, a1;
    if (!locked(p, a0)) return false;
    if (bytes <= 1) return true;
    hipDeviceptr_t base = nullptr;
    size_t block = 0;
    if (a0.devicePointer && hipMemGetAddressRange(&base, &block, (hipDeviceptr_t)a0.devicePointer) == hipSuccess && base && block) {
        const uint8_t *b = (const uint8_t *)base, *d = (const uint8_t *)a0.devicePointer;
        return d >= b && (size_t)(d - b) <= block && bytes <= block - (size_t)(d - b);
    }
    (void)hipGetLastError();
    if (!locked((const uint8_t *)p + (bytes - 1), a1)) return false;
    return a0.devicePointer && a1.devicePointer && (const uint8_t *)a1.devicePointer - (const uint8_t *)a0.devicePointer == (ptrdiff_t)(bytes - 1);
}
int ensure_out_stage(ycge_ctx *c, size_t bytes)
{
    if (c->out_stage_bytes >= bytes) return YCGE_OK;
    if (c->out_stage) { (void)hipHostFree(c->out_stage); c->out_stage = nullptr; c->out_stage_bytes = 0; }
    HIP_TRY(c, hipHostMalloc(&c->out_stage, bytes, hipHostMallocDefault));
    c->out_stage_bytes = bytes;
    return YCGE_OK;
}
// synchronous copy of `bytes` from device memory of the current device to `dst`; nothing of this context may be in flight on other streams
int copy_out(ycge_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return YCGE_OK;
    if (host_memory_is_page_locked(dst, bytes)) { HIP_TRY(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return YCGE_OK; }
    const size_t chunk = (size_t)32 << 20;
    const int rc = ensure_out_stage(c, bytes < chunk ? bytes : chunk);
    if (rc != YCGE_OK) return rc;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        HIP_TRY(c, hipMemcpyAsync(c->out_stage, (const uint8_t *)src + off, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::memcpy((uint8_t *)dst + off, c->out_stage, n);
    }
    return YCGE_OK;
}
// the SDR frame of a SYNCHRONOUS call into a pageable array: the copy was queued into the staging buffer (run_post); the stream has been waited for
void finish_staged_sdr(ycge_ctx *c)
{
    if (!c->staged_sdr_dst) return;
    std::memcpy(c->staged_sdr_dst, c->out_stage, c->staged_sdr_bytes);
    c->staged_sdr_dst = nullptr; c->staged_sdr_bytes = 0;
}

// every stream a frame of this context may still be running on
int quiesce(ycge_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    // a pipelined tiled caller may alternate several streams of its own between ycge_trace_tiles and ycge_resolve_gathered; the
    // scene updates rewrite live allocations in place (DevBuf::upload), so wait for the whole device - this is a per-scene-change
    // call, never part of a frame
    HIP_TRY(c, hipDeviceSynchronize());
    c->async_outstanding = false; c->set_read[0] = c->set_read[1] = c->set_read[2] = false;        // (frames in flight included)
    c->post_hist_pending = c->post_busy = c->post_set_pending[0] = c->post_set_pending[1] = c->post_set_pending[2] = false;
    return YCGE_OK;
}

// ---- argument checks of ycge_scene_upload (pure host code; also exported as ycge_validate_scene)
int validate_scene(const ycge_scene *s, std::string &msg)
{
    char buf[256];
    auto bad = [&](int code, const char *fmt, int a = 0, int b = 0, int c2 = 0) { std::snprintf(buf, sizeof buf, fmt, a, b, c2); msg = buf; return code; };
    if (!s) return bad(YCGE_ERR_INVALID_ARG, "null scene");
    if (s->n_prims < 0 || s->n_materials < 0 || s->n_lights < 0 || s->n_meshes < 0 || s->n_grids < 0) return bad(YCGE_ERR_INVALID_ARG, "negative count");
    if ((s->n_prims > 0 && !s->prims) || (s->n_materials > 0 && !s->materials) || (s->n_lights > 0 && !s->lights) || (s->n_meshes > 0 && !s->meshes) ||
        (s->n_grids > 0 && !s->grids))
        return bad(YCGE_ERR_INVALID_ARG, "null array with a non-zero count");
    auto mat_ok = [&](int mi) { return mi >= 0 && mi < s->n_materials; };
    for (int i = 0; i < s->n_materials; i++) {
        const int k = s->materials[i].kind;
        if (k == YCGE_MAT_TEXTURED) {
            if (s->materials[i].texture < 0 || s->materials[i].texture >= s->n_textures)
                return bad(YCGE_ERR_INVALID_ARG, "material %d: texture index %d out of range (%d textures)", i, s->materials[i].texture, s->n_textures);
        } else if (k != YCGE_MAT_CONSTANT && k != YCGE_MAT_CHECKER) return bad(YCGE_ERR_UNSUPPORTED, "material %d: unknown kind %d", i, k);
    }
    if (s->n_textures < 0 || (s->n_textures > 0 && !s->textures)) return bad(YCGE_ERR_INVALID_ARG, "bad texture array");
    for (int i = 0; i < s->n_textures; i++) {
        const ycge_texture &t = s->textures[i];
        if (t.frame_bytes_per_pixel != 0 && t.frame_bytes_per_pixel != 3 && t.frame_bytes_per_pixel != 4) return bad(YCGE_ERR_INVALID_ARG, "texture %d: frame_bytes_per_pixel %d (0 static, 3 BGR, 4 BGRA)", i, t.frame_bytes_per_pixel);
        if (t.width < 1 || t.height < 1 || (!t.pixels && t.frame_bytes_per_pixel == 0)) return bad(YCGE_ERR_INVALID_ARG, "texture %d: needs width, height >= 1 and pixels (%d x %d)", i, t.width, t.height);
        if ((long long)t.width * t.height > (1ll << 28)) return bad(YCGE_ERR_UNSUPPORTED, "texture %d: above 2^28 pixels", i);
    }
    for (int mi = 0; mi < s->n_meshes; mi++) {
        const ycge_mesh &m = s->meshes[mi];
        if (m.n_triangles < 0 || (m.n_triangles > 0 && !m.triangles)) return bad(YCGE_ERR_INVALID_ARG, "mesh %d: bad triangle array", mi);
        if (!m.tri_material && !mat_ok(m.material)) return bad(YCGE_ERR_INVALID_ARG, "mesh %d: material out of range", mi);
        if (m.tri_material)
            for (int t = 0; t < m.n_triangles; t++)
                if (!mat_ok(m.tri_material[t])) return bad(YCGE_ERR_INVALID_ARG, "mesh %d: triangle material out of range", mi);
    }
    for (int gi = 0; gi < s->n_grids; gi++) {
        const ycge_grid &g = s->grids[gi];
        if (g.nx <= 0 || g.ny <= 0 || g.nz <= 0 || !g.cells) return bad(YCGE_ERR_INVALID_ARG, "grid %d: empty", gi);
        if ((uint64_t)g.nx * g.ny * g.nz >= (1u << 30)) return bad(YCGE_ERR_UNSUPPORTED, "grid %d: more than 2^30 cells", gi);
        // (the voxel walk forms brick indices with 24-bit multiplies: ((z >> 3) * bricks_y + (y >> 3)) * bricks_x)
        if ((uint64_t)((g.nz + 7) >> 3) * (uint64_t)((g.ny + 7) >> 3) >= (1u << 23) || ((g.nx + 7) >> 3) >= (1 << 23))
            return bad(YCGE_ERR_UNSUPPORTED, "grid %d: more than 2^23 bricks across one face", gi);
        if (g.n_lookup < 0 || (g.n_lookup > 0 && !g.lookup)) return bad(YCGE_ERR_INVALID_ARG, "grid %d: bad lookup table", gi);
        for (int k = 0; k < g.n_lookup; k++)
            if (!mat_ok(g.lookup[k].material)) return bad(YCGE_ERR_INVALID_ARG, "grid %d: lookup entry %d names a material out of range", gi, k);
    }
    for (int i = 0; i < s->n_prims; i++) {
        const ycge_prim &q = s->prims[i];
        if (q.type < YCGE_PRIM_SPHERE || q.type > YCGE_PRIM_VOLUME_GRID) return bad(YCGE_ERR_INVALID_ARG, "prim %d: unknown type %d", i, q.type);
        if (q.type == YCGE_PRIM_MESH) { if (q.ref < 0 || q.ref >= s->n_meshes) return bad(YCGE_ERR_INVALID_ARG, "prim %d: mesh ref out of range", i); }
        else if (q.type == YCGE_PRIM_VOLUME_GRID) { if (q.ref < 0 || q.ref >= s->n_grids) return bad(YCGE_ERR_INVALID_ARG, "prim %d: grid ref out of range", i); }
        else if (!mat_ok(q.material)) return bad(YCGE_ERR_INVALID_ARG, "prim %d: material out of range", i);
    }
    return YCGE_OK;
}

// Scene.Objects -> object records + scene BVH (BVH ctor, BVH.cs:29-97), built on the host ONCE per update (rank 0's context
// keeps the metadata of the last full upload: mesh root boxes, grid bounds, material count) and installed on every device.
struct ObjectsHost {
    std::vector<GPrim> gprims;
    std::vector<GNode> scene_nodes;
    std::vector<uint32_t> leaf_prims;
    uint32_t scene_root = YCGE_REF_NONE_VALUE;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    int wf_rounds = 2, spill_levels = 0;
    std::vector<int32_t> grid_owner;       // per grid of the last upload: the object that holds it (-1: none), SceneDev::grid_owner
    bool grid_owner_unique = true;         // false: two objects hold the same grid - no walk tree
    float walk_t_limit = 0.0f;             // the smallest GGrid::cull_t_limit of the grids in Objects
    bool analytic_only = true;             // no mesh and no voxel grid among the objects (SceneDev::analytic_only)
};

// Scene.Objects as device records + what Scene.RebuildBVH gets from every object's TryGetBounds (BVH.cs:32-53); no tree yet
int flatten_objects(ycge_ctx *c, const ycge_prim *prims, int n_prims, ObjectsHost &oh, BoundsSoA &items)
{
    std::vector<GPrim> &gprims = oh.gprims;
    gprims.assign(n_prims, GPrim{});
    items.resize(n_prims);
    oh.grid_owner.assign(c->grid_solid.size(), -1); oh.grid_owner_unique = true; oh.walk_t_limit = HUGE_VALF;
    oh.analytic_only = true;
    for (int i = 0; i < n_prims; i++) {
        const ycge_prim &q = prims[i];
        if (q.type == YCGE_PRIM_MESH || q.type == YCGE_PRIM_VOLUME_GRID) oh.analytic_only = false;
        GPrim &g = gprims[i];
        std::memset(&g, 0, sizeof g);
        g.type = q.type; g.material = q.material; g.ref = q.ref; g.reflectivity = q.reflectivity;
        const float *p = q.p;
        switch (q.type) {
        case YCGE_PRIM_SPHERE: for (int k = 0; k < 4; k++) g.p[k] = p[k]; break;
        case YCGE_PRIM_PLANE: {      // Plane ctor, Surfaces.cs:19-28
            H3 n = h_norm(H3{p[3], p[4], p[5]});
            g.p[0] = n.x; g.p[1] = n.y; g.p[2] = n.z;
            g.p[3] = n.x * p[0] + n.y * p[1] + n.z * p[2];
            break;
        }
        case YCGE_PRIM_DISK: {       // Disk ctor, Surfaces.cs:84-94
            H3 n = h_norm(H3{p[3], p[4], p[5]});
            g.p[0] = p[0]; g.p[1] = p[1]; g.p[2] = p[2]; g.p[3] = n.x; g.p[4] = n.y; g.p[5] = n.z;
            g.p[6] = p[6] * p[6];
            g.p[7] = n.x * p[0] + n.y * p[1] + n.z * p[2];
            break;
        }
        case YCGE_PRIM_XYRECT: case YCGE_PRIM_XZRECT: case YCGE_PRIM_YZRECT: for (int k = 0; k < 5; k++) g.p[k] = p[k]; break;
        case YCGE_PRIM_BOX: for (int k = 0; k < 6; k++) g.p[k] = p[k]; break;
        case YCGE_PRIM_CYLINDER_Y:   // CylinderY ctor, BoundedObjects.cs:128-137
            g.p[0] = p[0]; g.p[1] = p[2]; g.p[2] = p[3]; g.p[3] = p[3] * p[3];
            g.p[4] = cs_min(p[4], p[5]); g.p[5] = cs_max(p[4], p[5]); g.p[6] = p[6];
            break;
        case YCGE_PRIM_TRIANGLE: {   // Triangle ctor, Triangle.cs:36-45
            float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
            float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
            float nnx = e1y * e2z - e1z * e2y, nny = e1z * e2x - e1x * e2z, nnz = e1x * e2y - e1y * e2x;
            float inv_len = 1.0f / cs_max(1e-20f, cs_sqrt(nnx * nnx + nny * nny + nnz * nnz));
            g.p[0] = p[0]; g.p[1] = p[1]; g.p[2] = p[2];
            g.p[3] = e1x; g.p[4] = e1y; g.p[5] = e1z; g.p[6] = e2x; g.p[7] = e2y; g.p[8] = e2z;
            g.p[9] = nnx * inv_len; g.p[10] = nny * inv_len; g.p[11] = nnz * inv_len;
            break;
        }
        case YCGE_PRIM_MESH: {      // root box + root reference ride in the object record (one fetch less per query)
            if (q.ref < 0 || q.ref >= (int)c->gmeshes_host.size()) return c->fail(YCGE_ERR_INVALID_ARG, "prim %d: mesh ref out of range", i);
            const GMesh &gm = c->gmeshes_host[q.ref];
            for (int a = 0; a < 3; a++) { g.p[a] = gm.root_min[a]; g.p[3 + a] = gm.root_max[a]; }
            g.p[6] = u2f(gm.root_ref);
            break;
        }
        case YCGE_PRIM_VOLUME_GRID:
            if (q.ref < 0 || q.ref >= (int)c->grid_bounds.size()) return c->fail(YCGE_ERR_INVALID_ARG, "prim %d: grid ref out of range", i);
            for (int k = 0; k < 7; k++) g.p[k] = c->grid_solid[q.ref][k];       // box of the grid's solid voxels + how far along a ray it may be trusted (grid_cull in the walk)
            if (oh.grid_owner[(size_t)q.ref] >= 0) oh.grid_owner_unique = false;
            oh.grid_owner[(size_t)q.ref] = i;
            oh.walk_t_limit = cs_min(oh.walk_t_limit, c->grid_solid[q.ref][6]);
            break;
        default: return c->fail(YCGE_ERR_INVALID_ARG, "prim %d: unknown type %d", i, q.type);
        }
        if (q.type != YCGE_PRIM_MESH && q.type != YCGE_PRIM_VOLUME_GRID && !(q.material >= 0 && q.material < c->n_materials))
            return c->fail(YCGE_ERR_INVALID_ARG, "prim %d: material out of range", i);
        float b[6], cen[3];
        if (!prim_bounds(q, c->meshes, c->grid_bounds, b, cen)) return c->fail(YCGE_ERR_INVALID_ARG, "Unbounded Hittable (prim %d)", i);   // BVH.cs:37-40
        for (int a = 0; a < 3; a++) { items.mn[a][i] = b[a]; items.mx[a][i] = b[3 + a]; items.c[a][i] = cen[a]; }
    }
    // can any surface take the mirror branch (Reflectivity >= MirrorThreshold, RaytraceRenderer.cs:559)?
    bool can_mirror = c->materials_can_mirror;
    for (int i = 0; i < n_prims; i++) {
        const int ty = prims[i].type;
        const bool overrides = ty == YCGE_PRIM_PLANE || ty == YCGE_PRIM_DISK || ty == YCGE_PRIM_XYRECT || ty == YCGE_PRIM_XZRECT || ty == YCGE_PRIM_YZRECT || ty == YCGE_PRIM_BOX;
        if (overrides && prims[i].reflectivity >= c->cfg.mirror_threshold) can_mirror = true;
    }
    oh.wf_rounds = can_mirror ? 2 + c->cfg.max_mirror_bounces : 2;
    return YCGE_OK;
}

int check_scene_depth(ycge_ctx *c, int max_depth, int &spill_levels)
{
    // (the mesh trees are built once, by the root context: a peer's own max_mesh_depth mirrors it - install_scene - and the root's is what counts)
    const int mesh_depth = c->parent ? c->parent->max_mesh_depth : c->max_mesh_depth;
    if (max_depth > 128) return c->fail(YCGE_ERR_STACK_DEPTH, "scene BVH depth %d exceeds the reference's 128-entry stack (BVH.cs:118)", max_depth);
    if (max_depth + 4 + mesh_depth + 2 > YCGE_TRAVERSAL_STACK)
        return c->fail(YCGE_ERR_STACK_DEPTH, "combined traversal depth %d + %d exceeds the device stack", max_depth, mesh_depth);
    // levels the per-lane stack can need beyond its LDS part: scene depth + 4 leaf objects + deepest mesh
    const int need = max_depth + 4 + mesh_depth + 2 - YCGE_LDS_STACK_LEVELS;
    spill_levels = need > 0 ? need : 0;
    return YCGE_OK;
}

// the scene-level BVH by the host builder (ycge_accel.cpp)
int build_scene_tree_host(ycge_ctx *c, const BoundsSoA &items, ObjectsHost &oh)
{
    build_tree(items, TreeFlavour::Scene, c->scene_tree);
    c->scene_tree_on_device = false;
    c->bvh_host_builds++;
    const int rc = check_scene_depth(c, c->scene_tree.max_depth, oh.spill_levels);
    if (rc != YCGE_OK) return rc;
    oh.scene_root = to_gpu_nodes(c->scene_tree, REF_SCENE_NODE, REF_SCENE_LEAF, 0, 0, 3, oh.scene_nodes);
    oh.leaf_prims.assign(c->scene_tree.leaf_index.begin(), c->scene_tree.leaf_index.end());
    if (c->scene_tree.root >= 0)
        for (int a = 0; a < 3; a++) { oh.root_min[a] = c->scene_tree.nodes[c->scene_tree.root].mn[a]; oh.root_max[a] = c->scene_tree.nodes[c->scene_tree.root].mx[a]; }
    return YCGE_OK;
}

// SceneDev::walk_nodes: the walk tree of a world of voxel grids, from the tree and the object records now on the device (k_scene_walk,
// ycge_bvh_build.hip) - whichever builder made the tree.  None (null) without a grid, when the root is a leaf, when two objects hold
// the same grid (grid_owner would be ambiguous) and under YCGE_NO_WALK_TREE.
int install_walk_tree(ycge_ctx *c, const ObjectsHost &oh, int n_inner, uint32_t scene_root)
{
    SceneDev &sd = c->sd;
    sd.walk_nodes = nullptr; sd.grid_owner = nullptr; sd.walk_root_ref = YCGE_REF_NONE_VALUE; sd.walk_t_limit = 0.0f; c->walk_scene_nodes = 0;
    if (!c->has_grid || n_inner <= 0 || c->knobs.no_walk_tree || !oh.grid_owner_unique || YCGE_REF_KIND(scene_root) != REF_SCENE_NODE) return YCGE_OK;
    HIP_TRY(c, c->d_grid_owner.upload(oh.grid_owner));
    const size_t n_walk = (size_t)n_inner * (1 + 2 * YCGE_WALK_LEAF_NODES);
    HIP_TRY(c, c->d_walk_nodes.reserve(n_walk));
    HIP_TRY(c, hipMemsetAsync(c->d_walk_nodes.p, 0, n_walk * sizeof(GNode), c->stream));
    const int e = ycge_launch_scene_walk(c->d_scene_nodes.p, n_inner, c->d_scene_leaf.p, c->d_prims.p, c->d_walk_nodes.p, c->stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_scene_walk launch failed: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    sd.walk_nodes = c->d_walk_nodes.p; sd.grid_owner = c->d_grid_owner.p;
    sd.walk_root_ref = YCGE_REF(REF_WALK_NODE, YCGE_REF_PAYLOAD(scene_root));
    sd.walk_t_limit = oh.walk_t_limit;
    c->walk_scene_nodes = n_inner;
    return YCGE_OK;
}

int install_objects(ycge_ctx *c, const ObjectsHost &oh)
{
    HIP_TRY(c, hipSetDevice(c->device));
    c->wf_rounds = oh.wf_rounds;
    if (oh.spill_levels != c->spill_levels) {
        c->spill_levels = oh.spill_levels;
        const int rc2 = alloc_tile_buffers(c);
        if (rc2 != YCGE_OK) return rc2;
    }
    HIP_TRY(c, c->d_prims.upload(oh.gprims)); HIP_TRY(c, c->d_scene_nodes.upload(oh.scene_nodes)); HIP_TRY(c, c->d_scene_leaf.upload(oh.leaf_prims));
    SceneDev &sd = c->sd;
    sd.scene_nodes = c->d_scene_nodes.p; sd.scene_leaf_prims = c->d_scene_leaf.p; sd.prims = c->d_prims.p;
    sd.scene_root_ref = oh.scene_root;
    sd.analytic_only = (oh.analytic_only && !c->knobs.no_analytic_walk) ? 1 : 0;
    for (int a = 0; a < 3; a++) { sd.scene_root_min[a] = oh.root_min[a]; sd.scene_root_max[a] = oh.root_max[a]; }
    c->block_order_valid = false; c->flight_order_frame[0] = c->flight_order_frame[1] = c->flight_order_frame[2] = -1;
    return install_walk_tree(c, oh, (int)oh.scene_nodes.size(), oh.scene_root);
}

int build_objects(ycge_ctx *c, const ycge_prim *prims, int n_prims, ObjectsHost &oh)
{
    BoundsSoA items;
    const int rc = flatten_objects(c, prims, n_prims, oh, items);
    return rc != YCGE_OK ? rc : build_scene_tree_host(c, items, oh);
}

// ycge_scene_update_objects, device form: the object records and their boxes go up, ycge_bvh_build.hip builds the tree where the
// trace reads it.  Returns 1 when the kernel declines (a tree deeper than the reference's stack; the caller then builds on the host).
int install_objects_device_built(ycge_ctx *c, ycge_ctx *root, const ObjectsHost &oh, const BoundsSoA &items)
{
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = (int)items.size();
    std::vector<float> planes((size_t)9 * n);
    for (int a = 0; a < 3; a++)
        for (int i = 0; i < n; i++) {
            planes[(size_t)a * n + i] = items.mn[a][i]; planes[(size_t)(3 + a) * n + i] = items.mx[a][i]; planes[(size_t)(6 + a) * n + i] = items.c[a][i];
        }
    HIP_TRY(c, c->d_prims.upload(oh.gprims)); HIP_TRY(c, c->d_bvh_items.upload(planes));
    HIP_TRY(c, c->d_scene_nodes.reserve((size_t)n)); HIP_TRY(c, c->d_scene_leaf.reserve((size_t)n));
    HIP_TRY(c, c->d_bvh_scratch.reserve(ycge_bvh_build_scratch_bytes(n))); HIP_TRY(c, c->d_bvh_ref.reserve((size_t)2 * n * sizeof(RefNode)));
    HIP_TRY(c, c->d_bvh_res.reserve(sizeof(BvhBuildResult)));
    const int e = ycge_launch_scene_bvh_build(c->d_bvh_items.p, n, c->d_bvh_scratch.p, c->d_bvh_ref.p, c->d_scene_nodes.p, c->d_scene_leaf.p, c->d_bvh_res.p, c->knobs.bvh_waves, c->stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_scene_bvh_build launch failed: %s", hipGetErrorString((hipError_t)e));
    BvhBuildResult res;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    { const int cr = copy_out(c, &res, c->d_bvh_res.p, sizeof res); if (cr != YCGE_OK) return cr; }       // (through the library's page-locked staging: the device writes no caller or stack memory)
    if (res.fallback) return 1;
    int spill = 0;
    const int rc = check_scene_depth(c, res.max_depth, spill);
    if (rc != YCGE_OK) return rc;
    c->wf_rounds = oh.wf_rounds;
    if (spill != c->spill_levels) {
        c->spill_levels = spill;
        const int rc2 = alloc_tile_buffers(c);
        if (rc2 != YCGE_OK) return rc2;
    }
    SceneDev &sd = c->sd;
    sd.scene_nodes = c->d_scene_nodes.p; sd.scene_leaf_prims = c->d_scene_leaf.p; sd.prims = c->d_prims.p;
    sd.scene_root_ref = res.root_ref;
    sd.analytic_only = (oh.analytic_only && !c->knobs.no_analytic_walk) ? 1 : 0;
    for (int a = 0; a < 3; a++) { sd.scene_root_min[a] = res.root_min[a]; sd.scene_root_max[a] = res.root_max[a]; }
    c->block_order_valid = false;
    const int rc3 = install_walk_tree(c, oh, res.n_inner, res.root_ref);
    if (rc3 != YCGE_OK) return rc3;
    if (c == root) { c->scene_tree_on_device = true; c->dev_tree_nodes = res.n_nodes; c->dev_tree_items = n; c->scene_tree.max_depth = res.max_depth; c->scene_tree.sort_fallbacks = (int32_t)res.sorts; }
    return YCGE_OK;
}

// the arrays of a full upload, built once and installed on every device
struct SceneArrays {
    std::vector<GMaterial> mats;
    std::vector<uint8_t> arena, cells;
    std::vector<GMesh> gmeshes;
    std::vector<GGrid> ggrids;
    std::vector<int32_t> lut;
    bool any_transparent = false, has_grid = false, any_textured = false;
    std::vector<uint32_t> tex_pixels;
    std::vector<int32_t> tex_info;          // per texture {first word, width, height, 0 | live-frame flags}
    uint32_t tl_offset = 0;          // treelet region of the arena (append_treelets), 0 = none
};

int install_scene(ycge_ctx *c, const SceneArrays &A, const ObjectsHost &oh, const ycge_scene *s)
{
    HIP_TRY(c, hipSetDevice(c->device));
    c->have_scene = false;
    HIP_TRY(c, c->d_materials.upload(A.mats)); HIP_TRY(c, c->d_mesh_arena.upload(A.arena));
    HIP_TRY(c, c->d_meshes.upload(A.gmeshes)); HIP_TRY(c, c->d_grids.upload(A.ggrids)); HIP_TRY(c, c->d_cells.upload(A.cells));
    HIP_TRY(c, c->d_lut.upload(A.lut));
    HIP_TRY(c, c->d_tex_pixels.upload(A.tex_pixels)); HIP_TRY(c, c->d_tex_info.upload(A.tex_info));
    c->tex_info_host = A.tex_info;
    SceneDev &sd = c->sd;
    std::memset(&sd, 0, sizeof sd);
    sd.mesh_arena = c->d_mesh_arena.p;
    sd.materials = c->d_materials.p; sd.meshes = c->d_meshes.p; sd.grids = c->d_grids.p;
    sd.grid_cells = c->d_cells.p; sd.grid_lut = c->d_lut.p;
    sd.tex_pixels = c->d_tex_pixels.p; sd.tex_info = c->d_tex_info.p;
    c->has_grid = A.has_grid;
    int rc = install_objects(c, oh);
    if (rc != YCGE_OK) return rc;
    sd.ambient[0] = s->ambient_color.x; sd.ambient[1] = s->ambient_color.y; sd.ambient[2] = s->ambient_color.z;
    sd.ambient_intensity = s->ambient_intensity;
    sd.bg_top[0] = s->background_top.x; sd.bg_top[1] = s->background_top.y; sd.bg_top[2] = s->background_top.z;
    sd.bg_bottom[0] = s->background_bottom.x; sd.bg_bottom[1] = s->background_bottom.y; sd.bg_bottom[2] = s->background_bottom.z;
    sd.is_volume_scene = s->is_volume_scene ? 1 : 0;
    c->has_dynamic_textures = s->has_dynamic_textures != 0;
    sd.any_transparent = A.any_transparent ? 1 : 0;
    sd.any_textured = A.any_textured ? 1 : 0;
    sd.tl_offset = A.tl_offset;
    sd.anyhit_bfs = (uint32_t)c->knobs.bfs_rays;
    if (!c->dbg_counters.p) { HIP_TRY(c, c->dbg_counters.alloc(16 + 64 * 256)); HIP_TRY(c, hipMemset(c->dbg_counters.p, 0, (16 + 64 * 256) * sizeof(unsigned long long))); }
    sd.dbg_counters = c->dbg_counters.p;
    rc = upload_lights(c, s->lights, s->n_lights);
    if (rc != YCGE_OK) return rc;
    c->have_scene = true;
    c->block_order_valid = false; c->flight_order_frame[0] = c->flight_order_frame[1] = c->flight_order_frame[2] = -1;
    return YCGE_OK;
}

} // namespace ycge_host
extern "C" {
// A mesh's device records appended to the arena (32-byte units): depth-first - an internal node (GNode, both child boxes),
// then its left child's records (a leaf's triangle pair records or the whole left subtree), then the right child's - so a
// walk's next fetch is usually the next cache line.  Returns the root reference (ycge_device.h).
static uint32_t emit_mesh_records(const BuiltTree &t, const float *tris9, const int32_t *tri_material, int32_t material, int n_materials,
                                  std::vector<uint8_t> &arena, bool &bad_leaf, bool &bad_material)
{
    if (t.root < 0) return YCGE_REF_NONE_VALUE;
    std::function<uint32_t(int32_t)> emit = [&](int32_t ni) -> uint32_t {
        const RefNode &nd = t.nodes[(size_t)ni];
        const uint32_t unit = (uint32_t)(arena.size() / 32);
        if (nd.count > 0) {
            if (nd.count > 15) { bad_leaf = true; return YCGE_REF_NONE_VALUE; }
            const uint32_t n_rec = ((uint32_t)nd.count + 1u) / 2u;
            arena.resize(arena.size() + (size_t)n_rec * sizeof(GTriPair), 0);        // an odd leaf's last slot stays all zeros
            for (int32_t k = 0; k < nd.count; k++) {
                const int32_t ti = t.leaf_index[(size_t)(nd.start + k)];
                const float *v = tris9 + 9 * (size_t)ti;
                uint8_t *rec = arena.data() + (size_t)unit * 32 + (size_t)(k / 2) * sizeof(GTriPair);
                GTriPair g;
                std::memcpy(&g, rec, sizeof g);
                const int sl = k & 1;
                g.ax[sl] = v[0]; g.ay[sl] = v[1]; g.az[sl] = v[2];
                g.e1x[sl] = v[3] - v[0]; g.e1y[sl] = v[4] - v[1]; g.e1z[sl] = v[5] - v[2];        // MeshBVH.cs:87-91
                g.e2x[sl] = v[6] - v[0]; g.e2y[sl] = v[7] - v[1]; g.e2z[sl] = v[8] - v[2];
                g.orig[sl] = ti;
                g.material[sl] = tri_material ? tri_material[ti] : material;
                if (g.material[sl] < 0 || g.material[sl] >= n_materials) bad_material = true;
                std::memcpy(rec, &g, sizeof g);
            }
            return YCGE_REF(REF_MESH_LEAF, (unit << 4) | (uint32_t)nd.count);
        }
        arena.resize(arena.size() + sizeof(GNode), 0);
        GNode g;
        std::memset(&g, 0, sizeof g);
        const RefNode &L = t.nodes[(size_t)nd.left];
        const RefNode &R = t.nodes[(size_t)nd.right];
        g.lmin_x = L.mn[0]; g.lmin_y = L.mn[1]; g.lmin_z = L.mn[2]; g.lmax_x = L.mx[0]; g.lmax_y = L.mx[1]; g.lmax_z = L.mx[2];
        g.rmin_x = R.mn[0]; g.rmin_y = R.mn[1]; g.rmin_z = R.mn[2]; g.rmax_x = R.mx[0]; g.rmax_y = R.mx[1]; g.rmax_z = R.mx[2];
        g.lref = emit(nd.left);
        g.rref = emit(nd.right);
        std::memcpy(arena.data() + (size_t)unit * 32, &g, sizeof g);
        return YCGE_REF(REF_MESH_NODE, unit << 4);
    };
    return emit(t.root);
}


// Treelets of every internal mesh node (GTreeSlot, ycge_device.h), appended to the arena: a pure function of the records above.
// Returns the region's byte offset, 0 when there is nothing to build or 32-bit offsets would not reach its end.
static uint32_t append_treelets(std::vector<uint8_t> &arena, const std::vector<GMesh> &gmeshes)
{
    const size_t n_units = arena.size() / 32;
    const size_t t0 = (arena.size() + 511) & ~(size_t)511;
    const size_t total = t0 + n_units * YCGE_TL_BYTES_PER_UNIT;
    if (n_units == 0 || total + 4096 >= (1ull << 32)) return 0;
    bool any = false;
    for (const GMesh &m : gmeshes) any |= YCGE_REF_KIND(m.root_ref) == REF_MESH_NODE && m.root_ref != YCGE_REF_NONE_VALUE;
    if (!any) return 0;
    arena.resize(total, 0);
    auto node_at = [&](uint32_t ref) { GNode g; std::memcpy(&g, arena.data() + (size_t)((ref & 0x1ffffff0u) >> 4) * 32, sizeof g); return g; };
    std::vector<uint32_t> todo;
    for (const GMesh &m : gmeshes) if (m.root_ref != YCGE_REF_NONE_VALUE && YCGE_REF_KIND(m.root_ref) == REF_MESH_NODE) todo.push_back(m.root_ref);
    while (!todo.empty()) {
        const uint32_t ref = todo.back(); todo.pop_back();
        const uint32_t unit = (ref & 0x1ffffff0u) >> 4;
        GTreeSlot slots[YCGE_TL_SLOTS];
        std::memset(slots, 0, sizeof slots);
        // fill(b, parent record): slots 2b+2 / 2b+3 from the record of the node in slot b (b = -1: the treelet's own node)
        std::function<void(int, const GNode &, int)> fill = [&](int b, const GNode &g, int depth) {
            const int l = 2 * b + 2, r = l + 1;
            slots[l].mn[0] = g.lmin_x; slots[l].mn[1] = g.lmin_y; slots[l].mn[2] = g.lmin_z; slots[l].mx_x = g.lmax_x; slots[l].mx_y = g.lmax_y; slots[l].mx_z = g.lmax_z;
            slots[r].mn[0] = g.rmin_x; slots[r].mn[1] = g.rmin_y; slots[r].mn[2] = g.rmin_z; slots[r].mx_x = g.rmax_x; slots[r].mx_y = g.rmax_y; slots[r].mx_z = g.rmax_z;
            slots[l].ref = g.lref; slots[r].ref = g.rref; slots[l].valid = slots[r].valid = 1;
            if (depth < 3)
                for (int c : {l, r})
                    if (YCGE_REF_KIND(slots[c].ref) == REF_MESH_NODE) fill(c, node_at(slots[c].ref), depth + 1);
        };
        const GNode g = node_at(ref);
        fill(-1, g, 1);
        std::memcpy(arena.data() + t0 + (size_t)unit * YCGE_TL_BYTES_PER_UNIT, slots, sizeof slots);
        if (YCGE_REF_KIND(g.lref) == REF_MESH_NODE) todo.push_back(g.lref);
        if (YCGE_REF_KIND(g.rref) == REF_MESH_NODE) todo.push_back(g.rref);
    }
    return (uint32_t)t0;
}

int ycge_validate_scene(const ycge_scene *scene, char *msg, size_t msg_bytes)
try {
    std::string m;
    const int rc = validate_scene(scene, m);
    if (msg && msg_bytes) { std::strncpy(msg, m.c_str(), msg_bytes - 1); msg[msg_bytes - 1] = 0; }
    return rc;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

int ycge_scene_upload(ycge_ctx *c, const ycge_scene *s)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->parent) return c->fail(YCGE_ERR_INVALID_ARG, "peer contexts are driven by their root");
    {
        std::string m;
        const int vrc = validate_scene(s, m);
        if (vrc != YCGE_OK) return c->fail(vrc, "%s", m.c_str());
    }
    int qrc = quiesce(c);
    for (ycge_ctx *p : c->peers) if (qrc == YCGE_OK) qrc = quiesce(p);
    if (qrc != YCGE_OK) return qrc;
    HIP_TRY(c, hipSetDevice(c->device));
    c->have_scene = false;
    SceneArrays A;

    // ---- materials
    A.mats.assign(s->n_materials, GMaterial{});
    for (int i = 0; i < s->n_materials; i++) {
        const ycge_material &m = s->materials[i];
        GMaterial &g = A.mats[i];
        std::memset(&g, 0, sizeof g);
        g.kind = m.kind;
        g.albedo[0] = m.albedo.x; g.albedo[1] = m.albedo.y; g.albedo[2] = m.albedo.z;
        g.albedo_b[0] = m.albedo_b.x; g.albedo_b[1] = m.albedo_b.y; g.albedo_b[2] = m.albedo_b.z;
        g.checker_scale = m.checker_scale;
        g.reflectivity = m.reflectivity;
        g.emission[0] = m.emission.x; g.emission[1] = m.emission.y; g.emission[2] = m.emission.z;
        g.transparency = m.transparency; g.ior = m.index_of_refraction;
        g.trans_color[0] = m.transmission_color.x; g.trans_color[1] = m.transmission_color.y; g.trans_color[2] = m.transmission_color.z;
        if (m.transparency > 0.0f) A.any_transparent = true;
        g.tex = -1;
        if (m.kind == YCGE_MAT_TEXTURED && m.texture_weight > 0.0) {     // SampleAlbedo, RaytraceRenderer.cs:726-733 (a weight <= 0: the plain albedo)
            g.tex = m.texture;
            A.any_textured = true;
            g.tex_tiles = (float)(m.uv_scale > 1e-6 ? m.uv_scale : 1e-6);                                            // (float)Math.Max(1e-6, mat.UVScale)
            g.tex_t = (float)(m.texture_weight < 0.0 ? 0.0 : m.texture_weight > 1.0 ? 1.0 : m.texture_weight);       // (float)Math.Clamp(mat.TextureWeight, 0.0, 1.0)
        }
    }
    for (int i = 0; i < s->n_textures; i++) {
        const ycge_texture &t = s->textures[i];
        // info.w: 0 = static (RGBA32 ints); else bytes per pixel of a live texture's frames | flipU << 4 | flipV << 5, the frame's bytes packed into words
        const int bpp = t.frame_bytes_per_pixel;
        const int32_t info[4] = {(int32_t)A.tex_pixels.size(), t.width, t.height, bpp ? (bpp | (t.flip_u ? 16 : 0) | (t.flip_v ? 32 : 0)) : 0};
        A.tex_info.insert(A.tex_info.end(), info, info + 4);
        if (!bpp) A.tex_pixels.insert(A.tex_pixels.end(), t.pixels, t.pixels + (size_t)t.width * t.height);
        else {
            const size_t nbytes = (size_t)t.width * t.height * bpp, nwords = (nbytes + 3) / 4;
            const size_t at = A.tex_pixels.size();
            A.tex_pixels.resize(at + nwords, 0u);
            if (t.frame) std::memcpy(A.tex_pixels.data() + at, t.frame, nbytes);
        }
    }
    auto mat_ok = [&](int mi) { return mi >= 0 && mi < s->n_materials; };

    // ---- meshes: MeshBVH ctor (MeshBVH.cs:41-130) -> paired nodes + leaf-ordered triangles
    c->meshes.assign(s->n_meshes, MeshHost{});
    std::vector<uint8_t> &arena = A.arena;         // GNode (64 B) and GTriPair (96 B) records of every mesh, addressed in 32-byte units
    std::vector<GMesh> &gmeshes = A.gmeshes;
    gmeshes.assign(s->n_meshes, GMesh{});
    int max_mesh_depth = 0;
    for (int mi = 0; mi < s->n_meshes; mi++) {
        const ycge_mesh &m = s->meshes[mi];
        BoundsSoA items;
        triangle_items(m.triangles, m.n_triangles, items);
        BuiltTree &t = c->meshes[mi].tree;
        build_tree(items, TreeFlavour::Mesh, t);
        if (t.max_depth > 64) return c->fail(YCGE_ERR_STACK_DEPTH, "mesh %d: BVH depth %d exceeds the reference's 64-entry stack (MeshBVH.cs:150)", mi, t.max_depth);
        if (t.max_depth > max_mesh_depth) max_mesh_depth = t.max_depth;
        GMesh &gm = gmeshes[mi];
        std::memset(&gm, 0, sizeof gm);
        bool bad_material = false, bad_leaf = false;
        gm.root_ref = emit_mesh_records(t, m.triangles, m.tri_material, m.material, s->n_materials, arena, bad_leaf, bad_material);
        if (t.root >= 0) for (int a = 0; a < 3; a++) { gm.root_min[a] = t.nodes[t.root].mn[a]; gm.root_max[a] = t.nodes[t.root].mx[a]; }
        if (bad_leaf) return c->fail(YCGE_ERR_UNSUPPORTED, "mesh %d: a leaf of more than 15 triangles", mi);
        if (bad_material) return c->fail(YCGE_ERR_INVALID_ARG, "mesh %d: triangle material out of range", mi);
        if (arena.size() / 32 >= (1u << 25)) return c->fail(YCGE_ERR_UNSUPPORTED, "mesh records exceed the 1 GB arena");
    }

    A.tl_offset = c->knobs.no_coop ? 0u : append_treelets(arena, gmeshes);

    // ---- voxel grids: VolumeGrid ctor (VolumeGrid.cs:55-93), one byte per voxel = index into a per-grid material table
    std::vector<GGrid> &ggrids = A.ggrids;
    ggrids.assign(s->n_grids, GGrid{});
    std::vector<uint8_t> &cells = A.cells;
    std::vector<int32_t> &lut = A.lut;
    for (int gi = 0; gi < s->n_grids; gi++) {
        const ycge_grid &g = s->grids[gi];
        GGrid &G = ggrids[gi];
        std::memset(&G, 0, sizeof G);
        G.nx = g.nx; G.ny = g.ny; G.nz = g.nz;
        G.nbx = (g.nx + 7) >> 3; G.nby = (g.ny + 7) >> 3; G.nbz = (g.nz + 7) >> 3;
        G.min_corner[0] = g.min_corner.x; G.min_corner[1] = g.min_corner.y; G.min_corner[2] = g.min_corner.z;
        G.voxel_size[0] = cs_max(1e-6f, g.voxel_size.x); G.voxel_size[1] = cs_max(1e-6f, g.voxel_size.y); G.voxel_size[2] = cs_max(1e-6f, g.voxel_size.z);
        G.wireframe = g.wireframe ? 1 : 0;
        float ww = g.wire_width_fraction; if (ww < 0.0f) ww = 0.0f; if (ww > 0.5f) ww = 0.5f;
        G.wire_width_frac = ww;
        float wm = g.wire_max_distance; if (wm < 0.0f) wm = 0.0f;
        G.wire_max_distance = wm;
        const size_t cap = (size_t)G.nbx * G.nby * G.nbz * 512;
        const size_t off = (cells.size() + 255) & ~(size_t)255;
        if (off + cap >= ((size_t)1 << 32)) return c->fail(YCGE_ERR_UNSUPPORTED, "voxel storage exceeds 4 GiB");
        G.cell_offset = (uint32_t)off;
        cells.resize(off + cap, 0);
        G.lut_offset = (uint32_t)lut.size();
        // code 0 = empty; codes 1.. = distinct (matId, metaId) pairs in first-seen order
        std::vector<std::pair<int32_t, int32_t>> seen;
        lut.push_back(-1);
        const bool maskable = (size_t)G.nbx * G.nby * G.nbz <= 64;
        uint64_t brick_mask = 0;
        int lo[3] = {g.nx, g.ny, g.nz}, hi[3] = {-1, -1, -1};          // index box of the solid voxels
        for (int iz = 0; iz < g.nz; iz++)
            for (int iy = 0; iy < g.ny; iy++)
                for (int ix = 0; ix < g.nx; ix++) {
                    const size_t src = ((size_t)ix * g.ny + iy) * g.nz + iz;
                    const int32_t mat = g.cells[2 * src], meta = g.cells[2 * src + 1];
                    if (mat <= 0) continue;
                    int code = -1;
                    for (size_t k = 0; k < seen.size(); k++) if (seen[k].first == mat && seen[k].second == meta) { code = (int)k + 1; break; }
                    if (code < 0) {
                        if (seen.size() >= 255) return c->fail(YCGE_ERR_UNSUPPORTED, "grid %d: more than 255 distinct (matId, metaId) pairs", gi);
                        int material = g.default_material;
                        for (int k = 0; k < g.n_lookup; k++) if (g.lookup[k].mat_id == mat && g.lookup[k].meta_id == meta) { material = g.lookup[k].material; break; }
                        if (!mat_ok(material)) return c->fail(YCGE_ERR_INVALID_ARG, "grid %d: no material for (matId %d, metaId %d)", gi, mat, meta);
                        seen.push_back({mat, meta});
                        lut.push_back(material);
                        code = (int)seen.size();
                    }
                    const int brick = (((iz >> 3) * G.nby) + (iy >> 3)) * G.nbx + (ix >> 3);
                    cells[off + (size_t)brick * 512 + morton3(ix & 7, iy & 7, iz & 7)] = (uint8_t)code;
                    if (maskable) brick_mask |= (uint64_t)1 << brick;
                    if (ix < lo[0]) lo[0] = ix; if (ix > hi[0]) hi[0] = ix;
                    if (iy < lo[1]) lo[1] = iy; if (iy > hi[1]) hi[1] = iy;
                    if (iz < lo[2]) lo[2] = iz; if (iz > hi[2]) hi[2] = iz;
                }
        for (int a = 0; a < 3; a++) {           // one voxel of margin on every side (GGrid::solid_lo / solid_hi)
            G.solid_lo[a] = hi[a] < 0 ? 1.0f : G.min_corner[a] + (float)(lo[a] - 1) * G.voxel_size[a];
            G.solid_hi[a] = hi[a] < 0 ? 0.0f : G.min_corner[a] + (float)(hi[a] + 2) * G.voxel_size[a];
        }
        // How far along a ray the one-voxel margin of that box is provably enough.  The reference's walk reaches a cell by repeated
        // binary32 additions to tMax (VolumeGrid.cs:205-226): after k steps its t is off by at most k * ulp(t) / 2, i.e. the cell path
        // may drift k * t * 2^-24 world units from the true ray, with k <= nx + ny + nz steps inside one grid.  The cull (and the early
        // end of a walk at the box's exit) assumes that drift stays below HALF a voxel: t <= voxel * 2^23 / (nx + ny + nz) - 87 000 voxel
        // lengths for a 32^3 chunk; half of that is what is stored.  Beyond it the timed kernels walk the grid as the reference does.
        {
            const float vs = cs_min(G.voxel_size[0], cs_min(G.voxel_size[1], G.voxel_size[2]));
            G.cull_t_limit = vs * 4194304.0f / (float)(G.nx + G.ny + G.nz);
        }
        G.has_brick_mask = maskable ? 1 : 0;
        G.brick_mask_lo = (uint32_t)brick_mask; G.brick_mask_hi = (uint32_t)(brick_mask >> 32);
    }

    // ---- what the object / scene-BVH step needs (kept for ycge_scene_update_objects)
    c->gmeshes_host = gmeshes;
    c->n_materials = s->n_materials;
    c->max_mesh_depth = max_mesh_depth;
    c->materials_can_mirror = false;
    for (int i = 0; i < s->n_materials; i++) if (s->materials[i].reflectivity >= c->cfg.mirror_threshold) c->materials_can_mirror = true;
    c->grid_bounds.assign(s->n_grids, std::array<float, 6>{{0, 0, 0, -1, -1, -1}});
    for (int gi = 0; gi < s->n_grids; gi++) {                // VolumeGrid.TryGetBounds, VolumeGrid.cs:95-97
        const ycge_grid &g = s->grids[gi];
        if (g.nx <= 0 || g.ny <= 0 || g.nz <= 0) continue;
        const float vs[3] = {cs_max(1e-6f, g.voxel_size.x), cs_max(1e-6f, g.voxel_size.y), cs_max(1e-6f, g.voxel_size.z)};
        c->grid_bounds[gi] = {{g.min_corner.x, g.min_corner.y, g.min_corner.z, g.min_corner.x + (float)g.nx * vs[0],
                               g.min_corner.y + (float)g.ny * vs[1], g.min_corner.z + (float)g.nz * vs[2]}};
    }
    c->grid_solid.resize(s->n_grids);
    for (int gi = 0; gi < s->n_grids; gi++)
    {
        for (int a = 0; a < 3; a++) { c->grid_solid[gi][a] = A.ggrids[gi].solid_lo[a]; c->grid_solid[gi][3 + a] = A.ggrids[gi].solid_hi[a]; }
        c->grid_solid[gi][6] = A.ggrids[gi].cull_t_limit;
    }
    A.has_grid = s->n_grids > 0;

    // ---- Scene.Objects + scene BVH, then every device gets the same arrays
    ObjectsHost oh;
    int rc = build_objects(c, s->prims, s->n_prims, oh);
    if (rc != YCGE_OK) return rc;
    rc = install_scene(c, A, oh, s);
    for (ycge_ctx *p : c->peers) {
        if (rc != YCGE_OK) break;
        p->max_mesh_depth = c->max_mesh_depth; p->n_materials = c->n_materials; p->materials_can_mirror = c->materials_can_mirror;
        rc = install_scene(p, A, oh, s);
        if (rc != YCGE_OK) c->err = p->err;
    }
    (void)hipSetDevice(c->device);
    return rc;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_scene_update_lights(ycge_ctx *c, const ycge_light *lights, int32_t n_lights, const ycge_vec3 *ambient_color,
                             float ambient_intensity, const ycge_vec3 *top, const ycge_vec3 *bottom)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "no scene uploaded");
    if (n_lights < 0 || (n_lights > 0 && !lights)) return c->fail(YCGE_ERR_INVALID_ARG, "bad light array");
    int rc = quiesce(c);
    if (rc != YCGE_OK) return rc;
    rc = upload_lights(c, lights, n_lights);
    if (rc != YCGE_OK) return rc;
    if (ambient_color) { c->sd.ambient[0] = ambient_color->x; c->sd.ambient[1] = ambient_color->y; c->sd.ambient[2] = ambient_color->z; c->sd.ambient_intensity = ambient_intensity; }
    if (top) { c->sd.bg_top[0] = top->x; c->sd.bg_top[1] = top->y; c->sd.bg_top[2] = top->z; }
    if (bottom) { c->sd.bg_bottom[0] = bottom->x; c->sd.bg_bottom[1] = bottom->y; c->sd.bg_bottom[2] = bottom->z; }
    for (ycge_ctx *p : c->peers) {
        rc = ycge_scene_update_lights(p, lights, n_lights, ambient_color, ambient_intensity, top, bottom);
        if (rc != YCGE_OK) { c->err = p->err; break; }
    }
    (void)hipSetDevice(c->device);
    return rc;
}
catch (...) { return ycge_host::abi_catch(c); }

// The next frame of a live texture (Renderer/Texture.cs:113-116: SampleBilinear reads IFrameReader.GetCurrentFramePtr())
int ycge_scene_update_texture(ycge_ctx *c, int32_t texture_index, const uint8_t *frame, size_t bytes)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->parent) return c->fail(YCGE_ERR_INVALID_ARG, "peer contexts are driven by their root");
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "no scene uploaded");
    if (texture_index < 0 || (size_t)texture_index * 4 + 3 >= c->tex_info_host.size()) return c->fail(YCGE_ERR_INVALID_ARG, "texture index %d out of range", texture_index);
    const int32_t *info = c->tex_info_host.data() + (size_t)texture_index * 4;
    const int bpp = info[3] & 15;
    if (bpp == 0) return c->fail(YCGE_ERR_INVALID_ARG, "texture %d is static: upload the scene again to change it", texture_index);
    if (!frame || bytes != (size_t)info[1] * info[2] * bpp) return c->fail(YCGE_ERR_INVALID_ARG, "texture %d: a frame is %d x %d x %d bytes", texture_index, info[1], info[2], bpp);
    int rc = YCGE_OK;
    if (c->cfg.world_size == 1 && c->peers.empty() && (!c->last_stream || c->last_stream == c->stream)) {
        // Single device, no tiled call on a stream of the caller's: the traces of this context run on its own stream and - frames in flight of
        // the stage pipeline (render_frame_in_flight: odd frames from 4 096 tiles on) or of YCGE_PATH=m - on its second trace stream.  The
        // copy is queued on the first and ORDERED against the second both ways: it waits for what the second stream holds now (the trace that
        // still samples the old frame), and the second stream's next trace waits for it.  No device-wide wait, frames in flight stay in flight.
        // The caller's array is its own again when this returns: the frame is staged in page-locked memory first.
        HIP_TRY(c, hipSetDevice(c->device));
        if (c->stream2) {
            if (!c->tex_order_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->tex_order_ev, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(c->tex_order_ev, c->stream2));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tex_order_ev, 0));
        }
        const int k = c->tex_stage_next; c->tex_stage_next ^= 1;
        if (c->tex_stage_busy[k]) { HIP_TRY(c, hipEventSynchronize(c->tex_stage_ev[k])); c->tex_stage_busy[k] = false; }      // (the copy of two updates ago)
        if (c->tex_stage_bytes[k] < bytes) {
            if (c->tex_stage[k]) { (void)hipHostFree(c->tex_stage[k]); c->tex_stage[k] = nullptr; c->tex_stage_bytes[k] = 0; }
            HIP_TRY(c, hipHostMalloc((void **)&c->tex_stage[k], bytes, hipHostMallocDefault));
            c->tex_stage_bytes[k] = bytes;
        }
        if (!c->tex_stage_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->tex_stage_ev[k], hipEventDisableTiming));
        std::memcpy(c->tex_stage[k], frame, bytes);
        HIP_TRY(c, hipMemcpyAsync(c->d_tex_pixels.p + info[0], c->tex_stage[k], bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->tex_stage_ev[k], c->stream));
        if (c->stream2) HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->tex_stage_ev[k], 0));
        c->tex_stage_busy[k] = true;
        return YCGE_OK;
    }
    // several devices / a rank of a tiled frame / tiled calls on the caller's own streams: the scene changes while nothing runs
    rc = quiesce(c);
    if (rc != YCGE_OK) return rc;
    HIP_TRY(c, hipMemcpy(c->d_tex_pixels.p + info[0], frame, bytes, hipMemcpyHostToDevice));
    for (ycge_ctx *p : c->peers) {
        if (hipSetDevice(p->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(p->d_tex_pixels.p + info[0], frame, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); rc = c->fail(YCGE_ERR_DEVICE, "texture frame copy failed on device %d", p->device); break; }
    }
    (void)hipSetDevice(c->device);
    return rc;
}
catch (...) { return ycge_host::abi_catch(c); }

// Scene.Update -> RebuildBVH when an entity moved (Scenes/Scene.cs:122-127, e.g. BobbingSphereEntity,
// TestScenesRandom.cs:708-714): new Scene.Objects records against the meshes, grids and materials of the last
// ycge_scene_upload.  Only the scene-level BVH is rebuilt (as in the reference: a Mesh keeps its own BVH).
int ycge_scene_update_objects(ycge_ctx *c, const ycge_prim *prims, int32_t n_prims)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->parent) return c->fail(YCGE_ERR_INVALID_ARG, "peer contexts are driven by their root");
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "no scene uploaded");
    if (n_prims < 0 || (n_prims > 0 && !prims)) return c->fail(YCGE_ERR_INVALID_ARG, "bad object array");
    ObjectsHost oh;
    BoundsSoA items;
    int rc = flatten_objects(c, prims, n_prims, oh, items);       // host work first: the devices keep rendering the old objects meanwhile
    if (rc != YCGE_OK) return rc;
    // the tree itself is built on the device (ycge_bvh_build.hip) - on the host only for what that kernel does not take: no objects,
    // more than YCGE_BVH_DEV_MAX_ITEMS, or (reported by the kernel) a tree deeper than the reference's stack allows
    // ... and fewer objects than the measured crossover: one CU building a small tree loses to the host builder (300 objects: 168 us
    // against 105; config 5's 976: 559 against 354; 2 300: 687 against 1 533 - profiles/r02/f2_update_objects_timing.txt)
    bool on_device = !c->knobs.scene_bvh_host && n_prims >= 1 && n_prims >= c->knobs.scene_bvh_device_min && n_prims <= YCGE_BVH_DEV_MAX_ITEMS;
    if (!on_device) { rc = build_scene_tree_host(c, items, oh); if (rc != YCGE_OK) return rc; }
    rc = quiesce(c);
    for (ycge_ctx *p : c->peers) if (rc == YCGE_OK) rc = quiesce(p);
    if (rc != YCGE_OK) return rc;
    c->have_scene = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (on_device) {
        rc = install_objects_device_built(c, c, oh, items);
        for (ycge_ctx *p : c->peers) {
            if (rc != YCGE_OK) break;
            rc = install_objects_device_built(p, c, oh, items);       // the same kernel on the same items: the same tree
            if (rc < 0) c->err = p->err;
        }
        if (rc == 1) {
            c->bvh_host_fallbacks++;
            on_device = false;
            rc = build_scene_tree_host(c, items, oh);
        } else if (rc == YCGE_OK) c->bvh_device_builds++;
    }
    if (!on_device && rc == YCGE_OK) {
        rc = install_objects(c, oh);
        for (ycge_ctx *p : c->peers) {
            if (rc != YCGE_OK) break;
            rc = install_objects(p, oh);
            if (rc != YCGE_OK) c->err = p->err;
        }
    }
    c->bvh_last_build_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    (void)hipSetDevice(c->device);
    if (rc != YCGE_OK) return rc;
    c->have_scene = true;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// profiling aid: the progress lines of the persistent in-place A-trous launch (32 words per band: [0] progress, [4..7] begin / end
// timestamps at 100 MHz, [8] passes) of the last frame with a post stage
int ycge_debug_read_post_progress(ycge_ctx *c, uint32_t *dst, size_t n_words)
try {
    if (!c || !dst || !c->post_progress.p || n_words > c->post_progress.n) return YCGE_ERR_INVALID_ARG;
    HIP_TRY(c, hipDeviceSynchronize());
    return copy_out(c, dst, c->post_progress.p, n_words * 4);
}
catch (...) { return ycge_host::abi_catch(c); }
// test / profiling hook: {device builds, host rebuilds after the kernel declined (a tree deeper than the reference's stack), host builds,
// microseconds of the last update's build + install, Array.Sort cases in the current tree (BVH.cs:389,419), depth of the current tree}
int ycge_debug_scene_bvh_stats(ycge_ctx *c, int64_t *out6)
try {
    if (!c || !out6) return YCGE_ERR_INVALID_ARG;
    out6[0] = c->bvh_device_builds; out6[1] = c->bvh_host_fallbacks; out6[2] = c->bvh_host_builds; out6[3] = (int64_t)c->bvh_last_build_us;
    out6[4] = c->scene_tree.sort_fallbacks; out6[5] = c->scene_tree.max_depth;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_resize(ycge_ctx *c, int32_t fbw, int32_t fbh, int32_t ss)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    int rc = quiesce(c);
    if (rc != YCGE_OK) return rc;
    rc = set_geometry(c, fbw, fbh, ss);
    for (ycge_ctx *p : c->peers) {
        if (rc != YCGE_OK) break;
        rc = ycge_resize(p, fbw, fbh, ss);
        if (rc != YCGE_OK) c->err = p->err;
    }
    (void)hipSetDevice(c->device);
    return rc;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_set_camera(ycge_ctx *c, const float pos[3], float yaw, float pitch, float fov_deg)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!pos) return c->fail(YCGE_ERR_INVALID_ARG, "null position");
    std::lock_guard<std::mutex> g(c->cam_lock);
    c->cam_pos[0] = pos[0]; c->cam_pos[1] = pos[1]; c->cam_pos[2] = pos[2];
    c->yaw = yaw; c->pitch = pitch; c->fov_deg = fov_deg;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_set_frame_counter(ycge_ctx *c, int64_t fc)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (fc < 0 || fc == INT64_MAX) return c->fail(YCGE_ERR_INVALID_ARG, "frame counter must be in [0, 2^63 - 2] (the reference's counter starts at 0 and only grows, RaytraceRenderer.cs:24,175)");
    c->frame_counter = fc;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// profiling builds: the 6 sums of the cooperative walk's statistics (ycge_coop.hip.h, -DYCGE_DBG_COOPSTAT), cumulative
int ycge_debug_read_coop_stats(ycge_ctx *c, uint64_t out[16])
try {
    if (!c || !out || !c->dbg_counters.p) return YCGE_ERR_INVALID_ARG;
    std::vector<unsigned long long> v(16 + 16 * 256);
    HIP_TRY(c, hipDeviceSynchronize());
    { const int rc2 = copy_out(c, v.data(), c->dbg_counters.p, v.size() * sizeof(unsigned long long)); if (rc2 != YCGE_OK) return rc2; }
    for (int k = 0; k < 16; k++) { out[k] = 0; for (int i = 0; i < 256; i++) out[k] += v[16 + 8 * 256 * (k >> 3) + 8 * i + (k & 7)]; }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// profiling builds (-DYCGE_DBG_BATCHSTAT, mesh_walk): 8 banks of 8 sums, cumulative; banks 2-4 every batch by kind (occlusion / closest hit / mixed),
// 5-7 the batches of >= 48 iterations
int ycge_debug_read_batch_stats(ycge_ctx *c, uint64_t out[64])
try {
    if (!c || !out || !c->dbg_counters.p) return YCGE_ERR_INVALID_ARG;
    std::vector<unsigned long long> v(16 + 64 * 256);
    HIP_TRY(c, hipDeviceSynchronize());
    { const int rc2 = copy_out(c, v.data(), c->dbg_counters.p, v.size() * sizeof(unsigned long long)); if (rc2 != YCGE_OK) return rc2; }
    for (int k = 0; k < 64; k++) { out[k] = 0; for (int i = 0; i < 256; i++) out[k] += v[16 + 8 * 256 * (k >> 3) + 8 * i + (k & 7)]; }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// test hook: the scene nodes as the device holds them and SceneDev::walk_nodes (64 B each; the walk tree has 1 + 2 * YCGE_WALK_LEAF_NODES
// entries per scene node) + grid_owner; returns the scene node count (0: no walk tree), < 0 on an error
int ycge_debug_read_walk_tree(ycge_ctx *c, void *gnodes_out, void *walk_out, int32_t capacity_nodes, int32_t *grid_owner_out, int32_t n_grids, uint32_t *root_and_limit_out)
try {
    if (!c || !c->have_scene) return YCGE_ERR_INVALID_ARG;
    const int n = c->walk_scene_nodes;
    if (n == 0) return 0;
    if (!gnodes_out || !walk_out || capacity_nodes < n || !grid_owner_out || n_grids != (int)c->grid_solid.size() || !root_and_limit_out) return YCGE_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    { int rc2 = copy_out(c, gnodes_out, c->d_scene_nodes.p, (size_t)n * sizeof(GNode)); if (rc2 != YCGE_OK) return rc2;
      rc2 = copy_out(c, walk_out, c->d_walk_nodes.p, (size_t)n * (1 + 2 * YCGE_WALK_LEAF_NODES) * sizeof(GNode)); if (rc2 != YCGE_OK) return rc2;
      rc2 = copy_out(c, grid_owner_out, c->d_grid_owner.p, (size_t)n_grids * sizeof(int32_t)); if (rc2 != YCGE_OK) return rc2; }
    root_and_limit_out[0] = c->sd.walk_root_ref; std::memcpy(&root_and_limit_out[1], &c->sd.walk_t_limit, 4);
    return n;
}
catch (...) { return ycge_host::abi_catch(c); }

// Page-locked host memory for the SDR frame.  hipHostRegister works on whole pages: two registered heap arrays that share a boundary page
// lose it when ONE of them is unregistered, and the other's next read-back is a device write to an unmapped host page ("Memory access fault
// by GPU" at a heap address, round 4).  So the library refuses a range that is not whole pages of its own - the caller proves it owns the
// pages by handing over page-aligned memory - and offers memory that is (ycge_alloc_host_buffer: hipHostMalloc).
size_t ycge_host_page_size(void)
try {
    const long p = sysconf(_SC_PAGESIZE);
    return p > 0 ? (size_t)p : (size_t)4096;
}
catch (...) { (void)ycge_host::abi_catch(nullptr); return 0; }
int ycge_pin_host_buffer(void *buffer, size_t bytes)
try {
    const size_t page = ycge_host_page_size();
    if (!buffer || bytes == 0 || ((uintptr_t)buffer % page) != 0 || (bytes % page) != 0) return YCGE_ERR_INVALID_ARG;
    if (hipHostRegister(buffer, bytes, hipHostRegisterDefault) == hipSuccess) return YCGE_OK;
    (void)hipGetLastError();
    return YCGE_ERR_DEVICE;
}
catch (...) { return ycge_host::abi_catch(nullptr); }
int ycge_unpin_host_buffer(void *buffer)
try {
    if (!buffer || ((uintptr_t)buffer % ycge_host_page_size()) != 0) return YCGE_ERR_INVALID_ARG;
    if (hipHostUnregister(buffer) == hipSuccess) return YCGE_OK;
    (void)hipGetLastError();
    return YCGE_ERR_DEVICE;
}
catch (...) { return ycge_host::abi_catch(nullptr); }
int ycge_alloc_host_buffer(size_t bytes, void **out)
try {
    if (!out) return YCGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (bytes == 0) return YCGE_ERR_INVALID_ARG;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return YCGE_ERR_OUT_OF_MEMORY; }
    std::memset(p, 0, bytes);
    *out = p;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(nullptr); }
int ycge_free_host_buffer(void *buffer)
try {
    if (!buffer) return YCGE_OK;
    if (hipHostFree(buffer) == hipSuccess) return YCGE_OK;
    (void)hipGetLastError();
    return YCGE_ERR_INVALID_ARG;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

int ycge_device_count(void)
try {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : -1;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

int ycge_read_timed_steps(ycge_ctx *c, uint64_t *lane_steps)
try {
    if (!c || !lane_steps) return YCGE_ERR_INVALID_ARG;
    unsigned long long total = 0;
    std::vector<ycge_ctx *> all{c};
    all.insert(all.end(), c->peers.begin(), c->peers.end());
    for (ycge_ctx *d : all) {
        std::vector<unsigned long long> v(YCGE_COUNTER_WORDS);
        if (hipSetDevice(d->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            copy_out(d, v.data(), d->counters.p, v.size() * sizeof(unsigned long long)) != YCGE_OK) { (void)hipSetDevice(c->device); return c->fail(YCGE_ERR_DEVICE, "timed-step read-back failed on device %d", d->device); }
        for (size_t i = 8; i < v.size(); i += 8) total += v[i];
    }
    (void)hipSetDevice(c->device);
    *lane_steps = total;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_tile_slab_bytes(const ycge_ctx *c, size_t *bytes)
try {
    if (!c || !bytes) return YCGE_ERR_INVALID_ARG;
    *bytes = (size_t)c->tiles_per_rank_padded * 256 * slab_floats(c) * sizeof(float);
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

} // extern "C"

namespace ycge_host {

// steps 1-3 of TryFlipAndBlit (RaytraceRenderer.cs:159-176): camera snapshot under the lock, frame = ++frameCounter.
// The reset decision (step 2) compares this pose with the camera the LAST RESOLVED frame committed, so it is taken where
// the frame is resolved (taa_and_commit) - in a pipelined caller the previous frame may not have been resolved yet.
void snapshot_frame(ycge_ctx *c, FrameState &fs)
{
    {
        std::lock_guard<std::mutex> g(c->cam_lock);
        fs.pos[0] = c->cam_pos[0]; fs.pos[1] = c->cam_pos[1]; fs.pos[2] = c->cam_pos[2];
        fs.yaw = c->yaw; fs.pitch = c->pitch; fs.fov = c->fov_deg;
    }
    fs.reset = false;
    fs.frame = ++c->frame_counter;
    fs.fan_blocks = 0;
}

// How the longest-first schedule cuts blocks into parts.  policy, octal: digit c = log2(parts) a block of class c
// (policy_class_of_order_class) is split into, class 7 leftmost.  A whole frame splits by rank instead (split_top: a per-class split
// costs more slots than it saves there).  On a rank's tiles slots are plentiful, thin wavefronts step faster and
// see a smaller maximum over their lanes, so the heavier classes are split, deeper the fewer blocks a rank holds.  Per-rank trace on
// config 4, maximum over ranks (profiles/rank_times.py): 8 ranks 0.359 -> 0.289 ms, 4 ranks 0.381 -> 0.351, 2 ranks 0.485 -> 0.441;
// one class deeper loses at every rank count (8 ranks 55543000: 0.356; 4 ranks 44432000: 0.420; 2 ranks 44320000: 0.471).
// With three or more traces of the tile-resident ring in flight the slots are no longer plentiful - the traces fill each other's tails - and
// the shallow cut wins at every rank count (8 ranks, ring of 4, one-GPU emulation: 033220000 0.147-0.172 ms a rank-frame, 044433000
// 0.186-0.20, unsplit 0.17-0.23; profiles/r04/g_resident_ring_emulation.txt).
// BATCHED launches (n frames of a rank's tiles per launch, two launches in flight) are throughput-bound: two overlapping launches of four
// eighth-frames are a whole frame's blocks and took 0.84 ms against the whole frame's 0.48 - the split's thin wavefronts (8 or 4 parts of a
// block's 64 pixels) are slot time nobody gets back.  Unsplit: 8 ranks, batches of 4: 0.100-0.108 -> 0.086-0.088 ms a rank-frame
// (011110000 0.087-0.093, 022110000 0.088-0.095, 022220000 0.101-0.112; profiles/r05/e_split_policy_by_form.txt).
void schedule_policy(const ycge_ctx *c, uint32_t &policy, uint32_t &split_top, int resident_ring, bool batched)
{
    const uint32_t world_policy = batched ? 0u : (resident_ring >= 3 && c->cfg.world_size >= 2) ? 033220000u
                                : c->cfg.world_size >= 8 ? 044433000u : c->cfg.world_size >= 4 ? 044320000u : c->cfg.world_size >= 2 ? 033220000u : 0u;
    policy = c->knobs.split_set ? c->knobs.split_policy : world_policy;
    // ... or, on a whole frame, the split_top blocks at the head of the schedule whatever their class (k_cost_scatter)
    split_top = (c->knobs.split_set || policy || batched || c->knobs.split_top <= 0) ? 0u : ((uint32_t)c->knobs.split_top & 0xffffu) | ((uint32_t)c->knobs.split_top_lg << 16);
    // Round 6: in TWO parts of 32 pixels, more blocks of them where the frame leaves the machine room.  A part is a wavefront slot for the
    // length of its chain (~0.7 of the block's in two parts, ~0.6 in four) and the bulk of a 1080p frame fills the slots to within 15 %
    // (slot time 0.39 of 0.48 ms): config 4 (32 400 blocks) 32 x 4 parts 0.4765 ms, 64 x 2 0.4716, 128 x 2 0.4757, 256 x 2 0.4848, 512 x 2 0.4969;
    // config 3 (14 400 blocks) 32 x 4 0.2781, 64 x 2 0.2769, 128 x 2 0.2716, 256 x 2 0.2671, 512 x 2 0.2693 (same call, profiles/r06/c_split_in_two.txt).
    // YCGE_SPLIT_TOP / YCGE_SPLIT_TOP_LG override.
    if (split_top && !c->knobs.split_top_set) {
        const uint32_t n_blocks = (uint32_t)c->n_owned * 4u;
        split_top = (n_blocks > 24000u ? 64u : 256u) | (1u << 16);
    }
}

// (a scene with a textured material takes the generic kernels: the flat ones - configs 3 and 4 - are compiled without the texture
// branch, which cost them 1.6 % when it was merely present, profiles/tex_ab.sh)
int scene_is_flat(const ycge_ctx *c)
{
    return (YCGE_REF_KIND(c->sd.scene_root_ref) == REF_SCENE_LEAF && !c->knobs.generic_walk && !c->sd.any_textured) ? 1 : 0;      // YCGE_GENERIC_WALK: experiment knob, same pixels
}
bool frame_is_single_launch(const ycge_ctx *c)
{
    // (auto: one-leaf scenes - the mesh viewers - and scenes of analytic objects only: a few dozen objects under a shallow tree are bound by
    // the latency of a pixel's chain of queries, not by throughput, and one launch lets the chains overlap - config 1's scene at 80x45 ..
    // 960x270 consoles: 0.080 / 0.083 / 0.106 / 0.199 ms against 0.101 / 0.110 / 0.145 / 0.218 as stages, profiles/r05)
    return c->sd.any_transparent || c->knobs.path_policy == 2 || (c->knobs.path_policy == 0 && (scene_is_flat(c) || c->sd.analytic_only));
}

// step 4 (RaytraceRenderer.cs:183-216): ray-gen + trace of this context's tiles for the frame `fs`
// (ResidentTarget - where a trace of the tile-resident form writes and which schedule it follows - is trace_frame's last argument: ycge_ctx.h)
int trace_frame(ycge_ctx *c, float *d_slab, hipStream_t stream, FrameState &fs, bool timed, hipEvent_t launch_begin, hipEvent_t launch_end /* frames in flight: around the trace launches alone */,
                const ResidentTarget *rt)
{
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "Scene BVH not built; call ycge_scene_upload first (Scene.cs:73)");
    FrameParams P;
    fill_frame_params(c, P, fs.frame, fs.pos, fs.yaw, fs.pitch, fs.fov);
    TraceOut O;
    std::memset(&O, 0, sizeof O);
    O.current_hdr = c->current_hdr.p; O.g_albedo = c->g_albedo.p; O.g_normal = c->g_normal.p; O.g_depth = c->g_depth.p; O.sky = c->sky.p;
    const bool slab = d_slab != nullptr;
    if (slab) {
        const size_t npx = (size_t)c->hiW * c->hiH;
        if (!c->t_hdr.p) {
            HIP_TRY(c, c->t_hdr.alloc(3 * npx)); HIP_TRY(c, c->t_albedo.alloc(3 * npx)); HIP_TRY(c, c->t_normal.alloc(3 * npx));
            HIP_TRY(c, c->t_depth.alloc(npx)); HIP_TRY(c, c->t_sky.alloc(npx));
        }
        O.current_hdr = c->t_hdr.p; O.g_albedo = c->t_albedo.p; O.g_normal = c->t_normal.p; O.g_depth = c->t_depth.p; O.sky = c->t_sky.p;
        // Two tiled traces at a time: a caller that queues frame N + 1 on another stream than frame N (bench.py alternates two) lets
        // the bulk of one fill the wavefront slots the tail of the other leaves empty - on a rank's share of a frame, where the
        // heaviest chains ARE the launch, that is most of the machine.  So what a trace writes or scratches exists per frame parity
        // (trace outputs: the second set is the one the frames in flight use on a single-device context, never both; stack spill
        // area), every tiled trace first waits for the trace of frame N - 2, whatever streams the caller uses, and a scene whose trace
        // shares more than that between frames (stage pipeline queues, refraction stacks) also for frame N - 1.
        const int par = (int)(fs.frame & 1);
        for (int k = 0; k < 2; k++) if (!c->tile_trace_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->tile_trace_ev[k], hipEventDisableTiming));
        if (par) {
            if (!c->alt_hdr.p) {
                HIP_TRY(c, c->alt_hdr.alloc(3 * npx)); HIP_TRY(c, c->alt_albedo.alloc(3 * npx)); HIP_TRY(c, c->alt_normal.alloc(3 * npx));
                HIP_TRY(c, c->alt_depth.alloc(npx)); HIP_TRY(c, c->alt_sky.alloc(npx));
            }
            O.current_hdr = c->alt_hdr.p; O.g_albedo = c->alt_albedo.p; O.g_normal = c->alt_normal.p; O.g_depth = c->alt_depth.p; O.sky = c->alt_sky.p;
            if (!c->stack_spill2.p) HIP_TRY(c, c->stack_spill2.alloc(c->stack_spill.n));
        }
        if (c->tile_trace_used[par]) HIP_TRY(c, hipStreamWaitEvent(stream, c->tile_trace_ev[par], 0));
        if (c->tile_trace_used[par ^ 1] && (!frame_is_single_launch(c) || c->sd.any_transparent)) HIP_TRY(c, hipStreamWaitEvent(stream, c->tile_trace_ev[par ^ 1], 0));
    }
    if (rt) {       // tile-resident form: the frame set of the ring (the albedo plane is not kept: the frame ends with TAA)
        O.current_hdr = rt->set->hdr.p; O.g_normal = rt->set->normal.p; O.g_depth = rt->set->depth.p; O.sky = rt->set->sky.p;
        O.g_albedo = c->g_albedo.p;          // (written, never read in this form: one shared plane)
    }
    const bool debug = c->cfg.capture_debug && !slab && !rt;
    if (debug) { O.rays = c->dbg_rays.p; O.prim_id = c->dbg_prim.p; O.sub_id = c->dbg_sub.p; O.hit_t = c->dbg_hit_t.p; O.rng_state = c->dbg_rng.p; }
    if (c->knobs.wave_prof_stage >= 0) { if (!c->wave_prof.p) HIP_TRY(c, c->wave_prof.alloc((size_t)c->n_tiles * 16)); O.wave_prof = c->wave_prof.p; O.wave_prof_stage = c->knobs.wave_prof_stage; }
    O.counters = c->counters.p;
    if (c->in_flight_call && c->placed_flag && c->placed_next) { O.placed_flag = c->placed_flag; O.placed_value = c->placed_next; }
    if (c->cfg.count_work) HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, 6 * sizeof(unsigned long long), stream));
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[0], stream));
    int e;
    O.stack_spill = rt ? rt->set->spill.p : c->spill_override ? c->spill_override : (slab && (fs.frame & 1)) ? c->stack_spill2.p : c->stack_spill.p;
    const uint32_t trace_lanes = (uint32_t)(c->n_owned > 0 ? c->n_owned : 1) * 256u * YCGE_SCHEDULE_SLACK;
    O.stack_lanes = trace_lanes + c->fan_cap * 192u;
    O.path_stack = c->path_stack.p;
    const int flat = scene_is_flat(c);
    // Path choice.  Refraction splits need TraceFull's per-pixel LIFO -> single launch.  Otherwise: scenes
    // that are one BVH leaf (mesh viewers) are bounded by the latency chain of their few heaviest tiles, and
    // one launch lets the chains of all stages overlap (measured 0.85 vs 1.24 ms on config 4); scenes with a
    // real top-level tree (voxel worlds) are throughput-bound and run 1.7x faster as occupancy-friendly stages.
    const bool single_launch = frame_is_single_launch(c);
    if (single_launch) {
        // TraceFull's per-pixel LIFO is only ever touched at a refractive hit: scenes without transparent materials get none (0.5 GB at 1080p)
        if (!c->path_stack.p && c->sd.any_transparent) HIP_TRY(c, c->path_stack.alloc((size_t)3 * 11 * O.stack_lanes));
        O.path_stack = c->path_stack.p;
        if (rt) {       // the ring's own cost slots and schedule buffers (ycge_trace_tiles_resident queues the schedule builds)
            O.block_cost = c->knobs.no_lpt ? nullptr : rt->cost;
            O.block_order = c->knobs.no_lpt ? nullptr : rt->order;
            O.n_order = rt->n_order;
            fs.scheduled = O.block_order != nullptr;
            if (c->batch_collect) { c->batch_P.push_back(P); c->batch_O.push_back(O); return YCGE_OK; }      // (ycge_trace_tiles_resident_batch launches the frames of a batch together)
            if (launch_begin) HIP_TRY(c, hipEventRecord(launch_begin, stream));
            e = ycge_launch_trace(&c->sd, &P, &O, c->cfg.count_work, flat, 0, stream);
            if (launch_end) HIP_TRY(c, hipEventRecord(launch_end, stream));
            if (e != 0) return c->fail(YCGE_ERR_DEVICE, "trace launch failed: %s", hipGetErrorString((hipError_t)e));
            if (timed) HIP_TRY(c, hipEventRecord(c->ev[1], stream));
            return YCGE_OK;
        }
        const uint32_t n_blocks = (uint32_t)c->n_owned * 4u;
        // Longest first only where there is a "first": a frame whose 8 x 8 blocks are all resident at once (one wavefront each; 4 per SIMD
        // for the kernels without a mesh walk, 3 with) is placed whole whatever the order, and the schedule - two kernels, a memset and
        // two stream hops behind every trace - is then a fifth of a small frame: config 2 (3 600 blocks) 0.094 -> see DESIGN section 8
        // (YCGE_LPT_ALWAYS=1: build it anyway).
        const uint32_t resident_blocks = (uint32_t)c->compute_units * 4u * (c->meshes.empty() ? 4u : 3u);
        const bool lpt = !c->knobs.no_lpt && (n_blocks > resident_blocks || c->knobs.lpt_always);
        const uint32_t cost_slot = (uint32_t)((uint64_t)fs.frame % YCGE_COST_FRAMES);       // this frame's array of the cost ring
        O.block_cost = lpt ? c->block_cost.p + (size_t)cost_slot * n_blocks : nullptr;
        const int fk = (int)(fs.frame & 1);
        const bool flight = c->in_flight_call && lpt;
        // A rank's share of a tiled frame (ycge_trace_tiles): callers queue the trace of frame N + 1 before frame N is gathered and
        // resolved (bench.py's two streams), and then the schedule built BETWEEN the two traces is all that stands between them
        // (34 us of kernels + two cross-stream hops against a rank's 0.29 ms at 8 ranks).  So a tiled frame's trace is followed by the
        // schedule of frame N + 2 - it leaves out the cost slot frame N + 1's trace may be writing and clears frame N + 2's - and
        // frame N + 1 runs with the order built behind frame N - 1.  The same buffers and rules as the frames in flight.
        const bool deferred = slab && lpt && !flight;
        fs.scheduled = lpt;
        fs.single_launch = true;
        uint32_t policy, split_top;
        schedule_policy(c, policy, split_top);
        if (deferred) {
            for (int k = 0; k < 2; k++) {
                if (!c->flight_order[k].p) { HIP_TRY(c, c->flight_order[k].alloc((size_t)n_blocks * YCGE_SCHEDULE_SLACK)); HIP_TRY(c, c->flight_ws[k].alloc(96)); HIP_TRY(c, hipMemset(c->flight_ws[k].p, 0, 96 * sizeof(uint32_t))); }
                if (!c->flight_order_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->flight_order_ev[k], hipEventDisableTiming));
            }
            if (c->last_frame_deferred != fs.frame - 1)      // the frame before was not a tiled one: nobody has cleared the NEXT frame's cost slot
                HIP_TRY(c, hipMemsetAsync(c->block_cost.p + (size_t)((cost_slot + 1u) % YCGE_COST_FRAMES) * n_blocks, 0, (size_t)n_blocks * sizeof(uint32_t), stream));
            c->last_frame_deferred = fs.frame;
        }
        int fo = -1;            // a schedule built ahead for exactly this frame (frames in flight, tiled frames; also the first synchronous frame after a burst)
        for (int q = 0; q < 3; q++) if (lpt && c->flight_order_frame[q] == fs.frame) fo = q;
        if (fo >= 0) {
            O.block_order = c->flight_order[fo].p;                   // (a synchronous caller has joined the frames in flight)
            if (deferred || flight) HIP_TRY(c, hipStreamWaitEvent(stream, c->flight_order_ev[fo], 0));       // (built frames ago on a side stream)
            O.n_order = c->flight_ws[fo].p + 16;
            if (c->order_pending) { HIP_TRY(c, hipStreamWaitEvent(stream, c->order_ev, 0)); c->order_pending = false; }        // (an older synchronous schedule still on the side stream: it cleared this frame's cost slot)
        } else {
        if (c->order_pending) { HIP_TRY(c, hipStreamWaitEvent(stream, c->order_ev, 0)); c->order_pending = false; }     // the schedule built beside the last frame's TAA
        O.block_order = (lpt && c->block_order_valid) ? c->block_order.p : nullptr;
        O.n_order = c->order_ws.p + 16;
        }
        const bool flight_order = O.block_order != nullptr && O.block_order != c->block_order.p;
        // the schedule's head (the heaviest blocks of the previous frame) goes to k_trace_fan, launched first and beside k_trace
        const int refill_steps = c->knobs.refill_steps;   // k_trace_refill: steps between refills (0 = k_trace)
        // a schedule without fanned blocks (analytic scenes, small frames) skips k_trace_fan and the side stream altogether: the count
        // comes back through pinned memory and is a frame or two old when read here - either answer traces every block exactly once,
        // because k_trace is told (n_fan pointer or null) which convention this frame uses
        const bool fan = O.block_order != nullptr && (!flight_order || deferred) && c->fan_cap > 0 && !(flat && refill_steps > 0) && c->h_n_fan && *(volatile uint32_t *)c->h_n_fan > 0;       // (schedules of ycge_render_frame_async have no fanned head)
        if (launch_begin) HIP_TRY(c, hipEventRecord(launch_begin, stream));       // (behind the wait for the schedule)
        if (fan) {
            fs.fan_blocks = *(volatile uint32_t *)c->h_n_fan;      // what the last finished schedule handed to k_trace_fan (this frame's may differ by a few)
            // k_trace_fan goes FIRST and on the frame's stream, so that its blocks - the frame's longest chains - are resident from
            // t = 0; the rest of the schedule follows on the side stream (forked before, joined after) and fills in around them
            O.n_fan = O.n_order + 2;          // (word 18 of the schedule's work space, whichever buffer this frame reads)
            TraceOut OF = O;
            OF.lane_base = trace_lanes;
            HIP_TRY(c, hipEventRecord(c->fan_ev[0], stream));
            e = ycge_launch_trace_fan(&c->sd, &P, &OF, c->cfg.count_work, flat, c->fan_cap, stream);
            if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_trace_fan launch failed: %s", hipGetErrorString((hipError_t)e));
            HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->fan_ev[0], 0));
            e = ycge_launch_trace(&c->sd, &P, &O, c->cfg.count_work, flat, 0, c->fan_stream);
            HIP_TRY(c, hipEventRecord(c->fan_ev[1], c->fan_stream));
            HIP_TRY(c, hipStreamWaitEvent(stream, c->fan_ev[1], 0));
        } else {
            if (c->fuse_request && !slab && !rt && !c->in_flight_call && refill_steps == 0 && c->cfg.world_size == 1) {
                const size_t nb = (size_t)c->n_tiles * 4;
                if (c->taa_block_ctr.n != nb) {
                    HIP_TRY(c, c->taa_block_ctr.alloc(nb)); HIP_TRY(c, c->taa_part_ctr.alloc(nb));
                    HIP_TRY(c, hipMemsetAsync(c->taa_block_ctr.p, 0, nb * sizeof(uint32_t), stream)); HIP_TRY(c, hipMemsetAsync(c->taa_part_ctr.p, 0, nb * sizeof(uint32_t), stream));
                }
                O.taa.block_ctr = c->taa_block_ctr.p; O.taa.part_ctr = c->taa_part_ctr.p;
                O.taa.hist = c->taa_hist.p; O.taa.prev_normal = c->prev_normal.p; O.taa.prev_depth = c->prev_depth.p; O.taa.prev_sky = c->prev_sky.p;
                O.taa.T = c->fuse_T;
                c->fuse_done = true;
            }
            e = ycge_launch_trace(&c->sd, &P, &O, c->cfg.count_work, flat, refill_steps, stream);
        }
        if (launch_end) HIP_TRY(c, hipEventRecord(launch_end, stream));
        if (e == 0 && flight) {
            // (frames in flight: the schedule of frame N + 2 follows this frame's TAA on the second stream, ycge_render_frame_async)
        } else if (e == 0 && deferred) {
            HIP_TRY(c, hipEventRecord(c->traced_ev, stream));
            HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->traced_ev, 0));
            e = ycge_launch_order_blocks(c->block_cost.p, n_blocks, policy, split_top, c->fan_class, c->fan_cap, (cost_slot + 2u) % YCGE_COST_FRAMES, 1u << ((cost_slot + 1u) % YCGE_COST_FRAMES),
                                         c->flight_ws[fk].p, c->flight_order[fk].p, c->fan_stream, 0, 0, c->cost_snap.p);          // (the trace of frame N - 1 may still be writing its costs: a copy is read)
            if (e == 0 && c->h_n_fan && c->fan_cap > 0) HIP_TRY(c, hipMemcpyAsync(c->h_n_fan, c->flight_ws[fk].p + 18, sizeof(uint32_t), hipMemcpyDeviceToHost, c->fan_stream));
            HIP_TRY(c, hipEventRecord(c->flight_order_ev[fk], c->fan_stream));
            c->flight_order_frame[fk] = fs.frame + 2;
        } else if (e == 0 && lpt) {
            // the next frame's schedule needs this frame's trace and nothing else: built on the side stream, beside TAA (or the slab
            // pack and all-gather), instead of 25 us in front of it; the next trace waits for it (order_ev)
            HIP_TRY(c, hipEventRecord(c->traced_ev, stream));
            HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->traced_ev, 0));
            e = ycge_launch_order_blocks(c->block_cost.p, n_blocks, policy, split_top, c->fan_class, c->fan_cap, (cost_slot + 1u) % YCGE_COST_FRAMES, 0u, c->order_ws.p, c->block_order.p, c->fan_stream);

            c->block_order_valid = true;
            if (e == 0 && c->h_n_fan && c->fan_cap > 0) HIP_TRY(c, hipMemcpyAsync(c->h_n_fan, c->order_ws.p + 18, sizeof(uint32_t), hipMemcpyDeviceToHost, c->fan_stream));
            HIP_TRY(c, hipEventRecord(c->order_ev, c->fan_stream));
            c->order_pending = true;
        }
    } else {
        const size_t nt = (size_t)(c->n_owned > 0 ? c->n_owned : 1);
        // (an odd frame in flight runs on the second trace stream beside the frame before it: the second set of stage queues)
        const bool second = c->in_flight_call && c->spill_override != nullptr;
        if (second && !c->wf2_q0.p) {
            HIP_TRY(c, c->wf2_q0.alloc(c->wf_q0.n)); HIP_TRY(c, c->wf2_q1.alloc(c->wf_q1.n)); HIP_TRY(c, c->wf2_hit.alloc(c->wf_hit.n)); HIP_TRY(c, c->wf2_lq.alloc(c->wf_lq.n));
            HIP_TRY(c, c->wf2_seg.alloc(c->wf_seg.n)); HIP_TRY(c, c->wf2_counts.alloc(c->wf_counts.n));
        }
        void *bufs[7] = {c->wf_q0.p, c->wf_q1.p, c->wf_hit.p, c->wf_lq.p, c->wf_counts.p, c->wf_counts.p + 7 * nt, c->wf_seg.p};
        if (second) { bufs[0] = c->wf2_q0.p; bufs[1] = c->wf2_q1.p; bufs[2] = c->wf2_hit.p; bufs[3] = c->wf2_lq.p; bufs[4] = c->wf2_counts.p; bufs[5] = c->wf2_counts.p + 7 * nt; bufs[6] = c->wf2_seg.p; }
        // persistent extend: 32 wavefronts per CU (6 per SIMD resident, the rest queue behind them; measured on the voxel world: 16 per CU 13.1 ms, 20 12.5, 24 12.2, 32 12.0, 40 12.1), within the stack-spill columns
        int pw = c->knobs.no_refill ? 0 : c->compute_units * c->knobs.pw_per_cu;
        if ((size_t)pw * 64 > O.stack_lanes) pw = (int)(O.stack_lanes / 64);
        if ((size_t)pw > nt * 4) pw = (int)(nt * 4);            // never more wavefronts than the round can have rays for
        // A persistent wavefront takes a WHOLE tile segment (up to 256 rays, four passes) before it asks for the next: with fewer tiles than
        // wavefronts the round is as long as one wavefront's four passes while most of the machine idles - config 1's 29 tiles: 90 us of a
        // 180 us frame.  Such frames take the plain extend stage instead, four wavefronts per tile side by side (YCGE_PERSIST_MIN_TILES).
        {
            const int pw_full = c->compute_units * c->knobs.pw_per_cu;
            const size_t min_tiles = c->knobs.persist_min_tiles >= 0 ? (size_t)c->knobs.persist_min_tiles : (size_t)(pw_full > 0 ? pw_full / 4 : 0);
            if (nt < min_tiles) pw = 0;
        }
        if (launch_begin) HIP_TRY(c, hipEventRecord(launch_begin, stream));
        // the light loop of a round beside the trace of the next (ycge_launch_wavefront): on the side stream, with a spill area of its own
        // (not where the light loop has nothing to trace - every light dark, timed kernels - nor for the small frames of a burst in flight,
        // where the two stream hops cost more than the overlap gives: config 2 in flight 0.053 -> 0.056 ms, config 5 lit 5.12 -> 4.87)
        const bool beside = !c->knobs.no_lights_beside && c->fan_stream && c->fan_ev[0] && c->fan_ev[1] && c->wf_rounds >= 2 && (c->any_light_lit || c->cfg.count_work) &&
                            (!c->in_flight_call || c->n_owned >= 4096);
        TraceOut O_side = O;
        if (beside) {
            DevBuf<uint64_t> &side = second ? c->stack_spill_side2 : c->stack_spill_side;
            if (side.n != c->stack_spill.n) HIP_TRY(c, side.alloc(c->stack_spill.n));
            O_side.stack_spill = side.p;
        }
        e = ycge_launch_wavefront(&c->sd, &P, &O, bufs, c->wf_rounds, c->has_grid ? 1 : 0, flat, c->cfg.count_work, pw, stream,
                                  beside ? c->fan_stream : nullptr, beside ? c->fan_ev[0] : nullptr, beside ? c->fan_ev[1] : nullptr, beside ? &O_side : nullptr);
        if (launch_end) HIP_TRY(c, hipEventRecord(launch_end, stream));
    }
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "trace launch failed: %s", hipGetErrorString((hipError_t)e));
    if (slab) {
        e = ycge_launch_pack_slab(&P, O.current_hdr, O.g_albedo, O.g_normal, O.g_depth, O.sky, d_slab, (int)slab_floats(c), stream);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_pack_slab launch failed: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(c, hipEventRecord(c->tile_trace_ev[fs.frame & 1], stream));
        c->tile_trace_used[fs.frame & 1] = true;
    }
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[1], stream));
    return YCGE_OK;
}

// step 5's parameters for the frame `fs` (RaytraceRenderer.cs:218, :285, :305) - what TemporalBlendWithClamp will be called with, known before the trace
void taa_decide(ycge_ctx *c, FrameState &fs, TaaParams &T, bool &did_reset)
{
    fs.reset = should_reset_history(c, fs.pos, fs.yaw, fs.pitch) || c->has_dynamic_textures;      // step 2 (:171): this frame's pose against the last committed one; a scene with live textures restarts every frame
    T.w = c->hiW; T.h = c->hiH;
    T.alpha = cs_max(0.0f, cs_min(1.0f, c->cfg.taa_alpha));      // :305
    T.radius = c->cfg.taa_clamp_radius > 0 ? c->cfg.taa_clamp_radius : 0;
    T.pad_lum = c->cfg.taa_luminance_pad;
    did_reset = !c->taa_valid || fs.reset;                        // :285
    T.reset = did_reset ? 1 : 0;
}
// steps 5 and 9: TemporalBlendWithClamp + CommitCamera.  (fused: the trace launch resolved TAA itself - ycge::TaaFuse - and only the commit is left)
int taa_and_commit(ycge_ctx *c, hipStream_t stream, FrameState &fs, bool &did_reset, bool timed, bool fused)
{
    TaaParams T;
    taa_decide(c, fs, T, did_reset);
    if (!fused) {
        int e = ycge_launch_taa(&T, c->current_hdr.p, c->g_normal.p, c->g_depth.p, c->sky.p, c->taa_hist.p, c->prev_normal.p, c->prev_depth.p,
                                c->prev_sky.p, stream, c->in_flight_taa ? 1 : 0);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_taa launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[2], stream));
    c->taa_valid = true;
    c->last_cam[0] = fs.pos[0]; c->last_cam[1] = fs.pos[1]; c->last_cam[2] = fs.pos[2]; c->last_yaw = fs.yaw; c->last_pitch = fs.pitch;   // :266
    return YCGE_OK;
}

// ---- steps 6-8 of TryFlipAndBlit (RaytraceRenderer.cs:221-264): A-trous denoise, auto-exposure, tonemap + downsample
// Level schedule of an in-place A-trous iteration (see ycge_post.hip): T(p) = 1 + max T(q) over every pixel q that
// precedes p in scan order and is stencil-related to it (p reads q -> p needs q's NEW value; q reads p -> q needed
// p's OLD value).  Pixels of one level are mutually unrelated.  Derived from the clamped stencil itself, so it is
// exact for every size, step and border case.
void build_inplace_schedule(int w, int h, int step, std::vector<uint32_t> &pixels, std::vector<uint32_t> &offsets)
{
    const size_t n = (size_t)w * h;
    std::vector<uint32_t> T(n, 0), R(n, 0);      // R[p] = max T over earlier pixels that read p
    uint32_t max_t = 0;
    for (int y = 0; y < h; y++) {
        int sys[5];
        for (int k = -2; k <= 2; k++) { int v = y + k * step; sys[k + 2] = v < 0 ? 0 : v >= h ? h - 1 : v; }
        for (int x = 0; x < w; x++) {
            int sxs[5];
            for (int k = -2; k <= 2; k++) { int v = x + k * step; sxs[k + 2] = v < 0 ? 0 : v >= w ? w - 1 : v; }
            const size_t p = (size_t)x + (size_t)y * w;
            uint32_t m = R[p];
            for (int ky = 0; ky < 5; ky++)
                for (int kx = 0; kx < 5; kx++) {
                    const size_t q = (size_t)sxs[kx] + (size_t)sys[ky] * w;
                    if (q < p && T[q] > m) m = T[q];
                }
            const uint32_t t = m + 1;
            T[p] = t;
            if (t > max_t) max_t = t;
            for (int ky = 0; ky < 5; ky++)
                for (int kx = 0; kx < 5; kx++) {
                    const size_t q = (size_t)sxs[kx] + (size_t)sys[ky] * w;
                    if (q > p && R[q] < t) R[q] = t;
                }
        }
    }
    offsets.assign((size_t)max_t + 1, 0);
    for (size_t p = 0; p < n; p++) offsets[T[p]]++;          // offsets[t] = count of level t (levels are 1-based)
    uint32_t run = 0;
    for (uint32_t t = 1; t <= max_t; t++) { const uint32_t c2 = offsets[t]; offsets[t - 1] = run; run += c2; }
    offsets[max_t] = run;                                    // offsets[l] .. offsets[l + 1] = level l + 1
    pixels.resize(n);
    std::vector<uint32_t> cursor(offsets.begin(), offsets.end() - 1);
    for (size_t p = 0; p < n; p++) pixels[cursor[T[p] - 1]++] = (uint32_t)p;
}

// The level lists regrouped per band of `rows_per_band` image rows: band_pixels sorted by (band, level),
// band_offsets[b * (levels + 1) + t] = start of level t (0-based) of band b.
// The level lists per band, every level padded to whole passes of 32 pixels (0xffffffff = no pixel): band_offsets[b * (levels + 1) + t]
// = first pass of level t of band b (passes are numbered through all bands; pass i covers band_pixels[32 i .. 32 i + 32)).
// max_level_pixels = the most pixels (padding included) one level of one band holds: bounds what a launch of K levels writes.
void band_inplace_schedule(int w, int h, int rows_per_band, const std::vector<uint32_t> &pixels, const std::vector<uint32_t> &offsets,
                           std::vector<uint32_t> &band_pixels, std::vector<uint32_t> &band_offsets, int &n_bands, uint32_t &max_level_pixels,
                           uint32_t G = 32u /* pixels per pass */, const std::vector<int32_t> *row_band = nullptr /* band of every row; n_bands given */)
{
    const int levels = (int)offsets.size() - 1;
    if (!row_band) n_bands = (h + rows_per_band - 1) / rows_per_band;
    auto band_of = [&](uint32_t p) -> size_t { const uint32_t y = p / (uint32_t)w; return row_band ? (size_t)(*row_band)[y] : (size_t)(y / (uint32_t)rows_per_band); };
    band_offsets.assign((size_t)n_bands * (levels + 1), 0);
    std::vector<uint32_t> count((size_t)n_bands * levels, 0);
    for (int t = 0; t < levels; t++)
        for (uint32_t i = offsets[t]; i < offsets[t + 1]; i++) count[band_of(pixels[i]) * levels + t]++;
    uint32_t run = 0;       // in passes
    max_level_pixels = 0;
    for (int b = 0; b < n_bands; b++) {
        for (int t = 0; t < levels; t++) {
            band_offsets[(size_t)b * (levels + 1) + t] = run;
            const uint32_t passes = (count[(size_t)b * levels + t] + G - 1u) / G;
            run += passes;
            if (passes * G > max_level_pixels) max_level_pixels = passes * G;
        }
        band_offsets[(size_t)b * (levels + 1) + levels] = run;
    }
    band_pixels.assign((size_t)run * G, 0xffffffffu);
    std::vector<uint32_t> cursor((size_t)n_bands * levels);
    for (int b = 0; b < n_bands; b++) for (int t = 0; t < levels; t++) cursor[(size_t)b * levels + t] = band_offsets[(size_t)b * (levels + 1) + t] * G;
    for (int t = 0; t < levels; t++)
        for (uint32_t i = offsets[t]; i < offsets[t + 1]; i++) {
            const uint32_t p = pixels[i];
            band_pixels[cursor[band_of(p) * levels + t]++] = (p % (uint32_t)w) | ((p / (uint32_t)w) << 16);      // x | y << 16
        }
}

// Bands for the persistent form at step 2, split by ROW PARITY.  A tap is 0, +-2 or +-4 rows away: rows of one parity only ever
// read rows of the same parity - except where the clamp at the image's top and bottom folds a tap onto row 0 or row h - 1.  So the
// first and the last four rows stay together (a band of 4 rows each), and the rows between them fall apart into two INDEPENDENT
// chains of half-bands (4 even rows, 4 odd rows of an 8-row stretch).  Every band then has 8 pixels a level, half a workgroup's
// wavefronts: the other half fetches the next pass meanwhile (k_atrous_stream's two sets).  desc = 8 ints a band: first row, rows, row stride, pixel groups a pass uses, the (at most two)
// bands it waits for and the (at most two) bands that wait for it (-1: none), derived from the clamped stencil itself.
// Band order: the first four rows, the even chain, the odd chain, the last four rows - neighbours in a chain are neighbours in
// the order.  Returns false where the layout does not apply (a grid below 24 rows).
bool split_band_layout(int h, int step, std::vector<int32_t> &row_band, std::vector<int32_t> &desc, int &n_bands)
{
    const int R = 8, edge = 2 * step;          // rows 0 .. 3 and h - 4 .. h - 1: where the clamp folds taps onto another parity
    if (step != 2 || h < 3 * R) return false;
    // stretches of 8 rows between the edges; every stretch keeps at least 4 rows (a tap reaches 4 rows up: it must not skip a stretch),
    // so a remainder of 1 .. 3 rows takes 4 rows from the stretch before it
    std::vector<int> stretch;
    for (int left = h - 2 * edge; left > 0; left -= R) stretch.push_back(left < R ? left : R);
    if (stretch.size() >= 2 && stretch.back() < 4) { stretch[stretch.size() - 2] -= 4; stretch.back() += 4; }
    if (stretch.empty() || stretch.back() < 4) return false;
    const int chunks = (int)stretch.size();
    n_bands = 2 + 2 * chunks;
    row_band.assign(h, 0);
    desc.assign((size_t)n_bands * 8, -1);
    auto set = [&](int b, int y0, int rows, int stride, int groups) { desc[8 * b] = y0; desc[8 * b + 1] = rows; desc[8 * b + 2] = stride; desc[8 * b + 3] = groups; };
    set(0, 0, edge, 1, 8);
    for (int y = 0; y < edge; y++) row_band[y] = 0;
    for (int k = 0, y_k = edge; k < chunks; y_k += stretch[k], k++)
        for (int par = 0; par < 2; par++) {
            const int b = 1 + par * chunks + k, y_first = y_k + ((y_k & 1) == par ? 0 : 1);       // the stretch's first row of this parity
            int rows = 0;
            for (int y = y_first; y < y_k + stretch[k]; y += 2) { row_band[y] = b; rows++; }
            set(b, y_first, rows, 2, 8);
        }
    const int last = n_bands - 1;
    set(last, h - edge, edge, 1, 8);
    for (int y = h - edge; y < h; y++) row_band[y] = last;
    // who waits for whom: band A needs band B's progress iff a pixel of A reads a row of B that lies above it (same row: same band)
    for (int y = 0; y < h; y++)
        for (int k = 1; k <= 2; k++) {
            int sy = y - k * step; if (sy < 0) sy = 0;
            const int a = row_band[y], b = row_band[sy];
            if (a == b) continue;
            int *up = &desc[8 * a + 4], *dn = &desc[8 * b + 6];
            if (up[0] != b && up[1] != b) { if (up[0] < 0) up[0] = b; else if (up[1] < 0) up[1] = b; else return false; }
            if (dn[0] != a && dn[1] != a) { if (dn[0] < 0) dn[0] = a; else if (dn[1] < 0) dn[1] = a; else return false; }
        }
    // ... and nothing may read DOWN into a row of another chain either (it would be an unordered read of a value in flux)
    for (int y = 0; y < h; y++)
        for (int k = 1; k <= 2; k++) {
            int sy = y + k * step; if (sy >= h) sy = h - 1;
            const int a = row_band[y], b = row_band[sy];
            if (a == b) continue;
            const int *dn = &desc[8 * a + 6];
            if (dn[0] != b && dn[1] != b) return false;         // a lower row read as OLD must belong to a band that waits for this one
        }
    return true;
}

// The narrowest power-of-two window width WX (64 ..) for which no two pixels that ONE launch of k_atrous_band writes - the levels
// [K g, K g + K) of one band - share the entry (row in the band) * WX + (x mod WX), with rows * WX <= capacity; 0 if there is none.
uint32_t band_window_width(const std::vector<uint32_t> &band_pixels, const std::vector<uint32_t> &band_offsets, int n_bands, int levels, int K,
                           int rows_per_band, uint32_t G, uint32_t capacity, const std::vector<int32_t> *desc = nullptr /* split layout: 8 ints a band */)
{
    std::vector<uint32_t> seen;
    const int max_rows = desc ? 8 : rows_per_band;
    for (uint32_t wx = 64; (size_t)wx * max_rows <= capacity; wx *= 2) {
        seen.assign((size_t)wx * max_rows, 0u);
        uint32_t stamp = 0;
        bool ok = true;
        for (int b = 0; b < n_bands && ok; b++) {
            const uint32_t y0 = desc ? (uint32_t)(*desc)[8 * b] : (uint32_t)b * rows_per_band, stride = desc ? (uint32_t)(*desc)[8 * b + 2] : 1u;
            for (int t0 = 0; t0 < levels && ok; t0 += K) {
                stamp++;
                const int t1 = t0 + K < levels ? t0 + K : levels;
                const size_t lo = (size_t)band_offsets[(size_t)b * (levels + 1) + t0] * G, hi = (size_t)band_offsets[(size_t)b * (levels + 1) + t1] * G;
                for (size_t i = lo; i < hi; i++) {
                    const uint32_t e = band_pixels[i];
                    if (e == 0xffffffffu) continue;
                    const uint32_t x = e & 0xffffu, y = e >> 16;
                    const size_t slot = (size_t)((y - y0) / stride) * wx + (x & (wx - 1u));
                    if (seen[slot] == stamp) { ok = false; break; }
                    seen[slot] = stamp;
                }
            }
        }
        if (ok) return wx;
    }
    return 0u;
}

// band workgroups of the persistent in-place A-trous a CU holds at once: what the runtime says of the instantiation that would be
// launched, capped by the YCGE_POST_RESIDENT knob.  0 (the question failed) keeps the persistent form off.
int post_resident_per_cu(ycge_ctx *c, bool split)
{
    int &q = c->post_resident_seen[split ? 1 : 0];
    ycge_atrous_duo_pad_lds(c->knobs.post_pad_lds);
    if (q < 0) q = ycge_atrous_persist_resident(c->knobs.post_groups, split ? 1 : 0, c->knobs.post_mode == 4 ? 0 : 1, c->knobs.post_probe_band >= 0 ? 1 : 0);
    if (c->knobs.post_assume_resident > 0) return c->knobs.post_assume_resident;
    return q < c->knobs.post_resident_per_cu ? q : c->knobs.post_resident_per_cu;
}

int run_post(ycge_ctx *c, hipStream_t stream, float *out_sdr_host, bool timed, hipEvent_t history_read = nullptr /* recorded once the TAA history has been read for the last time */,
             hipEvent_t before_copy = nullptr /* recorded in front of the read-back: the exposure state is this frame's */, bool second_sdr = false,
             hipEvent_t tone_wait = nullptr /* the frame before has left its exposure state: waited for in front of this frame's exposure step */, bool second_set = false)
{
    // (every other frame in flight: the names below stand for the second set of denoise buffers while this call queues its kernels)
    struct SwapPost { ycge_ctx *c; bool on;
        void swap() { std::swap(c->den_a, c->alt_post.den_a); std::swap(c->den_b, c->alt_post.den_b); std::swap(c->unit_n, c->alt_post.unit_n); std::swap(c->exp_terms, c->alt_post.exp_terms);
                      std::swap(c->atrous_statw, c->alt_post.atrous_statw); std::swap(c->exp_scratch, c->alt_post.exp_scratch); std::swap(c->post_progress, c->alt_post.post_progress);
                      std::swap(c->post_epoch, c->alt_post.post_epoch); std::swap(c->post_ticket, c->alt_post.post_ticket); }
        SwapPost(ycge_ctx *c_, bool on_) : c(c_), on(on_) { if (on) swap(); }
        ~SwapPost() { if (on) swap(); } } swap_post(c, second_set);
    const int w = c->hiW, h = c->hiH;
    const size_t n = (size_t)w * h;
    if (!c->den_a.p) {
        HIP_TRY(c, c->den_a.alloc(3 * n)); HIP_TRY(c, c->den_b.alloc(3 * n)); HIP_TRY(c, c->unit_n.alloc(3 * n));
        HIP_TRY(c, c->exp_terms.alloc(n));
    }
    if (!c->d_sdr.p) HIP_TRY(c, c->d_sdr.alloc((size_t)c->fbW * c->fbH * 6));
    if (!c->tone_state.p) {
        HIP_TRY(c, c->tone_state.alloc(ycge_post_state_bytes()));
        const float init[4] = {1.0f, 1.0f, 0.0f, 0.0f};     // aeExposure = 1, effectiveExposure = 1 (ToneMapper.cs:13,17), count = 0
        HIP_TRY(c, hipMemcpy(c->tone_state.p, init, sizeof init, hipMemcpyHostToDevice));
    }
    int e = ycge_launch_unit_normals(c->g_normal.p, c->unit_n.p, n, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_unit_normals launch failed: %s", hipGetErrorString((hipError_t)e));
    const float phi[4] = {cs_max(1e-6f, c->cfg.atrous_c_phi), cs_max(1e-6f, c->cfg.atrous_n_phi), cs_max(1e-6f, c->cfg.atrous_z_phi),
                          cs_max(1e-6f, c->cfg.atrous_a_phi)};
    // ApplyAtrousDenoise's buffer walk, :648-650 and :718 (odd iterations end up in place)
    const float *cur = c->taa_hist.p;
    float *A = c->den_a.p, *B = c->den_b.p, *dst = A;
    const int iters = c->cfg.atrous_iterations > 1 ? c->cfg.atrous_iterations : 1;
    // The in-place iteration (iteration 1, when there is one) reads colour-independent weight factors that need the G-buffer and the
    // unit normals only: they are computed on the side stream beside iteration 0 (fork here, join in front of the band launches)
    bool static_pending = false;
    if (iters >= 2 && c->fan_stream && c->cfg.atrous_inplace_exact) {
        if (!c->atrous_statw.p) HIP_TRY(c, c->atrous_statw.alloc(n * 75));
        HIP_TRY(c, hipEventRecord(c->fan_ev[0], stream));
        HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->fan_ev[0], 0));
        e = ycge_launch_atrous_static(w, h, 2, phi, c->g_albedo.p, c->unit_n.p, c->g_depth.p, c->sky.p, c->atrous_statw.p, c->fan_stream);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_atrous_static launch failed: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(c, hipEventRecord(c->fan_ev[1], c->fan_stream));
        static_pending = true;
    }
    for (int it = 0; it < iters; it++) {
        const int step = 1 << it;
        if (cur == dst) {
            if (w > 65535 || h > 65535 || n * 300 >= ((size_t)1 << 32))
                return c->fail(YCGE_ERR_UNSUPPORTED, "in-place A-trous: trace grid above 65535 pixels a side or 14.3 M pixels (32-bit offsets into the weight table)");
            ycge_ctx::InplaceSchedule *sc = nullptr;
            for (auto *k : c->schedules) if (k->w == w && k->h == h && k->step == step) sc = k;
            if (!sc) {
                sc = new ycge_ctx::InplaceSchedule();
                sc->w = w; sc->h = h; sc->step = step;
                std::vector<uint32_t> px, off, bpx, boff;
                build_inplace_schedule(w, h, step, px, off);
                // bands of whole rows; related pixels are at most 2 * step rows apart, so they share a band or sit in adjacent ones
                const int band_rows = c->knobs.post_band_rows;
                const int rows_per_band = 2 * step > band_rows ? 2 * step : band_rows;
                // the persistent form at step 2: bands split by row parity (split_band_layout) where every band then still finds a place
                std::vector<int32_t> row_band, desc;
                int split_bands = 0;
                if ((c->knobs.post_mode == 0 || c->knobs.post_mode == 3) && !c->knobs.post_no_split && !c->knobs.post_hash && c->knobs.post_groups == 16 && rows_per_band == 8 &&
                    split_band_layout(h, step, row_band, desc, split_bands) && c->compute_units > 0 &&
                    (c->knobs.post_pad_lds > 0 || ((split_bands + 7) / 8) * 8 <= post_resident_per_cu(c, true) * c->compute_units)) {
                    sc->split = true;
                    sc->bands = split_bands;
                    band_inplace_schedule(w, h, rows_per_band, px, off, bpx, boff, sc->bands, sc->max_level_pixels, (uint32_t)c->knobs.post_groups, &row_band);
                    // a half-band's level must fit the 8 pixel groups its workgroup keeps (one pass, the upper 8 entries padding)
                    const int levels_n = (int)off.size() - 1;
                    for (int b2 = 0; b2 < sc->bands && sc->split; b2++) {
                        const uint32_t gmax = (uint32_t)desc[8 * b2 + 3];
                        for (int t = 0; t < levels_n && sc->split; t++) {
                            const uint32_t p0 = boff[(size_t)b2 * (levels_n + 1) + t], p1 = boff[(size_t)b2 * (levels_n + 1) + t + 1];
                            if (p1 - p0 > 1 && gmax < 16u) sc->split = false;
                            for (uint32_t ps = p0; ps < p1 && sc->split; ps++)
                                for (uint32_t g2 = gmax; g2 < 16u; g2++) if (bpx[(size_t)ps * 16 + g2] != 0xffffffffu) sc->split = false;
                        }
                    }
                    if (sc->split) HIP_TRY(c, sc->band_desc.upload(desc));
                }
                if (!sc->split)
                band_inplace_schedule(w, h, rows_per_band, px, off, bpx, boff, sc->bands, sc->max_level_pixels, (uint32_t)c->knobs.post_groups);
                sc->levels = (int)off.size() - 1;
                sc->rows_per_band = rows_per_band;
                // levels per launch: a launch keeps what it writes in a 2048-entry LDS table (k_atrous_band), at most 3/4 full
                sc->levels_per_launch = c->knobs.post_k;
                const int k_cap = (int)(1536u / (sc->max_level_pixels > 0 ? sc->max_level_pixels : 32u));
                if (sc->levels_per_launch > k_cap) sc->levels_per_launch = k_cap;
                sc->window_width = sc->levels_per_launch >= 1 && !c->knobs.post_hash
                                       ? band_window_width(bpx, boff, sc->bands, sc->levels, sc->levels_per_launch, rows_per_band, (uint32_t)c->knobs.post_groups, 2048u, sc->split ? &desc : nullptr) : 0u;
                c->schedules.push_back(sc);
                HIP_TRY(c, sc->pixels.upload(bpx)); HIP_TRY(c, sc->offsets.upload(boff));
                std::vector<uint32_t> plevel(bpx.size() / (size_t)c->knobs.post_groups + 1, 0u);       // level of every pass (k_atrous_stream)
                for (int b = 0; b < sc->bands; b++)
                    for (int t = 0; t < sc->levels; t++)
                        for (uint32_t ps = boff[(size_t)b * (sc->levels + 1) + t]; ps < boff[(size_t)b * (sc->levels + 1) + t + 1]; ps++) plevel[ps] = (uint32_t)t;
                HIP_TRY(c, sc->pass_level.upload(plevel));
            }
            if (sc->split && sc->window_width == 0) return c->fail(YCGE_ERR_DEVICE, "in-place A-trous: the split band layout found no collision-free window (set YCGE_POST_NO_SPLIT=1)");
            const int levels_per_launch = sc->levels_per_launch;
            if (levels_per_launch < 1) return c->fail(YCGE_ERR_UNSUPPORTED, "in-place A-trous: a level of %u pixels in one band", sc->max_level_pixels);
            if (!c->atrous_statw.p) HIP_TRY(c, c->atrous_statw.alloc(n * 75));
            if (static_pending && step == 2) { HIP_TRY(c, hipStreamWaitEvent(stream, c->fan_ev[1], 0)); static_pending = false; }
            else {
                e = ycge_launch_atrous_static(w, h, step, phi, c->g_albedo.p, c->unit_n.p, c->g_depth.p, c->sky.p, c->atrous_statw.p, stream);
                if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_atrous_static launch failed: %s", hipGetErrorString((hipError_t)e));
            }
            // one persistent launch when the window form applies and every band's workgroup is resident at once (it waits for its
            // neighbour inside the kernel); else a launch per level group
            const bool persist = sc->split || (c->knobs.post_mode != 2 && (c->knobs.post_groups <= 16 || c->knobs.post_mode == 4) && sc->window_width != 0 && (size_t)sc->rows_per_band * sc->window_width <= 2048 &&
                                 c->compute_units > 0 && ((sc->bands + 7) / 8) * 8 <= post_resident_per_cu(c, sc->split) * c->compute_units);
            if (persist) {
                // Bands of one XCD adjacent (their colours meet in one L2) while every band has a CU of its own: 1080p 3.90 against 4.03 ms.
                // Where two bands must share a CU (a 4K grid: 270 bands) block order is the better one - 14.7 against 15.9 ms, launch
                // form 16.3: the pairs a CU gets are then far apart in the image and busy at different times.
                const int xcd_local = c->knobs.post_mode == 3 ? 0 : c->knobs.post_mode == 0 ? (((sc->bands + 7) / 8) * 8 <= c->compute_units ? 1 : 0) : 1;
                const uint32_t groups = (uint32_t)((sc->levels + levels_per_launch - 1) / levels_per_launch);
                if (c->post_progress.n < (size_t)sc->bands * 32 + 8000 || c->post_epoch > 0x60000000u) {
                    HIP_TRY(c, c->post_progress.reserve((size_t)sc->bands * 32 + 8000));       // + room for the profiling timeline of two bands
                    HIP_TRY(c, hipMemsetAsync(c->post_progress.p, 0, ((size_t)sc->bands * 32 + 8000) * sizeof(uint32_t), stream));
                    if (c->knobs.post_probe_band >= 0) { const uint32_t v = (uint32_t)c->knobs.post_probe_band + 1u; HIP_TRY(c, hipMemcpyAsync(c->post_progress.p + (size_t)sc->bands * 32 + 7999, &v, 4, hipMemcpyHostToDevice, stream)); HIP_TRY(c, hipStreamSynchronize(stream)); }
                    c->post_epoch = 0;
                    c->post_ticket = 0;
                }
                ycge_atrous_duo_pad_lds(c->knobs.post_pad_lds);
                e = ycge_launch_atrous_persist(w, h, step, phi, dst, c->sky.p, c->atrous_statw.p, sc->pixels.p, sc->offsets.p, sc->pass_level.p, sc->split ? sc->band_desc.p : nullptr, sc->levels, sc->bands,
                                               levels_per_launch, c->knobs.post_groups, sc->rows_per_band, sc->window_width, c->post_progress.p, c->post_epoch,
                                               xcd_local | (c->knobs.post_dbg_free ? 2 : 0), c->knobs.post_mode == 4 ? 0 : 1, c->knobs.post_probe_band >= 0 ? 1 : 0, c->post_ticket, stream);
                if (!xcd_local && c->knobs.post_mode != 4) c->post_ticket += (uint32_t)sc->bands;      // one number per workgroup of the launch
                c->post_epoch += (groups > (uint32_t)sc->levels ? groups : (uint32_t)sc->levels) + 1u;
            } else
            e = ycge_launch_atrous_inplace(w, h, step, phi, dst, c->g_albedo.p, c->unit_n.p, c->g_depth.p, c->sky.p, c->atrous_statw.p,
                                           sc->pixels.p, sc->offsets.p, sc->levels, sc->bands, levels_per_launch, c->knobs.post_groups, sc->rows_per_band, sc->window_width, stream);
        } else {
            e = ycge_launch_atrous(w, h, step, phi, cur, dst, c->g_albedo.p, c->unit_n.p, c->g_depth.p, c->sky.p, stream);
        }
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "A-trous launch failed: %s", hipGetErrorString((hipError_t)e));
        if (it == 0 && iters > 1 && history_read) { HIP_TRY(c, hipEventRecord(history_read, stream)); history_read = nullptr; }        // (iteration 0 is the only reader of taa_hist when there are more)
        // the reference's swap, :718 (`tmp` is the history after iteration 0, so iteration 1 gets dst = A = cur: in place); waived
        // (config.atrous_inplace_exact = 0): plain ping-pong between A and B
        const float *tmp = cur; cur = dst; dst = c->cfg.atrous_inplace_exact ? ((tmp == A) ? B : A) : ((cur == A) ? B : A);
    }
    if (static_pending) HIP_TRY(c, hipStreamWaitEvent(stream, c->fan_ev[1], 0));
    c->denoised = cur;
    const int step = c->ss * 2 > 2 ? c->ss * 2 : 2;            // :226
    const float tone_consts[5] = {1.0f, 0.18f, 0.2f, 0.10f, 1.50f};     // toneExposure, aeKey, aeSpeed, aeMin, aeMax (ToneMapper.cs:8-16)
    if (!c->exp_scratch.p) HIP_TRY(c, c->exp_scratch.alloc(ycge_exposure_scratch_bytes(w, h, step)));
    if (tone_wait) HIP_TRY(c, hipStreamWaitEvent(stream, tone_wait, 0));
    e = ycge_launch_exposure(cur, c->sky.p, w, h, step, c->exp_terms.p, c->tone_state.p, tone_consts, c->exp_scratch.p, c->knobs.exposure_serial ? 1 : 0, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "exposure launch failed: %s", hipGetErrorString((hipError_t)e));
    if (second_sdr && !c->d_sdr2.p) HIP_TRY(c, c->d_sdr2.alloc((size_t)c->fbW * c->fbH * 6));
    float *d_sdr = second_sdr ? c->d_sdr2.p : c->d_sdr.p;
    e = ycge_launch_tonemap(cur, w, c->fbW, c->fbH, c->ss, 2.2f, 2.0f, 0.0f, c->tone_state.p, d_sdr, stream);   // toneGamma, toneSaturation, toneVibrance
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "tonemap launch failed: %s", hipGetErrorString((hipError_t)e));
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[3], stream));
    if (before_copy) HIP_TRY(c, hipEventRecord(before_copy, stream));
    if (out_sdr_host) {
        const size_t sdr_bytes = (size_t)c->fbW * c->fbH * 6 * sizeof(float);
        float *target = out_sdr_host;
        if (!host_memory_is_page_locked(out_sdr_host, sdr_bytes)) {        // (synchronous callers only: the frames in flight refuse a pageable array up front)
            const int rs = ensure_out_stage(c, sdr_bytes);
            if (rs != YCGE_OK) return rs;
            target = (float *)c->out_stage;
            c->staged_sdr_dst = out_sdr_host; c->staged_sdr_bytes = sdr_bytes;
        }
        HIP_TRY(c, hipMemcpyAsync(target, d_sdr, sdr_bytes, hipMemcpyDeviceToHost, stream));
    }
    if (history_read) HIP_TRY(c, hipEventRecord(history_read, stream));        // (a single iteration: exposure and tonemap read the history itself)
    return YCGE_OK;
}

int fill_stats(ycge_ctx *c, ycge_frame_stats *st, const FrameState &fs, bool did_reset, bool have_taa, double wall_ms)
{
    if (!st) return YCGE_OK;
    std::memset(st, 0, sizeof *st);
    st->frame = fs.frame;
    st->history_reset = did_reset ? 1 : 0;
    st->fan_blocks = (int32_t)fs.fan_blocks;
    float ms = 0.0f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    st->trace_ms = ms;
    if (have_taa) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); st->taa_ms = ms; }
    st->total_ms = wall_ms;
    st->n_devices_traced = 1 + (int32_t)c->peers.size();
    st->device_tiles[0] = c->n_owned;
    for (size_t i = 0; i < c->peers.size() && i + 1 < YCGE_MAX_DEVICES; i++) st->device_tiles[i + 1] = c->peers[i]->n_owned;
    if (c->cfg.count_work) {
        unsigned long long h[8];
        { const int cr = copy_out(c, h, c->counters.p, sizeof h); if (cr != YCGE_OK) return cr; }
        st->n_rays = h[0]; st->n_box = h[1]; st->n_tri = h[2]; st->n_prim = h[3]; st->n_vox = h[4]; st->n_rays_dark = h[5];
    }
    st->exposure = 1.0f;
    return YCGE_OK;
}

} // namespace ycge_host

namespace ycge_host {
// every other entry point first waits for what ycge_render_frame_async left in flight (the frames-in-flight machinery is below)
int join_async(ycge_ctx *c)
{
    if (!c->async_outstanding) return YCGE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(c, hipStreamSynchronize(c->stream2));
    HIP_TRY(c, hipStreamSynchronize(c->taa_stream));
    c->async_outstanding = false;
    c->set_read[0] = c->set_read[1] = c->set_read[2] = false;
    c->post_hist_pending = c->post_busy = c->post_set_pending[0] = c->post_set_pending[1] = c->post_set_pending[2] = false;
    return YCGE_OK;
}
} // namespace ycge_host

extern "C" {

// one peer's share of the frame `fs`: its tiles traced on its own stream, then written into the root's frame buffers (k_push_tiles)
static int peer_trace_and_push(ycge_ctx *c, ycge_ctx *p, FrameState &pfs)
{
    if (hipSetDevice(p->device) != hipSuccess) return p->fail(YCGE_ERR_DEVICE, "hipSetDevice(%d) failed", p->device);
    p->frame_counter = pfs.frame;
    const bool rccl = c->exchange_mode == YCGE_EXCHANGE_RCCL;       // the tiles leave as a slab for the all-gather the root queues (trace_on_all_devices); only debug captures are still pushed
    int rc = trace_frame(p, rccl ? p->own_slab.p : nullptr, p->stream, pfs, false);
    if (rc != YCGE_OK) return rc;
    PushPlanes L;
    std::memset(&L, 0, sizeof L);
    auto plane = [&](const void *src, void *dst, int bpp) { if (src && dst) { L.src[L.n] = (const uint8_t *)src; L.dst[L.n] = (uint8_t *)dst; L.bytes_per_pixel[L.n] = bpp; L.n++; } };
    if (!rccl) {
    plane(p->current_hdr.p, c->current_hdr.p, 12); plane(p->g_albedo.p, c->g_albedo.p, 12); plane(p->g_normal.p, c->g_normal.p, 12);
    plane(p->g_depth.p, c->g_depth.p, 4); plane(p->sky.p, c->sky.p, 1);
    }
    if (c->cfg.capture_debug) {
        plane(p->dbg_rays.p, c->dbg_rays.p, 24); plane(p->dbg_prim.p, c->dbg_prim.p, 4); plane(p->dbg_sub.p, c->dbg_sub.p, 4);
        plane(p->dbg_hit_t.p, c->dbg_hit_t.p, 4); plane(p->dbg_rng.p, c->dbg_rng.p, 8);
    }
    FrameParams P;
    fill_frame_params(p, P, pfs.frame, pfs.pos, pfs.yaw, pfs.pitch, pfs.fov);
    const int e = L.n > 0 ? ycge_launch_push_tiles(&P, &L, p->stream) : 0;
    if (e != 0) return p->fail(YCGE_ERR_DEVICE, "k_push_tiles launch failed: %s", hipGetErrorString((hipError_t)e));
    if (hipEventRecord(p->pushed_ev, p->stream) != hipSuccess) return p->fail(YCGE_ERR_DEVICE, "hipEventRecord failed on device %d", p->device);
    return YCGE_OK;
}

static void peer_worker_main(ycge_ctx *c, ycge_ctx *p)
{
    ycge_ctx::PeerWorker &w = *p->worker;
    for (;;) {
        std::unique_lock<std::mutex> lk(w.m);
        w.cv.wait(lk, [&] { return w.job == 1 || w.job == -1; });
        if (w.job == -1) return;
        FrameState pfs = w.fs;
        lk.unlock();
        int rc;
        try { rc = peer_trace_and_push(c, p, pfs); } catch (...) { rc = abi_catch(p); }     // (an exception that left a thread function would end the process)
        lk.lock();
        w.fs = pfs; w.rc = rc; w.job = 2;
        lk.unlock();
        w.cv.notify_all();
    }
}

// n_devices >= 2: every device traces its tiles of the frame `fs` (each peer's launches issued by its own thread, side by side with
// the root's); the peers then copy theirs into this (rank 0's) frame buffers
static int trace_on_all_devices(ycge_ctx *c, FrameState &fs)
{
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    for (ycge_ctx *p : c->peers) {
        ycge_ctx::PeerWorker &w = *p->worker;
        { std::lock_guard<std::mutex> g(w.m); w.fs = fs; w.job = 1; }
        w.cv.notify_all();
    }
    const bool rccl = c->exchange_mode == YCGE_EXCHANGE_RCCL;
    int rc = trace_frame(c, rccl ? c->own_slab.p : nullptr, c->stream, fs, false);
    for (ycge_ctx *p : c->peers) {          // (every posted frame is collected, whatever the root's own share returned)
        ycge_ctx::PeerWorker &w = *p->worker;
        std::unique_lock<std::mutex> lk(w.m);
        w.cv.wait(lk, [&] { return w.job == 2; });
        w.job = 0;
        if (w.rc != YCGE_OK && rc == YCGE_OK) { rc = w.rc; c->err = p->err; }
        fs.fan_blocks += w.fs.fan_blocks;
    }
    if (rc != YCGE_OK) return rc;
    if (rccl) {
        // ONE all-gather of the tile slabs over xGMI (north_star), rank r's call on device r's stream behind its trace and pack,
        // all of them in one group from this thread; then the root's copy is un-permuted into the frame buffers TAA and the post stage read
        const RcclApi &R = load_rccl();
        const size_t per_rank = (size_t)c->tiles_per_rank_padded * 256 * slab_floats(c);
        std::vector<ycge_ctx *> ranks{c};
        ranks.insert(ranks.end(), c->peers.begin(), c->peers.end());
        int nr = R.GroupStart();
        for (size_t r = 0; r < ranks.size() && nr == 0; r++) {
            (void)hipSetDevice(ranks[r]->device);
            nr = R.AllGather(ranks[r]->own_slab.p, ranks[r]->all_slabs.p, per_rank, 7 /* ncclFloat32 */, c->nccl_comms[r], ranks[r]->stream);
        }
        const int ne = R.GroupEnd();
        (void)hipSetDevice(c->device);
        if (nr != 0 || ne != 0) return c->fail(YCGE_ERR_DEVICE, "ncclAllGather failed: %s", R.GetErrorString ? R.GetErrorString(nr != 0 ? nr : ne) : "?");
        const int e = ycge_launch_unpermute(c->all_slabs.p, per_rank, c->hiW, c->hiH, c->tiles_x, c->n_tiles, c->cfg.world_size, (int)slab_floats(c),
                                            c->current_hdr.p, c->g_albedo.p, c->g_normal.p, c->g_depth.p, c->sky.p, c->stream);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_unpermute launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    for (ycge_ctx *p : c->peers) HIP_TRY(c, hipStreamWaitEvent(c->stream, p->pushed_ev, 0));
    HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));        // trace_ms of a multi-device frame: until the last tile has arrived
    return YCGE_OK;
}

// Frames in flight.  ycge_render_frame is the reference's call: one frame, finished when it returns.  A caller that only wants
// frames per second (a benchmark, a recorder, a render thread that flips when a frame is there) loses the gap between two traces
// to it: TAA, the schedule of the next frame and the host's wake-up stand between them (57 us of a 0.55 ms frame on config 4) although
// the trace of frame N + 1 needs nothing of frame N's TAA.  ycge_render_frame_async queues steps 1-5 and 9 of a frame and returns:
// the trace on the context's stream, TAA on a second one, a frame's trace outputs alternating between two sets of buffers so that
// TAA of frame N reads one set while the trace of frame N + 1 writes the other.  Same kernels, same order of frames, same bits
// (tests/test_gpu_timed_variants.py); every other entry point first waits for what is in flight (join_async).

int ycge_wait(ycge_ctx *c)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    return join_async(c);
}
catch (...) { return ycge_host::abi_catch(c); }

// measurement: how long the trace launches of the frames queued since the last call took (HIP events around them on the context's
// stream, behind the wait for the schedule), oldest first, at most the last YCGE_FLIGHT_RING of them.  Waits for the frames in flight.
int ycge_async_trace_times(ycge_ctx *c, float *ms_out, int32_t capacity, int32_t *n_out)
try {
    if (!c || !n_out || (capacity > 0 && !ms_out)) return YCGE_ERR_INVALID_ARG;
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    if (c->knobs.flight_no_begin) { *n_out = 0; c->flight_frames = 0; return YCGE_OK; }
    const uint64_t have = c->flight_frames < YCGE_FLIGHT_RING ? c->flight_frames : YCGE_FLIGHT_RING;
    const uint64_t n = have < (uint64_t)(capacity > 0 ? capacity : 0) ? have : (uint64_t)(capacity > 0 ? capacity : 0);
    for (uint64_t i = 0; i < n; i++) {
        const size_t slot = (size_t)((c->flight_frames - n + i) % YCGE_FLIGHT_RING);
        HIP_TRY(c, hipEventElapsedTime(&ms_out[i], c->flight_ev[2 * slot], c->flight_ev[2 * slot + 1]));
    }
    if (getenv("YCGE_FLIGHT_DEBUG") && n > 12) {        // profiling aid: how the last frames' traces lie to each other (negative end -> next begin: they overlap)
        for (uint64_t i = n - 10; i + 1 < n; i++) {
            const size_t s0 = (size_t)((c->flight_frames - n + i) % YCGE_FLIGHT_RING), s1 = (size_t)((c->flight_frames - n + i + 1) % YCGE_FLIGHT_RING);
            float bb = 0, eb = 0, d = 0;
            (void)hipEventElapsedTime(&bb, c->flight_ev[2 * s0], c->flight_ev[2 * s1]);
            (void)hipEventElapsedTime(&eb, c->flight_ev[2 * s0 + 1], c->flight_ev[2 * s1]);
            (void)hipEventElapsedTime(&d, c->flight_ev[2 * s0], c->flight_ev[2 * s0 + 1]);
            fprintf(stderr, "frame %llu: duration %.4f begin->next begin %.4f end->next begin %.4f\n", (unsigned long long)i, d, bb, eb);
        }
    }
    *n_out = (int32_t)n;
    c->flight_frames = 0;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// what the frames in flight of this context do (timing machinery only: DESIGN section 6); the gate's state is what render_frame_in_flight
// last found - before the first frame in flight it is the knob's
int ycge_flight_query(ycge_ctx *c, ycge_flight_info *out)
try {
    if (!c || !out) return YCGE_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const bool single_launch = c->have_scene && frame_is_single_launch(c) && !c->sd.any_transparent;
    const bool stage_pair = c->have_scene && !frame_is_single_launch(c) && !c->sd.any_transparent && !c->knobs.no_flight_stage_overlap && c->n_owned >= 4096;      // (the stage pipeline with its second set of queues)
    out->two_trace_streams = (c->knobs.flight_overlap && c->stream2 && (single_launch || stage_pair)) ? 1 : 0;
    out->placed_gate = (out->two_trace_streams && single_launch && c->knobs.flight_placed_gate && c->knobs.refill_steps == 0) ? 1 : 0;      // (only the single-launch kernels store the value)
    out->post_gate = c->knobs.flight_post_gate ? 1 : 0;
    out->post_pair = (out->two_trace_streams && c->knobs.flight_post_pair) ? 1 : 0;
    out->frames_outstanding = c->async_outstanding ? 1 : 0;
    out->stage_pipeline = (c->have_scene && !frame_is_single_launch(c)) ? 1 : 0;
    out->placed_waits = c->placed_waits;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_exchange_query(ycge_ctx *c, int32_t *mode_out, int32_t *world_out)
try {
    if (!c || !mode_out) return YCGE_ERR_INVALID_ARG;
    *mode_out = c->exchange_mode;
    if (world_out) *world_out = 1 + (int32_t)c->peers.size();
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

static int render_frame_in_flight(ycge_ctx *c, float *out_sdr);
int ycge_render_frame_async(ycge_ctx *c)
try {
    return c ? render_frame_in_flight(c, nullptr) : YCGE_ERR_INVALID_ARG;
}
catch (...) { return ycge_host::abi_catch(c); }
// ... with steps 6-8 (denoise, exposure, tonemap + downsample) and the read-back into out_top_bottom_sdr, which is filled when the frame
// is complete (ycge_wait, or any other call): page-locked memory (ycge_pin_host_buffer) keeps the copy off the caller's thread, and a
// caller that queues several such frames passes a buffer per frame in flight.  The post stage of frame N runs beside the traces and TAA
// of the frames after it (its in-place iteration is a dependent chain that leaves most of the chip idle).
int ycge_render_frame_async_sdr(ycge_ctx *c, float *out_top_bottom_sdr)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!out_top_bottom_sdr) return c->fail(YCGE_ERR_INVALID_ARG, "null SDR buffer");
    return render_frame_in_flight(c, out_top_bottom_sdr);
}
catch (...) { return ycge_host::abi_catch(c); }

static int render_frame_in_flight(ycge_ctx *c, float *out_sdr)
{
    if (c->parent || !c->peers.empty() || c->cfg.world_size != 1)
        return c->fail(YCGE_ERR_INVALID_ARG, "ycge_render_frame_async is the single-device form (tiled frames overlap through ycge_trace_tiles / ycge_resolve_gathered on two streams)");
    if (c->cfg.capture_debug || c->cfg.count_work) return c->fail(YCGE_ERR_INVALID_ARG, "ycge_render_frame_async keeps neither debug captures nor per-frame counters: use ycge_render_frame");
    HIP_TRY(c, hipSetDevice(c->device));
    if (out_sdr && !host_memory_is_page_locked(out_sdr, (size_t)c->fbW * c->fbH * 6 * sizeof(float)))
        return c->fail(YCGE_ERR_INVALID_ARG, "ycge_render_frame_async_sdr fills its array while the caller runs on: it must be page-locked memory (ycge_alloc_host_buffer, or whole pages registered with ycge_pin_host_buffer)");
    if (!c->taa_stream) return c->fail(YCGE_ERR_INVALID_ARG, "no second stream: frames in flight need a single-device context");
    for (int k = 0; k < 3; k++) if (!c->set_resolved_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->set_resolved_ev[k], hipEventDisableTiming));
    for (int k = 0; k < 3; k++) if (!c->flight_order_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->flight_order_ev[k], hipEventDisableTiming));
    if (c->flight_ev.empty()) {
        c->flight_ev.resize(2 * YCGE_FLIGHT_RING, nullptr);
        for (hipEvent_t &ev : c->flight_ev) HIP_TRY(c, hipEventCreate(&ev));
    }
    const size_t npx = (size_t)c->hiW * c->hiH;
    if (!c->alt_hdr.p) {
        HIP_TRY(c, c->alt_hdr.alloc(3 * npx)); HIP_TRY(c, c->alt_albedo.alloc(3 * npx)); HIP_TRY(c, c->alt_normal.alloc(3 * npx));
        HIP_TRY(c, c->alt_depth.alloc(npx)); HIP_TRY(c, c->alt_sky.alloc(npx));
    }
    if (!c->alt2_hdr.p) {
        HIP_TRY(c, c->alt2_hdr.alloc(3 * npx)); HIP_TRY(c, c->alt2_albedo.alloc(3 * npx)); HIP_TRY(c, c->alt2_normal.alloc(3 * npx));
        HIP_TRY(c, c->alt2_depth.alloc(npx)); HIP_TRY(c, c->alt2_sky.alloc(npx));
    }
    const uint32_t n_blocks = (uint32_t)c->n_owned * 4u;
    for (int k = 0; k < 3; k++)
        if (!c->flight_order[k].p) { HIP_TRY(c, c->flight_order[k].alloc((size_t)n_blocks * YCGE_SCHEDULE_SLACK)); HIP_TRY(c, c->flight_ws[k].alloc(96)); HIP_TRY(c, hipMemset(c->flight_ws[k].p, 0, 96 * sizeof(uint32_t))); }
    FrameState fs;
    snapshot_frame(c, fs);
    // Two traces at a time: odd frames go to a second stream, so that the bulk of frame N + 1 fills the wavefront slots the tail of frame
    // N leaves empty (a third of a trace's slot-time on config 4: its last chains).  Nothing else changes - frame N + 1's trace never
    // needed frame N's: its order, its cost slot and its output set are ready once the second stream's work of frame N - 1 is done, which
    // is the one event it waits for.  Only the single-launch kernel (the stage pipeline of voxel worlds shares its queues between frames)
    // and only without refraction stacks.
    const bool overlap_scene = c->knobs.flight_overlap && c->stream2 && c->have_scene && !c->sd.any_transparent &&
                               (frame_is_single_launch(c) || (!c->knobs.no_flight_stage_overlap && c->n_owned >= 4096 && !out_sdr && !c->post_busy));
    // (the stage pipeline: a second set of queues, trace_frame; big frames only - small ones gain nothing from a second stream's hops; and not
    // while a post stage is in flight: its persistent in-place launch needs its band workgroups placed, and behind TWO frames' persistent
    // extend stages they are not - one run in four of `bench.py --config 5` stretched to 0.4 s a frame, profiles/r04/h_voxel_walk_tree.txt)
    const bool overlap = overlap_scene && (fs.frame & 1);
    hipStream_t ts = overlap ? c->stream2 : c->stream;
    if (overlap && !c->stack_spill2.p) HIP_TRY(c, c->stack_spill2.alloc(c->stack_spill.n));
    if (!c->async_outstanding) {
        // the first frame in flight after synchronous calls.  Whatever they left on the context's stream (a TAA, a post stage that reads
        // the current set) is ahead of this trace in stream order, and the second stream's first TAA waits for this trace.  The
        // synchronous schedule cleared THIS frame's cost slot; the next frame's would have been cleared between the two traces.
        // ORDER matters here and must not depend on the placed-value gate (it may be off: YCGE_FLIGHT_PLACED_GATE=0, no signal memory,
        // k_trace_refill): first the synchronous path's schedule still on the side stream (it writes the order buffer and its counters the
        // second stream's first trace reads), then the cost-slot clears, THEN the fork the second trace stream waits for.
        if (c->order_pending) { HIP_TRY(c, hipStreamWaitEvent(c->stream, c->order_ev, 0)); c->order_pending = false; }
        for (int ahead = 1; ahead <= 2; ahead++)        // (the schedules queued in flight clear the slot of the frame three ahead)
            HIP_TRY(c, hipMemsetAsync(c->block_cost.p + (size_t)((uint64_t)(fs.frame + ahead) % YCGE_COST_FRAMES) * n_blocks, 0, (size_t)n_blocks * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipEventRecord(c->flight_fork_ev, c->stream));
        if (c->stream2) HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->flight_fork_ev, 0));
    }
    // the other set becomes "the current frame's": every later reader (TAA below, a read-back, a synchronous frame's post stage) goes by these names
    // (three sets: current <- alt2, frame N - 3's; alt2 <- alt, N - 2's; alt <- the old current, N - 1's)
    auto rotate3 = [](auto &a, auto &b, auto &d) { std::swap(a, d); std::swap(b, d); };
    rotate3(c->current_hdr, c->alt_hdr, c->alt2_hdr); rotate3(c->g_albedo, c->alt_albedo, c->alt2_albedo); rotate3(c->g_normal, c->alt_normal, c->alt2_normal);
    rotate3(c->g_depth, c->alt_depth, c->alt2_depth); rotate3(c->sky, c->alt_sky, c->alt2_sky);
    rotate3(c->set_id[0], c->set_id[1], c->set_id[2]);
    const int k = c->out_set = c->set_id[0];
    const int par = (int)(fs.frame & 1);          // the trace stream, the denoise buffers and the device SDR array go by frame parity
    // ONE wait per frame on the trace's stream: the second stream's work of two frames ago - TAA (it read this set of buffers) and
    // behind it the schedule for THIS frame (it wrote the order buffer of this parity and cleared this frame's cost slot)
    if (c->set_read[k]) HIP_TRY(c, hipStreamWaitEvent(ts, c->set_resolved_ev[k], 0));
    // ... and (two traces at a time) for the moment the trace before has PLACED its last workgroup: started earlier, this frame's heaviest
    // blocks take wavefront places from that frame's bulk - both frames' longest chains then start late - and started later the machine
    // idles.  The last workgroup of a trace launch stores the frame's sequence number; this stream waits for the value.
    if (overlap_scene && c->knobs.flight_placed_gate && c->knobs.refill_steps == 0 /* (k_trace_refill stores no value) */) {
        if (!c->placed_flag) {
            // (a runtime without signal memory: no gate - the frames come out the same, a little later)
            if (hipExtMallocWithFlags((void **)&c->placed_flag, 8, hipMallocSignalMemory) != hipSuccess || hipMemset(c->placed_flag, 0, 8) != hipSuccess) {
                (void)hipGetLastError();
                if (c->placed_flag) { (void)hipFree(c->placed_flag); c->placed_flag = nullptr; }
                c->knobs.flight_placed_gate = false;
            }
            c->placed_expect = 0; c->placed_next = 0;
        }
    }
    if (overlap_scene && c->knobs.flight_placed_gate && c->knobs.refill_steps == 0 && c->placed_flag) {
        if (c->placed_expect) { HIP_TRY(c, hipStreamWaitValue32(ts, c->placed_flag, c->placed_expect, hipStreamWaitValueGte, 0xffffffffu)); c->placed_waits++; }
        c->placed_next = c->placed_expect + 1u;
        if (c->placed_next == 0u) c->placed_next = 1u;
    } else c->placed_next = 0;
    if (c->post_set_pending[k]) { HIP_TRY(c, hipStreamWaitEvent(ts, c->post_set_ev[k], 0)); c->post_set_pending[k] = false; }      // a post stage still reads this set's G-buffer
    // ... and a trace does not start while the post stage of the frame before has yet to place its persistent in-place launch: behind a
    // running trace's 64 800 pending workgroups its band workgroups (nine wavefronts each) find their CUs one by one, and every band
    // waits for the slowest to arrive (config 4 with the post stage in flight: 4.7 ms a frame instead of 3.6 synchronous)
    if (c->post_hist_pending && c->knobs.flight_post_gate) HIP_TRY(c, hipStreamWaitEvent(ts, c->post_hist_ev, 0));
    const size_t slot = (size_t)(c->flight_frames % YCGE_FLIGHT_RING);
    hipEvent_t ev_begin = c->knobs.flight_no_begin ? nullptr : c->flight_ev[2 * slot], ev_end = c->flight_ev[2 * slot + 1];
    c->in_flight_call = true;
    c->spill_override = overlap ? c->stack_spill2.p : nullptr;
    int rc = trace_frame(c, nullptr, ts, fs, false, ev_begin, nullptr);
    c->spill_override = nullptr;
    c->in_flight_call = false;
    if (rc != YCGE_OK) return rc;
    if (c->placed_next && fs.single_launch) c->placed_expect = c->placed_next;       // (only the single-launch kernels store the value)
    HIP_TRY(c, hipEventRecord(ev_end, ts));          // end of the trace: the timing ring's event is also what the second stream waits for
    c->flight_frames++;
    HIP_TRY(c, hipStreamWaitEvent(c->taa_stream, ev_end, 0));
    const bool small = c->knobs.flight_overlap && c->knobs.flight_small_groups;
    if (fs.scheduled) {
        // the schedule of frame N + 2, from the costs up to this frame's: the slot frame N + 1's trace is writing meanwhile is left out,
        // the one frame N + 2's will write is cleared; into the order buffer this frame's trace has just finished reading
        uint32_t policy, split_top;
        schedule_policy(c, policy, split_top);
        const uint32_t in_flight = (1u << ((uint64_t)(fs.frame + 1) % YCGE_COST_FRAMES)) | (1u << ((uint64_t)(fs.frame + 2) % YCGE_COST_FRAMES));
        const uint32_t target = (uint32_t)((uint64_t)(fs.frame + 3) % YCGE_COST_FRAMES);
        const int fk = (int)((uint64_t)fs.frame % 3u);
        const int e = ycge_launch_order_blocks(c->block_cost.p, n_blocks, policy, split_top, 0u, 0u, target, in_flight, c->flight_ws[fk].p, c->flight_order[fk].p, c->taa_stream, small ? 1 : 0, 0, c->cost_snap.p);       // (the trace of frame N - 1, on the other stream, may still be writing the slot this build reads and clears: a copy is read)
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "schedule launch failed: %s", hipGetErrorString((hipError_t)e));
        c->flight_order_frame[fk] = fs.frame + 3;
        HIP_TRY(c, hipEventRecord(c->flight_order_ev[fk], c->taa_stream));
    }
    if (c->post_hist_pending) { HIP_TRY(c, hipStreamWaitEvent(c->taa_stream, c->post_hist_ev, 0)); c->post_hist_pending = false; }      // the post stage of the frame before reads the history this TAA rewrites
    bool did_reset = false;
    c->in_flight_taa = small;
    rc = taa_and_commit(c, c->taa_stream, fs, did_reset, false, false);
    c->in_flight_taa = false;
    if (rc != YCGE_OK) return rc;
    if (out_sdr) {
        for (hipEvent_t *ev : {&c->flight_taa_ev, &c->post_hist_ev, &c->post_done_ev, &c->post_set_ev[0], &c->post_set_ev[1], &c->post_set_ev[2]})
            if (!*ev) HIP_TRY(c, hipEventCreateWithFlags(ev, hipEventDisableTiming));
        // Steps 6-8 of this frame.  On the stream of this frame's trace where two traces run at a time (the next trace on that stream is
        // frame N + 2's, which must wait for this post stage anyway: it overwrites the G-buffer set the denoiser reads); on the second
        // trace stream where all traces share one (stage pipeline).  Post stages follow each other (one set of denoise buffers, one
        // exposure state); TAA of frame N + 1 waits until iteration 0 has read this frame's history.
        hipStream_t ps = overlap_scene ? ts : c->stream2 ? c->stream2 : ts;
        HIP_TRY(c, hipEventRecord(c->flight_taa_ev, c->taa_stream));
        HIP_TRY(c, hipStreamWaitEvent(ps, c->flight_taa_ev, 0));
        // Where two traces run at a time the post stages of consecutive frames are on different streams and run SIDE BY SIDE: a set of
        // denoise buffers and a device SDR array per frame parity (the read-back, 0.5 ms of PCIe for a 1920 x 540 console, is in nobody's
        // way either); only the exposure step waits for the frame before - its state passes from frame to frame (ToneMapper.cs:49-91).
        // Elsewhere the post stages follow each other.
        const bool side_by_side = overlap_scene && c->knobs.flight_post_pair;
        if (c->post_busy && !side_by_side) HIP_TRY(c, hipStreamWaitEvent(ps, c->post_done_ev, 0));
        rc = run_post(c, ps, out_sdr, false, c->post_hist_ev, overlap_scene ? c->post_done_ev : nullptr, overlap_scene && par == 1,
                      (side_by_side && c->post_busy) ? c->post_done_ev : nullptr, side_by_side && par == 1);
        if (rc != YCGE_OK) return rc;
        if (!overlap_scene) HIP_TRY(c, hipEventRecord(c->post_done_ev, ps));
        HIP_TRY(c, hipEventRecord(c->post_set_ev[k], ps));
        c->post_busy = true; c->post_hist_pending = true; c->post_set_pending[k] = true;
    }
    HIP_TRY(c, hipEventRecord(c->set_resolved_ev[k], c->taa_stream));
    c->set_read[k] = true;
    c->async_outstanding = true;
    return YCGE_OK;
}

int ycge_render_frame(ycge_ctx *c, float *out_sdr, ycge_frame_stats *st)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->parent) return c->fail(YCGE_ERR_INVALID_ARG, "peer contexts are driven by their root");
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    const bool multi_dev = !c->peers.empty() || c->exchange_mode == YCGE_EXCHANGE_RCCL;
    if (c->cfg.world_size != 1 && !multi_dev)
        return c->fail(YCGE_ERR_INVALID_ARG, "ycge_render_frame is the one-process entry: set config.n_devices / devices[] to drive several GPUs from it, "
                                             "or use ycge_trace_tiles + ycge_resolve_gathered with one process per GPU (rank / world_size)");
    HIP_TRY(c, hipSetDevice(c->device));
    auto t0 = std::chrono::steady_clock::now();
    FrameState fs;
    snapshot_frame(c, fs);
    bool did_reset = false;
    // (experiment builds, YCGE_TAA_FUSE=1: TemporalBlendWithClamp inside the trace launch where the frame is ONE launch on ONE device - ycge::TaaFuse; trace_frame decides and says so)
    c->fuse_done = false;
    c->fuse_request = c->knobs.taa_fuse && !multi_dev && c->cfg.taa_clamp_radius == 1;          // (the reference's call, RaytraceRenderer.cs:218: clampRadius 1 - the window the block resolve stages)
    if (c->fuse_request) taa_decide(c, fs, c->fuse_T, did_reset);
    int rc = multi_dev ? trace_on_all_devices(c, fs) : trace_frame(c, nullptr, c->stream, fs, true);
    c->fuse_request = false;
    if (rc != YCGE_OK) return rc;
    rc = taa_and_commit(c, c->stream, fs, did_reset, true, c->fuse_done);
    if (rc != YCGE_OK) return rc;
    if (out_sdr) {      // steps 6-8; with NULL the frame stops after TAA (trace-only callers, benchmarks of the hot path)
        rc = run_post(c, c->stream, out_sdr, true);
        if (rc != YCGE_OK) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    finish_staged_sdr(c);
    double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = fill_stats(c, st, fs, did_reset, true, wall);
    if (rc == YCGE_OK && st && c->cfg.count_work)
        for (ycge_ctx *p : c->peers) {          // the counters of the peers' tiles
            unsigned long long h[8];
            if (hipSetDevice(p->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess || copy_out(p, h, p->counters.p, sizeof h) != YCGE_OK) { (void)hipSetDevice(c->device); return c->fail(YCGE_ERR_DEVICE, "counter read-back failed on device %d", p->device); }
            st->n_rays += h[0]; st->n_box += h[1]; st->n_tri += h[2]; st->n_prim += h[3]; st->n_vox += h[4]; st->n_rays_dark += h[5];
        }
    if (multi_dev) HIP_TRY(c, hipSetDevice(c->device));
    if (rc == YCGE_OK && st && out_sdr) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
        st->post_ms = ms;
        float tone[3];
        { const int cr = copy_out(c, tone, c->tone_state.p, sizeof tone); if (cr != YCGE_OK) return cr; }
        st->exposure = tone[1];
        uint32_t n_serial; std::memcpy(&n_serial, &tone[2], 4);
        st->exposure_serial_chunks = (float)n_serial;
    }
    return rc;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_trace_tiles(ycge_ctx *c, void *d_slab, void *hip_stream, ycge_frame_stats *st)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->parent || !c->peers.empty()) return c->fail(YCGE_ERR_INVALID_ARG, "ycge_trace_tiles is the one-process-per-GPU form; this context drives its devices through ycge_render_frame");
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    float *slab = d_slab ? (float *)d_slab : c->own_slab.p;
    if (!slab) return c->fail(YCGE_ERR_INVALID_ARG, "no slab: pass a device pointer of ycge_tile_slab_bytes() bytes");
    if (c->pending.size() >= 8) c->pending.pop_front();       // a traced frame nobody resolved (trace-only callers) is abandoned once 8 newer ones wait
    auto t0 = std::chrono::steady_clock::now();
    FrameState fs;
    snapshot_frame(c, fs);
    int rc = trace_frame(c, slab, stream, fs, st != nullptr);
    if (rc != YCGE_OK) return rc;
    c->pending.push_back(fs);           // the matching ycge_resolve_gathered resolves THIS frame: its pose, its number
    if (st) {
        HIP_TRY(c, hipStreamSynchronize(stream));
        double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return fill_stats(c, st, fs, false, false, wall);
    }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_resolve_gathered(ycge_ctx *c, const void *d_all_slabs, void *hip_stream, float *out_sdr, ycge_frame_stats *st)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!d_all_slabs) return c->fail(YCGE_ERR_INVALID_ARG, "null gathered slabs");
    if (c->pending.empty()) return c->fail(YCGE_ERR_INVALID_ARG, "no traced frame to resolve: every ycge_resolve_gathered follows its own ycge_trace_tiles");
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    auto t0 = std::chrono::steady_clock::now();
    if (out_sdr && !c->cfg.slab_albedo) return c->fail(YCGE_ERR_INVALID_ARG, "lean slabs (config.slab_albedo = 0) carry no albedo: the denoise stage cannot run");
    const size_t per_rank = (size_t)c->tiles_per_rank_padded * 256 * slab_floats(c);
    if (st) HIP_TRY(c, hipEventRecord(c->ev[1], stream));
    int e = ycge_launch_unpermute((const float *)d_all_slabs, per_rank, c->hiW, c->hiH, c->tiles_x, c->n_tiles, c->cfg.world_size,
                                  (int)slab_floats(c), c->current_hdr.p, c->g_albedo.p, c->g_normal.p, c->g_depth.p, c->sky.p, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_unpermute launch failed: %s", hipGetErrorString((hipError_t)e));
    // the frame being resolved is the oldest traced one: its snapshot (pose, frame number) - NOT the live camera, which a
    // pipelined caller may already have moved on for the next trace (RaytraceRenderer.cs:159-176 snapshots once per frame)
    FrameState fs = c->pending.front();
    c->pending.pop_front();
    bool did_reset = false;
    int rc = taa_and_commit(c, stream, fs, did_reset, st != nullptr, false);
    if (rc != YCGE_OK) return rc;
    if (out_sdr) {
        rc = run_post(c, stream, out_sdr, st != nullptr);
        if (rc != YCGE_OK) return rc;
        if (!st) { HIP_TRY(c, hipStreamSynchronize(stream)); finish_staged_sdr(c); }      // the caller's host buffer is filled when the call returns
    }
    if (st) {
        HIP_TRY(c, hipStreamSynchronize(stream));
        finish_staged_sdr(c);
        std::memset(st, 0, sizeof *st);
        st->frame = fs.frame; st->history_reset = did_reset ? 1 : 0;
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
        st->taa_ms = ms;
        st->exposure = 1.0f;
        if (out_sdr) {
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
            st->post_ms = ms;
            float tone[2];
            { const int cr = copy_out(c, tone, c->tone_state.p, sizeof tone); if (cr != YCGE_OK) return cr; }
            st->exposure = tone[1];
        }
        st->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// ---- host-only helpers exported for the `-m "not gpu"` tests (no device needed) -------------
// Build a tree over caller-supplied boxes with the product builder: bounds = n*6 (min xyz, max xyz),
// centroids = n*3.  flavour 0 = scene, 1 = mesh.  Outputs are malloc'ed by the caller:
// nodes_out must hold 2*n records of 10 x 4 B, leaf_out n int32.  Returns node count or <0.
int ycge_host_build_tree(const float *bounds, const float *centroids, int32_t n, int32_t flavour, void *nodes_out, int32_t *leaf_out,
                         int32_t *stats_out /* [root, max_depth, sort_fallbacks] */)
try {
    if (n < 0 || (n > 0 && (!bounds || !centroids || !nodes_out || !leaf_out))) return YCGE_ERR_INVALID_ARG;
    BoundsSoA it;
    it.resize(n);
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { it.mn[a][i] = bounds[6 * i + a]; it.mx[a][i] = bounds[6 * i + 3 + a]; it.c[a][i] = centroids[3 * i + a]; }
    BuiltTree t;
    build_tree(it, flavour == 0 ? TreeFlavour::Scene : TreeFlavour::Mesh, t);
    if (!t.nodes.empty()) std::memcpy(nodes_out, t.nodes.data(), t.nodes.size() * sizeof(RefNode));
    if (!t.leaf_index.empty()) std::memcpy(leaf_out, t.leaf_index.data(), t.leaf_index.size() * 4);
    if (stats_out) { stats_out[0] = t.root; stats_out[1] = t.max_depth; stats_out[2] = t.sort_fallbacks; }
    return (int)t.nodes.size();
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// Same for a triangle soup (MeshBVH ctor path incl. TryComputeBounds).
// test hook: the device records of one mesh (single material 0) as emit_mesh_records lays them out
int ycge_host_mesh_arena(const float *tris9, int32_t n, void *out, int64_t capacity_bytes, uint32_t *root_ref_out)
try {
    if (n < 0 || (n > 0 && !tris9) || !root_ref_out) return YCGE_ERR_INVALID_ARG;
    BoundsSoA it;
    triangle_items(tris9, n, it);
    BuiltTree t;
    build_tree(it, TreeFlavour::Mesh, t);
    std::vector<uint8_t> arena;
    bool bad_leaf = false, bad_material = false;
    *root_ref_out = emit_mesh_records(t, tris9, nullptr, 0, 1, arena, bad_leaf, bad_material);
    if (bad_leaf || bad_material) return YCGE_ERR_UNSUPPORTED;
    if (out && (int64_t)arena.size() <= capacity_bytes && !arena.empty()) std::memcpy(out, arena.data(), arena.size());
    return (int)arena.size();
}
catch (...) { return ycge_host::abi_catch(nullptr); }

// ... with the treelet region of the cooperative walk appended (append_treelets): returns the total size, *tl_offset_out = where it starts
int ycge_host_mesh_arena_treelets(const float *tris9, int32_t n, void *out, int64_t capacity_bytes, uint32_t *root_ref_out, uint32_t *tl_offset_out)
try {
    if (n < 0 || (n > 0 && !tris9) || !root_ref_out || !tl_offset_out) return YCGE_ERR_INVALID_ARG;
    BoundsSoA it;
    triangle_items(tris9, n, it);
    BuiltTree t;
    build_tree(it, TreeFlavour::Mesh, t);
    std::vector<uint8_t> arena;
    bool bad_leaf = false, bad_material = false;
    GMesh gm;
    std::memset(&gm, 0, sizeof gm);
    gm.root_ref = *root_ref_out = emit_mesh_records(t, tris9, nullptr, 0, 1, arena, bad_leaf, bad_material);
    if (bad_leaf || bad_material) return YCGE_ERR_UNSUPPORTED;
    *tl_offset_out = append_treelets(arena, std::vector<GMesh>{gm});
    if (out && (int64_t)arena.size() <= capacity_bytes && !arena.empty()) std::memcpy(out, arena.data(), arena.size());
    return (int)arena.size();
}
catch (...) { return ycge_host::abi_catch(nullptr); }

int ycge_host_build_mesh(const float *tris9, int32_t n, void *nodes_out, int32_t *leaf_out, int32_t *stats_out)
try {
    if (n < 0 || (n > 0 && (!tris9 || !nodes_out || !leaf_out))) return YCGE_ERR_INVALID_ARG;
    BoundsSoA it;
    triangle_items(tris9, n, it);
    BuiltTree t;
    build_tree(it, TreeFlavour::Mesh, t);
    if (!t.nodes.empty()) std::memcpy(nodes_out, t.nodes.data(), t.nodes.size() * sizeof(RefNode));
    if (!t.leaf_index.empty()) std::memcpy(leaf_out, t.leaf_index.data(), t.leaf_index.size() * 4);
    if (stats_out) { stats_out[0] = t.root; stats_out[1] = t.max_depth; stats_out[2] = t.sort_fallbacks; }
    return (int)t.nodes.size();
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// Level schedule of an in-place A-trous iteration (host only).  pixels_out: w*h uint32, offsets_out: capacity
// uint32.  Returns the number of levels (offsets_out holds levels + 1 entries) or <0.
int ycge_host_inplace_schedule(int32_t w, int32_t h, int32_t step, uint32_t *pixels_out, uint32_t *offsets_out, int32_t capacity)
try {
    if (w <= 0 || h <= 0 || step <= 0 || !pixels_out || !offsets_out) return YCGE_ERR_INVALID_ARG;
    std::vector<uint32_t> px, off;
    build_inplace_schedule(w, h, step, px, off);
    if ((int64_t)off.size() > capacity) return YCGE_ERR_INVALID_ARG;
    std::memcpy(pixels_out, px.data(), px.size() * 4);
    std::memcpy(offsets_out, off.data(), off.size() * 4);
    return (int)off.size() - 1;
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// test hook: the banded pass lists of an in-place iteration as k_atrous_band reads them.  Returns the number of passes (entries = 32 x
// passes, x | y << 16 or 0xffffffff); offsets_out gets n_bands x (levels + 1) pass offsets; info_out = {levels, n_bands, max_level_pixels}
int ycge_host_inplace_bands(int32_t w, int32_t h, int32_t step, int32_t rows_per_band, uint32_t *entries_out, int64_t entries_capacity,
                            uint32_t *offsets_out, int64_t offsets_capacity, int32_t *info_out)
try {
    if (w <= 0 || h <= 0 || step <= 0 || rows_per_band <= 0 || !info_out) return YCGE_ERR_INVALID_ARG;
    std::vector<uint32_t> px, off, bpx, boff;
    build_inplace_schedule(w, h, step, px, off);
    int n_bands = 0;
    uint32_t max_level_pixels = 0;
    band_inplace_schedule(w, h, rows_per_band, px, off, bpx, boff, n_bands, max_level_pixels);
    info_out[0] = (int32_t)off.size() - 1; info_out[1] = n_bands; info_out[2] = (int32_t)max_level_pixels;
    if (entries_out && (int64_t)bpx.size() <= entries_capacity) std::memcpy(entries_out, bpx.data(), bpx.size() * 4);
    if (offsets_out && (int64_t)boff.size() <= offsets_capacity) std::memcpy(offsets_out, boff.data(), boff.size() * 4);
    return (int)(bpx.size() / 32);
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// test hook: the row-parity band layout of the persistent in-place A-trous launch (split_band_layout) and, per band, the most pixels a
// level holds.  row_band_out: h ints; desc_out: 8 ints a band (first row, rows, stride, groups, up0, up1, dn0, dn1); max_px_out: a band.
// Returns the number of bands, 0 where the layout does not apply.
int ycge_host_split_bands(int32_t w, int32_t h, int32_t step, int32_t *row_band_out, int32_t *desc_out, int32_t desc_capacity, int32_t *max_px_out)
try {
    if (w <= 0 || h <= 0 || step <= 0 || !row_band_out || !desc_out) return YCGE_ERR_INVALID_ARG;
    std::vector<int32_t> row_band, desc;
    int n_bands = 0;
    if (!split_band_layout(h, step, row_band, desc, n_bands)) return 0;
    if ((int32_t)desc.size() > desc_capacity) return YCGE_ERR_INVALID_ARG;
    std::memcpy(row_band_out, row_band.data(), row_band.size() * 4);
    std::memcpy(desc_out, desc.data(), desc.size() * 4);
    if (max_px_out) {
        std::vector<uint32_t> px, off;
        build_inplace_schedule(w, h, step, px, off);
        const int levels = (int)off.size() - 1;
        std::vector<int32_t> cnt((size_t)n_bands * levels, 0);
        for (int t = 0; t < levels; t++)
            for (uint32_t i = off[t]; i < off[t + 1]; i++) cnt[(size_t)row_band[px[i] / (uint32_t)w] * levels + t]++;
        for (int b = 0; b < n_bands; b++) { int m = 0; for (int t = 0; t < levels; t++) if (cnt[(size_t)b * levels + t] > m) m = cnt[(size_t)b * levels + t]; max_px_out[b] = m; }
    }
    return n_bands;
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// test hook: the window width run_post would hand k_atrous_band for this schedule (0 = hash form)
int ycge_host_band_window_width(int32_t w, int32_t h, int32_t step, int32_t rows_per_band, int32_t K, int32_t G)
try {
    if (w <= 0 || h <= 0 || step <= 0 || rows_per_band <= 0 || K <= 0 || (G != 8 && G != 16 && G != 32)) return YCGE_ERR_INVALID_ARG;
    std::vector<uint32_t> px, off, bpx, boff;
    build_inplace_schedule(w, h, step, px, off);
    int n_bands = 0;
    uint32_t max_level_pixels = 0;
    band_inplace_schedule(w, h, rows_per_band, px, off, bpx, boff, n_bands, max_level_pixels, (uint32_t)G);
    return (int)band_window_width(bpx, boff, n_bands, (int)off.size() - 1, K, rows_per_band, (uint32_t)G, 2048u);
}
catch (...) { return ycge_host::abi_catch(nullptr); }
// profiling aid: per-wavefront {start, end, node iterations, leaf phases} of the last counted k_wf_primary launch
int ycge_debug_read_wave_prof(ycge_ctx *c, unsigned long long *dst, size_t n_u64)
try {
    if (!c || !dst || !c->wave_prof.p || n_u64 > c->wave_prof.n) return YCGE_ERR_INVALID_ARG;
    HIP_TRY(c, hipDeviceSynchronize());
    return copy_out(c, dst, c->wave_prof.p, n_u64 * 8);
}
catch (...) { return ycge_host::abi_catch(c); }
// sizeof of each ABI struct, for the ctypes mirror check
size_t ycge_abi_sizeof(int32_t which)
try {
    switch (which) {
    case 0: return sizeof(ycge_vec3); case 1: return sizeof(ycge_material); case 2: return sizeof(ycge_prim); case 3: return sizeof(ycge_mesh);
    case 4: return sizeof(ycge_voxel_lookup); case 5: return sizeof(ycge_grid); case 6: return sizeof(ycge_light); case 7: return sizeof(ycge_scene);
    case 8: return sizeof(ycge_config); case 9: return sizeof(ycge_frame_stats); case 10: return sizeof(ycge_flight_info);
    }
    return 0;
}
catch (...) { (void)ycge_host::abi_catch(nullptr); return 0; }

// tests: what copy_out / run_post / ycge_render_frame_async_sdr decide about a destination (1: the device may write [p, p + bytes) directly)
int ycge_debug_is_page_locked(const void *p, size_t bytes)
try {
    return (p && host_memory_is_page_locked(p, bytes)) ? 1 : 0;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

// tests (tests/test_host_cpu.py, no GPU needed): an exception of the given kind raised INSIDE an exported body - what comes back is the
// barrier's answer (abi_catch): 1 std::bad_alloc, 2 std::runtime_error, 3 something that is no std::exception, 4 std::length_error out of a
// vector, 5 std::system_error as a failing std::thread constructor raises it; 0 nothing
int ycge_debug_throw(ycge_ctx *c, int32_t kind)
try {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::runtime_error("requested by ycge_debug_throw");
    if (kind == 3) throw 42;
    if (kind == 4) { std::vector<uint64_t> v; v.resize(v.max_size() + (size_t)kind); }
    if (kind == 5) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread");
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

#if YCGE_FAULT_INJECTION
// lib/var_faultinject.so only (tests/test_gpu_abi_barrier.py): the n-th allocation from now fails.  The replaced operators serve this
// library's own code (every std::vector, std::string and `new` of the host side is compiled into it; -Bsymbolic binds them here).
int ycge_debug_fail_allocation(int64_t nth)
{
    const long long left = g_fail_alloc_countdown.exchange((long long)nth);
    return left < 0 ? -1 : (int)(left > 0x7fffffff ? 0x7fffffff : left);
}
#endif

} // extern "C"
#if YCGE_FAULT_INJECTION && !defined(__HIP_DEVICE_COMPILE__)
// (the variant is linked -Wl,-Bsymbolic: the library's own references bind to these definitions, whatever else the process has loaded)
static void *fi_alloc(std::size_t n)
{
    if (g_fail_alloc_countdown.load(std::memory_order_relaxed) >= 0 && g_fail_alloc_countdown.fetch_sub(1) == 0) throw std::bad_alloc();
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(std::size_t n) { return fi_alloc(n); }
void *operator new[](std::size_t n) { return fi_alloc(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
#endif
