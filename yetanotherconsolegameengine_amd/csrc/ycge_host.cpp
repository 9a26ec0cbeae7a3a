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
size_t slab_floats(const ycge_ctx *c) { return c->cfg.slab_albedo ? (size_t)YCGE_SLAB_FLOATS : (size_t)YCGE_SLAB_FLOATS - 3; }

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

static RcclApi g_rccl;
const RcclApi &load_rccl()
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
        peer->worker->th = std::thread(ycge_peer_worker_main, root, peer);
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
    bool any_grid_object = false;          // some object of the list IS a voxel grid (the scene's grid TABLE may hold grids no object uses: their entity left)
    float walk_t_limit = 0.0f;             // the smallest GGrid::cull_t_limit of the grids in Objects
    bool analytic_only = true;             // no mesh and no voxel grid among the objects (SceneDev::analytic_only)
};

// Scene.Objects as device records + what Scene.RebuildBVH gets from every object's TryGetBounds (BVH.cs:32-53); no tree yet
int flatten_objects(ycge_ctx *c, const ycge_prim *prims, int n_prims, ObjectsHost &oh, BoundsSoA &items)
{
    std::vector<GPrim> &gprims = oh.gprims;
    gprims.assign(n_prims, GPrim{});
    items.resize(n_prims);
    oh.grid_owner.assign(c->grid_solid.size(), -1); oh.grid_owner_unique = true; oh.any_grid_object = false; oh.walk_t_limit = HUGE_VALF;
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
            oh.any_grid_object = true;
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
// the same grid (grid_owner would be ambiguous), when NO object holds a grid although the scene's table has some (every voxel entity left
// through ycge_scene_update_objects, or the upload listed grids nothing uses: the objects are then analytic_only, and a walk tree beside
// that flag sent the timed k_trace out of bounds - a GPU memory fault found by the drawn call sequences of round 6) and under YCGE_NO_WALK_TREE.
int install_walk_tree(ycge_ctx *c, const ObjectsHost &oh, int n_inner, uint32_t scene_root)
{
    SceneDev &sd = c->sd;
    sd.walk_nodes = nullptr; sd.grid_owner = nullptr; sd.walk_root_ref = YCGE_REF_NONE_VALUE; sd.walk_t_limit = 0.0f; c->walk_scene_nodes = 0;
    if (!c->has_grid || !oh.any_grid_object || n_inner <= 0 || c->knobs.no_walk_tree || !oh.grid_owner_unique || YCGE_REF_KIND(scene_root) != REF_SCENE_NODE) return YCGE_OK;
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

extern "C" {

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
