// ycge_ctx.h - the context of the library (struct ycge_ctx), the launch entry points of the kernel files, and what the host translation
// units share (internal: the C-ABI is include/ycge.h).
//   ycge_host.cpp      context, scene flattening and upload, frame orchestration (TryFlipAndBlit steps 1-9), frames in flight, the slab
//                      form of the tiled frame, the post stage's schedule
//   ycge_resident.cpp  the tile-resident multi-GPU form, its batched launches and emulation loop; read-backs (ycge_read_buffer / _accel)
//   ycge_accel.cpp     the bit-faithful BVH builders
// All device work is in the .hip files; there is no CPU implementation of any per-pixel stage.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <string>
#include <vector>
#include <functional>

#include "../../include/ycge.h"
#include "ycge_accel.h"
#include "ycge_device.h"
#include "ycge_math.h"

extern "C" {
size_t ycge_wf_sizes(int which);
int ycge_launch_trace(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int count, int flat, int refill_steps,
                      hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
int ycge_launch_wavefront(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, void *const bufs[7], int rounds,
                          int has_grid, int flat, int count, int persistent_waves, hipStream_t stream, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join,
                          const ycge::TraceOut *O_side);
int ycge_launch_trace_batch(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int n, int count, int flat, hipStream_t stream);
int ycge_launch_scene_walk(const void *nodes, int n_inner, const uint32_t *leaf_prims, const void *prims, void *walk, hipStream_t stream);
int ycge_launch_order_blocks(uint32_t *cost, uint32_t n, uint32_t policy, uint32_t split_top, uint32_t fan_class, uint32_t fan_cap, uint32_t next_slot, uint32_t skip_mask, uint32_t *order_ws,
                             uint32_t *order, hipStream_t stream, int small_groups = 0, uint32_t n_frames = 0, uint32_t *snap = nullptr);
int ycge_launch_taa_tiles(const ycge::TaaParams *T, const ycge::FrameParams *P, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                          float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, float *slab, hipStream_t stream);
int ycge_launch_resolve_tiles(const ycge::TaaParams *T, const ycge::FrameParams *P, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                              const void *records, const uint32_t *halo_index, float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, float *slab, hipStream_t stream);
int ycge_launch_halo(int scatter, float *hdr, uint8_t *sky, const uint32_t *px, uint32_t n, void *records, hipStream_t stream);
int ycge_launch_pack_history(const ycge::FrameParams *P, const float *hist, float *slab, hipStream_t stream);
int ycge_launch_unpack_history(const float *all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size, float *hist, hipStream_t stream);
int ycge_launch_trace_fan(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int count, int flat, uint32_t fan_cap,
                          hipStream_t stream);
int ycge_launch_taa(const ycge::TaaParams *T, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                    float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, hipStream_t stream, int small_groups = 0, hipEvent_t stop = nullptr);
size_t ycge_post_state_bytes(void);
int ycge_atrous_persist_resident(int groups_per_pass, int split, int level_handover, int profile);
void ycge_atrous_duo_pad_lds(int bytes);
int ycge_launch_unit_normals(const float *normal, float *unit, size_t n, hipStream_t stream);
int ycge_launch_atrous(int w, int h, int step, const float phi[4], const float *cur, float *dst, const float *albedo, const float *unit_n,
                       const float *depth, const uint8_t *sky, hipStream_t stream);
int ycge_launch_atrous_static(int w, int h, int step, const float phi[4], const float *albedo, const float *unit_n, const float *depth,
                              const uint8_t *sky, float *statw, hipStream_t stream);
int ycge_launch_atrous_inplace(int w, int h, int step, const float phi[4], float *buf, const float *albedo, const float *unit_n,
                               const float *depth, const uint8_t *sky, float *statw, const uint32_t *d_pixels, const uint32_t *d_offsets,
                               int n_levels, int n_bands, int K, int groups_per_pass, int rows_per_band, unsigned window_width, hipStream_t stream);
int ycge_launch_atrous_persist(int w, int h, int step, const float phi[4], float *buf, const uint8_t *sky, float *statw, const uint32_t *d_pixels,
                               const uint32_t *d_offsets, const uint32_t *d_pass_level, const int32_t *d_band_desc, int n_levels, int n_bands, int K, int groups_per_pass, int rows_per_band, unsigned window_width,
                               uint32_t *progress, uint32_t epoch, int xcd_local, int level_handover, int profile, uint32_t ticket_base, hipStream_t stream);
size_t ycge_exposure_scratch_bytes(int w, int h, int step);
size_t ycge_bvh_build_scratch_bytes(int n);
int ycge_launch_scene_bvh_build(const float *items, int n, void *scratch, void *ref_out, void *gnodes_out, uint32_t *leaf_out, void *result,
                                int active_waves, hipStream_t stream);
int ycge_launch_exposure(const float *hdr, const uint8_t *sky, int w, int h, int step, float *terms, void *state, const float consts[5],
                         void *scratch, int serial, hipStream_t stream);
int ycge_launch_tonemap(const float *hdr, int hiW, int fbW, int fbH, int ss, float gamma, float saturation, float vibrance, const void *state,
                        float *out, hipStream_t stream);
int ycge_launch_pack_slab(const ycge::FrameParams *P, const float *hdr, const float *albedo, const float *normal, const float *depth,
                          const uint8_t *sky, float *slab, int slab_floats, hipStream_t stream);
int ycge_launch_push_tiles(const ycge::FrameParams *P, const ycge::PushPlanes *planes, hipStream_t stream);
int ycge_launch_unpermute(const float *all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size,
                          int slab_floats, float *hdr, float *albedo, float *normal, float *depth, uint8_t *sky, hipStream_t stream);
}

using namespace ycge;

namespace ycge_host {

inline std::string g_create_error;          // (one per library: C++17 inline variable)
// THE EXCEPTION BARRIER of the C-ABI (round 6): every exported function's body is a function-try-block whose handler ends here, so that no
// C++ exception - std::bad_alloc from a vector that flattens an 871 200-triangle mesh, std::system_error from a thread constructor - unwinds
// into the P/Invoke frame of the CLR host ("no exceptions/longjmp across the ABI").  Called INSIDE a catch (...) handler:
// rethrows to classify.  std::bad_alloc -> YCGE_ERR_OUT_OF_MEMORY, anything else -> YCGE_ERR_INTERNAL; the text goes to ycge_last_error.
int abi_catch(const ycge_ctx *c) noexcept;

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;          // elements in use / elements allocated
    void release() { if (p) { (void)hipFree(p); p = nullptr; } n = cap = 0; }
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) return hipSuccess;
        const hipError_t e = hipMalloc((void **)&p, count * sizeof(T) + 64);     // records are read with whole 64- / 72-byte fetches: room for the over-read past the last record
        if (e != hipSuccess) { p = nullptr; return e; }
        n = cap = count;
        return hipSuccess;
    }
    // room for `count` elements, contents undefined; the allocation is kept when it is large enough
    hipError_t reserve(size_t count)
    {
        if (count > cap || cap == 0) { const hipError_t e = alloc(count > 0 ? count : 1); if (e != hipSuccess) return e; }
        n = count;
        return hipSuccess;
    }
    // per-frame callers (lights, moved objects) reuse the allocation when the new contents fit
    hipError_t upload(const std::vector<T> &v)
    {
        if (v.size() > cap || (v.empty() && cap == 0)) {
            const hipError_t e = alloc(v.size());
            if (e != hipSuccess) return e;
        }
        n = v.size();
        if (v.empty()) return hipSuccess;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// Experiment knobs (none changes a pixel): read ONCE, when the context is created.
struct Knobs {
    int path_policy = 0;             // YCGE_PATH: 0 auto, 1 wavefront, 2 single launch
    bool xcd_strips = false, generic_walk = false, no_lpt = false, no_refill = false;
    int wave_prof_stage = -1;        // YCGE_WAVE_PROF: -1 off, 0 primary, 1 extend, 2 mega
    int refill_steps = YCGE_REFILL_STEPS_DEFAULT;
    bool split_set = false; uint32_t split_policy = 0;
    int split_top_lg = 2;                        // YCGE_SPLIT_TOP_LG: log2 of the parts such a block goes in (2 = 4 parts of 16 pixels)
    bool split_top_set = false;                  // YCGE_SPLIT_TOP / YCGE_SPLIT_TOP_LG given: they decide; else schedule_policy picks by the frame's block count
    int split_top = YCGE_SPLIT_TOP_DEFAULT;     // YCGE_SPLIT_TOP: this many blocks at the head of the schedule go in 4 parts of 16 pixels (0 = none)
    int pw_per_cu = 32;
    int post_band_rows = YCGE_POST_BAND_ROWS_DEFAULT, post_k = YCGE_POST_K_DEFAULT, post_groups = YCGE_POST_GROUPS_DEFAULT;
    int fan_class = -1, fan_cap = -1;   // -1 = default by world size
    bool split_resolve = false;      // YCGE_RES_SPLIT_RESOLVE=1: the tile-resident resolve as round 5's two launches (k_scatter_halo, k_taa_tiles) instead of k_resolve_tiles (A/B)
    bool taa_fuse = false;           // YCGE_TAA_FUSE=1: the synchronous single-launch frame resolves TAA inside the trace launch - experiment builds only (csrc/experiments/ycge_taa_in_trace.hip.h: bit-exact, slower)
    int post_mode = 0;               // YCGE_POST_MODE: in-place A-trous: 0 = one persistent launch, level-granular hand-over (k_atrous_stream), 2 = a launch per level group, 3 = as 0 with bands in block order, 4 = persistent with group hand-over (k_atrous_persist)
    bool post_no_split = false;      // YCGE_POST_NO_SPLIT: whole bands in the persistent in-place A-trous (no row-parity half-bands)
    int post_probe_band = -1;        // YCGE_POST_PROBE_BAND: this band and the next record a per-pass timeline (profiles/post_bands.py)
    int post_assume_resident = 0;    // YCGE_POST_ASSUME_RESIDENT (tests): take this for the runtime's answer - more bands than fit, to exercise the order-of-arrival numbering
    bool post_dbg_free = false;      // YCGE_POST_DBG_FREE (timing experiment, WRONG pixels): no band of the persistent in-place A-trous waits for the band above
    bool flight_small_groups = true; // YCGE_FLIGHT_SMALL_GROUPS: TAA and schedule kernels of the frames in flight in small workgroups (they find room beside a running trace)
    int flight_priority = 1;         // YCGE_FLIGHT_PRIORITY: the second stream's priority: 1 highest, 0 normal, -1 lowest
    bool flight_post_pair = true;    // YCGE_FLIGHT_POST_PAIR: the post stages of consecutive frames in flight side by side (second set of denoise buffers)
    bool flight_placed_gate = true;  // YCGE_FLIGHT_PLACED_GATE: a frame in flight traces once the trace before it has placed its last workgroup (a value that kernel stores)
    bool flight_post_gate = true;    // YCGE_FLIGHT_POST_GATE: a frame in flight traces only once the post stage before it has passed its first iteration
    bool flight_overlap = true;      // YCGE_FLIGHT_OVERLAP: frames in flight alternate between two trace streams (two traces may overlap)
    bool flight_no_begin = false;      // experiments on ycge_render_frame_async: second stream at normal priority; no begin-of-trace timing event
    int post_pad_lds = 0;            // YCGE_POST_PAD_LDS (experiment): bytes of unused LDS per band workgroup of the two-set form - fewer of them on a CU
    int post_resident_per_cu = 3;    // YCGE_POST_RESIDENT: band workgroups of the persistent in-place A-trous a CU may hold (3 fit: 576 threads, 46 KB of LDS each)
    bool post_hash = false;          // YCGE_POST_HASH_FORM=1: the hash form of k_atrous_band even where the window fits
    int bvh_waves = 16;              // YCGE_BVH_WAVES: wavefronts of k_scene_bvh_build that take nodes (tests: the order nodes are split in must not matter)
    bool scene_bvh_host = false;     // YCGE_SCENE_BVH_HOST: ycge_scene_update_objects builds the scene BVH on the host, not on the device
    int scene_bvh_device_min = YCGE_BVH_DEV_MIN_ITEMS_DEFAULT;   // YCGE_SCENE_BVH_DEVICE_MIN: fewer objects than this are built on the host (measured crossover, profiles/r02/f2_update_objects_timing.txt)
    int res_sched_every = 0;         // YCGE_RES_SCHED_EVERY: the tile-resident ring builds a new schedule behind every n-th frame (0 = the ring's depth)
    int bfs_rays = 0;                // YCGE_BFS=<n>: a wavefront's occlusion queries against a mesh go breadth-first from one shared work list when at most n of its lanes ask (mesh_anyhit_bfs; 0 = never, 64 = always)
    bool no_flight_stage_overlap = false;   // YCGE_NO_FLIGHT_STAGE_OVERLAP: frames in flight of the stage pipeline (voxel worlds) one trace at a time (A/B)
    bool no_lights_beside = false;   // YCGE_NO_LIGHTS_BESIDE: the stage pipeline strictly in sequence (A/B of the light loop beside the next round's trace)
    bool lpt_always = false;         // YCGE_LPT_ALWAYS: the longest-first schedule also for frames whose blocks are all resident at once
    int persist_min_tiles = -1;      // YCGE_PERSIST_MIN_TILES: frames of fewer tiles take k_wf_extend instead of the persistent extend stage (-1: a quarter of the persistent wavefronts)
    bool no_analytic_walk = false;   // YCGE_NO_ANALYTIC_WALK: scenes of analytic objects only are walked by tree_phase's general loop (A/B of analytic_walk; same pixels)
    bool no_walk_tree = false;       // YCGE_NO_WALK_TREE: voxel worlds are walked down the scene tree, leaves and object steps and all (A/B of SceneDev::walk_nodes)
    bool no_coop = false;            // YCGE_NO_COOP: no treelets are built, sparse wavefronts keep the regular walk (A/B of the cooperative walk)
    bool exposure_serial = false;    // YCGE_EXPOSURE_SERIAL: the one-lane chain instead of the chunked exact evaluation
    void read()
    {
        auto geti = [](const char *n, int dflt) { const char *e = getenv(n); return e ? atoi(e) : dflt; };
        if (const char *e = getenv("YCGE_PATH")) path_policy = e[0] == 'w' ? 1 : e[0] == 'm' ? 2 : 0;
        xcd_strips = getenv("YCGE_XCD_STRIPS") != nullptr; generic_walk = getenv("YCGE_GENERIC_WALK") != nullptr;
        no_analytic_walk = getenv("YCGE_NO_ANALYTIC_WALK") != nullptr;
        if (const char *e = getenv("YCGE_PERSIST_MIN_TILES")) persist_min_tiles = atoi(e);
        lpt_always = getenv("YCGE_LPT_ALWAYS") != nullptr;
        no_walk_tree = getenv("YCGE_NO_WALK_TREE") != nullptr; no_lights_beside = getenv("YCGE_NO_LIGHTS_BESIDE") != nullptr; no_flight_stage_overlap = getenv("YCGE_NO_FLIGHT_STAGE_OVERLAP") != nullptr;
        no_lpt = getenv("YCGE_NO_LPT") != nullptr; no_refill = getenv("YCGE_NO_REFILL") != nullptr;
        if (const char *e = getenv("YCGE_WAVE_PROF")) wave_prof_stage = e[0] == 'e' ? 1 : e[0] == 'm' ? 2 : 0;
        refill_steps = geti("YCGE_REFILL", YCGE_REFILL_STEPS_DEFAULT);
        if (const char *e = getenv("YCGE_SPLIT")) { split_set = true; split_policy = (uint32_t)strtoul(e, nullptr, 8); }
        split_top_set = getenv("YCGE_SPLIT_TOP") != nullptr || getenv("YCGE_SPLIT_TOP_LG") != nullptr;
        split_top = geti("YCGE_SPLIT_TOP", YCGE_SPLIT_TOP_DEFAULT);
        if (split_top < 0) split_top = 0;
        split_top_lg = geti("YCGE_SPLIT_TOP_LG", 2);
        if (split_top_lg < 1 || split_top_lg > 6) split_top_lg = 2;
        pw_per_cu = geti("YCGE_PW_PER_CU", 32);
        post_band_rows = geti("YCGE_POST_BAND_ROWS", YCGE_POST_BAND_ROWS_DEFAULT); post_k = geti("YCGE_POST_K", YCGE_POST_K_DEFAULT);
        post_groups = geti("YCGE_POST_GROUPS", YCGE_POST_GROUPS_DEFAULT);
        if (post_groups != 8 && post_groups != 16 && post_groups != 32) post_groups = YCGE_POST_GROUPS_DEFAULT;
        fan_class = geti("YCGE_FAN", -1); fan_cap = geti("YCGE_FAN_CAP", -1);
        taa_fuse = YCGE_EXPERIMENTS && geti("YCGE_TAA_FUSE", 0) != 0;
        split_resolve = geti("YCGE_RES_SPLIT_RESOLVE", 0) != 0;
        post_mode = geti("YCGE_POST_MODE", 0);
        post_hash = geti("YCGE_POST_HASH_FORM", 0) != 0;
        post_no_split = getenv("YCGE_POST_NO_SPLIT") != nullptr;
        post_probe_band = geti("YCGE_POST_PROBE_BAND", -1);
        post_resident_per_cu = geti("YCGE_POST_RESIDENT", 3);
        post_assume_resident = geti("YCGE_POST_ASSUME_RESIDENT", 0);
        if (post_resident_per_cu < 1 || post_resident_per_cu > 3) post_resident_per_cu = 3;
        post_pad_lds = geti("YCGE_POST_PAD_LDS", 0);
        flight_overlap = geti("YCGE_FLIGHT_OVERLAP", 1) != 0;
        flight_post_gate = geti("YCGE_FLIGHT_POST_GATE", 1) != 0;
        flight_placed_gate = geti("YCGE_FLIGHT_PLACED_GATE", 1) != 0;
        flight_post_pair = geti("YCGE_FLIGHT_POST_PAIR", 1) != 0;
        flight_small_groups = geti("YCGE_FLIGHT_SMALL_GROUPS", 1) != 0;
        flight_priority = geti("YCGE_FLIGHT_PRIORITY", 1);
        flight_no_begin = geti("YCGE_FLIGHT_NO_BEGIN", 0) != 0;
        post_dbg_free = geti("YCGE_POST_DBG_FREE", 0) != 0;
        exposure_serial = getenv("YCGE_EXPOSURE_SERIAL") != nullptr;
        no_coop = getenv("YCGE_NO_COOP") != nullptr;
        res_sched_every = geti("YCGE_RES_SCHED_EVERY", 0);
        bfs_rays = geti("YCGE_BFS", 0);
        if (bfs_rays < 0 || bfs_rays > 64) bfs_rays = 0;
        scene_bvh_host = getenv("YCGE_SCENE_BVH_HOST") != nullptr;
        bvh_waves = geti("YCGE_BVH_WAVES", 16);
        scene_bvh_device_min = geti("YCGE_SCENE_BVH_DEVICE_MIN", YCGE_BVH_DEV_MIN_ITEMS_DEFAULT);
#if !YCGE_EXPERIMENTS
        // the kernel forms of csrc/experiments/ (k_trace_refill, the group hand-over A-trous) are not in this build
        refill_steps = 0;
        if (post_mode == 4) post_mode = 0;
#endif
    }
};

// One frame's identity from snapshot to commit (TryFlipAndBlit steps 1-3, RaytraceRenderer.cs:159-176): the pose the
// frame is traced with is the pose its reset decision and CommitCamera use.
struct FrameState {
    float pos[3], yaw, pitch, fov;
    bool reset;
    int64_t frame;
    uint32_t fan_blocks;
    bool scheduled = false;      // the trace ran the single-launch kernel with a longest-first schedule (cost ring in use)
    bool single_launch = false;  // the trace ran the single-launch kernel (its last workgroup stores the placed value), scheduled or not
};

struct MeshHost {
    BuiltTree tree;
};

} // namespace ycge_host
using namespace ycge_host;

struct ycge_ctx {
    ycge_config cfg;
    Knobs knobs;
    std::string err;
    int device = 0;
    // one process, several GPUs (config.n_devices >= 2): this context is rank 0 and owns one context per further device
    std::vector<ycge_ctx *> peers;
    ycge_ctx *parent = nullptr;
    hipEvent_t pushed_ev = nullptr;            // a peer's tiles have arrived in the parent's frame buffers
    // config.multi_device_exchange = YCGE_EXCHANGE_RCCL (root only): the in-process communicators (ncclCommInitAll over devices[]), rank r's
    // on device r's stream; exchange_mode says what the frames really use (0 = peer push: not asked for, or librccl.so / its symbols absent)
    int exchange_mode = 0;
    std::vector<void *> nccl_comms;
    DevBuf<float> all_slabs;                   // every context of an RCCL frame: the gathered slabs of all ranks (the root un-permutes its copy)
    // A peer's share of a frame is ISSUED by a thread of its own (trace launches, tile push, event): eight devices driven one after the
    // other from the caller's thread would put 7 x ~0.1 ms of launch calls in front of the last device's first kernel - as long as
    // the frame itself.  The worker sleeps between frames; the root posts a frame, issues its own share, then collects the peers'.
    struct PeerWorker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        int job = 0;                           // 0 idle, 1 frame posted, 2 done, -1 quit
        FrameState fs{};
        int rc = 0;
    } *worker = nullptr;
    std::deque<FrameState> pending;            // frames traced by ycge_trace_tiles and not yet resolved (pipelined callers)
    hipStream_t last_stream = nullptr;         // the stream the last tiled call ran on (scene updates wait for it too)
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // k_trace runs beside k_trace_fan on a side stream, forked from and joined to the frame's stream
    hipStream_t fan_stream = nullptr;
    hipEvent_t fan_ev[2] = {nullptr, nullptr};
    hipEvent_t traced_ev = nullptr, order_ev = nullptr;   // the next frame's schedule is built on the side stream, beside TAA
    bool order_pending = false;
    uint32_t fan_class = 0, fan_cap = 0;       // schedule classes >= fan_class are fanned, at most fan_cap blocks (0 = off)
    uint32_t *h_n_fan = nullptr;               // pinned: how many entries the last finished schedule gave k_trace_fan (read without waiting)
    char device_name[256] = {0};
    int compute_units = 0;

    // geometry of the trace grid
    int fbW = 0, fbH = 0, ss = 1, hiW = 0, hiH = 0;
    int tiles_x = 0, tiles_y = 0, n_tiles = 0, n_owned = 0, tiles_per_rank_padded = 0;

    // camera (lock(camLock), RaytraceRenderer.cs:142-147)
    std::mutex cam_lock;
    float cam_pos[3] = {0.0f, 1.0f, 0.0f};
    float yaw = 0.0f, pitch = 0.0f, fov_deg = 45.0f;

    int64_t frame_counter = 0;                 // RaytraceRenderer.cs:24
    // TemporalAA camera memory (TemporalAA.cs:11-15) and history validity
    float last_cam[3] = {NAN, NAN, NAN}, last_yaw = NAN, last_pitch = NAN;
    bool taa_valid = false;

    // per-pixel buffers in HBM (row-major, x + y*hiW)
    DevBuf<float> current_hdr, g_albedo, g_normal, g_depth, taa_hist, prev_normal, prev_depth;
    DevBuf<uint8_t> sky, prev_sky;
    // tiled frame: the trace writes its tiles here (same full-frame indexing), ycge_resolve_gathered writes the buffers above -
    // so the trace of frame N+1 may run beside the all-gather and resolve of frame N (two streams, caller-ordered)
    DevBuf<float> t_hdr, t_albedo, t_normal, t_depth;
    DevBuf<uint8_t> t_sky;
    // frames in flight (ycge_render_frame_async): the trace of frame N + 1 runs beside the TAA of frame N, so a frame's trace outputs
    // alternate between the five buffers above and these (swapped before the trace: the names above are always the newest frame's)
    DevBuf<float> alt_hdr, alt_albedo, alt_normal, alt_depth;
    DevBuf<uint8_t> alt_sky;
    DevBuf<float> alt2_hdr, alt2_albedo, alt2_normal, alt2_depth;      // (three sets: the trace of frame N + 1 must not wait for the TAA of frame N - 1, which finds
    DevBuf<uint8_t> alt2_sky;                                          //  its places among frame N's wavefronts late; it waits for TAA of frame N - 2)
    int set_id[3] = {0, 1, 2};                     // which of the three sets the names current / alt / alt2 hold
    hipStream_t taa_stream = nullptr, stream2 = nullptr;      // stream2: the traces of odd frames in flight (two traces may overlap: the tail of one, the bulk of the next)
    DevBuf<uint64_t> stack_spill2;                 // ... which then need a traversal-stack spill area of their own
    uint64_t *spill_override = nullptr;            // set around trace_frame by ycge_render_frame_async
    hipEvent_t flight_fork_ev = nullptr;
    uint32_t *placed_flag = nullptr;               // signal memory: the number of the newest frame in flight whose trace has placed its last workgroup
    uint32_t placed_expect = 0, placed_next = 0;   // what the next trace waits for (0: nothing) / the value the next trace stores
    uint64_t placed_waits = 0;                     // traces queued behind a placed value so far (ycge_flight_query)
    // frames in flight WITH the post stage (ycge_render_frame_async_sdr): post of frame N beside the traces and TAA of the frames after it
    hipEvent_t flight_taa_ev = nullptr, post_hist_ev = nullptr, post_done_ev = nullptr, post_set_ev[3] = {nullptr, nullptr, nullptr};
    bool post_hist_pending = false, post_busy = false, post_set_pending[3] = {false, false, false};
    hipEvent_t tile_trace_ev[2] = {nullptr, nullptr};      // tiled frames: the trace (and slab pack) of the newest frame of each parity is done
    bool tile_trace_used[2] = {false, false};
    hipEvent_t set_resolved_ev[3] = {nullptr, nullptr, nullptr};
    bool set_read[3] = {false, false, false};             // a TAA launch on taa_stream has read this set: the next trace into it waits for set_resolved_ev
    int out_set = 0;                               // which set the names above hold
    bool async_outstanding = false;
    // ... and their schedules: the one for frame N + 1 is built WHILE frame N is traced, from the costs up to frame N - 1 (a frame
    // staler than the synchronous path's, which builds it between the two traces), into the buffers frame N is not reading
    DevBuf<uint32_t> flight_order[3], flight_ws[3];         // (frames in flight use all three, by frame number mod 3; tiled frames two, by parity)
    int64_t flight_order_frame[3] = {-1, -1, -1};  // the frame number each buffer's schedule was built for (-1: none)
    hipEvent_t flight_order_ev[3] = {nullptr, nullptr, nullptr};     // the schedule in each buffer is complete (side stream)
    int64_t last_frame_deferred = -2;              // the newest tiled frame whose trace was followed by a deferred schedule
    bool in_flight_taa = false;                    // taa_and_commit is called by ycge_render_frame_async with two traces overlapping: one-wavefront workgroups
    bool in_flight_call = false;                   // trace_frame is called by ycge_render_frame_async
    std::vector<hipEvent_t> flight_ev;             // begin / end of the trace launches of the frames in flight, a ring (ycge_async_trace_times)
    uint64_t flight_frames = 0;                    // queued since the last ycge_async_trace_times
    // ---- tile-resident form (one process per GPU; ycge_trace_tiles_resident / ycge_resolve_tiles_resident): TAA on this rank's own tiles
    // with a one-pixel halo of {hdr, sky} exchanged between the ranks, the history never leaves its rank; a ring of K frame sets so that K
    // tiled traces may be in flight (a rank's launch is its longest chains: its period per frame becomes max(slot time, chain / K))
    struct ResidentSet {
        DevBuf<float> hdr, normal, depth;
        DevBuf<uint8_t> sky;
        DevBuf<uint64_t> spill;
        hipEvent_t traced = nullptr, resolved = nullptr;
        bool traced_used = false, resolved_used = false;
    };
    std::vector<ResidentSet *> rsets;
    // ycge_trace_tiles_resident_batch: the frames of a batch leave their launch parameters here instead of launching (trace_frame), one
    // launch traces them all (the records travel as its arguments)
    bool batch_collect = false;
    std::vector<FrameParams> batch_P;
    std::vector<TraceOut> batch_O;
    static constexpr int kBatchMax = YCGE_TRACE_BATCH_MAX;
    DevBuf<uint64_t> batch_spill[2];               // a spill area as wide as the batch's frames together, per batch parity: two batches may run at a time
    uint64_t batch_count = 0;
    hipEvent_t batch_done[2] = {nullptr, nullptr}; // a batch's launch: the batch after the next may scratch its spill area after it
    bool batch_spill_used[2] = {false, false};
    static constexpr uint32_t kResCostFrames = 16; // the resident ring's own cost ring: K - 1 slots are being written, one is cleared, the rest are read
    DevBuf<uint32_t> res_cost;
    // three schedule buffers taken in turn: one is built behind the trace of every R-th frame M (R = the ring's depth; YCGE_RES_SCHED_EVERY) and
    // serves the frames from M + K on - a trace never waits for a trace younger than frame N - K - until a newer one does
    std::vector<DevBuf<uint32_t> *> res_order, res_ws;
    std::vector<hipEvent_t> res_order_ev, res_order_read_ev;
    std::vector<int64_t> res_order_frame;          // per buffer: the frame M its schedule was built behind (-1: none)
    int res_order_next = 0;                        // the buffer the next build writes
    hipEvent_t res_last_traced = nullptr;          // stage-pipeline scenes share their queues between frames: their traces follow each other
    bool res_last_traced_used = false;
    std::vector<int64_t> halo_send_counts, halo_recv_counts;          // records (4 floats) per peer rank
    DevBuf<uint32_t> d_halo_send_px, d_halo_recv_px;
    DevBuf<uint32_t> d_halo_index;             // [pixel of the frame] place of its halo record in the receive buffer (pixels other ranks own that border this rank's tiles; others: unused) - k_resolve_tiles
    bool halo_ready = false;
    DevBuf<float> dbg_rays, dbg_hit_t;
    DevBuf<int32_t> dbg_prim, dbg_sub;
    DevBuf<uint64_t> dbg_rng;
    DevBuf<unsigned long long> counters, wave_prof, dbg_counters;
    DevBuf<float> own_slab;                    // used when world_size > 1 and the caller passes no slab
    // wavefront pipeline storage (ycge_kernels.hip: QEntry / HitRec / LEntry), sized for one ray per pixel
    DevBuf<uint8_t> wf_q0, wf_q1, wf_hit, wf_lq;
    DevBuf<uint32_t> wf_seg;                      // segment counter of the persistent extend stage
    DevBuf<uint32_t> wf_counts, tile_order;
    // the stage pipeline's second set (frames in flight: two voxel-world traces at a time, ycge_render_frame_async): queues, counters, both spill areas
    DevBuf<uint8_t> wf2_q0, wf2_q1, wf2_hit, wf2_lq;
    DevBuf<uint32_t> wf2_seg, wf2_counts;
    DevBuf<uint64_t> stack_spill_side2;
    // denoise / exposure / tonemap stage (ycge_post.hip), allocated on the first frame that asks for SDR output
    DevBuf<float> den_a, den_b, unit_n, exp_terms, d_sdr, d_sdr2;      // d_sdr2: SDR frames in flight read back one array while the next frame's tonemap fills the other
    DevBuf<float> atrous_statw;                // [pixel][25 taps][3]: colour-independent weight factors of an in-place A-trous iteration
    DevBuf<uint8_t> exp_scratch;                  // chunk records of the exposure sum (k_exposure_sum)
    DevBuf<uint32_t> post_progress;               // k_atrous_persist: groups finished per band, one 128-byte line each
    uint32_t post_epoch = 0;                      // ... counted from here in the next launch
    uint32_t post_ticket = 0;                     // k_atrous_stream, bands in order of arrival: numbers drawn so far (the counter lives in post_progress)
    // a second set of everything the denoiser scratches, for the post stages of every other frame in flight: two of them run side by side
    // (each is a dependent chain that leaves the chip idle); the exposure state passes from one to the next in frame order
    struct PostSet { DevBuf<float> den_a, den_b, unit_n, exp_terms, atrous_statw; DevBuf<uint8_t> exp_scratch; DevBuf<uint32_t> post_progress; uint32_t post_epoch = 0, post_ticket = 0;
                     void release() { den_a.release(); den_b.release(); unit_n.release(); exp_terms.release(); atrous_statw.release(); exp_scratch.release(); post_progress.release(); post_epoch = post_ticket = 0; } } alt_post;
    int post_resident_seen[2] = {-1, -1};         // post_resident_per_cu: the runtime's answer for the whole-band / split-band instantiation (-1: not asked yet)
    DevBuf<uint8_t> tone_state;                   // ToneMapper state; lives as long as the context (not reset by Resize)
    struct InplaceSchedule { ~InplaceSchedule() { pixels.release(); offsets.release(); pass_level.release(); band_desc.release(); } int w = 0, h = 0, step = 0, levels = 0, bands = 0, rows_per_band = 0, levels_per_launch = 0; uint32_t max_level_pixels = 0, window_width = 0; bool split = false; DevBuf<uint32_t> pixels, offsets, pass_level; DevBuf<int32_t> band_desc; };
    std::vector<InplaceSchedule *> schedules;     // level schedules of the in-place A-trous iterations, by (w, h, step)
    // what ycge_scene_update_objects needs from the last full upload
    std::vector<GMesh> gmeshes_host;
    std::vector<std::array<float, 6>> grid_bounds;   // VolumeGrid.TryGetBounds per grid; max < min marks an empty grid
    std::vector<std::array<float, 7>> grid_solid;    // GGrid::solid_lo / solid_hi per grid (copied into the grid's object record: the walk culls before it enters)
    int n_materials = 0, max_mesh_depth = 0;
    bool materials_can_mirror = false;
    bool has_dynamic_textures = false;           // Scene.HasDynamicTextures: every frame restarts the TAA history (RaytraceRenderer.cs:171)
    const float *denoised = nullptr;              // result of the last post stage (one of den_a / den_b / taa_hist)
    DevBuf<uint32_t> block_cost, block_order, order_ws;   // k_trace scheduling feedback (4 blocks of 8x8 px per tile)
    DevBuf<uint32_t> cost_snap;                    // a schedule built while traces are in flight reads a copy of the cost ring (ycge_launch_order_blocks)
    bool block_order_valid = false;
    // TemporalBlendWithClamp inside the trace launch (ycge::TaaFuse): neighbourhood counters per 8 x 8 block (monotonic, zeroed when allocated) and
    // what ycge_render_frame asks of / hears back from trace_frame for the frame at hand
    DevBuf<uint32_t> taa_block_ctr, taa_part_ctr;
    bool fuse_request = false, fuse_done = false;
    ycge::TaaParams fuse_T;
    DevBuf<uint64_t> stack_spill;                 // [YCGE_TRAVERSAL_STACK - 12][persistent lanes]
    bool any_light_lit = false;                   // some light has a contribution (GLight::dark == 0): the timed light loop has shadow rays to trace
    DevBuf<uint64_t> stack_spill_side;            // ... of the stage kernel that runs on the side stream beside another (the light loop beside the next round's trace)
    DevBuf<float> path_stack;                  // [3][11][persistent lanes], only for scenes with transparent materials
    int spill_levels = 0;                      // traversal depth beyond the 12 LDS levels, from the uploaded trees
    int wf_rounds = 2;                         // 2 = primary + diffuse bounce; 4 when a surface can mirror (<= 2 mirror bounces)
    bool has_grid = false;

    // scene
    bool have_scene = false;
    SceneDev sd{};
    DevBuf<GNode> d_scene_nodes;
    DevBuf<GNode> d_walk_nodes;        // SceneDev::walk_nodes (worlds of voxel grids): the scene nodes + YCGE_WALK_LEAF_NODES entries per leaf child
    DevBuf<int32_t> d_grid_owner;      // SceneDev::grid_owner
    int walk_scene_nodes = 0;          // scene nodes the walk tree was made from (0: SceneDev::walk_nodes is null)
    DevBuf<uint8_t> d_mesh_arena;
    DevBuf<uint32_t> d_scene_leaf;
    DevBuf<GPrim> d_prims;
    DevBuf<GMaterial> d_materials;
    DevBuf<GMesh> d_meshes;
    DevBuf<GGrid> d_grids;
    DevBuf<uint8_t> d_cells;
    DevBuf<int32_t> d_lut;
    DevBuf<uint32_t> d_tex_pixels;             // textures of YCGE_MAT_TEXTURED materials
    // a live texture's next frame travels through page-locked staging (two buffers taken in turn) and a stream-ordered copy on the
    // context's stream: behind the traces that still read the old frame, ahead of the ones queued after the call
    uint8_t *tex_stage[2] = {nullptr, nullptr};
    size_t tex_stage_bytes[2] = {0, 0};
    hipEvent_t tex_stage_ev[2] = {nullptr, nullptr};
    // GPU -> host copies never target memory whose mapping the library does not control (copy_out below): page-locked staging of its own
    void *out_stage = nullptr; size_t out_stage_bytes = 0;
    float *staged_sdr_dst = nullptr; size_t staged_sdr_bytes = 0;       // a synchronous frame's SDR read-back into a pageable caller array: finished on the host after the stream
    hipEvent_t tex_order_ev = nullptr;         // "everything queued on the second trace stream so far": a live texture's copy waits for it
    bool tex_stage_busy[2] = {false, false};
    int tex_stage_next = 0;
    DevBuf<int32_t> d_tex_info;
    std::vector<int32_t> tex_info_host;        // {first word, width, height, flags} per texture (ycge_scene_update_texture)
    DevBuf<GLight> d_lights;
    BuiltTree scene_tree;                      // host copy of the scene BVH in the reference's format (ycge_read_accel)
    bool scene_tree_on_device = false;         // ... not fetched yet from the last device-side build (accel_view does it on demand)
    int32_t dev_tree_nodes = 0, dev_tree_items = 0;
    DevBuf<float> d_bvh_items;                 // device-side scene BVH build (ycge_bvh_build.hip): item boxes + centroids, nine planes
    DevBuf<uint8_t> d_bvh_scratch, d_bvh_ref, d_bvh_res;
    int64_t bvh_device_builds = 0, bvh_host_fallbacks = 0, bvh_host_builds = 0;
    double bvh_last_build_us = 0.0;
    std::vector<MeshHost> meshes;

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIP_TRY(ctx, call)                                                                                   \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return (ctx)->fail(e_ == hipErrorOutOfMemory ? YCGE_ERR_OUT_OF_MEMORY : YCGE_ERR_DEVICE, \
                                                 "%s failed: %s", #call, hipGetErrorString(e_));            \
    } while (0)

#define YCGE_FLIGHT_RING 1024u       // frames in flight whose trace launches keep their timing events (ycge_async_trace_times)

extern "C" void ycge_peer_worker_main(ycge_ctx *root, ycge_ctx *peer);      // ycge_frame.cpp: a peer device's thread (ycge_create starts it)
// ---- what the other translation units of the library call in ycge_host.cpp (defined there, in this namespace)
namespace ycge_host {
// where a trace of the tile-resident form writes and which schedule it follows (trace_frame's last argument)
struct ResidentTarget {
    ycge_ctx::ResidentSet *set;
    uint32_t *cost;                 // this frame's slot of the resident cost ring
    const uint32_t *order, *n_order;        // the schedule built for this frame (null: blocks in index order)
};
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
const RcclApi &load_rccl();
size_t slab_floats(const ycge_ctx *c);
bool host_memory_is_page_locked(const void *p, size_t bytes);
int ensure_out_stage(ycge_ctx *c, size_t bytes);
void finish_staged_sdr(ycge_ctx *c);
int run_post(ycge_ctx *c, hipStream_t stream, float *out_sdr_host, bool timed, hipEvent_t history_read = nullptr /* recorded once the TAA history has been read for the last time */,
             hipEvent_t before_copy = nullptr /* recorded in front of the read-back: the exposure state is this frame's */, bool second_sdr = false,
             hipEvent_t tone_wait = nullptr /* the frame before has left its exposure state: waited for in front of this frame's exposure step */, bool second_set = false);
int join_async(ycge_ctx *c);
int copy_out(ycge_ctx *c, void *dst, const void *src, size_t bytes);
int fill_stats(ycge_ctx *c, ycge_frame_stats *st, const FrameState &fs, bool did_reset, bool have_taa, double wall_ms);
int quiesce(ycge_ctx *c);
void release_resident(ycge_ctx *c);
void snapshot_frame(ycge_ctx *c, FrameState &fs);
void schedule_policy(const ycge_ctx *c, uint32_t &policy, uint32_t &split_top, int resident_ring = 0, bool batched = false);
int scene_is_flat(const ycge_ctx *c);
bool frame_is_single_launch(const ycge_ctx *c);
bool should_reset_history(const ycge_ctx *c, const float pos[3], float yaw, float pitch);
void fill_frame_params(ycge_ctx *c, ycge::FrameParams &P, int64_t frame, const float pos[3], float yaw, float pitch, float fov_deg);
int trace_frame(ycge_ctx *c, float *d_slab, hipStream_t stream, FrameState &fs, bool timed, hipEvent_t launch_begin = nullptr, hipEvent_t launch_end = nullptr, const ResidentTarget *rt = nullptr);
void halo_layout(int hiW, int hiH, int rank, int world, std::vector<int64_t> &send_counts, std::vector<int64_t> &recv_counts, std::vector<uint32_t> &send_px, std::vector<uint32_t> &recv_px);
} // namespace ycge_host
