// ycge_frame.cpp - frame orchestration behind the C-ABI (include/ycge.h): TryFlipAndBlit steps 1-5 and 9 (RaytraceRenderer.cs:159-218, 266) as
// trace_frame / taa_and_commit, the one-process multi-device frame (peer push or the RCCL all-gather), frames in flight, the slab form of the tiled frame.
// (Context, scene flattening and upload: ycge_host.cpp.  Steps 6-8 - the post stage - and its schedules: ycge_post_host.cpp.  The tile-resident form: ycge_resident.cpp.)
#include "ycge_ctx.h"

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
    // A timed synchronous frame that is ONE plain launch: ev[1] rides on the launch itself (the kernel's own end time, ycge_launch_trace) -
    // an event recorded on the stream is a packet of its own, and the one between the trace and TAA cost every such frame 8 us (round 6)
    // (only ev[1]: with ev[0] as the launch's start event and ev[2] as k_taa's stop event the frames got SLOWER again - config 4 0.5217 -> 0.5289 ms,
    // config 1 0.0845 -> 0.0885; recorded at the frame's two ends they wait behind nothing)
    const bool kernel_stop = timed && !slab && !rt && !c->in_flight_call && frame_is_single_launch(c) && c->knobs.refill_steps == 0 && c->fan_cap == 0;
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
            e = ycge_launch_trace(&c->sd, &P, &O, c->cfg.count_work, flat, refill_steps, stream, nullptr, kernel_stop ? c->ev[1] : nullptr);
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
    if (timed && !kernel_stop) HIP_TRY(c, hipEventRecord(c->ev[1], stream));
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

void ycge_peer_worker_main(ycge_ctx *c, ycge_ctx *p)           // (a peer's thread, started by ycge_create: declared in ycge_ctx.h)
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
    // (timing events only where the caller asks for statistics: an event between two kernels is a packet of its own that the next launch waits behind -
    // the C# wrapper passes no statistics and gets the frame without them; bench.py asks, its line is measured WITH them)
    const bool timed = st != nullptr;
    int rc = multi_dev ? trace_on_all_devices(c, fs) : trace_frame(c, nullptr, c->stream, fs, timed);
    c->fuse_request = false;
    if (rc != YCGE_OK) return rc;
    rc = taa_and_commit(c, c->stream, fs, did_reset, timed, c->fuse_done);
    if (rc != YCGE_OK) return rc;
    if (out_sdr) {      // steps 6-8; with NULL the frame stops after TAA (trace-only callers, benchmarks of the hot path)
        rc = run_post(c, c->stream, out_sdr, timed);
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

// profiling aid: per-wavefront {start, end, node iterations, leaf phases} of the last counted k_wf_primary launch
int ycge_debug_read_wave_prof(ycge_ctx *c, unsigned long long *dst, size_t n_u64)
try {
    if (!c || !dst || !c->wave_prof.p || n_u64 > c->wave_prof.n) return YCGE_ERR_INVALID_ARG;
    HIP_TRY(c, hipDeviceSynchronize());
    return copy_out(c, dst, c->wave_prof.p, n_u64 * 8);
}
catch (...) { return ycge_host::abi_catch(c); }
} // extern "C"
