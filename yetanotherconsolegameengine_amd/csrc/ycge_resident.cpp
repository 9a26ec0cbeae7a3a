// ycge_resident.cpp - the tile-resident multi-GPU form (include/ycge.h: ycge_trace_tiles_resident ...), one process per GPU:
// TAA on a rank's own tiles, halo exchange, ring of frame sets, batched launches; the per-rank emulation loop (ycge_debug_resident_loop).
#include "ycge_ctx.h"

extern "C" {
// ---------------------------------------------------------------------------------------------- tile-resident form (include/ycge.h)
static int ensure_resident(ycge_ctx *c)
{
    if (c->parent || !c->peers.empty()) return c->fail(YCGE_ERR_INVALID_ARG, "the tile-resident form is one process per GPU (rank / world_size); this context drives its devices through ycge_render_frame");
    if (c->cfg.capture_debug) return c->fail(YCGE_ERR_INVALID_ARG, "the tile-resident form keeps no debug captures");
    if (c->cfg.taa_clamp_radius > 1) return c->fail(YCGE_ERR_UNSUPPORTED, "the tile-resident form exchanges a one-pixel halo: taa_clamp_radius %d needs ycge_resolve_gathered", c->cfg.taa_clamp_radius);
    HIP_TRY(c, hipSetDevice(c->device));
    const int K = c->cfg.tile_ring <= 0 ? 2 : c->cfg.tile_ring;
    if (K < 2 || K > (int)ycge_ctx::kResCostFrames - 1) return c->fail(YCGE_ERR_INVALID_ARG, "config.tile_ring must be 2..%d", (int)ycge_ctx::kResCostFrames - 1);
    if ((int)c->rsets.size() != K || !c->halo_ready) {
        HIP_TRY(c, hipDeviceSynchronize());
        release_resident(c);
        {
            std::vector<uint32_t> spx, rpx;
            halo_layout(c->hiW, c->hiH, c->cfg.rank, c->cfg.world_size, c->halo_send_counts, c->halo_recv_counts, spx, rpx);
            if (spx.empty()) spx.push_back(0u);
            if (rpx.empty()) rpx.push_back(0u);
            HIP_TRY(c, c->d_halo_send_px.upload(spx)); HIP_TRY(c, c->d_halo_recv_px.upload(rpx));
            {   // k_resolve_tiles: pixel -> its record (a pixel that borders two of this rank's tiles arrives twice, with the same four floats: either will do)
                std::vector<uint32_t> index((size_t)c->hiW * c->hiH, 0u);
                size_t n_recv = 0;
                for (int64_t v : c->halo_recv_counts) n_recv += (size_t)v;
                for (size_t r = 0; r < n_recv; r++) index[rpx[r]] = (uint32_t)r;
                HIP_TRY(c, c->d_halo_index.upload(index));
            }
            c->halo_ready = true;
        }
        const size_t n = (size_t)c->hiW * c->hiH;
        for (int k = 0; k < K; k++) {
            auto *rs = new ycge_ctx::ResidentSet();
            c->rsets.push_back(rs);
            HIP_TRY(c, rs->hdr.alloc(3 * n)); HIP_TRY(c, rs->normal.alloc(3 * n)); HIP_TRY(c, rs->depth.alloc(n)); HIP_TRY(c, rs->sky.alloc(n));
            HIP_TRY(c, rs->spill.alloc(c->stack_spill.n));
            HIP_TRY(c, hipEventCreateWithFlags(&rs->traced, hipEventDisableTiming)); HIP_TRY(c, hipEventCreateWithFlags(&rs->resolved, hipEventDisableTiming));
        }
        const size_t nb = (size_t)(c->n_owned > 0 ? c->n_owned : 1) * 4;
        HIP_TRY(c, c->res_cost.alloc(nb * (ycge_ctx::kResCostFrames + 1)));          // (+ one slot nobody reads: what a schedule build "clears for the next frame")
        HIP_TRY(c, hipMemset(c->res_cost.p, 0, nb * (ycge_ctx::kResCostFrames + 1) * sizeof(uint32_t)));
        c->res_order_next = 0;
        for (int k = 0; k < 3; k++) {
            auto *o = new DevBuf<uint32_t>(); auto *w = new DevBuf<uint32_t>();
            c->res_order.push_back(o); c->res_ws.push_back(w);
            HIP_TRY(c, o->alloc(nb * YCGE_SCHEDULE_SLACK)); HIP_TRY(c, w->alloc(96)); HIP_TRY(c, hipMemset(w->p, 0, 96 * sizeof(uint32_t)));
            hipEvent_t e1 = nullptr, e2 = nullptr;
            HIP_TRY(c, hipEventCreateWithFlags(&e1, hipEventDisableTiming)); HIP_TRY(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            c->res_order_ev.push_back(e1); c->res_order_read_ev.push_back(e2);
            c->res_order_frame.push_back(-1);
        }
        HIP_TRY(c, hipEventCreateWithFlags(&c->res_last_traced, hipEventDisableTiming));
    }
    return YCGE_OK;
}

int ycge_halo_counts(ycge_ctx *c, int64_t *send_counts, int64_t *recv_counts)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!send_counts || !recv_counts) return c->fail(YCGE_ERR_INVALID_ARG, "null count array");
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    const int rc = ensure_resident(c);
    if (rc != YCGE_OK) return rc;
    for (int r = 0; r < c->cfg.world_size; r++) { send_counts[r] = c->halo_send_counts[(size_t)r]; recv_counts[r] = c->halo_recv_counts[(size_t)r]; }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

int ycge_history_slab_bytes(const ycge_ctx *c, size_t *bytes)
try {
    if (!c || !bytes) return YCGE_ERR_INVALID_ARG;
    *bytes = (size_t)c->tiles_per_rank_padded * 256 * 3 * sizeof(float);
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// steps 1-4 of TryFlipAndBlit on this rank's tiles into the frame set of the ring, then the halo records the other ranks need
int ycge_trace_tiles_resident(ycge_ctx *c, void *d_halo_send, void *hip_stream, ycge_frame_stats *st)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    int rc = ensure_resident(c);
    if (rc != YCGE_OK) return rc;
    size_t n_send = 0;
    for (int64_t v : c->halo_send_counts) n_send += (size_t)v;
    if (n_send > 0 && !d_halo_send) return c->fail(YCGE_ERR_INVALID_ARG, "null halo send buffer (%zu records of 16 bytes)", n_send);
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    const int K = (int)c->rsets.size();
    if ((int)c->pending.size() >= K) return c->fail(YCGE_ERR_INVALID_ARG, "%d traced frames wait to be resolved: the ring holds config.tile_ring = %d", (int)c->pending.size(), K);
    auto t0 = std::chrono::steady_clock::now();
    FrameState fs;
    snapshot_frame(c, fs);
    ycge_ctx::ResidentSet *rs = c->rsets[(size_t)((uint64_t)fs.frame % (uint64_t)K)];
    if (rs->resolved_used) HIP_TRY(c, hipStreamWaitEvent(stream, rs->resolved, 0));        // TAA of frame N - K has read this set
    if (rs->traced_used) HIP_TRY(c, hipStreamWaitEvent(stream, rs->traced, 0));            // (its spill area: the trace of frame N - K, on whatever stream)
    const bool single = frame_is_single_launch(c);
    if ((!single || c->sd.any_transparent) && c->res_last_traced_used) HIP_TRY(c, hipStreamWaitEvent(stream, c->res_last_traced, 0));      // shared stage queues / refraction stacks: one trace at a time
    const uint32_t n_blocks = (uint32_t)c->n_owned * 4u, RC = ycge_ctx::kResCostFrames;
    const uint32_t cost_slot = (uint32_t)((uint64_t)fs.frame % RC);
    ResidentTarget rt;
    rt.set = rs;
    rt.cost = c->res_cost.p + (size_t)cost_slot * n_blocks;
    rt.order = nullptr; rt.n_order = nullptr;
    const bool lpt = single && !c->knobs.no_lpt;
    int ob = -1;          // the newest schedule this frame may follow: built behind a frame M <= N - K (anything younger would make this trace wait for a trace of its own ring)
    if (lpt)
        for (int b = 0; b < 3; b++)
            if (c->res_order_frame[(size_t)b] >= 0 && c->res_order_frame[(size_t)b] + K <= fs.frame && (ob < 0 || c->res_order_frame[(size_t)b] > c->res_order_frame[(size_t)ob])) ob = b;
    if (ob >= 0) {
        rt.order = c->res_order[(size_t)ob]->p; rt.n_order = c->res_ws[(size_t)ob]->p + 16;
        HIP_TRY(c, hipStreamWaitEvent(stream, c->res_order_ev[(size_t)ob], 0));
    }
    if (lpt) HIP_TRY(c, hipMemsetAsync(rt.cost, 0, (size_t)n_blocks * sizeof(uint32_t), stream));          // this frame's cost slot (the kernel's atomicMax needs zeros)
    rc = trace_frame(c, nullptr, stream, fs, st != nullptr, nullptr, nullptr, &rt);
    if (rc != YCGE_OK) return rc;
    if (ob >= 0) HIP_TRY(c, hipEventRecord(c->res_order_read_ev[(size_t)ob], stream));
    int e = ycge_launch_halo(0, rs->hdr.p, rs->sky.p, c->d_halo_send_px.p, (uint32_t)n_send, d_halo_send, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_gather_halo launch failed: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(c, hipEventRecord(rs->traced, stream)); rs->traced_used = true;
    if (!single || c->sd.any_transparent) { HIP_TRY(c, hipEventRecord(c->res_last_traced, stream)); c->res_last_traced_used = true; }
    const int every = c->knobs.res_sched_every > 0 ? c->knobs.res_sched_every : K;
    if (lpt && fs.frame % every == 0) {
        // a new schedule, behind this trace on the side stream: from the cost slots of the frames up to this one (the K - 1 slots the traces of
        // frames N + 1 .. N + K - 1 may be writing are left out), into the oldest of the three buffers once its last reader is done.  Built
        // every `every`-th frame only: which blocks run long is a property of the image region, and a rank's host thread has ~25 driver
        // calls a frame to make as it is
        uint32_t policy, split_top;
        schedule_policy(c, policy, split_top, K);
        uint32_t skip = 0;
        for (int a = 1; a < K; a++) skip |= 1u << ((cost_slot + (uint32_t)a) % RC);
        const int tb = c->res_order_next;
        c->res_order_next = (tb + 1) % 3;
        HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, rs->traced, 0));
        if (c->res_order_frame[(size_t)tb] >= 0) HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->res_order_read_ev[(size_t)tb], 0));
        c->res_order_frame[(size_t)tb] = -1;          // (not to be picked while it is being rewritten ...)
        e = ycge_launch_order_blocks(c->res_cost.p, n_blocks, policy, split_top, 0u, 0u, RC /* the slot nobody reads */, skip, c->res_ws[(size_t)tb]->p, c->res_order[(size_t)tb]->p, c->fan_stream, 0, RC, c->cost_snap.p);      // (traces in flight write their costs meanwhile: a copy is read)
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "schedule launch failed: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(c, hipEventRecord(c->res_order_ev[(size_t)tb], c->fan_stream));
        c->res_order_frame[(size_t)tb] = fs.frame;          // (... and from frame N + K on it is the newest)
    }
    c->pending.push_back(fs);
    if (st) {
        HIP_TRY(c, hipStreamSynchronize(stream));
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return fill_stats(c, st, fs, false, false, wall);
    }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// n consecutive frames of this rank's tiles in ONE launch (k_trace_batch): what one frame's launch - a rank's share is a few thousand
// blocks - leaves idle around its longest chains, the other frames' blocks fill.  poses: n x {pos xyz, yaw, pitch, fov} (the camera of each
// frame, as n ycge_set_camera calls would set it; the last one stays the context's camera); d_halo_send: n buffers, filled as by n
// ycge_trace_tiles_resident calls.  The frames are then exchanged and resolved one by one, oldest first, as ever.  Same pixels (the frames
// never needed each other's traces).  Scenes that trace in stages or keep refraction stacks, and counting contexts, take the frames one
// by one here too.
int ycge_trace_tiles_resident_batch(ycge_ctx *c, int32_t n, const float *poses, void *const *d_halo_send, void *hip_stream)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (n < 1 || n > ycge_ctx::kBatchMax || !poses) return c->fail(YCGE_ERR_INVALID_ARG, "a batch is 1..%d frames with their poses", ycge_ctx::kBatchMax);
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    int rc = ensure_resident(c);
    if (rc != YCGE_OK) return rc;
    size_t n_send = 0;
    for (int64_t v : c->halo_send_counts) n_send += (size_t)v;
    if (n_send > 0) { if (!d_halo_send) return c->fail(YCGE_ERR_INVALID_ARG, "null halo send buffers"); for (int k = 0; k < n; k++) if (!d_halo_send[k]) return c->fail(YCGE_ERR_INVALID_ARG, "null halo send buffer of frame %d of the batch", k); }
    const int K = (int)c->rsets.size();
    if ((int)c->pending.size() + n > K) return c->fail(YCGE_ERR_INVALID_ARG, "%d traced frames wait to be resolved and %d more are asked for: the ring holds config.tile_ring = %d", (int)c->pending.size(), n, K);
    auto set_pose = [&](int k) {
        std::lock_guard<std::mutex> g(c->cam_lock);
        const float *q = poses + 6 * k;
        c->cam_pos[0] = q[0]; c->cam_pos[1] = q[1]; c->cam_pos[2] = q[2]; c->yaw = q[3]; c->pitch = q[4]; c->fov_deg = q[5];
    };
    // A batch is all or nothing for the HOST-side state, on both ways through this function: frames enter the pending list for good only
    // once the whole batch is queued, and a failure on the way (a launch refused, an allocation, an exception) puts the frame counter, the
    // pending list, the camera and the batch parity back where they were - the caller's exchange ring and the library stay in step.
    // (What a failed HIP call leaves on the device is the device's business: the context reports YCGE_ERR_DEVICE.)
    struct Rollback {
        ycge_ctx *c; int64_t counter; size_t pending; uint64_t batches; float pos[3], yaw, pitch, fov; bool armed = true;
        Rollback(ycge_ctx *c_) : c(c_), counter(c_->frame_counter), pending(c_->pending.size()), batches(c_->batch_count)
        {
            std::lock_guard<std::mutex> g(c->cam_lock);
            pos[0] = c->cam_pos[0]; pos[1] = c->cam_pos[1]; pos[2] = c->cam_pos[2]; yaw = c->yaw; pitch = c->pitch; fov = c->fov_deg;
        }
        ~Rollback()
        {
            if (!armed) return;
            c->frame_counter = counter; c->batch_count = batches;
            while (c->pending.size() > pending) c->pending.pop_back();
            c->batch_collect = false; c->batch_P.clear(); c->batch_O.clear();
            std::lock_guard<std::mutex> g(c->cam_lock);
            c->cam_pos[0] = pos[0]; c->cam_pos[1] = pos[1]; c->cam_pos[2] = pos[2]; c->yaw = yaw; c->pitch = pitch; c->fov_deg = fov;
        }
    } rollback(c);
    const bool single = frame_is_single_launch(c);
    if (!single || c->sd.any_transparent || c->cfg.count_work || n == 1) {
        for (int k = 0; k < n; k++) {
            set_pose(k);
            rc = ycge_trace_tiles_resident(c, d_halo_send ? d_halo_send[k] : nullptr, hip_stream, nullptr);
            if (rc != YCGE_OK) return rc;
        }
        rollback.armed = false;
        return YCGE_OK;
    }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    const uint32_t n_blocks = (uint32_t)c->n_owned * 4u, RC = ycge_ctx::kResCostFrames;
    const bool lpt = !c->knobs.no_lpt;
    std::vector<FrameState> fs((size_t)n);
    std::vector<ycge_ctx::ResidentSet *> sets((size_t)n, nullptr);
    c->batch_P.clear(); c->batch_O.clear();
    int ob = -1;
    for (int k = 0; k < n; k++) {
        set_pose(k);
        snapshot_frame(c, fs[(size_t)k]);
        ycge_ctx::ResidentSet *rs = sets[(size_t)k] = c->rsets[(size_t)((uint64_t)fs[(size_t)k].frame % (uint64_t)K)];
        if (rs->resolved_used) HIP_TRY(c, hipStreamWaitEvent(stream, rs->resolved, 0));
        if (rs->traced_used) HIP_TRY(c, hipStreamWaitEvent(stream, rs->traced, 0));
        const uint32_t cost_slot = (uint32_t)((uint64_t)fs[(size_t)k].frame % RC);
        ResidentTarget rt;
        rt.set = rs;
        rt.cost = c->res_cost.p + (size_t)cost_slot * n_blocks;
        rt.order = nullptr; rt.n_order = nullptr;
        if (k == 0 && lpt) {        // ONE schedule for the whole batch: the newest the batch's first frame may follow (ycge_trace_tiles_resident's rule)
            for (int b = 0; b < 3; b++)
                if (c->res_order_frame[(size_t)b] >= 0 && c->res_order_frame[(size_t)b] + K <= fs[0].frame && (ob < 0 || c->res_order_frame[(size_t)b] > c->res_order_frame[(size_t)ob])) ob = b;
            if (ob >= 0) HIP_TRY(c, hipStreamWaitEvent(stream, c->res_order_ev[(size_t)ob], 0));
        }
        if (ob >= 0) { rt.order = c->res_order[(size_t)ob]->p; rt.n_order = c->res_ws[(size_t)ob]->p + 16; }
        if (lpt) HIP_TRY(c, hipMemsetAsync(rt.cost, 0, (size_t)n_blocks * sizeof(uint32_t), stream));
        c->batch_collect = true;
        rc = trace_frame(c, nullptr, stream, fs[(size_t)k], false, nullptr, nullptr, &rt);
        c->batch_collect = false;
        if (rc != YCGE_OK) return rc;
    }
    if ((int)c->batch_P.size() != n || (int)c->batch_O.size() != n) return c->fail(YCGE_ERR_DEVICE, "batch: %zu of %d frames left their launch parameters", c->batch_P.size(), n);
    // one spill area for the launch, n frames wide: a workgroup's column is its index in the launch
    const uint32_t lanes = c->batch_O[0].stack_lanes;
    const size_t spill_words = (size_t)(c->spill_levels > 0 ? c->spill_levels : 1) * lanes * (size_t)n;
    const int bp = (int)(c->batch_count & 1u);
    for (int q = 0; q < 2; q++)          // (both areas at the first batch of a size: an allocation of gigabytes is no part of a later frame)
        if (c->batch_spill[q].n < spill_words) {
            if (c->batch_spill_used[q]) HIP_TRY(c, hipEventSynchronize(c->batch_done[q]));
            HIP_TRY(c, c->batch_spill[q].alloc(spill_words));
        }
    for (int k = 0; k < n; k++) { c->batch_O[(size_t)k].stack_spill = c->batch_spill[bp].p; c->batch_O[(size_t)k].stack_lanes = lanes * (uint32_t)n; }
    c->batch_count++;
    if (c->batch_spill_used[bp]) HIP_TRY(c, hipStreamWaitEvent(stream, c->batch_done[bp], 0));          // (the batch before the last scratched this area)
    int e = ycge_launch_trace_batch(&c->sd, c->batch_P.data(), c->batch_O.data(), n, 0, scene_is_flat(c), stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_trace_batch launch failed: %s", hipGetErrorString((hipError_t)e));
    if (!c->batch_done[bp]) HIP_TRY(c, hipEventCreateWithFlags(&c->batch_done[bp], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->batch_done[bp], stream)); c->batch_spill_used[bp] = true;
    if (ob >= 0) HIP_TRY(c, hipEventRecord(c->res_order_read_ev[(size_t)ob], stream));
    for (int k = 0; k < n; k++) {
        ycge_ctx::ResidentSet *rs = sets[(size_t)k];
        e = ycge_launch_halo(0, rs->hdr.p, rs->sky.p, c->d_halo_send_px.p, (uint32_t)n_send, d_halo_send ? d_halo_send[k] : nullptr, stream);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_gather_halo launch failed: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(c, hipEventRecord(rs->traced, stream)); rs->traced_used = true;
    }
    for (int k = 0; k < n; k++) c->pending.push_back(fs[(size_t)k]);
    rollback.armed = false;
    const int every = c->knobs.res_sched_every > 0 ? c->knobs.res_sched_every : K;
    bool build = false;
    for (int k = 0; k < n; k++) if (fs[(size_t)k].frame % every == 0) build = true;
    if (lpt && build) {         // a new schedule behind the batch, from the cost slots up to its last frame's (ycge_trace_tiles_resident's rule)
        uint32_t policy, split_top;
        schedule_policy(c, policy, split_top, K, true);
        const uint32_t last_slot = (uint32_t)((uint64_t)fs[(size_t)n - 1].frame % RC);
        uint32_t skip = 0;
        for (int a = 1; a < K; a++) skip |= 1u << ((last_slot + (uint32_t)a) % RC);
        const int tb = c->res_order_next;
        c->res_order_next = (tb + 1) % 3;
        HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, sets[(size_t)n - 1]->traced, 0));
        if (c->res_order_frame[(size_t)tb] >= 0) HIP_TRY(c, hipStreamWaitEvent(c->fan_stream, c->res_order_read_ev[(size_t)tb], 0));
        c->res_order_frame[(size_t)tb] = -1;
        e = ycge_launch_order_blocks(c->res_cost.p, n_blocks, policy, split_top, 0u, 0u, RC, skip, c->res_ws[(size_t)tb]->p, c->res_order[(size_t)tb]->p, c->fan_stream, 0, RC, c->cost_snap.p);      // (traces in flight write their costs meanwhile: a copy is read)
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "schedule launch failed: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(c, hipEventRecord(c->res_order_ev[(size_t)tb], c->fan_stream));
        c->res_order_frame[(size_t)tb] = fs[(size_t)n - 1].frame;
    }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// the records of this frame's halo into its set, TAA on this rank's own tiles (steps 5 and 9), the resolved history of those tiles as a slab
int ycge_resolve_tiles_resident(ycge_ctx *c, const void *d_halo_recv, void *d_history_slab, void *hip_stream, ycge_frame_stats *st)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (c->pending.empty() || c->rsets.empty()) return c->fail(YCGE_ERR_INVALID_ARG, "no traced frame to resolve: every ycge_resolve_tiles_resident follows its own ycge_trace_tiles_resident");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n_recv = 0;
    for (int64_t v : c->halo_recv_counts) n_recv += (size_t)v;
    if (n_recv > 0 && !d_halo_recv) return c->fail(YCGE_ERR_INVALID_ARG, "null halo receive buffer (%zu records of 16 bytes)", n_recv);
    // every refusal comes BEFORE the frame leaves the pending list: a refused call changes nothing, ring and caller stay in step
    if (c->cfg.taa_clamp_radius > 1) return c->fail(YCGE_ERR_UNSUPPORTED, "the tile-resident form exchanges a one-pixel halo: taa_clamp_radius %d needs ycge_resolve_gathered", c->cfg.taa_clamp_radius);
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    auto t0 = std::chrono::steady_clock::now();
    FrameState fs = c->pending.front();
    c->pending.pop_front();
    const int K = (int)c->rsets.size();
    ycge_ctx::ResidentSet *rs = c->rsets[(size_t)((uint64_t)fs.frame % (uint64_t)K)];
    HIP_TRY(c, hipStreamWaitEvent(stream, rs->traced, 0));         // (the caller's exchange already follows the trace; this holds whatever streams it uses)
    if (st) HIP_TRY(c, hipEventRecord(c->ev[1], stream));
    int e = 0;
    if (c->knobs.split_resolve) {
        e = ycge_launch_halo(1, rs->hdr.p, rs->sky.p, c->d_halo_recv_px.p, (uint32_t)n_recv, const_cast<void *>(d_halo_recv), stream);
        if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_scatter_halo launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    fs.reset = should_reset_history(c, fs.pos, fs.yaw, fs.pitch) || c->has_dynamic_textures;
    TaaParams T;
    T.w = c->hiW; T.h = c->hiH;
    T.alpha = cs_max(0.0f, cs_min(1.0f, c->cfg.taa_alpha));
    T.radius = c->cfg.taa_clamp_radius > 0 ? c->cfg.taa_clamp_radius : 0;
    T.pad_lum = c->cfg.taa_luminance_pad;
    const bool did_reset = !c->taa_valid || fs.reset;
    T.reset = did_reset ? 1 : 0;
    FrameParams P;
    fill_frame_params(c, P, fs.frame, fs.pos, fs.yaw, fs.pitch, fs.fov);
    // ONE launch: the halo taps read from the records where the exchange left them, TAA on this rank's tiles, the resolved history packed (k_resolve_tiles)
    if (c->knobs.split_resolve)
        e = ycge_launch_taa_tiles(&T, &P, rs->hdr.p, rs->normal.p, rs->depth.p, rs->sky.p, c->taa_hist.p, c->prev_normal.p, c->prev_depth.p, c->prev_sky.p, (float *)d_history_slab /* packed by the same launch */, stream);
    else
        e = ycge_launch_resolve_tiles(&T, &P, rs->hdr.p, rs->normal.p, rs->depth.p, rs->sky.p, d_halo_recv, c->d_halo_index.p, c->taa_hist.p, c->prev_normal.p, c->prev_depth.p, c->prev_sky.p, (float *)d_history_slab, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "resolve launch failed: %s", hipGetErrorString((hipError_t)e));
    if (st) HIP_TRY(c, hipEventRecord(c->ev[2], stream));
    c->taa_valid = true;
    c->last_cam[0] = fs.pos[0]; c->last_cam[1] = fs.pos[1]; c->last_cam[2] = fs.pos[2]; c->last_yaw = fs.yaw; c->last_pitch = fs.pitch;
    HIP_TRY(c, hipEventRecord(rs->resolved, stream)); rs->resolved_used = true;
    if (st) {
        HIP_TRY(c, hipStreamSynchronize(stream));
        std::memset(st, 0, sizeof *st);
        st->frame = fs.frame; st->history_reset = did_reset ? 1 : 0;
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
        st->taa_ms = ms;
        st->exposure = 1.0f;
        st->n_devices_traced = 1; st->device_tiles[0] = c->n_owned;
        st->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// the consumer's half: world_size history slabs (rank-major, as an all-gather or a gather leaves them) into the full-frame history
int ycge_unpack_history(ycge_ctx *c, const void *d_all_history_slabs, void *hip_stream)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!d_all_history_slabs) return c->fail(YCGE_ERR_INVALID_ARG, "null gathered history slabs");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->last_stream = stream;
    const int e = ycge_launch_unpack_history((const float *)d_all_history_slabs, (size_t)c->tiles_per_rank_padded * 256 * 3, c->hiW, c->hiH, c->tiles_x, c->n_tiles, c->cfg.world_size, c->taa_hist.p, stream);
    if (e != 0) return c->fail(YCGE_ERR_DEVICE, "k_unpack_history launch failed: %s", hipGetErrorString((hipError_t)e));
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// measurement (profiles/rank_flight.py): a rank's pipelined loop of the tile-resident form driven from C - K traces in flight over K streams,
// a device copy of the rank's own records standing in for the all-to-all, resolve + history slab on another stream - so that what is timed
// is the library's and the driver's host cost per frame, not a scripting language's.  period_ms: wall time per frame; issue_ms: host time
// to queue a frame (where the two agree the loop is host-bound).
int ycge_debug_resident_loop(ycge_ctx *c, int32_t frames, double *period_ms, double *issue_ms)
try {
    if (!c || frames <= 0 || !period_ms || !issue_ms) return YCGE_ERR_INVALID_ARG;
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    int rc = ensure_resident(c);
    if (rc != YCGE_OK) return rc;
    const int K = (int)c->rsets.size();
    size_t ns = 0, nr = 0;
    for (int64_t v : c->halo_send_counts) ns += (size_t)v;
    for (int64_t v : c->halo_recv_counts) nr += (size_t)v;
    const size_t sb = (ns ? ns : 1) * 16, rb = (nr ? nr : 1) * 16, hb = (size_t)c->tiles_per_rank_padded * 256 * 3 * sizeof(float);
    std::vector<hipStream_t> st((size_t)K, nullptr);
    std::vector<hipEvent_t> evt((size_t)K, nullptr), evr((size_t)K, nullptr);
    std::vector<void *> send((size_t)K, nullptr), recv((size_t)K, nullptr), hist((size_t)K, nullptr);
    hipStream_t comm = nullptr;
    auto cleanup = [&]() {
        (void)hipDeviceSynchronize();
        for (int k = 0; k < K; k++) { if (st[k]) (void)hipStreamDestroy(st[k]); if (evt[k]) (void)hipEventDestroy(evt[k]); if (evr[k]) (void)hipEventDestroy(evr[k]);
                                      if (send[k]) (void)hipFree(send[k]); if (recv[k]) (void)hipFree(recv[k]); if (hist[k]) (void)hipFree(hist[k]); }
        if (comm) (void)hipStreamDestroy(comm);
    };
#define LOOP_TRY(call) do { if ((call) != hipSuccess) { cleanup(); return c->fail(YCGE_ERR_DEVICE, "%s failed", #call); } } while (0)
    {   // the exchange + resolve stream at the highest priority (YCGE_RES_LOOP_PRIO=0: plain): its small kernels must not queue behind a
        // trace that happens to share its hardware queue - a resolve held up that way holds up the trace K frames later
        int lo = 0, hi = 0;
        const char *pe = getenv("YCGE_RES_LOOP_PRIO");
        if ((!pe || atoi(pe) != 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) LOOP_TRY(hipStreamCreateWithPriority(&comm, hipStreamNonBlocking, hi));
        else LOOP_TRY(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
    }
    for (int k = 0; k < K; k++) {
        LOOP_TRY(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        LOOP_TRY(hipEventCreateWithFlags(&evt[k], hipEventDisableTiming)); LOOP_TRY(hipEventCreateWithFlags(&evr[k], hipEventDisableTiming));
        LOOP_TRY(hipMalloc(&send[k], sb)); LOOP_TRY(hipMalloc(&recv[k], rb)); LOOP_TRY(hipMalloc(&hist[k], hb));
        LOOP_TRY(hipMemset(send[k], 0, sb)); LOOP_TRY(hipMemset(recv[k], 0, rb));
    }
    std::deque<int> issued;
    // YCGE_RES_LOOP_COMM=slot: exchange and resolve of a frame on the stream of ITS ring slot (the next trace there waits for that resolve
    // anyway) - K streams in all instead of K + 1: no stream shares a hardware queue with a trace while K <= 4
    const char *ce = getenv("YCGE_RES_LOOP_COMM");
    const bool comm_on_slot = ce && ce[0] == 's';
    auto resolve = [&](int k) -> int {
        hipStream_t cs = comm_on_slot ? st[k] : comm;
        if (!comm_on_slot && hipStreamWaitEvent(cs, evt[k], 0) != hipSuccess) return YCGE_ERR_DEVICE;
        if (hipMemcpyAsync(recv[k], send[k], sb < rb ? sb : rb, hipMemcpyDeviceToDevice, cs) != hipSuccess) return YCGE_ERR_DEVICE;      // stands in for the all-to-all
        const int r2 = ycge_resolve_tiles_resident(c, recv[k], hist[k], cs, nullptr);
        if (r2 != YCGE_OK) return r2;
        return hipEventRecord(evr[k], cs) == hipSuccess ? YCGE_OK : YCGE_ERR_DEVICE;
    };
    int64_t i = 0;
    // YCGE_RES_LOOP_TIMELINE=1: timing events around the last 16 frames' traces (begin: behind the stream's waits; end: behind the halo
    // gather), printed relative to the first - how the K traces in flight really lie to each other
    const bool timeline = getenv("YCGE_RES_LOOP_TIMELINE") != nullptr;
    std::vector<hipEvent_t> tl_b, tl_e;
    // YCGE_RES_LOOP_EAGER=0: a frame's exchange + resolve are queued only when its ring slot is needed again (round 4's loop).  Default: queued
    // right behind its trace (they wait for the trace's event on their own stream) - the trace that takes the slot K frames later then finds
    // the resolve done instead of waiting for one that was queued a moment ago and runs starved beside the traces in flight
    const char *ee = getenv("YCGE_RES_LOOP_EAGER");
    const bool eager = !(ee && atoi(ee) == 0);
    auto frame = [&]() -> int {
        const int k = (int)(i++ % K);
        if ((int)issued.size() == K) { const int r2 = resolve(issued.front()); issued.pop_front(); if (r2 != YCGE_OK) return r2; }
        if (hipStreamWaitEvent(st[k], evr[k], 0) != hipSuccess) return YCGE_ERR_DEVICE;
        const bool mark = timeline && i > 12 + (int64_t)frames - 16;
        if (mark) { hipEvent_t eb = nullptr; if (hipEventCreate(&eb) != hipSuccess || hipEventRecord(eb, st[k]) != hipSuccess) return YCGE_ERR_DEVICE; tl_b.push_back(eb); }
        const int r2 = ycge_trace_tiles_resident(c, send[k], st[k], nullptr);
        if (r2 != YCGE_OK) return r2;
        issued.push_back(k);
        if (mark) { hipEvent_t ee2 = nullptr; if (hipEventCreate(&ee2) != hipSuccess || hipEventRecord(ee2, st[k]) != hipSuccess) return YCGE_ERR_DEVICE; tl_e.push_back(ee2); }
        if (hipEventRecord(evt[k], st[k]) != hipSuccess) return YCGE_ERR_DEVICE;
        if (eager) { const int r3 = resolve(issued.front()); issued.pop_front(); if (r3 != YCGE_OK) return r3; }
        return YCGE_OK;
    };
    auto drain = [&]() -> int { while (!issued.empty()) { const int r2 = resolve(issued.front()); issued.pop_front(); if (r2 != YCGE_OK) return r2; } return YCGE_OK; };
    // YCGE_RES_LOOP_BATCH=n: the frames n at a time in one launch (ycge_trace_tiles_resident_batch), consecutive batches on two streams
    const int nb = getenv("YCGE_RES_LOOP_BATCH") ? atoi(getenv("YCGE_RES_LOOP_BATCH")) : 0;
    if (nb > 1) {
        if (nb > K || nb > ycge_ctx::kBatchMax) { cleanup(); return c->fail(YCGE_ERR_INVALID_ARG, "YCGE_RES_LOOP_BATCH=%d needs a ring of at least that many sets (and <= %d)", nb, ycge_ctx::kBatchMax); }
        float pose[6 * ycge_ctx::kBatchMax];
        { std::lock_guard<std::mutex> g(c->cam_lock); for (int k = 0; k < nb; k++) { pose[6 * k] = c->cam_pos[0]; pose[6 * k + 1] = c->cam_pos[1]; pose[6 * k + 2] = c->cam_pos[2]; pose[6 * k + 3] = c->yaw; pose[6 * k + 4] = c->pitch; pose[6 * k + 5] = c->fov_deg; } }
        int64_t batches = 0;
        auto batch = [&]() -> int {
            while ((int)issued.size() + nb > K) { const int r2 = resolve(issued.front()); issued.pop_front(); if (r2 != YCGE_OK) return r2; }
            hipStream_t bs = st[(size_t)((batches++ & 1) * (K >= 3 ? 2 : 1))];          // (streams 0 and 2 of the loop: neighbours share a hardware queue on this runtime - 4 queues, round robin)
            void *sends[ycge_ctx::kBatchMax];
            int slots[ycge_ctx::kBatchMax];
            for (int k = 0; k < nb; k++) { slots[k] = (int)(i++ % K); sends[k] = send[(size_t)slots[k]]; if (hipStreamWaitEvent(bs, evr[(size_t)slots[k]], 0) != hipSuccess) return YCGE_ERR_DEVICE; }
            const bool mark = timeline && batches > 4 + (int64_t)((frames + nb - 1) / nb) - 10;
            if (mark) { hipEvent_t eb = nullptr; if (hipEventCreate(&eb) != hipSuccess || hipEventRecord(eb, bs) != hipSuccess) return YCGE_ERR_DEVICE; tl_b.push_back(eb); }
            const int r2 = ycge_trace_tiles_resident_batch(c, nb, pose, sends, bs);
            if (r2 != YCGE_OK) return r2;
            if (mark) { hipEvent_t ee2 = nullptr; if (hipEventCreate(&ee2) != hipSuccess || hipEventRecord(ee2, bs) != hipSuccess) return YCGE_ERR_DEVICE; tl_e.push_back(ee2); }
            for (int k = 0; k < nb; k++) { issued.push_back(slots[k]); if (hipEventRecord(evt[(size_t)slots[k]], bs) != hipSuccess) return YCGE_ERR_DEVICE; }
            // the frames of the batch before the last are resolved NOW (they run beside the launches in flight, starved: a batch that re-uses
            // their sets should find them done - a ring of three batches' sets lets consecutive launches lie side by side)
            while ((int)issued.size() > (eager ? 0 : 2 * nb)) { const int r3 = resolve(issued.front()); issued.pop_front(); if (r3 != YCGE_OK) return r3; }
            return YCGE_OK;
        };
        const int nbat = (frames + nb - 1) / nb;
        for (int w = 0; w < 4 && rc == YCGE_OK; w++) rc = batch();
        if (rc == YCGE_OK) rc = drain();
        if (rc != YCGE_OK) { cleanup(); return rc; }
        LOOP_TRY(hipDeviceSynchronize());
        const auto b0 = std::chrono::steady_clock::now();
        for (int f = 0; f < nbat && rc == YCGE_OK; f++) rc = batch();
        if (rc == YCGE_OK) rc = drain();
        const auto b1 = std::chrono::steady_clock::now();
        if (rc != YCGE_OK) { cleanup(); return rc; }
        LOOP_TRY(hipDeviceSynchronize());
        const auto b2 = std::chrono::steady_clock::now();
        *issue_ms = std::chrono::duration<double, std::milli>(b1 - b0).count() / (nbat * nb);
        *period_ms = std::chrono::duration<double, std::milli>(b2 - b0).count() / (nbat * nb);
        for (size_t q = 0; q < tl_b.size() && q < tl_e.size(); q++) {
            float b = 0.0f, e2 = 0.0f;
            (void)hipEventElapsedTime(&b, tl_b[0], tl_b[q]); (void)hipEventElapsedTime(&e2, tl_b[0], tl_e[q]);
            fprintf(stderr, "  batch %2zu: begin %7.3f ms  end %7.3f ms  duration %6.3f\n", q, b, e2, e2 - b);
        }
        for (hipEvent_t ev : tl_b) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : tl_e) (void)hipEventDestroy(ev);
        cleanup();
        return YCGE_OK;
    }
    for (int w = 0; w < 12 && rc == YCGE_OK; w++) rc = frame();
    if (rc == YCGE_OK) rc = drain();
    if (rc != YCGE_OK) { cleanup(); return rc; }
    LOOP_TRY(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames && rc == YCGE_OK; f++) rc = frame();
    if (rc == YCGE_OK) rc = drain();
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != YCGE_OK) { cleanup(); return rc; }
    LOOP_TRY(hipDeviceSynchronize());
    const auto t2 = std::chrono::steady_clock::now();
#undef LOOP_TRY
    *issue_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / frames;
    *period_ms = std::chrono::duration<double, std::milli>(t2 - t0).count() / frames;
    for (size_t q = 0; q < tl_b.size() && q < tl_e.size(); q++) {
        float b = 0.0f, e2 = 0.0f;
        (void)hipEventElapsedTime(&b, tl_b[0], tl_b[q]); (void)hipEventElapsedTime(&e2, tl_b[0], tl_e[q]);
        fprintf(stderr, "  trace %2zu (slot %zu): begin %7.3f ms  end %7.3f ms  duration %6.3f\n", q, q % (size_t)K, b, e2, e2 - b);
    }
    for (hipEvent_t ev : tl_b) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : tl_e) (void)hipEventDestroy(ev);
    cleanup();
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

// the halo lists of (hiW, hiH, rank, world) as pure host code (CPU tests hold them to tiles.py): counts per peer, then the pixel lists
int ycge_host_halo_layout(int32_t hiW, int32_t hiH, int32_t rank, int32_t world, int64_t *send_counts, int64_t *recv_counts, uint32_t *send_px, uint32_t *recv_px, int64_t capacity)
try {
    if (hiW <= 0 || hiH <= 0 || world < 1 || rank < 0 || rank >= world || !send_counts || !recv_counts) return YCGE_ERR_INVALID_ARG;
    std::vector<int64_t> sc, rcv; std::vector<uint32_t> spx, rpx;
    halo_layout(hiW, hiH, rank, world, sc, rcv, spx, rpx);
    for (int r = 0; r < world; r++) { send_counts[r] = sc[(size_t)r]; recv_counts[r] = rcv[(size_t)r]; }
    if (send_px && (int64_t)spx.size() <= capacity) std::memcpy(send_px, spx.data(), spx.size() * 4);
    if (recv_px && (int64_t)rpx.size() <= capacity) std::memcpy(recv_px, rpx.data(), rpx.size() * 4);
    return ((int64_t)spx.size() <= capacity && (int64_t)rpx.size() <= capacity) || (!send_px && !recv_px) ? YCGE_OK : YCGE_ERR_INVALID_ARG;
}
catch (...) { return ycge_host::abi_catch(nullptr); }

int ycge_read_buffer(ycge_ctx *c, int32_t which, void *dst, size_t bytes)
try {
    if (!c) return YCGE_ERR_INVALID_ARG;
    if (!dst) return c->fail(YCGE_ERR_INVALID_ARG, "null destination");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->hiW * c->hiH;
    const void *src = nullptr; size_t want = 0;
    switch (which) {
    case YCGE_BUF_RAYS: src = c->dbg_rays.p; want = n * 24; break;
    case YCGE_BUF_PRIM_ID: src = c->dbg_prim.p; want = n * 4; break;
    case YCGE_BUF_SUB_ID: src = c->dbg_sub.p; want = n * 4; break;
    case YCGE_BUF_HIT_T: src = c->dbg_hit_t.p; want = n * 4; break;
    case YCGE_BUF_RNG_STATE: src = c->dbg_rng.p; want = n * 8; break;
    case YCGE_BUF_CURRENT_HDR: src = c->current_hdr.p; want = n * 12; break;
    case YCGE_BUF_G_ALBEDO: src = c->g_albedo.p; want = n * 12; break;
    case YCGE_BUF_G_NORMAL: src = c->g_normal.p; want = n * 12; break;
    case YCGE_BUF_G_DEPTH: src = c->g_depth.p; want = n * 4; break;
    case YCGE_BUF_SKY_MASK: src = c->sky.p; want = n; break;
    case YCGE_BUF_TAA_HISTORY: src = c->taa_hist.p; want = n * 12; break;
    case YCGE_BUF_PREV_NORMAL: src = c->prev_normal.p; want = n * 12; break;
    case YCGE_BUF_PREV_DEPTH: src = c->prev_depth.p; want = n * 4; break;
    case YCGE_BUF_PREV_SKY: src = c->prev_sky.p; want = n; break;
    case YCGE_BUF_DENOISED:
        if (!c->denoised) return c->fail(YCGE_ERR_INVALID_ARG, "no denoised frame yet: render with an SDR output buffer first");
        src = c->denoised; want = n * 12; break;
    default: return c->fail(YCGE_ERR_INVALID_ARG, "unknown buffer %d", which);
    }
    if (!src) return c->fail(YCGE_ERR_INVALID_ARG, "buffer %d needs config.capture_debug", which);
    if (bytes != want) return c->fail(YCGE_ERR_INVALID_ARG, "buffer %d is %zu bytes, caller passed %zu", which, want, bytes);
    { const int jr = join_async(c); if (jr != YCGE_OK) return jr; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return copy_out(c, dst, src, want);
}
catch (...) { return ycge_host::abi_catch(c); }

static int accel_view(ycge_ctx *c, int32_t which, int32_t index, const void **p, size_t *n)
{
    if (c->scene_tree_on_device && (which == YCGE_ACCEL_SCENE_NODES || which == YCGE_ACCEL_SCENE_LEAF_INDEX)) {
        // the tree was built on the device: fetch the reference-format copy the first time somebody asks for it
        c->scene_tree.nodes.resize((size_t)c->dev_tree_nodes);
        c->scene_tree.leaf_index.resize((size_t)c->dev_tree_items);
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            copy_out(c, c->scene_tree.nodes.data(), c->d_bvh_ref.p, (size_t)c->dev_tree_nodes * sizeof(RefNode)) != YCGE_OK ||
            copy_out(c, c->scene_tree.leaf_index.data(), c->d_scene_leaf.p, (size_t)c->dev_tree_items * 4) != YCGE_OK)
            return -1;
        c->scene_tree.root = 0;
        c->scene_tree_on_device = false;
    }
    switch (which) {
    case YCGE_ACCEL_SCENE_NODES: *p = c->scene_tree.nodes.data(); *n = c->scene_tree.nodes.size() * sizeof(RefNode); return 0;
    case YCGE_ACCEL_SCENE_LEAF_INDEX: *p = c->scene_tree.leaf_index.data(); *n = c->scene_tree.leaf_index.size() * 4; return 0;
    case YCGE_ACCEL_MESH_NODES:
        if (index < 0 || index >= (int)c->meshes.size()) return -1;
        *p = c->meshes[index].tree.nodes.data(); *n = c->meshes[index].tree.nodes.size() * sizeof(RefNode); return 0;
    case YCGE_ACCEL_MESH_LEAF_INDEX:
        if (index < 0 || index >= (int)c->meshes.size()) return -1;
        *p = c->meshes[index].tree.leaf_index.data(); *n = c->meshes[index].tree.leaf_index.size() * 4; return 0;
    }
    return -1;
}
int ycge_accel_size(ycge_ctx *c, int32_t which, int32_t index, size_t *bytes)
try {
    const void *p; size_t n;
    if (!c || !bytes) return YCGE_ERR_INVALID_ARG;
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "no scene uploaded");
    if (accel_view(c, which, index, &p, &n)) return c->fail(YCGE_ERR_INVALID_ARG, "bad accel selector");
    *bytes = n;
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }
int ycge_read_accel(ycge_ctx *c, int32_t which, int32_t index, void *dst, size_t bytes)
try {
    const void *p; size_t n;
    if (!c || !dst) return YCGE_ERR_INVALID_ARG;
    if (!c->have_scene) return c->fail(YCGE_ERR_NO_SCENE, "no scene uploaded");
    if (accel_view(c, which, index, &p, &n) || n != bytes) return c->fail(YCGE_ERR_INVALID_ARG, "bad accel selector or size");
    std::memcpy(dst, p, n);
    return YCGE_OK;
}
catch (...) { return ycge_host::abi_catch(c); }

} // extern "C"
