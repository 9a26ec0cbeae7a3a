// experiments/ycge_taa_in_trace.hip.h - TemporalBlendWithClamp INSIDE the trace launch.  Measured and rejected in round 6 (profiles/r06/b_taa_in_trace.txt):
// bit-exact - tests/test_gpu_parity.py::test_taa_inside_the_trace_launch_is_bit_exact holds it to the oracle through lib/var_experiments.so - and
// slower: config 4's launch 0.48 -> 0.71 ms with the window staged in LDS (0.91 with every tap a coherent load).  TAA is latency: three dependent
// memory phases a block, hidden by k_taa's 8 000 resident wavefronts and not by a k_trace wavefront that is one of 3 072.  Compiled only with
// -DYCGE_EXPERIMENTS=1; the host asks for it with YCGE_TAA_FUSE=1 (ycge_host.cpp: fuse_request) and the product build never honours it.
#pragma once
namespace ycge {
// TemporalBlendWithClamp inside the trace launch (TaaFuse; the synchronous single-device frame, round 6).  TAA of a pixel reads this frame's
// colour and sky flag of its (2 radius + 1)^2 window, radius <= 8: the 8 x 8 block's own pixels and one ring of the eight blocks around it.
// So a block's TAA can run the moment the nine blocks of its neighbourhood have finished their pixels - and most neighbourhoods complete
// long before the launch ends on its last chains: 33 us of TAA kernel, its launch gap and a stream hop leave the frame (config 4:
// 0.536 -> see DESIGN section 8).  Protocol, per finishing wavefront:
//   1. its pixel stores are written THROUGH (sc1) and waited for (s_waitcnt vmcnt(0));
//   2. a part of a split block counts itself at part_ctr[block]; only the block's last part goes on (and puts the count back);
//   3. lanes 0..8 add 1 to block_ctr of the nine neighbours (device-scope atomics); the counter is monotonic and a frame adds exactly the
//      neighbourhood's size `need` to it, so the add that makes it a multiple of `need` is the LAST of this frame: that wavefront resolves
//      the neighbour (taa_block_resolve): the 10 x 10 window of this frame's colours and sky flags comes into LDS with a handful of coalesced
//      device-coherent (sc1) loads - the first form read every tap of every pixel that way, 44 strided sc1 loads a lane, and the launch took
//      0.91 ms instead of 0.48 (profiles/r06/b_taa_in_trace.txt: the written-through stores cost 10 us of it, the counters 3, the taps 420).
// Store sc1 -> atomic -> atomic -> load sc1 is the pairing profiles/micro/xcdvis.hip measured never stale across XCDs (0 of 2 000; every other
// pairing: 2 000 of 2 000).  Blocks wholly outside the image are launched too (a tile is 4 blocks) and count like the others.
// History and guide copies are touched by the resolving wavefront alone: same reads, same arithmetic, same writes as k_taa - bit-identical.
// TAA of the 8 x 8 block at (x0, y0) by one wavefront (radius 1 - the host fuses nothing else).  Entry (wy, wx) of the window is the pixel
// (clamp(y0 - 1 + wy), clamp(x0 - 1 + wx)): tap (dx, dy) of pixel (lx, ly) is entry (ly + 1 + dy, lx + 1 + dx) - the clamped tap of
// RaytraceRenderer.cs:344-350.  The window lives where the block's shading contexts lived (g_shade_ctx: every pixel is done).
__device__ __forceinline__ void taa_block_resolve(const TraceOut &O, const int x0, const int y0, const int lane)
{
    const TaaParams &T = O.taa.T;
    float *win = g_shade_ctx;                               // [10][10][3] floats, then [100] sky bytes
    uint8_t *wsky = (uint8_t *)(g_shade_ctx + 300);
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; };
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const int idx = it * 64 + lane;
        if (idx < 300) {
            const int p = idx / 3, comp = idx - 3 * p, wy = p / 10, wx = p - 10 * wy;
            const size_t j = (size_t)clampi(x0 - 1 + wx, T.w - 1) + (size_t)clampi(y0 - 1 + wy, T.h - 1) * T.w;
            win[idx] = __hip_atomic_load(O.current_hdr + 3 * j + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int p = it * 64 + lane;
        if (p < 100) {
            const int wy = p / 10, wx = p - 10 * wy;
            const size_t j = (size_t)clampi(x0 - 1 + wx, T.w - 1) + (size_t)clampi(y0 - 1 + wy, T.h - 1) * T.w;
            wsky[p] = __hip_atomic_load(O.sky + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int lx = lane & 7, ly = lane >> 3, x = x0 + lx, y = y0 + ly;
    if (x >= T.w || y >= T.h) return;
    const size_t i = (size_t)x + (size_t)y * T.w;
    const float nx = __hip_atomic_load(O.g_normal + 3 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ny = __hip_atomic_load(O.g_normal + 3 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                nz = __hip_atomic_load(O.g_normal + 3 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float z_now = __hip_atomic_load(O.g_depth + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c = (ly + 1) * 10 + lx + 1;
    const float cr = win[3 * c], cg = win[3 * c + 1], cb = win[3 * c + 2];
    const uint8_t sky_now = wsky[c];
    if (T.reset) { taa_reset(i, cr, cg, cb, nx, ny, nz, z_now, sky_now, O.taa.hist, O.taa.prev_normal, O.taa.prev_depth, O.taa.prev_sky); return; }
    float min_l = YCGE_INF, max_l = -YCGE_INF;
#pragma unroll
    for (int k = 0; k < 9; k++) {           // (ky, kx) order as the reference's loops; same comparisons in the same order as taa_pixel
        const int e = (ly + k / 3) * 10 + lx + k % 3;
        const float l = luma(win[3 * e], win[3 * e + 1], win[3 * e + 2]);
        const bool use = wsky[e] == sky_now;
        if (use && l < min_l) min_l = l;
        if (use && l > max_l) max_l = l;
    }
    taa_blend(T, i, cr, cg, cb, nx, ny, nz, z_now, sky_now, min_l, max_l, O.taa.hist, O.taa.prev_normal, O.taa.prev_depth, O.taa.prev_sky);
}

__device__ __forceinline__ void taa_in_trace(const FrameParams &P, const TraceOut &O, const uint32_t bid, const uint32_t lg, const int lane)
{
#ifndef YCGE_TAAFUSE_DBG
#define YCGE_TAAFUSE_DBG 0      // TIMING ONLY (no TAA happens): 1 = the written-through stores alone, 2 = + the counters
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (YCGE_TAAFUSE_DBG == 1) return;
    if (lg) {
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(O.taa.part_ctr + bid, 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old + 1u != (1u << lg)) return;
        if (lane == 0) __hip_atomic_store(O.taa.part_ctr + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int nbx = P.tiles_x * 4, nby = P.tiles_y;
    const int bx = (int)(bid % (uint32_t)nbx), by = (int)(bid / (uint32_t)nbx);
    const int j = lane < 9 ? lane : 4;
    const int qx = bx + j % 3 - 1, qy = by + j / 3 - 1;
    bool win = false;
    if (lane < 9 && qx >= 0 && qy >= 0 && qx < nbx && qy < nby) {
        const int x0 = qx > 0 ? qx - 1 : 0, x1 = qx + 1 < nbx ? qx + 1 : nbx - 1, y0 = qy > 0 ? qy - 1 : 0, y1 = qy + 1 < nby ? qy + 1 : nby - 1;
        const uint32_t need = (uint32_t)((x1 - x0 + 1) * (y1 - y0 + 1));
        const uint32_t old = atomicAdd(O.taa.block_ctr + (size_t)qy * nbx + qx, 1u);
        win = (old + 1u) % need == 0u;
    }
    unsigned long long m = __ballot(win);
    asm volatile("" ::: "memory");          // (nothing of the resolve moves in front of the counters)
    if (YCGE_TAAFUSE_DBG == 2) return;
    while (m) {
        const int w = __builtin_ctzll(m);
        m &= m - 1ull;
        const int rx = bx + w % 3 - 1, ry = by + w / 3 - 1;
        taa_block_resolve(O, rx * 8, ry * 8, lane);
    }
}

} // namespace ycge
