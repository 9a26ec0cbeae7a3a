// ycge_post.hip — the steps of TryFlipAndBlit after TAA (reference RayTracing/RaytraceRenderer.cs:221-264):
//   ApplyAtrousDenoise (:622-722), ToneMapper.UpdateExposure (ToneMapper.cs:49-91), the ss x ss box average of the
//   top / bottom half-cells and ToneMapper.MapPixel (:229-264, ToneMapper.cs:204-260).  SURVEY.md section 8, row f1.
//
// Everything here reproduces the reference's floating-point ORDER, not just its formulas:
//   * a pixel's 25 taps are accumulated in (ky, kx) order;
//   * odd A-trous iterations run IN PLACE in the reference (buffer swap at :718: iter 0 src->A, iter 1 A->A,
//     iter 2 A->B), so a pixel sees the NEW values of the neighbours that precede it in scan order and the OLD
//     values of those that follow.  That recurrence is evaluated exactly by levels: the host derives, from the
//     stencil itself, the earliest level T(p) = 1 + max T(q) over every stencil-related pixel q that precedes p in
//     scan order (build_inplace_schedule); pixels of one level never read or write each other.  Levels run in order
//     inside a workgroup per band of rows, bands pipelined over consecutive launches (k_atrous_band);
//   * the auto-exposure is a serial fp32 sum over the sampled pixels in scan order: the log terms are produced
//     in parallel, the sum is one lane adding them one by one.
#include <hip/hip_runtime.h>

#include "ycge_rt.hip.h"

namespace ycge {

struct AtrousParams {
    int32_t w, h, step;
    float c_phi, n_phi, z_phi, a_phi;      // already max(1e-6f, phi), :693-696
};

struct ToneState {                          // ToneMapper fields that change (ToneMapper.cs:13,17)
    float ae_exposure;
    float effective;
    uint32_t count;                         // scratch: sampled pixels with lum > 0 this frame
    uint32_t pad;
};

__device__ __forceinline__ F3 ld3(const float *p, size_t i) { return f3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void st3(float *p, size_t i, F3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
// device-coherent forms (sc1): a store written through to memory, a load that never returns another XCD's stale copy - measured
// in profiles/micro/xcdvis.hip (store sc1 + load sc1: 0 stale reads of 2 000 across XCDs; every other pairing: 2 000 of 2 000)
__device__ __forceinline__ float ld_dev(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ F3 ld3_dev(const float *p, size_t i) { return f3(ld_dev(p + 3 * i), ld_dev(p + 3 * i + 1), ld_dev(p + 3 * i + 2)); }
__device__ __forceinline__ void st3_dev(float *p, size_t i, F3 v)
{
    __hip_atomic_store(p + 3 * i, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3 * i + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3 * i + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float luma3(F3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
__device__ __forceinline__ float kernel_tap(int k) { return k == 0 ? 3.0f / 8.0f : (k == 1 || k == -1) ? 1.0f / 4.0f : 1.0f / 16.0f; }   // :646

// normal[x, y].Normalized() is evaluated once per pixel per frame instead of once per tap (same function, same bits)
__global__ __launch_bounds__(256) void k_unit_normals(const float *__restrict__ normal, float *__restrict__ unit, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) st3(unit, i, normalized(ld3(normal, i)));
}

struct Center { F3 c0, a0, n0; float z0; uint8_t sky0; };

// one tap of the 5x5 stencil, :664-700, in two halves so that callers can issue every load before any arithmetic
__device__ __forceinline__ size_t atrous_tap_index(const AtrousParams &A, int x, int y, int kx, int ky)
{
    int sy = y + ky * A.step;
    if (sy < 0) sy = 0; else if (sy >= A.h) sy = A.h - 1;
    int sx = x + kx * A.step;
    if (sx < 0) sx = 0; else if (sx >= A.w) sx = A.w - 1;
    return (size_t)sx + (size_t)sy * A.w;
}
// m_exp(-d / phi).  exp(-0) is exactly 1 (m_exp_d: k = 0, r = -0, every Horner step p * -0 + c = c), and whole wavefronts agree on
// d == 0 all the time - one material (albedo distance 0), a flat axis-aligned surface (normal distance 0), the centre tap (luminance,
// depth, albedo): then the ~60 binary64 instructions are skipped.  Per wavefront, never per lane: a lane saves nothing by itself.
__device__ __forceinline__ float exp_neg_ratio(float d, float phi)
{
    if (!__any(d != 0.0f)) return 1.0f;
    return m_exp(-d / phi);
}
__device__ __forceinline__ float atrous_tap_weight(const AtrousParams &A, int kx, int ky, const Center &C, F3 c, F3 a, F3 n, float z)
{
    const float wy = kernel_tap(ky);
    const float wx = kernel_tap(kx);
    const float w_base = wx * wy;
    const float lum0 = luma3(C.c0);
    const float lum = luma3(c);
    const float dl = cs_abs(lum - lum0);
    const float dn = cs_max(0.0f, 1.0f - dot(C.n0, n));
    const float dz = cs_abs(z - C.z0);
    const float da = cs_abs(a.x - C.a0.x) + cs_abs(a.y - C.a0.y) + cs_abs(a.z - C.a0.z);
    const float wc = exp_neg_ratio(dl, A.c_phi);
    const float wn = exp_neg_ratio(dn, A.n_phi);
    const float wz = exp_neg_ratio(dz, A.z_phi);
    const float wa = exp_neg_ratio(da, A.a_phi);
    return w_base * wc * wn * wz * wa;
}
// Returns false for a tap the reference skips (`continue`).
__device__ __forceinline__ bool atrous_tap(const AtrousParams &A, const float *cur, const float *albedo, const float *unit_n, const float *depth,
                                           const uint8_t *sky, int x, int y, int kx, int ky, const Center &C, float &wght, F3 &c)
{
    const size_t j = atrous_tap_index(A, x, y, kx, ky);
    if (sky[j] != C.sky0) return false;
    c = ld3(cur, j);
    wght = atrous_tap_weight(A, kx, ky, C, c, ld3(albedo, j), ld3(unit_n, j), depth[j]);
    return true;
}

// an iteration whose source and destination differ: one thread per pixel, taps in order
__global__ __launch_bounds__(256) void k_atrous(const AtrousParams A, const float *__restrict__ cur, float *__restrict__ dst,
                                                const float *__restrict__ albedo, const float *__restrict__ unit_n,
                                                const float *__restrict__ depth, const uint8_t *__restrict__ sky)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= A.w || y >= A.h) return;
    const size_t i = (size_t)x + (size_t)y * A.w;
    Center C;
    C.c0 = ld3(cur, i);
    C.sky0 = sky[i];
    if (C.sky0) { st3(dst, i, C.c0); return; }
    C.a0 = ld3(albedo, i); C.n0 = ld3(unit_n, i); C.z0 = depth[i];
    float wsum = 0.0f;
    F3 accum = f3(0, 0, 0);
    for (int ky = -2; ky <= 2; ky++) {
        // a stencil row's five taps are fetched before any of them is looked at (atrous_tap asks for a tap's sky flag, waits, and only
        // then for its data: 25 dependent round trips per pixel); weights and sums as before, in (ky, kx) order
        size_t j[5];
        uint8_t sj[5];
        F3 cj[5], aj[5], nj[5];
        float zj[5];
#pragma unroll
        for (int k = 0; k < 5; k++) j[k] = atrous_tap_index(A, x, y, k - 2, ky);
#pragma unroll
        for (int k = 0; k < 5; k++) { sj[k] = sky[j[k]]; cj[k] = ld3(cur, j[k]); aj[k] = ld3(albedo, j[k]); nj[k] = ld3(unit_n, j[k]); zj[k] = depth[j[k]]; }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (sj[k] != C.sky0) continue;
            const float wght = atrous_tap_weight(A, k - 2, ky, C, cj[k], aj[k], nj[k], zj[k]);
            accum = f3(accum.x + cj[k].x * wght, accum.y + cj[k].y * wght, accum.z + cj[k].z * wght);
            wsum += wght;
        }
    }
    if (wsum > 1e-8f) {
        const float inv = 1.0f / wsum;
        st3(dst, i, f3(accum.x * inv, accum.y * inv, accum.z * inv));
    } else {
        st3(dst, i, C.c0);
    }
}

// The three factors of a tap's weight that do not depend on the colour buffer (normal, depth, albedo terms of
// atrous_tap_weight, the same expressions): an in-place iteration changes colours only, so these are computed for every
// (pixel, tap) by the whole chip before the iteration starts instead of inside its serial chain of levels - there a band
// is ONE workgroup, and four binary64 exponentials per tap made a level's time the fp64 issue rate of one CU.
// statw[(p * 25 + tap) * 3 + {0, 1, 2}] = wn, wz, wa.
// The three factors are symmetric in the two pixels of a tap - the dot product's terms commute, |a - b| = |b - a| bit for bit - so
// a pair of pixels is evaluated ONCE, by the one that comes first in scan order, and stored in both records: tap t of p and tap
// 24 - t of its partner.  16 lanes per pixel: lanes 0..12 take the centre tap and the 12 forward ones; a backward tap has no such
// partner only where the border clamps it (the partner's forward tap would then not lead back here), and there the lane computes
// it itself.  Half the binary64 exponentials, four pixels to a wavefront instead of two: 0.45 -> 0.32 ms at 1080p (beside iteration 0).
__device__ __forceinline__ void static_factors(const AtrousParams &A, const float *albedo, const float *unit_n, const float *depth,
                                               size_t p, size_t j, float &wn, float &wz, float &wa)
{
    const F3 n0 = ld3(unit_n, p), nj = ld3(unit_n, j), a0 = ld3(albedo, p), aj = ld3(albedo, j);
    const float dn = cs_max(0.0f, 1.0f - dot(n0, nj));
    const float dz = cs_abs(depth[j] - depth[p]);
    const float da = cs_abs(aj.x - a0.x) + cs_abs(aj.y - a0.y) + cs_abs(aj.z - a0.z);
    wn = exp_neg_ratio(dn, A.n_phi);
    wz = exp_neg_ratio(dz, A.z_phi);
    wa = exp_neg_ratio(da, A.a_phi);
}
__global__ __launch_bounds__(256) void k_atrous_static(const AtrousParams A, const float *__restrict__ albedo, const float *__restrict__ unit_n,
                                                       const float *__restrict__ depth, const uint8_t *__restrict__ sky, float *__restrict__ statw,
                                                       size_t n)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t p = gid >> 4;
    const int l = (int)(gid & 15);
    if (p >= n || sky[p]) return;
    const int x = (int)(p % (size_t)A.w), y = (int)(p / (size_t)A.w);
    if (l < 13) {                               // the centre tap (12) and the forward taps 13..24
        const int t = 12 + l;
        const int kx = t % 5 - 2, ky = t / 5 - 2;
        const size_t j = atrous_tap_index(A, x, y, kx, ky);
        if (!sky[j]) {                          // (a tap the reference skips: its weight is never read)
            float wn, wz, wa;
            static_factors(A, albedo, unit_n, depth, p, j, wn, wz, wa);
            float *o = statw + (p * 25 + (size_t)t) * 3;
            o[0] = wn; o[1] = wz; o[2] = wa;
            const int sx = x + kx * A.step, sy = y + ky * A.step;
            if (l > 0 && sx >= 0 && sx < A.w && sy >= 0 && sy < A.h) {      // not clamped: this pixel is tap 24 - t of j
                float *o2 = statw + (j * 25 + (size_t)(24 - t)) * 3;
                o2[0] = wn; o2[1] = wz; o2[2] = wa;
            }
        }
    }
    if (l < 12) {                               // a backward tap the border clamps: nobody else writes it
        const int t = l;
        const int kx = t % 5 - 2, ky = t / 5 - 2;
        const int sx = x + kx * A.step, sy = y + ky * A.step;
        if (sx < 0 || sx >= A.w || sy < 0 || sy >= A.h) {
            const size_t j = atrous_tap_index(A, x, y, kx, ky);
            if (!sky[j]) {
                float wn, wz, wa;
                static_factors(A, albedo, unit_n, depth, p, j, wn, wz, wa);
                float *o = statw + (p * 25 + (size_t)t) * 3;
                o[0] = wn; o[1] = wz; o[2] = wa;
            }
        }
    }
}

// ---- in-place iteration.  The image is cut into bands of whole rows, one 1024-thread workgroup per band; a band's pixels
// are listed level by level, every level padded to whole PASSES of 32 pixels (0xffffffff = no pixel).  A pass gives each
// pixel a 32-lane group: lanes 0..24 evaluate one tap each, lanes 0..3 then add the 25 products of one component
// (x, y, z, weight) in tap order.  Launch `launch` runs, for band b, the passes of its levels [K (launch - b),
// K (launch - b) + K) one after the other.  Two stencil-related pixels are either in the same band (same workgroup:
// ordered by pass) or in adjacent bands, and then the one that comes first in scan order is in the upper band and has
// the smaller level, hence a strictly smaller launch index band + level / K: kernel boundaries order them.
//
// What a pass costs is its chain of dependent operations, 2 600 times per 1080p frame, so the chain is kept short:
//  * everything a pass reads from memory - pixel index, colours of the pixel and of its taps, sky flags, the static
//    weights - is fetched ONE PASS AHEAD (two for the index), while the previous pass computes;
//  * the values a launch writes stay in an LDS hash table (pixel -> new colour) until the launch ends: a tap whose
//    source was rewritten earlier in this launch takes it from there, so a pass never waits for a store to reach L2,
//    and the prefetch may read a stale colour for exactly those pixels (the table wins);
//  * the 25 products of a pixel are written and read by ONE wavefront (a group is half of it): no workgroup barrier
//    between the taps and the sum, one barrier per pass (table visible to the next pass), and it is a bare
//    s_barrier - the fenced __syncthreads would wait for the prefetches in flight.
// Pixels per pass = the workgroup's size / 32.  A level of a 16-row band holds 12 pixels on average and never more than 32; a pass is
// bound by the instruction issue of its wavefronts on ONE CU (4 per SIMD at 32 pixels: 1.45 us), so a narrower workgroup - two passes
// for the rare wide level - is faster: template parameter G (8, 16 or 32 pixels; knob YCGE_POST_GROUPS).
#define YCGE_POST_HASH 2048         // entries of the hash form; the host keeps a launch's pixels below 3/4 of it
#define YCGE_POST_WIN 2048          // entries of the window form: rows of the band x window width
#define YCGE_POST_NONE 0xffffffffu
#define YCGE_POST_TICKET_WORD 16       // k_atrous_stream, bands in order of arrival: the counter (a free word of band 0's progress record; only ever incremented)
#define YCGE_POST_PROBE_LEVEL 1400    // profiling aid of k_atrous_stream: times of this level's hand-over (profiles/post_bands.py)
// Where a launch keeps its new colours.  WINDOW form (the default): entry (row in the band) * WX + (x mod WX) - the host has checked,
// list by list, that no two pixels one launch writes share an entry (a launch's levels cover a short diagonal stripe of the band: 16
// columns per row at step 2), so a lookup is one LDS read at an address known a pass ahead and an insert is a plain store: no probe
// loop, no LDS atomic and no cross-lane broadcast of the slot in the chain.  HASH form: open addressing by pixel index, for a schedule
// the window does not fit (never seen; kept as the general case).
template <int G> struct PostSharedT {
    float val[G][4][28];        // [component x, y, z, weight][tap], a row padded to 16-byte multiples: the sum reads its 25 terms as 6 x b128 + 1
    uint4 ent[YCGE_POST_HASH];  // {pixel (tag), r, g, b as bits}: a lookup is ONE 16-byte LDS read
    uint32_t out_slot[2][G];    // k_atrous_stream: the entry each group wrote in the last two passes (YCGE_POST_NONE: none), for the publishing wavefront
    uint32_t ticket;            // k_atrous_stream, bands in order of arrival: the band this workgroup drew
};
static_assert(YCGE_POST_WIN == YCGE_POST_HASH, "one LDS array serves both forms");
struct BandWindow { int y0, rows; uint32_t wx, use; int shift; };     // use == 0: hash form; shift 1: the band's rows are y0, y0 + 2, .. (k_atrous_stream's half-bands)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t post_hash(uint32_t p) { return (p * 2654435761u) >> (32 - 11); }
struct PassData {           // what one lane needs for one pass, fetched ahead
    F3 c0, cj;
    float wn, wz, wa;
    uint32_t p, j;          // pixel index and this lane's tap source
    uint32_t pslot, jslot;  // window form: their entries (YCGE_POST_NONE: the tap's row is outside the band)
    uint8_t sky0, sky_j;
};
template <bool COH = false, bool COH_ALL = false>
__device__ __forceinline__ PassData pass_fetch(const AtrousParams &A, const BandWindow &W, const float *buf, const float *statw, const uint8_t *sky,
                                               uint32_t p, int t, int sky_known = -1)
{
    PassData D;
    if (sky_known > 0 || (sky_known >= 0 && p == YCGE_POST_NONE)) {      // k_atrous_stream knows the flag a pass early: a sky pixel (or no pixel) fetches nothing
        D.sky0 = 1;             // nothing else of D is read for a sky pixel (pass_compute: work = false), so nothing else is set
        return D;
    }
    const uint32_t e = p == YCGE_POST_NONE ? 0u : p;        // list entries are x | y << 16: no division in the chain
    const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
    const uint32_t pp = (uint32_t)x + (uint32_t)y * (uint32_t)A.w;
    D.p = pp;
    int sx = x, sy = y;
    if (p != YCGE_POST_NONE && t < 25) {                    // atrous_tap_index, coordinates kept
        sy = y + (t / 5 - 2) * A.step; if (sy < 0) sy = 0; else if (sy >= A.h) sy = A.h - 1;
        sx = x + (t % 5 - 2) * A.step; if (sx < 0) sx = 0; else if (sx >= A.w) sx = A.w - 1;
    }
    D.j = (uint32_t)sx + (uint32_t)sy * (uint32_t)A.w;
    const int rj = sy - W.y0, rp = y - W.y0;             // (a half-band's taps stay on its own row parity, or leave its rows upwards / downwards)
    D.jslot = (rj >= 0 && (rj >> W.shift) < W.rows) ? (uint32_t)(rj >> W.shift) * W.wx + ((uint32_t)sx & (W.wx - 1u)) : YCGE_POST_NONE;
    D.pslot = ((uint32_t)(rp >> W.shift) * W.wx + ((uint32_t)x & (W.wx - 1u))) & (YCGE_POST_WIN - 1u);
    // 32-bit byte offsets against the (scalar) array bases: one address register per load instead of 64-bit arithmetic per lane
    // (the host keeps the in-place form to grids whose weight table stays below 4 GB: 14.3 M pixels)
    const float *c0p = (const float *)((const char *)buf + 12u * pp), *cjp = (const float *)((const char *)buf + 12u * D.j);
    D.sky0 = sky_known >= 0 ? (uint8_t)sky_known : *(const uint8_t *)((const char *)sky + pp);
    D.c0 = COH_ALL ? ld3_dev(c0p, 0) : ld3(c0p, 0);     // (own rows in k_atrous_stream: what this launch has rewritten comes from the LDS table, the rest is old in every cache)
    D.sky_j = *(const uint8_t *)((const char *)sky + D.j);
    D.cj = (COH_ALL || (COH && rj < 0)) ? ld3_dev(cjp, 0) : ld3(cjp, 0);       // ... a tap above the band is the only colour this launch reads that another workgroup writes
    const float *sw = (const float *)((const char *)statw + (pp * 25u + (uint32_t)(t < 25 ? t : 0)) * 12u);
    D.wn = sw[0]; D.wz = sw[1]; D.wa = sw[2];
    return D;
}
template <bool WIN, class SH>
__device__ __forceinline__ void pass_compute(const AtrousParams &A, uint32_t p, const PassData &D, SH &sh, uint32_t *out_entry = nullptr /* this group's */)
{
    const int g = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int kx = t % 5 - 2, ky = t / 5 - 2;
    const bool work = p != YCGE_POST_NONE && !D.sky0;        // sky pixel: dst[x, y] = cur[x, y] on the same buffer, nothing to do
    const bool valid = work && t < 25 && D.sky_j == D.sky0;
    if (valid) {
        F3 cj = D.cj;
        if (WIN) {                                           // rewritten earlier in this launch?
            if (D.jslot != YCGE_POST_NONE) {
                const uint4 en = sh.ent[D.jslot];
                if (en.x == D.j) cj = f3(__uint_as_float(en.y), __uint_as_float(en.z), __uint_as_float(en.w));
            }
        } else {
            for (uint32_t h = post_hash(D.j);; h = (h + 1u) & (YCGE_POST_HASH - 1u)) {
                const uint4 en = sh.ent[h];
                if (en.x == D.j) { cj = f3(__uint_as_float(en.y), __uint_as_float(en.z), __uint_as_float(en.w)); break; }
                if (en.x == YCGE_POST_NONE) break;
            }
        }
        // atrous_tap_weight with its three colour-independent exponentials read back: w_base * wc * wn * wz * wa, left to right
        const float w_base = kernel_tap(kx) * kernel_tap(ky);
        const float dl = cs_abs(luma3(cj) - luma3(D.c0));
        const float wc = m_exp(-dl / A.c_phi);
        const float wght = w_base * wc * D.wn * D.wz * D.wa;
        sh.val[g][0][t] = cj.x * wght; sh.val[g][1][t] = cj.y * wght; sh.val[g][2][t] = cj.z * wght; sh.val[g][3][t] = wght;
    } else if (work && t < 25) {
        // a tap the reference skips contributes +0: the running sums start at +0.0f and can therefore never be -0.0f, so
        // s + 0.0f == s for every value they take - no per-tap select in the chain of adds below
        sh.val[g][0][t] = 0.0f; sh.val[g][1][t] = 0.0f; sh.val[g][2][t] = 0.0f; sh.val[g][3][t] = 0.0f;
    }
    // the group's 25 products were written by this wavefront: in-order LDS, no barrier.  Lane c < 4 reads the 25 terms of component c
    // (seven reads issued together), then adds them in tap order
    float acc = 0.0f;
    if (work && t < 4) {
        float v[28];
        const float4 *row = (const float4 *)&sh.val[g][t][0];
#pragma unroll
        for (int k = 0; k < 6; k++) { const float4 q = row[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
        v[24] = sh.val[g][t][24];
#pragma unroll
        for (int k = 0; k < 25; k++) acc = acc + v[k];
    }
    if (WIN) {
        // lanes 0..2 hold the colour sums, lane 3 the weight sum: one DPP move hands it to its quad, every lane of the quad stores one word
        const float wsum = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xFF /* quad_perm 3,3,3,3 */, 0xF, 0xF, true));
        const bool changed = work && wsum > 1e-8f;          // else dst = c0: unchanged, nothing to record
        if (changed && t < 4) {
            const float inv = 1.0f / wsum;
            (&sh.ent[D.pslot].x)[(t + 1) & 3] = t < 3 ? __float_as_uint(acc * inv) : D.p;
        }
        if (out_entry && t == 3) *out_entry = changed ? D.pslot : YCGE_POST_NONE;      // k_atrous_stream: for the publishing wavefront
    } else {
        const float wsum = __shfl(acc, (threadIdx.x & 32) + 3, 64);
        uint32_t h = 0;
        const bool changed = work && wsum > 1e-8f;
        if (changed && t == 0) {
            for (h = post_hash(D.p);; h = (h + 1u) & (YCGE_POST_HASH - 1u)) {
                const uint32_t prev = atomicCAS(&sh.ent[h].x, YCGE_POST_NONE, D.p);
                if (prev == YCGE_POST_NONE || prev == D.p) break;
            }
        }
        h = (uint32_t)__shfl((int)h, threadIdx.x & 32, 64);
        if (changed && t < 3) {
            const float inv = 1.0f / wsum;
            (&sh.ent[h].y)[t] = __float_as_uint(acc * inv);
        }
    }
    lds_barrier();              // the table holds this pass's pixels before the next pass looks its taps up
}

// pixels: the padded pass list; off[b * (levels + 1) + t] = first pass of level t of band b (.. + 1: one past its last)
template <int G, bool WIN>
__global__ __launch_bounds__(32 * G) void k_atrous_band(const AtrousParams A, float *__restrict__ buf, const float *__restrict__ statw,
                                                        const uint8_t *__restrict__ sky, const uint32_t *__restrict__ pixels,
                                                        const uint32_t *__restrict__ off, int levels, int K, int launch, int first_band,
                                                        int rows_per_band, uint32_t wx)
{
    __shared__ __attribute__((aligned(16))) PostSharedT<G> sh;
    const int b = first_band + (int)blockIdx.x;
    const int g0 = launch - b;
    if (g0 < 0) return;
    const int t0 = g0 * K, t1 = t0 + K < levels ? t0 + K : levels;
    const uint32_t *o = off + (size_t)b * (levels + 1);
    const uint32_t pass_lo = o[t0], pass_hi = o[t1];
    if (pass_lo >= pass_hi) return;
    const int grp = threadIdx.x >> 5, t = threadIdx.x & 31;
    const BandWindow W = {b * rows_per_band, rows_per_band, WIN ? wx : 1u, WIN ? 1u : 0u, 0};
    const uint32_t n_ent = WIN ? (uint32_t)rows_per_band * wx : (uint32_t)YCGE_POST_HASH;
    for (uint32_t e = threadIdx.x; e < n_ent; e += 32 * G) sh.ent[e].x = YCGE_POST_NONE;
    uint32_t p1 = pixels[(size_t)pass_lo * G + grp];
    uint32_t p2 = pass_lo + 1 < pass_hi ? pixels[(size_t)(pass_lo + 1) * G + grp] : YCGE_POST_NONE;
    PassData D1 = pass_fetch(A, W, buf, statw, sky, p1, t);
    lds_barrier();              // table cleared
    for (uint32_t i = pass_lo; i < pass_hi; i++) {
        const uint32_t p3 = i + 2 < pass_hi ? pixels[(size_t)(i + 2) * G + grp] : YCGE_POST_NONE;
        const PassData D2 = pass_fetch(A, W, buf, statw, sky, p2, t);      // in flight while this pass computes
        pass_compute<WIN>(A, p1, D1, sh);
        p1 = p2; D1 = D2; p2 = p3;
    }
    // the launch's new colours go to memory together; the kernel boundary publishes them
    for (uint32_t e = threadIdx.x; e < n_ent; e += 32 * G) {
        const uint4 en = sh.ent[e];
        if (en.x != YCGE_POST_NONE) st3(buf, en.x, f3(__uint_as_float(en.y), __uint_as_float(en.z), __uint_as_float(en.w)));
    }
}

// The whole in-place iteration in ONE launch: a workgroup per band walks all of its level groups; before a group it waits for the
// band above to have finished the same group (progress[band] = epoch + groups finished, one 128-byte line per band), after it the
// new colours go out and the count goes up.  The launch form pays a kernel boundary per group AND lets a band start a group only a
// whole launch after its neighbour (8 levels of skew per band where the stencil needs 12 levels of offset anyway: 135 x 8 + 2 581
// level times); here a band trails its neighbour by what the data needs.  What makes it possible without cache-wide fences
// (a buffer_wbl2 / buffer_inv pair per group cost more than the launches, round 1): colours are stored write-through and loaded
// with device-coherent loads (sc1), which cost 25 ns more than plain ones when the line is in L2 (profiles/micro/ldflavour.hip)
// and are never stale (profiles/micro/xcdvis.hip).  Why the order is right: as in k_atrous_band - a tap in the band above is
// earlier in scan order and has a smaller level, hence a group <= this one, finished and published before this group starts; a
// tap in the band below must be read OLD, and that band does not start the group that rewrites it before this band has published
// the same group.  All bands must be resident at once (one workgroup each; the host checks the count against the chip).
// xcd_local: band = (block % 8) * per_xcd + block / 8 keeps neighbouring bands on one XCD under round-robin dispatch (their
// colours then meet in one L2); correctness does not depend on where a workgroup lands.
#if YCGE_EXPERIMENTS
} // namespace ycge
#include "experiments/ycge_atrous_persist_groups.hip.h"
namespace ycge {
#endif

// The persistent form with LEVEL-granular hand-over.  In k_atrous_persist a band starts a group when the band above has FINISHED
// the same group: it trails by a group plus the hand-over, like the launch form.  The data needs far less - a pixel of level T
// reads the band above up to level T - 1.  Here the LDS window is never cleared (an entry is reused 64 columns = 32 levels
// later, a tap reaches 8 levels back) and one extra wavefront per workgroup PUBLISHES: after the barrier of pass i it copies the
// colours that pass wrote from the window to memory (written through), waits for its own stores - off the computing wavefronts'
// chain, which a write-through acknowledgement (0.4 us) or a device-coherent load of a cold line (1.3 us) would lengthen by a
// third each (measured with YCGE_POST_DBG) - and raises progress[band] = epoch + (the first level this band has NOT completed).
// A band fetches the taps of level T (one pass ahead of computing them) once the band above has published >= T; it reads that
// word one pass ahead as well, so in the steady state nothing waits.  Only the taps ABOVE the band are read device-coherently:
// what the band rewrote itself comes from the window, everything else it reads is old and right in any cache.
template <int G, bool PROF, bool DUO = false>
__global__ __launch_bounds__(32 * G + 64) void k_atrous_stream(const AtrousParams A, float *__restrict__ buf, const float *__restrict__ statw,
                                                               const uint8_t *__restrict__ sky, const uint32_t *__restrict__ pixels,
                                                               const uint32_t *__restrict__ off, const uint32_t *__restrict__ pass_level,
                                                               const int32_t *__restrict__ band_desc, int levels, int n_bands, int rows_per_band, uint32_t wx, uint32_t *__restrict__ progress, uint32_t epoch,
                                                               int xcd_local, uint32_t ticket_base)
{
    __shared__ __attribute__((aligned(16))) PostSharedT<G> sh;
    int b = (int)blockIdx.x;
    const bool dbg_free = (xcd_local & 2) != 0;         // timing experiment (YCGE_POST_DBG_FREE, wrong pixels): no band waits for the band above
    if (xcd_local & 1) { const int per_xcd = (n_bands + 7) / 8; b = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8; }
    else {
        // Bands in order of ARRIVAL, not by block index: a band only ever waits for bands of lower numbers, and with a drawn number
        // all of those have started - whatever order the dispatcher places workgroups in, and however many of them fit the chip at
        // once.  (By block index a launch with 28 workgroups too many stalled for 1 - 59 s, until the queue was preempted.)
        if (threadIdx.x == 0) sh.ticket = atomicAdd(progress + YCGE_POST_TICKET_WORD, 1u) - ticket_base;
        __syncthreads();
        b = __builtin_amdgcn_readfirstlane((int)sh.ticket);
    }
    if (b >= n_bands) return;
    const uint32_t *o = off + (size_t)b * (levels + 1);
    const bool publisher = threadIdx.x >= 32 * G;               // the last wavefront
    const int grp = threadIdx.x >> 5, t = threadIdx.x & 31;
    // the band: rows y0, y0 + stride, .. (rows of them), how many 32-lane groups its passes use, whom it waits for, who waits for it
    // (band_desc: the row-parity split of the host's split_band_layout; without it whole bands of rows_per_band rows in a chain)
    int y0 = b * rows_per_band, rows = rows_per_band, stride = 1, groups = G, up0 = b > 0 ? b - 1 : -1, up1 = -1, dn0 = b + 1 < n_bands ? b + 1 : -1, dn1 = -1;
    if (band_desc) {
        const int4 d0 = ((const int4 *)band_desc)[2 * b], d1 = ((const int4 *)band_desc)[2 * b + 1];
        y0 = d0.x; rows = d0.y; stride = d0.z; groups = d0.w; up0 = d1.x; up1 = d1.y; dn0 = d1.z; dn1 = d1.w;
    }
    const BandWindow W = {y0, rows, wx, 1u, stride == 2 ? 1 : 0};
    const uint32_t n_ent = (uint32_t)rows * wx;
    uint32_t *mine = progress + (size_t)b * 32;
    const bool has_up = up0 >= 0 && !dbg_free;
    const uint32_t *above = progress + (size_t)(has_up ? up0 : 0) * 32, *above2 = progress + (size_t)(up1 >= 0 ? up1 : has_up ? up0 : 0) * 32;
    // how far the band(s) above are, as seen by one (device-coherent) look: the smaller of the two words
    auto look_up = [&]() -> int {
        const int s0 = (int32_t)(__hip_atomic_load(above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch);
        const int s1 = (int32_t)(__hip_atomic_load(above2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch);
        return s0 < s1 ? s0 : s1;
    };
    const uint32_t first = o[0], end = o[levels];
    if (first >= end) {         // a band without pixels: everything "done" (and an XCC id nobody shares: its neighbour writes through)
        if (threadIdx.x == 0) {
            __hip_atomic_store(mine, epoch + (uint32_t)levels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, epoch + 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int probe0 = PROF ? (int)(progress[(size_t)n_bands * 32 + 7999] & 0xffffu) - 1 : -1;     // profiling build: which two bands record a timeline
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    uint32_t my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xfu;
    if (threadIdx.x == 0) __hip_atomic_store(mine + 1, epoch + 1u + my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // A band of sky pixels only (the upper half of an outdoor frame) changes nothing: it says so at once, and the band below never
    // waits for it - otherwise its 12 levels of head start, 0.5 us each, stand in front of every band under it
    {
        int shaded = 0;
        for (int k = 0; k < rows; k++) {
            const int y = y0 + k * stride;
            if (y >= A.h) break;
            const uint8_t *bytes = sky + (size_t)y * A.w;               // (whole words where the row's bytes allow, bytes for the rest)
            const size_t head = (4 - ((uintptr_t)bytes & 3)) & 3, n_words = (size_t)A.w > head ? ((size_t)A.w - head) / 4 : 0;
            for (size_t i = threadIdx.x; i < n_words; i += 32 * G + 64) shaded |= ((const uint32_t *)(bytes + head))[i] != 0x01010101u;
            for (size_t i = threadIdx.x; i < (size_t)A.w; i += 32 * G + 64) if (i < head || i >= head + 4 * n_words) shaded |= bytes[i] != 1;
        }
        if (!__syncthreads_or(shaded)) {
            if (threadIdx.x == 0) __hip_atomic_store(mine, epoch + (uint32_t)levels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    if (threadIdx.x < 2 * G) (&sh.out_slot[0][0])[threadIdx.x] = YCGE_POST_NONE;
    if (!DUO && !publisher && grp >= groups) return;        // (a band whose passes use fewer groups than the workgroup has: the others leave before the first barrier)
    for (uint32_t e = threadIdx.x; e < n_ent; e += 32 * G + 64) sh.ent[e].x = YCGE_POST_NONE;
    if (publisher) {
        const int lane = (int)threadIdx.x - 32 * G;
        if (lane == 0) __hip_atomic_store(mine, epoch + pass_level[first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Does the band below run on this XCD?  Then its device-coherent loads find these colours in the L2 both share, and a plain
        // store (acknowledged by that L2 in 0.18 us, the line stays there) is enough; across XCDs the store must be written through
        // (0.38 us, and the reader's load goes to memory: ~1.5 us).  Measured, not assumed: every band announces its XCC id.
        // The answer is LOOKED for at every pass until it is there, never waited for: a band below that has not started yet (more
        // bands than the chip holds at once) starts only when a band above it has finished and left - until then colours are
        // written through.  (Waiting here deadlocked such launches until the queue was preempted: frames of 1 - 59 s.)
        bool same_xcd = false, dn_known = dn0 < 0;
        auto look_down = [&]() {
            bool all = true, same = true;
            for (int k = 0; k < 2; k++) {
                const int dn = k == 0 ? dn0 : dn1;
                if (dn < 0) continue;
                const uint32_t v = __hip_atomic_load(progress + (size_t)dn * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch;
                if (v == 0u || v > 16u) all = false; else same = same && v - 1u == my_xcc;
            }
            if (all) { dn_known = true; same_xcd = same; }
        };
        if (!dn_known) look_down();
        lds_barrier();          // table cleared
        // One pass of slack for the acknowledgement: the colours of pass i are sent after barrier i, and "pass i - 1 complete" is
        // published once all but that newest store instruction have been acknowledged (this wavefront issues nothing but stores,
        // and stores complete in order among themselves) - waiting for the newest one would make every pass as long as a
        // write-through round trip, because this wavefront stands at the workgroup's barrier like the others.
        uint32_t lv = pass_level[first], lv_done = lv;          // lv_done: what to publish once pass i - 1 is in memory
        bool have_prev = false;
        for (uint32_t i = first; i < end; i++) {
            const uint32_t lv_next = i + 1 < end ? pass_level[i + 1] : (uint32_t)levels;
            if (!dn_known) look_down();         // (its loads have returned before this pass's stores are issued: the count below sees stores only)
            lds_barrier();      // end of pass i: its colours are in the window, its entries in out_slot[i & 1]
            bool any = false;
            for (int l = lane; l < 4 * G; l += 64) {
                const uint32_t slot = sh.out_slot[(i - first) & 1u][l >> 2];
                if (slot != YCGE_POST_NONE && (l & 3) < 3) {
                    float *dst = buf + 3 * (size_t)sh.ent[slot].x + (l & 3);
                    const float v = __uint_as_float((&sh.ent[slot].y)[l & 3]);
                    if (same_xcd) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    any = true;
                }
            }
            // (G <= 16: the loop above is one iteration, at most one store instruction of each flavour - wait for all but the
            // newest; wider passes wait for everything)
            if (4 * G <= 64 && __any(any)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && have_prev) __hip_atomic_store(mine, epoch + lv_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have_prev = true;
            lv_done = lv_next;      // once pass i is in memory, the first level not completed is that of pass i + 1 (the same level if it has two passes)
            if (lv_next == lv) lv_done = lv;
            lv = lv_next;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(mine, epoch + (uint32_t)levels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {        // profiling aid (profiles/post_bands.py): when this band began and ended (100 MHz), how many passes it ran
            ((unsigned long long *)(mine + 4))[0] = t_begin; ((unsigned long long *)(mine + 4))[1] = __builtin_amdgcn_s_memrealtime(); mine[8] = end - first;
            uint32_t hw_id;         // where the band ran: cu_id [11:8], sh_id [12], se_id [15:13] (+ the XCC id in [31:28])
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            mine[9] = (hw_id & 0x0fffffffu) | my_xcc << 28;
        }
        return;
    }
    if (DUO) {
        // Every band of the split layout has 8 pixels a level: the 16 groups are TWO sets of 8 that take turns - while one set computes
        // pass i the other issues everything pass i + 1 reads (index arithmetic, the loads), and at the pass's barrier the roles swap.
        // A pass is one wavefront's instruction stream from the first look-up to the barrier (1.6 us with 4 wavefronts as with 8):
        // the ~100 instructions of the fetch now run on the SIMD's other wavefront instead of in front of the chain.
        const int set = grp >> 3, gl = grp & 7;
        auto entry = [&](uint32_t i) -> uint32_t { return i < end ? pixels[(size_t)i * G + gl] : YCGE_POST_NONE; };
        auto level_of = [&](uint32_t i) -> int { return i < end ? (int)pass_level[i] : levels; };
        auto sky_of = [&](uint32_t e) -> int { return e == YCGE_POST_NONE ? 1 : (int)sky[(size_t)(e & 0xffffu) + (size_t)(e >> 16) * (size_t)A.w]; };
        int up_seen = has_up ? 0 : 0x7fffffff;
        auto wait_above = [&](int need) {
            while (up_seen < need) { up_seen = look_up(); if (up_seen < need) __builtin_amdgcn_s_sleep(1); }
            asm volatile("" ::: "memory");
        };
        // this set computes the passes first + set, first + set + 2, ..: (pcur, scur, lcur) describe the next of them, (pnxt, lnxt) the one after
        uint32_t pcur = entry(first + set), pnxt = entry(first + set + 2);
        int lcur = level_of(first + set), lnxt = level_of(first + set + 2);
        int scur = set == 0 ? -1 : sky_of(pcur);
        PassData D;
        D.sky0 = 1;
        // list entry and level of the pass after next, sky flag of the next: asked for in the fetch phase (here for set 0's first
        // pass), not at the head of the computing one - a level's chain starts at the barrier
        uint32_t p_new = YCGE_POST_NONE;
        int l_new = levels, s_new = 1;
        uint32_t w0 = epoch, w1 = epoch;
        if (set == 0) {
            wait_above(lcur);
            if (threadIdx.x == 0) ((unsigned long long *)(mine + 18))[0] = __builtin_amdgcn_s_memrealtime();      // (profiles/post_bands.py: the band's first level may be fetched)
            D = pass_fetch<true>(A, W, buf, statw, sky, pcur, t, scur);
            p_new = entry(first + 4); l_new = level_of(first + 4); s_new = sky_of(pnxt);
            if (has_up) asm volatile("global_load_dword %0, %2, %3 sc1\n\tglobal_load_dword %1, %2, %4 sc1" : "=&v"(w0), "=&v"(w1) : "v"(0u), "s"(above), "s"(above2) : "memory");
        }
        lds_barrier();              // table cleared
        // profiling instantiation (profiles/post_bands.py): where set 0's shader clocks go - the head of a computing pass (its wait for
        // what it fetched), the pass itself up to its barrier, the fetch phase's wait for the band above, the fetch, its idle time at the barrier
        unsigned long long pf_head = 0, pf_comp = 0, pf_wait = 0, pf_fetch = 0, pf_bar = 0, pf_t0 = 0, pf_t1 = 0;
        for (uint32_t i = first; i < end; i++) {
            if (((i - first) & 1u) == (uint32_t)set) {
                if (PROF) { pf_t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pf_t1 = __builtin_amdgcn_s_memtime(); pf_head += pf_t1 - pf_t0; }
                // The words of the band(s) above, for the next fetch's decision: asked for here, looked at after the pass.  By hand:
                // written as look_up() the compiler put the subtraction and the minimum - and with them a wait for these two
                // device-coherent loads AND for the three list loads above - in FRONT of the pass, a round trip to memory in every
                // level's chain.  (The compiler's own waits stay right: loads return in order, these are the newest.)
                pass_compute<true>(A, pcur, D, sh, &sh.out_slot[set][gl]);       // ends with the workgroup's barrier
                if (PROF) pf_comp += __builtin_amdgcn_s_memtime() - pf_t1;
                if (has_up) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1) : : "memory");
                    const int s0 = (int32_t)(w0 - epoch), s1 = (int32_t)(w1 - epoch);
                    const int up_word = s0 < s1 ? s0 : s1;
                    if (up_word > up_seen) up_seen = up_word;
                }
                pcur = pnxt; scur = s_new; lcur = lnxt; pnxt = p_new; lnxt = l_new;
            } else {
                if (PROF) pf_t0 = __builtin_amdgcn_s_memtime();
                if (i + 1 < end) {
                    wait_above(lcur);
                    if (PROF) { pf_t1 = __builtin_amdgcn_s_memtime(); pf_wait += pf_t1 - pf_t0; }
                    D = pass_fetch<true>(A, W, buf, statw, sky, pcur, t, scur);
                    p_new = entry(i + 5); l_new = level_of(i + 5); s_new = sky_of(pnxt);
                    // (issued LAST in the phase: ahead of the wait for the band above - a whole phase earlier, so that the computing pass's
                    // first wait would not include this cross-XCD round trip - the word is a phase staler, the wait spins on fresh looks
                    // 1 250 instead of 740 shader clocks a pass and every configuration loses 10 %: measured in round 3, profiles/r03)
                    if (has_up) asm volatile("global_load_dword %0, %2, %3 sc1\n\tglobal_load_dword %1, %2, %4 sc1" : "=&v"(w0), "=&v"(w1) : "v"(0u), "s"(above), "s"(above2) : "memory");
                    if (PROF) { pf_t0 = __builtin_amdgcn_s_memtime(); pf_fetch += pf_t0 - pf_t1; }
                }
                lds_barrier();
                if (PROF) pf_bar += __builtin_amdgcn_s_memtime() - pf_t0;
            }
        }
        if (PROF && threadIdx.x == 0) {
            unsigned long long *rec = (unsigned long long *)(mine + 20);
            rec[0] = pf_head; rec[1] = pf_comp; rec[2] = pf_wait; rec[3] = pf_fetch; rec[4] = pf_bar;
        }
        return;
    }
    // pixel list entry and level of the passes i, i + 1, i + 2 travel in registers (fetched two passes ahead)
    int lvl1 = (int)pass_level[first];
    int lvl2 = first + 1 < end ? (int)pass_level[first + 1] : levels;
    int up_seen = has_up ? 0 : 0x7fffffff;                          // levels the band(s) above have completed, as far as this wavefront knows
    while (up_seen < lvl1) { up_seen = look_up(); if (up_seen < lvl1) __builtin_amdgcn_s_sleep(1); }
    asm volatile("" ::: "memory");
    // list entries run three passes ahead, the sky flag of an entry two: a pass of sky pixels (the upper half of an outdoor frame)
    // then fetches nothing and computes nothing - list entry, flag, barrier
    auto entry = [&](uint32_t i) -> uint32_t { return i < end ? pixels[(size_t)i * G + grp] : YCGE_POST_NONE; };
    auto sky_of = [&](uint32_t e) -> int { return e == YCGE_POST_NONE ? 1 : (int)sky[(size_t)(e & 0xffffu) + (size_t)(e >> 16) * (size_t)A.w]; };
    uint32_t p1 = entry(first), p2 = entry(first + 1), p3 = entry(first + 2);
    int s2 = sky_of(p2);
    PassData D1 = pass_fetch<true>(A, W, buf, statw, sky, p1, t);
    int up_word_old = 0;           // the word read in the pass before: a device-coherent load of a line its owner keeps rewriting takes longer than a pass
    lds_barrier();              // table cleared
    uint32_t i = first;
    // one pass: compute Dc (pass i) while Dn (pass i + 1) is fetched.  The loop below runs it twice per iteration with the two
    // register sets swapped - handing Dn over to Dc by assignment cost 50 register moves a pass
    auto one_pass = [&](const PassData &Dc, PassData &Dn) {
        const uint32_t p4 = entry(i + 3);
        const int s3 = sky_of(p3);
        const int lvl3 = i + 2 < end ? (int)pass_level[i + 2] : levels;
        // the word of the band above, for the NEXT pass's decision: asked for here by hand, looked at after the pass (as look_up() the
        // compiler waited for the two device-coherent loads on the spot - see the two-set form above)
        uint32_t w0 = epoch, w1 = epoch;
        if (has_up) asm volatile("global_load_dword %0, %2, %3 sc1\n\tglobal_load_dword %1, %2, %4 sc1" : "=&v"(w0), "=&v"(w1) : "v"(0u), "s"(above), "s"(above2) : "memory");
        uint32_t spins = 0;
        if (i + 1 < end) {
            while (up_seen < lvl2) {            // rare in the steady state: the band above is not far enough yet
                if (PROF) spins++;
                up_seen = look_up();
                if (up_seen < lvl2) __builtin_amdgcn_s_sleep(1);
            }
            if (PROF && threadIdx.x == 0 && lvl2 == YCGE_POST_PROBE_LEVEL) ((unsigned long long *)(mine + 12))[0] = __builtin_amdgcn_s_memrealtime();
            asm volatile("" ::: "memory");      // (the loads below are issued after the word was seen: program order; a fence would wait for everything in flight)
            Dn = pass_fetch<true>(A, W, buf, statw, sky, p2, t, s2);
        }
        if (PROF && threadIdx.x == 0 && lvl1 == YCGE_POST_PROBE_LEVEL) ((unsigned long long *)(mine + 14))[0] = __builtin_amdgcn_s_memrealtime();
        if (PROF && threadIdx.x == 0 && (b == probe0 || b == probe0 + 1) && i - first < 1000u) {      // timeline of two neighbouring bands
            uint32_t *tl = progress + (size_t)n_bands * 32 + (size_t)(b - probe0) * 4000 + (size_t)(i - first) * 4;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            tl[0] = (uint32_t)now; tl[1] = (uint32_t)(now >> 32); tl[2] = spins; tl[3] = (uint32_t)lvl1;
        }
        pass_compute<true>(A, p1, Dc, sh, &sh.out_slot[(i - first) & 1u][grp]);     // ends with the workgroup's barrier
        if (has_up) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1) : : "memory");
            const int s0 = (int32_t)(w0 - epoch), s1 = (int32_t)(w1 - epoch);
            const int up_word = s0 < s1 ? s0 : s1;
            if (up_word_old > up_seen) up_seen = up_word_old;
            up_word_old = up_word;
        }
        p1 = p2; p2 = p3; p3 = p4; s2 = s3; lvl1 = lvl2; lvl2 = lvl3;
        i++;
    };
    PassData D2 = D1;
    while (i < end) {
        one_pass(D1, D2);
        if (i >= end) break;
        one_pass(D2, D1);
    }
}

// ToneMapper.UpdateExposure, serial overload (ToneMapper.cs:49-91), part 1: the log term of every sampled pixel
// (0 for the ones the reference skips: s + 0.0f == s for every value the running sum can take) + how many count
__global__ __launch_bounds__(256) void k_exposure_terms(const float *__restrict__ hdr, const uint8_t *__restrict__ sky, int w, int h, int step,
                                                        int nsx, int nsy, float *__restrict__ terms, ToneState *__restrict__ state)
{
    // how many samples count is an integer sum (order-free): per workgroup through LDS into counts[block] = the words behind the
    // nsx * nsy terms, added up by k_exposure_sum - one global atomic per wavefront on a single word serialised the whole kernel
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool counted = false;
    if (i < nsx * nsy) {
        const int px = (i % nsx) * step, py = (i / nsx) * step;
        const size_t j = (size_t)px + (size_t)py * w;
        float term = 0.0f;
        const uint8_t is_sky = sky[j];
        const F3 c = ld3(hdr, j);
        if (!is_sky) {
            const float lum = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;
            if (lum > 0.0f) { term = m_log(1e-6f + lum); counted = true; }
        }
        terms[i] = term;
    }
    const unsigned long long m = __ballot(counted);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) ((uint32_t *)(terms + (size_t)nsx * nsy))[blockIdx.x] = s_cnt;
}

// part 2: logSum += term in scan order by ONE lane, then the exposure update
struct ToneConsts { float tone_exposure, ae_key, ae_speed, ae_min, ae_max; };
__global__ __launch_bounds__(64) void k_exposure_sum_serial(const float *__restrict__ terms, int n, ToneConsts K, ToneState *__restrict__ state)
{
    // The adds are one dependent chain (4 cycles each at best); everything else is kept off it: the wavefront
    // fetches the next 1024 terms with coalesced 16-byte loads while lane 0 adds the current 1024 out of LDS.
    __shared__ float4 s_buf[2][256];
    const int lane = threadIdx.x;
    uint32_t total_cnt = 0;     // samples that count: k_exposure_terms left one word per workgroup behind the terms
    {
        const uint32_t *counts = (const uint32_t *)(terms + n);
        const int n_blocks = (n + 255) / 256;
        for (int b = lane; b < n_blocks; b += 64) total_cnt += counts[b];
        for (int off = 32; off >= 1; off >>= 1) total_cnt += (uint32_t)__shfl_xor((int)total_cnt, off, 64);
    }
    const float4 *t4 = (const float4 *)terms;
    const int n_chunks = n / 1024;
    float log_sum = 0.0f;
    float4 r0, r1, r2, r3;
    if (n_chunks > 0) { r0 = t4[lane]; r1 = t4[64 + lane]; r2 = t4[128 + lane]; r3 = t4[192 + lane]; }
    for (int c = 0; c < n_chunks; c++) {
        float4 *buf = s_buf[c & 1];
        buf[lane] = r0; buf[64 + lane] = r1; buf[128 + lane] = r2; buf[192 + lane] = r3;
        if (c + 1 < n_chunks) {
            const float4 *nx = t4 + (size_t)(c + 1) * 256;
            r0 = nx[lane]; r1 = nx[64 + lane]; r2 = nx[128 + lane]; r3 = nx[192 + lane];
        }
        __syncthreads();
        if (lane == 0) {
            // The adds are ONE dependent chain; nothing else may sit on it.  32 terms in registers (set A) are added while the next 32
            // (set B) are already on their way out of LDS, and vice versa: reads issued by hand (ds_read_b128, no wait), consumed
            // behind an explicit s_waitcnt that leaves the OTHER set's eight reads outstanding.
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
            const uint32_t base = (uint32_t)(uintptr_t)buf;
#define YCGE_RD8(r0, r1, r2, r3, r4, r5, r6, r7, addr)                                                                                         \
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t" \
                         "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112" \
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr) : "memory")
#define YCGE_WAIT8(n, r0, r1, r2, r3, r4, r5, r6, r7)                                                                                          \
            asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7))
#define YCGE_ADD8(r0, r1, r2, r3, r4, r5, r6, r7)                                                                                              \
            log_sum += r0.x; log_sum += r0.y; log_sum += r0.z; log_sum += r0.w; log_sum += r1.x; log_sum += r1.y; log_sum += r1.z; log_sum += r1.w; \
            log_sum += r2.x; log_sum += r2.y; log_sum += r2.z; log_sum += r2.w; log_sum += r3.x; log_sum += r3.y; log_sum += r3.z; log_sum += r3.w; \
            log_sum += r4.x; log_sum += r4.y; log_sum += r4.z; log_sum += r4.w; log_sum += r5.x; log_sum += r5.y; log_sum += r5.z; log_sum += r5.w; \
            log_sum += r6.x; log_sum += r6.y; log_sum += r6.z; log_sum += r6.w; log_sum += r7.x; log_sum += r7.y; log_sum += r7.z; log_sum += r7.w
            YCGE_RD8(a0, a1, a2, a3, a4, a5, a6, a7, base);
#pragma unroll 1
            for (uint32_t blk = 0; blk < 30; blk += 2) {        // 32 blocks of 8 float4 = the chunk's 1024 terms; no branch inside:
                YCGE_RD8(b0, b1, b2, b3, b4, b5, b6, b7, base + (blk + 1u) * 128u);     // the two sets must stay in their registers
                YCGE_WAIT8(8, a0, a1, a2, a3, a4, a5, a6, a7);
                YCGE_ADD8(a0, a1, a2, a3, a4, a5, a6, a7);
                YCGE_RD8(a0, a1, a2, a3, a4, a5, a6, a7, base + (blk + 2u) * 128u);
                YCGE_WAIT8(8, b0, b1, b2, b3, b4, b5, b6, b7);
                YCGE_ADD8(b0, b1, b2, b3, b4, b5, b6, b7);
            }
            YCGE_RD8(b0, b1, b2, b3, b4, b5, b6, b7, base + 31u * 128u);
            YCGE_WAIT8(8, a0, a1, a2, a3, a4, a5, a6, a7);
            YCGE_ADD8(a0, a1, a2, a3, a4, a5, a6, a7);
            YCGE_WAIT8(0, b0, b1, b2, b3, b4, b5, b6, b7);
            YCGE_ADD8(b0, b1, b2, b3, b4, b5, b6, b7);
#undef YCGE_RD8
#undef YCGE_WAIT8
#undef YCGE_ADD8
        }
        // the other buffer is written next; it was last read two iterations ago, before the barrier above
    }
    if (lane != 0) return;
    for (int i = n_chunks * 1024; i < n; i++) log_sum += terms[i];
    const int cnt = (int)total_cnt;
    float ae = state->ae_exposure;
    const float avg_log = cnt > 0 ? log_sum / (float)(cnt > 1 ? cnt : 1) : 0.0f;
    const float avg_lum = m_exp(avg_log);
    float target = cnt > 0 ? K.ae_key / cs_max(1e-6f, avg_lum) : ae;
    if (target < K.ae_min) target = K.ae_min;
    if (target > K.ae_max) target = K.ae_max;
    const float s = 1.0f - m_exp(-K.ae_speed);
    ae = ae + (target - ae) * s;
    state->ae_exposure = ae;
    state->effective = K.tone_exposure * ae;
    state->count = 0;
}

// part 2, the default: the SAME serial fp32 sum, evaluated exactly without being serial.
//
// logSum += term is one chain of 518 400 dependent binary32 additions at 1080p (6.5 cycles each when hand-pipelined: 1.44 ms, the
// kernel above).  But while the running sum s stays inside ONE binade [2^e, 2^(e+1)) - which it does for tens of thousands of terms
// in a row once it has grown - every result is a multiple of u = 2^(e-23) and fl(s + t) is integer arithmetic on m = |s| / u:
//       m' = RNE(m + x),  x = sign(s) t / u   (exact in binary64: a binary32 value scaled by a power of two)
//          = m + floor(x) + inc,   inc = frac(x) > 1/2, or on a tie (frac(x) == 1/2) whatever makes m' even
// so the only thing a term needs to know about the sum before it is the PARITY of m, and only when it ties.  A chunk of terms is
// therefore a map (parity in) -> (total increment, parity out) that can be worked out for both parities without knowing s:
//   phase A  binary64 chunk sums and their prefix predict the binade each chunk starts in (all chunks in parallel);
//   phase B  every chunk simulates its terms for both input parities in units of its predicted u: increment D[p], parity out P[p]
//            and the lowest / highest running increment (all chunks in parallel, one lane each);
//   phase C  ONE lane walks the chunks: if the true s is in the predicted binade with the predicted sign and m + lowest / highest
//            stay inside [2^23, 2^24), the chunk is m += D[parity] - otherwise (the sum is still small, or it crosses a binade
//            boundary in this chunk: a few dozen chunks per frame) it adds that chunk's terms one by one in binary32.
// Same bits as the serial loop for every input (the fallback IS the serial loop); tested against the oracle's exposure on every
// post-stage parity test.  1.44 ms -> ~0.1 ms at 1080p.
#define YCGE_EXPO_CHUNK 512
#define YCGE_EXPO_BATCH 1024            // chunks whose records sit in LDS at a time during the walk
struct ExpoRec { long long d[2], lo[2], hi[2]; int32_t e, neg; };       // one chunk's map (parity in -> increment, bounds), 56 bytes

// phase A: binary64 sum of each chunk (one lane per chunk, spread over the chip; order inside a chunk is irrelevant for a prediction)
__global__ __launch_bounds__(64) void k_exposure_chunk_sums(const float *__restrict__ terms, int n, double *__restrict__ chunk_sum)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int n_chunks = (n + YCGE_EXPO_CHUNK - 1) / YCGE_EXPO_CHUNK;
    if (c >= n_chunks) return;
    const int lo = c * YCGE_EXPO_CHUNK, hi = lo + YCGE_EXPO_CHUNK < n ? lo + YCGE_EXPO_CHUNK : n;
    double a0 = 0.0, a1 = 0.0;
    const float4 *t4 = (const float4 *)(terms + lo);
    const int nq = (hi - lo) >> 2;
    int q4 = 0;
    for (; q4 + 2 <= nq; q4 += 2) {
        const float4 v = t4[q4], u = t4[q4 + 1];
        a0 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        a1 += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w);
    }
    for (int i = lo + 4 * q4; i < hi; i++) a0 += (double)terms[i];
    chunk_sum[c] = a0 + a1;
}
// exclusive prefix of the chunk sums (one workgroup; n_chunks is ~1000), in place: chunk_sum[c] <- what the sum is predicted to be
// when chunk c starts
__global__ __launch_bounds__(1024) void k_exposure_prefix(double *__restrict__ chunk_sum, int n_chunks)
{
    __shared__ double s[1024];
    __shared__ double s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0.0;
    for (int base = 0; base < n_chunks; base += 1024) {
        __syncthreads();
        const double own = base + tid < n_chunks ? chunk_sum[base + tid] : 0.0;
        s[tid] = own;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const double add = tid >= off ? s[tid - off] : 0.0;
            __syncthreads();
            s[tid] += add;
            __syncthreads();
        }
        const double carry = s_carry;
        if (base + tid < n_chunks) chunk_sum[base + tid] = carry + s[tid] - own;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s[1023];
    }
}
// phase B: every chunk simulates its terms for both input parities in units of its predicted ulp (one lane per chunk, spread over the chip)
__global__ __launch_bounds__(64) void k_exposure_chunk_maps(const float *__restrict__ terms, int n, const double *__restrict__ chunk_start,
                                                            ExpoRec *__restrict__ recs)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int n_chunks = (n + YCGE_EXPO_CHUNK - 1) / YCGE_EXPO_CHUNK;
    if (c >= n_chunks) return;
    const double s0 = chunk_start[c];
    const int neg = s0 < 0.0 ? 1 : 0;
    const double mag = s0 < 0.0 ? -s0 : s0;
    int e = -1000;
    if (mag >= 1.0e-30) e = (int)((double_to_bits(mag) >> 52) & 0x7ff) - 1023;      // 2^e <= |s0| < 2^(e+1)
    long long d0 = 0, d1 = 0, lo0 = 0, lo1 = 0, hi0 = 0, hi1 = 0;
    bool all_zero = true;           // a chunk of skipped samples (sky) leaves any sum as it is: s + 0.0f == s
    {
        const int lo = c * YCGE_EXPO_CHUNK, hi = lo + YCGE_EXPO_CHUNK < n ? lo + YCGE_EXPO_CHUNK : n;
        const float4 *t4 = (const float4 *)(terms + lo);
        const int nq = (hi - lo) >> 2;
        for (int q4 = 0; q4 < nq && all_zero; q4++) { const float4 v = t4[q4]; all_zero = v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f; }
        for (int i = lo + 4 * nq; i < hi && all_zero; i++) all_zero = terms[i] == 0.0f;
    }
    if (all_zero) e = -2000;
    else if (e >= -12 && e < 100) {      // below 2^-12 the sum is still tiny (first terms): serial
        const double inv_u = bits_to_double((uint64_t)(1023 + 23 - e) << 52) * (neg ? -1.0 : 1.0);      // sign(s) / u, a power of two
        const int lo = c * YCGE_EXPO_CHUNK, hi = lo + YCGE_EXPO_CHUNK < n ? lo + YCGE_EXPO_CHUNK : n;
        int p0 = 0, p1 = 1;
        auto one = [&](float term) {
            const double x = (double)term * inv_u;
            if (!(x > -1.0e12 && x < 1.0e12)) { lo0 = lo1 = -(1ll << 62); return; }      // absurd term (inf / nan): never take the fast path
            const double fx = floor(x);
            const double fr = x - fx;
            const long long ifx = (long long)fx;
            const int up = fr > 0.5 ? 1 : 0, tie = fr == 0.5 ? 1 : 0;
            const int inc0 = tie ? (int)((p0 + ifx) & 1) : up;
            const int inc1 = tie ? (int)((p1 + ifx) & 1) : up;
            d0 += ifx + inc0; d1 += ifx + inc1;
            p0 = (int)((p0 + ifx + inc0) & 1); p1 = (int)((p1 + ifx + inc1) & 1);
            lo0 = d0 < lo0 ? d0 : lo0; hi0 = d0 > hi0 ? d0 : hi0;
            lo1 = d1 < lo1 ? d1 : lo1; hi1 = d1 > hi1 ? d1 : hi1;
        };
        const float4 *t4 = (const float4 *)(terms + lo);
        const int nq = (hi - lo) >> 2;
        float4 nxt = nq > 0 ? t4[0] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int q4 = 0; q4 < nq; q4++) {           // the next four terms are on their way while these four are worked on
            const float4 v = nxt;
            if (q4 + 1 < nq) nxt = t4[q4 + 1];
            one(v.x); one(v.y); one(v.z); one(v.w);
        }
        for (int i = lo + 4 * nq; i < hi; i++) one(terms[i]);
    } else e = -1000;
    ExpoRec r;
    r.d[0] = d0; r.d[1] = d1; r.lo[0] = lo0; r.lo[1] = lo1; r.hi[0] = hi0; r.hi[1] = hi1; r.e = e; r.neg = neg;
    recs[c] = r;
}
// phase C: the walk (one workgroup) + the exposure update
__global__ __launch_bounds__(1024) void k_exposure_sum(const float *__restrict__ terms, int n, ToneConsts K, ToneState *__restrict__ state,
                                                       const ExpoRec *__restrict__ recs)
{
    const int n_chunks = (n + YCGE_EXPO_CHUNK - 1) / YCGE_EXPO_CHUNK;
    const int tid = threadIdx.x;
    // chunk records of the current batch, structure of arrays: the walking lane reads them at LDS latency, not memory latency
    __shared__ long long s_d[2][YCGE_EXPO_BATCH], s_lo[2][YCGE_EXPO_BATCH], s_hi[2][YCGE_EXPO_BATCH];
    __shared__ short s_e[YCGE_EXPO_BATCH];
    __shared__ unsigned char s_neg[YCGE_EXPO_BATCH];
    __shared__ long long s_gd[2][YCGE_EXPO_BATCH / 16], s_glo[2][YCGE_EXPO_BATCH / 16], s_ghi[2][YCGE_EXPO_BATCH / 16];
    __shared__ unsigned char s_gok[YCGE_EXPO_BATCH / 16];
    __shared__ float4 s_terms[YCGE_EXPO_CHUNK / 4];
    __shared__ uint32_t s_total_cnt;
    __shared__ int s_c, s_serial;
    __shared__ float s_sum;
    if (tid == 0) { s_total_cnt = 0; s_sum = 0.0f; s_serial = 0; }
    __syncthreads();
    {   // samples that count: k_exposure_terms left one word per workgroup behind the terms
        const uint32_t *counts = (const uint32_t *)(terms + n);
        const int n_blocks = (n + 255) / 256;
        uint32_t part = 0;
        for (int b = tid; b < n_blocks; b += 1024) part += counts[b];
        for (int off = 32; off >= 1; off >>= 1) part += (uint32_t)__shfl_xor((int)part, off, 64);
        if ((tid & 63) == 0 && part) atomicAdd(&s_total_cnt, part);
    }
    for (int batch = 0; batch < n_chunks; batch += YCGE_EXPO_BATCH) {
        const int batch_end = batch + YCGE_EXPO_BATCH < n_chunks ? batch + YCGE_EXPO_BATCH : n_chunks;
        __syncthreads();
        if (batch + tid < batch_end) {
            const ExpoRec r = recs[batch + tid];
            s_d[0][tid] = r.d[0]; s_d[1][tid] = r.d[1]; s_lo[0][tid] = r.lo[0]; s_lo[1][tid] = r.lo[1]; s_hi[0][tid] = r.hi[0]; s_hi[1][tid] = r.hi[1];
            s_e[tid] = (short)r.e; s_neg[tid] = (unsigned char)r.neg;
        }
        if (tid == 0) s_c = batch;
        __syncthreads();
        // groups of 16 chunks composed into one map each (valid when the 16 were simulated in the same binade with the same sign):
        // the walking lane then takes 64 steps per batch instead of 1024, and opens a group only when the group does not fit
        if (tid < YCGE_EXPO_BATCH / 16) {
            const int k0 = tid * 16;
            bool ok = batch + k0 + 16 <= batch_end;
            const int e0 = s_e[k0], n0 = s_neg[k0];
            long long gd[2] = {0, 0}, glo[2] = {0, 0}, ghi[2] = {0, 0};
            if (ok) {
                for (int pin = 0; pin < 2; pin++) {
                    long long D = 0, L = 0, H = 0;
                    int P = pin;
                    for (int j = 0; j < 16; j++) {
                        const int k = k0 + j;
                        if ((int)s_e[k] == -2000) continue;                 // an all-zero chunk changes nothing
                        if ((int)s_e[k] != e0 || (int)s_neg[k] != n0 || e0 <= -1000) ok = false;
                        const long long l = D + s_lo[P][k], h = D + s_hi[P][k];
                        L = l < L ? l : L; H = h > H ? h : H;
                        const long long dk = s_d[P][k];
                        D += dk;
                        P = (int)((P + dk) & 1);            // parity of m + d
                    }
                    gd[pin] = D; glo[pin] = L; ghi[pin] = H;
                }
            }
            s_gd[0][tid] = gd[0]; s_gd[1][tid] = gd[1]; s_glo[0][tid] = glo[0]; s_glo[1][tid] = glo[1]; s_ghi[0][tid] = ghi[0]; s_ghi[1][tid] = ghi[1];
            s_gok[tid] = ok ? 1 : 0;
        }
        __syncthreads();
        // s_sum is the reference's logSum after each chunk, bit for bit.  One lane walks the chunks that take the fast path; at a chunk
        // that does not, the whole workgroup stages its terms in LDS and the lane adds them one by one (the reference's loop)
        for (;;) {
            if (tid == 0) {
                float log_sum = s_sum;
                int cc = s_c;
                while (cc < batch_end) {
                    const int k = cc - batch;
                    const uint32_t bits = __float_as_uint(log_sum);
                    const int e_now = (int)((bits >> 23) & 0xff) - 127;
                    const int neg_now = (int)(bits >> 31);
                    const long long m = (long long)((bits & 0x7fffffu) | 0x800000u);       // |s| = m * 2^(e_now - 23) for a normal s
                    const int p = (int)(m & 1);
                    const bool normal = ((bits >> 23) & 0xff) != 0;
                    if ((int)s_e[k] == -2000) { cc++; continue; }           // nothing but + 0.0f
                    if ((k & 15) == 0 && s_gok[k >> 4] && normal && e_now == (int)s_e[k] && neg_now == (int)s_neg[k] &&
                        m + s_glo[p][k >> 4] > (1ll << 23) && m + s_ghi[p][k >> 4] < (1ll << 24)) {
                        const long long m2 = m + s_gd[p][k >> 4];         // sixteen chunks at once
                        log_sum = __uint_as_float((bits & 0xff800000u) | (uint32_t)(m2 & 0x7fffff));
                        cc += 16;
                        continue;
                    }
                    // strictly inside the binade: a sum that touches 2^e from above may have come from the finer grid below it
                    const bool fast = normal && e_now == (int)s_e[k] && neg_now == (int)s_neg[k] && m + s_lo[p][k] > (1ll << 23) && m + s_hi[p][k] < (1ll << 24);
                    if (!fast) break;
                    const long long m2 = m + s_d[p][k];
                    log_sum = __uint_as_float((bits & 0xff800000u) | (uint32_t)(m2 & 0x7fffff));
                    cc++;
                }
                s_sum = log_sum; s_c = cc;
            }
            __syncthreads();
            const int cc = s_c;
            if (cc >= batch_end) break;
            const int lo = cc * YCGE_EXPO_CHUNK, hi = lo + YCGE_EXPO_CHUNK < n ? lo + YCGE_EXPO_CHUNK : n;
            if (tid < YCGE_EXPO_CHUNK / 4) {
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // past the end: + 0.0f leaves the sum as it is
                const int i = lo + 4 * tid;
                if (i + 4 <= hi) v = ((const float4 *)(terms + lo))[tid];
                else { if (i < hi) v.x = terms[i]; if (i + 1 < hi) v.y = terms[i + 1]; if (i + 2 < hi) v.z = terms[i + 2]; }
                s_terms[tid] = v;
            }
            __syncthreads();
            if (tid == 0) {
                float log_sum = s_sum;
#pragma unroll 8
                for (int i = 0; i < YCGE_EXPO_CHUNK / 4; i++) { const float4 v = s_terms[i]; log_sum += v.x; log_sum += v.y; log_sum += v.z; log_sum += v.w; }
                s_sum = log_sum; s_c = cc + 1; s_serial = s_serial + 1;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid != 0) return;
    const float log_sum = s_sum;
    const int cnt = (int)s_total_cnt;
    float ae = state->ae_exposure;
    const float avg_log = cnt > 0 ? log_sum / (float)(cnt > 1 ? cnt : 1) : 0.0f;
    const float avg_lum = m_exp(avg_log);
    float target = cnt > 0 ? K.ae_key / cs_max(1e-6f, avg_lum) : ae;
    if (target < K.ae_min) target = K.ae_min;
    if (target > K.ae_max) target = K.ae_max;
    const float sp = 1.0f - m_exp(-K.ae_speed);
    ae = ae + (target - ae) * sp;
    state->ae_exposure = ae;
    state->effective = K.tone_exposure * ae;
    state->count = (uint32_t)s_serial;          // diagnostics: chunks that took the serial path this frame
}

// ToneMapper.ToneMapAndEncode + ApplySaturation, ToneMapper.cs:204-260
__device__ __forceinline__ float aces_film(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float num = x * (a * x + b);
    const float den = x * (c * x + d) + e;
    float y = den > 0.0f ? num / den : 0.0f;
    if (y < 0.0f) y = 0.0f;
    if (y > 1.0f) y = 1.0f;
    return y;
}
__device__ __forceinline__ F3 map_pixel(F3 hdr, float exposure, float gamma, float saturation, float vibrance)
{
    float r = cs_max(0.0f, hdr.x) * exposure;
    float g = cs_max(0.0f, hdr.y) * exposure;
    float b = cs_max(0.0f, hdr.z) * exposure;
    r = aces_film(r); g = aces_film(g); b = aces_film(b);
    const float inv_gamma = 1.0f / cs_max(0.1f, gamma);
    const float sr = m_pow(clamp01(r), inv_gamma);
    const float sg = m_pow(clamp01(g), inv_gamma);
    const float sb = m_pow(clamp01(b), inv_gamma);
    r = clamp01(sr); g = clamp01(sg); b = clamp01(sb);
    const float y = 0.2126f * r + 0.7152f * g + 0.0722f * b;
    const float maxc = cs_max(r, cs_max(g, b));
    const float minc = cs_min(r, cs_min(g, b));
    const float chroma = maxc - minc;
    const float vib = 1.0f + vibrance * (1.0f - chroma);
    const float f = saturation * vib;
    const float rr = y + (r - y) * f, gg = y + (g - y) * f, bb = y + (b - y) * f;
    return f3(clamp01(rr), clamp01(gg), clamp01(bb));
}

// step 8, :229-264: per chexel the ss x ss box average of its top and bottom half-cell, then MapPixel
__global__ __launch_bounds__(256) void k_tonemap_downsample(const float *__restrict__ hdr, int hiW, int fbW, int fbH, int ss, float gamma,
                                                            float saturation, float vibrance, const ToneState *__restrict__ state,
                                                            float *__restrict__ out /* fbW*fbH*6 */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= fbW * fbH) return;
    const int cx = i % fbW, cy = i / fbW;
    const int y_top0 = cy * 2 * ss, y_bot0 = (cy * 2 + 1) * ss, x0 = cx * ss;
    F3 top = f3(0, 0, 0), bot = f3(0, 0, 0);
    for (int sy = 0; sy < ss; sy++)
        for (int sx = 0; sx < ss; sx++) {
            top = top + ld3(hdr, (size_t)(x0 + sx) + (size_t)(y_top0 + sy) * hiW);
            bot = bot + ld3(hdr, (size_t)(x0 + sx) + (size_t)(y_bot0 + sy) * hiW);
        }
    const float inv = 1.0f / (float)(ss * ss);
    const float exposure = state->effective;
    const F3 t = map_pixel(f3(top.x * inv, top.y * inv, top.z * inv), exposure, gamma, saturation, vibrance);
    const F3 b = map_pixel(f3(bot.x * inv, bot.y * inv, bot.z * inv), exposure, gamma, saturation, vibrance);
    float *o = out + (size_t)i * 6;
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = b.x; o[4] = b.y; o[5] = b.z;
}

} // namespace ycge

template <int G>
static void launch_band(bool win, dim3 grid, hipStream_t stream, const ycge::AtrousParams &A, float *buf, const float *statw, const uint8_t *sky,
                        const uint32_t *d_pixels, const uint32_t *d_offsets, int n_levels, int K, int launch, int first, int rows_per_band, uint32_t wx)
{
    if (win) hipLaunchKernelGGL((ycge::k_atrous_band<G, true>), grid, dim3(32 * G), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, launch, first, rows_per_band, wx);
    else hipLaunchKernelGGL((ycge::k_atrous_band<G, false>), grid, dim3(32 * G), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, launch, first, rows_per_band, wx);
}

extern "C" {

size_t ycge_post_state_bytes(void) { return sizeof(ycge::ToneState); }

int ycge_launch_unit_normals(const float *normal, float *unit, size_t n, hipStream_t stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(ycge::k_unit_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, normal, unit, n);
    return (int)hipGetLastError();
}

int ycge_launch_atrous(int w, int h, int step, const float phi[4], const float *cur, float *dst, const float *albedo, const float *unit_n,
                       const float *depth, const uint8_t *sky, hipStream_t stream)
{
    ycge::AtrousParams A = {w, h, step, phi[0], phi[1], phi[2], phi[3]};
    hipLaunchKernelGGL(ycge::k_atrous, dim3((unsigned)((w + 31) / 32), (unsigned)((h + 7) / 8)), dim3(256), 0, stream, A, cur, dst, albedo, unit_n,
                       depth, sky);
    return (int)hipGetLastError();
}

// the colour-independent weight factors of an in-place iteration (k_atrous_static): needs the G-buffer only, so the host runs it on a
// side stream beside the iteration before it
int ycge_launch_atrous_static(int w, int h, int step, const float phi[4], const float *albedo, const float *unit_n, const float *depth,
                              const uint8_t *sky, float *statw, hipStream_t stream)
{
    ycge::AtrousParams A = {w, h, step, phi[0], phi[1], phi[2], phi[3]};
    const size_t n = (size_t)w * h;
    hipLaunchKernelGGL(ycge::k_atrous_static, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, stream, A, albedo, unit_n, depth, sky, statw, n);
    return (int)hipGetLastError();
}

// in-place iteration as a pipeline of bands (see k_atrous_band): n_bands + ceil(levels / K) - 1 launches.  window_width: the
// collision-free window the host found for this schedule and K (0: hash form)
int ycge_launch_atrous_inplace(int w, int h, int step, const float phi[4], float *buf, const float *albedo, const float *unit_n,
                               const float *depth, const uint8_t *sky, float *statw, const uint32_t *d_pixels, const uint32_t *d_offsets,
                               int n_levels, int n_bands, int K, int groups_per_pass, int rows_per_band, unsigned window_width, hipStream_t stream)
{
    ycge::AtrousParams A = {w, h, step, phi[0], phi[1], phi[2], phi[3]};
    const int groups = (n_levels + K - 1) / K;
    const bool win = window_width != 0 && (size_t)rows_per_band * window_width <= YCGE_POST_WIN && (window_width & (window_width - 1)) == 0;
    for (int launch = 0; launch < n_bands + groups - 1; launch++) {
        const int first = launch - (groups - 1) > 0 ? launch - (groups - 1) : 0;      // bands with a level group left to run
        const int last = launch < n_bands - 1 ? launch : n_bands - 1;
        const dim3 grid((unsigned)(last - first + 1));
        if (groups_per_pass == 8) launch_band<8>(win, grid, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, launch, first, rows_per_band, window_width);
        else if (groups_per_pass == 16) launch_band<16>(win, grid, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, launch, first, rows_per_band, window_width);
        else launch_band<32>(win, grid, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, launch, first, rows_per_band, window_width);
    }
    return (int)hipGetLastError();
}

// the same iteration as ONE persistent launch (k_atrous_persist).  progress: n_bands x 32 words, zero before the first use; epoch: a
// value that grows by more than the group count from call to call (the host's running sum)
// Unused dynamic LDS of the two-set instantiation: the only way to tell the dispatcher "at most N band workgroups on a CU"
static int g_duo_pad_lds = 0;
void ycge_atrous_duo_pad_lds(int bytes) { g_duo_pad_lds = bytes > 0 ? bytes : 0; }

int ycge_launch_atrous_persist(int w, int h, int step, const float phi[4], float *buf, const uint8_t *sky, float *statw, const uint32_t *d_pixels,
                               const uint32_t *d_offsets, const uint32_t *d_pass_level, const int32_t *d_band_desc, int n_levels, int n_bands, int K, int groups_per_pass, int rows_per_band, unsigned window_width,
                               uint32_t *progress, uint32_t epoch, int xcd_local, int level_handover, int profile, uint32_t ticket_base, hipStream_t stream)
{
    ycge::AtrousParams A = {w, h, step, phi[0], phi[1], phi[2], phi[3]};
    const int per_xcd = (n_bands + 7) / 8;
    const dim3 grid((unsigned)((xcd_local & 1) ? 8 * per_xcd : n_bands));
    if (level_handover) {       // 8 or 16 pixels a pass: the publishing wavefront is the workgroup's 5th or 9th
        if (groups_per_pass == 8) hipLaunchKernelGGL((ycge::k_atrous_stream<8, false>), grid, dim3(256 + 64), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, d_pass_level, d_band_desc, n_levels, n_bands, rows_per_band, window_width, progress, epoch, xcd_local, ticket_base);
        else if (groups_per_pass == 16 && d_band_desc && !profile) hipLaunchKernelGGL((ycge::k_atrous_stream<16, false, true>), grid, dim3(512 + 64), (size_t)g_duo_pad_lds, stream, A, buf, statw, sky, d_pixels, d_offsets, d_pass_level, d_band_desc, n_levels, n_bands, rows_per_band, window_width, progress, epoch, xcd_local, ticket_base);
        else if (groups_per_pass == 16 && d_band_desc) hipLaunchKernelGGL((ycge::k_atrous_stream<16, true, true>), grid, dim3(512 + 64), (size_t)g_duo_pad_lds, stream, A, buf, statw, sky, d_pixels, d_offsets, d_pass_level, d_band_desc, n_levels, n_bands, rows_per_band, window_width, progress, epoch, xcd_local, ticket_base);
        else if (groups_per_pass == 16 && !profile) hipLaunchKernelGGL((ycge::k_atrous_stream<16, false>), grid, dim3(512 + 64), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, d_pass_level, d_band_desc, n_levels, n_bands, rows_per_band, window_width, progress, epoch, xcd_local, ticket_base);
        else if (groups_per_pass == 16) hipLaunchKernelGGL((ycge::k_atrous_stream<16, true>), grid, dim3(512 + 64), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, d_pass_level, d_band_desc, n_levels, n_bands, rows_per_band, window_width, progress, epoch, xcd_local, ticket_base);
        else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
#if YCGE_EXPERIMENTS
    if (groups_per_pass == 8)
        hipLaunchKernelGGL((ycge::k_atrous_persist<8>), grid, dim3(256), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, n_bands, rows_per_band, window_width, progress, epoch, xcd_local);
    else if (groups_per_pass == 16)
        hipLaunchKernelGGL((ycge::k_atrous_persist<16>), grid, dim3(512), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, n_bands, rows_per_band, window_width, progress, epoch, xcd_local);
    else
        hipLaunchKernelGGL((ycge::k_atrous_persist<32>), grid, dim3(1024), 0, stream, A, buf, statw, sky, d_pixels, d_offsets, n_levels, K, n_bands, rows_per_band, window_width, progress, epoch, xcd_local);
    return (int)hipGetLastError();
#else
    return (int)hipErrorInvalidValue;          // (the group hand-over form, YCGE_POST_MODE=4, exists in experiment builds only: the host never asks for it here)
#endif
}

// How many band workgroups of the persistent launch one CU holds - asked of the runtime for the very instantiation
// ycge_launch_atrous_persist would start (they differ: 66 VGPRs and three workgroups for the two-set form, 82 and two for the
// others).  A band that is not resident while its neighbours spin on its progress word stalls the frame for seconds.
int ycge_atrous_persist_resident(int groups_per_pass, int split, int level_handover, int profile)
{
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    if (level_handover) {
        if (groups_per_pass == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_stream<8, false>, 256 + 64, 0);
        else if (groups_per_pass == 16 && split && !profile) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_stream<16, false, true>, 512 + 64, (size_t)g_duo_pad_lds);
        else if (groups_per_pass == 16 && split) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_stream<16, true, true>, 512 + 64, (size_t)g_duo_pad_lds);
        else if (groups_per_pass == 16 && !profile) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_stream<16, false>, 512 + 64, 0);
        else if (groups_per_pass == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_stream<16, true>, 512 + 64, 0);
    } else {
#if YCGE_EXPERIMENTS
        if (groups_per_pass == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_persist<8>, 256, 0);
        else if (groups_per_pass == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_persist<16>, 512, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ycge::k_atrous_persist<32>, 1024, 0);
#endif
    }
    return e == hipSuccess ? n : 0;
}

size_t ycge_exposure_scratch_bytes(int w, int h, int step)
{
    const int nsx = (w + step - 1) / step, nsy = (h + step - 1) / step;
    const size_t n_chunks = ((size_t)nsx * nsy + YCGE_EXPO_CHUNK - 1) / YCGE_EXPO_CHUNK;
    return ((n_chunks * sizeof(double) + 63) & ~(size_t)63) + n_chunks * sizeof(ycge::ExpoRec) + 64;
}

int ycge_launch_exposure(const float *hdr, const uint8_t *sky, int w, int h, int step, float *terms, void *state, const float consts[5],
                         void *scratch, int serial, hipStream_t stream)
{
    const int nsx = (w + step - 1) / step, nsy = (h + step - 1) / step;
    const int n = nsx * nsy;
    ycge::ToneConsts K = {consts[0], consts[1], consts[2], consts[3], consts[4]};
    hipLaunchKernelGGL(ycge::k_exposure_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, hdr, sky, w, h, step, nsx, nsy, terms,
                       (ycge::ToneState *)state);
    if (serial || !scratch) {       // YCGE_EXPOSURE_SERIAL: the one-lane chain (A/B and cross-check of the chunked evaluation)
        hipLaunchKernelGGL(ycge::k_exposure_sum_serial, dim3(1), dim3(64), 0, stream, terms, n, K, (ycge::ToneState *)state);
        return (int)hipGetLastError();
    }
    const size_t n_chunks = ((size_t)n + YCGE_EXPO_CHUNK - 1) / YCGE_EXPO_CHUNK;
    double *chunk_sum = (double *)scratch;
    ycge::ExpoRec *recs = (ycge::ExpoRec *)((char *)scratch + ((n_chunks * sizeof(double) + 63) & ~(size_t)63));
    const unsigned cb = (unsigned)((n_chunks + 63) / 64);
    hipLaunchKernelGGL(ycge::k_exposure_chunk_sums, dim3(cb), dim3(64), 0, stream, terms, n, chunk_sum);
    hipLaunchKernelGGL(ycge::k_exposure_prefix, dim3(1), dim3(1024), 0, stream, chunk_sum, (int)n_chunks);
    hipLaunchKernelGGL(ycge::k_exposure_chunk_maps, dim3(cb), dim3(64), 0, stream, terms, n, chunk_sum, recs);
    hipLaunchKernelGGL(ycge::k_exposure_sum, dim3(1), dim3(1024), 0, stream, terms, n, K, (ycge::ToneState *)state, recs);
    return (int)hipGetLastError();
}

int ycge_launch_tonemap(const float *hdr, int hiW, int fbW, int fbH, int ss, float gamma, float saturation, float vibrance, const void *state,
                        float *out, hipStream_t stream)
{
    const int n = fbW * fbH;
    hipLaunchKernelGGL(ycge::k_tonemap_downsample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, hdr, hiW, fbW, fbH, ss, gamma,
                       saturation, vibrance, (const ycge::ToneState *)state, out);
    return (int)hipGetLastError();
}

} // extern "C"
