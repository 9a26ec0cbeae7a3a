// ycge_post_host.cpp - host side of steps 6-8 of TryFlipAndBlit (RaytraceRenderer.cs:221-264): the level schedule of the in-place A-trous iteration,
// its band layouts, run_post (denoise, exact exposure sum, tonemap + downsample, the SDR read-back), and the host-only hooks the CPU tests hold them by.
// (Kernels: ycge_post.hip.)
#include "ycge_ctx.h"

namespace ycge_host {

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

int run_post(ycge_ctx *c, hipStream_t stream, float *out_sdr_host, bool timed, hipEvent_t history_read, hipEvent_t before_copy, bool second_sdr, hipEvent_t tone_wait, bool second_set)      // (defaults and what the events mean: ycge_ctx.h)
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

} // namespace ycge_host

extern "C" {

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
} // extern "C"
