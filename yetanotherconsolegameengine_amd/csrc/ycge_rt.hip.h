// ycge_rt.hip.h — device-side building blocks of the ray-trace core (gfx950, wave64):
// vectors, sampler, shading helpers, material evaluation, slab tests, analytic primitives,
// voxel DDA, the closest-hit traversal and hit-attribute reconstruction.  Shared by the
// wavefront stage kernels and the single-launch megakernel in ycge_kernels.hip.
//
// Arithmetic contract: see ycge_math.h (compiled with -ffp-contract=off; every expression is
// the reference's, operation for operation — citations are to /root/reference/ConsoleGame/).
#pragma once
#include <hip/hip_runtime.h>

#include "ycge_device.h"
#include "ycge_math.h"

namespace ycge {

// ------------------------------------------------------------------ vectors
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 f3(const float *p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 vdiv(F3 a, float s) { float inv = 1.0f / s; return f3(a.x * inv, a.y * inv, a.z * inv); }   // Vec3.cs:67-71
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ F3 normalized(F3 a)      // Vec3.cs:98-107
{
    float len_sq = a.x * a.x + a.y * a.y + a.z * a.z;
    if (len_sq <= 0.0f) return a;
    float inv_len = 1.0f / cs_sqrt(len_sq);
    return f3(a.x * inv_len, a.y * inv_len, a.z * inv_len);
}
__device__ __forceinline__ F3 saturate(F3 a) { return f3(clamp01(a.x), clamp01(a.y), clamp01(a.z)); }

struct RayQ {              // one closest-hit query: Scene.Hit(r, tMin, tMax)
    F3 o, d;
    float tmin, tmax;
    // A shadow query of a scene WITHOUT transparent materials only asks "is anything in [tmin, tmax]?": every hit has
    // Transparency <= 0, so ComputeTransmittanceToLight returns zero whichever occluder is the closest one
    // (RaytraceRenderer.cs:761-765, 773-781).  Such a query may stop at its first accepted hit: same radiance, fewer
    // steps.  The counting variants never do (their counters are the reference's full traversal).
    bool anyhit = false;
    // FULLWAVE callers (the single-launch kernels) enter traverse() with EVERY lane of the wavefront, so that the cooperative walk
    // (ycge_coop.hip.h) can use them all; a lane that has no query to trace carries live = false
    bool live = true;
};

struct Work {              // SURVEY 8(d) counters (COUNT variants) + this lane's traversal steps (always; scheduling feedback)
    unsigned rays, box, tri, prim, vox;
    unsigned steps;
    unsigned dark;             // COUNT variants: shadow queries towards lights of zero intensity - traced here as the reference does, skipped by the timed kernels
#if defined(YCGE_DBG_VOXSTAT)
    unsigned dbg[8];           // profiling build: scene-tree steps by kind (node, leaf, object), objects culled by their solid box, grids asked / entered, cell steps, cell fetches
#endif
};
#if defined(YCGE_DBG_VOXSTAT)
#define YCGE_VOXSTAT(w, i) ((w).dbg[i]++)
#else
#define YCGE_VOXSTAT(w, i) do { } while (0)
#endif

// wave-level iteration counters (profiling aid, only touched by COUNT variants)
static __shared__ unsigned int g_wave_iters[16];  // [wave][4 wave-level iteration counters] (profiling aid, COUNT variants)
__device__ __forceinline__ void prof_tick(int which)
{
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63;
    if ((m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&g_wave_iters[(threadIdx.x >> 6) * 4 + which], 1u);
}


// ------------------------------------------------------------------ sampler (RaytraceSampler.cs)
__device__ __constant__ uint8_t c_bayer8x8[64] = {
    0, 32, 8, 40, 2, 34, 10, 42, 48, 16, 56, 24, 50, 18, 58, 26, 12, 44, 4, 36, 14, 46, 6, 38, 60, 28, 52, 20, 62, 30, 54, 22,
    3, 35, 11, 43, 1, 33, 9, 41, 51, 19, 59, 27, 49, 17, 57, 25, 15, 47, 7, 39, 13, 45, 5, 37, 63, 31, 55, 23, 61, 29, 53, 21};

__device__ __forceinline__ float frac(float v) { return v - cs_floor(v); }
__device__ __forceinline__ float blue_noise_sample(int x, int y, int frame_idx, int channel)   // :27-34
{
    float base = ((float)c_bayer8x8[(y & 7) * 8 + (x & 7)] + 0.5f) * (1.0f / 64.0f);
    float rot = frac((float)(frame_idx + 1) * (channel == 0 ? 0.7548776662466927f : 0.5698402909980532f));
    return frac(base + rot);
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)                                     // :71-80
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t per_frame_seed(int x, int y, int64_t frame, uint64_t salt)  // :56-68, jx = jy = 0
{
    uint64_t h = 1469598103934665603ULL;
    h ^= (uint64_t)(int64_t)x * 0x9E3779B97F4A7C15ULL; h = splitmix64(h);
    h ^= (uint64_t)(int64_t)y * 0xC2B2AE3D27D4EB4FULL; h = splitmix64(h);
    h ^= (uint64_t)frame * 0x165667B19E3779F9ULL; h = splitmix64(h);
    h ^= 0ULL; h = splitmix64(h);
    h ^= salt; h = splitmix64(h);
    return h;
}
__device__ __forceinline__ float rng_next_unit(uint64_t &state)                                 // :43-52
{
    state = splitmix64(state);
    uint32_t m24 = (uint32_t)(state >> 40);
    return ((float)m24 + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ F3 cosine_sample_hemisphere(F3 n, uint64_t &rng)                     // :83-111
{
    float u1 = rng_next_unit(rng);
    float u2 = rng_next_unit(rng);
    float r = cs_sqrt(u1);
    float phi = 6.2831853071795864769f * u2;
    float sn, cs;
    m_sincos(phi, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = cs_sqrt(1.0f - u1);
    float wz = n.z;
    if (wz < -0.999999f) {
        F3 u = f3(0.0f, -1.0f, 0.0f);
        F3 v = f3(-1.0f, 0.0f, 0.0f);
        return u * x + v * y + n * z;
    }
    float a = 1.0f / (1.0f + wz);
    float b = (-n.x * n.y) * a;
    // new Vec3(double, double, double): `1.0 - (w.X*w.X)*a` is binary64, then narrowed
    F3 u_axis = f3((float)(1.0 - (double)((n.x * n.x) * a)), b, -n.x);
    F3 v_axis = f3(b, (float)(1.0 - (double)((n.y * n.y) * a)), -n.y);
    return u_axis * x + v_axis * y + n * z;
}

// ------------------------------------------------------------------ shading helpers (RaytraceRenderer.cs:737-831)
#define YCGE_PI 3.14159265358979323846f
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return v - n * (2.0f * dot(v, n)); }
__device__ __forceinline__ F3 lerp3(F3 a, F3 b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ bool refract(F3 v, F3 n, float eta, F3 &out)
{
    float cosi = -cs_max(-1.0f, cs_min(1.0f, dot(v, n)));
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) { out = f3(0, 0, 0); return false; }
    out = (v * eta) + (n * (eta * cosi - cs_sqrt(k)));
    return true;
}
__device__ __forceinline__ float fresnel_schlick(float cos_theta, float eta_i, float eta_t)
{
    float r0 = (eta_i - eta_t) / (eta_i + eta_t);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * m_pow5(1.0f - cos_theta);
}
// OrenNayarBRDF, RaytraceRenderer.cs:810-831.  A and B depend on sigma only; the host evaluates the
// same fp32 expressions once per context (FrameParams.on_a / on_b), which yields the same bits.
__device__ __forceinline__ F3 oren_nayar(F3 albedo, F3 n, F3 wo, F3 wi, float A, float B)
{
    const float inv_pi = 1.0f / YCGE_PI;
    float cos_i = cs_max(0.0f, dot(n, wi));
    float cos_o = cs_max(0.0f, dot(n, wo));
    if (cos_i <= 0.0f || cos_o <= 0.0f) return f3(0, 0, 0);
    float sin_i = cs_sqrt(cs_max(0.0f, 1.0f - cos_i * cos_i));
    float sin_o = cs_sqrt(cs_max(0.0f, 1.0f - cos_o * cos_o));
    F3 proj_i = normalized(wi - n * cos_i);
    F3 proj_o = normalized(wo - n * cos_o);
    float cos_phi = cs_max(0.0f, dot(proj_i, proj_o));
    float sin_alpha = cs_max(sin_i, sin_o);
    float tan_beta = cs_min(sin_i / cs_max(1e-6f, cos_i), sin_o / cs_max(1e-6f, cos_o));
    float on = (A + B * cos_phi * sin_alpha * tan_beta);
    F3 f = albedo * (on * inv_pi);
    return saturate(f);
}

// A light the timed kernels need no shadow query for (GLight::dark): zero intensity, finite colour, and the shaded point not ON the light
// (dist2 == 0 would make the attenuation 0 / 0).  The radiance it would add is a zero, and radiance (a sum of non-negative products that
// starts at +0) + (+-0) is radiance, bit for bit.
template <bool COUNT>
__device__ __forceinline__ bool light_is_dark(const GLight &L, float dist2, Work &w)
{
    const bool dark = L.dark != 0.0f && dist2 > 0.0f;
    if (COUNT && dark) w.dark++;           // the counting kernels trace the query (and say how many such there are)
    return !COUNT && dark;
}

// ------------------------------------------------------------------ materials
struct MatEval {
    F3 albedo, emission, trans_color;
    float reflectivity, transparency, ior;
};
__device__ __forceinline__ MatEval eval_material(const SceneDev &S, int mi, F3 pos)   // Scenes.cs:408-428
{
    const GMaterial *m = S.materials + mi;
    const float4 a = ((const float4 *)m)[0];   // kind, albedo
    const float4 b = ((const float4 *)m)[1];   // albedo_b, scale
    const float4 c = ((const float4 *)m)[2];   // refl, emission
    const float4 d = ((const float4 *)m)[3];   // transp, ior, trans_color.xy
    const float4 e = ((const float4 *)m)[4];   // trans_color.z
    MatEval o;
    if (__float_as_int(a.x) == 1) {
        int32_t cx = cs_f2i(cs_floor(pos.x / b.w));
        int32_t cz = cs_f2i(cs_floor(pos.z / b.w));
        bool check = (((uint32_t)cx + (uint32_t)cz) & 1u) == 0u;
        o.albedo = check ? f3(a.y, a.z, a.w) : f3(b.x, b.y, b.z);
    } else {
        o.albedo = f3(a.y, a.z, a.w);
    }
    o.reflectivity = c.x;
    o.emission = f3(c.y, c.z, c.w);
    o.transparency = d.x;
    o.ior = d.y;
    o.trans_color = f3(d.z, d.w, e.x);
    return o;
}
// Texture.SampleBilinear (Renderer/Texture.cs:142-163, static texture) + the blend of SampleAlbedo (RaytraceRenderer.cs:730-734):
// wrap by u - floor(u), texel coordinates over (size - 1), the right / lower neighbour wraps with %, RGBA32.toVec3 = byte / 255f,
// Lerp(a, b, t) = a * (1 - t) + b * t, Saturate; then albedo * (1 - t) + tex * t, Saturate.
__device__ __forceinline__ F3 sample_albedo(const SceneDev &S, F3 albedo, int tex_index, float tiles, float weight, float u, float v)
{
    const int4 info = ((const int4 *)S.tex_info)[tex_index];
    const int w = info.y, hgt = info.z;
    u = u * tiles; v = v * tiles;
    const float u_in = u, v_in = v;
    u = u - cs_floor(u);
    v = v - cs_floor(v);
    const float fx = u * (float)(w - 1), fy = v * (float)(hgt - 1);
    const int x0 = cs_f2i(cs_floor(fx)), y0 = cs_f2i(cs_floor(fy));
    F3 tex = f3(1.0f, 1.0f, 1.0f);
    if (info.w != 0) {
        // a LIVE texture, Texture.cs:113-140: uu = flipU ? 1 - u : u; Frac; neighbours clamped at the last column / row; bytes B, G, R
        // (LoadPixel, :173-182); two lerps per channel as a * (1 - t) + b * t; ONE Saturate at the end.  (u, v here are the tiled
        // coordinates SampleAlbedo passes, before this function's own wrap: restored from the arguments.)
        const int bpp = info.w & 15;
        const float uu = (info.w & 16) ? 1.0f - u_in : u_in, vv = (info.w & 32) ? 1.0f - v_in : v_in;
        const float dfx = frac(uu) * (float)(w - 1), dfy = frac(vv) * (float)(hgt - 1);
        const int dx0 = cs_f2i(cs_floor(dfx)), dy0 = cs_f2i(cs_floor(dfy));
        if (dx0 >= 0 && dx0 < w && dy0 >= 0 && dy0 < hgt) {
            const int dx1 = (dx0 + 1) >= w ? (w - 1) : (dx0 + 1), dy1 = (dy0 + 1) >= hgt ? (hgt - 1) : (dy0 + 1);
            const float dtx = dfx - (float)dx0, dty = dfy - (float)dy0;
            const uint8_t *bytes = (const uint8_t *)(S.tex_pixels + info.x);
            auto px3 = [&](int x, int y) { const uint8_t *q = bytes + ((size_t)y * w + x) * bpp; return f3((float)q[2] / 255.0f, (float)q[1] / 255.0f, (float)q[0] / 255.0f); };
            const F3 c00 = px3(dx0, dy0), c10 = px3(dx1, dy0), c01 = px3(dx0, dy1), c11 = px3(dx1, dy1);
            const float sx1 = 1.0f - dtx, sy1 = 1.0f - dty;
            const F3 r0 = f3(c00.x * sx1 + c10.x * dtx, c00.y * sx1 + c10.y * dtx, c00.z * sx1 + c10.z * dtx);
            const F3 r1 = f3(c01.x * sx1 + c11.x * dtx, c01.y * sx1 + c11.y * dtx, c01.z * sx1 + c11.z * dtx);
            tex = saturate(f3(r0.x * sy1 + r1.x * dty, r0.y * sy1 + r1.y * dty, r0.z * sy1 + r1.z * dty));
        }
    } else if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < hgt) {        // (not finite: an exception in the reference; white here, as in the oracle)
        const int x1 = (x0 + 1) % w, y1 = (y0 + 1) % hgt;
        const float tx = fx - (float)x0, ty = fy - (float)y0;
        const uint32_t *px = S.tex_pixels + info.x;
        const uint32_t c00 = px[(size_t)y0 * w + x0], c10 = px[(size_t)y0 * w + x1], c01 = px[(size_t)y1 * w + x0], c11 = px[(size_t)y1 * w + x1];
        auto texel = [](uint32_t c) { return f3((float)(c & 255u) / 255.0f, (float)((c >> 8) & 255u) / 255.0f, (float)((c >> 16) & 255u) / 255.0f); };
        auto lerp3 = [](F3 a, F3 b, float t) { const float s = 1.0f - t; return f3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t); };
        const F3 a = lerp3(texel(c00), texel(c10), tx), b = lerp3(texel(c01), texel(c11), tx);
        tex = saturate(lerp3(a, b, ty));
    }
    const float t = weight, s1 = 1.0f - t;
    return saturate(f3(albedo.x * s1 + tex.x * t, albedo.y * s1 + tex.y * t, albedo.z * s1 + tex.z * t));
}

// ------------------------------------------------------------------ box tests
// BVH.BoxHitFast, BVH.cs:201-236: NaN-propagating Max/Min, clamp to [tMin, tMax].
// MathF.Max / MathF.Min differ from the hardware's maxNum / minNum (v_max_f32 / v_min_f32) in ONE respect that a comparison can see:
// they propagate a NaN.  A slab product (plane - o) * inv is NaN only as 0 * inf, i.e. when a direction component is exactly zero
// (inv = +-inf) - boxes and origins are finite.  So a ray whose three reciprocals are finite takes the plain form: per axis min /
// max of the two products (the reference's swap), max3 / min3, clamp by max / min - results equal to the reference's up to the sign
// of a zero, which no later comparison sees (the value is only ever compared: entry order, the pop test against closest).  The
// rare ray with a zero component takes the reference's own sequence.  `inv` is constant over a query: the test is hoisted out of the
// walk.  (A scene-level node visit is two of these: config 5 makes 676 M of them a frame.)
__device__ __forceinline__ bool box_scene(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 inv,
                                          float tmin, float tmax, float &tnear, float &tfar)
{
    const bool plain = cs_abs(inv.x) < YCGE_INF && cs_abs(inv.y) < YCGE_INF && cs_abs(inv.z) < YCGE_INF;      // false for +-inf and NaN
    if (plain) {
        const float ax = (mnx - o.x) * inv.x, bx = (mxx - o.x) * inv.x;
        const float ay = (mny - o.y) * inv.y, by = (mxy - o.y) * inv.y;
        const float az = (mnz - o.z) * inv.z, bz = (mxz - o.z) * inv.z;
        float t_enter = __builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fmaxf(__builtin_fminf(ay, by), __builtin_fminf(az, bz)));
        float t_exit = __builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fminf(__builtin_fmaxf(ay, by), __builtin_fmaxf(az, bz)));
        t_enter = __builtin_fmaxf(t_enter, tmin);
        t_exit = __builtin_fminf(t_exit, tmax);
        tnear = t_enter; tfar = t_exit;
        return t_exit >= t_enter;
    }
    float en_x = (mnx - o.x) * inv.x, ex_x = (mxx - o.x) * inv.x;
    if (en_x > ex_x) { float t = en_x; en_x = ex_x; ex_x = t; }
    float en_y = (mny - o.y) * inv.y, ex_y = (mxy - o.y) * inv.y;
    if (en_y > ex_y) { float t = en_y; en_y = ex_y; ex_y = t; }
    float en_z = (mnz - o.z) * inv.z, ex_z = (mxz - o.z) * inv.z;
    if (en_z > ex_z) { float t = en_z; en_z = ex_z; ex_z = t; }
    float t_enter = cs_max(en_x, cs_max(en_y, en_z));
    float t_exit = cs_min(ex_x, cs_min(ex_y, ex_z));
    if (t_enter < tmin) t_enter = tmin;
    if (t_exit > tmax) t_exit = tmax;
    tnear = t_enter; tfar = t_exit;
    return t_exit >= t_enter;
}
__device__ __forceinline__ bool box_scene(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 inv,
                                          float tmin, float tmax, float &tnear)
{
    float tfar;
    return box_scene(mnx, mny, mnz, mxx, mxy, mxz, o, inv, tmin, tmax, tnear, tfar);
}
// MeshBVH.BoxHitFast, MeshBVH.cs:308-332: sign-indexed slabs.  The C# updates the interval with
// `if (tEnter > tMin) tMin = tEnter; if (tExit < tMax) tMax = tExit;` — a NaN candidate never wins and the
// running bound is never NaN, which is exactly IEEE maxNum / minNum (v_max_f32 / v_min_f32); the two
// returned values can differ from the C# only in the sign of a zero, which no later comparison sees.
// Its two early-outs are pure shortcuts: the final test fails whenever one of them would.
__device__ __forceinline__ bool box_mesh(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 inv,
                                         bool sx, bool sy, bool sz, float tmin, float tmax, float &tnear)
{
    float tx_en = ((sx ? mxx : mnx) - o.x) * inv.x;
    float tx_ex = ((sx ? mnx : mxx) - o.x) * inv.x;
    float ty_en = ((sy ? mxy : mny) - o.y) * inv.y;
    float ty_ex = ((sy ? mny : mxy) - o.y) * inv.y;
    float tz_en = ((sz ? mxz : mnz) - o.z) * inv.z;
    float tz_ex = ((sz ? mnz : mxz) - o.z) * inv.z;
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tmin, tx_en), ty_en), tz_en);
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fminf(tmax, tx_ex), ty_ex), tz_ex);
    tnear = tmin;
    return tmax >= tmin;
}

// ------------------------------------------------------------------ analytic primitives (t only; attributes are rebuilt in resolve_hit)
// XYRect/XZRect/YZRect.Hit, Surfaces.cs:184-214 / 256-286 / 328-358
__device__ __forceinline__ bool rect_t(int axis, float a0, float a1, float b0, float b1, float k, F3 o, F3 d, float tmin, float tmax, float &t)
{
    float dir_k = axis == 2 ? d.z : axis == 1 ? d.y : d.x;
    float org_k = axis == 2 ? o.z : axis == 1 ? o.y : o.x;
    float adir = cs_abs(dir_k);
    float safe = cs_copysign(cs_max(adir, 1e-8f), dir_k);
    t = (k - org_k) / safe;
    float pa, pb;
    if (axis == 2) { pa = o.x + t * d.x; pb = o.y + t * d.y; }
    else if (axis == 1) { pa = o.x + t * d.x; pb = o.z + t * d.z; }
    else { pa = o.y + t * d.y; pb = o.z + t * d.z; }
    bool ok = adir >= 1e-8f;
    ok &= (t >= tmin) & (t <= tmax);
    ok &= (pa >= a0) & (pa <= a1) & (pb >= b0) & (pb <= b1);
    return ok;
}
__device__ __forceinline__ void box_face(const float *p, int i, int &axis, float &a0, float &a1, float &b0, float &b1, float &k)
{   // Box ctor, BoundedObjects.cs:82-89: +Z, -Z, +Y, -Y, +X, -X
    const float mnx = p[0], mny = p[1], mnz = p[2], mxx = p[3], mxy = p[4], mxz = p[5];
    if (i < 2) { axis = 2; a0 = mnx; a1 = mxx; b0 = mny; b1 = mxy; k = i == 0 ? mxz : mnz; }
    else if (i < 4) { axis = 1; a0 = mnx; a1 = mxx; b0 = mnz; b1 = mxz; k = i == 2 ? mxy : mny; }
    else { axis = 0; a0 = mny; a1 = mxy; b0 = mnz; b1 = mxz; k = i == 4 ? mxx : mnx; }
}

template <bool COUNT>
__device__ __forceinline__ void analytic_prim(const float4 q0, const float4 q1, const float4 q2, const float4 q3, int type, int prim_index,
                                           F3 o, F3 d, float tmin, float &closest, int &hit_prim, int &hit_sub, Work &w)
{
    const float p[12] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    (void)q0;
    switch (type) {
    case 0: {   // Sphere.Hit, BoundedObjects.cs:31-69
        if (COUNT) w.prim++;
        float ox = o.x - p[0], oy = o.y - p[1], oz = o.z - p[2];
        float a = d.x * d.x + d.y * d.y + d.z * d.z;
        float half_b = ox * d.x + oy * d.y + oz * d.z;
        float c = ox * ox + oy * oy + oz * oz - p[3] * p[3];
        float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return;
        float s = cs_sqrt(disc);
        float inv_a = 1.0f / a;
        float t = (-half_b - s) * inv_a;
        if (t < tmin || t > closest) {
            t = (-half_b + s) * inv_a;
            if (t < tmin || t > closest) return;
        }
        closest = t; hit_prim = prim_index; hit_sub = 0;
        return;
    }
    case 1: {   // Plane.Hit, Surfaces.cs:39-71
        if (COUNT) w.prim++;
        float denom = p[0] * d.x + p[1] * d.y + p[2] * d.z;
        if (denom > -1e-6f && denom < 1e-6f) return;
        float t = (p[3] - (p[0] * o.x + p[1] * o.y + p[2] * o.z)) / denom;
        if (t < tmin || t > closest) return;
        closest = t; hit_prim = prim_index; hit_sub = 0;
        return;
    }
    case 2: {   // Disk.Hit, Surfaces.cs:108-142
        if (COUNT) w.prim++;
        F3 n = f3(p[3], p[4], p[5]);
        float denom = dot(n, d);
        float adenom = cs_abs(denom);
        float safe = cs_copysign(cs_max(adenom, 1e-8f), denom);
        float t = (p[7] - dot(n, o)) / safe;
        float px = o.x + t * d.x, pz = o.z + t * d.z;
        float dx = px - p[0], dz = pz - p[2];
        float rr = dx * dx + dz * dz;
        bool ok = adenom >= 1e-6f;
        ok &= (t >= tmin) & (t <= closest);
        ok &= rr <= p[6];
        if (!ok) return;
        closest = t; hit_prim = prim_index; hit_sub = 0;
        return;
    }
    case 3: case 4: case 5: {
        if (COUNT) w.prim++;
        int axis = type == 3 ? 2 : type == 4 ? 1 : 0;
        float t;
        if (!rect_t(axis, p[0], p[1], p[2], p[3], p[4], o, d, tmin, closest, t)) return;
        closest = t; hit_prim = prim_index; hit_sub = 0;
        return;
    }
    case 6: {   // Box.Hit, BoundedObjects.cs:100-115
        for (int i = 0; i < 6; i++) {
            if (COUNT) w.prim++;
            int axis; float a0, a1, b0, b1, k, t;
            box_face(p, i, axis, a0, a1, b0, b1, k);
            if (rect_t(axis, a0, a1, b0, b1, k, o, d, tmin, closest, t)) { closest = t; hit_prim = prim_index; hit_sub = i; }
        }
        return;
    }
    case 7: {   // CylinderY.Hit, BoundedObjects.cs:148-247; p = cx cz radius radius2 yMin yMax capped
        if (COUNT) w.prim++;
        float ox = o.x - p[0], oy = o.y, oz = o.z - p[1];
        float a = d.x * d.x + d.z * d.z;
        float hit_t = YCGE_FLT_MAX;
        int code = -1;      // 0 side, 1 top cap, 2 bottom cap
        if (a > 1e-12f) {
            float half_b = ox * d.x + oz * d.z;
            float c = ox * ox + oz * oz - p[3];
            float disc = half_b * half_b - a * c;
            if (disc >= 0.0f) {
                float s = cs_sqrt(disc);
                float inv_a = 1.0f / a;
                float t1 = (-half_b - s) * inv_a;
                if (t1 > tmin && t1 < closest) {
                    float y1 = oy + t1 * d.y;
                    if (y1 >= p[4] && y1 <= p[5]) { hit_t = t1; code = 0; }
                }
                if (code < 0) {
                    float t2 = (-half_b + s) * inv_a;
                    if (t2 > tmin && t2 < closest) {
                        float y2 = oy + t2 * d.y;
                        if (y2 >= p[4] && y2 <= p[5]) { hit_t = t2; code = 0; }
                    }
                }
            }
        }
        if (p[6] != 0.0f && cs_abs(d.y) > 1e-8f) {
            float t_top = (p[5] - oy) / d.y;
            if (t_top > tmin && t_top < closest) {
                float rx = ox + t_top * d.x, rz = oz + t_top * d.z;
                if (rx * rx + rz * rz <= p[3]) { if (t_top < hit_t) { hit_t = t_top; code = 1; } }
            }
            float t_bot = (p[4] - oy) / d.y;
            if (t_bot > tmin && t_bot < closest) {
                float rx = ox + t_bot * d.x, rz = oz + t_bot * d.z;
                if (rx * rx + rz * rz <= p[3]) { if (t_bot < hit_t) { hit_t = t_bot; code = 2; } }
            }
        }
        if (code < 0) return;
        closest = hit_t; hit_prim = prim_index; hit_sub = code;
        return;
    }
    case 8: {   // Triangle.Hit scalar path, Triangle.cs:131-175; p = A e1 e2 n
        if (COUNT) w.prim++;
        float e1x = p[3], e1y = p[4], e1z = p[5], e2x = p[6], e2y = p[7], e2z = p[8];
        float px = d.y * e2z - d.z * e2y;
        float py = d.z * e2x - d.x * e2z;
        float pz = d.x * e2y - d.y * e2x;
        float det = e1x * px + e1y * py + e1z * pz;
        if (cs_abs(det) < 1e-8f) return;
        float inv_det = 1.0f / det;
        float sx = o.x - p[0], sy = o.y - p[1], sz = o.z - p[2];
        float u = (sx * px + sy * py + sz * pz) * inv_det;
        if (u < 0.0f || u > 1.0f) return;
        float qx = sy * e1z - sz * e1y;
        float qy = sz * e1x - sx * e1z;
        float qz = sx * e1y - sy * e1x;
        float v = (d.x * qx + d.y * qy + d.z * qz) * inv_det;
        if (v < 0.0f || (u + v) > 1.0f) return;
        float t = (e2x * qx + e2y * qy + e2z * qz) * inv_det;
        if (t < tmin || t > closest) return;
        closest = t; hit_prim = prim_index; hit_sub = 0;
        return;
    }
    default: return;
    }
}

// ------------------------------------------------------------------ voxel grid
// The low three bits of x, y, z interleaved (x lowest).  All three are spread at once: the fields sit 9 bits apart in one
// word, one shift-or pair opens the gaps, one mask keeps bits 0, 3, 6 of each field (12 instructions, not 23 bit by bit).
__device__ __forceinline__ int morton3_3bits(int x, int y, int z)   // VolumeGrid.cs:246-252
{
    const uint32_t w = (uint32_t)(x & 7) | ((uint32_t)(y & 7) << 9) | ((uint32_t)(z & 7) << 18);
    const uint32_t t = (w | (w << 2) | (w << 4)) & 0x01249249u;
    return (int)((t | (t >> 8) | (t >> 16)) & 0x1ffu);
}
// cell coordinates and brick counts are far below 2^23: the one-instruction 24-bit multiply-add, not the 64-bit one
__device__ __forceinline__ int grid_brick(int ix, int iy, int iz, int nbx, int nby)
{
    return __mul24(__mul24(iz >> 3, nby) + (iy >> 3), nbx) + (ix >> 3);
}
__device__ __forceinline__ uint32_t grid_index(const GGrid &g, int ix, int iy, int iz)   // VolumeGrid.cs:235-242
{
    const int brick = grid_brick(ix, iy, iz, g.nbx, g.nby);
    return (uint32_t)(brick * 512 + morton3_3bits(ix, iy, iz));
}
__device__ __forceinline__ bool grid_slab(float ro, float rd, float inv, float mn, float mx, float &t_enter, float &t_exit, int axis, int &enter_axis)
{   // VolumeGrid.Slab, VolumeGrid.cs:331-355.  `inv` = 1.0f / rd, the C#'s own expression, evaluated once per ray
    if (cs_abs(rd) < 1e-12f) {
        if (ro < mn || ro > mx) return false;
        return true;
    }
    float t0 = (mn - ro) * inv;
    float t1 = (mx - ro) * inv;
    if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
    if (t0 > t_enter) { t_enter = t0; enter_axis = axis; }
    if (t1 < t_exit) t_exit = t1;
    return t_exit >= t_enter;
}
// x / s with the exact shortcut x / 1.0f == x (every voxel world of the reference has unit voxels)
__device__ __forceinline__ float div_by_size(float x, float s) { return s == 1.0f ? x : x / s; }

// true: between tmin and tmax the ray stays outside the box of the grid's solid voxels (GGrid::solid_lo / solid_hi, one voxel of
// margin): VolumeGrid.Hit would walk through air and return false, so the timed kernels skip the walk (a bounce ray that leaves
// the ground crosses the empty upper part of chunk after chunk).  Not in the counting kernels: their voxel-step and grid-entry
// counters are the reference's.  A NaN (0 * inf: ray parallel to a face, origin in its plane) is a point one voxel away from
// every solid voxel: either outcome of the comparison is right.
// t_out: where the ray leaves the box - beyond it the walk meets air only, so the timed kernels end it there (a ray that rises from
// the ground stops two voxels above the chunk's highest block instead of at the chunk's top).
__device__ __forceinline__ bool solid_box_missed(float lx, float ly, float lz, float hx, float hy, float hz, float t_limit, F3 o, F3 inv, float tmin, float tmax,
                                                 float &t_out)
{
    t_out = YCGE_INF;
    if (hx < lx) return true;                   // no solid voxel at all
    const float ax = (lx - o.x) * inv.x, bx = (hx - o.x) * inv.x;
    const float ay = (ly - o.y) * inv.y, by = (hy - o.y) * inv.y;
    const float az = (lz - o.z) * inv.z, bz = (hz - o.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    // so far from the ray's origin that the reference's cell walk (t by repeated binary32 additions) may stray more than the box's
    // one-voxel margin from the true ray (GGrid::cull_t_limit): no verdict, and no early end - the grid is walked as the reference walks it
    if (!(t_out <= t_limit)) { t_out = YCGE_INF; return false; }
    return t_in > t_out || t_out < tmin || t_in > tmax;
}
__device__ __forceinline__ bool grid_cull(const GGrid &g, F3 o, F3 inv, float tmin, float tmax, float &t_out)
{
    return solid_box_missed(g.solid_lo[0], g.solid_lo[1], g.solid_lo[2], g.solid_hi[0], g.solid_hi[1], g.solid_hi[2], g.cull_t_limit, o, inv, tmin, tmax, t_out);
}

// VolumeGrid.Hit, VolumeGrid.cs:99-231 (Amanatides-Woo DDA; first cell with matId > 0 hits).
// Every floating-point step of the C# is performed (the t of a hit is the running sum of its tDelta
// additions), but cell bytes are only FETCHED inside bricks the grid's 64-bit occupancy mask marks
// non-empty: in open air the walk is pure ALU instead of one dependent load per voxel.
template <bool COUNT>
__device__ __forceinline__ void grid_dda(const SceneDev &S, int grid_index_, int prim_index, F3 o, F3 d, F3 inv, float tmin, float &closest,
                                      int &hit_prim, int &hit_sub, Work &w)
{
    if (COUNT) { w.prim++; prof_tick(2); }
    YCGE_VOXSTAT(w, 4);
    const GGrid g = S.grids[grid_index_];
    if (!COUNT && prim_index < 0) prim_index = S.grid_owner[grid_index_];       // (reached through the walk tree, which names grids: SceneDev::walk_nodes)
    float t_solid_out = YCGE_INF;
    if (!COUNT && grid_cull(g, o, inv, tmin, closest, t_solid_out)) return;
    const float min_x = g.min_corner[0], min_y = g.min_corner[1], min_z = g.min_corner[2];
    const float size_x = g.voxel_size[0], size_y = g.voxel_size[1], size_z = g.voxel_size[2];
    const float max_x = min_x + (float)g.nx * size_x, max_y = min_y + (float)g.ny * size_y, max_z = min_z + (float)g.nz * size_z;
    const float tmax = COUNT ? closest : fminf(closest, t_solid_out);       // (only the loop's end test reads it)
    int enter_axis = -1;
    float t_enter = -YCGE_INF, t_exit = YCGE_INF;
    if (!grid_slab(o.x, d.x, inv.x, min_x, max_x, t_enter, t_exit, 0, enter_axis)) return;
    if (!grid_slab(o.y, d.y, inv.y, min_y, max_y, t_enter, t_exit, 1, enter_axis)) return;
    if (!grid_slab(o.z, d.z, inv.z, min_z, max_z, t_enter, t_exit, 2, enter_axis)) return;
    if (!(t_exit >= cs_max(0.0f, t_enter))) return;
    float t = t_enter; if (t < tmin) t = tmin; if (t > tmax || t > t_exit) return;
    t += 1e-6f;
    float px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    int ix = cs_f2i(cs_floor(div_by_size(px - min_x, size_x))); if (ix < 0) ix = 0; else if (ix >= g.nx) ix = g.nx - 1;
    int iy = cs_f2i(cs_floor(div_by_size(py - min_y, size_y))); if (iy < 0) iy = 0; else if (iy >= g.ny) iy = g.ny - 1;
    int iz = cs_f2i(cs_floor(div_by_size(pz - min_z, size_z))); if (iz < 0) iz = 0; else if (iz >= g.nz) iz = g.nz - 1;
    const int step_x = d.x > 0.0f ? 1 : d.x < 0.0f ? -1 : 0;
    const int step_y = d.y > 0.0f ? 1 : d.y < 0.0f ? -1 : 0;
    const int step_z = d.z > 0.0f ? 1 : d.z < 0.0f ? -1 : 0;
    const float inv_dx = step_x == 0 ? 0.0f : inv.x;
    const float inv_dy = step_y == 0 ? 0.0f : inv.y;
    const float inv_dz = step_z == 0 ? 0.0f : inv.z;
    const float next_vx = min_x + (step_x > 0 ? (float)(ix + 1) * size_x : (float)ix * size_x);
    const float next_vy = min_y + (step_y > 0 ? (float)(iy + 1) * size_y : (float)iy * size_y);
    const float next_vz = min_z + (step_z > 0 ? (float)(iz + 1) * size_z : (float)iz * size_z);
    float t_max_x = step_x == 0 ? YCGE_INF : (next_vx - o.x) * inv_dx;
    float t_max_y = step_y == 0 ? YCGE_INF : (next_vy - o.y) * inv_dy;
    float t_max_z = step_z == 0 ? YCGE_INF : (next_vz - o.z) * inv_dz;
    const float t_delta_x = step_x == 0 ? YCGE_INF : cs_abs(size_x * inv_dx);
    const float t_delta_y = step_y == 0 ? YCGE_INF : cs_abs(size_y * inv_dy);
    const float t_delta_z = step_z == 0 ? YCGE_INF : cs_abs(size_z * inv_dz);
    int last_axis = enter_axis < 0 ? (t_max_x <= t_max_y && t_max_x <= t_max_z ? 0 : t_max_y <= t_max_z ? 1 : 2) : enter_axis;
    const uint8_t *cells = S.grid_cells + g.cell_offset;
    const unsigned long long mask = ((unsigned long long)g.brick_mask_hi << 32) | g.brick_mask_lo;
    const bool use_mask = g.has_brick_mask != 0;
#if defined(YCGE_DBG_VOXSTAT)
    if (t <= t_exit && t <= tmax) w.dbg[5]++;
#endif
    while (t <= t_exit && t <= tmax) {
        if (COUNT) prof_tick(3);
        w.steps++;
        {   // (VolumeGrid.cs:153 tests the cell against the grid's bounds here: always true - the entry cell is clamped into the grid
            // and the walk leaves the loop the moment a step takes it outside, :224-227)
            if (COUNT) w.vox++;
            YCGE_VOXSTAT(w, 6);
            const int brick = grid_brick(ix, iy, iz, g.nbx, g.nby);
            if (!use_mask || ((mask >> brick) & 1ull)) {
                YCGE_VOXSTAT(w, 7);
                if (cells[(uint32_t)(brick * 512 + morton3_3bits(ix, iy, iz))] != 0) {
                    closest = cs_max(t, tmin);
                    hit_prim = prim_index;
                    hit_sub = (ix + g.nx * (iy + g.ny * iz)) | (last_axis << 30);
                    return;
                }
            }
        }
        // the three-way `if` of VolumeGrid.cs:203-223 as selects (one axis advances; the others keep their bits)
        const bool ax = t_max_x <= t_max_y && t_max_x <= t_max_z;
        const bool ay = !ax && t_max_y <= t_max_z;
        const bool az = !ax && !ay;
        t = ax ? t_max_x : ay ? t_max_y : t_max_z;
        ix += ax ? step_x : 0; iy += ay ? step_y : 0; iz += az ? step_z : 0;
        const float nx_ = t_max_x + t_delta_x, ny_ = t_max_y + t_delta_y, nz_ = t_max_z + t_delta_z;
        t_max_x = ax ? nx_ : t_max_x; t_max_y = ay ? ny_ : t_max_y; t_max_z = az ? nz_ : t_max_z;
        last_axis = ax ? 0 : ay ? 1 : 2;
        if ((uint32_t)ix >= (uint32_t)g.nx || (uint32_t)iy >= (uint32_t)g.ny || (uint32_t)iz >= (uint32_t)g.nz) break;
    }
}
// The same walk as grid_dda, split into "enter the grid" and "one cell step", so that a lane can pause inside a
// grid while the wavefront hands finished lanes new rays (k_wf_extend_p).  Operation for operation the code of grid_dda.
struct DdaState {
    int ix, iy, iz;
    float t, t_max_x, t_max_y, t_max_z, t_delta_x, t_delta_y, t_delta_z, t_exit, tmax;
    int step_x, step_y, step_z, last_axis, use_mask;        // (kept apart: the extend stage has registers to spare, a cell step no instructions)
    int nx, ny, nz, nbx, nby;
    uint32_t cell_offset, mask_lo, mask_hi;
    int prim;
};
template <bool COUNT>
__device__ __forceinline__ bool dda_begin(const SceneDev &S, int grid_index_, int prim_index, F3 o, F3 d, F3 inv, float tmin, float closest,
                                          DdaState &D, Work &w)
{
    if (COUNT) { w.prim++; prof_tick(2); }
    YCGE_VOXSTAT(w, 4);
    const GGrid g = S.grids[grid_index_];
    if (!COUNT && prim_index < 0) {
        // reached through the walk tree (SceneDev::walk_nodes), which names the grid and has tested the box of its solid voxels against
        // the `closest` of that moment: the object it belongs to, and where the ray leaves that box - the walk ends there (the object
        // step's own test, solid_box_missed, for the rays that come down the scene tree)
        prim_index = S.grid_owner[grid_index_];
        float t_solid_out;
        if (grid_cull(g, o, inv, tmin, closest, t_solid_out)) return false;
        closest = fminf(closest, t_solid_out);
    }
    const float min_x = g.min_corner[0], min_y = g.min_corner[1], min_z = g.min_corner[2];
    const float size_x = g.voxel_size[0], size_y = g.voxel_size[1], size_z = g.voxel_size[2];
    const float max_x = min_x + (float)g.nx * size_x, max_y = min_y + (float)g.ny * size_y, max_z = min_z + (float)g.nz * size_z;
    const float tmax = closest;
    int enter_axis = -1;
    float t_enter = -YCGE_INF, t_exit = YCGE_INF;
    if (!grid_slab(o.x, d.x, inv.x, min_x, max_x, t_enter, t_exit, 0, enter_axis)) return false;
    if (!grid_slab(o.y, d.y, inv.y, min_y, max_y, t_enter, t_exit, 1, enter_axis)) return false;
    if (!grid_slab(o.z, d.z, inv.z, min_z, max_z, t_enter, t_exit, 2, enter_axis)) return false;
    if (!(t_exit >= cs_max(0.0f, t_enter))) return false;
    float t = t_enter; if (t < tmin) t = tmin; if (t > tmax || t > t_exit) return false;
    t += 1e-6f;
    float px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    int ix = cs_f2i(cs_floor(div_by_size(px - min_x, size_x))); if (ix < 0) ix = 0; else if (ix >= g.nx) ix = g.nx - 1;
    int iy = cs_f2i(cs_floor(div_by_size(py - min_y, size_y))); if (iy < 0) iy = 0; else if (iy >= g.ny) iy = g.ny - 1;
    int iz = cs_f2i(cs_floor(div_by_size(pz - min_z, size_z))); if (iz < 0) iz = 0; else if (iz >= g.nz) iz = g.nz - 1;
    const int step_x = d.x > 0.0f ? 1 : d.x < 0.0f ? -1 : 0;
    const int step_y = d.y > 0.0f ? 1 : d.y < 0.0f ? -1 : 0;
    const int step_z = d.z > 0.0f ? 1 : d.z < 0.0f ? -1 : 0;
    const float inv_dx = step_x == 0 ? 0.0f : inv.x;
    const float inv_dy = step_y == 0 ? 0.0f : inv.y;
    const float inv_dz = step_z == 0 ? 0.0f : inv.z;
    const float next_vx = min_x + (step_x > 0 ? (float)(ix + 1) * size_x : (float)ix * size_x);
    const float next_vy = min_y + (step_y > 0 ? (float)(iy + 1) * size_y : (float)iy * size_y);
    const float next_vz = min_z + (step_z > 0 ? (float)(iz + 1) * size_z : (float)iz * size_z);
    D.t_max_x = step_x == 0 ? YCGE_INF : (next_vx - o.x) * inv_dx;
    D.t_max_y = step_y == 0 ? YCGE_INF : (next_vy - o.y) * inv_dy;
    D.t_max_z = step_z == 0 ? YCGE_INF : (next_vz - o.z) * inv_dz;
    D.t_delta_x = step_x == 0 ? YCGE_INF : cs_abs(size_x * inv_dx);
    D.t_delta_y = step_y == 0 ? YCGE_INF : cs_abs(size_y * inv_dy);
    D.t_delta_z = step_z == 0 ? YCGE_INF : cs_abs(size_z * inv_dz);
    const int last_axis = enter_axis < 0 ? (D.t_max_x <= D.t_max_y && D.t_max_x <= D.t_max_z ? 0 : D.t_max_y <= D.t_max_z ? 1 : 2) : enter_axis;
    D.ix = ix; D.iy = iy; D.iz = iz; D.t = t; D.t_exit = t_exit; D.tmax = tmax;
    D.step_x = step_x; D.step_y = step_y; D.step_z = step_z; D.last_axis = last_axis; D.use_mask = g.has_brick_mask ? 1 : 0;
    D.nx = g.nx; D.ny = g.ny; D.nz = g.nz; D.nbx = (g.nx + 7) >> 3; D.nby = (g.ny + 7) >> 3;
    D.cell_offset = g.cell_offset; D.mask_lo = g.brick_mask_lo; D.mask_hi = g.brick_mask_hi;
    D.prim = prim_index;
#if defined(YCGE_DBG_VOXSTAT)
    if (t <= t_exit && t <= tmax) w.dbg[5]++;
#endif
    return t <= t_exit && t <= tmax;            // the while condition of VolumeGrid.cs:151 before the first cell
}
// one pass of the while loop of VolumeGrid.Hit (:151-228).  Returns false when the lane has left the grid (hit or exit).
template <bool COUNT>
__device__ __forceinline__ bool dda_step(const SceneDev &S, DdaState &D, float tmin, float &closest, int &hit_prim, int &hit_sub, Work &w)
{
    if (COUNT) prof_tick(3);
    w.steps++;
    {   // (the cell is inside the grid: see grid_dda)
        if (COUNT) w.vox++;
        const int brick = grid_brick(D.ix, D.iy, D.iz, D.nbx, D.nby);
        const unsigned long long mask = ((unsigned long long)D.mask_hi << 32) | D.mask_lo;
        YCGE_VOXSTAT(w, 6);
        if (!D.use_mask || ((mask >> brick) & 1ull)) {
            YCGE_VOXSTAT(w, 7);
            if (S.grid_cells[D.cell_offset + (uint32_t)(brick * 512 + morton3_3bits(D.ix, D.iy, D.iz))] != 0) {
                closest = cs_max(D.t, tmin);
                hit_prim = D.prim;
                hit_sub = (D.ix + D.nx * (D.iy + D.ny * D.iz)) | (D.last_axis << 30);
                return false;
            }
        }
    }
    const bool ax = D.t_max_x <= D.t_max_y && D.t_max_x <= D.t_max_z;
    const bool ay = !ax && D.t_max_y <= D.t_max_z;
    const bool az = !ax && !ay;
    D.t = ax ? D.t_max_x : ay ? D.t_max_y : D.t_max_z;
    D.ix += ax ? D.step_x : 0; D.iy += ay ? D.step_y : 0; D.iz += az ? D.step_z : 0;
    const float nx_ = D.t_max_x + D.t_delta_x, ny_ = D.t_max_y + D.t_delta_y, nz_ = D.t_max_z + D.t_delta_z;
    D.t_max_x = ax ? nx_ : D.t_max_x; D.t_max_y = ay ? ny_ : D.t_max_y; D.t_max_z = az ? nz_ : D.t_max_z;
    D.last_axis = ax ? 0 : ay ? 1 : 2;
    if ((uint32_t)D.ix >= (uint32_t)D.nx || (uint32_t)D.iy >= (uint32_t)D.ny || (uint32_t)D.iz >= (uint32_t)D.nz) return false;
    return D.t <= D.t_exit && D.t <= D.tmax;
}
__device__ __forceinline__ double edge_distance(double v, double v0, double v1)   // VolumeGrid.cs:291-296
{
    double a = v - v0, b = v1 - v;
    if (a < 0.0) a = 0.0;
    if (b < 0.0) b = 0.0;
    return cs_min_d(a, b);
}
__device__ __forceinline__ bool is_wire_on_face(const GGrid &g, F3 p, int ix, int iy, int iz, int axis)   // VolumeGrid.cs:256-283 (fp64)
{
    double x0 = (double)(g.min_corner[0] + (float)ix * g.voxel_size[0]); double x1 = x0 + (double)g.voxel_size[0];
    double y0 = (double)(g.min_corner[1] + (float)iy * g.voxel_size[1]); double y1 = y0 + (double)g.voxel_size[1];
    double z0 = (double)(g.min_corner[2] + (float)iz * g.voxel_size[2]); double z1 = z0 + (double)g.voxel_size[2];
    if (axis == 0) {
        double dy = edge_distance((double)p.y, y0, y1), dz = edge_distance((double)p.z, z0, z1);
        double wv = (double)(g.wire_width_frac * cs_min(g.voxel_size[1], g.voxel_size[2]));
        return dy <= wv || dz <= wv;
    } else if (axis == 1) {
        double dx = edge_distance((double)p.x, x0, x1), dz = edge_distance((double)p.z, z0, z1);
        double wv = (double)(g.wire_width_frac * cs_min(g.voxel_size[0], g.voxel_size[2]));
        return dx <= wv || dz <= wv;
    }
    double dx = edge_distance((double)p.x, x0, x1), dy = edge_distance((double)p.y, y0, y1);
    double wv = (double)(g.wire_width_frac * cs_min(g.voxel_size[0], g.voxel_size[1]));
    return dx <= wv || dy <= wv;
}


// ------------------------------------------------------------------ per-lane traversal stacks
// Entries are {reference, tNear}.  The first YCGE_LDS_STACK levels of every lane live in LDS (entry e of
// thread t at g_lds_stack[e * 256 + t]: consecutive lanes hit consecutive 8-byte slots, so ds_read_b64 /
// ds_write_b64 are conflict-free whatever the per-lane depth); deeper levels — rare, a near-first DFS
// stacks one entry per level where both children are hit — go to a preallocated HBM spill area laid out
// [level][global lane] so that a wavefront's accesses coalesce.  No private arrays: scratch-backed
// kernels are admitted at about one wavefront per SIMD on this part (measured), which costs far more
// than the spill traffic.
#define YCGE_LDS_STACK YCGE_LDS_STACK_LEVELS
#define YCGE_BLOCK 256
static __shared__ uint2 g_lds_stack[YCGE_LDS_STACK * YCGE_BLOCK];     // 256-thread workgroups (tile = workgroup)
static __shared__ __attribute__((aligned(16))) uint2 g_lds_stack64[YCGE_LDS_STACK * 64];           // 64-thread workgroups (8x8 block = workgroup); while the stacks are empty: the work list of mesh_anyhit_bfs
static __shared__ uint2 g_lds_stack192[YCGE_LDS_STACK * 192];         // 192-thread workgroups (k_trace_fan: a block's three wavefronts)

// The LDS part is accessed with explicit ds_read_b64 / ds_write_b64: written as plain C++ the compiler merges the
// "LDS or spill" choice into ONE flat_load through a selected generic pointer (seen in the ISA), which puts the
// LDS pop on the slow flat path in the middle of the traversal's dependency chain.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_b64(uint32_t byte_addr, uint32_t x, uint32_t y)
{
    u32x2 v; v.x = x; v.y = y;
    asm volatile("ds_write_b64 %0, %1" : : "v"(byte_addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x2 lds_read_b64(uint32_t byte_addr)
{
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
    return v;
}
template <int BS>
struct StackT {
    static constexpr int kBS = BS;
    uint2 *spill;          // this lane's column of the spill area
    uint32_t spill_stride; // lanes in the grid
    uint32_t lds_base;     // LDS byte address of this lane's level-0 slot (levels are BS * 8 bytes apart)
    int sp;
    __device__ __forceinline__ void init(void *spill_base, uint32_t n_lanes, uint32_t first_lane = 0)
    {
        spill = (uint2 *)spill_base + (first_lane + blockIdx.x * BS + threadIdx.x);
        spill_stride = n_lanes;
        lds_base = (uint32_t)(uintptr_t)(BS == 64 ? (void *)g_lds_stack64 : BS == 192 ? (void *)g_lds_stack192 : (void *)g_lds_stack) + threadIdx.x * 8u;
        sp = 0;
    }
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push(uint32_t ref, float tnear)
    {
        if (sp < YCGE_LDS_STACK) lds_write_b64(lds_base + (uint32_t)sp * (BS * 8u), ref, __float_as_uint(tnear));
        else spill[(size_t)(sp - YCGE_LDS_STACK) * spill_stride] = make_uint2(ref, __float_as_uint(tnear));
        sp++;
    }
    __device__ __forceinline__ bool pop(uint32_t &ref, float &tnear)
    {
        if (sp == 0) return false;
        sp--;
        if (sp < YCGE_LDS_STACK) {
            const u32x2 v = lds_read_b64(lds_base + (uint32_t)sp * (BS * 8u));
            ref = v.x; tnear = __uint_as_float(v.y);
        } else {
            const uint2 v = spill[(size_t)(sp - YCGE_LDS_STACK) * spill_stride];
            ref = v.x; tnear = __uint_as_float(v.y);
        }
        return true;
    }
    // entry `lvl` of the stack of thread `owner` of this workgroup (the cooperative walk: a group of lanes works on the owner's stack)
    __device__ __forceinline__ void write_at(uint32_t owner, int lvl, uint32_t ref, float tnear) const
    {
        if (lvl < YCGE_LDS_STACK) lds_write_b64(lds_base - threadIdx.x * 8u + owner * 8u + (uint32_t)lvl * (BS * 8u), ref, __float_as_uint(tnear));
        else (spill - threadIdx.x + owner)[(size_t)(lvl - YCGE_LDS_STACK) * spill_stride] = make_uint2(ref, __float_as_uint(tnear));
    }
    // The same read asked for EARLY (LDS levels only; any other level asks for level 0 and the caller does not use the answer).  Plain C++
    // on purpose: the compiler tracks this load itself and puts the wait in front of the first USE, wherever the value has travelled by
    // then (a hand-issued asm load would be invisible to its wait-count bookkeeping, and a register copy at a loop edge would read stale data).
    __device__ __forceinline__ uint2 read_early(uint32_t owner, int lvl) const
    {
        const uint32_t l = lvl >= 0 && lvl < YCGE_LDS_STACK ? (uint32_t)lvl : 0u;
        const uint2 *base = BS == 64 ? g_lds_stack64 : BS == 192 ? g_lds_stack192 : g_lds_stack;
        return base[l * BS + owner];
    }
    __device__ __forceinline__ void read_at(uint32_t owner, int lvl, uint32_t &ref, float &tnear) const
    {
        if (lvl < YCGE_LDS_STACK) {
            const u32x2 v = lds_read_b64(lds_base - threadIdx.x * 8u + owner * 8u + (uint32_t)lvl * (BS * 8u));
            ref = v.x; tnear = __uint_as_float(v.y);
        } else {
            const uint2 v = (spill - threadIdx.x + owner)[(size_t)(lvl - YCGE_LDS_STACK) * spill_stride];
            ref = v.x; tnear = __uint_as_float(v.y);
        }
    }
};
using Stack = StackT<256>;

// MeshBVH.TriHit, MeshBVH.cs:239-304: scaled-numerator Moller-Trumbore, one divide on accept - for the TWO
// triangles of a GTriPair record at once.  Every product, sum and comparison is the one TriHit performs for that
// triangle, in its order (contraction is off), evaluated for both slots by packed operations; the accepts are
// then taken in leaf order, slot 1 against the `closest` slot 0 may just have lowered, which is what the
// reference's loop over the leaf does (MeshBVH.cs:171-186).  No early exits: a mixed wavefront runs every branch
// of them anyway (measured: skipping the two IEEE divisions when no lane accepts is slower than executing them).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct TriPairRec { f32x4 r0, r1, r2, r3; f32x2 e2z; };
__device__ __forceinline__ TriPairRec load_tri_pair(const GTriPair *tp)
{
    TriPairRec r;
    const f32x4 *q = (const f32x4 *)tp;
    r.r0 = q[0]; r.r1 = q[1]; r.r2 = q[2]; r.r3 = q[3];
    r.e2z = *(const f32x2 *)(q + 4);
    return r;
}
// `left` = triangles of the leaf still to test (>= 1; slot 1 exists when >= 2), `first` = index of slot 0
__device__ __forceinline__ void tri_pair_hit(const TriPairRec &T, uint32_t left, uint32_t first, int mesh_prim, F3 o, F3 d, float tmin,
                                             float &closest, int &hit_prim, int &hit_sub)
{
    const f32x2 ax = T.r0.xy, ay = T.r0.zw, az = T.r1.xy, e1x = T.r1.zw, e1y = T.r2.xy, e1z = T.r2.zw, e2x = T.r3.xy, e2y = T.r3.zw, e2z = T.e2z;
    const f32x2 dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const f32x2 px = dy * e2z - dz * e2y;
    const f32x2 py = dz * e2x - dx * e2z;
    const f32x2 pz = dx * e2y - dy * e2x;
    const f32x2 det = e1x * px + e1y * py + e1z * pz;
    const f32x2 sxx = ox - ax, syy = oy - ay, szz = oz - az;
    const f32x2 u_num = sxx * px + syy * py + szz * pz;
    const f32x2 sgn = {det.x > 0.0f ? 1.0f : -1.0f, det.y > 0.0f ? 1.0f : -1.0f};
    const f32x2 det_abs = det * sgn;
    const f32x2 u_num_s = u_num * sgn;
    const f32x2 qx = syy * e1z - szz * e1y;
    const f32x2 qy = szz * e1x - sxx * e1z;
    const f32x2 qz = sxx * e1y - syy * e1x;
    const f32x2 v_num = dx * qx + dy * qy + dz * qz;
    const f32x2 v_num_s = v_num * sgn;
    const f32x2 uv_sum_s = u_num_s + v_num_s;
    const f32x2 t_num = e2x * qx + e2y * qy + e2z * qz;
    const f32x2 t_num_s = t_num * sgn;
    const f32x2 tmin2 = {tmin, tmin};
    const f32x2 t_min_scaled = tmin2 * det_abs;
    bool ok0 = !(det.x > -1e-8f && det.x < 1e-8f);
    ok0 &= !(u_num_s.x < 0.0f || u_num_s.x > det_abs.x);
    ok0 &= !(v_num_s.x < 0.0f || uv_sum_s.x > det_abs.x);
    ok0 &= !(t_num_s.x < t_min_scaled.x || t_num_s.x > closest * det_abs.x);
    const float t_0 = t_num.x * (1.0f / det.x);
    if (ok0) { closest = t_0; hit_prim = mesh_prim; hit_sub = (int)first; }
    bool ok1 = left >= 2u;
    ok1 &= !(det.y > -1e-8f && det.y < 1e-8f);
    ok1 &= !(u_num_s.y < 0.0f || u_num_s.y > det_abs.y);
    ok1 &= !(v_num_s.y < 0.0f || uv_sum_s.y > det_abs.y);
    ok1 &= !(t_num_s.y < t_min_scaled.y || t_num_s.y > closest * det_abs.y);
    const float t_1 = t_num.y * (1.0f / det.y);
    if (ok1) { closest = t_1; hit_prim = mesh_prim; hit_sub = (int)(first + 1u); }
}
// The same test for ONE triangle in scalar form with TriHit's early exits (identical operations per triangle): the
// generic walk's leaves use it, where register count decides how many wavefronts a stage kernel of the wavefront path
// keeps resident (the packed form's pairs of intermediates cost every such kernel 15-30 VGPRs, i.e. a wavefront per
// SIMD, meshes in the scene or not; measured on the voxel world: 13.2 -> 14.4 ms).
__device__ __forceinline__ bool tri_hit(float ax, float ay, float az, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, F3 o, F3 d,
                                        float tmin, float tmax, float &t)
{
    float px = d.y * e2z - d.z * e2y;
    float py = d.z * e2x - d.x * e2z;
    float pz = d.x * e2y - d.y * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (det > -1e-8f && det < 1e-8f) return false;
    float sxx = o.x - ax, syy = o.y - ay, szz = o.z - az;
    float u_num = sxx * px + syy * py + szz * pz;
    float sgn = det > 0.0f ? 1.0f : -1.0f;
    float det_abs = det * sgn;
    float u_num_s = u_num * sgn;
    if (u_num_s < 0.0f || u_num_s > det_abs) return false;
    float qx = syy * e1z - szz * e1y;
    float qy = szz * e1x - sxx * e1z;
    float qz = sxx * e1y - syy * e1x;
    float v_num = d.x * qx + d.y * qy + d.z * qz;
    float v_num_s = v_num * sgn;
    float uv_sum_s = u_num_s + v_num_s;
    if (v_num_s < 0.0f || uv_sum_s > det_abs) return false;
    float t_num = e2x * qx + e2y * qy + e2z * qz;
    float t_num_s = t_num * sgn;
    float t_min_scaled = tmin * det_abs;
    float t_max_scaled = tmax * det_abs;
    if (t_num_s < t_min_scaled || t_num_s > t_max_scaled) return false;
    float inv_det = 1.0f / det;
    t = t_num * inv_det;
    return true;
}
// One leaf (<= 15 triangles, tested in leaf order against the shrinking `closest`) of the generic walk
template <bool COUNT>
__device__ __forceinline__ void leaf_triangles(const SceneDev &S, uint32_t pay, int mesh_prim, F3 o, F3 d, float tmin, float &closest,
                                               int &hit_prim, int &hit_sub, Work &w)
{
    const uint32_t unit = pay >> 4, count = pay & 15u;
    if (COUNT) w.tri += (int)count;
#pragma unroll 1
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t tri = ((unit + 3u * (i >> 1)) << 1) | (i & 1u);             // (record unit << 1) | slot
        const float *tp = (const float *)(S.mesh_arena + (size_t)(tri >> 1) * 32u) + (tri & 1u);       // component c of a slot is at tp[2 c]
        float t;
        if (tri_hit(tp[0], tp[2], tp[4], tp[6], tp[8], tp[10], tp[12], tp[14], tp[16], o, d, tmin, closest, t)) {
            closest = t;
            hit_prim = mesh_prim;
            hit_sub = (int)tri;
        }
    }
}

// ------------------------------------------------------------------ closest-hit traversal
// Scene.Hit -> BVH.Hit (BVH.cs:99-198) with Mesh -> MeshBVH.Hit (MeshBVH.cs:132-236) run on ONE stack.
//
// Visit order and pruning are exactly the reference's:
//  * children are tested against [tMin, closest]; both hit -> the nearer is visited first and the
//    other is stacked (ties: `lNear < rNear` false -> right first);
//  * the reference re-tests a node's own box when it pops it; with the entry distance tNear kept
//    next to the reference on the stack that re-test is `closest >= tNear` (same predicate, no
//    re-fetch: the raw slab values cannot change, only `closest` shrank);
//  * a scene leaf's objects are queued in order; a Mesh object opens its own tree on the same
//    stack and runs to completion before the next object of the leaf is tried.
// hit_sub of a mesh hit is the LEAF-ORDER triangle index (resolve_hit maps it back).
// Voxel grids are handled in PHASES: a lane that reaches a VolumeGrid object parks, the node loop keeps
// running until every lane of the wavefront is parked or finished, then all parked lanes run their DDA
// together.  Incoherent rays otherwise serialise: measured on config 5's bounce rays, running each DDA where
// it is met left 4.4 of 64 lanes active per VALU instruction.  Per-lane order of events is unchanged.
// The tree part of a query: runs until the lane reaches a VolumeGrid object (returns TREE_AT_GRID, the grid /
// object in parked_grid / parked_prim), has nothing left to visit (TREE_DONE) or has used its step budget (TREE_YIELD).
enum : int { TREE_DONE = 0, TREE_AT_GRID = 1, TREE_YIELD = 2 };
template <bool COUNT, bool HAS_GRID, class STK>
__device__ __forceinline__ int tree_phase(const SceneDev &S, uint32_t &cur, int &mesh_prim, STK &st, F3 o, F3 d, F3 inv, bool sx, bool sy, bool sz,
                                          float tmin, float &closest, int &hit_prim, int &hit_sub, int &parked_grid, int &parked_prim, float &parked_tend,
                                          Work &w, int budget = 0x7fffffff, bool anyhit = false)
{
    for (;;) {
        if (!COUNT && anyhit && hit_prim >= 0) { st.reset(); cur = YCGE_REF_NONE_VALUE; return TREE_DONE; }      // occlusion query answered
        if (budget-- <= 0) return TREE_YIELD;
        if (cur == YCGE_REF_NONE_VALUE) {
            float tn;
            if (!st.pop(cur, tn)) { cur = YCGE_REF_NONE_VALUE; return TREE_DONE; }
            if (!(closest >= tn)) { cur = YCGE_REF_NONE_VALUE; continue; }
        }
        const uint32_t kind = YCGE_REF_KIND(cur), pay = YCGE_REF_PAYLOAD(cur);
        if (COUNT) prof_tick(0);
        w.steps++;
        if (kind == REF_MESH_NODE || kind == REF_SCENE_NODE || (HAS_GRID && !COUNT && kind == REF_WALK_NODE)) {
            YCGE_VOXSTAT(w, 0);
            const float4 *np = kind == REF_MESH_NODE ? (const float4 *)(S.mesh_arena + (size_t)(pay >> 4) * 32u)
                             : (HAS_GRID && !COUNT && kind == REF_WALK_NODE) ? (const float4 *)(S.walk_nodes + (pay & ~YCGE_WALK_IN_ORDER)) : (const float4 *)(S.scene_nodes + pay);
            const float4 a = np[0], b = np[1], c = np[2], e = np[3];
            float ln, rn;
            bool hl, hr;
            if (COUNT) w.box += 2;
            if (kind == REF_MESH_NODE) {
                hl = box_mesh(a.x, a.y, a.z, b.x, b.y, a.w, o, inv, sx, sy, sz, tmin, closest, ln);      // GNode plane order
                hr = box_mesh(b.z, b.w, c.x, c.z, c.w, c.y, o, inv, sx, sy, sz, tmin, closest, rn);
            } else {
                hl = box_scene(a.x, a.y, a.z, b.x, b.y, a.w, o, inv, tmin, closest, ln);
                hr = box_scene(b.z, b.w, c.x, c.z, c.w, c.y, o, inv, tmin, closest, rn);
            }
            const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
            // (a leaf node of the walk tree: the leaf's objects in index order whatever the distances)
            const bool left_first = (HAS_GRID && !COUNT && kind == REF_WALK_NODE && (pay & YCGE_WALK_IN_ORDER)) || ln < rn;
            if (hl & hr) {
                if (left_first) { st.push(rref, rn); cur = lref; }
                else { st.push(lref, ln); cur = rref; }
            } else if (hl) cur = lref;
            else if (hr) cur = rref;
            else cur = YCGE_REF_NONE_VALUE;
        } else if (kind == REF_MESH_LEAF) {
            leaf_triangles<COUNT>(S, pay, mesh_prim, o, d, tmin, closest, hit_prim, hit_sub, w);
            cur = YCGE_REF_NONE_VALUE;
        } else if (kind == REF_SCENE_LEAF) {
            const uint32_t start = pay >> 3, count = pay & 7u;
            YCGE_VOXSTAT(w, 1);
            for (uint32_t i = count - 1; i >= 1; i--) st.push(YCGE_REF(REF_PRIM, S.scene_leaf_prims[start + i]), -YCGE_INF);
            cur = YCGE_REF(REF_PRIM, S.scene_leaf_prims[start]);
        } else if (HAS_GRID && !COUNT && kind == REF_GRID) {      // walk tree: the ray meets the box of this grid's solid voxels (the node above tested it)
            cur = YCGE_REF_NONE_VALUE;
            parked_grid = (int)pay; parked_prim = -1;       // (the cell walk finds the end of the solid box and the owning object itself: dda_begin, grid_dda)
            return TREE_AT_GRID;
        } else {    // REF_PRIM: objectHit[objId](r, tMin, closest, ...), BVH.cs:139-149
            const float4 *pp = (const float4 *)(S.prims + pay);
            const float4 q0 = pp[0];
            const int type = __float_as_int(q0.x);
            cur = YCGE_REF_NONE_VALUE;
            YCGE_VOXSTAT(w, 2);
            if (type == 10) {
                if (HAS_GRID) {
                    // (the object record carries the box of the grid's solid voxels: a grid the ray cannot hit costs this test, not the
                    // rest of the lane's tree steps of the round plus a voxel phase spent waiting - see grid_cull)
                    if (!COUNT) { const float4 q1 = pp[1], q2 = pp[2]; if (solid_box_missed(q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, o, inv, tmin, closest, parked_tend)) { YCGE_VOXSTAT(w, 3); continue; } }
                    parked_grid = __float_as_int(q0.z); parked_prim = (int)pay; return TREE_AT_GRID;
                }
            } else {
                const float4 q1 = pp[1], q2 = pp[2], q3 = pp[3];
                if (type == 9) {    // Mesh.Hit -> MeshBVH.Hit: root pushed, popped, own box tested (p = root box, ref)
                    const uint32_t root_ref = __float_as_uint(q2.z);
                    if (root_ref != YCGE_REF_NONE_VALUE) {
                        float tn;
                        if (COUNT) w.box++;
                        if (box_mesh(q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, o, inv, sx, sy, sz, tmin, closest, tn)) {
                            cur = root_ref;
                            mesh_prim = (int)pay;
                        }
                    }
                } else {
                    analytic_prim<COUNT>(q0, q1, q2, q3, type, (int)pay, o, d, tmin, closest, hit_prim, hit_sub, w);
                }
            }
        }
    }
}

// tree_phase for a lane in the walk tree of a voxel world (SceneDev::walk_nodes) and nothing else in the loop: a node visit (the plain
// form of the entry test - the caller sends only rays whose three reciprocals are finite, box_scene) or a grid reached.  TREE_OTHER: the
// lane stands at something else (an object that is no grid) with `cur` untouched - the caller goes on in tree_phase.  Operation for
// operation what tree_phase does for these references; the point is the instruction stream a wavefront issues per step under partial
// masks: tree_phase's loop body is 1 600 instructions of which a voxel world's lanes enter 200, and every block skipped costs its branch.
enum : int { TREE_OTHER = 3 };
#ifndef YCGE_WALK_PHASE
#define YCGE_WALK_PHASE 2          // 0: A/B build - the walk tree through tree_phase's general loop; 1: round 4's loop (a pop or a node visit per iteration); 2: below
#endif
#if YCGE_WALK_PHASE == 2
// Round 5's form of the loop.  What changed, and why it is the same walk:
//  * the stack entries a closer hit has made useless (`closest >= tNear` fails, BVH.cs:120-123) are discarded in a loop of their own - a
//    dozen instructions a pop - instead of one iteration of the whole body each: a ray that comes back from a grid with a hit finds
//    its three to five stacked siblings all farther than the hit, and round 4's loop spent a node step's issue slots on each;
//  * the twelve slab products of a node as six packed operations (v_pk_add_f32 / v_pk_mul_f32 - a - o is a + (-o) bit for bit), as mesh_walk
//    does for the mesh tree: WalkNode keeps its planes in the pairs this needs;
//  * no result code carried through the loop: why a lane left is read off (cur, sp) afterwards.
// (A lane may now report a grid, or the end of its walk, in the round its budget ran out in instead of the next: which round a step
// falls in never changes a result.)
template <class STK>
__device__ __forceinline__ int walk_phase(const SceneDev &S, uint32_t &cur, STK &st, F3 o, F3 inv, float tmin, float closest, int &parked_grid, int &parked_prim, Work &w, int budget)
{
    const f32x2 oxy = {o.x, o.y}, ozz = {o.z, o.z}, ixy = {inv.x, inv.y}, izz = {inv.z, inv.z};
    for (; budget > 0; budget--) {
        if (cur == YCGE_REF_NONE_VALUE) {
            for (;;) {
                uint32_t r; float tn;
                if (!st.pop(r, tn)) break;
                if (closest >= tn) { cur = r; break; }
            }
            if (cur == YCGE_REF_NONE_VALUE) break;          // nothing left
        }
        if (YCGE_REF_KIND(cur) != REF_WALK_NODE) break;     // a grid, or an object that is none
        const uint32_t pay = YCGE_REF_PAYLOAD(cur);
        w.steps++;
        YCGE_VOXSTAT(w, 0);
        const f32x4 *np = (const f32x4 *)(S.walk_nodes + (pay & ~YCGE_WALK_IN_ORDER));
        const f32x4 a = np[0], b = np[1], c = np[2], e = np[3];
        // box_scene, plain form, both children: left (a.x a.y a.z | b.x b.y a.w), right (b.z b.w c.x | c.z c.w c.y)
        const f32x2 la = (a.xy - oxy) * ixy, lb = (b.xy - oxy) * ixy, lz = (a.zw - ozz) * izz;
        const f32x2 ra = (b.zw - oxy) * ixy, rb = (c.zw - oxy) * ixy, rz = (c.xy - ozz) * izz;
        float ln = __builtin_fmaxf(__builtin_fminf(la.x, lb.x), __builtin_fmaxf(__builtin_fminf(la.y, lb.y), __builtin_fminf(lz.x, lz.y)));
        float lf = __builtin_fminf(__builtin_fmaxf(la.x, lb.x), __builtin_fminf(__builtin_fmaxf(la.y, lb.y), __builtin_fmaxf(lz.x, lz.y)));
        float rn = __builtin_fmaxf(__builtin_fminf(ra.x, rb.x), __builtin_fmaxf(__builtin_fminf(ra.y, rb.y), __builtin_fminf(rz.x, rz.y)));
        float rf = __builtin_fminf(__builtin_fmaxf(ra.x, rb.x), __builtin_fminf(__builtin_fmaxf(ra.y, rb.y), __builtin_fmaxf(rz.x, rz.y)));
        ln = __builtin_fmaxf(ln, tmin); lf = __builtin_fminf(lf, closest);
        rn = __builtin_fmaxf(rn, tmin); rf = __builtin_fminf(rf, closest);
        const bool hl = lf >= ln, hr = rf >= rn;
        const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
        const bool left_first = (pay & YCGE_WALK_IN_ORDER) || ln < rn;
        if (hl & hr) st.push(left_first ? rref : lref, left_first ? rn : ln);
        cur = (hl & hr) ? (left_first ? lref : rref) : hl ? lref : hr ? rref : YCGE_REF_NONE_VALUE;
    }
    if (cur == YCGE_REF_NONE_VALUE) return st.sp == 0 ? TREE_DONE : TREE_YIELD;
    const uint32_t kind = YCGE_REF_KIND(cur);
    if (kind == REF_WALK_NODE) return TREE_YIELD;
    if (kind != REF_GRID) return TREE_OTHER;
    w.steps++;
    parked_grid = (int)YCGE_REF_PAYLOAD(cur); parked_prim = -1;
    cur = YCGE_REF_NONE_VALUE;
    return TREE_AT_GRID;
}
#else
template <class STK>
__device__ __forceinline__ int walk_phase(const SceneDev &S, uint32_t &cur, STK &st, F3 o, F3 inv, float tmin, float closest, int &parked_grid, int &parked_prim, Work &w, int budget)
{
    for (;;) {
        if (budget-- <= 0) return TREE_YIELD;
        if (cur == YCGE_REF_NONE_VALUE) {
            float tn;
            if (!st.pop(cur, tn)) { cur = YCGE_REF_NONE_VALUE; return TREE_DONE; }
            if (!(closest >= tn)) { cur = YCGE_REF_NONE_VALUE; continue; }
        }
        const uint32_t kind = YCGE_REF_KIND(cur), pay = YCGE_REF_PAYLOAD(cur);
        if (kind == REF_GRID) {
            w.steps++;
            cur = YCGE_REF_NONE_VALUE;
            parked_grid = (int)pay; parked_prim = -1;
            return TREE_AT_GRID;
        }
        if (kind != REF_WALK_NODE) return TREE_OTHER;
        w.steps++;
        YCGE_VOXSTAT(w, 0);
        const float4 *np = (const float4 *)(S.walk_nodes + (pay & ~YCGE_WALK_IN_ORDER));
        const float4 a = np[0], b = np[1], c = np[2], e = np[3];
        // box_scene, plain form, both children
        const float lax = (a.x - o.x) * inv.x, lbx = (b.x - o.x) * inv.x, lay = (a.y - o.y) * inv.y, lby = (b.y - o.y) * inv.y, laz = (a.z - o.z) * inv.z, lbz = (a.w - o.z) * inv.z;
        const float rax = (b.z - o.x) * inv.x, rbx = (c.z - o.x) * inv.x, ray_ = (b.w - o.y) * inv.y, rby = (c.w - o.y) * inv.y, raz = (c.x - o.z) * inv.z, rbz = (c.y - o.z) * inv.z;
        float ln = __builtin_fmaxf(__builtin_fminf(lax, lbx), __builtin_fmaxf(__builtin_fminf(lay, lby), __builtin_fminf(laz, lbz)));
        float lf = __builtin_fminf(__builtin_fmaxf(lax, lbx), __builtin_fminf(__builtin_fmaxf(lay, lby), __builtin_fmaxf(laz, lbz)));
        float rn = __builtin_fmaxf(__builtin_fminf(rax, rbx), __builtin_fmaxf(__builtin_fminf(ray_, rby), __builtin_fminf(raz, rbz)));
        float rf = __builtin_fminf(__builtin_fmaxf(rax, rbx), __builtin_fminf(__builtin_fmaxf(ray_, rby), __builtin_fmaxf(raz, rbz)));
        ln = __builtin_fmaxf(ln, tmin); lf = __builtin_fminf(lf, closest);
        rn = __builtin_fmaxf(rn, tmin); rf = __builtin_fminf(rf, closest);
        const bool hl = lf >= ln, hr = rf >= rn;
        const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
        const bool left_first = (pay & YCGE_WALK_IN_ORDER) || ln < rn;
        if (hl & hr) st.push(left_first ? rref : lref, left_first ? rn : ln);          // (one push site: selects, not two copies of the spill test)
        cur = (hl & hr) ? (left_first ? lref : rref) : hl ? lref : hr ? rref : YCGE_REF_NONE_VALUE;
    }
}
#endif

// The scene tree of a scene that holds ANALYTIC objects only (SceneDev::analytic_only: no mesh, no voxel grid - configs 1 and 2 of the
// benchmark, BVH.cs:99-198 over the objects of Scenes/Scenes.cs:269-335), as two loops that each hold ONE kind of step: node and leaf
// steps until the lane stands at an object (or has nothing left), then - all lanes of the wavefront that stand at one together - the
// object's own test.  tree_phase's single loop carries the node step, the leaf step, nine primitive tests, the mesh entry and the grid entry
// through every iteration a wavefront makes (1 400 instructions of which a lane at a node enters 150), and a wavefront whose lanes stand at
// nodes, leaves and objects at once pays for all of it each time.  Per lane the sequence of visits, tests and updates of `closest` is
// tree_phase's, operation for operation: same hits, same counters, same step count.
template <bool COUNT, class STK>
__device__ __forceinline__ void analytic_walk(const SceneDev &S, uint32_t cur, STK &st, F3 o, F3 d, F3 inv, float tmin, float &closest, int &hit_prim,
                                              int &hit_sub, Work &w, bool anyhit)
{
    uint32_t at_prim = YCGE_REF_NONE_VALUE;          // payload of the object this lane stands at
    bool more = cur != YCGE_REF_NONE_VALUE;          // the stack is empty at entry
    for (;;) {
        while (more) {
            if (!COUNT && anyhit && hit_prim >= 0) { st.reset(); cur = YCGE_REF_NONE_VALUE; more = false; break; }      // occlusion query answered
            if (cur == YCGE_REF_NONE_VALUE) {
                float tn;
                if (!st.pop(cur, tn)) { cur = YCGE_REF_NONE_VALUE; more = false; break; }
                if (!(closest >= tn)) { cur = YCGE_REF_NONE_VALUE; continue; }
            }
            const uint32_t kind = YCGE_REF_KIND(cur), pay = YCGE_REF_PAYLOAD(cur);
            if (kind == REF_PRIM) { at_prim = pay; cur = YCGE_REF_NONE_VALUE; break; }
            if (COUNT) prof_tick(0);
            w.steps++;
            if (kind == REF_SCENE_NODE) {
                const float4 *np = (const float4 *)(S.scene_nodes + pay);
                const float4 a = np[0], b = np[1], c = np[2], e = np[3];
                float ln, rn;
                if (COUNT) w.box += 2;
                const bool hl = box_scene(a.x, a.y, a.z, b.x, b.y, a.w, o, inv, tmin, closest, ln);      // GNode plane order
                const bool hr = box_scene(b.z, b.w, c.x, c.z, c.w, c.y, o, inv, tmin, closest, rn);
                const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
                const bool left_first = ln < rn;
                if (hl & hr) st.push(left_first ? rref : lref, left_first ? rn : ln);
                cur = (hl & hr) ? (left_first ? lref : rref) : hl ? lref : hr ? rref : YCGE_REF_NONE_VALUE;
            } else {        // REF_SCENE_LEAF: its objects in index order (BVH.cs:139-149)
                const uint32_t start = pay >> 3, count = pay & 7u;
                for (uint32_t i = count - 1; i >= 1; i--) st.push(YCGE_REF(REF_PRIM, S.scene_leaf_prims[start + i]), -YCGE_INF);
                cur = YCGE_REF(REF_PRIM, S.scene_leaf_prims[start]);
            }
        }
        if (!__any(at_prim != YCGE_REF_NONE_VALUE)) break;
        if (at_prim != YCGE_REF_NONE_VALUE) {       // objectHit[objId](r, tMin, closest, ...), BVH.cs:139-149
            if (COUNT) prof_tick(0);
            w.steps++;
            const float4 *pp = (const float4 *)(S.prims + at_prim);
            const float4 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3];
            analytic_prim<COUNT>(q0, q1, q2, q3, __float_as_int(q0.x), (int)at_prim, o, d, tmin, closest, hit_prim, hit_sub, w);
            at_prim = YCGE_REF_NONE_VALUE;
        }
    }
}

// where a query enters the scene: the walk tree of a voxel world (SceneDev::walk_nodes) for a ray of the timed kernels that leaves the
// root box (at t_far) before the smallest distance any grid's cull verdict holds to, else the scene tree
template <bool COUNT, bool HAS_GRID>
__device__ __forceinline__ uint32_t scene_entry(const SceneDev &S, float t_far)
{
    return (HAS_GRID && !COUNT && S.walk_nodes && t_far <= S.walk_t_limit) ? S.walk_root_ref : S.scene_root_ref;
}

template <bool COUNT, bool HAS_GRID, class STK>
__device__ __forceinline__ void walk(const SceneDev &S, uint32_t cur, int mesh_prim, STK &st, F3 o, F3 d, F3 inv, bool sx, bool sy, bool sz,
                                     float tmin, float &closest, int &hit_prim, int &hit_sub, Work &w, bool anyhit = false)
{
    if (S.analytic_only) {       // (wave-uniform: a scalar branch)
        analytic_walk<COUNT>(S, cur, st, o, d, inv, tmin, closest, hit_prim, hit_sub, w, anyhit);
        return;
    }
    bool more = cur != YCGE_REF_NONE_VALUE;      // the stack is empty at entry
    // a ray that enters the walk tree of a voxel world with three finite reciprocals takes walk_phase - the loop that holds nothing else
    const bool fast = YCGE_WALK_PHASE && HAS_GRID && !COUNT && more && YCGE_REF_KIND(cur) == REF_WALK_NODE && cs_abs(inv.x) < YCGE_INF && cs_abs(inv.y) < YCGE_INF && cs_abs(inv.z) < YCGE_INF;
    for (;;) {
        int parked_grid = -1, parked_prim = -1;
        float parked_tend = YCGE_INF;       // (grid_dda finds the end of the solid box itself)
        bool parked = false;
        if (more) {
            int r = TREE_OTHER;
            if (HAS_GRID && !COUNT && fast) {
                if (anyhit && hit_prim >= 0) { st.reset(); cur = YCGE_REF_NONE_VALUE; r = TREE_DONE; }      // occlusion query answered (tree_phase's first line)
                else r = walk_phase(S, cur, st, o, inv, tmin, closest, parked_grid, parked_prim, w, 0x7fffffff);
            }
            if (r == TREE_OTHER)
                r = tree_phase<COUNT, HAS_GRID>(S, cur, mesh_prim, st, o, d, inv, sx, sy, sz, tmin, closest, hit_prim, hit_sub, parked_grid, parked_prim, parked_tend, w, 0x7fffffff, anyhit);
            parked = r == TREE_AT_GRID;
        }
        more = parked;
        if (!HAS_GRID) break;
        if (!__any(parked)) break;
        if (parked) grid_dda<COUNT>(S, parked_grid, parked_prim, o, d, inv, tmin, closest, hit_prim, hit_sub, w);
    }
}

// MeshBVH.Hit (MeshBVH.cs:132-236) for one mesh as a UNIFIED-STEP loop: in every iteration each live lane
// either visits one internal node (two slab tests, order, stack the far child) or tests one triangle of its
// current leaf, whichever it is at.  Both record kinds are fetched by the SAME four 16-byte loads from a
// per-lane address (a 48-byte triangle record is over-read by 16 bytes; the arrays are padded), issued
// back to back and waited for once: an iteration has ONE memory round trip.  (Left to the compiler the
// node's last 28 bytes were fetched after the node/triangle branch - a second dependent round trip per
// visit.)  The kernel's duration is the serial latency chain of its slowest wavefront, and this form makes a
// wavefront's iteration count ~ max over lanes of (nodes + triangles).  Visit order is unchanged.
// a node (64 B) or a triangle pair record (its first 72 B) in one round trip: every fetch is issued before the
// single wait.  The fifth fetch only has a use for leaf lanes; for node lanes it reads the start of the next
// node (the arrays are padded at upload).
__device__ __forceinline__ void load_record72(const uint8_t *base, uint32_t byte_offset, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &e, f32x2 &f)
{
    asm volatile("global_load_dwordx4 %0, %5, %6\n\t"
                 "global_load_dwordx4 %1, %5, %6 offset:16\n\t"
                 "global_load_dwordx4 %2, %5, %6 offset:32\n\t"
                 "global_load_dwordx4 %3, %5, %6 offset:48\n\t"
                 "global_load_dwordx2 %4, %5, %6 offset:64\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e), "=&v"(f)
                 : "v"(byte_offset), "s"(base)
                 : "memory");
}
} // namespace ycge
#include "ycge_coop.hip.h"
#include "ycge_anyhit.hip.h"
namespace ycge {

template <bool COUNT, bool BOUNDED = false, bool FULLWAVE = false, class STK>
__device__ __forceinline__ void mesh_walk(const SceneDev &S, uint32_t &cur, int mesh_prim, STK &st, F3 o, F3 inv, F3 d, bool sx, bool sy,
                                          bool sz, float tmin, float &closest, int &hit_prim, int &hit_sub, Work &w, int budget = 0x7fffffff,
                                          bool anyhit = false)
{
    // the twelve slab products of a node as six packed operations (v_pk_add_f32 / v_pk_mul_f32): GNode keeps its
    // planes as (x y)(z Z)(X Y) pairs per child, so two pairings of the ray's origin and reciprocal direction serve
    // all of them.  Each product is the same single subtract and multiply box_mesh performs; selecting by the
    // direction sign afterwards picks the same values.
    const f32x2 oxy = {o.x, o.y}, ozz = {o.z, o.z};
    const f32x2 ixy = {inv.x, inv.y}, izz = {inv.z, inv.z};
    // the timed kernels hand the last few walking lanes of a wavefront to the cooperative walk (ycge_coop.hip.h): same visits, same
    // answers, a third of the round trips.  The counting instances keep the lane-serial form (their counters are the reference's).
    // (only where the caller entered with every lane of the wavefront: the groups of the cooperative walk need them all)
    const bool cwalk = FULLWAVE && !COUNT && !BOUNDED && S.tl_offset != 0u;
#if defined(YCGE_DBG_BATCHSTAT)
    // profiling build: where a batch's iterations go - until 16 / 8 / 4 lanes are left, in the cooperative walk - by query kind
    uint32_t bs_it = 0u, bs_16 = 0xffffffffu, bs_8 = 0xffffffffu, bs_n0 = 0u, bs_kind = 3u, bs_coop = 0u;
    if (FULLWAVE && !COUNT && !BOUNDED) {
        const unsigned long long all = __ballot(cur != YCGE_REF_NONE_VALUE), any = __ballot(cur != YCGE_REF_NONE_VALUE && anyhit);
        bs_n0 = (uint32_t)__popcll(all);
        bs_kind = all == 0ull ? 3u : any == all ? 0u : any == 0ull ? 1u : 2u;
    }
#endif
    for (;;) {
        const bool act = cur != YCGE_REF_NONE_VALUE && (!BOUNDED || budget > 0);      // budget: refill mode yields with the walk's state in (cur, stack)
#if defined(YCGE_DBG_BATCHSTAT)
        if (FULLWAVE && !COUNT && !BOUNDED) {
            const uint32_t nn = (uint32_t)__popcll(__ballot(act));
            if (nn <= 16u && bs_16 == 0xffffffffu) bs_16 = bs_it;
            if (nn <= 8u && bs_8 == 0xffffffffu) bs_8 = bs_it;
            if (nn > YCGE_COOP_RAYS || !cwalk) bs_it++;
        }
#endif
        if (!act) break;
        if (cwalk && __popcll(__ballot(act)) <= YCGE_COOP_RAYS) break;
        if (BOUNDED) budget--;
        const bool is_node = YCGE_REF_KIND(cur) == REF_MESH_NODE;
        const uint32_t unit2 = (cur & 0x1ffffff0u) >> 3;       // record's 32-byte unit, times two
        f32x4 a, b, c, e;
        f32x2 f;
        load_record72(S.mesh_arena, unit2 << 4, a, b, c, e, f);
        if (COUNT) prof_tick(0);
        w.steps++;
        uint32_t next;
        if (is_node) {
            if (COUNT) w.box += 2;
            const f32x2 t0 = (a.xy - oxy) * ixy, t1 = (a.zw - ozz) * izz, t2 = (b.xy - oxy) * ixy;      // left:  (x y)(z Z)(X Y)
            const f32x2 t3 = (b.zw - oxy) * ixy, t4 = (c.xy - ozz) * izz, t5 = (c.zw - oxy) * ixy;      // right: (x y)(z Z)(X Y)
            float ln = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tmin, sx ? t2.x : t0.x), sy ? t2.y : t0.y), sz ? t1.y : t1.x);
            float lx = __builtin_fminf(__builtin_fminf(__builtin_fminf(closest, sx ? t0.x : t2.x), sy ? t0.y : t2.y), sz ? t1.x : t1.y);
            float rn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tmin, sx ? t5.x : t3.x), sy ? t5.y : t3.y), sz ? t4.y : t4.x);
            float rx = __builtin_fminf(__builtin_fminf(__builtin_fminf(closest, sx ? t3.x : t5.x), sy ? t3.y : t5.y), sz ? t4.x : t4.y);
            const bool hl = lx >= ln, hr = rx >= rn;
            const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
            const bool left_first = ln < rn;
            if (hl & hr) st.push(left_first ? rref : lref, left_first ? rn : ln);
            next = (hl & hr) ? (left_first ? lref : rref) : hl ? lref : hr ? rref : YCGE_REF_NONE_VALUE;
        } else {
            const uint32_t left = cur & 15u;
            if (COUNT) w.tri += left >= 2u ? 2 : 1;
            TriPairRec T;
            T.r0 = a; T.r1 = b; T.r2 = c; T.r3 = e; T.e2z = f;
            tri_pair_hit(T, left, unit2, mesh_prim, o, d, tmin, closest, hit_prim, hit_sub);
            next = left > 2u ? cur + ((3u << 4) - 2u) : YCGE_REF_NONE_VALUE;       // next record: 3 units on, two triangles fewer
            if (!COUNT && anyhit && hit_prim >= 0) { next = YCGE_REF_NONE_VALUE; st.reset(); }      // occlusion query answered: drop what is left
        }
        cur = next;
        if (cur == YCGE_REF_NONE_VALUE) {
            float tn; uint32_t r;
            while (st.pop(r, tn)) { if (closest >= tn) { cur = r; break; } }
        }
    }
    if (FULLWAVE && !COUNT && !BOUNDED) {
#if defined(YCGE_DBG_BATCHSTAT)
        const uint32_t bs_steps0 = w.steps;
#endif
        if (cwalk && __any(cur != YCGE_REF_NONE_VALUE)) coop_walk(S, cur, mesh_prim, st, o, inv, d, tmin, closest, hit_prim, hit_sub, w, anyhit);
#if defined(YCGE_DBG_BATCHSTAT)
        {
            uint32_t cm = w.steps - bs_steps0;          // the cooperative walk's iterations = the largest step count it handed back
            for (int off = 32; off >= 1; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)cm, off, 64); cm = o2 > cm ? o2 : cm; }
            bs_coop = cm;
            if ((threadIdx.x & 63u) == 0u && S.dbg_counters && bs_kind < 3u) {
                // bank 2 + kind: every batch; bank 5 + kind: batches of >= 48 iterations.  [0] batches [1] lanes at entry [2] serial iterations
                // [3] ... until <= 16 lanes [4] ... until <= 8 lanes [5] cooperative iterations [6] (serial + cooperative)^2 / 16 [7] batches that reached the cooperative walk
                const uint32_t tot = bs_it + bs_coop;
                for (int rep = 0; rep < (tot >= 48u ? 2 : 1); rep++) {
                    unsigned long long *dc = S.dbg_counters + 16 + (size_t)8 * 256 * (2 + bs_kind + 3 * rep) + (size_t)((blockIdx.x * 2654435761u) >> 24) * 8;
                    atomicAdd(dc + 0, 1ull); atomicAdd(dc + 1, (unsigned long long)bs_n0); atomicAdd(dc + 2, (unsigned long long)bs_it);
                    atomicAdd(dc + 3, (unsigned long long)(bs_16 < bs_it ? bs_16 : bs_it)); atomicAdd(dc + 4, (unsigned long long)(bs_8 < bs_it ? bs_8 : bs_it));
                    atomicAdd(dc + 5, (unsigned long long)bs_coop); atomicAdd(dc + 6, (unsigned long long)tot * tot / 16ull); atomicAdd(dc + 7, bs_coop ? 1ull : 0ull);
                }
            }
        }
#endif
    }
}

// One closest-hit query.  FLAT (the whole scene is one BVH leaf, <= 4 objects: every mesh-viewer scene of
// the reference): the object list is walked in leaf order with WAVE-UNIFORM control flow, so object records
// come through the scalar cache and only the per-lane mesh walk diverges.  Otherwise the generic walk
// starts at the scene root.  Both give the reference's visit order.
template <bool COUNT, bool HAS_GRID, bool FLAT, bool FULLWAVE = false, class STK>
__device__ __forceinline__ void traverse(const SceneDev &S, RayQ &q, STK &st, float &closest, int &hit_prim, int &hit_sub, Work &w)
{
    F3 o = q.o, d = q.d;
    float tmin = q.tmin;
    closest = q.tmax;
    hit_prim = -1;
    hit_sub = 0;
    st.reset();
    const bool live = !FULLWAVE || q.live;
    if (COUNT && live) w.rays++;
    if (S.scene_root_ref == YCGE_REF_NONE_VALUE) return;
    F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    float tn, tf;
    if (COUNT && live) w.box++;
    const bool root_hit = box_scene(S.scene_root_min[0], S.scene_root_min[1], S.scene_root_min[2], S.scene_root_max[0], S.scene_root_max[1],
                                    S.scene_root_max[2], o, inv, tmin, closest, tn, tf) && live;
    const bool anyhit = !COUNT && q.anyhit;
    if (!FLAT) {
        walk<COUNT, HAS_GRID>(S, root_hit ? scene_entry<COUNT, HAS_GRID>(S, tf) : YCGE_REF_NONE_VALUE, -1, st, o, d, inv, sx, sy, sz, tmin, closest, hit_prim, hit_sub, w, anyhit);
        return;
    }
    const uint32_t leaf_start = YCGE_REF_PAYLOAD(S.scene_root_ref) >> 3;
    const int n_top = (int)(YCGE_REF_PAYLOAD(S.scene_root_ref) & 7u);
    for (int i = 0; i < n_top; i++) {
        const int pi = (int)S.scene_leaf_prims[leaf_start + i];               // uniform address -> scalar loads
        const float4 *pp = (const float4 *)(S.prims + pi);
        const float4 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3];
        const int type = __float_as_int(q0.x);
        if (type == 9) {
            uint32_t start = YCGE_REF_NONE_VALUE;
            const uint32_t root_ref = __float_as_uint(q2.z);
            const bool open = root_hit && !(anyhit && hit_prim >= 0);      // an answered occlusion query looks at no further object
            if (open && root_ref != YCGE_REF_NONE_VALUE) {
                float tm;
                if (COUNT) w.box++;
                if (box_mesh(q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, o, inv, sx, sy, sz, tmin, closest, tm)) start = root_ref;
            }
            if (FULLWAVE && !COUNT && STK::kBS == 64 && S.anyhit_bfs) {
                // occlusion queries of this wavefront against the mesh: order-free, breadth-first from one shared work list (ycge_anyhit.hip.h);
                // the stacks are empty here - between the objects of the flat scene - and lend it their LDS
                const bool bfs = anyhit && start != YCGE_REF_NONE_VALUE;
                const uint32_t n_bfs = (uint32_t)__popcll(__ballot(bfs));
                if (n_bfs != 0u && n_bfs <= S.anyhit_bfs) {
                    // (every lane's ray waits in LDS meanwhile and comes back from there - the caller's copy too: nothing of it is live across the rounds)
                    bfs_park_ray(o, inv, d, tmin, closest);
                    const unsigned long long occluded = mesh_anyhit_bfs(S, bfs, start, w);
                    bfs_unpark_ray(o, inv, d, tmin, closest);
                    q.o = o; q.d = d; q.tmin = tmin;
                    if (bfs) {
                        start = YCGE_REF_NONE_VALUE;
                        if ((occluded >> (threadIdx.x & 63u)) & 1ull) { hit_prim = pi; hit_sub = 0; }      // (an occlusion query's t and triangle are never looked at)
                    }
                }
            }
            mesh_walk<COUNT, false, FULLWAVE>(S, start, pi, st, o, inv, d, sx, sy, sz, tmin, closest, hit_prim, hit_sub, w, 0x7fffffff, anyhit);     // start is consumed
        } else if (type == 10) {
            if (HAS_GRID) { if (root_hit && !(anyhit && hit_prim >= 0)) grid_dda<COUNT>(S, __float_as_int(q0.z), pi, o, d, inv, tmin, closest, hit_prim, hit_sub, w); }
        } else {
            if (root_hit && !(anyhit && hit_prim >= 0)) analytic_prim<COUNT>(q0, q1, q2, q3, type, pi, o, d, tmin, closest, hit_prim, hit_sub, w);
        }
    }
}

#if YCGE_EXPERIMENTS
} // namespace ycge
#include "experiments/ycge_refill.hip.h"
namespace ycge {
#endif

// ------------------------------------------------------------------ hit attributes
// Rebuild HitRecord {P, N, Mat} of the winning primitive from (prim, sub, t) with the same
// expressions its C# Hit() uses, so the values are the bits the reference would have stored.
struct HitAttr {
    F3 p, n;
    MatEval m;
    int sub_public;     // triangle index in input order / box face / voxel cell
};
template <bool HAS_GRID>
__device__ __forceinline__ void resolve_hit(const SceneDev &S, int prim_index, int sub, float t, F3 o, F3 d, HitAttr &h)
{
    const GPrim *P = S.prims + prim_index;
    const float4 q0 = ((const float4 *)P)[0];
    const int type = __float_as_int(q0.x);
    int material = __float_as_int(q0.y);
    const float refl_override = q0.w;
    bool override_refl = false;
    bool wire_black = false;
    h.sub_public = sub;
    if (type == 9) {    // MeshBVH.cs:177-185
        const GTriPair *tp = (const GTriPair *)(S.mesh_arena + (size_t)((uint32_t)sub >> 1) * 32u);      // sub = (record unit << 1) | slot
        const int sl = sub & 1;
        const float e1x = tp->e1x[sl], e1y = tp->e1y[sl], e1z = tp->e1z[sl], e2x = tp->e2x[sl], e2y = tp->e2y[sl], e2z = tp->e2z[sl];
        // unit normal exactly as the MeshBVH ctor computes it, MeshBVH.cs:93-97
        float nnx = e1y * e2z - e1z * e2y;
        float nny = e1z * e2x - e1x * e2z;
        float nnz = e1x * e2y - e1y * e2x;
        float inv_len = 1.0f / cs_max(1e-20f, cs_sqrt(nnx * nnx + nny * nny + nnz * nnz));
        float nx = nnx * inv_len, ny = nny * inv_len, nz = nnz * inv_len;
        h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
        float ndotd = nx * d.x + ny * d.y + nz * d.z;
        h.n = ndotd < 0.0f ? f3(nx, ny, nz) : f3(-nx, -ny, -nz);
        material = tp->material[sl];
        h.sub_public = tp->orig[sl];
    } else if (type == 10) {   // VolumeGrid.cs:160-198
        if (HAS_GRID) {
            const GGrid g = S.grids[__float_as_int(q0.z)];
            const int axis = (int)((uint32_t)sub >> 30);
            const int cell = sub & 0x3fffffff;
            const int ix = cell % g.nx, iy = (cell / g.nx) % g.ny, iz = cell / (g.nx * g.ny);
            h.sub_public = cell;
            if (axis == 0) h.n = f3(d.x > 0.0f ? -1.0f : 1.0f, 0.0f, 0.0f);
            else if (axis == 1) h.n = f3(0.0f, d.y > 0.0f ? -1.0f : 1.0f, 0.0f);
            else h.n = f3(0.0f, 0.0f, d.z > 0.0f ? -1.0f : 1.0f);
            h.p = f3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);     // Ray.At
            const uint8_t code = S.grid_cells[g.cell_offset + grid_index(g, ix, iy, iz)];
            material = S.grid_lut[g.lut_offset + code];
            if (g.wireframe) {
                const float wire_max2 = g.wire_max_distance <= 0.0f ? -1.0f : g.wire_max_distance * g.wire_max_distance;
                bool within = false;
                if (wire_max2 >= 0.0f) {
                    float dir_len2 = d.x * d.x + d.y * d.y + d.z * d.z;
                    float dist2 = t * t * dir_len2;
                    within = dist2 <= wire_max2;
                }
                // centre-block highlight (VolumeGrid.cs:176-187): dead code under this renderer at EVERY console size - it needs a query
                // with |screenV - 0.5| <= 1e-6, and vCenter = (py + 0.5f) / hiH with hiH = fbH * 2 * ss always even (RaytraceRenderer.cs:86-87,
                // 204-205) misses 0.5 by >= 0.5 / hiH; every other caller of Scene.Hit passes (0, 0).  The wire colour is always WireColor.
                if (within && is_wire_on_face(g, h.p, ix, iy, iz, axis)) wire_black = true;
            }
        }
    } else {
        const float4 q1 = ((const float4 *)P)[1], q2 = ((const float4 *)P)[2], q3 = ((const float4 *)P)[3];
        const float p[12] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        switch (type) {
        case 0: {
            float px = o.x + t * d.x, py = o.y + t * d.y, pz = o.z + t * d.z;
            float inv_r = 1.0f / p[3];
            h.p = f3(px, py, pz);
            h.n = f3((px - p[0]) * inv_r, (py - p[1]) * inv_r, (pz - p[2]) * inv_r);
            break;
        }
        case 1: {
            float denom = p[0] * d.x + p[1] * d.y + p[2] * d.z;
            h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
            h.n = denom < 0.0f ? f3(p[0], p[1], p[2]) : f3(-p[0], -p[1], -p[2]);
            override_refl = true;
            break;
        }
        case 2: {
            F3 n = f3(p[3], p[4], p[5]);
            float denom = dot(n, d);
            h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
            h.n = denom < 0.0f ? n : -n;
            override_refl = true;
            break;
        }
        case 3: case 4: case 5: case 6: {
            int axis; float a0, a1, b0, b1, k;
            if (type == 6) box_face(p, sub, axis, a0, a1, b0, b1, k);
            else { axis = type == 3 ? 2 : type == 4 ? 1 : 0; a0 = p[0]; a1 = p[1]; b0 = p[2]; b1 = p[3]; k = p[4]; }
            float dir_k = axis == 2 ? d.z : axis == 1 ? d.y : d.x;
            float nk = cs_copysign(1.0f, -dir_k);
            if (axis == 2) { h.p = f3(o.x + t * d.x, o.y + t * d.y, k); h.n = f3(0.0f, 0.0f, nk); }
            else if (axis == 1) { h.p = f3(o.x + t * d.x, k, o.z + t * d.z); h.n = f3(0.0f, nk, 0.0f); }
            else { h.p = f3(k, o.y + t * d.y, o.z + t * d.z); h.n = f3(nk, 0.0f, 0.0f); }
            override_refl = true;
            break;
        }
        case 7: {
            float ox = o.x - p[0], oz = o.z - p[1];
            F3 hn;
            if (sub == 0) hn = f3((ox + t * d.x) / p[2], 0.0f, (oz + t * d.z) / p[2]);
            else if (sub == 1) hn = f3(0.0f, 1.0f, 0.0f);
            else hn = f3(0.0f, -1.0f, 0.0f);
            h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
            h.n = dot(hn, d) < 0.0f ? hn : -hn;
            break;
        }
        default: {  // 8 Triangle
            h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
            float nd = p[9] * d.x + p[10] * d.y + p[11] * d.z;
            h.n = nd < 0.0f ? f3(p[9], p[10], p[11]) : f3(-p[9], -p[10], -p[11]);
            break;
        }
        }
    }
    h.m = eval_material(S, material, h.p);
    if (override_refl) h.m.reflectivity = refl_override;
    if (wire_black) h.m.albedo = f3(0.0f, 0.0f, 0.0f);
}

#ifndef YCGE_TEXTURES
#define YCGE_TEXTURES 1         // 0: an A/B build without the texture branch at the shade sites (profiles/tex_ab.sh)
#endif
// SampleAlbedo's texture branch (RaytraceRenderer.cs:724-735) for the hit resolve_hit has just described - called where a hit is SHADED,
// and only in scenes that have a textured material (S.any_textured is uniform: untextured scenes skip it with one scalar branch).
__device__ __forceinline__ void apply_texture(const SceneDev &S, int prim_index, int sub, F3 o, F3 d, HitAttr &h)
{
    const GPrim *P = S.prims + prim_index;
    const float4 q0 = ((const float4 *)P)[0];
    const int type = __float_as_int(q0.x);
    int material = __float_as_int(q0.y);
    if (type == 9) material = ((const GTriPair *)(S.mesh_arena + (size_t)((uint32_t)sub >> 1) * 32u))->material[sub & 1];
    if (type == 10) return;                                  // VolumeGrid.Hit leaves (U, V) = (0, 0) and its palette materials carry no texture
    const float4 mk = ((const float4 *)(S.materials + material))[0], me = ((const float4 *)(S.materials + material))[4];
    const int tex = __float_as_int(me.y);
    if (__float_as_int(mk.x) != 2 || tex < 0) return;
    {
        // HitRecord.U / V as the hit routines leave them, recomputed from the ray: barycentrics of a mesh triangle (MeshBVH.cs:296-303) or
        // a Triangle (Triangle.cs:136-174), the plane coordinates of a rectangle / box face over its span (Surfaces.cs:209-212, 281-284,
        // 353-356), 0 for everything else
        float u = 0.0f, v = 0.0f;
        if (type == 9) {
            const GTriPair *tp = (const GTriPair *)(S.mesh_arena + (size_t)((uint32_t)sub >> 1) * 32u);
            const int sl = sub & 1;
            const float e1x = tp->e1x[sl], e1y = tp->e1y[sl], e1z = tp->e1z[sl], e2x = tp->e2x[sl], e2y = tp->e2y[sl], e2z = tp->e2z[sl];
            const float px = d.y * e2z - d.z * e2y, py = d.z * e2x - d.x * e2z, pz = d.x * e2y - d.y * e2x;
            const float det = e1x * px + e1y * py + e1z * pz;
            const float sx = o.x - tp->ax[sl], sy = o.y - tp->ay[sl], sz = o.z - tp->az[sl];
            const float u_num = sx * px + sy * py + sz * pz;
            const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
            const float v_num = d.x * qx + d.y * qy + d.z * qz;
            const float inv_det = 1.0f / det;
            u = u_num * inv_det; v = v_num * inv_det;
        } else if (type == 8) {
            const float4 q1 = ((const float4 *)P)[1], q2 = ((const float4 *)P)[2], q3 = ((const float4 *)P)[3];
            const float e1x = q1.w, e1y = q2.x, e1z = q2.y, e2x = q2.z, e2y = q2.w, e2z = q3.x;
            const float px = d.y * e2z - d.z * e2y, py = d.z * e2x - d.x * e2z, pz = d.x * e2y - d.y * e2x;
            const float det = e1x * px + e1y * py + e1z * pz;
            const float inv_det = 1.0f / det;
            const float sx = o.x - q1.x, sy = o.y - q1.y, sz = o.z - q1.z;
            u = (sx * px + sy * py + sz * pz) * inv_det;
            const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
            v = (d.x * qx + d.y * qy + d.z * qz) * inv_det;
        } else if (type >= 3 && type <= 6) {
            const float4 q1 = ((const float4 *)P)[1], q2 = ((const float4 *)P)[2], q3 = ((const float4 *)P)[3];
            const float p[12] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            int axis; float a0, a1, b0, b1, k;
            if (type == 6) box_face(p, sub, axis, a0, a1, b0, b1, k);
            else { axis = type == 3 ? 2 : type == 4 ? 1 : 0; a0 = p[0]; a1 = p[1]; b0 = p[2]; b1 = p[3]; k = p[4]; }
            const float pa = axis == 0 ? h.p.y : h.p.x, pb = axis == 2 ? h.p.y : h.p.z;     // the two in-plane coordinates in the class's field order
            u = (pa - a0) * (1.0f / (a1 - a0));
            v = (pb - b0) * (1.0f / (b1 - b0));
        }
        h.m.albedo = sample_albedo(S, h.m.albedo, tex, me.z, me.w, u, v);
    }
}

} // namespace ycge
