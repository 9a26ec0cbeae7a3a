// ycge_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ray-trace core.
//
// K_trace   camera-ray generation + the whole per-pixel path of
//           RaytraceRenderer.TraceFull (reference RayTracing/RaytraceRenderer.cs:419-620)
//           as ONE persistent-state "while-while" kernel: every lane carries a small
//           path state machine and all lanes of a wavefront — whether they are on a
//           primary, bounce or shadow query — share the one traversal loop, so the
//           hot loop (node fetch, two slab tests, triangle tests) is the only code the
//           wave spends divergent time in.
// K_taa     TemporalBlendWithClamp (RaytraceRenderer.cs:274-398), one thread per pixel.
// K_unpermute  tile slabs gathered from all ranks -> full-frame buffers (multi-GPU).
//
// Arithmetic contract: see ycge_math.h (compiled with -ffp-contract=off; results are
// bit-identical to the C# scalar path for everything built from + - * / sqrt floor).
// No MFMA: this is branchy pointer-chasing, bounded by HBM/L2 latency and bandwidth.
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
};

struct Work {              // SURVEY 8(d) counters
    unsigned rays, box, tri, prim, vox;
};

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
__device__ __noinline__ F3 oren_nayar(F3 albedo, F3 n, F3 wo, F3 wi, float sigma_rad)
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
    float sigma2 = sigma_rad * sigma_rad;
    float A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    float B = 0.45f * sigma2 / (sigma2 + 0.09f);
    float sin_alpha = cs_max(sin_i, sin_o);
    float tan_beta = cs_min(sin_i / cs_max(1e-6f, cos_i), sin_o / cs_max(1e-6f, cos_o));
    float on = (A + B * cos_phi * sin_alpha * tan_beta);
    F3 f = albedo * (on * inv_pi);
    return saturate(f);
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

// ------------------------------------------------------------------ box tests
// BVH.BoxHitFast, BVH.cs:201-236: NaN-propagating Max/Min, clamp to [tMin, tMax]
__device__ __forceinline__ bool box_scene(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 inv,
                                          float tmin, float tmax, float &tnear)
{
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
    tnear = t_enter;
    return t_exit >= t_enter;
}
// MeshBVH.BoxHitFast, MeshBVH.cs:308-332: sign-indexed slabs, compare chain (NaN never wins).
// The two early-outs of the C# are pure shortcuts: the final test fails whenever one of them would.
__device__ __forceinline__ bool box_mesh(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 inv,
                                         bool sx, bool sy, bool sz, float tmin, float tmax, float &tnear)
{
    float tx_en = ((sx ? mxx : mnx) - o.x) * inv.x;
    float tx_ex = ((sx ? mnx : mxx) - o.x) * inv.x;
    if (tx_en > tmin) tmin = tx_en;
    if (tx_ex < tmax) tmax = tx_ex;
    float ty_en = ((sy ? mxy : mny) - o.y) * inv.y;
    float ty_ex = ((sy ? mny : mxy) - o.y) * inv.y;
    if (ty_en > tmin) tmin = ty_en;
    if (ty_ex < tmax) tmax = ty_ex;
    float tz_en = ((sz ? mxz : mnz) - o.z) * inv.z;
    float tz_ex = ((sz ? mnz : mxz) - o.z) * inv.z;
    if (tz_en > tmin) tmin = tz_en;
    if (tz_ex < tmax) tmax = tz_ex;
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
__device__ __noinline__ void analytic_prim(const float4 q0, const float4 q1, const float4 q2, const float4 q3, int type, int prim_index,
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
__device__ __forceinline__ int morton3_3bits(int x, int y, int z)   // VolumeGrid.cs:246-252
{
    return ((x & 1) << 0) | ((y & 1) << 1) | ((z & 1) << 2) | ((x & 2) << 2) | ((y & 2) << 3) | ((z & 2) << 4) | ((x & 4) << 4) | ((y & 4) << 5) | ((z & 4) << 6);
}
__device__ __forceinline__ uint32_t grid_index(const GGrid &g, int ix, int iy, int iz)   // VolumeGrid.cs:235-242
{
    int brick = (((iz >> 3) * g.nby) + (iy >> 3)) * g.nbx + (ix >> 3);
    return (uint32_t)(brick * 512 + morton3_3bits(ix & 7, iy & 7, iz & 7));
}
__device__ __forceinline__ bool grid_slab(float ro, float rd, float mn, float mx, float &t_enter, float &t_exit, int axis, int &enter_axis)
{   // VolumeGrid.Slab, VolumeGrid.cs:331-355
    if (cs_abs(rd) < 1e-12f) {
        if (ro < mn || ro > mx) return false;
        return true;
    }
    float inv = 1.0f / rd;
    float t0 = (mn - ro) * inv;
    float t1 = (mx - ro) * inv;
    if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
    if (t0 > t_enter) { t_enter = t0; enter_axis = axis; }
    if (t1 < t_exit) t_exit = t1;
    return t_exit >= t_enter;
}

// VolumeGrid.Hit, VolumeGrid.cs:99-231 (Amanatides-Woo DDA; first cell with matId > 0 hits)
template <bool COUNT>
__device__ __noinline__ void grid_dda(const SceneDev &S, int grid_index_, int prim_index, F3 o, F3 d, float tmin, float &closest,
                                      int &hit_prim, int &hit_sub, Work &w)
{
    if (COUNT) w.prim++;
    const GGrid g = S.grids[grid_index_];
    const float min_x = g.min_corner[0], min_y = g.min_corner[1], min_z = g.min_corner[2];
    const float size_x = g.voxel_size[0], size_y = g.voxel_size[1], size_z = g.voxel_size[2];
    const float max_x = min_x + (float)g.nx * size_x, max_y = min_y + (float)g.ny * size_y, max_z = min_z + (float)g.nz * size_z;
    const float tmax = closest;
    int enter_axis = -1;
    float t_enter = -YCGE_INF, t_exit = YCGE_INF;
    if (!grid_slab(o.x, d.x, min_x, max_x, t_enter, t_exit, 0, enter_axis)) return;
    if (!grid_slab(o.y, d.y, min_y, max_y, t_enter, t_exit, 1, enter_axis)) return;
    if (!grid_slab(o.z, d.z, min_z, max_z, t_enter, t_exit, 2, enter_axis)) return;
    if (!(t_exit >= cs_max(0.0f, t_enter))) return;
    float t = t_enter; if (t < tmin) t = tmin; if (t > tmax || t > t_exit) return;
    t += 1e-6f;
    float px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    int ix = cs_f2i(cs_floor((px - min_x) / size_x)); if (ix < 0) ix = 0; else if (ix >= g.nx) ix = g.nx - 1;
    int iy = cs_f2i(cs_floor((py - min_y) / size_y)); if (iy < 0) iy = 0; else if (iy >= g.ny) iy = g.ny - 1;
    int iz = cs_f2i(cs_floor((pz - min_z) / size_z)); if (iz < 0) iz = 0; else if (iz >= g.nz) iz = g.nz - 1;
    const int step_x = d.x > 0.0f ? 1 : d.x < 0.0f ? -1 : 0;
    const int step_y = d.y > 0.0f ? 1 : d.y < 0.0f ? -1 : 0;
    const int step_z = d.z > 0.0f ? 1 : d.z < 0.0f ? -1 : 0;
    const float inv_dx = step_x == 0 ? 0.0f : 1.0f / d.x;
    const float inv_dy = step_y == 0 ? 0.0f : 1.0f / d.y;
    const float inv_dz = step_z == 0 ? 0.0f : 1.0f / d.z;
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
    while (t <= t_exit && t <= tmax) {
        if ((uint32_t)ix < (uint32_t)g.nx && (uint32_t)iy < (uint32_t)g.ny && (uint32_t)iz < (uint32_t)g.nz) {
            if (COUNT) w.vox++;
            if (cells[grid_index(g, ix, iy, iz)] != 0) {
                closest = cs_max(t, tmin);
                hit_prim = prim_index;
                hit_sub = (ix + g.nx * (iy + g.ny * iz)) | (last_axis << 30);
                return;
            }
        }
        if (t_max_x <= t_max_y && t_max_x <= t_max_z) { ix += step_x; t = t_max_x; t_max_x += t_delta_x; last_axis = 0; }
        else if (t_max_y <= t_max_z) { iy += step_y; t = t_max_y; t_max_y += t_delta_y; last_axis = 1; }
        else { iz += step_z; t = t_max_z; t_max_z += t_delta_z; last_axis = 2; }
        if ((uint32_t)ix >= (uint32_t)g.nx || (uint32_t)iy >= (uint32_t)g.ny || (uint32_t)iz >= (uint32_t)g.nz) break;
    }
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

// ------------------------------------------------------------------ closest-hit traversal
// Scene.Hit -> BVH.Hit (BVH.cs:99-198) with Mesh -> MeshBVH.Hit (MeshBVH.cs:132-236) inlined into ONE loop.
//
// Visit order and pruning are exactly the reference's:
//  * children are tested against [tMin, closest]; both hit -> the nearer is visited first and the
//    other is stacked (ties: `lNear < rNear` false -> right first);
//  * the reference re-tests a node's own box when it pops it; with the entry distance tNear kept
//    next to the reference on the stack that re-test is `closest >= tNear` (same predicate, no
//    re-fetch: the raw slab values cannot change, only `closest` shrank);
//  * a scene leaf's objects are queued in order; a Mesh object opens its own tree on the same
//    stack and runs to completion before the next object of the leaf is tried.
// Per-lane stack entries are {ref, tNear}; hit_sub of a mesh hit is the LEAF-ORDER triangle index.
struct StackScratch {
    uint2 e[YCGE_TRAVERSAL_STACK];
    int sp;
    __device__ __forceinline__ void init() { sp = 0; }
    __device__ __forceinline__ void push(uint32_t ref, float tnear) { e[sp] = make_uint2(ref, __float_as_uint(tnear)); sp++; }
    __device__ __forceinline__ bool pop(uint32_t &ref, float &tnear)
    {
        if (sp == 0) return false;
        sp--;
        uint2 v = e[sp];
        ref = v.x; tnear = __uint_as_float(v.y);
        return true;
    }
};

template <bool COUNT>
__device__ __forceinline__ void traverse(const SceneDev &S, const RayQ &q, float &closest, int &hit_prim, int &hit_sub, Work &w)
{
    const F3 o = q.o, d = q.d;
    const float tmin = q.tmin;
    closest = q.tmax;
    hit_prim = -1;
    hit_sub = 0;
    if (COUNT) w.rays++;
    if (S.scene_root_ref == YCGE_REF_NONE_VALUE) return;
    const F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;

    StackScratch st;
    st.init();
    uint32_t cur = YCGE_REF_NONE_VALUE;
    int mesh_prim = -1;
    {
        float tn;
        if (COUNT) w.box++;
        if (box_scene(S.scene_root_min[0], S.scene_root_min[1], S.scene_root_min[2], S.scene_root_max[0], S.scene_root_max[1],
                      S.scene_root_max[2], o, inv, tmin, closest, tn))
            cur = S.scene_root_ref;
    }
    for (;;) {
        if (cur == YCGE_REF_NONE_VALUE) {
            float tn;
            if (!st.pop(cur, tn)) break;
            if (!(closest >= tn)) { cur = YCGE_REF_NONE_VALUE; continue; }
        }
        const uint32_t kind = YCGE_REF_KIND(cur), pay = YCGE_REF_PAYLOAD(cur);
        if (kind == REF_MESH_NODE || kind == REF_SCENE_NODE) {
            const float4 *np = (const float4 *)((kind == REF_MESH_NODE ? S.mesh_nodes : S.scene_nodes) + pay);
            const float4 a = np[0], b = np[1], c = np[2], e = np[3];
            float ln, rn;
            bool hl, hr;
            if (COUNT) w.box += 2;
            if (kind == REF_MESH_NODE) {
                hl = box_mesh(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, sx, sy, sz, tmin, closest, ln);
                hr = box_mesh(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, sx, sy, sz, tmin, closest, rn);
            } else {
                hl = box_scene(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tmin, closest, ln);
                hr = box_scene(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tmin, closest, rn);
            }
            const uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
            if (hl & hr) {
                if (ln < rn) { st.push(rref, rn); cur = lref; }
                else { st.push(lref, ln); cur = rref; }
            } else if (hl) cur = lref;
            else if (hr) cur = rref;
            else cur = YCGE_REF_NONE_VALUE;
        } else if (kind == REF_MESH_LEAF) {
            const uint32_t start = pay >> 4, count = pay & 15u;
            for (uint32_t i = 0; i < count; i++) {
                // MeshBVH.TriHit, MeshBVH.cs:239-304
                const float4 *tp = (const float4 *)(S.tris + start + i);
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                const float ax = t0.x, ay = t0.y, az = t0.z, e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
                if (COUNT) w.tri++;
                float px = d.y * e2z - d.z * e2y;
                float py = d.z * e2x - d.x * e2z;
                float pz = d.x * e2y - d.y * e2x;
                float det = e1x * px + e1y * py + e1z * pz;
                if (det > -1e-8f && det < 1e-8f) continue;
                float sxx = o.x - ax, syy = o.y - ay, szz = o.z - az;
                float u_num = sxx * px + syy * py + szz * pz;
                float sgn = det > 0.0f ? 1.0f : -1.0f;
                float det_abs = det * sgn;
                float u_num_s = u_num * sgn;
                if (u_num_s < 0.0f || u_num_s > det_abs) continue;
                float qx = syy * e1z - szz * e1y;
                float qy = szz * e1x - sxx * e1z;
                float qz = sxx * e1y - syy * e1x;
                float v_num = d.x * qx + d.y * qy + d.z * qz;
                float v_num_s = v_num * sgn;
                float uv_sum_s = u_num_s + v_num_s;
                if (v_num_s < 0.0f || uv_sum_s > det_abs) continue;
                float t_num = e2x * qx + e2y * qy + e2z * qz;
                float t_num_s = t_num * sgn;
                float t_min_scaled = tmin * det_abs;
                float t_max_scaled = closest * det_abs;
                if (t_num_s < t_min_scaled || t_num_s > t_max_scaled) continue;
                float inv_det = 1.0f / det;
                closest = t_num * inv_det;
                hit_prim = mesh_prim;
                hit_sub = (int)(start + i);
            }
            cur = YCGE_REF_NONE_VALUE;
        } else if (kind == REF_SCENE_LEAF) {
            const uint32_t start = pay >> 3, count = pay & 7u;
            for (uint32_t i = count - 1; i >= 1; i--) st.push(YCGE_REF(REF_PRIM, S.scene_leaf_prims[start + i]), -YCGE_INF);
            cur = YCGE_REF(REF_PRIM, S.scene_leaf_prims[start]);
        } else {    // REF_PRIM: objectHit[objId](r, tMin, closest, ...), BVH.cs:139-149
            const float4 *pp = (const float4 *)(S.prims + pay);
            const float4 q0 = pp[0];
            const int type = __float_as_int(q0.x);
            cur = YCGE_REF_NONE_VALUE;
            if (type == 9) {            // Mesh.Hit -> MeshBVH.Hit: root pushed, popped, own box tested
                const GMesh *m = S.meshes + __float_as_int(q0.z);
                const float4 m0 = ((const float4 *)m)[0], m1 = ((const float4 *)m)[1];
                const uint32_t root_ref = __float_as_uint(m1.z);
                if (root_ref != YCGE_REF_NONE_VALUE) {
                    float tn;
                    if (COUNT) w.box++;
                    if (box_mesh(m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, o, inv, sx, sy, sz, tmin, closest, tn)) {
                        cur = root_ref;
                        mesh_prim = (int)pay;
                    }
                }
            } else if (type == 10) {
                grid_dda<COUNT>(S, __float_as_int(q0.z), (int)pay, o, d, tmin, closest, hit_prim, hit_sub, w);
            } else {
                analytic_prim<COUNT>(q0, pp[1], pp[2], pp[3], type, (int)pay, o, d, tmin, closest, hit_prim, hit_sub, w);
            }
        }
    }
}

// ------------------------------------------------------------------ hit attributes
// Rebuild HitRecord {P, N, Mat} of the winning primitive from (prim, sub, t) with the same
// expressions its C# Hit() uses, so the values are the bits the reference would have stored.
struct HitAttr {
    F3 p, n;
    MatEval m;
    int sub_public;     // triangle index in input order / box face / voxel cell
};
__device__ __noinline__ void resolve_hit(const SceneDev &S, int prim_index, int sub, float t, F3 o, F3 d, HitAttr &h)
{
    const GPrim *P = S.prims + prim_index;
    const float4 q0 = ((const float4 *)P)[0];
    const int type = __float_as_int(q0.x);
    const int material = __float_as_int(q0.y);
    const float refl_override = q0.w;
    h.sub_public = sub;
    if (type == 9) {    // MeshBVH.cs:177-185
        const float4 *tp = (const float4 *)(S.tris + sub);
        const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
        const float e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
        // unit normal exactly as the MeshBVH ctor computes it, MeshBVH.cs:93-97
        float nnx = e1y * e2z - e1z * e2y;
        float nny = e1z * e2x - e1x * e2z;
        float nnz = e1x * e2y - e1y * e2x;
        float inv_len = 1.0f / cs_max(1e-20f, cs_sqrt(nnx * nnx + nny * nny + nnz * nnz));
        float nx = nnx * inv_len, ny = nny * inv_len, nz = nnz * inv_len;
        h.p = f3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
        float ndotd = nx * d.x + ny * d.y + nz * d.z;
        h.n = ndotd < 0.0f ? f3(nx, ny, nz) : f3(-nx, -ny, -nz);
        h.m = eval_material(S, __float_as_int(t2.z), h.p);
        h.sub_public = __float_as_int(t2.y);
        return;
    }
    if (type == 10) {   // VolumeGrid.cs:160-198
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
        h.m = eval_material(S, S.grid_lut[g.lut_offset + code], h.p);
        if (g.wireframe) {
            const float wire_max2 = g.wire_max_distance <= 0.0f ? -1.0f : g.wire_max_distance * g.wire_max_distance;
            bool within = false;
            if (wire_max2 >= 0.0f) {
                float dir_len2 = d.x * d.x + d.y * d.y + d.z * d.z;
                float dist2 = t * t * dir_len2;
                within = dist2 <= wire_max2;
            }
            // centre-block highlight (VolumeGrid.cs:176-187): shared mutable state raced by all pixel
            // threads in the reference; unreachable when hiW or hiH is even. Not modelled.
            if (within && is_wire_on_face(g, h.p, ix, iy, iz, axis)) h.m.albedo = f3(0.0f, 0.0f, 0.0f);
        }
        return;
    }
    const float4 q1 = ((const float4 *)P)[1], q2 = ((const float4 *)P)[2], q3 = ((const float4 *)P)[3];
    const float p[12] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    bool override_refl = false;
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
    h.m = eval_material(S, material, h.p);
    if (override_refl) h.m.reflectivity = refl_override;
}

// ------------------------------------------------------------------ K_trace
enum Phase : int { PH_PATH = 0, PH_SHADOW_OCC = 1, PH_SHADOW_TR = 2, PH_DONE = 3 };

struct PathItem {       // PathWorkItem, RaytraceRenderer.cs:439-446 (IsPrimary is false for every pushed item)
    F3 o, d, beta;
    int mirror_depth, diffuse_depth;
};

template <bool COUNT, bool DEBUG, bool SLAB>
__global__ __launch_bounds__(256) void k_trace(const SceneDev S, const FrameParams P, const TraceOut O)
{
    // ---- pixel of this lane: 32x8 tile per block, 8x8 sub-tile per wavefront
    const int k = blockIdx.x;
    const int tile_id = P.rank + k * P.world_size;
    const int tx = tile_id % P.tiles_x, ty = tile_id / P.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = wave * 8 + (lane & 7), ly = lane >> 3;
    const int px = tx * YCGE_TILE_W + lx, py = ty * YCGE_TILE_H + ly;
    const bool in_image = px < P.hiW && py < P.hiH;

    Work w = {0, 0, 0, 0, 0};

    // ---- MakeJitteredRay, RaytraceRenderer.cs:419-437
    RayQ q;
    {
        float jx = frac(blue_noise_sample(px, py, P.frame_idx, 0) + P.rot_x) - 0.5f;
        float jy = frac(blue_noise_sample(px, py, P.frame_idx, 1) + P.rot_y) - 0.5f;
        float u = (((float)px + 0.5f + jx) / (float)P.hiW) * 2.0f - 1.0f;
        float v = 1.0f - (((float)py + 0.5f + jy) / (float)P.hiH) * 2.0f;
        F3 fwd = f3(P.fwd), right = f3(P.right), up = f3(P.up);
        F3 dir = normalized(fwd + right * (u * P.half_w) + up * (v * P.half_h));
        q.o = f3(P.cam_pos);
        q.d = normalized(dir);      // Ray ctor normalises again, Ray.cs:8-12
        q.tmin = 0.001f;
        q.tmax = YCGE_FLT_MAX;
    }
    if (DEBUG && in_image && O.rays) {
        float *r = O.rays + ((size_t)px + (size_t)py * P.hiW) * 6;
        r[0] = q.o.x; r[1] = q.o.y; r[2] = q.o.z; r[3] = q.d.x; r[4] = q.d.y; r[5] = q.d.z;
    }

    // ---- TraceFull state, RaytraceRenderer.cs:448-460
    uint64_t rng = per_frame_seed(px, py, P.frame, P.seed_salt);
    if (rng == 0) rng = 0x9E3779B97F4A7C15ULL;
    F3 radiance = f3(0, 0, 0), beta = f3(1, 1, 1);
    int mirror_depth = 0, diffuse_depth = 0;
    bool item_is_primary = true, primary_hit_something = false, gbuf_valid = false, is_sky = false;
    F3 g_albedo = f3(0, 0, 0), g_normal = f3(0, 0, 0);
    float g_depth = YCGE_FLT_MAX;
    int g_prim = -1, g_sub = 0;
    PathItem pstack[3];     // a 16-entry stack in the C#; occupancy never exceeds 3 (mirrorDepth < 2 gates pushes)
    int psp = 0;
    // shading context kept across the shadow queries of one hit
    F3 sh_p = f3(0, 0, 0), sh_n = f3(0, 0, 0), sh_alb = f3(0, 0, 0), sh_wo = f3(0, 0, 0);
    F3 path_d = q.d;
    int light = 0;
    float tr_r = 1.0f, tr_g = 1.0f, tr_b = 1.0f, sh_maxdist = 0.0f;
    int tr_counter = 0;
    int phase = in_image ? PH_PATH : PH_DONE;

    for (;;) {
        if (!__any(phase != PH_DONE)) break;
        float t_hit = 0.0f;
        int hit_prim = -1, hit_sub = 0;
        if (phase != PH_DONE) traverse<COUNT>(S, q, t_hit, hit_prim, hit_sub, w);
        if (phase == PH_DONE) continue;
        const bool hit = hit_prim >= 0;

        // ================= result of a path query (scene.Hit at :472) =================
        bool go_lights = false, go_next_light = false, go_contrib = false, go_next_item = false;
        if (phase == PH_PATH) {
            if (!hit) {
                float tbg = 0.5f * (q.d.y + 1.0f);
                F3 sky = lerp3(f3(S.bg_bottom), f3(S.bg_top), tbg);
                if (item_is_primary && !primary_hit_something) {
                    is_sky = true;
                    if (!gbuf_valid) { g_albedo = f3(0, 0, 0); g_normal = f3(0, 0, 0); g_depth = YCGE_FLT_MAX; g_prim = -1; g_sub = 0; gbuf_valid = true; }
                }
                radiance = radiance + f3(beta.x * sky.x, beta.y * sky.y, beta.z * sky.z);
                go_next_item = true;
            } else {
                HitAttr h;
                resolve_hit(S, hit_prim, hit_sub, t_hit, q.o, q.d, h);
                if (item_is_primary) {
                    primary_hit_something = true;
                    is_sky = false;
                    if (!gbuf_valid) { g_albedo = h.m.albedo; g_normal = h.n; g_depth = t_hit; g_prim = hit_prim; g_sub = h.sub_public; gbuf_valid = true; }
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
                            pstack[psp++] = it;
                        }
                        if (T > 0.0f && psp < 3) {
                            PathItem it;
                            it.o = h.p - nl * P.eps;
                            it.d = normalized(normalized(refr_dir));
                            F3 tint = h.m.trans_color;
                            it.beta = f3(beta.x * tint.x * T, beta.y * tint.y * T, beta.z * tint.z * T);
                            it.mirror_depth = mirror_depth + 1; it.diffuse_depth = diffuse_depth;
                            pstack[psp++] = it;
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
                        beta = f3(beta.x * base_albedo.x, beta.y * base_albedo.y, beta.z * base_albedo.z);
                        mirror_depth++;
                        phase = PH_PATH;
                    }
                } else {
                    if (S.ambient_intensity > 0.0f) {   // :571-576
                        F3 a = f3(S.ambient[0] * S.ambient_intensity, S.ambient[1] * S.ambient_intensity, S.ambient[2] * S.ambient_intensity);
                        F3 amb = f3(a.x * base_albedo.x, a.y * base_albedo.y, a.z * base_albedo.z);
                        radiance = radiance + f3(beta.x * amb.x, beta.y * amb.y, beta.z * amb.z);
                    }
                    sh_p = h.p; sh_n = h.n; sh_alb = base_albedo;
                    sh_wo = normalized(q.d * -1.0f);
                    path_d = q.d;
                    light = 0;
                    go_lights = true;
                }
            }
        } else if (phase == PH_SHADOW_OCC) {            // VolumeScene: binary occlusion, :761-765
            if (hit) { tr_r = tr_g = tr_b = 0.0f; } else { tr_r = tr_g = tr_b = 1.0f; }
            go_contrib = true;
        } else {                                        // ComputeTransmittanceToLight loop body, :773-796
            if (!hit) {
                go_contrib = true;
            } else {
                tr_counter++;
                HitAttr h;
                resolve_hit(S, hit_prim, hit_sub, t_hit, q.o, q.d, h);
                float tr = h.m.transparency;
                if (tr <= 0.0f) { tr_r = tr_g = tr_b = 0.0f; go_contrib = true; }
                else {
                    F3 tint = h.m.trans_color;
                    tr_r *= tint.x * tr; tr_g *= tint.y * tr; tr_b *= tint.z * tr;
                    if (tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f) { tr_r = tr_g = tr_b = 0.0f; go_contrib = true; }
                    else if (t_hit > sh_maxdist) go_contrib = true;
                    else {
                        q.tmin = t_hit + P.eps;
                        if (tr_counter < P.max_refractions) { /* same ray, next segment */ }
                        else go_contrib = true;
                    }
                }
            }
        }

        // ================= light contribution after its shadow query, :592-602 =================
        if (go_contrib) {
            if (!(tr_r <= 1e-6f && tr_g <= 1e-6f && tr_b <= 1e-6f)) {
                const GLight &L = S.lights[light];
                F3 to_l = f3(L.pos) - sh_p;
                float dist2 = dot(to_l, to_l);
                float dist = cs_sqrt(dist2);
                F3 ldir = vdiv(to_l, dist);
                float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                float atten = L.intensity / dist2;
                F3 f_diffuse = oren_nayar(sh_alb, sh_n, sh_wo, ldir, P.sigma_rad);
                F3 Li = f3(L.color) * atten;
                F3 contrib = (f_diffuse * n_dot_l) * Li;
                contrib = f3(contrib.x * tr_r, contrib.y * tr_g, contrib.z * tr_b);
                radiance = radiance + f3(beta.x * contrib.x, beta.y * contrib.y, beta.z * contrib.z);
            }
            light++;
            go_next_light = true;
        }

        // ================= light loop head, :578-591 =================
        if (go_lights || go_next_light) {
            bool queued = false;
            for (; light < S.n_lights; light++) {
                const GLight &L = S.lights[light];
                F3 to_l = f3(L.pos) - sh_p;
                float dist2 = dot(to_l, to_l);
                float dist = cs_sqrt(dist2);
                F3 ldir = vdiv(to_l, dist);
                float n_dot_l = cs_max(0.0f, dot(sh_n, ldir));
                if (n_dot_l <= 0.0f) continue;
                q.o = sh_p + sh_n * P.eps;
                q.d = normalized(ldir);                 // new Ray(..., ldir)
                sh_maxdist = dist - P.eps;
                q.tmax = sh_maxdist;
                if (S.is_volume_scene) { q.tmin = 0.001f; phase = PH_SHADOW_OCC; }
                else { q.tmin = 0.0f + P.eps; tr_r = tr_g = tr_b = 1.0f; tr_counter = 0; phase = PH_SHADOW_TR; }
                if (!S.is_volume_scene && !(tr_counter < P.max_refractions)) {
                    // while-condition false before the first Scene.Hit: transmittance stays 1
                    continue;   // unreachable with MaxRefractions = 2; kept for the contract
                }
                queued = true;
                break;
            }
            if (!queued) {                              // bounce, :604-616
                if (diffuse_depth < P.diffuse_bounces) {
                    F3 bounce = cosine_sample_hemisphere(sh_n, rng);
                    F3 f_on = oren_nayar(sh_alb, sh_n, sh_wo, bounce, P.sigma_rad);
                    const float factor = YCGE_PI;
                    F3 mult = f3(f_on.x * factor, f_on.y * factor, f_on.z * factor);
                    q.o = sh_p + sh_n * P.eps;
                    q.d = normalized(bounce);
                    q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
                    beta = f3(beta.x * mult.x, beta.y * mult.y, beta.z * mult.z);
                    diffuse_depth++;
                    phase = PH_PATH;
                } else {
                    go_next_item = true;
                }
            }
        }

        // ================= outer while (sp > 0), :461-468 =================
        if (go_next_item) {
            if (psp == 0) {
                phase = PH_DONE;
            } else {
                PathItem it = pstack[--psp];
                q.o = it.o; q.d = it.d; q.tmin = 0.001f; q.tmax = YCGE_FLT_MAX;
                beta = it.beta; mirror_depth = it.mirror_depth; diffuse_depth = it.diffuse_depth;
                item_is_primary = false;
                phase = PH_PATH;
            }
        }
    }
    (void)path_d;

    // ---- per-pixel outputs, RaytraceRenderer.cs:210-215
    if (in_image) {
        if (SLAB) {
            float *s = O.slab + ((size_t)k * (YCGE_TILE_W * YCGE_TILE_H) + (size_t)(ly * YCGE_TILE_W + lx)) * YCGE_SLAB_FLOATS;
            s[0] = radiance.x; s[1] = radiance.y; s[2] = radiance.z;
            s[3] = g_albedo.x; s[4] = g_albedo.y; s[5] = g_albedo.z;
            s[6] = g_normal.x; s[7] = g_normal.y; s[8] = g_normal.z;
            s[9] = g_depth; s[10] = is_sky ? 1.0f : 0.0f;
        } else {
            const size_t i = (size_t)px + (size_t)py * P.hiW;
            O.current_hdr[3 * i + 0] = radiance.x; O.current_hdr[3 * i + 1] = radiance.y; O.current_hdr[3 * i + 2] = radiance.z;
            O.g_albedo[3 * i + 0] = g_albedo.x; O.g_albedo[3 * i + 1] = g_albedo.y; O.g_albedo[3 * i + 2] = g_albedo.z;
            O.g_normal[3 * i + 0] = g_normal.x; O.g_normal[3 * i + 1] = g_normal.y; O.g_normal[3 * i + 2] = g_normal.z;
            O.g_depth[i] = g_depth;
            O.sky[i] = is_sky ? 1 : 0;
        }
        if (DEBUG) {
            const size_t i = (size_t)px + (size_t)py * P.hiW;
            if (O.prim_id) O.prim_id[i] = g_prim;
            if (O.sub_id) O.sub_id[i] = g_sub;
            if (O.hit_t) O.hit_t[i] = g_depth;
            if (O.rng_state) O.rng_state[i] = rng;
        }
    }
    if (COUNT && O.counters) {
        // one atomic per wave per counter: butterfly-reduce across the 64 lanes first
        unsigned v[5] = {w.rays, w.box, w.tri, w.prim, w.vox};
        for (int c = 0; c < 5; c++) {
            unsigned long long x = v[c];
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
            if (lane == 0 && x) atomicAdd(O.counters + c, x);
        }
    }
}

// ------------------------------------------------------------------ K_taa
__device__ __forceinline__ float luma(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

// TemporalBlendWithClamp, RaytraceRenderer.cs:274-398.  One thread per pixel; the history and
// guide updates touch only the thread's own pixel, so the serial loops of the C# fuse into one pass.
__global__ __launch_bounds__(256) void k_taa(const TaaParams T, const float *__restrict__ current, const float *__restrict__ normal,
                                             const float *__restrict__ depth, const uint8_t *__restrict__ sky, float *__restrict__ hist,
                                             float *__restrict__ prev_normal, float *__restrict__ prev_depth, uint8_t *__restrict__ prev_sky)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= T.w || y >= T.h) return;
    const size_t i = (size_t)x + (size_t)y * T.w;
    const float cr = current[3 * i], cg = current[3 * i + 1], cb = current[3 * i + 2];
    const float nx = normal[3 * i], ny = normal[3 * i + 1], nz = normal[3 * i + 2];
    const float z_now = depth[i];
    const uint8_t sky_now = sky[i];
    if (T.reset) {
        hist[3 * i] = cr; hist[3 * i + 1] = cg; hist[3 * i + 2] = cb;
        prev_normal[3 * i] = nx; prev_normal[3 * i + 1] = ny; prev_normal[3 * i + 2] = nz;
        prev_depth[i] = z_now;
        prev_sky[i] = sky_now;
        return;
    }
    float pr = hist[3 * i], pg = hist[3 * i + 1], pb = hist[3 * i + 2];
    float local_alpha = T.alpha;
    if ((sky_now != 0) != (prev_sky[i] != 0)) {
        local_alpha = 1.0f;
    } else {
        const float z_prev = prev_depth[i];
        F3 n_now = normalized(f3(nx, ny, nz));
        F3 n_prev = normalized(f3(prev_normal[3 * i], prev_normal[3 * i + 1], prev_normal[3 * i + 2]));
        if (!cs_isfinite(z_now) || !cs_isfinite(z_prev)) {
            local_alpha = 1.0f;
        } else {
            float dz = cs_abs(z_now - z_prev);
            float rel = dz / cs_max(1e-4f, cs_min(z_now, z_prev));
            float ndot = dot(n_now, n_prev);
            if (rel > 0.05f || ndot < 0.8f) local_alpha = 1.0f;
        }
    }
    float min_l = YCGE_INF, max_l = -YCGE_INF;
    const int r = T.radius;
    for (int oy = -r; oy <= r; oy++) {
        int sy = y + oy; if (sy < 0) sy = 0; else if (sy >= T.h) sy = T.h - 1;
        for (int ox = -r; ox <= r; ox++) {
            int sx = x + ox; if (sx < 0) sx = 0; else if (sx >= T.w) sx = T.w - 1;
            const size_t j = (size_t)sx + (size_t)sy * T.w;
            if (sky[j] != sky_now) continue;
            float l = luma(current[3 * j], current[3 * j + 1], current[3 * j + 2]);
            if (l < min_l) min_l = l;
            if (l > max_l) max_l = l;
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
    hist[3 * i] = pr * (1.0f - local_alpha) + cr * local_alpha;
    hist[3 * i + 1] = pg * (1.0f - local_alpha) + cg * local_alpha;
    hist[3 * i + 2] = pb * (1.0f - local_alpha) + cb * local_alpha;
    prev_normal[3 * i] = nx; prev_normal[3 * i + 1] = ny; prev_normal[3 * i + 2] = nz;
    prev_depth[i] = z_now;
    prev_sky[i] = sky_now;
}

// ------------------------------------------------------------------ K_unpermute (multi-GPU)
// all_slabs: world_size equal-sized slabs, rank-major, as an all-gather leaves them.
__global__ __launch_bounds__(256) void k_unpermute(const float *__restrict__ all_slabs, size_t slab_floats_per_rank, int hiW, int hiH,
                                                   int tiles_x, int n_tiles, int world_size, float *__restrict__ hdr,
                                                   float *__restrict__ albedo, float *__restrict__ normal, float *__restrict__ depth,
                                                   uint8_t *__restrict__ sky)
{
    const int tile_id = blockIdx.x;
    if (tile_id >= n_tiles) return;
    const int rank = tile_id % world_size, k = tile_id / world_size;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int px = (tile_id % tiles_x) * YCGE_TILE_W + lx, py = (tile_id / tiles_x) * YCGE_TILE_H + ly;
    if (px >= hiW || py >= hiH) return;
    const float *s = all_slabs + (size_t)rank * slab_floats_per_rank + ((size_t)k * 256 + (size_t)(ly * YCGE_TILE_W + lx)) * YCGE_SLAB_FLOATS;
    const size_t i = (size_t)px + (size_t)py * hiW;
    hdr[3 * i] = s[0]; hdr[3 * i + 1] = s[1]; hdr[3 * i + 2] = s[2];
    albedo[3 * i] = s[3]; albedo[3 * i + 1] = s[4]; albedo[3 * i + 2] = s[5];
    normal[3 * i] = s[6]; normal[3 * i + 1] = s[7]; normal[3 * i + 2] = s[8];
    depth[i] = s[9];
    sky[i] = s[10] != 0.0f ? 1 : 0;
}

} // namespace ycge

// ------------------------------------------------------------------ host-callable launchers
extern "C" {

int ycge_launch_trace(const ycge::SceneDev *S, const ycge::FrameParams *P, const ycge::TraceOut *O, int count, int debug, int slab,
                      hipStream_t stream)
{
    using namespace ycge;
    if (P->n_owned_tiles <= 0) return 0;
    dim3 grid((unsigned)P->n_owned_tiles), block(256);
#define YCGE_LAUNCH(C, D, SL) hipLaunchKernelGGL((k_trace<C, D, SL>), grid, block, 0, stream, *S, *P, *O)
    if (slab) {
        if (count) YCGE_LAUNCH(true, false, true); else YCGE_LAUNCH(false, false, true);
    } else if (debug) {
        if (count) YCGE_LAUNCH(true, true, false); else YCGE_LAUNCH(false, true, false);
    } else {
        if (count) YCGE_LAUNCH(true, false, false); else YCGE_LAUNCH(false, false, false);
    }
#undef YCGE_LAUNCH
    return (int)hipGetLastError();
}

int ycge_launch_taa(const ycge::TaaParams *T, const float *current, const float *normal, const float *depth, const uint8_t *sky,
                    float *hist, float *prev_normal, float *prev_depth, uint8_t *prev_sky, hipStream_t stream)
{
    dim3 grid((unsigned)((T->w + 31) / 32), (unsigned)((T->h + 7) / 8)), block(256);
    hipLaunchKernelGGL(ycge::k_taa, grid, block, 0, stream, *T, current, normal, depth, sky, hist, prev_normal, prev_depth, prev_sky);
    return (int)hipGetLastError();
}

int ycge_launch_unpermute(const float *all_slabs, size_t slab_floats_per_rank, int hiW, int hiH, int tiles_x, int n_tiles, int world_size,
                          float *hdr, float *albedo, float *normal, float *depth, uint8_t *sky, hipStream_t stream)
{
    dim3 grid((unsigned)n_tiles), block(256);
    hipLaunchKernelGGL(ycge::k_unpermute, grid, block, 0, stream, all_slabs, slab_floats_per_rank, hiW, hiH, tiles_x, n_tiles, world_size,
                       hdr, albedo, normal, depth, sky);
    return (int)hipGetLastError();
}

} // extern "C"
