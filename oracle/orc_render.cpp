/*
 * orc_render.cpp — ORACLE: the frame pipeline of RaytraceRenderer.TryFlipAndBlit
 * (ray-gen, TraceFull, TAA, À-trous, exposure, tonemap/downsample) and the C
 * entry points the tests bind with ctypes.  Paths relative to
 * /root/reference/ConsoleGame/.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
 */
#include "orc_scene.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace orc {

/* ---- RaytraceSampler.cs -------------------------------------------------- */
static const uint8_t kBlueNoise8x8[8][8] = {   /* RaytraceSampler.cs:9-19 */
    {0, 32, 8, 40, 2, 34, 10, 42},  {48, 16, 56, 24, 50, 18, 58, 26},
    {12, 44, 4, 36, 14, 46, 6, 38}, {60, 28, 52, 20, 62, 30, 54, 22},
    {3, 35, 11, 43, 1, 33, 9, 41},  {51, 19, 59, 27, 49, 17, 57, 25},
    {15, 47, 7, 39, 13, 45, 5, 37}, {63, 31, 55, 23, 61, 29, 53, 21}};

static inline float frac(float v) { return v - cs_floor(v); }          /* RaytraceSampler.cs:22-25 */

static inline float blue_noise_sample(int x, int y, int frame_idx, int channel)   /* RaytraceSampler.cs:27-34 */
{
    int ix = x & 7, iy = y & 7;
    float base = ((float)kBlueNoise8x8[iy][ix] + 0.5f) * (1.0f / 64.0f);
    float rot = frac((float)(frame_idx + 1) * (channel == 0 ? 0.7548776662466927f : 0.5698402909980532f));
    return frac(base + rot);
}

static inline uint64_t splitmix64(uint64_t z)                          /* RaytraceSampler.cs:71-80 */
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t per_frame_seed(int x, int y, int64_t frame, int jx, int jy, uint64_t salt)   /* RaytraceSampler.cs:56-68 */
{
    uint64_t h = 1469598103934665603ULL;
    h ^= (uint64_t)(int64_t)x * 0x9E3779B97F4A7C15ULL; h = splitmix64(h);
    h ^= (uint64_t)(int64_t)y * 0xC2B2AE3D27D4EB4FULL; h = splitmix64(h);
    h ^= (uint64_t)frame * 0x165667B19E3779F9ULL; h = splitmix64(h);
    h ^= ((uint64_t)(uint8_t)jx << 8) ^ (uint64_t)(uint8_t)jy; h = splitmix64(h);
    h ^= salt; h = splitmix64(h);
    return h;
}
struct Rng {                                                            /* RaytraceSampler.cs:36-53 */
    uint64_t state;
    explicit Rng(uint64_t seed) : state(seed != 0 ? seed : 0x9E3779B97F4A7C15ULL) {}
    float next_unit()
    {
        state = splitmix64(state);
        uint32_t m24 = (uint32_t)(state >> 40);
        return ((float)m24 + 0.5f) * (1.0f / 16777216.0f);
    }
};

static V3 cosine_sample_hemisphere(V3 n, Rng &rng)                      /* RaytraceSampler.cs:83-111 */
{
    float u1 = rng.next_unit();
    float u2 = rng.next_unit();
    float r = cs_sqrt(u1);
    float phi = 6.2831853071795864769f * u2;
    float sn, cs;
    m_sincos(phi, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = cs_sqrt(1.0f - u1);
    V3 w = n;
    float wz = w.z;
    if (wz < -0.999999f) {
        V3 u = v3(0.0f, -1.0f, 0.0f);
        V3 v = v3(-1.0f, 0.0f, 0.0f);
        return u * x + v * y + w * z;
    }
    float a = 1.0f / (1.0f + wz);
    float b = (-w.x * w.y) * a;
    /* new Vec3(double,double,double): 1.0 - (w.X*w.X)*a is evaluated in binary64 then narrowed */
    V3 u_axis = v3((float)(1.0 - (double)((w.x * w.x) * a)), b, -w.x);
    V3 v_axis = v3(b, (float)(1.0 - (double)((w.y * w.y) * a)), -w.y);
    return u_axis * x + v_axis * y + w * z;
}

/* ---- shading helpers, RaytraceRenderer.cs:737-831 ------------------------ */
static const float kPi = 3.14159265358979323846f;
static const float kInvPi = 1.0f / kPi;

static inline V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }                /* :800-803 */
static inline V3 lerp(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }              /* :805-808 */
static bool refract(V3 v, V3 n, float eta, V3 &out)                                         /* :737-748 */
{
    float cosi = -cs_max(-1.0f, cs_min(1.0f, dot(v, n)));
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) { out = v3(0, 0, 0); return false; }
    out = (v * eta) + (n * (eta * cosi - cs_sqrt(k)));
    return true;
}
static float fresnel_schlick(float cos_theta, float eta_i, float eta_t)                    /* :750-755 */
{
    float r0 = (eta_i - eta_t) / (eta_i + eta_t);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * m_pow5(1.0f - cos_theta);
}
static V3 oren_nayar(V3 albedo, V3 n, V3 wo, V3 wi, float sigma_rad)                       /* :810-831 */
{
    float cos_i = cs_max(0.0f, dot(n, wi));
    float cos_o = cs_max(0.0f, dot(n, wo));
    if (cos_i <= 0.0f || cos_o <= 0.0f) return v3(0, 0, 0);
    float sin_i = cs_sqrt(cs_max(0.0f, 1.0f - cos_i * cos_i));
    float sin_o = cs_sqrt(cs_max(0.0f, 1.0f - cos_o * cos_o));
    V3 proj_i = normalized(wi - n * cos_i);
    V3 proj_o = normalized(wo - n * cos_o);
    float cos_phi = cs_max(0.0f, dot(proj_i, proj_o));
    float sigma2 = sigma_rad * sigma_rad;
    float A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    float B = 0.45f * sigma2 / (sigma2 + 0.09f);
    float sin_alpha = cs_max(sin_i, sin_o);
    float tan_beta = cs_min(sin_i / cs_max(1e-6f, cos_i), sin_o / cs_max(1e-6f, cos_o));
    float on = (A + B * cos_phi * sin_alpha * tan_beta);
    V3 f = albedo * (on * kInvPi);
    return saturate(f);
}

struct Consts {
    int max_mirror_bounces = 2, max_refractions = 2, diffuse_bounces = 1;
    float mirror_threshold = 0.9f, eps = 1e-4f, sigma_deg = 25.0f;
};

static V3 transmittance_to_light(const SceneData &S, const Consts &K, const Ray &shadow, float max_dist, Counters &cnt)   /* :757-798 */
{
    if (S.is_volume_scene) {
        bool blocked = S.occluded(shadow, max_dist, cnt);
        return blocked ? v3(0, 0, 0) : v3(1.0f, 1.0f, 1.0f);
    }
    float tr_r = 1.0f, tr_g = 1.0f, tr_b = 1.0f;
    Hit block{};
    float tmin = 0.0f + K.eps;
    int counter = 0;
    const float cutoff = 1e-6f;
    while (counter < K.max_refractions && S.hit(shadow, tmin, max_dist, block, cnt)) {
        counter++;
        float tr = block.m.transparency;
        if (tr <= 0.0f) return v3(0, 0, 0);
        V3 tint = block.m.trans_color;
        tr_r *= tint.x * tr;
        tr_g *= tint.y * tr;
        tr_b *= tint.z * tr;
        if (tr_r <= cutoff && tr_g <= cutoff && tr_b <= cutoff) return v3(0, 0, 0);
        float t_hit = block.t;
        if (t_hit > max_dist) break;
        tmin = t_hit + K.eps;
    }
    return v3(tr_r, tr_g, tr_b);
}

struct GBuf { V3 albedo, normal; float depth; int32_t prim, sub; float t; };

/* ---- TraceFull, RaytraceRenderer.cs:448-620 ------------------------------ */
static V3 trace_full(const SceneData &S, const Consts &K, Ray r, Rng &rng, bool &is_sky, GBuf &primary, Counters &cnt)
{
    struct Item { Ray ray; V3 beta; int mirror_depth, diffuse_depth; bool is_primary; };
    const int kMaxStack = 16;
    Item stack[kMaxStack];
    int sp = 0;
    stack[sp++] = Item{r, v3(1, 1, 1), 0, 0, true};
    V3 radiance = v3(0, 0, 0);
    bool primary_hit_something = false;
    is_sky = false;
    bool gbuf_valid = false;
    primary = GBuf{v3(0, 0, 0), v3(0, 0, 0), kFloatMax, -1, 0, kFloatMax};
    float sigma_rad = K.sigma_deg * (kPi / 180.0f);
    while (sp > 0) {
        sp--;
        Item item = stack[sp];
        Ray cur = item.ray;
        V3 beta = item.beta;
        int mirror_depth = item.mirror_depth;
        int diffuse_depth = item.diffuse_depth;
        for (;;) {
            Hit rec{};
            if (!S.hit(cur, 0.001f, kFloatMax, rec, cnt)) {
                float tbg = 0.5f * (cur.d.y + 1.0f);
                V3 sky = lerp(S.bg_bottom, S.bg_top, tbg);
                if (item.is_primary && !primary_hit_something) {
                    is_sky = true;
                    if (!gbuf_valid) {
                        primary = GBuf{v3(0, 0, 0), v3(0, 0, 0), kFloatMax, -1, 0, kFloatMax};
                        gbuf_valid = true;
                    }
                }
                radiance = radiance + v3(beta.x * sky.x, beta.y * sky.y, beta.z * sky.z);
                break;
            }
            if (item.is_primary) {
                primary_hit_something = true;
                is_sky = false;
                if (!gbuf_valid) {
                    primary = GBuf{S.sample_albedo(rec.m, rec.u, rec.v), rec.n, rec.t, rec.prim, rec.sub, rec.t};     /* :494-495 */
                    gbuf_valid = true;
                }
                item.is_primary = false;
            }
            if (rec.m.emission.x != 0.0f || rec.m.emission.y != 0.0f || rec.m.emission.z != 0.0f) {
                V3 e = rec.m.emission;
                radiance = radiance + v3(beta.x * e.x, beta.y * e.y, beta.z * e.z);
            }
            V3 base_albedo = S.sample_albedo(rec.m, rec.u, rec.v);     /* :505, SampleAlbedo :724-735 */
            if (rec.m.transparency > 0.0f) {
                if (mirror_depth >= K.max_mirror_bounces) break;
                V3 n = rec.n;
                V3 wo = cur.d;
                bool front = dot(n, wo) < 0.0f;
                V3 nl = front ? n : n * -1.0f;
                float eta_i = front ? 1.0f : rec.m.ior;
                float eta_t = front ? rec.m.ior : 1.0f;
                float eta = eta_i / eta_t;
                V3 refl_dir = normalized(reflect(wo, nl));
                V3 refr_dir;
                bool has_refract = refract(wo, nl, eta, refr_dir);
                float cos_theta = cs_abs(dot(nl, wo * -1.0f));
                float fres = fresnel_schlick(cos_theta, eta_i, eta_t);
                float R = fres;
                float Tr = cs_clamp(rec.m.transparency, 0.0f, 1.0f);
                float T = has_refract ? (1.0f - R) * Tr : 0.0f;
                R = cs_clamp(R + rec.m.reflectivity * (1.0f - R), 0.0f, 1.0f);
                if (R > 0.0f) {
                    if (sp < kMaxStack) {
                        Item it;
                        it.ray = make_ray(rec.p + nl * K.eps, refl_dir);
                        it.beta = v3(beta.x * base_albedo.x * R, beta.y * base_albedo.y * R, beta.z * base_albedo.z * R);
                        it.mirror_depth = mirror_depth + 1;
                        it.diffuse_depth = diffuse_depth;
                        it.is_primary = false;
                        stack[sp++] = it;
                    }
                }
                if (T > 0.0f) {
                    if (sp < kMaxStack) {
                        Item it;
                        it.ray = make_ray(rec.p - nl * K.eps, normalized(refr_dir));
                        V3 tint = rec.m.trans_color;
                        it.beta = v3(beta.x * tint.x * T, beta.y * tint.y * T, beta.z * tint.z * T);
                        it.mirror_depth = mirror_depth + 1;
                        it.diffuse_depth = diffuse_depth;
                        it.is_primary = false;
                        stack[sp++] = it;
                    }
                }
                break;
            }
            if (rec.m.reflectivity >= K.mirror_threshold) {
                if (mirror_depth >= K.max_mirror_bounces) break;
                V3 refl_dir = normalized(reflect(cur.d, rec.n));
                cur = make_ray(rec.p + rec.n * K.eps, refl_dir);
                beta = v3(beta.x * base_albedo.x, beta.y * base_albedo.y, beta.z * base_albedo.z);
                mirror_depth++;
                continue;
            }
            if (S.ambient_intensity > 0.0f) {
                V3 a = v3(S.ambient_color.x * S.ambient_intensity, S.ambient_color.y * S.ambient_intensity, S.ambient_color.z * S.ambient_intensity);
                V3 amb = v3(a.x * base_albedo.x, a.y * base_albedo.y, a.z * base_albedo.z);
                radiance = radiance + v3(beta.x * amb.x, beta.y * amb.y, beta.z * amb.z);
            }
            V3 wo_view = normalized(cur.d * -1.0f);
            for (size_t i = 0; i < S.lights.size(); i++) {
                const Light &light = S.lights[i];
                V3 to_l = light.pos - rec.p;
                float dist2 = dot(to_l, to_l);
                float dist = cs_sqrt(dist2);
                V3 ldir = to_l / dist;
                float n_dot_l = cs_max(0.0f, dot(rec.n, ldir));
                if (n_dot_l <= 0.0f) continue;
                Ray shadow = make_ray(rec.p + rec.n * K.eps, ldir);
                /* analysis counter only (the oracle traces every such ray, as the reference does): shadow rays towards lights of zero
                 * intensity - the product's timed kernels skip them, their contribution being a zero whatever the ray finds */
                if (light.intensity == 0.0f && std::isfinite(light.color.x) && std::isfinite(light.color.y) && std::isfinite(light.color.z) && dist2 > 0.0f) cnt.dark++;
                V3 trans = transmittance_to_light(S, K, shadow, dist - K.eps, cnt);
                if (trans.x <= 1e-6f && trans.y <= 1e-6f && trans.z <= 1e-6f) continue;
                float atten = light.intensity / dist2;
                V3 f_diffuse = oren_nayar(base_albedo, rec.n, wo_view, ldir, sigma_rad);
                V3 Li = light.color * atten;
                V3 contrib = (f_diffuse * n_dot_l) * Li;
                contrib = v3(contrib.x * trans.x, contrib.y * trans.y, contrib.z * trans.z);
                radiance = radiance + v3(beta.x * contrib.x, beta.y * contrib.y, beta.z * contrib.z);
            }
            if (diffuse_depth < K.diffuse_bounces) {
                V3 bounce = cosine_sample_hemisphere(rec.n, rng);
                V3 f_on = oren_nayar(base_albedo, rec.n, wo_view, bounce, sigma_rad);
                float factor = kPi;
                V3 mult = v3(f_on.x * factor, f_on.y * factor, f_on.z * factor);
                cur = make_ray(rec.p + rec.n * K.eps, bounce);
                beta = v3(beta.x * mult.x, beta.y * mult.y, beta.z * mult.z);
                diffuse_depth++;
                continue;
            }
            break;
        }
    }
    return radiance;
}

/* ---- camera basis: ForwardFromYawPitch + MakeJitteredRay invariants ------ */
struct CamBasis { V3 pos, fwd, right, up; float half_w, half_h; };
static CamBasis make_basis(V3 pos, float yaw, float pitch, float fov_deg, float aspect)   /* :413-417, 428-434 */
{
    CamBasis b;
    b.pos = pos;
    float cp = std::cos(pitch);
    V3 f = v3(std::sin(yaw) * cp, std::sin(pitch), -std::cos(yaw) * cp);
    float fov_rad = fov_deg * (kPi / 180.0f);
    b.half_h = std::tan(0.5f * fov_rad);
    b.half_w = b.half_h * aspect;
    b.fwd = normalized(f);
    b.right = normalized(cross(b.fwd, v3(0.0f, 1.0f, 0.0f)));
    b.up = normalized(cross(b.right, b.fwd));
    return b;
}
static Ray make_jittered_ray(const CamBasis &b, int px, int py, int W, int H, float rot_x, float rot_y, int frame_idx)   /* :419-437 */
{
    float jx_base = blue_noise_sample(px, py, frame_idx, 0);
    float jy_base = blue_noise_sample(px, py, frame_idx, 1);
    float jx = frac(jx_base + rot_x) - 0.5f;
    float jy = frac(jy_base + rot_y) - 0.5f;
    float u = (((float)px + 0.5f + jx) / (float)W) * 2.0f - 1.0f;
    float v = 1.0f - (((float)py + 0.5f + jy) / (float)H) * 2.0f;
    V3 dir = normalized(b.fwd + b.right * (u * b.half_w) + b.up * (v * b.half_h));
    return make_ray(b.pos, dir);
}

static inline float luma(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }   /* :269-272 */

/* ---- ToneMapper.cs ------------------------------------------------------- */
struct ToneMapper {
    float tone_exposure = 1.0f, tone_gamma = 2.2f;
    float ae_key = 0.18f, ae_speed = 0.2f, ae_exposure = 1.0f, ae_min = 0.10f, ae_max = 1.50f;
    float effective = 1.0f;
    float saturation = 2.0f, vibrance = 0.0f;

    void update_exposure(const V3 *hdr, const uint8_t *sky, int w, int h, int sample_step)   /* ToneMapper.cs:49-91 */
    {
        int step = sample_step > 2 ? sample_step : 2;
        float log_sum = 0.0f;
        int cnt = 0;
        for (int py = 0; py < h; py += step)
            for (int px = 0; px < w; px += step) {
                if (sky[px + py * w]) continue;
                V3 c = hdr[px + py * w];
                float lum = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;
                if (lum > 0.0f) { log_sum += m_log(1e-6f + lum); cnt++; }
            }
        float avg_log = cnt > 0 ? log_sum / (float)(cnt > 1 ? cnt : 1) : 0.0f;
        float avg_lum = m_exp(avg_log);
        float target = cnt > 0 ? ae_key / cs_max(1e-6f, avg_lum) : ae_exposure;
        if (target < ae_min) target = ae_min;
        if (target > ae_max) target = ae_max;
        float s = 1.0f - m_exp(-ae_speed);
        ae_exposure = ae_exposure + (target - ae_exposure) * s;
        effective = tone_exposure * ae_exposure;
    }
    static float aces(float x)                                                              /* ToneMapper.cs:247-260 */
    {
        float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
        float num = x * (a * x + b);
        float den = x * (c * x + d) + e;
        float y = den > 0.0f ? num / den : 0.0f;
        if (y < 0.0f) y = 0.0f;
        if (y > 1.0f) y = 1.0f;
        return y;
    }
    V3 map_pixel(V3 hdr) const                                                              /* ToneMapper.cs:204-238 */
    {
        float r = cs_max(0.0f, hdr.x) * effective;
        float g = cs_max(0.0f, hdr.y) * effective;
        float b = cs_max(0.0f, hdr.z) * effective;
        r = aces(r); g = aces(g); b = aces(b);
        float inv_gamma = 1.0f / cs_max(0.1f, tone_gamma);
        float sr = m_pow(clamp01(r), inv_gamma);
        float sg = m_pow(clamp01(g), inv_gamma);
        float sb = m_pow(clamp01(b), inv_gamma);
        r = clamp01(sr); g = clamp01(sg); b = clamp01(sb);
        float y = 0.2126f * r + 0.7152f * g + 0.0722f * b;
        float maxc = cs_max(r, cs_max(g, b));
        float minc = cs_min(r, cs_min(g, b));
        float chroma = maxc - minc;
        float vib = 1.0f + vibrance * (1.0f - chroma);
        float f = saturation * vib;
        float rr = y + (r - y) * f, gg = y + (g - y) * f, bb = y + (b - y) * f;
        return v3(clamp01(rr), clamp01(gg), clamp01(bb));
    }
};

/* ---- the renderer -------------------------------------------------------- */
struct Renderer {
    ycge_config cfg;
    Consts K;
    SceneData scene;
    bool have_scene = false;
    int fbW = 0, fbH = 0, ss = 1, hiW = 0, hiH = 0;
    int64_t frame_counter = 0;
    V3 cam_pos = v3(0.0f, 1.0f, 0.0f);
    float yaw = 0.0f, pitch = 0.0f, fov_deg = 45.0f;
    /* TemporalAA camera memory, TemporalAA.cs:11-15 */
    float last_x = NAN, last_y = NAN, last_z = NAN, last_yaw = NAN, last_pitch = NAN;
    bool taa_valid = false;
    ToneMapper tone;
    std::vector<Ray> rays;
    std::vector<V3> current, g_albedo, g_normal, taa_hist, prev_normal, spatial_a, spatial_b, denoised;
    std::vector<float> g_depth, prev_depth, hit_t;
    std::vector<uint8_t> sky, prev_sky;
    std::vector<int32_t> prim_id, sub_id;
    std::vector<uint64_t> rng_state;
    std::string err;

    void resize(int w, int h, int s)
    {
        fbW = w; fbH = h; ss = s < 1 ? 1 : s;
        hiW = fbW * ss; hiH = fbH * 2 * ss;
        size_t n = (size_t)hiW * hiH;
        rays.assign(n, Ray{}); current.assign(n, V3{}); g_albedo.assign(n, V3{}); g_normal.assign(n, V3{});
        taa_hist.assign(n, V3{}); prev_normal.assign(n, V3{}); spatial_a.assign(n, V3{}); spatial_b.assign(n, V3{});
        denoised.assign(n, V3{});
        g_depth.assign(n, 0.0f); prev_depth.assign(n, 0.0f); hit_t.assign(n, 0.0f);
        sky.assign(n, 0); prev_sky.assign(n, 0);
        prim_id.assign(n, -1); sub_id.assign(n, 0); rng_state.assign(n, 0);
        taa_valid = false;
        last_x = last_y = last_z = last_yaw = last_pitch = NAN;   /* TemporalAA.Resize :34-46 */
    }

    bool should_reset_history() const                                   /* TemporalAA.cs:58-67 */
    {
        float dx = cam_pos.x - last_x, dy = cam_pos.y - last_y, dz = cam_pos.z - last_z;
        float trans = (dx != dx) ? 0.0f : cs_sqrt(dx * dx + dy * dy + dz * dz);
        float dyaw = (last_yaw != last_yaw) ? 0.0f : cs_abs(yaw - last_yaw);
        float dpitch = (last_pitch != last_pitch) ? 0.0f : cs_abs(pitch - last_pitch);
        return trans > cfg.motion_trans_reset || dyaw > cfg.motion_rot_reset || dpitch > cfg.motion_rot_reset;
    }

    /* taa_threads: 1 = the reference's serial loops; n > 1 = the same per-pixel work in n row bands (every pixel reads its
     * own history and the CURRENT frame of its neighbours only, so bands are independent and the result is identical) - the
     * "also reported parallel" CPU figure of SURVEY 8(d), not something the reference does. */
    int taa_threads = 1;
    bool temporal_blend(bool force_reset)                               /* RaytraceRenderer.cs:274-398 */
    {
        int w = hiW, h = hiH;
        if (!taa_valid || force_reset) {
            for (size_t i = 0; i < (size_t)w * h; i++) {
                taa_hist[i] = current[i]; prev_normal[i] = g_normal[i]; prev_depth[i] = g_depth[i]; prev_sky[i] = sky[i];
            }
            taa_valid = true;
            return true;
        }
        float alpha = cs_max(0.0f, cs_min(1.0f, cfg.taa_alpha));
        int r = cfg.taa_clamp_radius > 0 ? cfg.taa_clamp_radius : 0;
        auto rows = [&](int y_begin, int y_end) {
        for (int y = y_begin; y < y_end; y++)
            for (int x = 0; x < w; x++) {
                size_t i = (size_t)x + (size_t)y * w;
                V3 cur = current[i];
                V3 prev = taa_hist[i];
                bool sky_now = sky[i] != 0, sky_prev = prev_sky[i] != 0;
                float local_alpha = alpha;
                if (sky_now != sky_prev) {
                    local_alpha = 1.0f;
                } else {
                    float z_now = g_depth[i], z_prev = prev_depth[i];
                    V3 n_now = normalized(g_normal[i]);
                    V3 n_prev = normalized(prev_normal[i]);
                    if (!cs_isfinite(z_now) || !cs_isfinite(z_prev)) {
                        local_alpha = 1.0f;
                    } else {
                        float dz = cs_abs(z_now - z_prev);
                        float rel = dz / cs_max(1e-4f, cs_min(z_now, z_prev));
                        float ndot = dot(n_now, n_prev);
                        if (rel > 0.05f || ndot < 0.8f) local_alpha = 1.0f;
                    }
                }
                float min_l = kInf, max_l = -kInf;
                for (int oy = -r; oy <= r; oy++) {
                    int sy = y + oy; if (sy < 0) sy = 0; else if (sy >= h) sy = h - 1;
                    for (int ox = -r; ox <= r; ox++) {
                        int sx = x + ox; if (sx < 0) sx = 0; else if (sx >= w) sx = w - 1;
                        size_t j = (size_t)sx + (size_t)sy * w;
                        if (sky[j] != sky[i]) continue;
                        float l = luma(current[j]);
                        if (l < min_l) min_l = l;
                        if (l > max_l) max_l = l;
                    }
                }
                float pad = cfg.taa_luminance_pad;
                float range = max_l - min_l;
                float l_min = min_l - range * pad;
                float l_max = max_l + range * pad;
                float prev_l = luma(prev);
                if (prev_l > l_max) {
                    float s = l_max / cs_max(1e-6f, prev_l);
                    prev = v3(prev.x * s, prev.y * s, prev.z * s);
                } else if (prev_l < l_min) {
                    float s = l_min / cs_max(1e-6f, prev_l);
                    prev = v3(prev.x * s, prev.y * s, prev.z * s);
                }
                taa_hist[i] = v3(prev.x * (1.0f - local_alpha) + cur.x * local_alpha,
                                 prev.y * (1.0f - local_alpha) + cur.y * local_alpha,
                                 prev.z * (1.0f - local_alpha) + cur.z * local_alpha);
            }
        };
        if (taa_threads <= 1) rows(0, h);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < taa_threads; k++) th.emplace_back(rows, (int)((long long)k * h / taa_threads), (int)((long long)(k + 1) * h / taa_threads));
            for (auto &t : th) t.join();
        }
        for (size_t i = 0; i < (size_t)w * h; i++) { prev_normal[i] = g_normal[i]; prev_depth[i] = g_depth[i]; prev_sky[i] = sky[i]; }
        return false;
    }

    /* ApplyAtrousDenoise, RaytraceRenderer.cs:622-722.  Buffers ping-pong exactly as
     * the C# swap at :718 does: iter0 src->A, iter1 A->A (in place, scan order!), iter2 A->B. */
    const V3 *atrous(const V3 *src)
    {
        int w = hiW, h = hiH;
        const float k[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
        const V3 *cur = src;
        V3 *scratch_a = spatial_a.data(), *scratch_b = spatial_b.data();
        V3 *dst = scratch_a;
        int iters = cfg.atrous_iterations > 1 ? cfg.atrous_iterations : 1;
        float c_phi = cfg.atrous_c_phi, n_phi = cfg.atrous_n_phi, z_phi = cfg.atrous_z_phi, a_phi = cfg.atrous_a_phi;
        for (int it = 0; it < iters; it++) {
            int step = 1 << it;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    size_t i = (size_t)x + (size_t)y * w;
                    if (sky[i]) { dst[i] = cur[i]; continue; }
                    V3 c0 = cur[i];
                    V3 a0 = g_albedo[i];
                    V3 n0 = normalized(g_normal[i]);
                    float z0 = g_depth[i];
                    float wsum = 0.0f;
                    V3 accum = v3(0, 0, 0);
                    for (int ky = -2; ky <= 2; ky++) {
                        int sy = y + ky * step;
                        if (sy < 0) sy = 0; else if (sy >= h) sy = h - 1;
                        float wy = k[ky + 2];
                        for (int kx = -2; kx <= 2; kx++) {
                            int sx = x + kx * step;
                            if (sx < 0) sx = 0; else if (sx >= w) sx = w - 1;
                            size_t j = (size_t)sx + (size_t)sy * w;
                            if (sky[j] != sky[i]) continue;
                            float wx = k[kx + 2];
                            float w_base = wx * wy;
                            V3 c = cur[j];
                            V3 a = g_albedo[j];
                            V3 n = normalized(g_normal[j]);
                            float z = g_depth[j];
                            float lum0 = 0.2126f * c0.x + 0.7152f * c0.y + 0.0722f * c0.z;
                            float lum = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;
                            float dl = cs_abs(lum - lum0);
                            float dn = cs_max(0.0f, 1.0f - dot(n0, n));
                            float dz = cs_abs(z - z0);
                            float da = cs_abs(a.x - a0.x) + cs_abs(a.y - a0.y) + cs_abs(a.z - a0.z);
                            float wc = m_exp(-dl / cs_max(1e-6f, c_phi));
                            float wn = m_exp(-dn / cs_max(1e-6f, n_phi));
                            float wz = m_exp(-dz / cs_max(1e-6f, z_phi));
                            float wa = m_exp(-(da) / cs_max(1e-6f, a_phi));
                            float wght = w_base * wc * wn * wz * wa;
                            accum = v3(accum.x + c.x * wght, accum.y + c.y * wght, accum.z + c.z * wght);
                            wsum += wght;
                        }
                    }
                    if (wsum > 1e-8f) {
                        float inv = 1.0f / wsum;
                        dst[i] = v3(accum.x * inv, accum.y * inv, accum.z * inv);
                    } else {
                        dst[i] = c0;
                    }
                }
            /* :718.  config.atrous_inplace_exact = 0 (include/ycge.h) is the WAIVED form: plain ping-pong, no iteration in place */
            const V3 *tmp = cur; cur = dst;
            dst = cfg.atrous_inplace_exact ? ((tmp == scratch_a) ? scratch_b : scratch_a) : ((cur == scratch_a) ? scratch_b : scratch_a);
        }
        return cur;
    }

    /* TryFlipAndBlit, RaytraceRenderer.cs:157-267 */
    int render(float *out_sdr, ycge_frame_stats *st, int threads, int stages)
    {
        if (!have_scene) { err = "Scene BVH not built; call RebuildBVH() after populating Objects."; return YCGE_ERR_NO_SCENE; }
        auto t0 = std::chrono::steady_clock::now();
        float aspect = (float)hiW / (float)hiH;
        bool reset = should_reset_history() || scene.has_dynamic_textures;      /* RaytraceRenderer.cs:171 */
        int64_t frame = ++frame_counter;
        int frame_idx = (int)(frame & 0x7fffffff);
        float rot_x = frac((float)(frame_idx + 1) * 0.61803398875f);
        float rot_y = frac((float)(frame_idx + 1) * 0.38196601125f);
        CamBasis basis = make_basis(cam_pos, yaw, pitch, fov_deg, aspect);
        if (threads < 1) threads = 1;

        /* step 3: ray-gen in row bands (FixedThreadFor) */
        auto raygen = [&](int worker) {
            int y0 = worker * hiH / threads, y1 = (worker + 1) * hiH / threads;
            for (int py = y0; py < y1; py++)
                for (int px = 0; px < hiW; px++)
                    rays[(size_t)px + (size_t)py * hiW] = make_jittered_ray(basis, px, py, hiW, hiH, rot_x, rot_y, frame_idx);
        };
        /* step 4: per-pixel trace, pixels interleaved across workers (PixelThreadPool.For2D stride) */
        std::vector<Counters> counters(threads);
        size_t npx = (size_t)hiW * hiH;
        auto trace = [&](int worker) {
            Counters c;
            for (size_t i = worker; i < npx; i += threads) {
                int px = (int)(i % hiW), py = (int)(i / hiW);
                Rng rng(per_frame_seed(px, py, frame, 0, 0, cfg.seed_salt));
                bool is_sky; GBuf g;
                V3 cur = trace_full(scene, K, rays[i], rng, is_sky, g, c);
                sky[i] = is_sky ? 1 : 0;
                current[i] = cur;
                g_albedo[i] = g.albedo; g_normal[i] = g.normal; g_depth[i] = g.depth;
                prim_id[i] = g.prim; sub_id[i] = g.sub; hit_t[i] = g.t;
                rng_state[i] = rng.state;
            }
            counters[worker] = c;
        };
        auto run = [&](auto &fn) {
            if (threads == 1) { fn(0); return; }
            std::vector<std::thread> th;
            for (int w = 0; w < threads; w++) th.emplace_back(fn, w);
            for (auto &t : th) t.join();
        };
        run(raygen);
        run(trace);
        auto t1 = std::chrono::steady_clock::now();
        bool did_reset = false;
        auto t2 = t1, t3 = t1;
        if (stages >= 1) {
            did_reset = temporal_blend(reset);
            t2 = std::chrono::steady_clock::now();
            t3 = t2;
        }
        if (stages >= 2) {
            const V3 *den = atrous(taa_hist.data());
            std::memcpy(denoised.data(), den, npx * sizeof(V3));
            int step = (ss * 2 > 2) ? ss * 2 : 2;
            tone.update_exposure(denoised.data(), sky.data(), hiW, hiH, step);
            /* step 8: ss x ss box average of top / bottom half-cells, :229-264 */
            if (out_sdr) {
                for (int cy = 0; cy < fbH; cy++) {
                    int y_top0 = cy * 2 * ss, y_bot0 = (cy * 2 + 1) * ss;
                    for (int cx = 0; cx < fbW; cx++) {
                        int x0 = cx * ss;
                        V3 top = v3(0, 0, 0), bot = v3(0, 0, 0);
                        for (int sy = 0; sy < ss; sy++)
                            for (int sx = 0; sx < ss; sx++) {
                                top = top + denoised[(size_t)(x0 + sx) + (size_t)(y_top0 + sy) * hiW];
                                bot = bot + denoised[(size_t)(x0 + sx) + (size_t)(y_bot0 + sy) * hiW];
                            }
                        float inv = 1.0f / (float)(ss * ss);
                        V3 t_sdr = tone.map_pixel(v3(top.x * inv, top.y * inv, top.z * inv));
                        V3 b_sdr = tone.map_pixel(v3(bot.x * inv, bot.y * inv, bot.z * inv));
                        float *o = out_sdr + ((size_t)cx + (size_t)cy * fbW) * 6;
                        o[0] = t_sdr.x; o[1] = t_sdr.y; o[2] = t_sdr.z; o[3] = b_sdr.x; o[4] = b_sdr.y; o[5] = b_sdr.z;
                    }
                }
            }
            t3 = std::chrono::steady_clock::now();
        }
        /* taa.CommitCamera :266 */
        last_x = cam_pos.x; last_y = cam_pos.y; last_z = cam_pos.z; last_yaw = yaw; last_pitch = pitch;
        if (st) {
            Counters tot;
            for (auto &c : counters) tot.add(c);
            std::memset(st, 0, sizeof(*st));
            st->frame = frame;
            st->history_reset = did_reset ? 1 : 0;
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            st->trace_ms = ms(t0, t1); st->taa_ms = ms(t1, t2); st->post_ms = ms(t2, t3); st->total_ms = ms(t0, t3);
            st->n_rays = tot.rays; st->n_box = tot.box; st->n_tri = tot.tri; st->n_prim = tot.prim; st->n_vox = tot.vox; st->n_rays_dark = tot.dark;
            st->exposure = tone.effective;
        }
        return YCGE_OK;
    }
};

} // namespace orc

/* ======================================================================
 * C entry points (ctypes).  Same shapes as include/ycge.h so the parity
 * tests drive oracle and product with identical calls.
 * ====================================================================== */
using orc::Renderer;

extern "C" {

int orc_create(const ycge_config *cfg, void **out)
{
    if (!cfg || !out) return YCGE_ERR_INVALID_ARG;
    if (cfg->fb_width <= 0 || cfg->fb_height <= 0) return YCGE_ERR_INVALID_ARG;
    Renderer *r = new Renderer();
    r->cfg = *cfg;
    r->K.max_mirror_bounces = cfg->max_mirror_bounces; r->K.max_refractions = cfg->max_refractions;
    r->K.diffuse_bounces = cfg->diffuse_bounces; r->K.mirror_threshold = cfg->mirror_threshold;
    r->K.eps = cfg->eps; r->K.sigma_deg = cfg->diffuse_sigma_deg;
    r->fov_deg = cfg->fov_deg;
    r->resize(cfg->fb_width, cfg->fb_height, cfg->super_sample);
    *out = r;
    return YCGE_OK;
}
void orc_destroy(void *ctx) { delete (Renderer *)ctx; }
const char *orc_last_error(void *ctx) { return ctx ? ((Renderer *)ctx)->err.c_str() : ""; }

int orc_scene_upload(void *ctx, const ycge_scene *s)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !s) return YCGE_ERR_INVALID_ARG;
    std::string e = r->scene.load(s);
    if (!e.empty()) { r->err = e; r->have_scene = false; return YCGE_ERR_INVALID_ARG; }
    if (r->scene.stats.max_depth > 128) { r->err = "scene BVH deeper than 128"; return YCGE_ERR_STACK_DEPTH; }
    for (auto &m : r->scene.meshes)
        if (m.stats.max_depth > 64) { r->err = "mesh BVH deeper than 64"; return YCGE_ERR_STACK_DEPTH; }
    r->have_scene = true;
    return YCGE_OK;
}
/* the frame IFrameReader.GetCurrentFramePtr() shows from now on (Texture.cs:116) */
int orc_scene_update_texture(void *ctx, int32_t texture_index, const uint8_t *frame, size_t bytes)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene || texture_index < 0 || (size_t)texture_index >= r->scene.textures.size() || !frame) return YCGE_ERR_INVALID_ARG;
    orc::Texture &t = r->scene.textures[(size_t)texture_index];
    if (t.frame_bpp == 0 || bytes != t.frame.size()) return YCGE_ERR_INVALID_ARG;
    std::memcpy(t.frame.data(), frame, bytes);
    return YCGE_OK;
}
int orc_scene_update_lights(void *ctx, const ycge_light *lights, int32_t n, const ycge_vec3 *amb, float amb_i,
                            const ycge_vec3 *top, const ycge_vec3 *bottom)
{
    Renderer *r = (Renderer *)ctx;
    if (!r) return YCGE_ERR_INVALID_ARG;
    r->scene.lights.clear();
    for (int i = 0; i < n; i++) r->scene.lights.push_back(orc::Light{orc::v3(lights[i].position), orc::v3(lights[i].color), lights[i].intensity});
    if (amb) { r->scene.ambient_color = orc::v3(*amb); r->scene.ambient_intensity = amb_i; }
    if (top) r->scene.bg_top = orc::v3(*top);
    if (bottom) r->scene.bg_bottom = orc::v3(*bottom);
    return YCGE_OK;
}
int orc_resize(void *ctx, int32_t w, int32_t h, int32_t ss)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || w <= 0 || h <= 0) return YCGE_ERR_INVALID_ARG;
    r->resize(w, h, ss);
    return YCGE_OK;
}
int orc_set_camera(void *ctx, const float pos[3], float yaw, float pitch, float fov)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !pos) return YCGE_ERR_INVALID_ARG;
    r->cam_pos = orc::v3(pos[0], pos[1], pos[2]); r->yaw = yaw; r->pitch = pitch; r->fov_deg = fov;
    return YCGE_OK;
}
/* stages: 0 = ray-gen + trace only, 1 = + TAA, 2 = + denoise/exposure/tonemap (full TryFlipAndBlit) */
int orc_render_frame(void *ctx, float *out_sdr, ycge_frame_stats *st, int threads, int stages)
{
    Renderer *r = (Renderer *)ctx;
    if (!r) return YCGE_ERR_INVALID_ARG;
    return r->render(out_sdr, st, threads, stages);
}
int orc_set_taa_threads(void *ctx, int n)
{
    Renderer *r = (Renderer *)ctx;
    if (!r) return YCGE_ERR_INVALID_ARG;
    r->taa_threads = n < 1 ? 1 : n;
    return YCGE_OK;
}
int orc_set_frame_counter(void *ctx, int64_t fc)
{
    Renderer *r = (Renderer *)ctx;
    if (!r) return YCGE_ERR_INVALID_ARG;
    r->frame_counter = fc;
    return YCGE_OK;
}
int orc_read_buffer(void *ctx, int32_t which, void *dst, size_t bytes)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !dst) return YCGE_ERR_INVALID_ARG;
    const void *src = nullptr; size_t n = 0;
    size_t npx = (size_t)r->hiW * r->hiH;
    switch (which) {
    case YCGE_BUF_RAYS: src = r->rays.data(); n = npx * 24; break;
    case YCGE_BUF_PRIM_ID: src = r->prim_id.data(); n = npx * 4; break;
    case YCGE_BUF_SUB_ID: src = r->sub_id.data(); n = npx * 4; break;
    case YCGE_BUF_HIT_T: src = r->hit_t.data(); n = npx * 4; break;
    case YCGE_BUF_CURRENT_HDR: src = r->current.data(); n = npx * 12; break;
    case YCGE_BUF_G_ALBEDO: src = r->g_albedo.data(); n = npx * 12; break;
    case YCGE_BUF_G_NORMAL: src = r->g_normal.data(); n = npx * 12; break;
    case YCGE_BUF_G_DEPTH: src = r->g_depth.data(); n = npx * 4; break;
    case YCGE_BUF_SKY_MASK: src = r->sky.data(); n = npx; break;
    case YCGE_BUF_TAA_HISTORY: src = r->taa_hist.data(); n = npx * 12; break;
    case YCGE_BUF_PREV_NORMAL: src = r->prev_normal.data(); n = npx * 12; break;
    case YCGE_BUF_PREV_DEPTH: src = r->prev_depth.data(); n = npx * 4; break;
    case YCGE_BUF_PREV_SKY: src = r->prev_sky.data(); n = npx; break;
    case YCGE_BUF_DENOISED: src = r->denoised.data(); n = npx * 12; break;
    case YCGE_BUF_RNG_STATE: src = r->rng_state.data(); n = npx * 8; break;
    default: return YCGE_ERR_INVALID_ARG;
    }
    if (bytes != n) return YCGE_ERR_INVALID_ARG;
    std::memcpy(dst, src, n);
    return YCGE_OK;
}
static int accel_view(Renderer *r, int32_t which, int32_t index, const void **p, size_t *n)
{
    switch (which) {
    case YCGE_ACCEL_SCENE_NODES: *p = r->scene.nodes.data(); *n = r->scene.nodes.size() * sizeof(orc::Node); return 0;
    case YCGE_ACCEL_SCENE_LEAF_INDEX: *p = r->scene.leaf_obj.data(); *n = r->scene.leaf_obj.size() * 4; return 0;
    case YCGE_ACCEL_MESH_NODES:
        if (index < 0 || index >= (int)r->scene.meshes.size()) return -1;
        *p = r->scene.meshes[index].nodes.data(); *n = r->scene.meshes[index].nodes.size() * sizeof(orc::Node); return 0;
    case YCGE_ACCEL_MESH_LEAF_INDEX:
        if (index < 0 || index >= (int)r->scene.meshes.size()) return -1;
        *p = r->scene.meshes[index].leaf_tri.data(); *n = r->scene.meshes[index].leaf_tri.size() * 4; return 0;
    }
    return -1;
}
int orc_accel_size(void *ctx, int32_t which, int32_t index, size_t *bytes)
{
    Renderer *r = (Renderer *)ctx; const void *p; size_t n;
    if (!r || !bytes || accel_view(r, which, index, &p, &n)) return YCGE_ERR_INVALID_ARG;
    *bytes = n; return YCGE_OK;
}
int orc_read_accel(void *ctx, int32_t which, int32_t index, void *dst, size_t bytes)
{
    Renderer *r = (Renderer *)ctx; const void *p; size_t n;
    if (!r || !dst || accel_view(r, which, index, &p, &n) || n != bytes) return YCGE_ERR_INVALID_ARG;
    std::memcpy(dst, p, n); return YCGE_OK;
}
/* builder statistics: [scene sort fallbacks, scene max depth, mesh sort fallbacks, mesh max depth] */
int orc_build_stats(void *ctx, int32_t mesh_index, int32_t out[4])
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !out) return YCGE_ERR_INVALID_ARG;
    out[0] = r->scene.stats.sort_fallbacks; out[1] = r->scene.stats.max_depth; out[2] = out[3] = 0;
    if (mesh_index >= 0 && mesh_index < (int)r->scene.meshes.size()) {
        out[2] = r->scene.meshes[mesh_index].stats.sort_fallbacks; out[3] = r->scene.meshes[mesh_index].stats.max_depth;
    }
    return YCGE_OK;
}

/* ---- scalar probes for known-answer tests ------------------------------- */
uint64_t orc_splitmix64(uint64_t z) { return orc::splitmix64(z); }
uint64_t orc_per_frame_seed(int x, int y, int64_t frame, int jx, int jy, uint64_t salt) { return orc::per_frame_seed(x, y, frame, jx, jy, salt); }
float orc_rng_next_unit(uint64_t *state) { orc::Rng r(*state); r.state = *state; float v = r.next_unit(); *state = r.state; return v; }
uint64_t orc_rng_init(uint64_t seed) { return orc::Rng(seed).state; }
float orc_blue_noise_sample(int x, int y, int frame_idx, int ch) { return orc::blue_noise_sample(x, y, frame_idx, ch); }
float orc_frac(float v) { return orc::frac(v); }
void orc_sincos(float x, float *s, float *c) { orc::m_sincos(x, s, c); }
float orc_pow5(float x) { return orc::m_pow5(x); }
float orc_exp(float x) { return orc::m_exp(x); }
float orc_log(float x) { return orc::m_log(x); }
float orc_pow(float x, float y) { return orc::m_pow(x, y); }
int32_t orc_f2i(float f) { return orc::cs_f2i(f); }
float orc_max(float a, float b) { return orc::cs_max(a, b); }
float orc_min(float a, float b) { return orc::cs_min(a, b); }
void orc_cosine_sample_hemisphere(const float n[3], uint64_t *state, float out[3])
{
    orc::Rng r(1); r.state = *state;
    orc::V3 d = orc::cosine_sample_hemisphere(orc::v3(n[0], n[1], n[2]), r);
    *state = r.state; out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void orc_oren_nayar(const float alb[3], const float n[3], const float wo[3], const float wi[3], float sigma_rad, float out[3])
{
    orc::V3 f = orc::oren_nayar(orc::v3(alb[0], alb[1], alb[2]), orc::v3(n[0], n[1], n[2]), orc::v3(wo[0], wo[1], wo[2]), orc::v3(wi[0], wi[1], wi[2]), sigma_rad);
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}
/* sort `n` keys on one axis the way .NET's Array.Sort would; keys carry their original index */
void orc_introsort(float *keys, int32_t *index, int n)
{
    std::vector<orc::BuildItem> it(n);
    for (int i = 0; i < n; i++) { it[i] = orc::BuildItem{}; it[i].index = index[i]; it[i].cx = keys[i]; }
    orc::dotnet_introsort(it.data(), n, 0);
    for (int i = 0; i < n; i++) { keys[i] = it[i].cx; index[i] = it[i].index; }
}
/* closest hit of one ray against the uploaded scene: out = {hit, prim, sub, t, px,py,pz, nx,ny,nz, albedo rgb} */
int orc_scene_hit(void *ctx, const float o[3], const float d[3], float t_min, float t_max, float out[13])
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene) return YCGE_ERR_NO_SCENE;
    orc::Counters c; orc::Hit h{};
    orc::Ray ray = orc::make_ray(orc::v3(o[0], o[1], o[2]), orc::v3(d[0], d[1], d[2]));
    bool hit = r->scene.hit(ray, t_min, t_max, h, c);
    out[0] = hit ? 1.0f : 0.0f; out[1] = (float)h.prim; out[2] = (float)h.sub; out[3] = h.t;
    out[4] = h.p.x; out[5] = h.p.y; out[6] = h.p.z; out[7] = h.n.x; out[8] = h.n.y; out[9] = h.n.z;
    out[10] = h.m.albedo.x; out[11] = h.m.albedo.y; out[12] = h.m.albedo.z;
    return YCGE_OK;
}
/* SampleAlbedo of material `mi` of the uploaded scene at n (u, v) pairs (known-answer tests of the texture branch): out = n x 3;
 * and the (u, v) the closest hit of a ray carries: hit_uv out = {hit, u, v, sampled albedo rgb} */
int orc_sample_albedo(void *ctx, int32_t mi, const float *uv, int n, float *out)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene || mi < 0 || mi >= (int)r->scene.materials.size()) return YCGE_ERR_INVALID_ARG;
    const orc::Mat m = r->scene.eval_material(mi, orc::v3(0, 0, 0));
    for (int i = 0; i < n; i++) {
        const orc::V3 a = r->scene.sample_albedo(m, uv[2 * i], uv[2 * i + 1]);
        out[3 * i] = a.x; out[3 * i + 1] = a.y; out[3 * i + 2] = a.z;
    }
    return YCGE_OK;
}
int orc_scene_hit_uv(void *ctx, const float o[3], const float d[3], float out[6])
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene) return YCGE_ERR_NO_SCENE;
    orc::Counters c; orc::Hit h{};
    orc::Ray ray = orc::make_ray(orc::v3(o[0], o[1], o[2]), orc::v3(d[0], d[1], d[2]));
    const bool hit = r->scene.hit(ray, 0.001f, 3.402823466e+38f, h, c);
    out[0] = hit ? 1.0f : 0.0f; out[1] = h.u; out[2] = h.v;
    const orc::V3 a = hit ? r->scene.sample_albedo(h.m, h.u, h.v) : orc::v3(0, 0, 0);
    out[3] = a.x; out[4] = a.y; out[5] = a.z;
    return YCGE_OK;
}
/* closest hits of n rays with per-ray work counters: counts = n x {box, tri, prim, vox} (analysis aid) */
int orc_scene_hit_many(void *ctx, const float *od /* n x 6 */, int n, float t_min, float t_max, float *t_out, int32_t *prim_out, uint32_t *counts)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene) return YCGE_ERR_NO_SCENE;
    for (int i = 0; i < n; i++) {
        orc::Counters c; orc::Hit h{};
        orc::Ray ray = orc::make_ray(orc::v3(od[6 * i], od[6 * i + 1], od[6 * i + 2]), orc::v3(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5]));
        bool hit = r->scene.hit(ray, t_min, t_max, h, c);
        if (t_out) t_out[i] = hit ? h.t : t_max;
        if (prim_out) prim_out[i] = hit ? h.prim : -1;
        if (counts) { counts[4 * i] = (uint32_t)c.box; counts[4 * i + 1] = (uint32_t)c.tri; counts[4 * i + 2] = (uint32_t)c.prim; counts[4 * i + 3] = (uint32_t)c.vox; }
    }
    return YCGE_OK;
}
/* probe for known-answer tests of steps 6-8 (denoise, exposure, downsample + tonemap) on caller-supplied buffers:
 * hdr / albedo / normal = hiW*hiH*3 f32, depth = hiW*hiH f32, sky = hiW*hiH u8; denoised_out = hiW*hiH*3,
 * sdr_out = fbW*fbH*6; exposure_io = {aeExposure before -> after, effective exposure after} */
int orc_post_probe(int fb_w, int fb_h, int ss, const float *hdr, const float *albedo, const float *normal, const float *depth,
                   const uint8_t *sky, int iterations, const float phi[4], float *denoised_out, float *exposure_io, float *sdr_out)
{
    if (fb_w <= 0 || fb_h <= 0 || !hdr || !albedo || !normal || !depth || !sky || !phi || !denoised_out || !exposure_io || !sdr_out) return YCGE_ERR_INVALID_ARG;
    Renderer r;
    ycge_config cfg{};
    cfg.atrous_iterations = iterations; cfg.atrous_c_phi = phi[0]; cfg.atrous_n_phi = phi[1]; cfg.atrous_z_phi = phi[2]; cfg.atrous_a_phi = phi[3];
    cfg.atrous_inplace_exact = 1;        /* the reference's buffer walk (RaytraceRenderer.cs:718) */
    r.cfg = cfg;
    r.resize(fb_w, fb_h, ss);
    size_t n = (size_t)r.hiW * r.hiH;
    std::memcpy(r.taa_hist.data(), hdr, n * 12); std::memcpy(r.g_albedo.data(), albedo, n * 12); std::memcpy(r.g_normal.data(), normal, n * 12);
    std::memcpy(r.g_depth.data(), depth, n * 4); std::memcpy(r.sky.data(), sky, n);
    const orc::V3 *den = r.atrous(r.taa_hist.data());
    std::memcpy(denoised_out, den, n * 12);
    r.tone.ae_exposure = exposure_io[0];
    r.tone.update_exposure(den, r.sky.data(), r.hiW, r.hiH, (r.ss * 2 > 2) ? r.ss * 2 : 2);
    exposure_io[0] = r.tone.ae_exposure; exposure_io[1] = r.tone.effective;
    for (int cy = 0; cy < r.fbH; cy++) {
        int y_top0 = cy * 2 * r.ss, y_bot0 = (cy * 2 + 1) * r.ss;
        for (int cx = 0; cx < r.fbW; cx++) {
            int x0 = cx * r.ss;
            orc::V3 top = orc::v3(0, 0, 0), bot = orc::v3(0, 0, 0);
            for (int sy = 0; sy < r.ss; sy++)
                for (int sx = 0; sx < r.ss; sx++) {
                    top = top + den[(size_t)(x0 + sx) + (size_t)(y_top0 + sy) * r.hiW];
                    bot = bot + den[(size_t)(x0 + sx) + (size_t)(y_bot0 + sy) * r.hiW];
                }
            float inv = 1.0f / (float)(r.ss * r.ss);
            orc::V3 t_sdr = r.tone.map_pixel(orc::v3(top.x * inv, top.y * inv, top.z * inv));
            orc::V3 b_sdr = r.tone.map_pixel(orc::v3(bot.x * inv, bot.y * inv, bot.z * inv));
            float *o = sdr_out + ((size_t)cx + (size_t)cy * r.fbW) * 6;
            o[0] = t_sdr.x; o[1] = t_sdr.y; o[2] = t_sdr.z; o[3] = b_sdr.x; o[4] = b_sdr.y; o[5] = b_sdr.z;
        }
    }
    return YCGE_OK;
}
/* analysis aid: re-trace the last rendered frame and report, per pixel, the traversal steps of each of its first
 * `per_pixel` Scene.Hit calls in call order (primary, shadow rays of vertex 1, bounce, shadow rays of vertex 2, ...) */
int orc_query_profile2(void *ctx, uint32_t *out, uint32_t *out2, int per_pixel);
int orc_query_profile(void *ctx, uint32_t *out, int per_pixel) { return orc_query_profile2(ctx, out, nullptr, per_pixel); }
/* ... out2 (may be null): leaves opened | left descents << 16 of the same calls */
int orc_query_profile2(void *ctx, uint32_t *out, uint32_t *out2, int per_pixel)
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene || !out || per_pixel <= 0) return YCGE_ERR_INVALID_ARG;
    size_t npx = (size_t)r->hiW * r->hiH;
    std::memset(out, 0, npx * per_pixel * sizeof(uint32_t));
    if (out2) std::memset(out2, 0, npx * per_pixel * sizeof(uint32_t));
    for (size_t i = 0; i < npx; i++) {
        int px = (int)(i % r->hiW), py = (int)(i / r->hiW);
        orc::Rng rng(orc::per_frame_seed(px, py, r->frame_counter, 0, 0, r->cfg.seed_salt));
        orc::Counters c; c.qlog = out + i * per_pixel; c.qcap = per_pixel;
        if (out2) c.qlog2 = out2 + i * per_pixel;
        bool is_sky; orc::GBuf g;
        (void)orc::trace_full(r->scene, r->K, r->rays[i], rng, is_sky, g, c);
    }
    return YCGE_OK;
}
/* brute-force closest hit over Scene.Objects in order (no BVH), same tie rule */
int orc_scene_hit_bruteforce(void *ctx, const float o[3], const float d[3], float t_min, float t_max, float out[4])
{
    Renderer *r = (Renderer *)ctx;
    if (!r || !r->have_scene) return YCGE_ERR_NO_SCENE;
    orc::Counters c;
    orc::Ray ray = orc::make_ray(orc::v3(o[0], o[1], o[2]), orc::v3(d[0], d[1], d[2]));
    bool any = false; float closest = t_max; orc::Hit best{};
    for (int i = 0; i < (int)r->scene.prims.size(); i++) {
        orc::Hit tmp{};
        if (r->scene.prim_hit(i, ray, t_min, closest, tmp, c)) { any = true; closest = tmp.t; best = tmp; }
    }
    out[0] = any ? 1.0f : 0.0f; out[1] = (float)best.prim; out[2] = (float)best.sub; out[3] = best.t;
    return YCGE_OK;
}
int orc_morton3(int x, int y, int z)
{
    return ((x & 1) << 0) | ((y & 1) << 1) | ((z & 1) << 2) | ((x & 2) << 2) | ((y & 2) << 3) | ((z & 2) << 4) | ((x & 4) << 4) | ((y & 4) << 5) | ((z & 4) << 6);
}

} // extern "C"
