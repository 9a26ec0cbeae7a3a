/*
 * orc_scene.h — ORACLE scene model: vectors, rays, materials, primitives,
 * the two BVH classes and the voxel grid, restated from the reference C#.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
 * Paths below are relative to /root/reference/ConsoleGame/.
 */
#ifndef ORC_SCENE_H
#define ORC_SCENE_H

#include "orc_math.h"
#include "../include/ycge.h"

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace orc {

/* ---- RayTracing/Vec3.cs ------------------------------------------------ */
struct V3 {
    float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3(const ycge_vec3 &v) { return V3{v.x, v.y, v.z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }        /* Vec3.cs:31-34 */
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }        /* Vec3.cs:37-40 */
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }                              /* Vec3.cs:43-46 */
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }        /* Vec3.cs:49-52 */
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }            /* Vec3.cs:55-58 */
static inline V3 operator/(V3 a, float s) { float inv = 1.0f / s; return V3{a.x * inv, a.y * inv, a.z * inv}; } /* Vec3.cs:67-71 */
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              /* Vec3.cs:74-77 */
static inline V3 cross(V3 a, V3 b)                                                             /* Vec3.cs:80-83 */
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline V3 normalized(V3 a)                                                              /* Vec3.cs:98-107 */
{
    float len_sq = a.x * a.x + a.y * a.y + a.z * a.z;
    if (len_sq <= 0.0f) return a;
    float inv_len = 1.0f / cs_sqrt(len_sq);
    return V3{a.x * inv_len, a.y * inv_len, a.z * inv_len};
}
static inline float clamp01(float v) { if (v < 0.0f) return 0.0f; if (v > 1.0f) return 1.0f; return v; } /* Vec3.cs:116-127 */
static inline V3 saturate(V3 a) { return V3{clamp01(a.x), clamp01(a.y), clamp01(a.z)}; }      /* Vec3.cs:110-113 */

/* ---- RayTracing/Ray.cs: ctor re-normalises Dir (Ray.cs:8-12) ------------ */
struct Ray {
    V3 o, d;
};
static inline Ray make_ray(V3 o, V3 d) { return Ray{o, normalized(d)}; }

/* ---- RayTracing/Material.cs, evaluated (textures are out of scope) ------ */
struct Mat {
    V3 albedo;
    float reflectivity;     /* (float)Material.Reflectivity */
    V3 emission;
    float transparency;     /* (float)Material.Transparency */
    float ior;
    V3 trans_color;
    int32_t tex = -1;       /* Material.DiffuseTexture (index into SceneData::textures), -1 = none */
    double tex_weight = 1.0, uv_scale = 1.0;    /* Material.TextureWeight / UVScale, doubles as in Material.cs:17-18 */
};

/* ---- RayTracing/HitRecord.cs + ids for the parity buffers ---------------- */
struct Hit {
    float t;
    V3 p, n;
    Mat m;
    float u, v;
    int32_t prim;           /* index in Scene.Objects */
    int32_t sub;            /* triangle index / box face / cylinder part (0 side, 1 top cap, 2 bottom cap) / voxel cell */
};

struct Counters {
    uint64_t rays = 0, box = 0, tri = 0, prim = 0, vox = 0, dark = 0;
    /* analysis aid (orc_query_profile): traversal steps (node visits + triangle tests) of each Scene.Hit call, in call order */
    uint32_t *qlog = nullptr; int qn = 0, qcap = 0;
    /* ... and, for the same calls, mesh leaves opened | node visits that go on into the node's LEFT child (the next record in the
     * device arena's depth-first order) << 16: what a wave-cooperative walk would save (tests/analysis_coop_model.py) */
    uint32_t *qlog2 = nullptr; uint64_t leaves = 0, left_desc = 0;
    void add(const Counters &o) { rays += o.rays; box += o.box; tri += o.tri; prim += o.prim; vox += o.vox; dark += o.dark; }
};

struct SceneData; /* fwd */

/* flat node record shared by both BVH classes (BVH.cs:11-20, MeshBVH.cs:18-27) */
struct Node {
    float min_x, min_y, min_z, max_x, max_y, max_z;
    int32_t left, right, start, count;
};

struct BuildItem {          /* BVH.cs:245-250 / MeshBVH.cs:342-347 */
    int32_t index;
    float min_x, min_y, min_z, max_x, max_y, max_z;
    float cx, cy, cz;
};

struct BuildStats {
    int32_t sort_fallbacks = 0;  /* nodes that took an Array.Sort path */
    int32_t max_depth = 0;
};

/* ---- Objects/MeshBVH.cs ------------------------------------------------- */
struct MeshAccel {
    std::vector<Node> nodes;
    std::vector<int32_t> leaf_tri;
    std::vector<float> ax, ay, az, e1x, e1y, e1z, e2x, e2y, e2z, nx, ny, nz;
    std::vector<int32_t> tri_mat;   /* material index per triangle */
    int32_t root = -1;
    BuildStats stats;
    void build(const float *tris9, int32_t n, int32_t material, const int32_t *tri_material);
};

/* ---- Objects/VolumeGrid.cs ---------------------------------------------- */
struct Grid {
    int32_t nx, ny, nz, nbx, nby, nbz;
    std::vector<int32_t> mat, meta;     /* bricked, Morton inside brick */
    V3 min_corner, voxel_size;
    bool wireframe;
    float wire_width_frac, wire_max_distance;
    std::vector<ycge_voxel_lookup> lookup;
    int32_t default_material;
    int index_of(int ix, int iy, int iz) const;
};

struct Prim {
    int32_t type, material, ref;
    float p[12];
    float specular, reflectivity;
    /* derived at construction like the C# ctors do */
    V3 normal;          /* Plane / Disk: Normal = n.Normalized() */
    V3 normal_neg;      /* Plane.NormalNeg */
    float ndot;         /* Plane.ndotPoint / Disk.ndotCenter */
    float radius2;      /* Disk / CylinderY */
    float y_min, y_max; /* CylinderY ctor sorts them */
    /* Triangle cached edges + unit normal (Triangle.cs:36-45) */
    float e1x, e1y, e1z, e2x, e2y, e2z, tnx, tny, tnz;
};

struct Light { V3 pos, color; float intensity; };

/* Renderer/Texture.cs, static texture: pixels[y * width + x] = RGBA32.ToInt() */
struct Texture {
    int32_t width = 0, height = 0;
    std::vector<uint32_t> pixels;
    /* a live texture (Texture.cs:51-66): bytes per pixel of its frames (3 BGR / 4 BGRA; 0 = static), flips, the current frame */
    int32_t frame_bpp = 0; bool flip_u = false, flip_v = false;
    std::vector<uint8_t> frame;
    V3 sample_bilinear(float u, float v) const;     /* Texture.cs:108-163, both branches */
};

struct SceneData {
    std::vector<ycge_material> materials;
    std::vector<Prim> prims;
    std::vector<MeshAccel> meshes;
    std::vector<Grid> grids;
    std::vector<Light> lights;
    std::vector<Texture> textures;
    V3 ambient_color; float ambient_intensity;
    V3 bg_top, bg_bottom;
    bool is_volume_scene;
    bool has_dynamic_textures = false;      /* Scene.HasDynamicTextures, Scenes/Scene.cs:30 */
    /* scene-level BVH (Objects/BVH.cs) */
    std::vector<Node> nodes;
    std::vector<int32_t> leaf_obj;
    int32_t root = -1;
    BuildStats stats;

    std::string load(const ycge_scene *s);   /* returns error text or "" */
    void rebuild_bvh();                       /* Scene.RebuildBVH, Scene.cs:66-69 */
    bool hit(const Ray &r, float t_min, float t_max, Hit &rec, Counters &c) const;        /* Scene.cs:71-75 */
    bool occluded(const Ray &r, float max_dist, Counters &c) const;                       /* Scene.cs:77-82 */
    bool prim_hit(int32_t prim_index, const Ray &r, float t_min, float t_max, Hit &rec, Counters &c) const;
    bool prim_bounds(int32_t prim_index, float b[6], float c[3]) const;
    Mat eval_material(int32_t material, V3 pos) const;
    V3 sample_albedo(const Mat &m, float u, float v) const;      /* RaytraceRenderer.SampleAlbedo, RaytraceRenderer.cs:724-735 */
};

/* .NET 8 Array.Sort(T[], int, int, IComparer<T>) = ArraySortHelper<T>.IntrospectiveSort */
void dotnet_introsort(BuildItem *keys, int n, int axis);

} // namespace orc
#endif
