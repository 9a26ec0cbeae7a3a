/*
 * orc_scene.cpp — ORACLE: primitives, BVH builders and closest-hit queries,
 * restated from the reference C# (paths relative to /root/reference/ConsoleGame/).
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
 */
#include "orc_scene.h"

#include <algorithm>
#include <cstdio>

namespace orc {

/* ======================================================================
 * .NET 8 Array.Sort(arr, start, count, Comparer<Item>.Create(cmp))
 * (BVH.cs:389,419; MeshBVH.cs:506,536).  Not vendored in the reference:
 * System.Private.CoreLib 8.0, ArraySortHelper<T>.IntrospectiveSort with a
 * Comparison<T>.  Restated from its published algorithm: introsort with
 * depth limit 2*(floor(log2 n)+1), insertion sort at <= 16 elements,
 * median-of-three pivot moved to hi-1, heapsort on depth exhaustion.  It is
 * unstable, so equal centroids keep the order THIS procedure leaves them in.
 * ====================================================================== */
namespace {

struct AxisCmp {
    int axis;
    /* float.CompareTo (a.Cx.CompareTo(b.Cx)): <0, 0, >0; NaN sorts first */
    int operator()(const BuildItem &a, const BuildItem &b) const
    {
        float x = axis == 0 ? a.cx : axis == 1 ? a.cy : a.cz;
        float y = axis == 0 ? b.cx : axis == 1 ? b.cy : b.cz;
        if (x < y) return -1;
        if (x > y) return 1;
        if (x == y) return 0;
        if (x != x) return (y != y) ? 0 : -1;
        return 1;
    }
};

inline void swap_items(BuildItem *k, int i, int j) { BuildItem t = k[i]; k[i] = k[j]; k[j] = t; }
inline void swap_if_greater(BuildItem *k, const AxisCmp &c, int i, int j)
{
    if (c(k[i], k[j]) > 0) swap_items(k, i, j);
}
void insertion_sort(BuildItem *k, int n, const AxisCmp &c)
{
    for (int i = 0; i < n - 1; i++) {
        BuildItem t = k[i + 1];
        int j = i;
        while (j >= 0 && c(t, k[j]) < 0) {
            k[j + 1] = k[j];
            j--;
        }
        k[j + 1] = t;
    }
}
void down_heap(BuildItem *k, int i, int n, const AxisCmp &c)
{
    BuildItem d = k[i - 1];
    while (i <= n >> 1) {
        int child = 2 * i;
        if (child < n && c(k[child - 1], k[child]) < 0) child++;
        if (!(c(d, k[child - 1]) < 0)) break;
        k[i - 1] = k[child - 1];
        i = child;
    }
    k[i - 1] = d;
}
void heap_sort(BuildItem *k, int n, const AxisCmp &c)
{
    for (int i = n >> 1; i >= 1; i--) down_heap(k, i, n, c);
    for (int i = n; i > 1; i--) {
        swap_items(k, 0, i - 1);
        down_heap(k, 1, i - 1, c);
    }
}
int pick_pivot_and_partition(BuildItem *k, int n, const AxisCmp &c)
{
    int hi = n - 1;
    int middle = hi >> 1;
    swap_if_greater(k, c, 0, middle);
    swap_if_greater(k, c, 0, hi);
    swap_if_greater(k, c, middle, hi);
    BuildItem pivot = k[middle];
    swap_items(k, middle, hi - 1);
    int left = 0, right = hi - 1;
    while (left < right) {
        while (c(k[++left], pivot) < 0) {}
        while (c(pivot, k[--right]) < 0) {}
        if (left >= right) break;
        swap_items(k, left, right);
    }
    if (left != hi - 1) swap_items(k, left, hi - 1);
    return left;
}
void intro_sort(BuildItem *k, int n, int depth_limit, const AxisCmp &c)
{
    int partition_size = n;
    while (partition_size > 1) {
        if (partition_size <= 16) {
            if (partition_size == 2) { swap_if_greater(k, c, 0, 1); return; }
            if (partition_size == 3) {
                swap_if_greater(k, c, 0, 1);
                swap_if_greater(k, c, 0, 2);
                swap_if_greater(k, c, 1, 2);
                return;
            }
            insertion_sort(k, partition_size, c);
            return;
        }
        if (depth_limit == 0) { heap_sort(k, partition_size, c); return; }
        depth_limit--;
        int p = pick_pivot_and_partition(k, partition_size, c);
        intro_sort(k + p + 1, partition_size - (p + 1), depth_limit, c);
        partition_size = p;
    }
}
inline int log2_floor(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

} // namespace

void dotnet_introsort(BuildItem *keys, int n, int axis)
{
    if (n > 1) {
        AxisCmp c{axis};
        intro_sort(keys, n, 2 * (log2_floor((uint32_t)n) + 1), c);
    }
}

/* ======================================================================
 * Shared recursive binned-SAH builder.
 *   scene flavour  : Objects/BVH.cs:258-459      (leaf 4, partition quirk 394-396)
 *   mesh flavour   : Objects/MeshBVH.cs:371-576  (leaf 8, partitions with the binning bounds 511-513)
 * ====================================================================== */
namespace {

const int kBins = 16;   /* SAH_Bins, BVH.cs:8 / MeshBVH.cs:15 */

inline void surround(float &mnx, float &mny, float &mnz, float &mxx, float &mxy, float &mxz,
                     float ox0, float oy0, float oz0, float ox1, float oy1, float oz1)
{   /* BVH.cs:252-256 */
    if (ox0 < mnx) mnx = ox0; if (oy0 < mny) mny = oy0; if (oz0 < mnz) mnz = oz0;
    if (ox1 > mxx) mxx = ox1; if (oy1 > mxy) mxy = oy1; if (oz1 > mxz) mxz = oz1;
}
inline float surface_area(float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{   /* BVH.cs:462-466 */
    float dx = mxx - mnx, dy = mxy - mny, dz = mxz - mnz;
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
inline float centroid(const BuildItem &it, int ax) { return ax == 0 ? it.cx : ax == 1 ? it.cy : it.cz; }

struct Builder {
    std::vector<Node> &nodes;
    std::vector<int32_t> &leaf_idx;
    BuildItem *arr;
    bool scene_flavour;
    int leaf_size;
    BuildStats &stats;

    int build(int start, int count, int depth)
    {
        if (count <= 0) return -1;
        if (depth > stats.max_depth) stats.max_depth = depth;
        if (count <= leaf_size) {
            Node leaf{};
            float mnx = arr[start].min_x, mny = arr[start].min_y, mnz = arr[start].min_z;
            float mxx = arr[start].max_x, mxy = arr[start].max_y, mxz = arr[start].max_z;
            for (int i = 1; i < count; i++) {
                const BuildItem &o = arr[start + i];
                surround(mnx, mny, mnz, mxx, mxy, mxz, o.min_x, o.min_y, o.min_z, o.max_x, o.max_y, o.max_z);
            }
            int base = (int)leaf_idx.size();
            for (int i = 0; i < count; i++) leaf_idx.push_back(arr[start + i].index);
            leaf.min_x = mnx; leaf.min_y = mny; leaf.min_z = mnz;
            leaf.max_x = mxx; leaf.max_y = mxy; leaf.max_z = mxz;
            leaf.left = -1; leaf.right = -1; leaf.start = base; leaf.count = count;
            int idx = (int)nodes.size();
            nodes.push_back(leaf);
            return idx;
        }

        float cminx = arr[start].cx, cminy = arr[start].cy, cminz = arr[start].cz;
        float cmaxx = cminx, cmaxy = cminy, cmaxz = cminz;
        for (int i = start + 1; i < start + count; i++) {
            float cx = arr[i].cx, cy = arr[i].cy, cz = arr[i].cz;
            if (cx < cminx) cminx = cx; if (cy < cminy) cminy = cy; if (cz < cminz) cminz = cz;
            if (cx > cmaxx) cmaxx = cx; if (cy > cmaxy) cmaxy = cy; if (cz > cmaxz) cmaxz = cz;
        }
        float ext_x = cmaxx - cminx, ext_y = cmaxy - cminy, ext_z = cmaxz - cminz;
        int axis = 0;
        if (ext_y > ext_x && ext_y >= ext_z) axis = 1; else if (ext_z > ext_x && ext_z >= ext_y) axis = 2;

        int split_bin = -1;
        int best_axis = axis;
        float best_cost = kInf;

        for (int ax = 0; ax < 3; ax++) {
            float extent = ax == 0 ? ext_x : ax == 1 ? ext_y : ext_z;
            if (!(extent > 0.0f)) continue;
            float origin = ax == 0 ? cminx : ax == 1 ? cminy : cminz;
            float inv_extent = 1.0f / extent;

            int counts[kBins];
            float lminx[kBins], lminy[kBins], lminz[kBins], lmaxx[kBins], lmaxy[kBins], lmaxz[kBins];
            for (int b = 0; b < kBins; b++) {
                lminx[b] = lminy[b] = lminz[b] = kInf;
                lmaxx[b] = lmaxy[b] = lmaxz[b] = -kInf;
                counts[b] = 0;
            }
            for (int i = start; i < start + count; i++) {
                float c = centroid(arr[i], ax);
                int b = cs_f2i((c - origin) * inv_extent * (float)(kBins - 1));
                if (b < 0) b = 0; if (b >= kBins) b = kBins - 1;
                counts[b]++;
                surround(lminx[b], lminy[b], lminz[b], lmaxx[b], lmaxy[b], lmaxz[b],
                         arr[i].min_x, arr[i].min_y, arr[i].min_z, arr[i].max_x, arr[i].max_y, arr[i].max_z);
            }
            int left_count[kBins], right_count[kBins];
            float left_area[kBins], right_area[kBins];
            float cx0 = kInf, cy0 = kInf, cz0 = kInf, cx1 = -kInf, cy1 = -kInf, cz1 = -kInf;
            int acc = 0;
            for (int b = 0; b < kBins; b++) {
                if (counts[b] > 0) surround(cx0, cy0, cz0, cx1, cy1, cz1, lminx[b], lminy[b], lminz[b], lmaxx[b], lmaxy[b], lmaxz[b]);
                acc += counts[b];
                left_count[b] = acc;
                left_area[b] = surface_area(cx0, cy0, cz0, cx1, cy1, cz1);
            }
            cx0 = cy0 = cz0 = kInf; cx1 = cy1 = cz1 = -kInf;
            acc = 0;
            for (int b = kBins - 1; b >= 0; b--) {
                if (counts[b] > 0) surround(cx0, cy0, cz0, cx1, cy1, cz1, lminx[b], lminy[b], lminz[b], lmaxx[b], lmaxy[b], lmaxz[b]);
                acc += counts[b];
                right_count[b] = acc;
                right_area[b] = surface_area(cx0, cy0, cz0, cx1, cy1, cz1);
            }
            for (int b = 0; b < kBins - 1; b++) {
                int lc = left_count[b];
                int rc = right_count[b + 1];
                if (lc == 0 || rc == 0) continue;
                float cost = left_area[b] * (float)lc + right_area[b + 1] * (float)rc;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; split_bin = b; }
            }
        }

        int mid;
        if (split_bin < 0) {
            dotnet_introsort(arr + start, count, best_axis);
            stats.sort_fallbacks++;
            mid = start + (count >> 1);
        } else {
            float origin, extent, inv_extent;
            if (scene_flavour) {
                /* BVH.cs:394-396 — origin/extent from the first and last ITEM, not the binning bounds */
                origin = centroid(arr[start], best_axis);
                extent = centroid(arr[start + count - 1], best_axis) - origin;
                inv_extent = extent != 0.0f ? 1.0f / extent : 0.0f;
            } else {
                /* MeshBVH.cs:511-513 */
                origin = best_axis == 0 ? cminx : best_axis == 1 ? cminy : cminz;
                extent = best_axis == 0 ? ext_x : best_axis == 1 ? ext_y : ext_z;
                inv_extent = 1.0f / extent;
            }
            int i0 = start, i1 = start + count - 1;
            while (i0 <= i1) {
                float c0 = centroid(arr[i0], best_axis);
                int b0;
                if (scene_flavour)
                    b0 = inv_extent != 0.0f ? cs_f2i((c0 - origin) * inv_extent * (float)(kBins - 1)) : 0;
                else
                    b0 = cs_f2i((c0 - origin) * inv_extent * (float)(kBins - 1));
                if (b0 <= split_bin) {
                    i0++;
                } else {
                    BuildItem tmp = arr[i0]; arr[i0] = arr[i1]; arr[i1] = tmp; i1--;
                }
            }
            mid = i0;
            if (mid == start || mid == start + count) {
                dotnet_introsort(arr + start, count, best_axis);
                stats.sort_fallbacks++;
                mid = start + (count >> 1);
            }
        }

        int my_index = (int)nodes.size();
        nodes.push_back(Node{});
        int left_index = build(start, mid - start, depth + 1);
        int right_index = build(mid, start + count - mid, depth + 1);

        Node cur{};
        cur.left = left_index;
        cur.right = right_index;
        if (left_index >= 0 && right_index >= 0) {
            const Node &L = nodes[left_index];
            const Node &R = nodes[right_index];
            cur.min_x = cs_min(L.min_x, R.min_x); cur.min_y = cs_min(L.min_y, R.min_y); cur.min_z = cs_min(L.min_z, R.min_z);
            cur.max_x = cs_max(L.max_x, R.max_x); cur.max_y = cs_max(L.max_y, R.max_y); cur.max_z = cs_max(L.max_z, R.max_z);
        } else if (left_index >= 0) {
            const Node &L = nodes[left_index];
            cur.min_x = L.min_x; cur.min_y = L.min_y; cur.min_z = L.min_z; cur.max_x = L.max_x; cur.max_y = L.max_y; cur.max_z = L.max_z;
        } else {
            const Node &R = nodes[right_index];
            cur.min_x = R.min_x; cur.min_y = R.min_y; cur.min_z = R.min_z; cur.max_x = R.max_x; cur.max_y = R.max_y; cur.max_z = R.max_z;
        }
        cur.start = 0; cur.count = 0;
        nodes[my_index] = cur;
        return my_index;
    }
};

} // namespace

/* ---- MeshBVH ctor, MeshBVH.cs:41-130 ------------------------------------ */
void MeshAccel::build(const float *t9, int32_t n, int32_t material, const int32_t *tri_material)
{
    nodes.clear(); leaf_tri.clear(); root = -1; stats = BuildStats{};
    ax.resize(n); ay.resize(n); az.resize(n);
    e1x.resize(n); e1y.resize(n); e1z.resize(n);
    e2x.resize(n); e2y.resize(n); e2z.resize(n);
    nx.resize(n); ny.resize(n); nz.resize(n);
    tri_mat.resize(n);
    if (n == 0) return;
    std::vector<BuildItem> items(n);
    const float eps = 1e-4f;   /* TryComputeBounds, MeshBVH.cs:349-363 */
    for (int i = 0; i < n; i++) {
        const float *t = t9 + 9 * (size_t)i;
        float Ax = t[0], Ay = t[1], Az = t[2], Bx = t[3], By = t[4], Bz = t[5], Cx = t[6], Cy = t[7], Cz = t[8];
        BuildItem it;
        it.min_x = cs_min(Ax, cs_min(Bx, Cx)) - eps;
        it.min_y = cs_min(Ay, cs_min(By, Cy)) - eps;
        it.min_z = cs_min(Az, cs_min(Bz, Cz)) - eps;
        it.max_x = cs_max(Ax, cs_max(Bx, Cx)) + eps;
        it.max_y = cs_max(Ay, cs_max(By, Cy)) + eps;
        it.max_z = cs_max(Az, cs_max(Bz, Cz)) + eps;
        it.index = i;
        it.cx = 0.5f * (it.min_x + it.max_x);
        it.cy = 0.5f * (it.min_y + it.max_y);
        it.cz = 0.5f * (it.min_z + it.max_z);
        items[i] = it;
        /* triangle SoA, MeshBVH.cs:82-100 */
        ax[i] = Ax; ay[i] = Ay; az[i] = Az;
        float lx = Bx - Ax, ly = By - Ay, lz = Bz - Az;
        float mx = Cx - Ax, my = Cy - Ay, mz = Cz - Az;
        e1x[i] = lx; e1y[i] = ly; e1z[i] = lz;
        e2x[i] = mx; e2y[i] = my; e2z[i] = mz;
        float nnx = ly * mz - lz * my;
        float nny = lz * mx - lx * mz;
        float nnz = lx * my - ly * mx;
        float inv_len = 1.0f / cs_max(1e-20f, cs_sqrt(nnx * nnx + nny * nny + nnz * nnz));
        nx[i] = nnx * inv_len; ny[i] = nny * inv_len; nz[i] = nnz * inv_len;
        tri_mat[i] = tri_material ? tri_material[i] : material;
    }
    nodes.reserve(2 * (size_t)n);
    leaf_tri.reserve(n);
    Builder b{nodes, leaf_tri, items.data(), false, 8 /* MeshBVH.cs:14 */, stats};
    root = b.build(0, n, 1);
}

/* ---- VolumeGrid.IndexOf / Morton3_3bits, VolumeGrid.cs:235-252 ---------- */
static inline int morton3_3bits(int x, int y, int z)
{
    return ((x & 1) << 0) | ((y & 1) << 1) | ((z & 1) << 2)
         | ((x & 2) << 2) | ((y & 2) << 3) | ((z & 2) << 4)
         | ((x & 4) << 4) | ((y & 4) << 5) | ((z & 4) << 6);
}
int Grid::index_of(int ix, int iy, int iz) const
{
    int bx = ix >> 3, by = iy >> 3, bz = iz >> 3;
    int lx = ix & 7, ly = iy & 7, lz = iz & 7;
    int brick_linear = ((bz * nby) + by) * nbx + bx;
    return brick_linear * 512 + morton3_3bits(lx, ly, lz);
}

/* ---- scene ingest ------------------------------------------------------- */
std::string SceneData::load(const ycge_scene *s)
{
    if (!s) return "null scene";
    materials.assign(s->materials, s->materials + s->n_materials);
    textures.clear();
    for (int i = 0; i < s->n_textures; i++) {
        const ycge_texture &t = s->textures[i];
        if (t.width < 1 || t.height < 1 || (!t.pixels && t.frame_bytes_per_pixel == 0)) return "texture without pixels";
        Texture T;
        T.width = t.width; T.height = t.height;
        T.frame_bpp = t.frame_bytes_per_pixel; T.flip_u = t.flip_u != 0; T.flip_v = t.flip_v != 0;
        if (T.frame_bpp) {
            T.frame.assign((size_t)t.width * t.height * T.frame_bpp, 0);
            if (t.frame) std::memcpy(T.frame.data(), t.frame, T.frame.size());
        } else
        T.pixels.assign(t.pixels, t.pixels + (size_t)t.width * t.height);
        textures.push_back(std::move(T));
    }
    for (int i = 0; i < s->n_materials; i++)
        if (s->materials[i].kind == YCGE_MAT_TEXTURED && (s->materials[i].texture < 0 || s->materials[i].texture >= s->n_textures)) return "material texture index out of range";
    lights.clear();
    for (int i = 0; i < s->n_lights; i++)
        lights.push_back(Light{v3(s->lights[i].position), v3(s->lights[i].color), s->lights[i].intensity});
    ambient_color = v3(s->ambient_color); ambient_intensity = s->ambient_intensity;
    bg_top = v3(s->background_top); bg_bottom = v3(s->background_bottom);
    is_volume_scene = s->is_volume_scene != 0;
    has_dynamic_textures = s->has_dynamic_textures != 0;

    meshes.clear(); meshes.resize(s->n_meshes);
    for (int i = 0; i < s->n_meshes; i++) {
        const ycge_mesh &m = s->meshes[i];
        meshes[i].build(m.triangles, m.n_triangles, m.material, m.tri_material);
    }
    grids.clear(); grids.resize(s->n_grids);
    for (int gi = 0; gi < s->n_grids; gi++) {
        const ycge_grid &g = s->grids[gi];
        Grid &G = grids[gi];
        /* VolumeGrid ctor, VolumeGrid.cs:55-93 */
        G.nx = g.nx; G.ny = g.ny; G.nz = g.nz;
        G.nbx = (g.nx + 7) >> 3; G.nby = (g.ny + 7) >> 3; G.nbz = (g.nz + 7) >> 3;
        size_t cap = (size_t)G.nbx * G.nby * G.nbz * 512;
        G.mat.assign(cap, 0); G.meta.assign(cap, 0);
        G.min_corner = v3(g.min_corner);
        G.voxel_size = v3(cs_max(1e-6f, g.voxel_size.x), cs_max(1e-6f, g.voxel_size.y), cs_max(1e-6f, g.voxel_size.z));
        G.wireframe = g.wireframe != 0;
        float ww = g.wire_width_fraction;
        if (ww < 0.0f) ww = 0.0f; if (ww > 0.5f) ww = 0.5f;
        G.wire_width_frac = ww;
        float wm = g.wire_max_distance;
        if (wm < 0.0f) wm = 0.0f;
        G.wire_max_distance = wm;
        G.lookup.assign(g.lookup, g.lookup + g.n_lookup);
        G.default_material = g.default_material;
        for (int iz = 0; iz < g.nz; iz++)
            for (int iy = 0; iy < g.ny; iy++)
                for (int ix = 0; ix < g.nx; ix++) {
                    size_t src = ((size_t)ix * g.ny + iy) * g.nz + iz;
                    int idx = G.index_of(ix, iy, iz);
                    G.mat[idx] = g.cells[2 * src];
                    G.meta[idx] = g.cells[2 * src + 1];
                }
    }

    prims.clear(); prims.resize(s->n_prims);
    for (int i = 0; i < s->n_prims; i++) {
        const ycge_prim &q = s->prims[i];
        Prim &P = prims[i];
        P = Prim{};
        P.type = q.type; P.material = q.material; P.ref = q.ref;
        for (int k = 0; k < 12; k++) P.p[k] = q.p[k];
        P.specular = q.specular; P.reflectivity = q.reflectivity;
        switch (q.type) {
        case YCGE_PRIM_PLANE: {   /* Surfaces.cs:19-28 */
            P.normal = normalized(v3(q.p[3], q.p[4], q.p[5]));
            P.ndot = P.normal.x * q.p[0] + P.normal.y * q.p[1] + P.normal.z * q.p[2];
            P.normal_neg = v3(-P.normal.x, -P.normal.y, -P.normal.z);
            break;
        }
        case YCGE_PRIM_DISK: {    /* Surfaces.cs:84-94 */
            P.normal = normalized(v3(q.p[3], q.p[4], q.p[5]));
            P.ndot = dot(P.normal, v3(q.p[0], q.p[1], q.p[2]));
            P.radius2 = q.p[6] * q.p[6];
            break;
        }
        case YCGE_PRIM_CYLINDER_Y: { /* BoundedObjects.cs:128-137 */
            P.y_min = cs_min(q.p[4], q.p[5]);
            P.y_max = cs_max(q.p[4], q.p[5]);
            P.radius2 = q.p[3] * q.p[3];
            break;
        }
        case YCGE_PRIM_TRIANGLE: { /* Triangle.cs:36-45 */
            P.e1x = q.p[3] - q.p[0]; P.e1y = q.p[4] - q.p[1]; P.e1z = q.p[5] - q.p[2];
            P.e2x = q.p[6] - q.p[0]; P.e2y = q.p[7] - q.p[1]; P.e2z = q.p[8] - q.p[2];
            float nnx = P.e1y * P.e2z - P.e1z * P.e2y;
            float nny = P.e1z * P.e2x - P.e1x * P.e2z;
            float nnz = P.e1x * P.e2y - P.e1y * P.e2x;
            float inv_len = 1.0f / cs_max(1e-20f, cs_sqrt(nnx * nnx + nny * nny + nnz * nnz));
            P.tnx = nnx * inv_len; P.tny = nny * inv_len; P.tnz = nnz * inv_len;
            break;
        }
        case YCGE_PRIM_MESH:
            if (q.ref < 0 || q.ref >= s->n_meshes) return "mesh ref out of range";
            break;
        case YCGE_PRIM_VOLUME_GRID:
            if (q.ref < 0 || q.ref >= s->n_grids) return "grid ref out of range";
            break;
        default: break;
        }
    }
    rebuild_bvh();
    if (!prims.empty() && root < 0) return "Unbounded Hittable";
    return "";
}

/* Hittable.TryGetBounds of each primitive class */
bool SceneData::prim_bounds(int32_t pi, float b[6], float c[3]) const
{
    const Prim &P = prims[pi];
    const float *p = P.p;
    const float eps = 1e-4f;
    bool centre_from_box = true;
    switch (P.type) {
    case YCGE_PRIM_SPHERE:     /* BoundedObjects.cs:20-28 */
        b[0] = p[0] - p[3]; b[1] = p[1] - p[3]; b[2] = p[2] - p[3];
        b[3] = p[0] + p[3]; b[4] = p[1] + p[3]; b[5] = p[2] + p[3];
        break;
    case YCGE_PRIM_PLANE:      /* Surfaces.cs:30-36 */
        b[0] = b[1] = b[2] = -1e6f; b[3] = b[4] = b[5] = 1e6f;
        c[0] = c[1] = c[2] = 0.0f; centre_from_box = false;
        break;
    case YCGE_PRIM_DISK:       /* Surfaces.cs:97-105 */
        b[0] = p[0] - p[6]; b[1] = p[1] - p[6]; b[2] = p[2] - p[6];
        b[3] = p[0] + p[6]; b[4] = p[1] + p[6]; b[5] = p[2] + p[6];
        break;
    case YCGE_PRIM_XYRECT:     /* Surfaces.cs:174-181 */
        b[0] = p[0]; b[1] = p[2]; b[2] = p[4] - eps; b[3] = p[1]; b[4] = p[3]; b[5] = p[4] + eps;
        break;
    case YCGE_PRIM_XZRECT:     /* Surfaces.cs:246-253 */
        b[0] = p[0]; b[1] = p[4] - eps; b[2] = p[2]; b[3] = p[1]; b[4] = p[4] + eps; b[5] = p[3];
        break;
    case YCGE_PRIM_YZRECT:     /* Surfaces.cs:318-325 */
        b[0] = p[4] - eps; b[1] = p[0]; b[2] = p[2]; b[3] = p[4] + eps; b[4] = p[1]; b[5] = p[3];
        break;
    case YCGE_PRIM_BOX:        /* BoundedObjects.cs:92-97 */
        for (int k = 0; k < 6; k++) b[k] = p[k];
        break;
    case YCGE_PRIM_CYLINDER_Y: /* BoundedObjects.cs:140-145 */
        b[0] = p[0] - p[3]; b[1] = P.y_min; b[2] = p[2] - p[3];
        b[3] = p[0] + p[3]; b[4] = P.y_max; b[5] = p[2] + p[3];
        break;
    case YCGE_PRIM_TRIANGLE: { /* Triangle.cs:54-64 */
        b[0] = cs_min(p[0], cs_min(p[3], p[6])) - eps;
        b[1] = cs_min(p[1], cs_min(p[4], p[7])) - eps;
        b[2] = cs_min(p[2], cs_min(p[5], p[8])) - eps;
        b[3] = cs_max(p[0], cs_max(p[3], p[6])) + eps;
        b[4] = cs_max(p[1], cs_max(p[4], p[7])) + eps;
        b[5] = cs_max(p[2], cs_max(p[5], p[8])) + eps;
        break;
    }
    case YCGE_PRIM_MESH: {     /* MeshBVH.cs:585-602 via Mesh.cs:34-37 */
        const MeshAccel &m = meshes[P.ref];
        if (m.root < 0) return false;
        const Node &r = m.nodes[m.root];
        b[0] = r.min_x; b[1] = r.min_y; b[2] = r.min_z; b[3] = r.max_x; b[4] = r.max_y; b[5] = r.max_z;
        break;
    }
    case YCGE_PRIM_VOLUME_GRID: { /* VolumeGrid.cs:393-410 */
        const Grid &g = grids[P.ref];
        if (g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return false;
        b[0] = g.min_corner.x; b[1] = g.min_corner.y; b[2] = g.min_corner.z;
        b[3] = g.min_corner.x + (float)g.nx * g.voxel_size.x;
        b[4] = g.min_corner.y + (float)g.ny * g.voxel_size.y;
        b[5] = g.min_corner.z + (float)g.nz * g.voxel_size.z;
        break;
    }
    default: return false;
    }
    if (centre_from_box) {
        c[0] = 0.5f * (b[0] + b[3]); c[1] = 0.5f * (b[1] + b[4]); c[2] = 0.5f * (b[2] + b[5]);
    }
    return true;
}

/* BVH ctor, BVH.cs:29-97 */
void SceneData::rebuild_bvh()
{
    nodes.clear(); leaf_obj.clear(); root = -1; stats = BuildStats{};
    int n = (int)prims.size();
    if (n == 0) return;
    std::vector<BuildItem> items(n);
    for (int i = 0; i < n; i++) {
        float b[6], c[3];
        if (!prim_bounds(i, b, c)) { nodes.clear(); root = -1; return; }
        BuildItem it;
        it.index = i;
        it.min_x = b[0]; it.min_y = b[1]; it.min_z = b[2]; it.max_x = b[3]; it.max_y = b[4]; it.max_z = b[5];
        it.cx = c[0]; it.cy = c[1]; it.cz = c[2];
        items[i] = it;
    }
    Builder bld{nodes, leaf_obj, items.data(), true, 4 /* BVH.cs:7 */, stats};
    root = bld.build(0, n, 1);
}

/* material delegates: Solid / Emissive / Checker, Scenes.cs:408-428 */
Mat SceneData::eval_material(int32_t mi, V3 pos) const
{
    const ycge_material &m = materials[mi];
    Mat out;
    if (m.kind == YCGE_MAT_CHECKER) {
        int32_t cx = cs_f2i(cs_floor(pos.x / m.checker_scale));
        int32_t cz = cs_f2i(cs_floor(pos.z / m.checker_scale));
        bool check = (((uint32_t)cx + (uint32_t)cz) & 1u) == 0u;
        out.albedo = check ? v3(m.albedo) : v3(m.albedo_b);
    } else {
        out.albedo = v3(m.albedo);
    }
    out.reflectivity = m.reflectivity;
    out.emission = v3(m.emission);
    out.transparency = m.transparency;
    out.ior = m.index_of_refraction;
    out.trans_color = v3(m.transmission_color);
    if (m.kind == YCGE_MAT_TEXTURED) { out.tex = m.texture; out.tex_weight = m.texture_weight; out.uv_scale = m.uv_scale; }
    return out;
}

/* Texture.SampleBilinear, Texture.cs:142-163: wrap by u - floor(u), texel coordinates over (size - 1), right / lower neighbour
 * wraps with %, RGBA32.toVec3 = byte / 255f (RGBA32.cs:82-85), Lerp(a, b, t) = a * (1 - t) + b * t per component
 * (Texture.cs:165-168), Saturate.  A u or v that is not finite would index out of range in the reference (an exception);
 * here such an index yields white - not reachable from finite barycentrics and rectangle coordinates. */
V3 Texture::sample_bilinear(float u, float v) const
{
    if (width <= 0 || height <= 0) return v3(1.0f, 1.0f, 1.0f);
    if (frame_bpp != 0) {
        /* Texture.cs:113-140, the live branch: flips, Frac, neighbours CLAMPED at the last column / row, LoadPixel reads B, G, R
         * (:173-182), r0 = r00 * (1 - dTx) + r10 * dTx per channel and row, one Saturate at the end.  (An index from a NaN coordinate
         * reads outside the frame in the reference; white here, as in the static branch.) */
        float uu = flip_u ? (1.0f - u) : u;
        float vv = flip_v ? (1.0f - v) : v;
        float dfx = (uu - cs_floor(uu)) * (float)(width - 1);
        float dfy = (vv - cs_floor(vv)) * (float)(height - 1);
        int x0 = cs_f2i(cs_floor(dfx)), y0 = cs_f2i(cs_floor(dfy));
        if (x0 < 0 || x0 >= width || y0 < 0 || y0 >= height) return v3(1.0f, 1.0f, 1.0f);
        int x1 = (x0 + 1) >= width ? (width - 1) : (x0 + 1);
        int y1 = (y0 + 1) >= height ? (height - 1) : (y0 + 1);
        float tx = dfx - (float)x0, ty = dfy - (float)y0;
        auto load = [&](int x, int y) {
            const uint8_t *q = frame.data() + ((size_t)y * width + x) * frame_bpp;
            return v3((float)q[2] / 255.0f, (float)q[1] / 255.0f, (float)q[0] / 255.0f);
        };
        V3 c00 = load(x0, y0), c10 = load(x1, y0), c01 = load(x0, y1), c11 = load(x1, y1);
        float sx = 1.0f - tx, sy = 1.0f - ty;
        V3 r0 = v3(c00.x * sx + c10.x * tx, c00.y * sx + c10.y * tx, c00.z * sx + c10.z * tx);
        V3 r1 = v3(c01.x * sx + c11.x * tx, c01.y * sx + c11.y * tx, c01.z * sx + c11.z * tx);
        return v3(clamp01(r0.x * sy + r1.x * ty), clamp01(r0.y * sy + r1.y * ty), clamp01(r0.z * sy + r1.z * ty));
    }
    if (pixels.empty()) return v3(1.0f, 1.0f, 1.0f);
    u = u - cs_floor(u);
    v = v - cs_floor(v);
    float fx = u * (float)(width - 1);
    float fy = v * (float)(height - 1);
    int x0 = cs_f2i(cs_floor(fx));
    int y0 = cs_f2i(cs_floor(fy));
    if (x0 < 0 || x0 >= width || y0 < 0 || y0 >= height) return v3(1.0f, 1.0f, 1.0f);
    int x1 = (x0 + 1) % width;
    int y1 = (y0 + 1) % height;
    float tx = fx - (float)x0;
    float ty = fy - (float)y0;
    auto texel = [&](int x, int y) {
        uint32_t c = pixels[(size_t)y * width + x];
        return v3((float)(c & 255u) / 255.0f, (float)((c >> 8) & 255u) / 255.0f, (float)((c >> 16) & 255u) / 255.0f);
    };
    auto lerp3 = [](V3 a, V3 b, float t) {
        float s = 1.0f - t;
        return v3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
    };
    V3 a = lerp3(texel(x0, y0), texel(x1, y0), tx);
    V3 b = lerp3(texel(x0, y1), texel(x1, y1), tx);
    V3 c = lerp3(a, b, ty);
    return v3(clamp01(c.x), clamp01(c.y), clamp01(c.z));
}

/* RaytraceRenderer.SampleAlbedo, RaytraceRenderer.cs:724-735 */
V3 SceneData::sample_albedo(const Mat &m, float u, float v) const
{
    if (m.tex < 0 || m.tex_weight <= 0.0) return m.albedo;
    double sc = m.uv_scale > 1e-6 ? m.uv_scale : 1e-6;                    /* Math.Max(1e-6, mat.UVScale); a NaN scale gives NaN in .NET: not modelled */
    float tiles = (float)sc;
    V3 tex = textures[m.tex].sample_bilinear(u * tiles, v * tiles);
    double w = m.tex_weight < 0.0 ? 0.0 : m.tex_weight > 1.0 ? 1.0 : m.tex_weight;      /* Math.Clamp */
    float t = (float)w;
    float s = 1.0f - t;
    V3 o = v3(m.albedo.x * s + tex.x * t, m.albedo.y * s + tex.y * t, m.albedo.z * s + tex.z * t);
    return v3(clamp01(o.x), clamp01(o.y), clamp01(o.z));
}

/* ======================================================================
 * Primitive intersection
 * ====================================================================== */
namespace {

/* XYRect.Hit Surfaces.cs:184-214 (axis 2), XZRect.Hit 256-286 (axis 1), YZRect.Hit 328-358 (axis 0).
 * a0,a1 / b0,b1 = the two in-plane ranges in the class's field order, k = plane offset. */
inline bool rect_hit(int axis, float a0, float a1, float b0, float b1, float k, const Ray &r,
                     float t_min, float t_max, float &t, V3 &P, V3 &N, float &U, float &V)
{
    float dir_k = axis == 2 ? r.d.z : axis == 1 ? r.d.y : r.d.x;
    float org_k = axis == 2 ? r.o.z : axis == 1 ? r.o.y : r.o.x;
    float adir = cs_abs(dir_k);
    float safe = cs_copysign(cs_max(adir, 1e-8f), dir_k);
    t = (k - org_k) / safe;
    float pa, pb;
    if (axis == 2) { pa = r.o.x + t * r.d.x; pb = r.o.y + t * r.d.y; }       /* px, py */
    else if (axis == 1) { pa = r.o.x + t * r.d.x; pb = r.o.z + t * r.d.z; }  /* px, pz */
    else { pa = r.o.y + t * r.d.y; pb = r.o.z + t * r.d.z; }                 /* py, pz */
    bool ok = adir >= 1e-8f;
    ok &= (t >= t_min) & (t <= t_max);
    ok &= (pa >= a0) & (pa <= a1) & (pb >= b0) & (pb <= b1);
    if (!ok) return false;
    float nk = cs_copysign(1.0f, -dir_k);
    if (axis == 2) { P = v3(pa, pb, k); N = v3(0.0f, 0.0f, nk); }
    else if (axis == 1) { P = v3(pa, k, pb); N = v3(0.0f, nk, 0.0f); }
    else { P = v3(k, pa, pb); N = v3(nk, 0.0f, 0.0f); }
    float inv_a = 1.0f / (a1 - a0);
    float inv_b = 1.0f / (b1 - b0);
    U = (pa - a0) * inv_a;
    V = (pb - b0) * inv_b;
    return true;
}

} // namespace

static bool grid_hit(const SceneData &S, const Grid &g, const Ray &r, float t_min, float t_max, Hit &rec, Counters &c);
static bool mesh_hit(const SceneData &S, const MeshAccel &m, const Ray &r, float t_min, float t_max, Hit &rec, Counters &c);

bool SceneData::prim_hit(int32_t pi, const Ray &r, float t_min, float t_max, Hit &rec, Counters &cnt) const
{
    const Prim &P = prims[pi];
    const float *p = P.p;
    switch (P.type) {
    case YCGE_PRIM_SPHERE: {   /* BoundedObjects.cs:31-69 */
        cnt.prim++;
        float ox = r.o.x - p[0], oy = r.o.y - p[1], oz = r.o.z - p[2];
        float dx = r.d.x, dy = r.d.y, dz = r.d.z;
        float a = dx * dx + dy * dy + dz * dz;
        float half_b = ox * dx + oy * dy + oz * dz;
        float c = ox * ox + oy * oy + oz * oz - p[3] * p[3];
        float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return false;
        float s = cs_sqrt(disc);
        float inv_a = 1.0f / a;
        float t = (-half_b - s) * inv_a;
        if (t < t_min || t > t_max) {
            t = (-half_b + s) * inv_a;
            if (t < t_min || t > t_max) return false;
        }
        float px = r.o.x + t * dx, py = r.o.y + t * dy, pz = r.o.z + t * dz;
        float inv_r = 1.0f / p[3];
        rec.t = t;
        rec.p = v3(px, py, pz);
        rec.n = v3((px - p[0]) * inv_r, (py - p[1]) * inv_r, (pz - p[2]) * inv_r);
        rec.m = eval_material(P.material, rec.p);
        rec.u = 0.0f; rec.v = 0.0f;
        rec.prim = pi; rec.sub = 0;
        return true;
    }
    case YCGE_PRIM_PLANE: {    /* Surfaces.cs:39-71 */
        cnt.prim++;
        float nx = P.normal.x, ny = P.normal.y, nz = P.normal.z;
        float dx = r.d.x, dy = r.d.y, dz = r.d.z;
        float ox = r.o.x, oy = r.o.y, oz = r.o.z;
        float denom = nx * dx + ny * dy + nz * dz;
        if (denom > -1e-6f && denom < 1e-6f) return false;
        float t = (P.ndot - (nx * ox + ny * oy + nz * oz)) / denom;
        if (t < t_min || t > t_max) return false;
        rec.t = t;
        rec.p = v3(ox + t * dx, oy + t * dy, oz + t * dz);
        rec.n = denom < 0.0f ? P.normal : P.normal_neg;
        rec.m = eval_material(P.material, rec.p);
        rec.m.reflectivity = P.reflectivity;
        rec.u = 0.0f; rec.v = 0.0f;
        rec.prim = pi; rec.sub = 0;
        return true;
    }
    case YCGE_PRIM_DISK: {     /* Surfaces.cs:108-142 */
        cnt.prim++;
        float denom = dot(P.normal, r.d);
        float adenom = cs_abs(denom);
        float safe = cs_copysign(cs_max(adenom, 1e-8f), denom);
        float t = (P.ndot - dot(P.normal, r.o)) / safe;
        float px = r.o.x + t * r.d.x, py = r.o.y + t * r.d.y, pz = r.o.z + t * r.d.z;
        float dx = px - p[0];
        float dz = pz - p[2];
        float rr = dx * dx + dz * dz;          /* quirk 4: x,z only */
        bool ok = adenom >= 1e-6f;
        ok &= (t >= t_min) & (t <= t_max);
        ok &= rr <= P.radius2;
        if (!ok) return false;
        rec.t = t;
        rec.p = v3(px, py, pz);
        rec.n = denom < 0.0f ? P.normal : -P.normal;
        rec.m = eval_material(P.material, rec.p);
        rec.m.reflectivity = P.reflectivity;
        rec.u = 0.0f; rec.v = 0.0f;
        rec.prim = pi; rec.sub = 0;
        return true;
    }
    case YCGE_PRIM_XYRECT: case YCGE_PRIM_XZRECT: case YCGE_PRIM_YZRECT: {
        cnt.prim++;
        int axis = P.type == YCGE_PRIM_XYRECT ? 2 : P.type == YCGE_PRIM_XZRECT ? 1 : 0;
        float t, U, V; V3 Pn, N;
        if (!rect_hit(axis, p[0], p[1], p[2], p[3], p[4], r, t_min, t_max, t, Pn, N, U, V)) return false;
        rec.t = t; rec.p = Pn; rec.n = N;
        rec.m = eval_material(P.material, rec.p);
        rec.m.reflectivity = P.reflectivity;
        rec.u = U; rec.v = V;
        rec.prim = pi; rec.sub = 0;
        return true;
    }
    case YCGE_PRIM_BOX: {      /* BoundedObjects.cs:82-89 (faces), 100-115 (Hit) */
        bool hit_anything = false;
        float closest = t_max;
        const float mnx = p[0], mny = p[1], mnz = p[2], mxx = p[3], mxy = p[4], mxz = p[5];
        for (int i = 0; i < 6; i++) {
            cnt.prim++;
            float t, U, V; V3 Pn, N;
            bool h;
            switch (i) {
            case 0: h = rect_hit(2, mnx, mxx, mny, mxy, mxz, r, t_min, closest, t, Pn, N, U, V); break;
            case 1: h = rect_hit(2, mnx, mxx, mny, mxy, mnz, r, t_min, closest, t, Pn, N, U, V); break;
            case 2: h = rect_hit(1, mnx, mxx, mnz, mxz, mxy, r, t_min, closest, t, Pn, N, U, V); break;
            case 3: h = rect_hit(1, mnx, mxx, mnz, mxz, mny, r, t_min, closest, t, Pn, N, U, V); break;
            case 4: h = rect_hit(0, mny, mxy, mnz, mxz, mxx, r, t_min, closest, t, Pn, N, U, V); break;
            default: h = rect_hit(0, mny, mxy, mnz, mxz, mnx, r, t_min, closest, t, Pn, N, U, V); break;
            }
            if (h) {
                hit_anything = true;
                closest = t;
                rec.t = t; rec.p = Pn; rec.n = N;
                rec.m = eval_material(P.material, rec.p);
                rec.m.reflectivity = P.reflectivity;
                rec.u = U; rec.v = V;
                rec.prim = pi; rec.sub = i;
            }
        }
        return hit_anything;
    }
    case YCGE_PRIM_CYLINDER_Y: { /* BoundedObjects.cs:148-247 */
        cnt.prim++;
        const float radius = p[3];
        float ox = r.o.x - p[0];
        float oy = r.o.y;                      /* quirk 6: Center.Y ignored */
        float oz = r.o.z - p[2];
        float dx = r.d.x, dy = r.d.y, dz = r.d.z;
        float a = dx * dx + dz * dz;
        float hit_t = kFloatMax;
        V3 hit_n = v3(0.0f, 0.0f, 0.0f);
        bool hit = false;
        int part = 0;                          /* debug capture only (YCGE_BUF_SUB_ID): 0 side, 1 top cap, 2 bottom cap */
        if (a > 1e-12f) {
            float half_b = ox * dx + oz * dz;
            float c = ox * ox + oz * oz - P.radius2;
            float disc = half_b * half_b - a * c;
            if (disc >= 0.0f) {
                float s = cs_sqrt(disc);
                float inv_a = 1.0f / a;
                float t1 = (-half_b - s) * inv_a;
                if (t1 > t_min && t1 < t_max) {
                    float y1 = oy + t1 * dy;
                    if (y1 >= P.y_min && y1 <= P.y_max) {
                        hit_t = t1;
                        float nx = (ox + t1 * dx) / radius;
                        float nz = (oz + t1 * dz) / radius;
                        hit_n = v3(nx, 0.0f, nz);
                        hit = true;
                    }
                }
                if (!hit) {
                    float t2 = (-half_b + s) * inv_a;
                    if (t2 > t_min && t2 < t_max) {
                        float y2 = oy + t2 * dy;
                        if (y2 >= P.y_min && y2 <= P.y_max) {
                            hit_t = t2;
                            float nx = (ox + t2 * dx) / radius;
                            float nz = (oz + t2 * dz) / radius;
                            hit_n = v3(nx, 0.0f, nz);
                            hit = true;
                        }
                    }
                }
            }
        }
        bool capped = p[6] != 0.0f;
        if (capped && cs_abs(dy) > 1e-8f) {
            float t_top = (P.y_max - oy) / dy;
            if (t_top > t_min && t_top < t_max) {
                float rx = ox + t_top * dx, rz = oz + t_top * dz;
                if (rx * rx + rz * rz <= P.radius2) {
                    if (t_top < hit_t) { hit_t = t_top; hit_n = v3(0.0f, 1.0f, 0.0f); hit = true; part = 1; }
                }
            }
            float t_bot = (P.y_min - oy) / dy;
            if (t_bot > t_min && t_bot < t_max) {
                float rx = ox + t_bot * dx, rz = oz + t_bot * dz;
                if (rx * rx + rz * rz <= P.radius2) {
                    if (t_bot < hit_t) { hit_t = t_bot; hit_n = v3(0.0f, -1.0f, 0.0f); hit = true; part = 2; }
                }
            }
        }
        if (!hit) return false;
        rec.t = hit_t;
        rec.p = v3(r.o.x + hit_t * dx, r.o.y + hit_t * dy, r.o.z + hit_t * dz);
        rec.n = dot(hit_n, r.d) < 0.0f ? hit_n : -hit_n;
        rec.m = eval_material(P.material, rec.p);
        rec.u = 0.0f; rec.v = 0.0f;
        rec.prim = pi; rec.sub = part;
        return true;
    }
    case YCGE_PRIM_TRIANGLE: { /* Triangle.cs:131-175 (scalar path; the SSE4.1 path 71-128 computes the same products) */
        cnt.prim++;
        float px = r.d.y * P.e2z - r.d.z * P.e2y;
        float py = r.d.z * P.e2x - r.d.x * P.e2z;
        float pz = r.d.x * P.e2y - r.d.y * P.e2x;
        float det = P.e1x * px + P.e1y * py + P.e1z * pz;
        if (cs_abs(det) < 1e-8f) return false;
        float inv_det = 1.0f / det;
        float sx = r.o.x - p[0], sy = r.o.y - p[1], sz = r.o.z - p[2];
        float u = (sx * px + sy * py + sz * pz) * inv_det;
        if (u < 0.0f || u > 1.0f) return false;
        float qx = sy * P.e1z - sz * P.e1y;
        float qy = sz * P.e1x - sx * P.e1z;
        float qz = sx * P.e1y - sy * P.e1x;
        float v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv_det;
        if (v < 0.0f || (u + v) > 1.0f) return false;
        float t = (P.e2x * qx + P.e2y * qy + P.e2z * qz) * inv_det;
        if (t < t_min || t > t_max) return false;
        rec.t = t;
        rec.p = v3(r.o.x + t * r.d.x, r.o.y + t * r.d.y, r.o.z + t * r.d.z);
        float nd = P.tnx * r.d.x + P.tny * r.d.y + P.tnz * r.d.z;
        rec.n = nd < 0.0f ? v3(P.tnx, P.tny, P.tnz) : v3(-P.tnx, -P.tny, -P.tnz);
        rec.m = eval_material(P.material, rec.p);
        rec.u = u; rec.v = v;
        rec.prim = pi; rec.sub = 0;
        return true;
    }
    case YCGE_PRIM_MESH: {     /* Mesh.cs:23-26 */
        Hit tmp = rec;
        if (!mesh_hit(*this, meshes[P.ref], r, t_min, t_max, tmp, cnt)) return false;
        rec = tmp; rec.prim = pi;
        return true;
    }
    case YCGE_PRIM_VOLUME_GRID: {
        Hit tmp = rec;
        if (!grid_hit(*this, grids[P.ref], r, t_min, t_max, tmp, cnt)) return false;
        rec = tmp; rec.prim = pi;
        return true;
    }
    default: return false;
    }
}

/* ---- BVH.BoxHitFast, BVH.cs:201-236 (NaN-propagating Max/Min) ----------- */
static inline bool scene_box_hit(const Node &n, const Ray &r, float t_min, float t_max,
                                 float inv_dx, float inv_dy, float inv_dz, float &t_near)
{
    float ox = r.o.x, oy = r.o.y, oz = r.o.z;
    float en_x = (n.min_x - ox) * inv_dx, ex_x = (n.max_x - ox) * inv_dx;
    if (en_x > ex_x) { float t = en_x; en_x = ex_x; ex_x = t; }
    float en_y = (n.min_y - oy) * inv_dy, ex_y = (n.max_y - oy) * inv_dy;
    if (en_y > ex_y) { float t = en_y; en_y = ex_y; ex_y = t; }
    float en_z = (n.min_z - oz) * inv_dz, ex_z = (n.max_z - oz) * inv_dz;
    if (en_z > ex_z) { float t = en_z; en_z = ex_z; ex_z = t; }
    float t_enter = cs_max(en_x, cs_max(en_y, en_z));
    float t_exit = cs_min(ex_x, cs_min(ex_y, ex_z));
    if (t_enter < t_min) t_enter = t_min;
    if (t_exit > t_max) t_exit = t_max;
    t_near = t_enter;
    return t_exit >= t_enter;
}

/* ---- BVH.Hit, BVH.cs:99-198 ------------------------------------------- */
bool SceneData::hit(const Ray &r, float t_min, float t_max, Hit &rec, Counters &cnt) const
{
    struct QueryLog {       /* analysis aid only */
        Counters &c; uint64_t b0, t0, l0, d0;
        explicit QueryLog(Counters &cc) : c(cc), b0(cc.box), t0(cc.tri), l0(cc.leaves), d0(cc.left_desc) {}
        ~QueryLog() { if (c.qlog2 && c.qn < c.qcap) c.qlog2[c.qn] = (uint32_t)(c.leaves - l0) | ((uint32_t)(c.left_desc - d0) << 16);
                      if (c.qlog && c.qn < c.qcap) c.qlog[c.qn++] = (uint32_t)((c.box - b0) / 2 + (c.tri - t0)) | ((uint32_t)(c.tri - t0) << 16);   /* steps | triangle tests << 16 */ }
    } query_log(cnt);
    cnt.rays++;
    if (root < 0) return false;
    float inv_dx = 1.0f / r.d.x, inv_dy = 1.0f / r.d.y, inv_dz = 1.0f / r.d.z;
    bool hit_anything = false;
    float closest = t_max;
    Hit best{};
    int stack[256];   /* C#: stackalloc int[128]; depth is validated at build */
    int sp = 0;
    stack[sp++] = root;
    cnt.box++;                                   /* root evaluation */
    while (sp > 0) {
        int ni = stack[--sp];
        float t_near;
        if (!scene_box_hit(nodes[ni], r, t_min, closest, inv_dx, inv_dy, inv_dz, t_near)) continue;  /* re-test on pop: not counted */
        int cnt_n = nodes[ni].count;
        if (cnt_n > 0) {
            int start = nodes[ni].start;
            for (int i = 0; i < cnt_n; i++) {
                int obj = leaf_obj[start + i];
                Hit tmp{};
                if (prim_hit(obj, r, t_min, closest, tmp, cnt)) {
                    hit_anything = true;
                    closest = tmp.t;
                    best = tmp;
                }
            }
        } else {
            int l = nodes[ni].left, rr = nodes[ni].right;
            float l_near = 0.0f, r_near = 0.0f;
            bool hit_l = false, hit_r = false;
            if (l >= 0) { cnt.box++; hit_l = scene_box_hit(nodes[l], r, t_min, closest, inv_dx, inv_dy, inv_dz, l_near); }
            if (rr >= 0) { cnt.box++; hit_r = scene_box_hit(nodes[rr], r, t_min, closest, inv_dx, inv_dy, inv_dz, r_near); }
            if (hit_l & hit_r) {
                if (l_near < r_near) { stack[sp++] = rr; stack[sp++] = l; }
                else { stack[sp++] = l; stack[sp++] = rr; }
            } else if (hit_l) {
                stack[sp++] = l;
            } else if (hit_r) {
                stack[sp++] = rr;
            }
        }
    }
    if (hit_anything) rec = best;
    return hit_anything;
}

bool SceneData::occluded(const Ray &r, float max_dist, Counters &cnt) const
{
    Hit rec{};
    return hit(r, 0.001f, max_dist, rec, cnt);
}

/* ---- MeshBVH.BoxHitFast, MeshBVH.cs:308-332 (compare chain ignores NaN) -- */
static inline bool mesh_box_hit(const Node &n, const Ray &r, float t_min, float t_max,
                                float inv_dx, float inv_dy, float inv_dz, int sx, int sy, int sz, float &t_near)
{
    float ox = r.o.x, oy = r.o.y, oz = r.o.z;
    float tx_en = ((sx == 0 ? n.min_x : n.max_x) - ox) * inv_dx;
    float tx_ex = ((sx == 0 ? n.max_x : n.min_x) - ox) * inv_dx;
    if (tx_en > t_min) t_min = tx_en;
    if (tx_ex < t_max) t_max = tx_ex;
    if (t_max < t_min) { t_near = t_min; return false; }
    float ty_en = ((sy == 0 ? n.min_y : n.max_y) - oy) * inv_dy;
    float ty_ex = ((sy == 0 ? n.max_y : n.min_y) - oy) * inv_dy;
    if (ty_en > t_min) t_min = ty_en;
    if (ty_ex < t_max) t_max = ty_ex;
    if (t_max < t_min) { t_near = t_min; return false; }
    float tz_en = ((sz == 0 ? n.min_z : n.max_z) - oz) * inv_dz;
    float tz_ex = ((sz == 0 ? n.max_z : n.min_z) - oz) * inv_dz;
    if (tz_en > t_min) t_min = tz_en;
    if (tz_ex < t_max) t_max = tz_ex;
    t_near = t_min;
    return t_max >= t_min;
}

/* ---- MeshBVH.TriHit, MeshBVH.cs:239-304 --------------------------------- */
static inline bool tri_hit(const MeshAccel &m, int i, const Ray &r, float t_min, float t_max, float &t, float &u, float &v)
{
    float dirx = r.d.x, diry = r.d.y, dirz = r.d.z;
    float e1x = m.e1x[i], e1y = m.e1y[i], e1z = m.e1z[i];
    float e2x = m.e2x[i], e2y = m.e2y[i], e2z = m.e2z[i];
    float ax = m.ax[i], ay = m.ay[i], az = m.az[i];
    float px = diry * e2z - dirz * e2y;
    float py = dirz * e2x - dirx * e2z;
    float pz = dirx * e2y - diry * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    const float eps = 1e-8f;
    if (det > -eps && det < eps) return false;
    float sx = r.o.x - ax, sy = r.o.y - ay, sz = r.o.z - az;
    float u_num = sx * px + sy * py + sz * pz;
    float sgn = det > 0.0f ? 1.0f : -1.0f;
    float det_abs = det * sgn;
    float u_num_s = u_num * sgn;
    if (u_num_s < 0.0f || u_num_s > det_abs) return false;
    float qx = sy * e1z - sz * e1y;
    float qy = sz * e1x - sx * e1z;
    float qz = sx * e1y - sy * e1x;
    float v_num = dirx * qx + diry * qy + dirz * qz;
    float v_num_s = v_num * sgn;
    float uv_sum_s = u_num_s + v_num_s;
    if (v_num_s < 0.0f || uv_sum_s > det_abs) return false;
    float t_num = e2x * qx + e2y * qy + e2z * qz;
    float t_num_s = t_num * sgn;
    float t_min_scaled = t_min * det_abs;
    float t_max_scaled = t_max * det_abs;
    if (t_num_s < t_min_scaled || t_num_s > t_max_scaled) return false;
    float inv_det = 1.0f / det;
    t = t_num * inv_det;
    u = u_num * inv_det;
    v = v_num * inv_det;
    return true;
}

/* ---- MeshBVH.Hit, MeshBVH.cs:132-236 ------------------------------------ */
static bool mesh_hit(const SceneData &S, const MeshAccel &m, const Ray &r, float t_min, float t_max, Hit &rec, Counters &cnt)
{
    if (m.root < 0) return false;
    float inv_dx = 1.0f / r.d.x, inv_dy = 1.0f / r.d.y, inv_dz = 1.0f / r.d.z;
    int sx = inv_dx < 0.0f ? 1 : 0, sy = inv_dy < 0.0f ? 1 : 0, sz = inv_dz < 0.0f ? 1 : 0;
    bool hit_anything = false;
    float closest = t_max;
    int best_tri = -1; float best_t = 0, best_u = 0, best_v = 0;
    int stack[256];   /* C#: stackalloc int[64]; depth is validated at build (orc_scene_upload) */
    int sp = 0;
    stack[sp++] = m.root;
    cnt.box++;
    while (sp > 0) {
        int ni = stack[--sp];
        float t_near;
        if (!mesh_box_hit(m.nodes[ni], r, t_min, closest, inv_dx, inv_dy, inv_dz, sx, sy, sz, t_near)) continue;
        int cnt_n = m.nodes[ni].count;
        if (cnt_n > 0) {
            int start = m.nodes[ni].start;
            cnt.leaves++;
            for (int i = 0; i < cnt_n; i++) {
                int tri = m.leaf_tri[start + i];
                float t_hit, u, v;
                cnt.tri++;
                if (tri_hit(m, tri, r, t_min, closest, t_hit, u, v)) {
                    closest = t_hit;
                    hit_anything = true;
                    best_tri = tri; best_t = t_hit; best_u = u; best_v = v;
                }
            }
        } else {
            int l = m.nodes[ni].left, rr = m.nodes[ni].right;
            float l_near = 0.0f, r_near = 0.0f;
            bool hit_l = false, hit_r = false;
            if (l >= 0) { cnt.box++; hit_l = mesh_box_hit(m.nodes[l], r, t_min, closest, inv_dx, inv_dy, inv_dz, sx, sy, sz, l_near); }
            if (rr >= 0) { cnt.box++; hit_r = mesh_box_hit(m.nodes[rr], r, t_min, closest, inv_dx, inv_dy, inv_dz, sx, sy, sz, r_near); }
            if (hit_l & hit_r) {
                if (l_near < r_near) { stack[sp++] = rr; stack[sp++] = l; cnt.left_desc++; }
                else { stack[sp++] = l; stack[sp++] = rr; }
            } else if (hit_l) {
                stack[sp++] = l; cnt.left_desc++;
            } else if (hit_r) {
                stack[sp++] = rr;
            }
        }
    }
    if (!hit_anything) return false;
    /* MeshBVH.cs:177-185 */
    int tri = best_tri;
    rec.t = best_t;
    rec.p = v3(r.o.x + best_t * r.d.x, r.o.y + best_t * r.d.y, r.o.z + best_t * r.d.z);
    float ndotd = m.nx[tri] * r.d.x + m.ny[tri] * r.d.y + m.nz[tri] * r.d.z;
    rec.n = ndotd < 0.0f ? v3(m.nx[tri], m.ny[tri], m.nz[tri]) : v3(-m.nx[tri], -m.ny[tri], -m.nz[tri]);
    rec.m = S.eval_material(m.tri_mat[tri], rec.p);
    rec.u = best_u; rec.v = best_v;
    rec.sub = tri;
    return true;
}

/* ---- VolumeGrid.Slab / RayAabb, VolumeGrid.cs:319-355 ------------------- */
static inline bool grid_slab(float ro, float rd, float mn, float mx, float &t_enter, float &t_exit, int axis, int &enter_axis)
{
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

/* VolumeGrid.EdgeDistance / IsWireOnFace, VolumeGrid.cs:256-299 (fp64) */
static inline double edge_distance(double v, double v0, double v1)
{
    double a = v - v0; double b = v1 - v;
    if (a < 0.0) a = 0.0; if (b < 0.0) b = 0.0;
    return cs_min_d(a, b);
}
static bool is_wire_on_face(const Grid &g, V3 p, int ix, int iy, int iz, int axis)
{
    /* minCorner.X + ix * voxelSize.X is an fp32 expression widened afterwards */
    double x0 = (double)(g.min_corner.x + (float)ix * g.voxel_size.x); double x1 = x0 + (double)g.voxel_size.x;
    double y0 = (double)(g.min_corner.y + (float)iy * g.voxel_size.y); double y1 = y0 + (double)g.voxel_size.y;
    double z0 = (double)(g.min_corner.z + (float)iz * g.voxel_size.z); double z1 = z0 + (double)g.voxel_size.z;
    if (axis == 0) {
        double dy = edge_distance((double)p.y, y0, y1);
        double dz = edge_distance((double)p.z, z0, z1);
        double w = (double)(g.wire_width_frac * cs_min(g.voxel_size.y, g.voxel_size.z));
        return dy <= w || dz <= w;
    } else if (axis == 1) {
        double dx = edge_distance((double)p.x, x0, x1);
        double dz = edge_distance((double)p.z, z0, z1);
        double w = (double)(g.wire_width_frac * cs_min(g.voxel_size.x, g.voxel_size.z));
        return dx <= w || dz <= w;
    } else {
        double dx = edge_distance((double)p.x, x0, x1);
        double dy = edge_distance((double)p.y, y0, y1);
        double w = (double)(g.wire_width_frac * cs_min(g.voxel_size.x, g.voxel_size.y));
        return dx <= w || dy <= w;
    }
}

/* ---- VolumeGrid.Hit, VolumeGrid.cs:99-231 ------------------------------- */
static bool grid_hit(const SceneData &S, const Grid &g, const Ray &r, float t_min, float t_max, Hit &rec, Counters &cnt)
{
    cnt.prim++;
    float min_x = g.min_corner.x, min_y = g.min_corner.y, min_z = g.min_corner.z;
    float size_x = g.voxel_size.x, size_y = g.voxel_size.y, size_z = g.voxel_size.z;
    float max_x = min_x + (float)g.nx * size_x, max_y = min_y + (float)g.ny * size_y, max_z = min_z + (float)g.nz * size_z;

    int enter_axis = -1;
    float t_enter = -kInf, t_exit = kInf;
    if (!grid_slab(r.o.x, r.d.x, min_x, max_x, t_enter, t_exit, 0, enter_axis)) return false;
    if (!grid_slab(r.o.y, r.d.y, min_y, max_y, t_enter, t_exit, 1, enter_axis)) return false;
    if (!grid_slab(r.o.z, r.d.z, min_z, max_z, t_enter, t_exit, 2, enter_axis)) return false;
    if (!(t_exit >= cs_max(0.0f, t_enter))) return false;
    float t = t_enter; if (t < t_min) t = t_min; if (t > t_max || t > t_exit) return false;

    const float eps = 1e-6f;
    t += eps;
    float ox = r.o.x, oy = r.o.y, oz = r.o.z;
    float dx = r.d.x, dy = r.d.y, dz = r.d.z;
    float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;

    int ix = cs_f2i(cs_floor((px - min_x) / size_x)); if (ix < 0) ix = 0; else if (ix >= g.nx) ix = g.nx - 1;
    int iy = cs_f2i(cs_floor((py - min_y) / size_y)); if (iy < 0) iy = 0; else if (iy >= g.ny) iy = g.ny - 1;
    int iz = cs_f2i(cs_floor((pz - min_z) / size_z)); if (iz < 0) iz = 0; else if (iz >= g.nz) iz = g.nz - 1;

    int step_x = dx > 0.0f ? 1 : dx < 0.0f ? -1 : 0;
    int step_y = dy > 0.0f ? 1 : dy < 0.0f ? -1 : 0;
    int step_z = dz > 0.0f ? 1 : dz < 0.0f ? -1 : 0;
    float inv_dx = step_x == 0 ? 0.0f : 1.0f / dx;
    float inv_dy = step_y == 0 ? 0.0f : 1.0f / dy;
    float inv_dz = step_z == 0 ? 0.0f : 1.0f / dz;
    float next_vx = min_x + (step_x > 0 ? (float)(ix + 1) * size_x : (float)ix * size_x);
    float next_vy = min_y + (step_y > 0 ? (float)(iy + 1) * size_y : (float)iy * size_y);
    float next_vz = min_z + (step_z > 0 ? (float)(iz + 1) * size_z : (float)iz * size_z);
    float t_max_x = step_x == 0 ? kInf : (next_vx - ox) * inv_dx;
    float t_max_y = step_y == 0 ? kInf : (next_vy - oy) * inv_dy;
    float t_max_z = step_z == 0 ? kInf : (next_vz - oz) * inv_dz;
    float t_delta_x = step_x == 0 ? kInf : cs_abs(size_x * inv_dx);
    float t_delta_y = step_y == 0 ? kInf : cs_abs(size_y * inv_dy);
    float t_delta_z = step_z == 0 ? kInf : cs_abs(size_z * inv_dz);

    int last_axis = enter_axis < 0 ? (t_max_x <= t_max_y && t_max_x <= t_max_z ? 0 : t_max_y <= t_max_z ? 1 : 2) : enter_axis;

    float wire_max2 = g.wire_max_distance <= 0.0f ? -1.0f : g.wire_max_distance * g.wire_max_distance;
    float dir_len2 = dx * dx + dy * dy + dz * dz;

    while (t <= t_exit && t <= t_max) {
        if ((uint32_t)ix < (uint32_t)g.nx && (uint32_t)iy < (uint32_t)g.ny && (uint32_t)iz < (uint32_t)g.nz) {
            cnt.vox++;
            int idx = g.index_of(ix, iy, iz);
            int mat_id = g.mat[idx];
            if (mat_id > 0) {
                int meta_id = g.meta[idx];
                int normal_axis = last_axis;           /* never < 0 here (line 150) */
                float hit_t = cs_max(t, t_min);
                V3 n;
                if (normal_axis == 0) n = v3(step_x > 0 ? -1.0f : 1.0f, 0.0f, 0.0f);
                else if (normal_axis == 1) n = v3(0.0f, step_y > 0 ? -1.0f : 1.0f, 0.0f);
                else n = v3(0.0f, 0.0f, step_z > 0 ? -1.0f : 1.0f);
                V3 hit_point = r.o + r.d * hit_t;      /* Ray.At */
                bool within_wire = false;
                if (g.wireframe && wire_max2 >= 0.0f) {
                    float dist2 = hit_t * hit_t * dir_len2;
                    within_wire = dist2 <= wire_max2;
                }
                /* The centre-block highlight (VolumeGrid.cs:176-187) is DEAD CODE under this renderer, at every console size: it needs
                 * a query with |screenV - 0.5| <= 1e-6 (IsCenterUV, :286-289); the tracer passes vCenter = (py + 0.5f) / hiH
                 * (RaytraceRenderer.cs:204-205) with hiH = fbH * 2 * ss (:86-87, :119) - EVEN for every framebuffer, so vCenter misses
                 * 0.5 by at least 0.5 / hiH (> 1e-6 below half a million rows) - and every other caller of Scene.Hit passes (0, 0)
                 * (VolumeScenes.cs:288-525).  centerValid never becomes true; the wire colour is always WireColor.  (KAT:
                 * tests/test_oracle_kats.py::test_center_block_highlight_is_unreachable.) */
                int mi = g.default_material;
                for (size_t k = 0; k < g.lookup.size(); k++)
                    if (g.lookup[k].mat_id == mat_id && g.lookup[k].meta_id == meta_id) { mi = g.lookup[k].material; break; }
                Mat m = S.eval_material(mi, hit_point);
                if (g.wireframe && within_wire && is_wire_on_face(g, hit_point, ix, iy, iz, normal_axis))
                    m.albedo = v3(0.0f, 0.0f, 0.0f);   /* WireColor */
                rec.t = hit_t; rec.p = hit_point; rec.n = n; rec.m = m; rec.u = 0.0f; rec.v = 0.0f;
                rec.sub = ix + g.nx * (iy + g.ny * iz);
                return true;
            }
        }
        if (t_max_x <= t_max_y && t_max_x <= t_max_z) { ix += step_x; t = t_max_x; t_max_x += t_delta_x; last_axis = 0; }
        else if (t_max_y <= t_max_z) { iy += step_y; t = t_max_y; t_max_y += t_delta_y; last_axis = 1; }
        else { iz += step_z; t = t_max_z; t_max_z += t_delta_z; last_axis = 2; }
        if ((uint32_t)ix >= (uint32_t)g.nx || (uint32_t)iy >= (uint32_t)g.ny || (uint32_t)iz >= (uint32_t)g.nz) break;
    }
    return false;
}

} // namespace orc
