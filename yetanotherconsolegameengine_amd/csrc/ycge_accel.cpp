// ycge_accel.cpp — host-side construction of the scene BVH and mesh BVHs.
// See ycge_accel.h for the contract (same topology / node order / leaf order as
// reference Objects/BVH.cs:258-459 and Objects/MeshBVH.cs:371-576).
#include "ycge_accel.h"
#include "ycge_keysort.h"

#include <cstring>

#include "ycge_math.h"

namespace ycge {

namespace {

constexpr int kBins = 16;           // SAH_Bins (BVH.cs:8, MeshBVH.cs:15)

struct Box3 {
    float mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; a++) { mn[a] = YCGE_INF; mx[a] = -YCGE_INF; } }
    // Surround(): plain compare-assign per component (BVH.cs:252-256)
    void grow(const float omn[3], const float omx[3])
    {
        for (int a = 0; a < 3; a++) { if (omn[a] < mn[a]) mn[a] = omn[a]; }
        for (int a = 0; a < 3; a++) { if (omx[a] > mx[a]) mx[a] = omx[a]; }
    }
    float area() const      // SurfaceArea (BVH.cs:462-466)
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.0f * (dx * dy + dx * dz + dy * dz);
    }
};

struct Task {
    int32_t start, count, depth, parent;
    bool is_right;
};

} // namespace

void build_tree(const BoundsSoA &it, TreeFlavour flavour, BuiltTree &out)
{
    out = BuiltTree{};
    const int n = (int)it.size();
    if (n == 0) return;
    const int leaf_max = flavour == TreeFlavour::Scene ? 4 : 8;    // TargetLeafSize (BVH.cs:7, MeshBVH.cs:14)
    std::vector<int32_t> ord(n);
    for (int i = 0; i < n; i++) ord[i] = i;
    out.nodes.reserve(2 * (size_t)n);
    out.leaf_index.reserve(n);

    auto item_box = [&](int32_t id, float mn[3], float mx[3]) {
        for (int a = 0; a < 3; a++) { mn[a] = it.mn[a][id]; mx[a] = it.mx[a][id]; }
    };

    std::vector<Task> todo;
    todo.push_back(Task{0, n, 1, -1, false});
    while (!todo.empty()) {
        Task t = todo.back();
        todo.pop_back();
        const int s = t.start, cnt = t.count;
        const int my = (int)out.nodes.size();
        if (t.parent >= 0) { if (t.is_right) out.nodes[t.parent].right = my; else out.nodes[t.parent].left = my; }
        else out.root = my;
        if (t.depth > out.max_depth) out.max_depth = t.depth;

        if (cnt <= leaf_max) {
            RefNode leaf{};
            item_box(ord[s], leaf.mn, leaf.mx);
            for (int i = 1; i < cnt; i++) {
                float mn[3], mx[3];
                item_box(ord[s + i], mn, mx);
                for (int a = 0; a < 3; a++) if (mn[a] < leaf.mn[a]) leaf.mn[a] = mn[a];
                for (int a = 0; a < 3; a++) if (mx[a] > leaf.mx[a]) leaf.mx[a] = mx[a];
            }
            leaf.left = leaf.right = -1;
            leaf.start = (int32_t)out.leaf_index.size();
            leaf.count = cnt;
            for (int i = 0; i < cnt; i++) out.leaf_index.push_back(ord[s + i]);
            out.nodes.push_back(leaf);
            continue;
        }

        // centroid bounds
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; a++) cmin[a] = cmax[a] = it.c[a][ord[s]];
        for (int i = s + 1; i < s + cnt; i++)
            for (int a = 0; a < 3; a++) {
                float c = it.c[a][ord[i]];
                if (c < cmin[a]) cmin[a] = c;
                if (c > cmax[a]) cmax[a] = c;
            }
        float ext[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
        int axis = 0;
        if (ext[1] > ext[0] && ext[1] >= ext[2]) axis = 1; else if (ext[2] > ext[0] && ext[2] >= ext[1]) axis = 2;

        int split_bin = -1, best_axis = axis;
        float best_cost = YCGE_INF;
        for (int ax = 0; ax < 3; ax++) {
            const float extent = ext[ax];
            if (!(extent > 0.0f)) continue;
            const float origin = cmin[ax];
            const float inv_extent = 1.0f / extent;
            int counts[kBins] = {0};
            Box3 bin[kBins];
            for (int b = 0; b < kBins; b++) bin[b].reset();
            for (int i = s; i < s + cnt; i++) {
                const int32_t id = ord[i];
                int b = cs_f2i((it.c[ax][id] - origin) * inv_extent * (float)(kBins - 1));
                if (b < 0) b = 0;
                if (b >= kBins) b = kBins - 1;
                counts[b]++;
                float mn[3], mx[3];
                item_box(id, mn, mx);
                bin[b].grow(mn, mx);
            }
            int lcount[kBins], rcount[kBins];
            float larea[kBins], rarea[kBins];
            Box3 acc_box;
            acc_box.reset();
            int acc = 0;
            for (int b = 0; b < kBins; b++) {
                if (counts[b] > 0) acc_box.grow(bin[b].mn, bin[b].mx);
                acc += counts[b];
                lcount[b] = acc;
                larea[b] = acc_box.area();
            }
            acc_box.reset();
            acc = 0;
            for (int b = kBins - 1; b >= 0; b--) {
                if (counts[b] > 0) acc_box.grow(bin[b].mn, bin[b].mx);
                acc += counts[b];
                rcount[b] = acc;
                rarea[b] = acc_box.area();
            }
            for (int b = 0; b + 1 < kBins; b++) {
                const int lc = lcount[b], rc = rcount[b + 1];
                if (lc == 0 || rc == 0) continue;
                const float cost = larea[b] * (float)lc + rarea[b + 1] * (float)rc;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; split_bin = b; }
            }
        }

        int mid;
        KeySorter<int32_t> sorter{ord.data(), it.c[best_axis].data()};
        if (split_bin < 0) {
            sorter.sort(s, cnt);
            out.sort_fallbacks++;
            mid = s + (cnt >> 1);
        } else {
            const float *key = it.c[best_axis].data();
            float origin, inv_extent;
            bool zero_guard = false;
            if (flavour == TreeFlavour::Scene) {
                // BVH.cs:394-396: the partition pass re-derives origin/extent from the first and the
                // last item currently in the range, not from the binning bounds
                origin = key[ord[s]];
                const float extent = key[ord[s + cnt - 1]] - origin;
                inv_extent = extent != 0.0f ? 1.0f / extent : 0.0f;
                zero_guard = true;
            } else {
                origin = cmin[best_axis];                       // MeshBVH.cs:511-513
                inv_extent = 1.0f / ext[best_axis];
            }
            int i0 = s, i1 = s + cnt - 1;
            while (i0 <= i1) {
                int b0;
                if (zero_guard && !(inv_extent != 0.0f)) b0 = 0;
                else b0 = cs_f2i((key[ord[i0]] - origin) * inv_extent * (float)(kBins - 1));
                if (b0 <= split_bin) i0++;
                else { int32_t tmp = ord[i0]; ord[i0] = ord[i1]; ord[i1] = tmp; i1--; }
            }
            mid = i0;
            if (mid == s || mid == s + cnt) {
                sorter.sort(s, cnt);
                out.sort_fallbacks++;
                mid = s + (cnt >> 1);
            }
        }

        RefNode inner{};
        inner.left = inner.right = -1;
        inner.start = 0;
        inner.count = 0;
        out.nodes.push_back(inner);
        // left subtree is numbered first (pre-order): push right, then left
        todo.push_back(Task{mid, s + cnt - mid, t.depth + 1, my, true});
        todo.push_back(Task{s, mid - s, t.depth + 1, my, false});
    }

    // interior bounds bottom-up: children always have larger indices than their parent
    for (int i = (int)out.nodes.size() - 1; i >= 0; i--) {
        RefNode &nd = out.nodes[i];
        if (nd.count > 0) continue;
        const RefNode &L = out.nodes[nd.left];
        const RefNode &R = out.nodes[nd.right];
        for (int a = 0; a < 3; a++) {
            nd.mn[a] = cs_min(L.mn[a], R.mn[a]);    // MathF.Min / MathF.Max, BVH.cs:438-443
            nd.mx[a] = cs_max(L.mx[a], R.mx[a]);
        }
    }
}

void triangle_items(const float *t9, int32_t n, BoundsSoA &out)
{
    out.resize(n);
    const float eps = 1e-4f;    // MeshBVH.cs:351
    for (int i = 0; i < n; i++) {
        const float *t = t9 + 9 * (size_t)i;
        for (int a = 0; a < 3; a++) {
            const float A = t[a], B = t[3 + a], C = t[6 + a];
            const float mn = cs_min(A, cs_min(B, C)) - eps;
            const float mx = cs_max(A, cs_max(B, C)) + eps;
            out.mn[a][i] = mn;
            out.mx[a][i] = mx;
            out.c[a][i] = 0.5f * (mn + mx);
        }
    }
}

uint32_t to_gpu_nodes(const BuiltTree &t, uint32_t node_kind, uint32_t leaf_kind, uint32_t node_base, uint32_t leaf_base,
                      int leaf_count_bits, std::vector<GNode> &gnodes, const std::vector<uint32_t> *leaf_slot)
{
    if (t.root < 0) return YCGE_REF_NONE_VALUE;
    // internal nodes keep their relative (pre-)order so a subtree stays contiguous in memory
    std::vector<int32_t> inner_index(t.nodes.size(), -1);
    uint32_t n_inner = 0;
    for (size_t i = 0; i < t.nodes.size(); i++)
        if (t.nodes[i].count == 0) inner_index[i] = (int32_t)n_inner++;
    auto ref_of = [&](int32_t ni) -> uint32_t {
        const RefNode &nd = t.nodes[ni];
        if (nd.count > 0) {
            const uint32_t start = leaf_slot ? (*leaf_slot)[(size_t)nd.start] : (uint32_t)nd.start;
            return YCGE_REF(leaf_kind, ((start + leaf_base) << leaf_count_bits) | (uint32_t)nd.count);
        }
        return YCGE_REF(node_kind, node_base + (uint32_t)inner_index[ni]);
    };
    const size_t base = gnodes.size();
    gnodes.resize(base + n_inner);
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const RefNode &nd = t.nodes[i];
        if (nd.count > 0) continue;
        GNode &g = gnodes[base + inner_index[i]];
        const RefNode &L = t.nodes[nd.left];
        const RefNode &R = t.nodes[nd.right];
        g.lmin_x = L.mn[0]; g.lmin_y = L.mn[1]; g.lmin_z = L.mn[2]; g.lmax_x = L.mx[0]; g.lmax_y = L.mx[1]; g.lmax_z = L.mx[2];
        g.rmin_x = R.mn[0]; g.rmin_y = R.mn[1]; g.rmin_z = R.mn[2]; g.rmax_x = R.mx[0]; g.rmax_y = R.mx[1]; g.rmax_z = R.mx[2];
        g.lref = ref_of(nd.left);
        g.rref = ref_of(nd.right);
        g.pad[0] = g.pad[1] = 0;
    }
    return ref_of(t.root);
}

} // namespace ycge
