// ycge_accel.h — host-side BVH construction for the MI355X ray-trace core.
//
// Builds, on the host and once per scene upload, the two trees the reference
// traverses — the scene-level BVH over Scene.Objects and the per-mesh triangle
// BVH — with the SAME topology, node order and leaf order as the reference
// builders (ConsoleGame/RayTracing/Objects/BVH.cs:258-459 and
// Objects/MeshBVH.cs:371-576), because traversal order is part of the result
// (exact-t ties are accepted, so the later-visited primitive wins).  The
// builder here is an explicit-stack, index-permutation formulation, not the
// reference's recursion over item structs; tests compare its output node for
// node with the oracle's.
#pragma once
#include <cstdint>
#include <vector>

#include "ycge_device.h"

namespace ycge {

// reference-format node (BVH.cs:11-20 / MeshBVH.cs:18-27), kept for ycge_read_accel
struct RefNode {
    float mn[3], mx[3];
    int32_t left, right, start, count;
};

struct BoundsSoA {
    std::vector<float> mn[3], mx[3], c[3];   // per item: box min/max and centroid, per axis
    size_t size() const { return mn[0].size(); }
    void resize(size_t n) { for (int a = 0; a < 3; a++) { mn[a].resize(n); mx[a].resize(n); c[a].resize(n); } }
};

struct BuiltTree {
    std::vector<RefNode> nodes;      // pre-order, as the reference numbers them
    std::vector<int32_t> leaf_index; // leafObjIndex / leafTriIndex
    int32_t root = -1;
    int32_t max_depth = 0;           // root = depth 1
    int32_t sort_fallbacks = 0;
};

enum class TreeFlavour { Scene, Mesh };

// Build one tree over `items`.
void build_tree(const BoundsSoA &items, TreeFlavour flavour, BuiltTree &out);

// triangle bounds + centroids as MeshBVH's ctor computes them (MeshBVH.cs:47-60,349-363)
void triangle_items(const float *tris9, int32_t n, BoundsSoA &out);

// Convert a reference-format tree to the paired-children GPU layout.
//   node_kind / leaf_kind : REF_SCENE_NODE/REF_SCENE_LEAF or REF_MESH_NODE/REF_MESH_LEAF
//   node_base / leaf_base : global offsets added to internal-node indices and leaf starts
//   leaf_slot             : optional map from a leaf's start in leaf_index to its start in the device array
//                           (mesh leaves start on triangle-pair records); identity when null
// Appends to gnodes; returns the reference for the root.
uint32_t to_gpu_nodes(const BuiltTree &t, uint32_t node_kind, uint32_t leaf_kind, uint32_t node_base, uint32_t leaf_base,
                      int leaf_count_bits, std::vector<GNode> &gnodes, const std::vector<uint32_t> *leaf_slot = nullptr);

} // namespace ycge
