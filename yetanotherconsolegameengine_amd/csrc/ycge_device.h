// ycge_device.h — data layout in HBM, shared by the host side (ycge_host.cpp,
// ycge_accel.cpp) and the gfx950 kernels (ycge_kernels.hip).
//
// Every record is sized so that one lane fetches it with whole 16-byte loads
// (global_load_dwordx4): lanes of a wavefront diverge during traversal, so the
// unit of HBM/L2 traffic is the record, not a wave-wide row.
//
//   GNode   64 B  an INTERNAL node of either BVH with BOTH child boxes inline:
//                 one fetch per visit replaces the reference's three
//                 (own box on pop + two child boxes; BVH.cs:126-166).
//   GTriPair 96 B TWO leaf triangles, components interleaved, in LEAF ORDER
//                 (leafTriIndex applied at upload): A, e1, e2 + original indices;
//                 the unit normal is recomputed from e1 x e2 on the final hit
//                 only (MeshBVH.cs:93-97 ops).  Mesh GNodes and GTriPairs share
//                 one arena in depth-first order, addressed in 32-byte units.
//   GPrim   64 B  one Scene.Objects entry with its ctor-derived constants.
//   GMaterial 80 B, GGrid 96 B + 1 byte per voxel (bricked 8^3, Morton inside).
#pragma once
#include <stdint.h>

#ifndef YCGE_EXPERIMENTS
#define YCGE_EXPERIMENTS 0          // 1: the measured-and-rejected kernel forms of csrc/experiments/ are compiled in (lib/var_experiments.so; their parity tests load that build)
#endif

namespace ycge {

// ---- child / stack reference encoding (uint32) ---------------------------
// bits 31..29 kind, bits 28..0 payload
enum : uint32_t {
    REF_SCENE_NODE = 0u,   // payload = index into scene_nodes
    REF_SCENE_LEAF = 1u,   // payload = (start << 3) | count        (count 1..4, BVH.cs:7)
    REF_MESH_NODE = 2u,    // payload = (32-byte unit of the GNode in the mesh arena) << 4
    REF_MESH_LEAF = 3u,    // payload = (unit of the leaf's first GTriPair << 4) | triangles left in the leaf (1..15; MeshBVH.cs:14 caps a leaf at 8)
    REF_PRIM = 4u,         // payload = index into prims
    REF_WALK_NODE = 5u,    // payload = index into walk_nodes (SceneDev::walk_nodes) | YCGE_WALK_IN_ORDER for a leaf node: left child first whatever the distances
    REF_GRID = 6u,         // payload = index into grids (walk tree only)
    REF_NONE = 7u
};
#define YCGE_REF(kind, payload) (((uint32_t)(kind) << 29) | (uint32_t)(payload))
#define YCGE_REF_KIND(r) ((r) >> 29)
#define YCGE_REF_PAYLOAD(r) ((r) & 0x1fffffffu)
#define YCGE_REF_NONE_VALUE 0xffffffffu
#define YCGE_WALK_IN_ORDER 0x10000000u      // in a REF_WALK_NODE payload (the rest is the index)

// device-side build of the scene-level BVH (ycge_bvh_build.hip): what the kernel hands back to the host
#define YCGE_TRACE_BATCH_MAX 8          // frames in one launch of k_trace_batch: their parameter and output records ride in the kernel arguments
#define YCGE_WALK_LEAF_NODES 6          // walk tree: entries set aside per leaf child (a leaf holds at most 7 objects: 6 nodes)
#define YCGE_BVH_DEV_MAX_ITEMS 2560     // one workgroup keeps the item order and its node queue in LDS
#define YCGE_BVH_DEV_MIN_ITEMS_DEFAULT 1400     // below this ycge_scene_update_objects builds on the host: the measured crossover of the two builders
struct BvhBuildResult {
    uint32_t root_ref;
    float root_min[3], root_max[3];
    int32_t n_nodes, n_inner, max_depth, fallback;      // fallback: deeper than the reference's stack - the host builder redoes the tree and reports it
    uint32_t sorts;                                      // how often the reference's Array.Sort case ran (BVH.cs:389,419)
    uint32_t pad[4];
};
static_assert(sizeof(BvhBuildResult) == 64, "BvhBuildResult");

// Plane order: (x y)(z Z)(X Y) per child, lower case = min, upper case = max.  Every 8-byte pair is then an
// (x, y) or a (z, z) pair, so the twelve slab products of a visit are six packed operations against just two
// pairings of the ray's origin / reciprocal direction (mesh_walk).
struct alignas(16) GNode {
    float lmin_x, lmin_y, lmin_z, lmax_z;
    float lmax_x, lmax_y, rmin_x, rmin_y;
    float rmin_z, rmax_z, rmax_x, rmax_y;
    uint32_t lref, rref;        // child references
    uint32_t pad[2];
};
static_assert(sizeof(GNode) == 64, "GNode must be 64 B");

// A TREELET: the boxes and references of the 2 children, 4 grandchildren and 8 great-grandchildren of one internal mesh node, one
// 32-byte slot each, heap order (slot 0 = left child, 1 = right; the children of slot b are 2b+2 and 2b+3).  Sixteen lanes fetch
// it in ONE round trip and test the fourteen boxes side by side: the near-first descent through up to three levels is then bit
// logic over the hit masks (coop_walk).  A copy of what the GNode records hold - the regular walk never reads it.
struct alignas(16) GTreeSlot {
    float mn[3], mx_x;
    float mx_y, mx_z;
    uint32_t ref;               // the child's reference as its parent's GNode holds it (REF_MESH_NODE / REF_MESH_LEAF)
    uint32_t valid;             // 0: no such descendant (its parent position is a leaf)
};
static_assert(sizeof(GTreeSlot) == 32, "GTreeSlot must be 32 B");
#define YCGE_TL_SLOTS 14
#define YCGE_TL_BYTES_PER_UNIT 256u      // a GNode owns two 32-byte units: 512 bytes of treelet (14 slots = 448)

// Leaf triangles, TWO to a record with the components interleaved (slot 0, slot 1): a 16-byte fetch then lands
// each component pair in an aligned register pair, and the Moller-Trumbore test of both triangles runs as packed
// operations (v_pk_mul_f32 / v_pk_add_f32) - one traversal step per two triangles.  A leaf starts on a record
// boundary; the second slot of a leaf's last record is all zeros when its count is odd (det = 0, and the walk
// masks it by the count anyway).  A triangle is addressed as record * 2 + slot.
struct alignas(16) GTriPair {
    float ax[2], ay[2];
    float az[2], e1x[2];
    float e1y[2], e1z[2];
    float e2x[2], e2y[2];
    float e2z[2];
    int32_t orig[2];            // index in the mesh's input triangle order
    int32_t material[2];        // material index (per-triangle or the mesh's)
    int32_t pad[2];
};
static_assert(sizeof(GTriPair) == 96, "GTriPair must be 96 B");

// prim types: identical numbering to ycge_prim_type in include/ycge.h
struct alignas(16) GPrim {
    int32_t type;
    int32_t material;
    int32_t ref;                // mesh / grid index
    float reflectivity;         // ctor override for plane/disk/rect/box (Surfaces.cs:65-66)
    // per-type constants (derived on the host exactly as the C# ctors do):
    //  SPHERE     cx cy cz radius
    //  PLANE      nx ny nz (normalised) ndotPoint
    //  DISK       cx cy cz nx ny nz radius2 ndotCenter
    //  XY/XZ/YZ   a0 a1 b0 b1 k
    //  BOX        min xyz max xyz
    //  CYLINDER_Y cx cz radius radius2 yMin yMax capped
    //  TRIANGLE   A xyz e1 xyz e2 xyz n xyz
    //  VOLUME_GRID  box of the grid's solid voxels, one voxel of margin: lo xyz, hi xyz (GGrid::solid_lo / solid_hi)
    float p[12];
};
static_assert(sizeof(GPrim) == 64, "GPrim must be 64 B");

struct alignas(16) GMaterial {
    int32_t kind;
    float albedo[3];
    float albedo_b[3];
    float checker_scale;
    float reflectivity;
    float emission[3];
    float transparency;
    float ior;
    float trans_color[3];
    // kind == YCGE_MAT_TEXTURED (SampleAlbedo, RaytraceRenderer.cs:724-735): texture index (-1: weight <= 0, no texture), and the two
    // doubles as SampleAlbedo narrows them: tiles = (float)Math.Max(1e-6, UVScale), t = (float)Math.Clamp(TextureWeight, 0, 1)
    int32_t tex;
    float tex_tiles, tex_t;
};
static_assert(sizeof(GMaterial) == 80, "GMaterial must be 80 B");

struct alignas(16) GMesh {
    float root_min[3], root_max[3];
    uint32_t root_ref;          // REF_MESH_NODE or REF_MESH_LEAF
    uint32_t pad;
};
static_assert(sizeof(GMesh) == 32, "GMesh must be 32 B");

struct alignas(16) GGrid {
    int32_t nx, ny, nz;
    int32_t nbx, nby, nbz;
    float min_corner[3];
    float voxel_size[3];        // already max(1e-6, v) (VolumeGrid.cs:76)
    uint32_t cell_offset;       // byte offset of this grid's cells in grid_cells
    int32_t wireframe;
    float wire_width_frac;
    float wire_max_distance;
    uint32_t lut_offset;        // index of this grid's first entry in grid_lut (cell code -> material)
    uint32_t brick_mask_lo, brick_mask_hi;   // bit b set = brick b holds a solid voxel (grids of <= 64 bricks, e.g. 32^3 chunks)
    int32_t has_brick_mask;
    // World-space box of the grid's SOLID voxels, one voxel wider on every side (solid_hi[0] < solid_lo[0]: no solid voxel).  A ray
    // that misses it between tmin and tmax cannot hit anything here - the voxel walk's rounding errors are ~1e-5 of a voxel - so the
    // timed kernels do not enter the grid at all (grid_cull); the counting kernels walk it as the reference does.
    float solid_lo[3], solid_hi[3];
    float cull_t_limit;         // ... as long as the ray leaves that box at t <= this (the walk's accumulated rounding stays below half a voxel; host: ycge_scene_upload)
    uint32_t pad;
};
static_assert(sizeof(GGrid) == 112, "GGrid must be 112 B");

struct GLight {
    float pos[3];
    float color[3];
    float intensity;
    // 1.0f: Intensity == 0 with a finite colour.  Its contribution is `radiance += throughput * (f * nDotL * (Color * (0 / dist2)) * tr)` =
    // radiance + (+-0) = radiance, whatever the shadow query answers (RaytraceRenderer.cs:592-602; the reference traces the ray all the
    // same - SURVEY appendix A, quirk 8).  The TIMED kernels skip the query and the addition (light_is_dark); the counting ones trace it.
    float dark;
};

#define YCGE_SPLIT_TOP_DEFAULT 32     // blocks at the head of k_trace's schedule that go in 4 parts of 16 pixels (k_cost_scatter)
#define YCGE_TIMED_STEP_SLOTS_LG 10  // counters[8 + 8 i], i < 1024: lane steps of the timed kernel instances, spread over cache lines (flush_work)
#define YCGE_COUNTER_WORDS (8 + 8 * (1 << YCGE_TIMED_STEP_SLOTS_LG))
#define YCGE_LDS_STACK_LEVELS 12    // levels of a lane's traversal stack kept in LDS; deeper ones go to the HBM spill area
#define YCGE_TRAVERSAL_STACK 96    // >= scene depth + 4 leaf prims + mesh depth, checked at upload
#ifndef YCGE_COST_FRAMES
#define YCGE_COST_FRAMES 4          // a block's schedule cost is its largest cost over this many frames
#endif
#define YCGE_SCHEDULE_SLACK 2u      // k_trace grid = blocks x this: room for the parts of split blocks
#define YCGE_FAN_CAP_DEFAULT 2048u   // k_trace_fan: at most this many blocks of the schedule's head
#define YCGE_REFILL_STEPS_DEFAULT 0
#define YCGE_POST_BAND_ROWS_DEFAULT 8 // rows per band of the in-place A-trous iteration (at least 2 x step)
#define YCGE_POST_GROUPS_DEFAULT 16    // pixels per pass of the banded in-place A-trous iteration (workgroup = 32 x this many threads)
#define YCGE_POST_K_DEFAULT 8         // levels per launch of the banded in-place A-trous iteration
#define YCGE_TILE_W 32
#define YCGE_TILE_H 8
#define YCGE_SLAB_FLOATS 11        // hdr rgb, albedo rgb, normal xyz, depth, sky

// everything the trace kernel needs besides per-frame camera values
struct SceneDev {
    const GNode *scene_nodes;
    const uint32_t *scene_leaf_prims;   // leafObjIndex
    const uint8_t *mesh_arena;          // GNode / GTriPair records of all meshes in depth-first order; byte offset = (ref & 0x1ffffff0) << 1
    const GPrim *prims;
    const GMaterial *materials;
    const GMesh *meshes;
    const GGrid *grids;
    const uint8_t *grid_cells;
    const int32_t *grid_lut;
    float scene_root_min[3], scene_root_max[3];
    uint32_t scene_root_ref;            // REF_NONE when Objects is empty
    int32_t n_lights;
    const GLight *lights;               // n_lights records in HBM (indexed per lane)
    float ambient[3];                   // Ambient.Color
    float ambient_intensity;
    float bg_top[3], bg_bottom[3];
    int32_t is_volume_scene;
    int32_t any_transparent;            // some material has Transparency > 0
    int32_t analytic_only;              // Scene.Objects holds neither a mesh nor a voxel grid: the scene tree is walked by analytic_walk (ycge_rt.hip.h)
    const uint32_t *tex_pixels;         // every texture's RGBA32 pixels (Renderer/Texture.cs:15), back to back
    const int32_t *tex_info;            // per texture: {first pixel, width, height, 0}
    int32_t any_textured;               // some material samples a texture (SampleAlbedo, RaytraceRenderer.cs:724-735)
    // Treelets for the wave-cooperative walk of sparse wavefronts (coop_walk, ycge_rt.hip.h): byte offset, inside the mesh arena's
    // allocation, of the treelet region - the treelet of the internal node at 32-byte unit u is at tl_offset + u * YCGE_TL_BYTES_PER_UNIT.
    // 0 = none (no mesh, or the region would not fit 32-bit offsets): the regular walk serves everything.
    uint32_t tl_offset;
    // occlusion queries against a mesh breadth-first from a shared work list (mesh_anyhit_bfs, ycge_anyhit.hip.h) when at most this many lanes
    // of the wavefront ask (YCGE_BFS; 0: the ordered walk serves them all)
    uint32_t anyhit_bfs;
    // The WALK TREE of a world of voxel grids (timed kernels; k_scene_walk, ycge_bvh_build.hip): the scene nodes again, every leaf of
    // 2..7 objects opened into nodes of its own - so that one kind of step, a node visit, serves the whole way down to a grid.
    //   [0, n)      the scene nodes, child boxes as they are (entry order and the entry test are the reference's), child references
    //               rewritten: node j -> REF_WALK_NODE j; a leaf of one object -> that object; a leaf of more -> its first leaf node
    //   [n, ...)    leaf nodes: the leaf's objects in index order, halved until single; a child's box is the union of the SOLID-voxel
    //               boxes below it (GGrid::solid_lo / solid_hi; everything for an object that is no grid); references to them carry
    //               YCGE_WALK_IN_ORDER: left child first - a leaf's objects are asked in index order (BVH.cs:139-149), not by distance
    //               (in the reference, not in the record: the record's last words would be a second fetch behind the box tests).
    // A grid is referenced as REF_GRID (its index); grid_owner[grid] = the object that holds it (hit_prim).  Grids without a solid
    // voxel are left out where a leaf has others.  What the walk skips is what the object step would cull one by one
    // (solid_box_missed); visit order, hence every hit, is unchanged.  A ray is sent down this tree only if it leaves the scene's
    // root box before walk_t_limit (the smallest GGrid::cull_t_limit: every verdict below holds), else down scene_nodes as before.
    // Null: no walk tree (no grid, the root is a leaf, two objects share a grid, YCGE_NO_WALK_TREE).
    const GNode *walk_nodes;
    const int32_t *grid_owner;
    uint32_t walk_root_ref;
    float walk_t_limit;
    unsigned long long *dbg_counters;   // profiling builds (-DYCGE_DBG_COOPSTAT): statistics of the cooperative walk, else unused
};

struct FrameParams {
    int32_t hiW, hiH;
    int64_t frame;
    int32_t frame_idx;
    float rot_x, rot_y;                 // jitterRotX/Y
    float cam_pos[3];
    float fwd[3], right[3], up[3];
    float half_w, half_h;
    uint64_t seed_salt;
    float eps;
    float mirror_threshold;
    float sigma_rad;
    float on_a, on_b;                   // Oren-Nayar A, B of sigma (RaytraceRenderer.cs:823-825)
    int32_t max_mirror_bounces, max_refractions, diffuse_bounces;
    // tile partition
    int32_t tiles_x, tiles_y;
    int32_t rank, world_size;
    int32_t n_owned_tiles;
    // optional blockIdx.x -> owned-tile ordinal table (null = identity).  Workgroup b runs on XCD b % 8 (observed
    // dispatch order); the table can hand each XCD vertical image strips for L2 affinity.  Off by default:
    // measured 2.1x slower than round-robin on config 4 (load balance of the heavy tiles matters more).
    const uint32_t *tile_order;
};

struct TaaParams {
    int32_t w, h;
    float alpha;
    int32_t radius;
    float pad_lum;
    int32_t reset;
};

// TemporalBlendWithClamp INSIDE the trace launch (round 6, measured and rejected: csrc/experiments/ycge_taa_in_trace.hip.h; honoured only by
// -DYCGE_EXPERIMENTS=1 builds): a block that finishes counts itself in at the blocks of its 3 x 3 neighbourhood, and whoever completes a
// neighbourhood resolves its centre block.  block_ctr == null - always, in the product: TAA is a launch of its own behind the trace.
struct TaaFuse {
    uint32_t *block_ctr;                // [8x8 block of the frame] finished blocks of its neighbourhood, itself included; monotonic: a frame adds the neighbourhood's size
    uint32_t *part_ctr;                 // [block] finished parts of a split block; the last part puts it back to 0
    float *hist, *prev_normal, *prev_depth;
    uint8_t *prev_sky;
    TaaParams T;
};

struct TraceOut {
    // full-frame buffers (row-major x + y*hiW); with several GPUs only the owned tiles are written
    float *current_hdr;     // 3 f32 / px
    float *g_albedo;
    float *g_normal;
    float *g_depth;
    uint8_t *sky;
    // debug capture (may be null)
    float *rays;
    int32_t *prim_id;
    int32_t *sub_id;
    float *hit_t;
    uint64_t *rng_state;
    // traversal-stack overflow area [level][global lane] and refraction path stack [slot][field][global lane]
    void *stack_spill;                  // uint2 entries {ref, tNear}
    uint32_t stack_lanes;
    uint32_t lane_base;                 // first column of this launch (k_trace_fan runs beside k_trace)
    float *path_stack;
    // per-wavefront profile of k_wf_primary (COUNT variant; may be null): 4 x u64 {start, end, node iters, leaf phases}
    unsigned long long *wave_prof;
    int32_t wave_prof_stage;            // 0 = k_wf_primary, 1 = k_wf_extend of round 1
    // k_trace scheduling feedback: block_cost[b] = traversal loop iterations of 8x8 block b's wavefront(s) this frame
    // (atomicMax; cleared by k_cost_scatter); block_order / n_order = THIS frame's schedule built from the previous
    // frame's costs (entries: see k_trace), or null = one wavefront per block in index order
    uint32_t *block_cost;
    const uint32_t *block_order;
    const uint32_t *n_order;
    // frames in flight: the LAST workgroup of the launch stores placed_value here when it starts - every workgroup before it has a
    // place then, and the next frame's trace (which waits for the value, hipStreamWaitValue32) gets what this launch leaves free
    uint32_t *placed_flag;
    uint32_t placed_value;
    const uint32_t *n_fan;              // the first *n_fan schedule entries are traced by k_trace_fan (null or 0: none)
    // traversal counters (may be null): rays, box, tri, prim, vox
    unsigned long long *counters;
    TaaFuse taa;                        // the single-launch kernels only
};

// planes a peer device copies from its own frame buffers into rank 0's (one process, several GPUs): its tiles only
struct PushPlanes {
    const uint8_t *src[12];
    uint8_t *dst[12];
    int32_t bytes_per_pixel[12];
    int32_t n;
};

} // namespace ycge
