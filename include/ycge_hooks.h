/* ycge_hooks.h - what libycge_hip.so exports BESIDE the drop-in boundary (include/ycge.h).
 *
 * NOT part of the boundary: nothing here is bound by the C# host (bindings/csharp/), nothing here is covered by
 * YCGE_ABI_VERSION, any of it may change between builds.  These are the handles tests/ and profiles/ hold the
 * library by - the host-side builders and schedules called WITHOUT a GPU (the CPU suite compares them with the
 * oracle node for node), read-outs of profiling instantiations, fault and layout probes - plus a few functions that
 * cross from the host translation units into the kernel ones (csrc/*.cpp -> csrc/*.hip).  The kernel launchers
 * themselves (ycge_launch_*: one per kernel, plain C linkage so that the host .cpp files need no HIP compiler) are
 * exported as well and are not listed one by one.  tests/test_host_cpu.py holds this list to `nm -D` of the library.
 */
#ifndef YCGE_HOOKS_H
#define YCGE_HOOKS_H

#include "ycge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- layout probe: sizeof the C structs as the library was compiled (0 ycge_vec3, 1 material, 2 prim, 3 mesh, 4 voxel_lookup, 5 grid,
 * 6 light, 7 scene, 8 config, 9 frame_stats, 10 flight_info) - the ctypes and C# mirrors are held to it */
size_t ycge_abi_sizeof(int32_t which);

/* ---- host-side builders and schedules, no GPU needed (tests/test_host_cpu.py: against the oracle, bit for bit) */
/* Objects/BVH.cs:258-459 (flavour 0) / MeshBVH.cs:371-576 (flavour 1) over given boxes and centroids; returns the node count */
int ycge_host_build_tree(const float *bounds, const float *centroids, int32_t n, int32_t flavour, void *nodes_out, int32_t *leaf_out,
                         int32_t *stats_out /* [root, max_depth, sort_fallbacks] */);
int ycge_host_build_mesh(const float *tris9, int32_t n, void *nodes_out, int32_t *leaf_out, int32_t *stats_out);
/* the device records of one mesh as ycge_scene_upload lays them out (64-byte nodes, 96-byte triangle pairs; with the cooperative walk's treelets) */
int ycge_host_mesh_arena(const float *tris9, int32_t n, void *out, int64_t capacity_bytes, uint32_t *root_ref_out);
int ycge_host_mesh_arena_treelets(const float *tris9, int32_t n, void *out, int64_t capacity_bytes, uint32_t *root_ref_out, uint32_t *tl_offset_out);
/* the in-place A-trous iteration's level schedule (RaytraceRenderer.cs:718), its bands, the row-parity split, the LDS window's width */
int ycge_host_inplace_schedule(int32_t w, int32_t h, int32_t step, uint32_t *pixels_out, uint32_t *offsets_out, int32_t capacity);
int ycge_host_inplace_bands(int32_t w, int32_t h, int32_t step, int32_t rows_per_band, uint32_t *entries_out, int64_t entries_capacity,
                            uint32_t *offsets_out, int64_t offsets_capacity, int32_t *info_out);
int ycge_host_split_bands(int32_t w, int32_t h, int32_t step, int32_t *row_band_out, int32_t *desc_out, int32_t desc_capacity, int32_t *max_px_out);
int ycge_host_band_window_width(int32_t w, int32_t h, int32_t step, int32_t rows_per_band, int32_t K, int32_t G);
/* the halo lists of the tile-resident form (what ycge_halo_counts counts), rank by rank */
int ycge_host_halo_layout(int32_t hiW, int32_t hiH, int32_t rank, int32_t world, int64_t *send_counts, int64_t *recv_counts,
                          uint32_t *send_px, uint32_t *recv_px, int64_t capacity);

/* ---- read-outs for tests and profiles (a context, a destination, a capacity; YCGE_OK or an error code) */
int ycge_debug_scene_bvh_stats(ycge_ctx *c, int64_t *out6);          /* how ycge_scene_update_objects built the tree: device / fallback / host builds, us, sort fallbacks, depth */
int ycge_debug_device_bvh(const float *bounds, const float *centroids, int32_t n, void *nodes_out, int32_t *leaf_out, uint32_t *result_out, void *build_out);   /* k_scene_bvh_build alone */
int ycge_debug_read_walk_tree(ycge_ctx *c, void *gnodes_out, void *walk_out, int32_t capacity_nodes, int32_t *grid_owner_out, int32_t n_grids, uint32_t *root_and_limit_out);
int ycge_debug_read_post_progress(ycge_ctx *c, uint32_t *dst, size_t n_words);               /* k_atrous_stream's per-band records (profiles/post_bands.py) */
int ycge_debug_read_wave_prof(ycge_ctx *c, unsigned long long *dst, size_t n_u64);            /* per-wavefront begin / end / steps of a profiling build (profiles/mega_prof.py) */
int ycge_debug_read_coop_stats(ycge_ctx *c, uint64_t out[16]);                               /* -DYCGE_DBG_COOPSTAT builds */
int ycge_debug_read_batch_stats(ycge_ctx *c, uint64_t out[64]);
int ycge_debug_resident_loop(ycge_ctx *c, int32_t frames, double *period_ms, double *issue_ms);   /* a rank's tile-resident ring driven from C (profiles/rank_times.py) */
int ycge_debug_is_page_locked(const void *p, size_t bytes);          /* the verdict ycge_render_frame takes on a caller's SDR buffer: 1 page-locked over its whole range */
int ycge_debug_throw(ycge_ctx *c, int32_t kind);                     /* throws INSIDE an export (1 std::bad_alloc, 2 std::runtime_error, 3 an int, 4 std::length_error, 5 std::system_error; 0 nothing): the exception barrier's test */
int ycge_debug_fail_allocation(int64_t nth);                         /* lib/var_faultinject.so ONLY (-DYCGE_FAULT_INJECTION): the library's n-th allocation from now throws std::bad_alloc */

/* ---- host translation units -> kernel translation units (sizes and knobs of what the .hip files define) */
size_t ycge_wf_sizes(int which);
size_t ycge_post_state_bytes(void);
size_t ycge_exposure_scratch_bytes(int w, int h, int step);
size_t ycge_bvh_build_scratch_bytes(int n);
int ycge_atrous_persist_resident(int groups_per_pass, int split, int level_handover, int profile);
void ycge_atrous_duo_pad_lds(int bytes);
void ycge_peer_worker_main(ycge_ctx *c, ycge_ctx *p);                /* a peer device's thread function (started by ycge_create) */

#ifdef __cplusplus
}
#endif
#endif
