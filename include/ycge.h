/*
 * ycge.h — C-ABI of the MI355X ray-trace core for YetAnotherConsoleGameEngine.
 *
 * This is the drop-in boundary for ONE path of the reference: the per-pixel
 * ray-trace loop that RaytraceEntity drives through the private seam
 * RaytraceEntity.IConsoleRenderer (reference ConsoleGame/RaytraceEntity.cs:12-18:
 * SetCamera / SetFov / TryFlipAndBlit / Resize).  A third IConsoleRenderer
 * wrapper on the C# side P/Invokes the entry points below (binding source in
 * INTEGRATION.md).  Everything is plain C: PODs, pointers and sizes; no C++
 * or torch types cross this line.
 *
 * Conventions
 *   - every function returns YCGE_OK (0) or a negative ycge_status; nothing
 *     throws across the ABI (the reference throws on misuse, e.g.
 *     Scenes/Scene.cs:73; the C# wrapper turns codes back into exceptions);
 *   - the caller owns every host pointer it passes; the library copies during
 *     the call; the context owns all device memory;
 *   - calls on one context come from one thread (reference: the Terminal loop
 *     thread, Renderer/Terminal.cs:136-176) except ycge_set_camera, which is
 *     safe against a concurrent ycge_render_frame (reference lock(camLock),
 *     RayTracing/RaytraceRenderer.cs:142-147).
 *
 * All float data is IEEE binary32, little endian.  "hi-res grid" below is the
 * reference's trace grid hiW = fbW*ss, hiH = fbH*2*ss
 * (RayTracing/RaytraceRenderer.cs:83-87); per-pixel buffers are row-major,
 * index = x + y*hiW (RayTracing/Fast2D.cs:21-24).
 */
#ifndef YCGE_H
#define YCGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YCGE_ABI_VERSION 9
#define YCGE_MAX_DEVICES 8

typedef enum ycge_status {
    YCGE_OK = 0,
    YCGE_ERR_INVALID_ARG = -1,   /* null pointer, bad size, bad enum            */
    YCGE_ERR_NO_SCENE = -2,      /* render before scene upload (Scene.cs:73)    */
    YCGE_ERR_DEVICE = -3,        /* HIP runtime error; see ycge_last_error      */
    YCGE_ERR_UNSUPPORTED = -4,   /* feature outside the path (textures, ...)    */
    YCGE_ERR_OUT_OF_MEMORY = -5,
    YCGE_ERR_STACK_DEPTH = -6,   /* BVH deeper than the reference's fixed stacks
                                    (BVH.cs:118 = 128, MeshBVH.cs:150 = 64)     */
    YCGE_ERR_NO_DEVICE_CODE = -7,/* HIP kernels missing / no gfx950 device      */
    YCGE_ERR_INTERNAL = -8       /* a C++ exception other than std::bad_alloc (which is
                                    YCGE_ERR_OUT_OF_MEMORY) was stopped at the boundary:
                                    every export is a function-try-block, nothing unwinds
                                    into the caller (csrc/ycge_ctx.h: abi_catch)        */
} ycge_status;

typedef struct ycge_vec3 { float x, y, z; } ycge_vec3;

/* ---------------------------------------------------------------- materials
 * The reference passes opaque delegates Func<Vec3,Vec3,float,Material>
 * (Objects/Surfaces.cs:64) / Func<int,int,Material> (Objects/VolumeGrid.cs:189).
 * Only three shapes occur in its scene builders (Scenes/Scenes.cs:408-428,
 * Scenes/VoxelMaterialPalette.cs:29-98): constant, emissive constant, checker.
 * Material doubles (Material.cs:7-18) are narrowed to float here; the tracer
 * only ever reads them through (float) casts or compares against 0.0.
 */
typedef enum ycge_material_kind {
    YCGE_MAT_CONSTANT = 0,       /* Solid / Emissive / plain Material struct    */
    YCGE_MAT_CHECKER = 1,        /* Scenes.cs:418-428: parity of floor(x/s)+floor(z/s) */
    YCGE_MAT_TEXTURED = 2        /* a constant Material with DiffuseTexture != null: SampleAlbedo blends the albedo with a
                                    bilinear sample of the texture at the hit's (U, V) (RaytraceRenderer.cs:724-735 ->
                                    Renderer/Texture.cs:142-163 for a static texture, Texture.cs:113-140 for a LIVE one - camera /
                                    video frames, see ycge_texture.frame_bytes_per_pixel and ycge_scene_update_texture) */
} ycge_material_kind;

/* Renderer/Texture.cs:15,22-23: `pixels[y * width + x]` as RGBA32.ToInt() packs them (RGBA32.cs:14-31: byte 0 = r,
 * 1 = g, 2 = b, 3 = a of the little-endian int). */
typedef struct ycge_texture {
    int32_t width, height;       /* both >= 1                                   */
    const uint32_t *pixels;      /* width * height (static textures)            */
    /* A LIVE texture (`new Texture(IFrameReader, useRGBA, flipU, flipV)`, Renderer/Texture.cs:51-66: camera / video frames): bytes per
     * pixel of the frames, 3 = BGR or 4 = BGRA (dynamicBytesPerPixel); 0 = a static texture.  SampleBilinear then takes its other
     * branch (Texture.cs:113-140: flips, neighbours CLAMPED at the last row / column, bytes in B, G, R order, no per-lerp Saturate). */
    int32_t frame_bytes_per_pixel;
    int32_t flip_u, flip_v;
    /* the frame GetCurrentFramePtr() shows at upload: width * height * frame_bytes_per_pixel bytes (NULL: black until the first
     * ycge_scene_update_texture).  `pixels` is not read for a live texture. */
    const uint8_t *frame;
} ycge_texture;

typedef struct ycge_material {
    int32_t kind;                /* ycge_material_kind                          */
    ycge_vec3 albedo;            /* constant albedo, or checker colour A        */
    ycge_vec3 albedo_b;          /* checker colour B                            */
    float checker_scale;
    float specular;              /* carried for symmetry; never read (quirk 2)  */
    float reflectivity;
    ycge_vec3 emission;
    float transparency;
    float index_of_refraction;
    ycge_vec3 transmission_color;
    /* YCGE_MAT_TEXTURED only (Material.cs:16-18; doubles as in the reference: SampleAlbedo compares and clamps them as
     * doubles before it narrows) */
    int32_t texture;             /* index into ycge_scene.textures              */
    int32_t reserved;
    double texture_weight;       /* Material.TextureWeight; <= 0 means no texture */
    double uv_scale;             /* Material.UVScale                            */
} ycge_material;

/* --------------------------------------------------------------- primitives
 * One record per entry of Scene.Objects, in Objects order (order is part of
 * the result: BVH leaf order and tie-breaking depend on it).
 */
typedef enum ycge_prim_type {
    YCGE_PRIM_SPHERE = 0,        /* p = cx,cy,cz,radius              BoundedObjects.cs:7-69   */
    YCGE_PRIM_PLANE = 1,         /* p = px,py,pz, nx,ny,nz (n as given; normalised inside) Surfaces.cs:8-71 */
    YCGE_PRIM_DISK = 2,          /* p = cx,cy,cz, nx,ny,nz, radius   Surfaces.cs:73-142       */
    YCGE_PRIM_XYRECT = 3,        /* p = x0,x1,y0,y1,z                Surfaces.cs:144-214      */
    YCGE_PRIM_XZRECT = 4,        /* p = x0,x1,z0,z1,y                Surfaces.cs:216-286      */
    YCGE_PRIM_YZRECT = 5,        /* p = y0,y1,z0,z1,x                Surfaces.cs:288-358      */
    YCGE_PRIM_BOX = 6,           /* p = min xyz, max xyz             BoundedObjects.cs:72-116 */
    YCGE_PRIM_CYLINDER_Y = 7,    /* p = cx,cy,cz,radius,yMin,yMax,capped(0/1) BoundedObjects.cs:118-247 */
    YCGE_PRIM_TRIANGLE = 8,      /* p = A xyz, B xyz, C xyz          Objects/Triangle.cs      */
    YCGE_PRIM_MESH = 9,          /* ref = index into ycge_scene.meshes                        */
    YCGE_PRIM_VOLUME_GRID = 10   /* ref = index into ycge_scene.grids                         */
} ycge_prim_type;

typedef struct ycge_prim {
    int32_t type;                /* ycge_prim_type                              */
    int32_t material;            /* index into ycge_scene.materials (unused for mesh / grid) */
    int32_t ref;                 /* mesh / grid index                           */
    int32_t reserved;
    float p[12];
    /* Plane / Disk / Rects / Box overwrite the material func's Specular and
     * Reflectivity with their ctor arguments (Surfaces.cs:65-66,136-137,
     * 208-209,280-281,352-353).  Ignored for the other types. */
    float specular;
    float reflectivity;
} ycge_prim;

/* Triangle soup of one Mesh (RayTracing/Mesh.cs:16-21): 9 floats per triangle
 * (A, B, C), already transformed as MeshLoader.FromObj leaves them. */
typedef struct ycge_mesh {
    const float *triangles;      /* 9 * n_triangles floats                      */
    int32_t n_triangles;
    int32_t material;            /* material of every triangle                  */
    const int32_t *tri_material; /* optional per-triangle material, or NULL     */
} ycge_mesh;

/* (matId, metaId) -> material; stands in for Func<int,int,Material>. */
typedef struct ycge_voxel_lookup {
    int32_t mat_id;
    int32_t meta_id;
    int32_t material;            /* index into ycge_scene.materials             */
} ycge_voxel_lookup;

/* One VolumeGrid (Objects/VolumeGrid.cs:55-93).  cells = the ctor's
 * (int,int)[nx,ny,nz] in its native memory order: pair index
 * (ix*ny + iy)*nz + iz, Item1 = matId, Item2 = metaId.
 * Limits (YCGE_ERR_UNSUPPORTED beyond them): fewer than 2^30 cells per grid,
 * fewer than 2^23 8x8x8 bricks across any face, at most 255 distinct
 * (matId, metaId) pairs per grid, 4 GiB of voxel bytes per scene. */
typedef struct ycge_grid {
    int32_t nx, ny, nz;
    ycge_vec3 min_corner;
    ycge_vec3 voxel_size;
    const int32_t *cells;        /* 2 * nx*ny*nz int32                          */
    const ycge_voxel_lookup *lookup;
    int32_t n_lookup;
    int32_t default_material;    /* lookup miss -> this material; <0 = error    */
    int32_t wireframe;           /* ctor default true                           */
    float wire_width_fraction;   /* ctor default 0.06f                          */
    float wire_max_distance;     /* ctor default 16.0f                          */
} ycge_grid;

typedef struct ycge_light {      /* Objects/PointLight.cs:3-15 */
    ycge_vec3 position;
    ycge_vec3 color;
    float intensity;
} ycge_light;

typedef struct ycge_scene {      /* Scenes/Scene.cs:12-24 */
    const ycge_material *materials; int32_t n_materials;
    const ycge_prim *prims;         int32_t n_prims;
    const ycge_mesh *meshes;        int32_t n_meshes;
    const ycge_grid *grids;         int32_t n_grids;
    const ycge_light *lights;       int32_t n_lights;
    ycge_vec3 ambient_color;  float ambient_intensity;
    ycge_vec3 background_top;
    ycge_vec3 background_bottom;
    int32_t is_volume_scene;     /* `scene is VolumeScene` (RaytraceRenderer.cs:761) */
    int32_t n_textures;
    const ycge_texture *textures;   /* what YCGE_MAT_TEXTURED materials index */
    /* Scene.HasDynamicTextures (Scenes/Scene.cs:30): the host rewrites texture pixels between frames (video, camera), so
     * every frame (re)initialises the TAA history - `resetHistory = ... || scene.HasDynamicTextures`, RaytraceRenderer.cs:171. */
    int32_t has_dynamic_textures;
} ycge_scene;

/* ------------------------------------------------------------------- config */
typedef struct ycge_config {
    int32_t abi_version;         /* YCGE_ABI_VERSION                            */
    int32_t fb_width;            /* chexel columns  (Framebuffer.Width)         */
    int32_t fb_height;           /* chexel rows     (Framebuffer.Height)        */
    int32_t super_sample;        /* ss >= 1                                     */
    float fov_deg;
    int32_t device;              /* HIP device ordinal                          */
    /* framebuffer tiling across GPUs: this context traces the 32x8-pixel
     * tiles with tile_id % world_size == rank (one process per GPU). */
    int32_t rank;
    int32_t world_size;
    /* constants of RaytraceRenderer.cs:31-43,65,218 (same defaults) */
    int32_t diffuse_bounces;     /* 1 */
    int32_t max_mirror_bounces;  /* 2 */
    int32_t max_refractions;     /* 2 */
    float mirror_threshold;      /* 0.9f */
    float eps;                   /* 1e-4f */
    uint64_t seed_salt;          /* 0x9E3779B97F4A7C15 */
    float taa_alpha;             /* 0.01f */
    float motion_trans_reset;    /* 0.0025f */
    float motion_rot_reset;      /* 0.0025f */
    float diffuse_sigma_deg;     /* 25.0f */
    int32_t taa_clamp_radius;    /* 1 */
    float taa_luminance_pad;     /* 0.10f */
    /* denoise / tonemap stage after TAA (RaytraceRenderer.cs:221-227) */
    int32_t atrous_iterations;   /* 3 */
    float atrous_c_phi, atrous_n_phi, atrous_z_phi, atrous_a_phi; /* 3, 0.35, 2, 0.20 */
    int32_t capture_debug;       /* also keep rays / primId / hitT buffers      */
    int32_t count_work;          /* keep per-frame traversal counters           */
    /* tiled frame (world_size > 1): 1 = slabs carry the albedo plane (11 floats
     * per pixel; the denoise stage needs it), 0 = lean slabs of 8 floats per
     * pixel (hdr, normal, depth, sky: what TAA needs) - 27 % less all-gather
     * traffic; ycge_resolve_gathered then refuses an SDR buffer. Default 1.  */
    int32_t slab_albedo;
    /* One process, several GPUs (the reference makes ONE TryFlipAndBlit call from one process,
     * RaytraceEntity.cs:230): n_devices >= 2 makes ycge_render_frame drive devices[0 .. n_devices) itself -
     * tiles dealt round-robin over them, every device traces its share, the peers write their tiles straight
     * into devices[0]'s frame buffers over xGMI (no staging slabs), TAA and the post stage run on devices[0].
     * rank / world_size must then be 0 / 1.  n_devices 0 or 1 = the single GPU `device`. */
    int32_t n_devices;
    int32_t devices[YCGE_MAX_DEVICES];
    /* ApplyAtrousDenoise's buffer swap (RaytraceRenderer.cs:718) makes iteration 1 run IN PLACE - scan-order dependent, a dependent chain of
     * W/2 + 3H/2 pixel levels that costs 1.9 ms of a 1080p frame (8.8 ms at 3840x2160) however it is scheduled.  1 (default) reproduces it
     * bit for bit.  0 WAIVES it (SURVEY 8-f1 allows "reproduce or explicitly waive"): iteration 1 reads A and writes B like any other
     * iteration - the denoiser the C# text reads like, fully parallel (~0.2 ms).  The two differ by what INTEGRATION.md states (a filter
     * tap that sees a neighbour's already-filtered value instead of its unfiltered one); everything up to TAA is unaffected. */
    int32_t atrous_inplace_exact;
    /* tile-resident form (ycge_trace_tiles_resident): frame sets in the ring = tiled traces that may be in flight at a time, 2..15; 0 = 2 */
    int32_t tile_ring;
    /* One process, several GPUs (n_devices >= 1): how the devices' tiles come together on devices[0] (ABI 9).
     * YCGE_EXCHANGE_PEER_PUSH (0, default): the peers write their tiles straight into devices[0]'s frame buffers (k_push_tiles, no collective).
     * YCGE_EXCHANGE_RCCL (1): every device packs its tiles into a slab, ONE ncclAllGather over the devices' in-process communicators
     * (ncclCommInitAll; librccl.so is dlopen'ed, the library does not link it) reassembles the frame, devices[0] un-permutes it and runs
     * TAA and the post stage - the all-gather of SURVEY 8(e) behind the one TryFlipAndBlit call a single-process host makes.  Same pixels.
     * Where librccl.so or its symbols are missing the context falls back to the peer push; ycge_exchange_query says which is in use.
     * n_devices = 1 with YCGE_EXCHANGE_RCCL is a world of one (the frame goes through slab, all-gather and un-permute): what a one-GPU
     * box can test.  RCCL refuses two ranks on one device: devices[] must then be distinct. */
    int32_t multi_device_exchange;
} ycge_config;

typedef enum ycge_exchange { YCGE_EXCHANGE_PEER_PUSH = 0, YCGE_EXCHANGE_RCCL = 1 } ycge_exchange;

typedef struct ycge_frame_stats {
    int64_t frame;               /* frameCounter after the increment            */
    int32_t history_reset;       /* TAA history was (re)initialised this frame  */
    int32_t fan_blocks;          /* 8x8 blocks of this frame's schedule that went to k_trace_fan (0 = kernel not launched) */
    double trace_ms;             /* device time of ray-gen + trace              */
    double taa_ms;
    double post_ms;              /* denoise + exposure + tonemap/downsample     */
    double total_ms;             /* wall time of the call                       */
    /* traversal counters; valid when config.count_work != 0 */
    uint64_t n_rays;             /* Scene.Hit + Scene.Occluded calls            */
    uint64_t n_box;              /* AABB evaluations as root or as child        */
    uint64_t n_tri;              /* MeshBVH.TriHit calls                        */
    uint64_t n_prim;             /* analytic primitive tests (box = 6 rects)    */
    uint64_t n_vox;              /* DDA cells visited                           */
    float exposure;              /* ToneMapper.EffectiveExposure                */
    float exposure_serial_chunks;/* diagnostics: 512-term chunks of the exposure sum that took the one-by-one path */
    /* of n_rays: shadow queries towards lights of Intensity == 0 (counted with config.count_work).  The reference traces them although
     * their contribution is a zero whatever they find (RaytraceRenderer.cs:586-602); the timed kernels do not. */
    uint64_t n_rays_dark;
    /* tiles traced by each device of this frame (one entry for a single-GPU context; devices[] order) */
    int32_t n_devices_traced;
    int32_t device_tiles[YCGE_MAX_DEVICES];
} ycge_frame_stats;

typedef enum ycge_buffer {
    YCGE_BUF_RAYS = 0,           /* 6 f32/px: origin, dir      (capture_debug)  */
    YCGE_BUF_PRIM_ID = 1,        /* i32/px: Objects index of primary hit, -1 miss (capture_debug) */
    YCGE_BUF_SUB_ID = 2,         /* i32/px: triangle index / box face / voxel cell (capture_debug) */
    YCGE_BUF_HIT_T = 3,          /* f32/px: primary hit t, FLT_MAX miss (capture_debug) */
    YCGE_BUF_CURRENT_HDR = 4,    /* 3 f32/px */
    YCGE_BUF_G_ALBEDO = 5,       /* 3 f32/px */
    YCGE_BUF_G_NORMAL = 6,       /* 3 f32/px */
    YCGE_BUF_G_DEPTH = 7,        /* f32/px   */
    YCGE_BUF_SKY_MASK = 8,       /* u8/px    */
    YCGE_BUF_TAA_HISTORY = 9,    /* 3 f32/px */
    YCGE_BUF_PREV_NORMAL = 10,   /* 3 f32/px */
    YCGE_BUF_PREV_DEPTH = 11,    /* f32/px   */
    YCGE_BUF_PREV_SKY = 12,      /* u8/px    */
    YCGE_BUF_DENOISED = 13,      /* 3 f32/px */
    YCGE_BUF_RNG_STATE = 14      /* u64/px: Rng state when the pixel finished (capture_debug) */
} ycge_buffer;

/* flattened acceleration structures, for parity tests of the builders */
typedef enum ycge_accel {
    YCGE_ACCEL_SCENE_NODES = 0,  /* 10 x 4 B per node: min xyz, max xyz, left, right, start, count (BVH.cs:11-20) */
    YCGE_ACCEL_SCENE_LEAF_INDEX = 1, /* i32 leafObjIndex                        */
    YCGE_ACCEL_MESH_NODES = 2,   /* same node record, mesh `index`              */
    YCGE_ACCEL_MESH_LEAF_INDEX = 3   /* i32 leafTriIndex                        */
} ycge_accel;

typedef struct ycge_ctx ycge_ctx;

/* Fill *cfg with the reference defaults (RaytraceRenderer.cs:31-43,65,218-224). */
int ycge_config_default(ycge_config *cfg);

/* new RaytraceRenderer(fb, scene, fov, pxW, pxH, ss)   RaytraceEntity.cs:97,240,262 */
int ycge_create(const ycge_config *cfg, ycge_ctx **out);
/* dispose */
void ycge_destroy(ycge_ctx *ctx);
/* message of the last failing call on ctx (or of ycge_create when ctx == NULL) */
const char *ycge_last_error(const ycge_ctx *ctx);

/* scene.RebuildBVH()   RaytraceRenderer.cs:107, RaytraceEntity.cs:244, Scene.cs:122-127.
 * Builds the scene BVH (Objects/BVH.cs:258-459) and every mesh BVH
 * (Objects/MeshBVH.cs:371-576) bit-faithfully and uploads them. */
int ycge_scene_upload(ycge_ctx *ctx, const ycge_scene *scene);
/* per-frame entity animation of lights / sky (Scenes/DayNightCycle.cs:80-89) */
int ycge_scene_update_lights(ycge_ctx *ctx, const ycge_light *lights, int32_t n_lights,
                             const ycge_vec3 *ambient_color, float ambient_intensity,
                             const ycge_vec3 *background_top, const ycge_vec3 *background_bottom);

/* Scene.Update() -> RebuildBVH() after an entity moved its geometry (Scenes/Scene.cs:122-127; e.g.
 * BobbingSphereEntity.Update, Scenes/TestScenesRandom.cs:708-714): replaces the Scene.Objects records and
 * rebuilds the scene-level BVH only.  `prims` index the materials, meshes and grids of the last
 * ycge_scene_upload (a Mesh keeps its own BVH in the reference too, Mesh.cs:14).  The tree (Objects/BVH.cs:258-459,
 * same nodes, numbering and leaf order) is built on the device for up to 2 560 objects - the host only flattens the
 * records and their boxes - and by the host builder above that. */
int ycge_scene_update_objects(ycge_ctx *ctx, const ycge_prim *prims, int32_t n_prims);

/* The argument checks of ycge_scene_upload on their own: pure host code, no device and no context needed
 * (every index in range, counts non-negative, pointers present, material kinds known).  Returns YCGE_OK or the
 * status ycge_scene_upload would return; `msg` (may be NULL) receives the reason. */
int ycge_validate_scene(const ycge_scene *scene, char *msg, size_t msg_bytes);

/* Resize(fb, ss)   RaytraceEntity.cs:289; drops TAA history (RaytraceRenderer.cs:137) */
int ycge_resize(ycge_ctx *ctx, int32_t fb_width, int32_t fb_height, int32_t super_sample);
/* SetCamera(pos,yaw,pitch) + SetFov(deg)   RaytraceEntity.cs:229,99 */
int ycge_set_camera(ycge_ctx *ctx, const float pos[3], float yaw, float pitch, float fov_deg);

/* TryFlipAndBlit(fb)   RaytraceEntity.cs:230 / RaytraceRenderer.cs:157-267.
 * out_top_bottom_sdr: fbW*fbH*2*3 f32, caller-owned host memory, per chexel
 * {top rgb, bottom rgb}; the host then does fb.SetChexel(cx,cy,new Chexel('▀',
 * top, bottom)) unchanged (RaytraceRenderer.cs:260-261).  May be NULL (frame is
 * still rendered; read buffers with ycge_read_buffer).  stats may be NULL. */
int ycge_render_frame(ycge_ctx *ctx, float *out_top_bottom_sdr, ycge_frame_stats *stats);
/* Frames in flight (no counterpart in the reference, whose TryFlipAndBlit returns a finished frame): queues steps 1-5 and 9 of the
 * next frame - camera snapshot, trace, TAA, camera commit - and returns without waiting.  The trace of frame N + 1 runs beside the
 * TAA of frame N (its own stream; three sets of trace outputs taken in turn) and two traces run at a time on two streams - the
 * single-launch kernel always, the stage pipeline of voxel worlds from 4096 tiles on and with no post stage in flight (a second set
 * of stage queues) - so a sequence of such calls costs max(trace, TAA + schedule) per frame or less, instead of their sum plus the
 * host's wake-up.  The frames are the ones the same sequence of ycge_render_frame(ctx, NULL, NULL)
 * calls produces, bit for bit.  Single device, no debug captures, no per-frame counters.  Every other entry point (and
 * ycge_wait) first waits for the frames in flight; ycge_set_camera between two calls moves the camera of the next frame. */
int ycge_render_frame_async(ycge_ctx *ctx);
/* ... with steps 6-8 (denoise, exposure, tonemap + downsample) and the read-back: out_top_bottom_sdr (as for ycge_render_frame) is
 * filled when the frame is complete - after ycge_wait or any other call - so a caller that queues several such frames passes one
 * buffer per frame in flight (page-locked, ycge_pin_host_buffer, or the copy blocks the calling thread).  The post stage of frame N
 * runs beside the traces and TAA of the frames after it; same pixels as ycge_render_frame(ctx, out, NULL) in the same order. */
int ycge_render_frame_async_sdr(ycge_ctx *ctx, float *out_top_bottom_sdr);
int ycge_wait(ycge_ctx *ctx);
/* measurement: durations (ms) of the trace launches of the frames queued since the last call, oldest first (at most the last 1024);
 * waits for the frames in flight */
int ycge_async_trace_times(ycge_ctx *ctx, float *ms_out, int32_t capacity, int32_t *n_out);

/* What the frames-in-flight machinery of this context does (it decides timing only, never a pixel - a host or a benchmark reports it
 * beside its numbers).  The placed-value gate - "a trace starts when the trace before it has placed its last workgroup" - rests on an
 * OBSERVED property of the dispatcher (workgroups are placed in index order, so the last index is the last placed); where signal memory
 * or hipStreamWaitValue32 is not available the gate switches itself off and says so here. */
typedef struct ycge_flight_info {
    int32_t two_trace_streams;   /* consecutive frames in flight alternate between two trace streams                               */
    int32_t placed_gate;         /* 1: the placed-value gate is armed; 0: off (YCGE_FLIGHT_PLACED_GATE=0, no signal memory, ...) */
    int32_t post_gate;           /* a trace waits until the post stage before it has placed its persistent launch                */
    int32_t post_pair;           /* the post stages of consecutive frames run side by side                                      */
    int32_t frames_outstanding;  /* 1: frames in flight have not been joined yet                                                 */
    int32_t stage_pipeline;      /* 1: this scene's frames are traced by the stage kernels (k_wf_*), 0: by the single launch      */
    uint64_t placed_waits;       /* traces that were queued behind a placed value since the context was created                 */
} ycge_flight_info;
int ycge_flight_query(ycge_ctx *ctx, ycge_flight_info *out);
/* which exchange the one-process multi-device frame uses: *mode_out = YCGE_EXCHANGE_PEER_PUSH or YCGE_EXCHANGE_RCCL (the latter only when
 * config.multi_device_exchange asked for it AND librccl.so was found and its communicators came up); *world_out = devices that trace */
int ycge_exchange_query(ycge_ctx *ctx, int32_t *mode_out, int32_t *world_out);

/* --- multi-GPU halves of a frame (one process per GPU; the exchange between
 * them is one all-gather of the tile slabs, done by the caller with RCCL).
 * slab layout: for each owned tile in ascending tile_id, for each of its
 * 32x8 pixels in row-major order: 11 f32 {hdr rgb, albedo rgb, normal xyz,
 * depth, sky(0/1)}; ycge_tile_slab_bytes = padded per-rank size (equal on
 * all ranks so that a plain all-gather applies). */
int ycge_tile_slab_bytes(const ycge_ctx *ctx, size_t *bytes);
/* steps 1-4 of TryFlipAndBlit on this rank's tiles; d_slab = DEVICE pointer */
int ycge_trace_tiles(ycge_ctx *ctx, void *d_slab, void *hip_stream, ycge_frame_stats *stats);
/* un-permute world_size gathered slabs (DEVICE pointer, rank-major) into the
 * full-frame buffers, then steps 5-9 (TAA ... tonemap) on the full frame */
int ycge_resolve_gathered(ycge_ctx *ctx, const void *d_all_slabs, void *hip_stream,
                          float *out_top_bottom_sdr, ycge_frame_stats *stats);

/* --- the tile-RESIDENT form of the same partition (one process per GPU): TAA runs on every rank's OWN tiles and its history never
 * leaves the rank.  TemporalBlendWithClamp reads a 3x3 window (clampRadius = 1, RaytraceRenderer.cs:218), so a rank needs {hdr, sky}
 * of the one-pixel ring around each of its tiles from the ranks that own those pixels: one small all-to-all of halo records (4 floats
 * each: 1.3 KB per tile instead of the 8-11 KB of its slab), then TAA, then - for whoever shows the frame - a gather of the RESOLVED
 * history, 12 bytes per pixel.  Per frame and rank at 1920x1080 on 8 ranks: ~1.4 MB of halo records each way + 3.1 MB of history out,
 * against 8.3 MB out / 66 MB in for the all-gather of lean slabs.  The frame ends with TAA (no G-buffer on the consumer: the post stage
 * needs ycge_resolve_gathered).  K = config.tile_ring frame sets: K traces may be in flight, a trace waits for the resolve of frame
 * N - K.  Same pixels as the single-device frame, bit for bit (the halo records are copies, the per-pixel arithmetic is k_taa's).
 *
 *   every frame, every rank:   ycge_trace_tiles_resident(ctx, d_send, stream)        trace + gather of the records other ranks need
 *                              all_to_all(d_recv <- d_send) with ycge_halo_counts' split sizes (records of 4 floats)
 *                              ycge_resolve_tiles_resident(ctx, d_recv, d_hist_slab, stream)   ONE launch: TAA on own tiles with the halo taps read from
 *                                                                                 d_recv where the exchange left them, history slab
 *   the consumer:              all_gather / gather of the history slabs -> ycge_unpack_history(ctx, d_all_hist_slabs, stream) */
int ycge_halo_counts(ycge_ctx *ctx, int64_t *send_counts /* [world_size] */, int64_t *recv_counts /* [world_size] */);
int ycge_history_slab_bytes(const ycge_ctx *ctx, size_t *bytes);      /* padded per-rank size: equal on all ranks */
int ycge_trace_tiles_resident(ycge_ctx *ctx, void *d_halo_send, void *hip_stream, ycge_frame_stats *stats);
/* n (1..8, <= the free sets of the ring) CONSECUTIVE frames of this rank's tiles in ONE launch - for a caller that knows the next n camera
 * poses (a recorder, a benchmark, a render thread that runs ahead): poses = n x {pos x y z, yaw, pitch, fov_deg}, as n ycge_set_camera
 * calls would give them (the last stays the context's camera); d_halo_send = n buffers, filled as by n ycge_trace_tiles_resident calls.
 * A rank's share of ONE frame is a few thousand blocks whose longest chains leave most of the machine idle; n frames' blocks in one
 * launch are the work of a rank of world_size / n ranks - throughput approaches N-fold on N GPUs as n approaches N, at n frames of
 * latency.  The frames are then exchanged and resolved one by one, oldest first, as after n single calls; same pixels. */
int ycge_trace_tiles_resident_batch(ycge_ctx *ctx, int32_t n, const float *poses /* n x 6 */, void *const *d_halo_send /* n */, void *hip_stream);
int ycge_resolve_tiles_resident(ycge_ctx *ctx, const void *d_halo_recv, void *d_history_slab /* may be NULL */, void *hip_stream, ycge_frame_stats *stats);
int ycge_unpack_history(ycge_ctx *ctx, const void *d_all_history_slabs, void *hip_stream);

/* A live texture's next frame (what IFrameReader.GetCurrentFramePtr() will return while the coming frames are traced): bytes =
 * width * height * frame_bytes_per_pixel of texture `texture_index` of the last ycge_scene_upload.  The host sets
 * ycge_scene.has_dynamic_textures for such scenes (Scene.cs:30), which restarts the TAA history every frame. */
int ycge_scene_update_texture(ycge_ctx *ctx, int32_t texture_index, const uint8_t *frame, size_t bytes);

/* tests only */
int ycge_read_buffer(ycge_ctx *ctx, int32_t which /* ycge_buffer */, void *dst, size_t bytes);
int ycge_set_frame_counter(ycge_ctx *ctx, int64_t frame_counter);
/* measurement: what the TIMED kernel instances (config.count_work == 0) have done since the context was created, all devices:
 * traversal-loop steps summed over lanes - one step = one node visit (a 64-byte record), one leaf record (72 bytes: two triangles)
 * or one voxel cell.  The counting instances walk the reference's full traversal (SURVEY 8d counters in ycge_frame_stats); the
 * timed ones stop shadow queries at the first hit and skip culled grids, so their own work is reported apart.  Waits for the
 * context's stream; never called inside a timed region. */
int ycge_read_timed_steps(ycge_ctx *ctx, uint64_t *lane_steps);
/* The SDR frame leaves the device by one copy into the caller's buffer at the end of ycge_render_frame / ycge_resolve_gathered (24 bytes per
 * chexel: 25 MB at 1920x540).  Into page-locked memory that copy is a DMA at the link's full rate instead of a staged copy into pageable
 * pages, and a frame in flight (ycge_render_frame_async_sdr) never blocks the calling thread.  Two ways to get such memory:
 *   ycge_alloc_host_buffer / ycge_free_host_buffer   the library's own (hipHostMalloc, zeroed): what a host should use for its SDR frames
 *                                                     (the C# wrapper reads it through a Span<float>);
 *   ycge_pin_host_buffer / ycge_unpin_host_buffer     registers memory the caller owns (hipHostRegister).  Registration is page-granular:
 *                                                     two registered ranges that share a page lose it when one is unregistered, and the
 *                                                     other's next read-back faults on the device.  So the range must be WHOLE PAGES OF ITS
 *                                                     OWN: `buffer` aligned to ycge_host_page_size() and `bytes` a multiple of it, else
 *                                                     YCGE_ERR_INVALID_ARG (a pinned managed array - GCHandle of a float[] - never qualifies).
 * ABI 8.  Optional either way; unpin / free before the memory goes away, and only after ycge_wait. */
size_t ycge_host_page_size(void);
int ycge_alloc_host_buffer(size_t bytes, void **out_buffer);
int ycge_free_host_buffer(void *buffer);
int ycge_pin_host_buffer(void *buffer, size_t bytes);
int ycge_unpin_host_buffer(void *buffer);
/* visible HIP devices (hipGetDeviceCount; does not initialise a device context), < 0 on error: what a host checks before it
 * fills config.devices[] */
int ycge_device_count(void);
int ycge_accel_size(ycge_ctx *ctx, int32_t which /* ycge_accel */, int32_t index, size_t *bytes);
int ycge_read_accel(ycge_ctx *ctx, int32_t which, int32_t index, void *dst, size_t bytes);
/* name of the device the context runs on + whether the gfx950 code object loaded */
int ycge_device_info(ycge_ctx *ctx, char *name, size_t name_bytes, int32_t *compute_units);

#ifdef __cplusplus
}
#endif
#endif /* YCGE_H */
