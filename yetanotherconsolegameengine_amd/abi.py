"""ctypes mirror of include/ycge.h — the C-ABI of the MI355X ray-trace core.

Every structure here has the same field order and types as its C twin; the
`-m "not gpu"` tests check sizes/offsets against the header through the built
library (ycge_abi_sizeof).  The product library is libycge_hip.so (host C++ +
gfx950 HIP kernels); there is NO CPU fallback: if the library or its device
code is missing, loading or ycge_create fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

YCGE_ABI_VERSION = 9
YCGE_MAX_DEVICES = 8

# ycge_status
YCGE_OK = 0
YCGE_ERR_INVALID_ARG = -1
YCGE_ERR_NO_SCENE = -2
YCGE_ERR_DEVICE = -3
YCGE_ERR_UNSUPPORTED = -4
YCGE_ERR_OUT_OF_MEMORY = -5
YCGE_ERR_STACK_DEPTH = -6
YCGE_ERR_NO_DEVICE_CODE = -7
YCGE_ERR_INTERNAL = -8

STATUS_NAMES = {
    0: "YCGE_OK", -1: "YCGE_ERR_INVALID_ARG", -2: "YCGE_ERR_NO_SCENE", -3: "YCGE_ERR_DEVICE",
    -4: "YCGE_ERR_UNSUPPORTED", -5: "YCGE_ERR_OUT_OF_MEMORY", -6: "YCGE_ERR_STACK_DEPTH",
    -7: "YCGE_ERR_NO_DEVICE_CODE",
    -8: "YCGE_ERR_INTERNAL",
}

# ycge_exchange
EXCHANGE_PEER_PUSH, EXCHANGE_RCCL = 0, 1
# ycge_material_kind
MAT_CONSTANT, MAT_CHECKER, MAT_TEXTURED = 0, 1, 2
# ycge_prim_type
(PRIM_SPHERE, PRIM_PLANE, PRIM_DISK, PRIM_XYRECT, PRIM_XZRECT, PRIM_YZRECT, PRIM_BOX,
 PRIM_CYLINDER_Y, PRIM_TRIANGLE, PRIM_MESH, PRIM_VOLUME_GRID) = range(11)
# ycge_buffer
(BUF_RAYS, BUF_PRIM_ID, BUF_SUB_ID, BUF_HIT_T, BUF_CURRENT_HDR, BUF_G_ALBEDO, BUF_G_NORMAL,
 BUF_G_DEPTH, BUF_SKY_MASK, BUF_TAA_HISTORY, BUF_PREV_NORMAL, BUF_PREV_DEPTH, BUF_PREV_SKY,
 BUF_DENOISED, BUF_RNG_STATE) = range(15)
# ycge_accel
ACCEL_SCENE_NODES, ACCEL_SCENE_LEAF_INDEX, ACCEL_MESH_NODES, ACCEL_MESH_LEAF_INDEX = range(4)

# (numpy dtype, elements per pixel) of each per-pixel buffer
BUFFER_LAYOUT = {
    BUF_RAYS: ("<f4", 6), BUF_PRIM_ID: ("<i4", 1), BUF_SUB_ID: ("<i4", 1), BUF_HIT_T: ("<f4", 1),
    BUF_CURRENT_HDR: ("<f4", 3), BUF_G_ALBEDO: ("<f4", 3), BUF_G_NORMAL: ("<f4", 3), BUF_G_DEPTH: ("<f4", 1),
    BUF_SKY_MASK: ("u1", 1), BUF_TAA_HISTORY: ("<f4", 3), BUF_PREV_NORMAL: ("<f4", 3),
    BUF_PREV_DEPTH: ("<f4", 1), BUF_PREV_SKY: ("u1", 1), BUF_DENOISED: ("<f4", 3), BUF_RNG_STATE: ("<u8", 1),
}


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("albedo", Vec3), ("albedo_b", Vec3), ("checker_scale", C.c_float),
        ("specular", C.c_float), ("reflectivity", C.c_float), ("emission", Vec3),
        ("transparency", C.c_float), ("index_of_refraction", C.c_float), ("transmission_color", Vec3),
        ("texture", C.c_int32), ("reserved", C.c_int32), ("texture_weight", C.c_double), ("uv_scale", C.c_double),
    ]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixels", C.POINTER(C.c_uint32)),
                ("frame_bytes_per_pixel", C.c_int32), ("flip_u", C.c_int32), ("flip_v", C.c_int32), ("frame", C.POINTER(C.c_uint8))]


class Prim(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("material", C.c_int32), ("ref", C.c_int32), ("reserved", C.c_int32),
        ("p", C.c_float * 12), ("specular", C.c_float), ("reflectivity", C.c_float),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("triangles", C.POINTER(C.c_float)), ("n_triangles", C.c_int32), ("material", C.c_int32),
        ("tri_material", C.POINTER(C.c_int32)),
    ]


class VoxelLookup(C.Structure):
    _fields_ = [("mat_id", C.c_int32), ("meta_id", C.c_int32), ("material", C.c_int32)]


class Grid(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("min_corner", Vec3), ("voxel_size", Vec3),
        ("cells", C.POINTER(C.c_int32)), ("lookup", C.POINTER(VoxelLookup)), ("n_lookup", C.c_int32),
        ("default_material", C.c_int32), ("wireframe", C.c_int32), ("wire_width_fraction", C.c_float),
        ("wire_max_distance", C.c_float),
    ]


class Light(C.Structure):
    _fields_ = [("position", Vec3), ("color", Vec3), ("intensity", C.c_float)]


class Scene(C.Structure):
    _fields_ = [
        ("materials", C.POINTER(Material)), ("n_materials", C.c_int32),
        ("prims", C.POINTER(Prim)), ("n_prims", C.c_int32),
        ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_int32),
        ("grids", C.POINTER(Grid)), ("n_grids", C.c_int32),
        ("lights", C.POINTER(Light)), ("n_lights", C.c_int32),
        ("ambient_color", Vec3), ("ambient_intensity", C.c_float),
        ("background_top", Vec3), ("background_bottom", Vec3),
        ("is_volume_scene", C.c_int32), ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
        ("has_dynamic_textures", C.c_int32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("fb_width", C.c_int32), ("fb_height", C.c_int32), ("super_sample", C.c_int32),
        ("fov_deg", C.c_float), ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("diffuse_bounces", C.c_int32), ("max_mirror_bounces", C.c_int32), ("max_refractions", C.c_int32),
        ("mirror_threshold", C.c_float), ("eps", C.c_float), ("seed_salt", C.c_uint64),
        ("taa_alpha", C.c_float), ("motion_trans_reset", C.c_float), ("motion_rot_reset", C.c_float),
        ("diffuse_sigma_deg", C.c_float), ("taa_clamp_radius", C.c_int32), ("taa_luminance_pad", C.c_float),
        ("atrous_iterations", C.c_int32), ("atrous_c_phi", C.c_float), ("atrous_n_phi", C.c_float),
        ("atrous_z_phi", C.c_float), ("atrous_a_phi", C.c_float), ("capture_debug", C.c_int32),
        ("count_work", C.c_int32),
        ("slab_albedo", C.c_int32),
        ("n_devices", C.c_int32), ("devices", C.c_int32 * YCGE_MAX_DEVICES),
        ("atrous_inplace_exact", C.c_int32), ("tile_ring", C.c_int32), ("multi_device_exchange", C.c_int32),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("frame", C.c_int64), ("history_reset", C.c_int32), ("fan_blocks", C.c_int32),
        ("trace_ms", C.c_double), ("taa_ms", C.c_double), ("post_ms", C.c_double), ("total_ms", C.c_double),
        ("n_rays", C.c_uint64), ("n_box", C.c_uint64), ("n_tri", C.c_uint64), ("n_prim", C.c_uint64),
        ("n_vox", C.c_uint64), ("exposure", C.c_float), ("exposure_serial_chunks", C.c_float),
        ("n_rays_dark", C.c_uint64), ("n_devices_traced", C.c_int32), ("device_tiles", C.c_int32 * 8),
    ]


class FlightInfo(C.Structure):
    _fields_ = [("two_trace_streams", C.c_int32), ("placed_gate", C.c_int32), ("post_gate", C.c_int32), ("post_pair", C.c_int32),
                ("frames_outstanding", C.c_int32), ("stage_pipeline", C.c_int32), ("placed_waits", C.c_uint64)]


def default_config() -> Config:
    """Reference defaults (RaytraceRenderer.cs:31-43,65,218-224) — pure data, no library needed."""
    c = Config()
    c.abi_version = YCGE_ABI_VERSION
    c.fb_width, c.fb_height, c.super_sample = 80, 45, 1
    c.fov_deg = 45.0
    c.device, c.rank, c.world_size = 0, 0, 1
    c.diffuse_bounces, c.max_mirror_bounces, c.max_refractions = 1, 2, 2
    c.mirror_threshold, c.eps = 0.9, 1e-4
    c.seed_salt = 0x9E3779B97F4A7C15
    c.taa_alpha, c.motion_trans_reset, c.motion_rot_reset = 0.01, 0.0025, 0.0025
    c.diffuse_sigma_deg = 25.0
    c.taa_clamp_radius, c.taa_luminance_pad = 1, 0.10
    c.atrous_iterations = 3
    c.atrous_c_phi, c.atrous_n_phi, c.atrous_z_phi, c.atrous_a_phi = 3.0, 0.35, 2.0, 0.20
    c.capture_debug, c.count_work = 0, 0
    c.slab_albedo, c.n_devices = 1, 0
    c.atrous_inplace_exact = 1
    c.tile_ring = 0
    c.multi_device_exchange = EXCHANGE_PEER_PUSH
    return c


_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "lib" / "libycge_hip.so"

# name -> (restype, argtypes); every symbol include/ycge.h declares
_PROTOTYPES = {
    "ycge_config_default": (C.c_int, [C.POINTER(Config)]),
    "ycge_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "ycge_destroy": (None, [C.c_void_p]),
    "ycge_last_error": (C.c_char_p, [C.c_void_p]),
    "ycge_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(Scene)]),
    "ycge_validate_scene": (C.c_int, [C.POINTER(Scene), C.c_char_p, C.c_size_t]),
    "ycge_scene_update_lights": (C.c_int, [C.c_void_p, C.POINTER(Light), C.c_int32, C.POINTER(Vec3), C.c_float,
                                           C.POINTER(Vec3), C.POINTER(Vec3)]),
    "ycge_scene_update_objects": (C.c_int, [C.c_void_p, C.POINTER(Prim), C.c_int32]),
    "ycge_scene_update_texture": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "ycge_resize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "ycge_set_camera": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float]),
    "ycge_render_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(FrameStats)]),
    "ycge_render_frame_async": (C.c_int, [C.c_void_p]),
    "ycge_render_frame_async_sdr": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ycge_wait": (C.c_int, [C.c_void_p]),
    "ycge_async_trace_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "ycge_flight_query": (C.c_int, [C.c_void_p, C.POINTER(FlightInfo)]),
    "ycge_exchange_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ycge_tile_slab_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "ycge_trace_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]),
    "ycge_resolve_gathered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(FrameStats)]),
    "ycge_halo_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ycge_history_slab_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "ycge_trace_tiles_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]),
    "ycge_trace_tiles_resident_batch": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.c_void_p]),
    "ycge_resolve_tiles_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]),
    "ycge_unpack_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ycge_read_buffer": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "ycge_set_frame_counter": (C.c_int, [C.c_void_p, C.c_int64]),
    "ycge_read_timed_steps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ycge_device_count": (C.c_int, []),
    "ycge_host_page_size": (C.c_size_t, []),
    "ycge_alloc_host_buffer": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "ycge_free_host_buffer": (C.c_int, [C.c_void_p]),
    "ycge_pin_host_buffer": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ycge_unpin_host_buffer": (C.c_int, [C.c_void_p]),
    "ycge_accel_size": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "ycge_read_accel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "ycge_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


class YcgeError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def bind(lib: C.CDLL, prefix: str = "ycge_", names=None) -> C.CDLL:
    """Attach prototypes to `lib`; `prefix` lets the tests bind the oracle twin (orc_*)."""
    for name, (res, args) in _PROTOTYPES.items():
        if names is not None and name not in names:
            continue
        fn = getattr(lib, prefix + name[len("ycge_"):])
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load libycge_hip.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path is not None else Path(os.environ.get("YCGE_LIB", LIB_PATH))
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the ray-trace path.")
    lib = bind(C.CDLL(str(p)))
    if path is None:
        _lib = lib
    return lib
