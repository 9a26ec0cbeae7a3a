"""ctypes binding of the ORACLE (oracle/liborc_oracle.so) — test infrastructure only.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from yetanotherconsolegameengine_amd import abi  # noqa: E402
from yetanotherconsolegameengine_amd.scene import Scene, flatten  # noqa: E402

ORACLE_DIR = ROOT / "oracle"
ORACLE_LIB = ORACLE_DIR / "liborc_oracle.so"

_lib = None


def build_oracle(force: bool = False) -> Path:
    srcs = [ORACLE_DIR / n for n in ("orc_scene.cpp", "orc_render.cpp", "orc_scene.h", "orc_math.h")] + [ROOT / "include" / "ycge.h"]
    stale = (not ORACLE_LIB.exists()) or any(s.stat().st_mtime > ORACLE_LIB.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", "liborc_oracle.so"], check=True, capture_output=True)
    return ORACLE_LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(str(ORACLE_LIB))
        names = ["ycge_create", "ycge_destroy", "ycge_last_error", "ycge_scene_upload", "ycge_scene_update_lights",
                 "ycge_resize", "ycge_set_camera", "ycge_read_buffer", "ycge_set_frame_counter", "ycge_accel_size",
                 "ycge_read_accel"]
        abi.bind(L, prefix="orc_", names=names)
        L.orc_render_frame.restype = C.c_int
        L.orc_render_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(abi.FrameStats), C.c_int, C.c_int]
        L.orc_scene_update_texture.restype = C.c_int
        L.orc_scene_update_texture.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]
        L.orc_set_taa_threads.restype = C.c_int
        L.orc_set_taa_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_build_stats.restype = C.c_int
        L.orc_build_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.orc_splitmix64.restype = C.c_uint64; L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_per_frame_seed.restype = C.c_uint64
        L.orc_per_frame_seed.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_uint64]
        L.orc_rng_init.restype = C.c_uint64; L.orc_rng_init.argtypes = [C.c_uint64]
        L.orc_rng_next_unit.restype = C.c_float; L.orc_rng_next_unit.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_blue_noise_sample.restype = C.c_float; L.orc_blue_noise_sample.argtypes = [C.c_int] * 4
        L.orc_frac.restype = C.c_float; L.orc_frac.argtypes = [C.c_float]
        L.orc_sincos.restype = None; L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        for n in ("orc_pow5", "orc_exp", "orc_log"):
            getattr(L, n).restype = C.c_float; getattr(L, n).argtypes = [C.c_float]
        for n in ("orc_pow", "orc_max", "orc_min"):
            getattr(L, n).restype = C.c_float; getattr(L, n).argtypes = [C.c_float, C.c_float]
        L.orc_f2i.restype = C.c_int32; L.orc_f2i.argtypes = [C.c_float]
        L.orc_cosine_sample_hemisphere.restype = None
        L.orc_cosine_sample_hemisphere.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        L.orc_oren_nayar.restype = None
        L.orc_oren_nayar.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_float, C.POINTER(C.c_float)]
        L.orc_introsort.restype = None
        L.orc_introsort.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int]
        L.orc_scene_hit.restype = C.c_int
        L.orc_scene_hit.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_scene_hit_bruteforce.restype = C.c_int
        L.orc_scene_hit_bruteforce.argtypes = L.orc_scene_hit.argtypes
        L.orc_morton3.restype = C.c_int; L.orc_morton3.argtypes = [C.c_int] * 3
        _lib = L
    return _lib


def _check(L, ctx, rc):
    if rc != 0:
        msg = L.orc_last_error(ctx) if ctx else b""
        raise abi.YcgeError(rc, (msg or b"").decode())


NODE_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("start", "<i4"), ("count", "<i4")])


class OracleRenderer:
    """Drives the oracle through the same call sequence the product's RaytraceRenderer uses."""

    def __init__(self, scene: Scene, fb_w: int, fb_h: int, ss: int = 1, pose=None, cfg: abi.Config | None = None, flat=None):
        self.L = lib()
        c = cfg if cfg is not None else abi.default_config()
        c.fb_width, c.fb_height, c.super_sample = fb_w, fb_h, ss
        if pose is not None:
            c.fov_deg = pose.get("fov", 45.0)
        self.cfg = c
        self.ctx = C.c_void_p()
        _check(self.L, None, self.L.orc_create(C.byref(c), C.byref(self.ctx)))
        self.flat = flat if flat is not None else flatten(scene)
        _check(self.L, self.ctx, self.L.orc_scene_upload(self.ctx, self.flat.byref()))
        self.hiW, self.hiH = fb_w * ss, fb_h * 2 * ss
        self.fbW, self.fbH = fb_w, fb_h
        if pose is not None:
            self.set_camera(pose["pos"], pose["yaw"], pose["pitch"], pose.get("fov", 45.0))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self.ctx:
            self.L.orc_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def set_camera(self, pos, yaw, pitch, fov=45.0):
        p = (C.c_float * 3)(*pos)
        _check(self.L, self.ctx, self.L.orc_set_camera(self.ctx, p, yaw, pitch, fov))

    def update_texture(self, texture):
        """the oracle's twin of RaytraceRenderer.UpdateTexture: the live texture's next frame"""
        idx = next(i for i, t in enumerate(self.flat.texture_objects) if t is texture)
        f = texture.frame
        _check(self.L, self.ctx, self.L.orc_scene_update_texture(self.ctx, idx, f.ctypes.data_as(C.c_void_p), f.nbytes))

    def resize(self, fb_w: int, fb_h: int, ss: int):
        """Resize(fb, ss) (RaytraceEntity.cs:289): new buffers, TAA history dropped (RaytraceRenderer.cs:137)"""
        _check(self.L, self.ctx, self.L.orc_resize(self.ctx, fb_w, fb_h, ss))
        self.fbW, self.fbH = fb_w, fb_h
        self.hiW, self.hiH = fb_w * ss, fb_h * 2 * ss

    def set_frame_counter(self, n: int):
        _check(self.L, self.ctx, self.L.orc_set_frame_counter(self.ctx, n))

    def set_taa_threads(self, n: int):
        """1 = the reference's serial TAA loops; n > 1 = row bands (identical result; the 'also reported parallel' baseline)."""
        _check(self.L, self.ctx, self.L.orc_set_taa_threads(self.ctx, n))

    def render(self, stages: int = 2, threads: int = 1, want_sdr: bool = False):
        st = abi.FrameStats()
        sdr = np.zeros((self.fbH, self.fbW, 2, 3), dtype=np.float32) if want_sdr else None
        ptr = sdr.ctypes.data_as(C.POINTER(C.c_float)) if want_sdr else None
        _check(self.L, self.ctx, self.L.orc_render_frame(self.ctx, ptr, C.byref(st), threads, stages))
        self.stats = st
        return sdr if want_sdr else st

    def read(self, which: int) -> np.ndarray:
        dt, n = abi.BUFFER_LAYOUT[which]
        shape = (self.hiH, self.hiW, n) if n > 1 else (self.hiH, self.hiW)
        a = np.zeros(shape, dtype=dt)
        _check(self.L, self.ctx, self.L.orc_read_buffer(self.ctx, which, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def accel(self, which: int, index: int = 0) -> np.ndarray:
        n = C.c_size_t()
        _check(self.L, self.ctx, self.L.orc_accel_size(self.ctx, which, index, C.byref(n)))
        dt = NODE_DTYPE if which in (abi.ACCEL_SCENE_NODES, abi.ACCEL_MESH_NODES) else np.dtype("<i4")
        a = np.zeros(n.value // dt.itemsize, dtype=dt)
        if n.value:
            _check(self.L, self.ctx, self.L.orc_read_accel(self.ctx, which, index, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def build_stats(self, mesh_index: int = 0):
        out = (C.c_int32 * 4)()
        _check(self.L, self.ctx, self.L.orc_build_stats(self.ctx, mesh_index, out))
        return dict(scene_sort_fallbacks=out[0], scene_max_depth=out[1], mesh_sort_fallbacks=out[2], mesh_max_depth=out[3])

    def scene_hit(self, o, d, t_min=0.001, t_max=3.4028234663852886e38, brute=False):
        out = (C.c_float * 13)()
        fn = self.L.orc_scene_hit_bruteforce if brute else self.L.orc_scene_hit
        _check(self.L, self.ctx, fn(self.ctx, (C.c_float * 3)(*o), (C.c_float * 3)(*d), t_min, t_max, out))
        return np.array(out[:], dtype=np.float32)
