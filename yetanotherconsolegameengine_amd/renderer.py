"""RaytraceRenderer — host-side mirror of the reference's render entry point.

Same surface as ConsoleGame/RayTracing/RaytraceRenderer.cs (ctor :74, Resize :110,
SetCamera :140, SetFov :150, TryFlipAndBlit :157), i.e. the IConsoleRenderer seam of
RaytraceEntity.cs:12-18, implemented by calls through the C-ABI (include/ycge.h) into
the gfx950 kernels.  No per-pixel work happens in Python and there is no CPU fallback:
constructing a renderer without the built library or without an MI355X raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import abi
from .scene import FlatScene, Scene, flatten

NODE_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("start", "<i4"), ("count", "<i4")])


class _PageLockedOwner:
    """Owns one ycge_alloc_host_buffer allocation; frees it when collected (after the last numpy view over it)."""

    def __init__(self, lib, address: int):
        self._lib, self.address = lib, address

    def __del__(self):
        try:
            if self.address:
                self._lib.ycge_free_host_buffer(C.c_void_p(self.address))
                self.address = 0
        except Exception:
            pass


class RaytraceRenderer:
    def __init__(self, scene: Scene | FlatScene, fb_width: int, fb_height: int, fovDeg: float = 45.0, superSample: int = 1, *,
                 cfg: Optional[abi.Config] = None, capture_debug: bool = False, count_work: bool = False, device: int = 0,
                 rank: int = 0, world_size: int = 1, slab_albedo: bool = True, devices=None, lib=None, tile_ring: int = 0):
        self.L = lib if lib is not None else abi.load_library()
        c = cfg if cfg is not None else abi.default_config()
        c.fb_width, c.fb_height, c.super_sample = fb_width, fb_height, max(1, superSample)
        c.fov_deg = fovDeg
        c.capture_debug, c.count_work = int(capture_debug), int(count_work)
        c.device, c.rank, c.world_size = device, rank, world_size
        c.tile_ring = int(tile_ring)              # tile-resident form: frame sets in the ring (0 = 2)
        c.slab_albedo = int(slab_albedo)          # tiled frame: lean 8-float slabs when the denoise stage will not run
        if devices is not None:                   # one process, several GPUs: TryFlipAndBlit drives them all (config.n_devices)
            c.n_devices = len(devices)
            for i, d in enumerate(devices):
                c.devices[i] = int(d)
        self.cfg = c
        self.ctx = C.c_void_p()
        rc = self.L.ycge_create(C.byref(c), C.byref(self.ctx))
        if rc != 0:
            raise abi.YcgeError(rc, (self.L.ycge_last_error(None) or b"").decode())
        self._set_dims(fb_width, fb_height, c.super_sample)
        self._pos, self._yaw, self._pitch, self._fov = (0.0, 1.0, 0.0), 0.0, 0.0, fovDeg
        self.flat = None
        self.stats = abi.FrameStats()
        if scene is not None:
            self.UploadScene(scene)          # the C# ctor ends with scene.RebuildBVH() (:107)

    # ---------------------------------------------------------------- plumbing
    def _set_dims(self, w, h, ss):
        self.fbW, self.fbH, self.ss = w, h, ss
        self.hiW, self.hiH = w * ss, h * 2 * ss

    def _check(self, rc: int):
        if rc != 0:
            raise abi.YcgeError(rc, (self.L.ycge_last_error(self.ctx) or b"").decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.L.ycge_destroy(self.ctx)          # (waits for the frames in flight: their SDR arrays are still the wrapper's)
            self.ctx = C.c_void_p()
        self._drop_sdr_buffer()
        self._drop_sdr_ring()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- reference surface
    def UploadScene(self, scene: Scene | FlatScene):
        """scene.RebuildBVH() + upload (RaytraceRenderer.cs:107, RaytraceEntity.cs:244)."""
        self.flat = scene if hasattr(scene, "byref") else flatten(scene)          # (a FlatScene, or a scene file read back: tools/scene_file.py)
        self._check(self.L.ycge_scene_upload(self.ctx, self.flat.byref()))

    def UpdateLights(self, lights, ambient=None, background_top=None, background_bottom=None):
        arr = (abi.Light * max(1, len(lights)))()
        for i, l in enumerate(lights):
            arr[i].position, arr[i].color, arr[i].intensity = abi.Vec3(*l.Position), abi.Vec3(*l.Color), float(l.Intensity)
        amb = abi.Vec3(*ambient.Color) if ambient is not None else None
        top = abi.Vec3(*background_top) if background_top is not None else None
        bot = abi.Vec3(*background_bottom) if background_bottom is not None else None
        self._check(self.L.ycge_scene_update_lights(
            self.ctx, arr, len(lights), C.byref(amb) if amb is not None else None,
            float(ambient.Intensity) if ambient is not None else 0.0,
            C.byref(top) if top is not None else None, C.byref(bot) if bot is not None else None))

    def UpdateTexture(self, texture) -> None:
        """The next frame of a live texture (LiveTexture.set_frame before this call): what IFrameReader.GetCurrentFramePtr() returns
        while the coming frames are traced (Renderer/Texture.cs:116)."""
        idx = next(i for i, t in enumerate(self.flat.texture_objects) if t is texture)
        f = texture.frame
        self._check(self.L.ycge_scene_update_texture(self.ctx, idx, f.ctypes.data_as(C.c_void_p), f.nbytes))

    def UpdateObjects(self, scene: Scene | FlatScene):
        """Scene.Update() -> RebuildBVH() after entities moved (Scene.cs:122-127): same materials, meshes and grids
        as the uploaded scene (in the same first-use order), new object records; only the scene BVH is rebuilt."""
        f = scene if isinstance(scene, FlatScene) else flatten(scene, against=self.flat if hasattr(self.flat, "_mat_index") else None)          # (a Scene: numbered against the upload; NeedsUpload if it holds something new)
        self._check(self.L.ycge_scene_update_objects(self.ctx, C.cast(f.prims, C.POINTER(abi.Prim)), f.struct.n_prims))
        self.flat = f

    def scene_bvh_stats(self) -> dict:
        """How ycge_scene_update_objects built the scene BVH so far: on the device (csrc/ycge_bvh_build.hip), on the host after the
        kernel declined, on the host outright; microseconds of the last build + install; how often the current tree took the
        reference's Array.Sort path (BVH.cs:389,419) and its depth."""
        out = (C.c_int64 * 6)()
        self.L.ycge_debug_scene_bvh_stats.restype = C.c_int
        self.L.ycge_debug_scene_bvh_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.ycge_debug_scene_bvh_stats(self.ctx, out))
        return dict(device_builds=int(out[0]), host_fallbacks=int(out[1]), host_builds=int(out[2]), last_build_us=int(out[3]),
                    sort_fallbacks=int(out[4]), max_depth=int(out[5]))

    def Resize(self, fb_width: int, fb_height: int, superSample: int):
        self._check(self.L.ycge_resize(self.ctx, fb_width, fb_height, superSample))          # (joins the frames in flight: nothing writes the old arrays any more)
        self._set_dims(fb_width, fb_height, max(1, superSample))
        self._drop_sdr_ring(keep_shape=(self.fbH, self.fbW, 2, 3))

    def _drop_sdr_ring(self, keep_shape=None):
        """The page-locked SDR arrays of the frames in flight: those of another console size go back to the library (nothing is in flight:
        the callers are Resize, which joins first, and close, after ycge_destroy)."""
        ring = self.__dict__.get("_sdr_ring", {})
        for key in [k for k, (a, _) in ring.items() if a.shape != keep_shape]:
            a, handle = ring.pop(key)
            self._free_page_locked(handle)

    def SetCamera(self, pos, yaw: float, pitch: float):
        self._pos, self._yaw, self._pitch = tuple(pos), yaw, pitch
        self._push_camera()

    def SetFov(self, fovDeg: float):
        self._fov = fovDeg
        self._push_camera()

    def _push_camera(self):
        p = (C.c_float * 3)(*self._pos)
        self._check(self.L.ycge_set_camera(self.ctx, p, self._yaw, self._pitch, self._fov))

    def _page_locked_zeros(self, shape):
        """A zeroed float32 array in page-locked memory OF THE LIBRARY (ycge_alloc_host_buffer: hipHostMalloc): (array, owner).  The
        device writes SDR frames straight into it.  Round 4 registered numpy arrays instead (hipHostRegister) and met GPU memory faults
        at heap addresses: registration is page-granular, and - more to the point - a mapping of process heap lives and dies with the
        allocator, not with the array (csrc/ycge_host.cpp: copy_out).
        LIFETIME: the allocation belongs to the ARRAY, not to the renderer - the ctypes buffer under the numpy array carries a
        _PageLockedOwner whose finaliser hands the pages back (ycge_free_host_buffer) when the last view of the array dies.  close(),
        Resize() and a change of console size only drop the renderer's own reference (after joining the frames in flight, so the device
        is done with the pages): an array a caller still holds - TryFlipAndBlit(copy=False), RenderAsync(sdr_slot=k) - stays readable."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        rc = self.L.ycge_alloc_host_buffer(n * 4, C.byref(p))
        if rc != 0 or not p.value:
            raise abi.YcgeError(rc, f"no page-locked memory for an SDR frame of {n * 4} bytes")
        owner = _PageLockedOwner(self.L, p.value)
        buf = (C.c_float * n).from_address(p.value)
        buf._ycge_owner = owner          # (numpy keeps `buf` alive through the buffer protocol; `buf` keeps the owner)
        a = np.ctypeslib.as_array(buf).reshape(shape)
        return a, owner

    def _free_page_locked(self, handle):
        """Drops the renderer's reference; the pages go back when no array over them is left (see _page_locked_zeros)."""
        return None

    def _sdr_buffer(self):
        """The wrapper's ONE SDR buffer (the C# side keeps one for the life of the renderer, bindings/csharp/HipRaytraceWrapper.cs), page-locked
        memory of the library: the frame's read-back is a plain DMA."""
        shape = (self.fbH, self.fbW, 2, 3)
        if getattr(self, "_sdr", None) is None or self._sdr.shape != shape:
            if getattr(self, "_sdr", None) is not None and getattr(self, "ctx", None):
                self.L.ycge_wait(self.ctx)
            self._drop_sdr_buffer()
            self._sdr, self._sdr_handle = self._page_locked_zeros(shape)
        return self._sdr

    def _drop_sdr_buffer(self):
        if getattr(self, "_sdr", None) is not None:
            self._free_page_locked(getattr(self, "_sdr_handle", None))
        self._sdr = None
        self._sdr_handle = None

    def TryFlipAndBlit(self, want_sdr: bool = False, copy: bool = True):
        """One frame.  Returns the fbH x fbW x 2 x 3 SDR array (top, bottom per chexel) when want_sdr (a copy of the wrapper's
        buffer; copy=False hands out the buffer itself: OVERWRITTEN by the next frame of this size, left alone - and valid for as long as
        the caller holds it - after Resize() to another size or close()), else the frame statistics."""
        sdr = self._sdr_buffer() if want_sdr else None
        ptr = sdr.ctypes.data_as(C.POINTER(C.c_float)) if want_sdr else None
        self._check(self.L.ycge_render_frame(self.ctx, ptr, C.byref(self.stats)))
        return (sdr.copy() if copy else sdr) if want_sdr else self.stats

    def RenderAsync(self, sdr_slot=None):
        """Frames in flight (ycge_render_frame_async): queues the next frame through TAA and returns; the trace of the frame after it
        runs beside this one's TAA.  Same frames as TryFlipAndBlit() in the same order.  Wait() - or any other call - joins.
        With sdr_slot = k the frame runs the post stage too (ycge_render_frame_async_sdr) into the wrapper's k-th page-locked SDR array,
        which is returned and holds the frame once Wait() has returned: one slot per frame in flight."""
        if sdr_slot is None:
            self._check(self.L.ycge_render_frame_async(self.ctx))
            return None
        ring = self.__dict__.setdefault("_sdr_ring", {})
        key = int(sdr_slot)
        if key in ring and ring[key][0].shape != (self.fbH, self.fbW, 2, 3):
            self._drop_sdr_ring(keep_shape=(self.fbH, self.fbW, 2, 3))
        if key not in ring:
            ring[key] = self._page_locked_zeros((self.fbH, self.fbW, 2, 3))
        a = ring[key][0]
        self._check(self.L.ycge_render_frame_async_sdr(self.ctx, a.ctypes.data_as(C.POINTER(C.c_float))))
        return a

    def Wait(self):
        self._check(self.L.ycge_wait(self.ctx))

    def flight_info(self) -> dict:
        """What the frames-in-flight machinery of this context does (ycge_flight_query): timing only, never a pixel."""
        fi = abi.FlightInfo()
        self._check(self.L.ycge_flight_query(self.ctx, C.byref(fi)))
        return {f: int(getattr(fi, f)) for f, _ in abi.FlightInfo._fields_}

    def async_trace_ms(self, capacity: int = 1024) -> np.ndarray:
        """Durations (ms) of the trace launches of the frames queued since the last call (waits for them), oldest first."""
        a = np.zeros(capacity, dtype=np.float32)
        n = C.c_int32()
        self._check(self.L.ycge_async_trace_times(self.ctx, a.ctypes.data_as(C.POINTER(C.c_float)), capacity, C.byref(n)))
        return a[:n.value].copy()

    # ---------------------------------------------------------------- multi-GPU halves
    def tile_slab_bytes(self) -> int:
        n = C.c_size_t()
        self._check(self.L.ycge_tile_slab_bytes(self.ctx, C.byref(n)))
        return n.value

    def trace_tiles(self, d_slab_ptr: int, stream_ptr: int = 0, want_stats: bool = False):
        st = C.byref(self.stats) if want_stats else None
        self._check(self.L.ycge_trace_tiles(self.ctx, C.c_void_p(d_slab_ptr), C.c_void_p(stream_ptr), st))

    def resolve_gathered(self, d_all_slabs_ptr: int, stream_ptr: int = 0, want_stats: bool = False, want_sdr: bool = False):
        """TAA on the gathered frame; with want_sdr also the denoise / exposure / tonemap stage (needs slab_albedo), returns the SDR array."""
        st = C.byref(self.stats) if want_stats else None
        sdr = np.zeros((self.fbH, self.fbW, 2, 3), dtype=np.float32) if want_sdr else None
        ptr = sdr.ctypes.data_as(C.POINTER(C.c_float)) if want_sdr else None
        self._check(self.L.ycge_resolve_gathered(self.ctx, C.c_void_p(d_all_slabs_ptr), C.c_void_p(stream_ptr), ptr, st))
        return sdr

    # the tile-resident form (include/ycge.h): TAA on this rank's own tiles, a halo exchange instead of the all-gather of slabs
    def halo_counts(self):
        w = self.cfg.world_size
        s, r = (C.c_int64 * w)(), (C.c_int64 * w)()
        self._check(self.L.ycge_halo_counts(self.ctx, s, r))
        return list(s), list(r)

    def history_slab_bytes(self) -> int:
        n = C.c_size_t()
        self._check(self.L.ycge_history_slab_bytes(self.ctx, C.byref(n)))
        return n.value

    def trace_tiles_resident(self, d_halo_send_ptr: int, stream_ptr: int = 0, want_stats: bool = False):
        st = C.byref(self.stats) if want_stats else None
        self._check(self.L.ycge_trace_tiles_resident(self.ctx, C.c_void_p(d_halo_send_ptr), C.c_void_p(stream_ptr), st))

    def trace_tiles_resident_batch(self, poses, d_halo_send_ptrs, stream_ptr: int = 0):
        """n consecutive frames of this rank's tiles in one launch (ycge_trace_tiles_resident_batch): poses = n x (pos, yaw, pitch, fov)."""
        n = len(poses)
        flat = (C.c_float * (6 * n))()
        for k, (pos, yaw, pitch, fov) in enumerate(poses):
            flat[6 * k:6 * k + 6] = [pos[0], pos[1], pos[2], yaw, pitch, fov]
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in d_halo_send_ptrs])
        self._check(self.L.ycge_trace_tiles_resident_batch(self.ctx, n, flat, ptrs, C.c_void_p(stream_ptr)))

    def resolve_tiles_resident(self, d_halo_recv_ptr: int, d_history_slab_ptr: int = 0, stream_ptr: int = 0, want_stats: bool = False):
        st = C.byref(self.stats) if want_stats else None
        self._check(self.L.ycge_resolve_tiles_resident(self.ctx, C.c_void_p(d_halo_recv_ptr), C.c_void_p(d_history_slab_ptr), C.c_void_p(stream_ptr), st))

    def unpack_history(self, d_all_history_slabs_ptr: int, stream_ptr: int = 0):
        self._check(self.L.ycge_unpack_history(self.ctx, C.c_void_p(d_all_history_slabs_ptr), C.c_void_p(stream_ptr)))

    # ---------------------------------------------------------------- tests only
    def set_frame_counter(self, n: int):
        self._check(self.L.ycge_set_frame_counter(self.ctx, n))

    def timed_steps(self) -> int:
        """Cumulative traversal-loop steps (summed over lanes, all devices) of the timed kernel instances (ycge_read_timed_steps)."""
        n = C.c_uint64()
        self._check(self.L.ycge_read_timed_steps(self.ctx, C.byref(n)))
        return int(n.value)

    def read(self, which: int) -> np.ndarray:
        dt, n = abi.BUFFER_LAYOUT[which]
        shape = (self.hiH, self.hiW, n) if n > 1 else (self.hiH, self.hiW)
        a = np.zeros(shape, dtype=dt)
        self._check(self.L.ycge_read_buffer(self.ctx, which, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def accel(self, which: int, index: int = 0) -> np.ndarray:
        n = C.c_size_t()
        self._check(self.L.ycge_accel_size(self.ctx, which, index, C.byref(n)))
        dt = NODE_DTYPE if which in (abi.ACCEL_SCENE_NODES, abi.ACCEL_MESH_NODES) else np.dtype("<i4")
        a = np.zeros(n.value // dt.itemsize, dtype=dt)
        if n.value:
            self._check(self.L.ycge_read_accel(self.ctx, which, index, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int32()
        self._check(self.L.ycge_device_info(self.ctx, name, 256, C.byref(cu)))
        return name.value.decode(), cu.value
