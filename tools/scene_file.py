"""YSC1 - a flattened scene (`ycge_scene` of include/ycge.h) as ONE little-endian file, plus the console size and the camera pose.

Why: the reference is C#/.NET and cannot run here, so nothing it produces pins the oracle.  tools/ReferenceDump/ is
a C# console project that rebuilds a scene from this file with the REFERENCE's own constructors (Sphere, Box, Mesh, VolumeGrid, ...),
runs the reference's own RaytraceRenderer on it and dumps its buffers; tests/test_reference_goldens.py then holds the oracle and the HIP
path to those dumps.  The file is the struct bytes of the C-ABI records themselves - the layouts bindings/csharp/Ycge.cs mirrors and
tests/test_csharp_binding.py checks - so the C# reader is `MemoryMarshal.Read<YMaterial>(...)`, not a second parser.

    python tools/scene_file.py <config 1..5> <out.ysc> [--t01 0.25] [--size WxH]

Layout (all little-endian; `struct` = the bytes of the ctypes / C struct, pointers zeroed):
    char[4] "YSC1"; u32 version = 1
    i32 fb_width, fb_height, super_sample; f32 fov_deg; f32 pos[3], yaw, pitch
    struct ycge_scene
    struct ycge_material[n_materials]; struct ycge_prim[n_prims]; struct ycge_light[n_lights]
    per mesh:    struct ycge_mesh;  f32 triangles[9 * n_triangles];  u32 has_tri_material;  i32 tri_material[n_triangles] if it has
    per grid:    struct ycge_grid;  i32 cells[2 * nx * ny * nz];  struct ycge_voxel_lookup[n_lookup]
    per texture: struct ycge_texture;  u32 pixels[width * height] (static)  or  u8 frame[width * height * frame_bytes_per_pixel] (live)
"""
from __future__ import annotations

import ctypes as C
import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from yetanotherconsolegameengine_amd import abi  # noqa: E402

MAGIC = b"YSC1"


def _zeroed(s, *pointer_fields):
    """the bytes of a ctypes struct with its pointer members set to null (addresses mean nothing in a file)"""
    t = type(s)()
    C.memmove(C.byref(t), C.byref(s), C.sizeof(s))
    for f in pointer_fields:
        setattr(t, f, None)
    return bytes(t)


def write_ysc(path, flat, fb_width, fb_height, super_sample, pose, compress=None):
    """`flat`: a FlatScene (yetanotherconsolegameengine_amd.scene.flatten); pose: {"pos", "yaw", "pitch", "fov"}."""
    sc = flat.struct
    out = [MAGIC, struct.pack("<I", 1),
           struct.pack("<iiif", fb_width, fb_height, super_sample, pose["fov"]),
           struct.pack("<5f", *pose["pos"], pose["yaw"], pose["pitch"]),
           _zeroed(sc, "materials", "prims", "meshes", "grids", "lights", "textures")]
    out += [bytes(flat.materials[i]) for i in range(sc.n_materials)]
    out += [bytes(flat.prims[i]) for i in range(sc.n_prims)]
    out += [bytes(flat.lights[i]) for i in range(sc.n_lights)]
    for i in range(sc.n_meshes):
        m = flat.meshes[i]
        out.append(_zeroed(m, "triangles", "tri_material"))
        out.append(np.ctypeslib.as_array(m.triangles, shape=(m.n_triangles * 9,)).astype("<f4").tobytes())
        has = bool(m.tri_material)
        out.append(struct.pack("<I", 1 if has else 0))
        if has:
            out.append(np.ctypeslib.as_array(m.tri_material, shape=(m.n_triangles,)).astype("<i4").tobytes())
    for i in range(sc.n_grids):
        g = flat.grids[i]
        out.append(_zeroed(g, "cells", "lookup"))
        out.append(np.ctypeslib.as_array(g.cells, shape=(2 * g.nx * g.ny * g.nz,)).astype("<i4").tobytes())
        out += [bytes(g.lookup[k]) for k in range(g.n_lookup)]
    for i in range(sc.n_textures):
        t = flat.textures[i]
        out.append(_zeroed(t, "pixels", "frame"))
        if t.frame_bytes_per_pixel:
            n = t.width * t.height * t.frame_bytes_per_pixel
            out.append(bytes(n) if not t.frame else np.ctypeslib.as_array(t.frame, shape=(n,)).tobytes())
        else:
            out.append(np.ctypeslib.as_array(t.pixels, shape=(t.width * t.height,)).astype("<u4").tobytes())
    data = b"".join(out)
    if compress or (compress is None and str(path).endswith(".gz")):           # (voxel worlds are mostly air: 8 MB -> ~100 KB; both readers take either form, told apart by gzip's magic number)
        import gzip
        data = gzip.compress(data, compresslevel=9, mtime=0)
    Path(path).write_bytes(data)


class LoadedScene:
    """A YSC1 file back as a `ycge_scene` (ctypes) with the arrays it points to kept alive - what the C# reader does, in Python:
    the round-trip check of the format (tests/test_reference_dump_tools.py) and the input of tests/test_reference_goldens.py."""

    def __init__(self, path):
        b = Path(path).read_bytes()
        if b[:2] == b"\x1f\x8b":
            import gzip
            b = gzip.decompress(b)
        if b[:4] != MAGIC or struct.unpack_from("<I", b, 4)[0] != 1:
            raise ValueError(f"{path}: not a YSC1 version 1 file")
        o = 8
        self.fb_width, self.fb_height, self.super_sample, fov = struct.unpack_from("<iiif", b, o); o += 16
        px, py, pz, yaw, pitch = struct.unpack_from("<5f", b, o); o += 20
        self.pose = {"pos": (px, py, pz), "yaw": yaw, "pitch": pitch, "fov": fov}
        self._keep = []

        def take(ctype, n=1):
            nonlocal o
            a = (ctype * max(1, n)).from_buffer_copy(b[o:o + C.sizeof(ctype) * n].ljust(C.sizeof(ctype) * max(1, n), b"\0"))
            o += C.sizeof(ctype) * n
            self._keep.append(a)
            return a

        def blob(dtype, n):
            nonlocal o
            a = np.frombuffer(b, dtype=dtype, count=n, offset=o).copy()
            o += a.nbytes
            self._keep.append(a)
            return a

        sc = take(abi.Scene)[0]
        self.materials = take(abi.Material, sc.n_materials)
        self.prims = take(abi.Prim, sc.n_prims)
        self.lights = take(abi.Light, sc.n_lights)
        self.meshes = (abi.Mesh * max(1, sc.n_meshes))()
        for i in range(sc.n_meshes):
            m = take(abi.Mesh)[0]
            tris = blob("<f4", 9 * m.n_triangles)
            m.triangles = tris.ctypes.data_as(C.POINTER(C.c_float))
            (has,) = struct.unpack_from("<I", b, o); o += 4
            m.tri_material = blob("<i4", m.n_triangles).ctypes.data_as(C.POINTER(C.c_int32)) if has else None
            self.meshes[i] = m
        self.grids = (abi.Grid * max(1, sc.n_grids))()
        for i in range(sc.n_grids):
            g = take(abi.Grid)[0]
            g.cells = blob("<i4", 2 * g.nx * g.ny * g.nz).ctypes.data_as(C.POINTER(C.c_int32))
            g.lookup = C.cast(take(abi.VoxelLookup, g.n_lookup), C.POINTER(abi.VoxelLookup))
            self.grids[i] = g
        self.textures = (abi.Texture * max(1, sc.n_textures))()
        for i in range(sc.n_textures):
            t = take(abi.Texture)[0]
            if t.frame_bytes_per_pixel:
                t.frame = blob("u1", t.width * t.height * t.frame_bytes_per_pixel).ctypes.data_as(C.POINTER(C.c_uint8))
            else:
                t.pixels = blob("<u4", t.width * t.height).ctypes.data_as(C.POINTER(C.c_uint32))
            self.textures[i] = t
        if o != len(b):
            raise ValueError(f"{path}: {len(b) - o} bytes left over")
        sc.materials = C.cast(self.materials, C.POINTER(abi.Material))
        sc.prims = C.cast(self.prims, C.POINTER(abi.Prim))
        sc.lights = C.cast(self.lights, C.POINTER(abi.Light))
        sc.meshes = C.cast(self.meshes, C.POINTER(abi.Mesh))
        sc.grids = C.cast(self.grids, C.POINTER(abi.Grid))
        sc.textures = C.cast(self.textures, C.POINTER(abi.Texture))
        self.struct = sc
        self.n_triangles = int(sum(self.meshes[i].n_triangles for i in range(sc.n_meshes)))
        self.texture_objects = []

    def byref(self):
        return C.byref(self.struct)


def main(argv):
    import argparse
    from yetanotherconsolegameengine_amd import scenes
    from yetanotherconsolegameengine_amd.scene import flatten
    ap = argparse.ArgumentParser(description="one of the five BASELINE configurations as a YSC1 file for tools/ReferenceDump")
    ap.add_argument("config", type=int)
    ap.add_argument("out")
    ap.add_argument("--t01", type=float, default=0.25, help="config 5: day phase (0.25 = the survey's; 0.5 = noon)")
    ap.add_argument("--size", default=None, help="console size WxH instead of the configuration's (e.g. 96x27 for a quick run)")
    ap.add_argument("--gzip", action="store_true", help="write the file gzip-compressed whatever its name (both readers tell by the magic number)")
    ap.add_argument("--small", action="store_true", help="configs 4 and 5: the reduced scene of the test suite (8 712 triangles; a 96 x 128 x 96 voxel world) - files small enough to commit")
    a = ap.parse_args(argv)
    sc, w, h, ss, pose = scenes.config_scene(a.config, small=a.small, t01=a.t01)
    if a.size:
        w, h = (int(v) for v in a.size.lower().split("x"))
    write_ysc(a.out, flatten(sc), w, h, ss, pose, compress=True if a.gzip else None)
    print(f"{a.out}: config {a.config}, console {w}x{h}, ss {ss}, {Path(a.out).stat().st_size} bytes")


if __name__ == "__main__":
    main(sys.argv[1:])
