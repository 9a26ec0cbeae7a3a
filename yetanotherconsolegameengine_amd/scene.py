"""Host-side mirror of the reference's Scene / primitive builder API.

The C# host keeps its own builders (ConsoleGame/RayTracing/Scenes/*.cs,
Objects/*.cs); this module mirrors their names and argument meaning so that
tests, bench.py and the Python driver describe scenes the way the reference
does, and flattens a scene into the POD `ycge_scene` the C-ABI takes
(include/ycge.h).  It holds NO intersection or shading code.

All numeric narrowing follows the C# constructors: `new Vec3(double,double,double)`
casts each component to float (Vec3.cs:21-26); Material doubles are narrowed
to float at the ABI (Material.cs:7-18, see ycge.h).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import abi

f32 = np.float32


def vec3(x, y, z):
    """new Vec3(x, y, z): three binary32 components."""
    return (float(f32(x)), float(f32(y)), float(f32(z)))


ZERO = vec3(0, 0, 0)
ONE = vec3(1, 1, 1)


class Texture:
    """Renderer/Texture.cs:13-23, a static texture: `pixels[y * width + x]` = RGBA32.ToInt() (byte 0 = r, 1 = g, 2 = b, 3 = a,
    RGBA32.cs:14-31).  Built from an (h, w, 3 or 4) uint8 RGB(A) array; live video textures (Texture.cs:113-140) have no pixels
    to pass and are not mirrored."""

    def __init__(self, rgba):
        a = np.asarray(rgba, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] not in (3, 4) or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError("texture must be (height, width, 3 or 4) uint8")
        if a.shape[2] == 3:
            a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], axis=2)
        self.height, self.width = int(a.shape[0]), int(a.shape[1])
        a = a.astype(np.uint32)
        self.pixels = np.ascontiguousarray((a[..., 0] | (a[..., 1] << 8) | (a[..., 2] << 16) | (a[..., 3] << 24)).reshape(-1), dtype=np.uint32)
        self.frame_bpp = 0


class LiveTexture:
    """`new Texture(IFrameReader reader, useRGBA, flipU, flipV)`, Renderer/Texture.cs:51-66: a texture whose pixels are the reader's
    CURRENT frame (camera / video), BGR (3 bytes) or BGRA (4) per pixel, sampled by SampleBilinear's live branch (Texture.cs:113-140).
    `frame` is an (h, w, 3 or 4) uint8 array in the reader's byte order (B, G, R[, A]); set_frame() swaps in the next one - the
    renderer's UpdateTexture then hands it to the library (the host side of IFrameReader.GetCurrentFramePtr)."""

    def __init__(self, frame, flipU: bool = False, flipV: bool = False):
        self.flipU, self.flipV = bool(flipU), bool(flipV)
        self.set_frame(frame)

    def set_frame(self, frame):
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] not in (3, 4) or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError("a frame must be (height, width, 3 or 4) uint8, bytes in B, G, R[, A] order")
        if hasattr(self, "frame") and a.shape != self.frame.shape:
            raise ValueError("a live texture keeps its size and format")
        self.frame = a
        self.height, self.width, self.frame_bpp = int(a.shape[0]), int(a.shape[1]), int(a.shape[2])


@dataclass
class Material:
    """RayTracing/Material.cs:7-46.  DiffuseTexture / TextureWeight / UVScale (Material.cs:16-18, doubles) feed SampleAlbedo
    (RaytraceRenderer.cs:724-735); a material with a texture is passed as YCGE_MAT_TEXTURED."""
    Albedo: tuple
    Specular: float = 0.0
    Reflectivity: float = 0.0
    Emission: tuple = ZERO
    Transparency: float = 0.0
    IndexOfRefraction: float = 1.5
    TransmissionColor: tuple = ONE
    # checker extension of the delegate shapes (Scenes.cs:418-428)
    Kind: int = abi.MAT_CONSTANT
    AlbedoB: tuple = ZERO
    CheckerScale: float = 1.0
    DiffuseTexture: Optional[Texture] = None
    TextureWeight: float = 1.0
    UVScale: float = 1.0


# material delegates Func<Vec3,Vec3,float,Material>, Scenes/Scenes.cs:408-428
def Solid(albedo) -> Material:
    return Material(Albedo=albedo)


def Emissive(emission) -> Material:
    return Material(Albedo=ZERO, Emission=emission)


def Checker(a, b, scale: float) -> Material:
    return Material(Albedo=a, Kind=abi.MAT_CHECKER, AlbedoB=b, CheckerScale=float(f32(scale)))


@dataclass
class Hittable:
    pass


@dataclass
class Sphere(Hittable):           # BoundedObjects.cs:7-18
    Center: tuple
    Radius: float
    Mat: Material


@dataclass
class Plane(Hittable):            # Surfaces.cs:19-28
    Point: tuple
    Normal: tuple
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class Disk(Hittable):             # Surfaces.cs:84-94
    Center: tuple
    Normal: tuple
    Radius: float
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class XYRect(Hittable):           # Surfaces.cs:158-171
    X0: float
    X1: float
    Y0: float
    Y1: float
    Z: float
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class XZRect(Hittable):           # Surfaces.cs:230-243
    X0: float
    X1: float
    Z0: float
    Z1: float
    Y: float
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class YZRect(Hittable):           # Surfaces.cs:302-315
    Y0: float
    Y1: float
    Z0: float
    Z1: float
    X: float
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class Box(Hittable):              # BoundedObjects.cs:78-90
    Min: tuple
    Max: tuple
    MaterialFunc: Material
    Specular: float
    Reflectivity: float


@dataclass
class CylinderY(Hittable):        # BoundedObjects.cs:128-137
    Center: tuple
    Radius: float
    YMin: float
    YMax: float
    Capped: bool
    Mat: Material


@dataclass
class Triangle(Hittable):         # Objects/Triangle.cs:30-35
    A: tuple
    B: tuple
    C: tuple
    Mat: Material


@dataclass
class Mesh(Hittable):             # RayTracing/Mesh.cs:16-21
    """triangles: float32 array [n,3,3] (A,B,C per triangle), already transformed.  TriMaterials / TriMaterialIndex: the ABI's optional
    per-triangle material (`ycge_mesh.tri_material`, include/ycge.h) - a palette of Materials and, per triangle, an index into it."""
    Triangles: np.ndarray
    Mat: Material
    TriMaterials: Optional[List[Material]] = None
    TriMaterialIndex: Optional[np.ndarray] = None


@dataclass
class VolumeGrid(Hittable):       # Objects/VolumeGrid.cs:55
    """cells: int32 array [nx,ny,nz,2] = the ctor's (matId, metaId)[,,]."""
    Cells: np.ndarray
    MinCorner: tuple
    VoxelSize: tuple
    MaterialLookup: Callable[[int, int], Material]
    EnableWireframe: bool = True
    WireWidthFraction: float = 0.06
    WireMaxDistance: float = 16.0


@dataclass
class PointLight:                 # Objects/PointLight.cs:9-14
    Position: tuple
    Color: tuple
    Intensity: float


@dataclass
class AmbientLight:               # Objects/AmbientLight.cs:8-12
    Color: tuple
    Intensity: float


@dataclass
class Scene:
    """Scenes/Scene.cs:12-24 — the data members the tracer reads."""
    Objects: List[Hittable] = field(default_factory=list)
    Lights: List[PointLight] = field(default_factory=list)
    BackgroundTop: tuple = vec3(0.6, 0.8, 1.0)
    BackgroundBottom: tuple = vec3(1.0, 1.0, 1.0)
    Ambient: AmbientLight = field(default_factory=lambda: AmbientLight(vec3(1.0, 1.0, 1.0), 0.075))
    DefaultFovDeg: float = 45.0
    DefaultCameraPos: tuple = vec3(0.0, 1.0, 0.0)
    DefaultYaw: float = 0.0
    DefaultPitch: float = 0.0
    IsVolumeScene: bool = False   # `scene is VolumeScene`, RaytraceRenderer.cs:761
    HasDynamicTextures: bool = False   # Scene.cs:30: a texture is rewritten between frames -> TAA history restarts every frame (RaytraceRenderer.cs:171)

    def Add(self, h: Hittable) -> Hittable:   # Scene.cs:505-511 (entity layer collapses to Objects order)
        self.Objects.append(h)
        return h


class NeedsUpload(ValueError):
    """flatten(scene, against=uploaded): the objects use a record the uploaded scene does not hold - upload the scene again"""


class FlatScene:
    """A `ycge_scene` plus the ctypes/numpy storage that keeps its pointers alive."""

    def __init__(self, scene: Scene, against: "FlatScene | None" = None):
        """against: only Scene.Objects again, numbered against the materials / meshes / grids of an UPLOADED scene - what
        ycge_scene_update_objects wants after entities moved, appeared or left (bindings/csharp/SceneFlattener.cs: ObjectsAgainst).  Raises
        NeedsUpload when the objects use a Material, Mesh or VolumeGrid object the uploaded scene did not hold."""
        self._keep = []
        mats: List[Material] = []
        mat_index = {} if against is None else dict(against._mat_index)
        self._mat_index, self._mesh_index, self._grid_index = mat_index, {}, {}

        def mat_id(m: Material) -> int:
            key = id(m)
            if key not in mat_index:
                if against is not None:
                    raise NeedsUpload("a material the uploaded scene does not hold")
                mat_index[key] = len(mats)
                mats.append(m)
                self._keep.append(m)
            return mat_index[key]

        prims, meshes, grids = [], [], []

        def prim(ptype, material=-1, p=(), specular=0.0, reflectivity=0.0, ref=-1):
            q = abi.Prim()
            q.type, q.material, q.ref = ptype, material, ref
            vals = [float(f32(v)) for v in p] + [0.0] * (12 - len(p))
            q.p = (C.c_float * 12)(*vals)
            q.specular, q.reflectivity = float(f32(specular)), float(f32(reflectivity))
            prims.append(q)

        for o in scene.Objects:
            if isinstance(o, Sphere):
                prim(abi.PRIM_SPHERE, mat_id(o.Mat), [*o.Center, o.Radius])
            elif isinstance(o, Plane):
                prim(abi.PRIM_PLANE, mat_id(o.MaterialFunc), [*o.Point, *o.Normal], o.Specular, o.Reflectivity)
            elif isinstance(o, Disk):
                prim(abi.PRIM_DISK, mat_id(o.MaterialFunc), [*o.Center, *o.Normal, o.Radius], o.Specular, o.Reflectivity)
            elif isinstance(o, XYRect):
                prim(abi.PRIM_XYRECT, mat_id(o.MaterialFunc), [o.X0, o.X1, o.Y0, o.Y1, o.Z], o.Specular, o.Reflectivity)
            elif isinstance(o, XZRect):
                prim(abi.PRIM_XZRECT, mat_id(o.MaterialFunc), [o.X0, o.X1, o.Z0, o.Z1, o.Y], o.Specular, o.Reflectivity)
            elif isinstance(o, YZRect):
                prim(abi.PRIM_YZRECT, mat_id(o.MaterialFunc), [o.Y0, o.Y1, o.Z0, o.Z1, o.X], o.Specular, o.Reflectivity)
            elif isinstance(o, Box):
                prim(abi.PRIM_BOX, mat_id(o.MaterialFunc), [*o.Min, *o.Max], o.Specular, o.Reflectivity)
            elif isinstance(o, CylinderY):
                prim(abi.PRIM_CYLINDER_Y, mat_id(o.Mat), [*o.Center, o.Radius, o.YMin, o.YMax, 1.0 if o.Capped else 0.0])
            elif isinstance(o, Triangle):
                prim(abi.PRIM_TRIANGLE, mat_id(o.Mat), [*o.A, *o.B, *o.C])
            elif isinstance(o, Mesh) and against is not None:
                if id(o) not in against._mesh_index: raise NeedsUpload("a mesh the uploaded scene does not hold")
                prim(abi.PRIM_MESH, -1, [], ref=against._mesh_index[id(o)])
            elif isinstance(o, VolumeGrid) and against is not None:
                if id(o) not in against._grid_index: raise NeedsUpload("a voxel grid the uploaded scene does not hold")
                prim(abi.PRIM_VOLUME_GRID, -1, [], ref=against._grid_index[id(o)])
            elif isinstance(o, Mesh):
                self._mesh_index[id(o)] = len(meshes)
                tris = np.ascontiguousarray(o.Triangles, dtype=np.float32).reshape(-1, 9)
                self._keep.append(tris)
                m = abi.Mesh()
                m.triangles = tris.ctypes.data_as(C.POINTER(C.c_float))
                m.n_triangles = tris.shape[0]
                m.material = mat_id(o.Mat)
                m.tri_material = None
                if o.TriMaterials is not None:
                    ids = np.array([mat_id(mm) for mm in o.TriMaterials], np.int32)
                    tm = np.ascontiguousarray(ids[np.asarray(o.TriMaterialIndex, np.int64)], dtype=np.int32)
                    assert tm.shape == (tris.shape[0],)
                    self._keep.append(tm)
                    m.tri_material = tm.ctypes.data_as(C.POINTER(C.c_int32))
                prim(abi.PRIM_MESH, -1, [], ref=len(meshes))
                meshes.append(m)
            elif isinstance(o, VolumeGrid):
                self._grid_index[id(o)] = len(grids)
                cells = np.ascontiguousarray(o.Cells, dtype=np.int32)
                assert cells.ndim == 4 and cells.shape[3] == 2
                self._keep.append(cells)
                pairs = np.unique(cells.reshape(-1, 2), axis=0)
                lut = []
                for mid, meta in pairs:
                    if mid <= 0:
                        continue
                    lut.append((int(mid), int(meta), mat_id(o.MaterialLookup(int(mid), int(meta)))))
                lut_arr = (abi.VoxelLookup * max(1, len(lut)))()
                for i, (a, b, c) in enumerate(lut):
                    lut_arr[i].mat_id, lut_arr[i].meta_id, lut_arr[i].material = a, b, c
                self._keep.append(lut_arr)
                g = abi.Grid()
                g.nx, g.ny, g.nz = cells.shape[0], cells.shape[1], cells.shape[2]
                g.min_corner = abi.Vec3(*o.MinCorner)
                g.voxel_size = abi.Vec3(*o.VoxelSize)
                g.cells = cells.ctypes.data_as(C.POINTER(C.c_int32))
                g.lookup = C.cast(lut_arr, C.POINTER(abi.VoxelLookup))
                g.n_lookup = len(lut)
                g.default_material = -1
                g.wireframe = 1 if o.EnableWireframe else 0
                g.wire_width_fraction = float(f32(o.WireWidthFraction))
                g.wire_max_distance = float(f32(o.WireMaxDistance))
                prim(abi.PRIM_VOLUME_GRID, -1, [], ref=len(grids))
                grids.append(g)
            else:
                raise TypeError(f"not a Hittable the path knows: {type(o).__name__}")

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self._keep.append(a)
            return a

        if against is not None:          # the uploaded scene's tables with these objects: what the library holds after ycge_scene_update_objects
            self._keep.append(against)
            self._mesh_index, self._grid_index = against._mesh_index, against._grid_index
            self.materials, self.meshes, self.grids, self.lights, self.textures = against.materials, against.meshes, against.grids, against.lights, against.textures
            self.texture_objects, self.n_triangles = against.texture_objects, against.n_triangles
            self.prims = arr(abi.Prim, prims)
            sc = abi.Scene()
            C.memmove(C.byref(sc), C.byref(against.struct), C.sizeof(sc))
            sc.prims, sc.n_prims = C.cast(self.prims, C.POINTER(abi.Prim)), len(prims)
            self.struct = sc
            return

        mat_structs = []
        textures: List[Texture] = []
        tex_index = {}
        for m in mats:
            s = abi.Material()
            s.kind = m.Kind
            s.texture, s.texture_weight, s.uv_scale = -1, float(m.TextureWeight), float(m.UVScale)
            if m.DiffuseTexture is not None:
                if m.Kind != abi.MAT_CONSTANT:
                    raise ValueError("a texture belongs to a plain Material (the checker delegate builds untextured ones)")
                if id(m.DiffuseTexture) not in tex_index:
                    tex_index[id(m.DiffuseTexture)] = len(textures)
                    textures.append(m.DiffuseTexture)
                s.kind, s.texture = abi.MAT_TEXTURED, tex_index[id(m.DiffuseTexture)]
            s.albedo = abi.Vec3(*m.Albedo)
            s.albedo_b = abi.Vec3(*m.AlbedoB)
            s.checker_scale = m.CheckerScale
            s.specular = float(f32(m.Specular))
            s.reflectivity = float(f32(m.Reflectivity))
            s.emission = abi.Vec3(*m.Emission)
            s.transparency = float(f32(m.Transparency))
            s.index_of_refraction = float(f32(m.IndexOfRefraction))
            s.transmission_color = abi.Vec3(*m.TransmissionColor)
            mat_structs.append(s)
        lights = []
        for l in scene.Lights:
            s = abi.Light()
            s.position, s.color, s.intensity = abi.Vec3(*l.Position), abi.Vec3(*l.Color), float(f32(l.Intensity))
            lights.append(s)

        self.materials = arr(abi.Material, mat_structs)
        self.prims = arr(abi.Prim, prims)
        self.meshes = arr(abi.Mesh, meshes)
        self.grids = arr(abi.Grid, grids)
        self.lights = arr(abi.Light, lights)
        tex_structs = []
        for t in textures:
            ts = abi.Texture()
            if getattr(t, "frame_bpp", 0):
                ts.width, ts.height, ts.frame_bytes_per_pixel = t.width, t.height, t.frame_bpp
                ts.flip_u, ts.flip_v = int(t.flipU), int(t.flipV)
                ts.frame = t.frame.ctypes.data_as(C.POINTER(C.c_uint8))
            else:
                ts.width, ts.height, ts.pixels = t.width, t.height, t.pixels.ctypes.data_as(C.POINTER(C.c_uint32))
            self._keep.append(t)
            tex_structs.append(ts)
        self.textures = arr(abi.Texture, tex_structs)
        self.texture_objects = textures          # index in ycge_scene.textures -> the Texture / LiveTexture it came from

        sc = abi.Scene()
        sc.materials, sc.n_materials = C.cast(self.materials, C.POINTER(abi.Material)), len(mat_structs)
        sc.prims, sc.n_prims = C.cast(self.prims, C.POINTER(abi.Prim)), len(prims)
        sc.meshes, sc.n_meshes = C.cast(self.meshes, C.POINTER(abi.Mesh)), len(meshes)
        sc.grids, sc.n_grids = C.cast(self.grids, C.POINTER(abi.Grid)), len(grids)
        sc.lights, sc.n_lights = C.cast(self.lights, C.POINTER(abi.Light)), len(lights)
        sc.ambient_color = abi.Vec3(*scene.Ambient.Color)
        sc.ambient_intensity = float(f32(scene.Ambient.Intensity))
        sc.background_top = abi.Vec3(*scene.BackgroundTop)
        sc.background_bottom = abi.Vec3(*scene.BackgroundBottom)
        sc.is_volume_scene = 1 if scene.IsVolumeScene else 0
        sc.has_dynamic_textures = 1 if getattr(scene, 'HasDynamicTextures', False) else 0
        sc.textures, sc.n_textures = C.cast(self.textures, C.POINTER(abi.Texture)), len(tex_structs)
        self.struct = sc
        self.n_triangles = int(sum(m.n_triangles for m in meshes))

    def byref(self):
        return C.byref(self.struct)


def flatten(scene: Scene, against: "FlatScene | None" = None) -> FlatScene:
    return FlatScene(scene, against)
