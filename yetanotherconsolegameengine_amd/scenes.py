"""The five BASELINE.json configuration scenes, described through the builder API.

Values are the reference's (cited per builder); the two assets the reference
cannot supply here are replaced as SURVEY.md §8(d) states:
  * config 4: `xyzrgb_dragon.obj` is a missing blob -> a seeded procedural
    871,200-triangle (2,3) torus-knot tube with fBm radial displacement;
  * config 5: the procedural island generator (WorldGeneration/*, out of scope)
    -> a seeded value-noise heightfield with the same extents and palette.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Tuple

import numpy as np

from . import mesh_loader
from .scene import (AmbientLight, Box, Checker, Emissive, Material, Mesh, Plane, PointLight, Scene, Solid, Sphere,
                    VolumeGrid, XYRect, XZRect, YZRect, f32, vec3, ZERO)

# ---------------------------------------------------------------- configs 1, 2


def BuildCornellBox() -> Scene:
    """Scenes/Scenes.cs:269-309."""
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.00)
    white = Solid(vec3(0.82, 0.82, 0.82))
    red = Solid(vec3(0.80, 0.10, 0.10))
    green = Solid(vec3(0.10, 0.80, 0.10))
    light_emit = Emissive(vec3(0.6, 0.6, 0.6))
    xL, xR, yB, yT, zF, zB = -3.0, 3.0, 0.0, 5.0, 0.0, -5.0
    s.Add(YZRect(yB, yT, zB, zF, xL, red, 0.0, 0.0))
    s.Add(YZRect(yB, yT, zB, zF, xR, green, 0.0, 0.0))
    s.Add(XZRect(xL, xR, zB, zF, yB, white, 0.0, 0.0))
    s.Add(XZRect(xL, xR, zB, zF, yT, white, 0.0, 0.0))
    s.Add(XYRect(xL, xR, yB, yT, zB, white, 0.0, 0.0))
    lx0, lx1, lz0, lz1 = f32(-0.9), f32(0.9), f32(-3.2), f32(-2.2)
    ly = f32(yT) - f32(0.01)
    s.Add(XZRect(lx0, lx1, lz0, lz1, ly, light_emit, 0.0, 0.0))
    s.Add(Box(vec3(-2.2, 0.0, -4.0), vec3(-0.8, 1.0, -2.8), white, 0.0, 0.0))
    s.Add(Box(vec3(0.6, 0.0, -3.3), vec3(2.0, 1.8, -2.1), white, 0.0, 0.0))
    s.Lights.append(PointLight(vec3(0.0, 4.6, -2.7), vec3(1.0, 1.0, 1.0), 20.0))
    s.BackgroundTop = vec3(0.0, 0.0, 0.0)
    s.BackgroundBottom = vec3(0.0, 0.0, 0.0)
    return s


def BuildMirrorSpheresOnChecker() -> Scene:
    """Scenes/Scenes.cs:311-335 (reflectivities 0.1/0.6/0.85 < MirrorThreshold: all shade diffuse)."""
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.01)
    floor = Checker(vec3(0.8, 0.8, 0.8), vec3(0.15, 0.15, 0.15), 0.6)
    s.Add(XZRect(-8.0, 8.0, -8.0, 4.0, 0.0, floor, 0.1, 0.0))
    gold = Material(vec3(1.0, 0.85, 0.57), 0.25, 0.1, ZERO)
    glassy = Material(vec3(0.9, 0.95, 1.0), 0.0, 0.6, ZERO)
    mirror = Material(vec3(0.98, 0.98, 0.98), 0.0, 0.85, ZERO)
    s.Add(Sphere(vec3(-1.2, 1.0, -2.0), 1.0, gold))
    s.Add(Sphere(vec3(1.3, 1.0, -2.6), 1.0, glassy))
    s.Add(Sphere(vec3(0.0, 0.5, -4.2), 0.5, mirror))
    s.Lights.append(PointLight(vec3(-2.5, 3.5, -1.5), vec3(1.0, 0.95, 0.9), 90.0))
    s.Lights.append(PointLight(vec3(2.0, 2.8, -3.8), vec3(0.9, 0.95, 1.0), 70.0))
    s.BackgroundTop = vec3(0.55, 0.75, 1.0)
    s.BackgroundBottom = vec3(0.95, 0.98, 1.0)
    return s


# ---------------------------------------------------------------- configs 3, 4

# MeshSwatches, Scenes/MeshScenes.cs:19-37,47-53,66-73
def _scale(c, k):
    k = f32(min(max(k, 0.0), 1.0))
    return (float(f32(c[0]) * k), float(f32(c[1]) * k), float(f32(c[2]) * k))


Emerald = _scale((0.0, 1.0, 0.0), 0.85)
Sapphire = _scale((0.0, 0.0, 1.0), 0.85)
Gold = _scale((1.0, 1.0, 0.0), 0.90)


def Matte(albedo, specular=0.10, reflectivity=0.00) -> Material:     # MeshScenes.cs:89-92
    return Material(albedo, specular, reflectivity, ZERO)


def MirrorMat(tint, reflectivity=0.85) -> Material:                   # MeshScenes.cs:94-97
    return Material(tint, 0.0, reflectivity, ZERO)


def NewBaseScene() -> Scene:
    """Scenes/MeshScenes.cs:160-171."""
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.15)
    s.Objects.append(Plane(vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), Solid(vec3(1, 1, 1)), 0.01, 0.00))
    s.Lights.append(PointLight(vec3(0.0, 30.6, -4.2), vec3(1.0, 0.95, 0.88), 110.0))
    s.Lights.append(PointLight(vec3(0.0, 30.0, 4.2), vec3(0.85, 0.90, 1.0), 85.0))
    s.BackgroundTop = vec3(0.0, 0.0, 0.0)
    s.BackgroundBottom = vec3(0.0, 0.0, 0.0)
    return s


# config 3's mesh: the reference's own data file (ConsoleGame/assets/stanford-bunny.obj) as parsed arrays - scene data of the package
# (written by tests/golden/make_fixtures.py bunny in the authoring container, where /root/reference exists)
ASSETS_DIR = Path(__file__).resolve().parent / "assets"
BUNNY_FIXTURE = ASSETS_DIR / "stanford_bunny_mesh.npz"

# benchmark pose for the mesh scenes (the reference default pose faces away from the mesh)
MESH_BENCH_POSE = dict(pos=(0.0, 1.0, -1.0), yaw=float(f32(3.14159274)), pitch=0.0, fov=45.0)


def load_bunny_arrays() -> Tuple[np.ndarray, np.ndarray]:
    """positions/faces of assets/stanford-bunny.obj as parsed by MeshLoader (assets/stanford_bunny_mesh.npz; see tests/golden/make_fixtures.py)."""
    z = np.load(BUNNY_FIXTURE)
    return z["positions"].astype(np.float32), z["faces"].astype(np.int32)


def BuildMeshScene(pos: np.ndarray, faces: np.ndarray, mat: Material, target_pos=(0.0, 0.5, 1.0)) -> Scene:
    """MeshScenes.BuildBunnyScene pattern, MeshScenes.cs:117-124: base scene + auto-grounded mesh."""
    s = NewBaseScene()
    tris = mesh_loader.add_mesh_auto_ground(pos, faces, 1.0, target_pos)
    s.Objects.append(Mesh(tris, mat))
    return s


def BuildBunnyScene() -> Scene:
    pos, faces = load_bunny_arrays()
    return BuildMeshScene(pos, faces, Matte(Emerald, 0.12, 0.00))


def make_torus_knot(nu: int = 1320, nv: int = 330, seed: int = 1337) -> Tuple[np.ndarray, np.ndarray]:
    """Dragon-class stand-in: (2,3) torus-knot tube, nu x nv quads = 2*nu*nv triangles, with a
    seeded fBm radial displacement so that triangle sizes and normals vary like a scanned mesh."""
    u = np.arange(nu, dtype=np.float64) * (2.0 * math.pi / nu)
    v = np.arange(nv, dtype=np.float64) * (2.0 * math.pi / nv)
    p, q = 2.0, 3.0
    r = 2.0 + np.cos(q * u)
    cx, cy, cz = r * np.cos(p * u), np.sin(q * u), r * np.sin(p * u)
    centre = np.stack([cx, cy, cz], axis=1)
    # Frenet-ish frame from finite differences
    tang = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    ref = np.array([0.0, 1.0, 0.0])
    nrm = np.cross(tang, ref)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bin_ = np.cross(tang, nrm)
    rng = np.random.RandomState(seed)
    # fBm: 5 octaves of periodic sinusoid lattice noise with seeded phases
    disp = np.zeros((nu, nv))
    amp, fu, fv = 0.08, 6, 3
    for _ in range(5):
        ph = rng.uniform(0, 2 * math.pi, size=4)
        disp += amp * (np.sin(fu * u[:, None] + ph[0]) * np.cos(fv * v[None, :] + ph[1])
                       + 0.5 * np.sin((fu + 1) * u[:, None] + (fv + 2) * v[None, :] + ph[2]))
        amp *= 0.5
        fu, fv = fu * 2 + 1, fv * 2 + 1
    tube = 0.45 * (1.0 + disp)
    ring = (np.cos(v)[None, :, None] * nrm[:, None, :] + np.sin(v)[None, :, None] * bin_[:, None, :])
    pts = centre[:, None, :] + tube[:, :, None] * ring
    pos = pts.reshape(-1, 3).astype(np.float32)
    iu = np.arange(nu)[:, None]
    iv = np.arange(nv)[None, :]
    a = (iu * nv + iv)
    b = (((iu + 1) % nu) * nv + iv)
    c = (((iu + 1) % nu) * nv + (iv + 1) % nv)
    d = (iu * nv + (iv + 1) % nv)
    faces = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)], axis=0)
    return pos, faces.astype(np.int32)


def BuildDragonStandInScene(nu: int = 1320, nv: int = 330) -> Scene:
    """MeshScenes.BuildDragonScene (MeshScenes.cs:135-143) with the procedural stand-in mesh."""
    pos, faces = make_torus_knot(nu, nv)
    s = BuildMeshScene(pos, faces, MirrorMat(Sapphire, 0.70))
    s.DefaultCameraPos = vec3(0, 10, 0)
    return s


# -------------------------------------------------------------------- config 5

# WorldGenSettings.Blocks, WorldGeneration/WorldGenSettings.cs:10-21
AIR, STONE, DIRT, GRASS, WATER, SAND, WOOD, LEAVES, SNOW, ORE, TALLGRASS, FLOWER = range(12)

_PALETTE16 = [  # Scenes/VoxelMaterialPalette.cs:9-27
    (0.00, 0.00, 0.00), (0.00, 0.00, 0.50), (0.00, 0.50, 0.00), (0.00, 0.50, 0.50), (0.50, 0.00, 0.00),
    (0.50, 0.00, 0.50), (0.50, 0.50, 0.00), (0.75, 0.75, 0.75), (0.50, 0.50, 0.50), (0.00, 0.00, 1.00),
    (0.00, 1.00, 0.00), (0.00, 1.00, 1.00), (1.00, 0.00, 0.00), (1.00, 0.00, 1.00), (1.00, 1.00, 0.00),
    (1.00, 1.00, 1.00)]
_PAL_MATS = [Material(vec3(*c), 0.05, 0.00, ZERO) for c in _PALETTE16]          # PalMat, :29-33


def VoxelMaterialLookup(mat_id: int, meta: int) -> Material:
    """VoxelMaterialPalette.MaterialLookup, Scenes/VoxelMaterialPalette.cs:35-98."""
    clamp = lambda v, lo, hi: lo if v < lo else hi if v > hi else v
    norm = {AIR: (0, 0), STONE: (1, clamp(meta, 0, 2)), DIRT: (2, 0), GRASS: (3, 0), WATER: (4, 0), SAND: (5, 0),
            WOOD: (6, 0), LEAVES: (7, 0), SNOW: (8, 0), ORE: (9, clamp(meta, 0, 2)), TALLGRASS: (10, 0),
            FLOWER: (11, 0)}.get(mat_id, (1, 0))
    k, m = norm
    table = {0: 0, 2: 6, 3: 10, 4: 9, 5: 14, 6: 6, 7: 2, 8: 15, 10: 10, 11: 12}
    if k == 1:
        return _PAL_MATS[{0: 8, 1: 7}.get(m, 15)]
    if k == 9:
        return _PAL_MATS[{0: 0, 1: 7}.get(m, 14)]
    return _PAL_MATS[table.get(k, 7)]


def _value_noise(nx, nz, cell, seed):
    """Seeded bilinear value noise on an integer lattice (pure integer hash + fp64 lerp)."""
    gx, gz = nx // cell + 2, nz // cell + 2
    ix, iz = np.meshgrid(np.arange(gx, dtype=np.uint64), np.arange(gz, dtype=np.uint64), indexing="ij")
    h = (ix * np.uint64(0x9E3779B97F4A7C15) + iz * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(seed) * np.uint64(0x165667B19E3779F9))
    h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
    h ^= h >> np.uint64(31)
    lat = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    x = np.arange(nx, dtype=np.float64) / cell
    z = np.arange(nz, dtype=np.float64) / cell
    x0, z0 = np.floor(x).astype(int), np.floor(z).astype(int)
    fx, fz = x - x0, z - z0
    fx, fz = fx * fx * (3 - 2 * fx), fz * fz * (3 - 2 * fz)
    a = lat[np.ix_(x0, z0)]; b = lat[np.ix_(x0 + 1, z0)]; c = lat[np.ix_(x0, z0 + 1)]; d = lat[np.ix_(x0 + 1, z0 + 1)]
    return (a * (1 - fx)[:, None] + b * fx[:, None]) * (1 - fz)[None, :] + (c * (1 - fx)[:, None] + d * fx[:, None]) * fz[None, :]


def make_island_heightfield(nx=544, nz=544, sea_level=64, max_rise=115, seed=0):
    """Island heightfield stand-in (shape per WorldConfig.cs:32, IslandSettings.cs:8): radial falloff x fBm."""
    n = np.zeros((nx, nz))
    amp, cell, tot = 1.0, 128, 0.0
    for o in range(5):
        n += amp * _value_noise(nx, nz, cell, seed * 16 + o)
        tot += amp
        amp *= 0.5
        cell = max(2, cell // 2)
    n /= tot
    xs = (np.arange(nx) - nx / 2 + 0.5) / (nx / 2)
    zs = (np.arange(nz) - nz / 2 + 0.5) / (nz / 2)
    rad = np.sqrt(xs[:, None] ** 2 + zs[None, :] ** 2)
    fall = np.clip(1.15 - rad, 0.0, 1.0) ** 1.5
    h = sea_level - 20 + (20 + max_rise) * np.clip(n * 1.6 - 0.25, 0, 1) * fall
    return np.clip(np.floor(h), 1, sea_level + max_rise).astype(np.int32)


def make_voxel_world(nx=544, ny=256, nz=544, sea_level=64, max_rise=115, seed=0) -> np.ndarray:
    """int32 [nx,ny,nz,2] (matId, metaId) world: stone/dirt/grass/sand/snow columns + water to sea level."""
    h = np.minimum(make_island_heightfield(nx, nz, sea_level, max_rise, seed), ny - 1)
    y = np.arange(ny, dtype=np.int32)[None, :, None]
    hh = h[:, None, :]
    mat = np.zeros((nx, ny, nz), dtype=np.int32)
    mat[y < hh] = STONE
    mat[(y < hh) & (y >= hh - 4)] = DIRT
    top = (y == hh - 1)
    mat[top & (hh - 1 <= sea_level + 1)] = SAND
    mat[top & (hh - 1 > sea_level + 1)] = GRASS
    mat[top & (hh - 1 > sea_level + 90)] = SNOW
    mat[(y >= hh) & (y <= sea_level)] = WATER
    meta = np.zeros_like(mat)
    # stone meta 0..2 bands so that the palette's three stone shades appear (VoxelMaterialPalette.cs:70-75)
    meta[mat == STONE] = (np.broadcast_to(y, mat.shape)[mat == STONE] // 24) % 3
    return np.stack([mat, meta], axis=-1)


def sun_moon_lights(t01: float = 0.25, sun_radius: float = 2000.0):
    """DayNightEntity.Update at phase t01, Scenes/DayNightCycle.cs:41-89 (host-side per-frame animation)."""
    theta = f32(t01) * f32(2.0) * f32(math.pi) - f32(math.pi) * f32(0.5)
    sx, sy, sz = f32(math.cos(float(theta))), f32(math.sin(float(theta))), f32(0.25)
    norm = f32(math.sqrt(float(sx * sx + sy * sy + sz * sz)))
    sx, sy, sz = sx / norm, sy / norm, sz / norm
    sun_pos = vec3(sx * f32(sun_radius), max(50.0, float(sy * f32(sun_radius))), sz * f32(sun_radius))
    moon_pos = vec3(-sun_pos[0], max(50.0, -sun_pos[1]), -sun_pos[2])
    sun_n, moon_n = max(f32(0.0), sy), max(f32(0.0), -sy)
    sun_i = sun_n * sun_n
    moon_i = f32(math.sqrt(float(moon_n))) * f32(0.10)
    sky_blend = min(max(float(sun_i * f32(1.5)), 0.0), 1.0)
    lerp = lambda a, b, t: tuple(float(f32(x) * f32(1.0 - t) + f32(y) * f32(t)) for x, y in zip(a, b))
    top = lerp(vec3(0.02, 0.03, 0.06), vec3(0.30, 0.55, 0.95), sky_blend)
    bottom = lerp(vec3(0.00, 0.00, 0.00), vec3(0.80, 0.90, 1.00), sky_blend)
    lights = [PointLight(sun_pos, vec3(1.00, 0.96, 0.88), float(f32(300000.0) * sun_i)),
              PointLight(moon_pos, vec3(0.65, 0.70, 0.90), float(f32(8000.0) * moon_i))]
    return lights, top, bottom


def BuildMinecraftLike(nx=544, ny=256, nz=544, chunk=32, seed=0, t01=0.25):
    """VolumeScenes.BuildMinecraftLike steady state (Scenes/VolumeScenes.cs:569-627):
    every non-empty chunk^3 block of the world is one VolumeGrid under the scene BVH; Ambient 0;
    sun+moon lights.  Returns (scene, camera pose on the surface at x=z=0)."""
    world = make_voxel_world(nx, ny, nz, seed=seed)
    s = Scene()
    s.IsVolumeScene = True
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.0)
    wmin = (-nx // 2, 0, -nz // 2)
    for cx in range(nx // chunk):                   # WorldManager attaches non-air chunks (WorldManager.cs:696-731)
        for cy in range(ny // chunk):
            for cz in range(nz // chunk):
                cells = world[cx * chunk:(cx + 1) * chunk, cy * chunk:(cy + 1) * chunk, cz * chunk:(cz + 1) * chunk]
                if not (cells[..., 0] != 0).any():
                    continue
                min_corner = vec3(f32(wmin[0]) + f32(cx) * f32(chunk) * f32(1), f32(wmin[1]) + f32(cy) * f32(chunk) * f32(1),
                                  f32(wmin[2]) + f32(cz) * f32(chunk) * f32(1))
                s.Objects.append(VolumeGrid(np.ascontiguousarray(cells), min_corner, vec3(1, 1, 1), VoxelMaterialLookup))
    lights, top, bottom = sun_moon_lights(t01)
    s.Lights.extend(lights)
    s.BackgroundTop, s.BackgroundBottom = top, bottom
    col = world[nx // 2, :, nz // 2, 0]
    surface = int(np.nonzero(col)[0].max()) + 1 if col.any() else 64
    pose = dict(pos=(0.0, float(surface) + 1.8, 0.0), yaw=0.0, pitch=-0.2, fov=45.0)
    return s, pose


# ------------------------------------------------------------------- registry

def config_scene(n: int, small: bool = False, t01: float = 0.25):
    """(scene, fbW, fbH, ss, pose) of BASELINE.json configs[n-1].
    small=True shrinks the asset-sized inputs (mesh / world) for CPU-only tests.
    t01 = day phase of config 5's sun and moon (DayNightCycle.cs:48-82).  SURVEY §8(d)'s 0.25 puts the sun ON the horizon: both
    lights have intensity 0 there; 0.5 is noon (sun 300000 * sy^2, moon 0), 0.8 is night (moon 8000 * sqrt(-sy) * 0.1, sun 0)."""
    default_pose = dict(pos=(0.0, 1.0, 0.0), yaw=0.0, pitch=0.0, fov=45.0)
    if n == 1:
        return BuildCornellBox(), 80, 45, 1, default_pose
    if n == 2:
        return BuildMirrorSpheresOnChecker(), 640, 180, 1, default_pose
    if n == 3:
        return BuildBunnyScene(), 1280, 360, 1, dict(MESH_BENCH_POSE)
    if n == 4:
        sc = BuildDragonStandInScene(132, 33) if small else BuildDragonStandInScene()
        return sc, 1920, 540, 1, dict(MESH_BENCH_POSE)
    if n == 5:
        if small:
            sc, pose = BuildMinecraftLike(96, 128, 96, 32, t01=t01)
        else:
            sc, pose = BuildMinecraftLike(t01=t01)
        return sc, 1920, 540, 2, pose
    raise ValueError(n)
