"""Host-side mirror of the reference's voxel world file and chunk attachment.

Reference: ConsoleGame/RayTracing/Scenes/WorldGeneration/WorldManager.cs
  * `VG01` file: writer :612-629, reader ReloadFromExistingFile :399-441 — 4 header bytes 'V','G','0','1'
    (BinaryWriter.Write(char) = one UTF-8 byte each), int32 nx, ny, nz (little endian), then for x, y, z
    (z fastest) the pair int32 matId, int32 metaId.  That is exactly the cell order `ycge_grid.cells` takes.
  * BuildDesiredSet :372-397 — the chunk keys within ViewDistanceChunks of the camera column, every cy.
  * AttachChunkFromPreloaded :696-731 — slice a chunk out of the preloaded world, skip it when it is all air,
    place it at WorldMin + c * ChunkSize * VoxelSize.

Setup code on the data side of the hot path: it produces the VolumeGrid objects `Scene.Objects` holds; all
arithmetic that reaches the tracer is binary32, operation for operation as in the C#.
"""
from __future__ import annotations

import math
import struct
from typing import Callable, Iterable, List, Optional, Tuple

import numpy as np

from .scene import Scene, VolumeGrid, vec3

f32 = np.float32
MAGIC = b"VG01"


def write_vg01(path, cells: np.ndarray) -> None:
    """cells: int32 [nx, ny, nz, 2] = (matId, metaId).  WorldManager.cs:612-629."""
    c = np.ascontiguousarray(cells, dtype="<i4")
    if c.ndim != 4 or c.shape[3] != 2:
        raise ValueError("cells must be [nx, ny, nz, 2]")
    nx, ny, nz = c.shape[:3]
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<iii", nx, ny, nz))
        fh.write(c.tobytes())


def read_vg01(path) -> np.ndarray:
    """Returns int32 [nx, ny, nz, 2].  Error behaviour of ReloadFromExistingFile (:399-441): missing file ->
    FileNotFoundError, bad header / dimensions -> ValueError (InvalidDataException), short file -> EOFError."""
    if path is None or str(path).strip() == "":
        raise ValueError("filename is null or empty.")
    with open(path, "rb") as fh:
        head = fh.read(4)
        if head != MAGIC:
            raise ValueError("Unsupported world file header. Expected 'VG01'.")
        dims = fh.read(12)
        if len(dims) < 12:
            raise EOFError("Unable to read beyond the end of the stream.")
        nx, ny, nz = struct.unpack("<iii", dims)
        if nx <= 0 or ny <= 0 or nz <= 0:
            raise ValueError("Invalid world dimensions.")
        want = nx * ny * nz * 8
        data = fh.read(want)
        if len(data) < want:
            raise EOFError("Unable to read beyond the end of the stream.")
    return np.frombuffer(data, dtype="<i4").reshape(nx, ny, nz, 2).copy()


def build_desired_set(center, world_min, voxel_size, chunk_size: int, view_distance_chunks: int, chunks_y: int) -> List[Tuple[int, int, int]]:
    """BuildDesiredSet, WorldManager.cs:372-397.  Returned in insertion order (cx, then cz, then cy), which is
    the order EnsureViewLoaded (:634-656) attaches chunks in - and so the order of Scene.Objects."""
    scale_x = f32(f32(voxel_size[0]) * f32(chunk_size))
    scale_z = f32(f32(voxel_size[2]) * f32(chunk_size))
    cx_center = int(math.floor(float(f32((f32(center[0]) - f32(world_min[0])) / scale_x))))
    cz_center = int(math.floor(float(f32((f32(center[2]) - f32(world_min[2])) / scale_z))))
    out = []
    for cx in range(cx_center - view_distance_chunks, cx_center + view_distance_chunks + 1):
        for cz in range(cz_center - view_distance_chunks, cz_center + view_distance_chunks + 1):
            for cy in range(chunks_y):
                out.append((cx, cy, cz))
    return out


def slice_chunk(world: np.ndarray, cx: int, cy: int, cz: int, chunk_size: int) -> Optional[np.ndarray]:
    """The cells of chunk (cx, cy, cz), clipped at the world's far faces; None when outside or all air (:698-718)."""
    nx, ny, nz = world.shape[:3]
    if cx < 0 or cy < 0 or cz < 0:
        return None             # the C# would index out of range; the reference never asks (keys are clamped by its callers)
    sx, sy, sz = min(chunk_size, nx - cx * chunk_size), min(chunk_size, ny - cy * chunk_size), min(chunk_size, nz - cz * chunk_size)
    if sx <= 0 or sy <= 0 or sz <= 0:
        return None
    cells = world[cx * chunk_size:cx * chunk_size + sx, cy * chunk_size:cy * chunk_size + sy, cz * chunk_size:cz * chunk_size + sz]
    if not (cells[..., 0] != 0).any():
        return None
    return np.ascontiguousarray(cells)


def chunk_min_corner(world_min, voxel_size, chunk_size: int, cx: int, cy: int, cz: int):
    """:720-724: WorldMin + c * ChunkSize * VoxelSize, evaluated left to right in binary32 (int * int first)."""
    return vec3(f32(world_min[0]) + f32(cx * chunk_size) * f32(voxel_size[0]),
                f32(world_min[1]) + f32(cy * chunk_size) * f32(voxel_size[1]),
                f32(world_min[2]) + f32(cz * chunk_size) * f32(voxel_size[2]))


def attach_view(scene: Scene, world: np.ndarray, center, world_min, voxel_size, chunk_size: int, view_distance_chunks: int,
                material_lookup: Callable, chunks_y: Optional[int] = None, loaded: Optional[dict] = None) -> List[Tuple[int, int, int]]:
    """EnsureViewLoaded over a preloaded world (:634-656): attaches every desired, not yet loaded, non-air chunk to
    scene.Objects in desired-set order.  `loaded` (key -> VolumeGrid) persists between calls like loadedChunkMap."""
    if chunks_y is None:
        chunks_y = (world.shape[1] + chunk_size - 1) // chunk_size
    loaded = {} if loaded is None else loaded
    added = []
    for key in build_desired_set(center, world_min, voxel_size, chunk_size, view_distance_chunks, chunks_y):
        if key in loaded:
            continue
        cells = slice_chunk(world, *key, chunk_size)
        if cells is None:
            continue
        vg = VolumeGrid(cells, chunk_min_corner(world_min, voxel_size, chunk_size, *key), vec3(*voxel_size), material_lookup)
        loaded[key] = vg
        scene.Objects.append(vg)
        added.append(key)
    return added
