"""Host-side mirror of MeshLoader.FromObj and MeshScenes.AddMeshAutoGround.

Reference: ConsoleGame/RayTracing/MeshLoader.cs:12-149 (OBJ subset: `v`, `f` with
`/` and negative indices, fan triangulation; bbox normalise; scale+translate) and
ConsoleGame/RayTracing/Scenes/MeshScenes.cs:173-330 (auto-ground placement from
the largest connected component's centroid-normalised bounds).

Setup code, not the hot path: it only produces the float32 triangle soup the
C-ABI takes.  All arithmetic is binary32, operation for operation as in the C#.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

f32 = np.float32


def parse_obj(text_lines) -> Tuple[np.ndarray, np.ndarray]:
    """MeshLoader.cs:23-55: returns (positions float32 [nv,3], faces int32 [nt,3])."""
    pos, faces = [], []
    for line in text_lines:
        line = line.rstrip("\r\n")
        if len(line) == 0 or line[0] == "#":
            continue
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v" and len(tok) >= 4:
            pos.append((float(tok[1]), float(tok[2]), float(tok[3])))
        elif tok[0] == "f" and len(tok) >= 4:
            idx = []
            for t in tok[1:]:
                s = t.split("/")[0]
                if s == "":
                    idx.append(0)                      # ParseIndex: empty -> 0
                    continue
                i = int(s)
                idx.append(i - 1 if i > 0 else len(pos) + i)
            for i in range(2, len(idx)):
                faces.append((idx[0], idx[i - 1], idx[i]))
    return np.asarray(pos, dtype=np.float32).reshape(-1, 3), np.asarray(faces, dtype=np.int32).reshape(-1, 3)


def load_obj(path) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "r") as fh:
        return parse_obj(fh)


def normalize_all_used_vertices(pos: np.ndarray, faces: np.ndarray, target_size: float) -> np.ndarray:
    """MeshLoader.cs:107-148: bbox of the vertices any face uses; every vertex is moved."""
    used = np.unique(faces.reshape(-1))
    p = pos[used]
    mn, mx = p.min(axis=0), p.max(axis=0)
    c = (mn + mx) * f32(0.5)
    r = mx - mn
    max_extent = r[0]
    if r[1] > max_extent:
        max_extent = r[1]
    if r[2] > max_extent:
        max_extent = r[2]
    if max_extent <= 0:
        max_extent = f32(1.0)
    s = f32(target_size) / max_extent
    return ((pos - c) * s).astype(np.float32)


def from_obj_arrays(pos: np.ndarray, faces: np.ndarray, scale: float = 1.0, translate=(0.0, 0.0, 0.0),
                    normalize: bool = True, target_size: float = 1.0) -> np.ndarray:
    """MeshLoader.FromObj after parsing (MeshLoader.cs:58-96): float32 triangles [nt,3,3]."""
    pos = np.asarray(pos, dtype=np.float32)
    if normalize:
        pos = normalize_all_used_vertices(pos, faces, target_size)
    t = np.asarray(translate, dtype=np.float32)
    if f32(scale) != f32(1.0) or t[0] != 0 or t[1] != 0 or t[2] != 0:
        pos = (pos * f32(scale) + t).astype(np.float32)
    return np.ascontiguousarray(pos[faces])           # [nt, 3 (A,B,C), 3 (xyz)]


def read_obj_bounds_normalized(pos: np.ndarray, faces: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """MeshScenes.TryReadObjBoundsNormalized, MeshScenes.cs:186-330."""
    nv, nf = pos.shape[0], faces.shape[0]
    if nv == 0 or nf == 0:
        return None
    parent = list(range(nv))
    rank = [0] * nv

    def find(x):
        while x != parent[x]:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    def union(x, y):
        rx, ry = find(x), find(y)
        if rx == ry:
            return
        if rank[rx] < rank[ry]:
            parent[rx] = ry
        elif rank[rx] > rank[ry]:
            parent[ry] = rx
        else:
            parent[ry] = rx
            rank[rx] += 1

    fl = faces.tolist()
    for a, b, c in fl:
        union(a, b)
        union(b, c)
    comp = {}
    for i, (a, b, c) in enumerate(fl):
        comp.setdefault(find(a), []).append(i)
    best_root, best_count = -1, -1
    for root, lst in comp.items():             # Dictionary enumerates in insertion order (no removals)
        if len(lst) > best_count:
            best_count, best_root = len(lst), root
    if best_root == -1:
        return None
    kept = faces[np.asarray(comp[best_root], dtype=np.int64)]
    A, B, Cc = pos[kept[:, 0]], pos[kept[:, 1]], pos[kept[:, 2]]
    third = f32(1.0) / f32(3.0)
    terms = ((A + B) + Cc) * third               # per-triangle (A+B+C)*(1/3f), float32
    # sequential float32 accumulation in face order (cx += ...), MeshScenes.cs:292-302
    c = np.zeros(3, dtype=np.float32)
    acc = np.cumsum(terms, axis=0, dtype=np.float32)
    c = acc[-1]
    inv_t = f32(1.0) / f32(kept.shape[0])
    c = c * inv_t
    used = np.unique(kept.reshape(-1))
    rel = pos[used] - c
    rmin, rmax = rel.min(axis=0), rel.max(axis=0)
    r = rmax - rmin
    max_extent = r[0]
    if r[1] > max_extent:
        max_extent = r[1]
    if r[2] > max_extent:
        max_extent = r[2]
    if max_extent <= 0:
        max_extent = f32(1.0)
    s = f32(1.0) / max_extent
    return (rmin * s).astype(np.float32), (rmax * s).astype(np.float32)


def add_mesh_auto_ground(pos: np.ndarray, faces: np.ndarray, scale: float, target_pos) -> np.ndarray:
    """MeshScenes.AddMeshAutoGround, MeshScenes.cs:173-184: returns the placed triangles."""
    b = read_obj_bounds_normalized(pos, faces)
    if b is None:
        raise FileNotFoundError("OBJ not found or empty")
    min_y_norm = b[0][1]
    y_translate = f32(target_pos[1]) - min_y_norm * f32(scale) + f32(0.01)
    translate = (f32(target_pos[0]), y_translate, f32(target_pos[2]))
    return from_obj_arrays(pos, faces, scale=scale, translate=translate, normalize=True, target_size=1.0)
