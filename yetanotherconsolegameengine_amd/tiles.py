"""Framebuffer tiling across GPUs — the host-side statement of the layout the kernels use.

The trace grid is cut into 32x8-pixel tiles, numbered row-major; rank r of `world` traces the tiles with
tile_id % world == r (round-robin, so the expensive tiles of a mesh in the image centre spread over all
GPUs).  Each rank packs its tiles into a slab — for its k-th owned tile (ascending tile_id), 256 pixel
records in row-major order inside the tile (j = y*32 + x), 11 float32 each {hdr rgb, albedo rgb, normal
xyz, depth, sky} or, with config.slab_albedo = 0 (no denoise stage to feed), 8 {hdr rgb, normal xyz, depth,
sky} — and
one all-gather of equal-sized slabs (RCCL over xGMI, torch.distributed backend "nccl") reassembles the
frame; k_unpermute scatters the gathered slabs back to row-major full-frame buffers on every rank.

(Inside the kernels a tile is one 256-thread workgroup: thread t -> wavefront w = t // 64, lane l = t % 64
traces pixel x = 8*w + l % 8, y = l // 8, i.e. one 8x8 pixel block per wavefront.)

This module is pure layout arithmetic (numpy); it is what the multi-process CPU tests exercise and what
the GPU tests compare k_pack_slab / k_unpermute against.
"""
from __future__ import annotations

import numpy as np

TILE_W, TILE_H, SLAB_FLOATS, LEAN_SLAB_FLOATS = 32, 8, 11, 8
LEAN_COLUMNS = [0, 1, 2, 6, 7, 8, 9, 10]        # the 11-float record's columns a lean record keeps, in order


def tile_grid(hiW: int, hiH: int):
    tx, ty = (hiW + TILE_W - 1) // TILE_W, (hiH + TILE_H - 1) // TILE_H
    return tx, ty, tx * ty


def owned_tiles(rank: int, world: int, n_tiles: int) -> np.ndarray:
    return np.arange(rank, n_tiles, world, dtype=np.int64)


def tiles_per_rank_padded(world: int, n_tiles: int) -> int:
    return (n_tiles + world - 1) // world


def slab_floats(world: int, n_tiles: int, floats: int = SLAB_FLOATS) -> int:
    return tiles_per_rank_padded(world, n_tiles) * 256 * floats


def _local_xy():
    j = np.arange(256)
    return j % TILE_W, j // TILE_W


def pack_slab(frame: np.ndarray, rank: int, world: int) -> np.ndarray:
    """frame: [hiH, hiW, F] float32 (F = 11, or 8 for lean slabs) -> this rank's slab [padded_tiles*256*F] (k_pack_slab)."""
    hiH, hiW, floats = frame.shape
    tx, ty, n = tile_grid(hiW, hiH)
    out = np.zeros((tiles_per_rank_padded(world, n), 256, floats), dtype=np.float32)
    lx, ly = _local_xy()
    for k, tid in enumerate(owned_tiles(rank, world, n)):
        px, py = (tid % tx) * TILE_W + lx, (tid // tx) * TILE_H + ly
        ok = (px < hiW) & (py < hiH)
        out[k, ok] = frame[py[ok], px[ok]]
    return out.reshape(-1)


def unpermute(all_slabs: np.ndarray, hiW: int, hiH: int, world: int, floats: int = SLAB_FLOATS) -> np.ndarray:
    """all_slabs: [world * slab_floats] rank-major (as all_gather leaves it) -> frame [hiH, hiW, floats] (k_unpermute)."""
    tx, ty, n = tile_grid(hiW, hiH)
    per = tiles_per_rank_padded(world, n)
    s = all_slabs.reshape(world, per, 256, floats)
    frame = np.zeros((hiH, hiW, floats), dtype=np.float32)
    lx, ly = _local_xy()
    for tid in range(n):
        r, k = tid % world, tid // world
        px, py = (tid % tx) * TILE_W + lx, (tid // tx) * TILE_H + ly
        ok = (px < hiW) & (py < hiH)
        frame[py[ok], px[ok]] = s[r, k, ok]
    return frame


# ---------------------------------------------------------------------------------------------- tile-resident form: halo lists
# TAA reads a 3x3 window (TemporalBlendWithClamp, clampRadius = 1, RaytraceRenderer.cs:218), so a rank that keeps the history of its own
# tiles needs {hdr, sky} of the one-pixel ring around each of them from the ranks that own those pixels.  Sender and receiver enumerate
# the ring pixels of the RECEIVER's tiles in one fixed order - tiles ascending; per tile the row above (x0 - 1 .. x0 + 32), the row below,
# the column left (y0 .. y0 + 7), the column right; pixels outside the image do not exist - and keep those the sender owns.  This is the
# statement of csrc/ycge_host.cpp: halo_layout (the CPU tests hold the two to each other, element for element).

def halo_ring(tile: int, hiW: int, hiH: int) -> np.ndarray:
    """pixel indices (x + y * hiW) of the one-pixel ring around `tile`, in the exchange's order"""
    tx, _, _ = tile_grid(hiW, hiH)
    x0, y0 = (tile % tx) * TILE_W, (tile // tx) * TILE_H
    xs = np.arange(x0 - 1, x0 + TILE_W + 1)
    ys = np.arange(y0, y0 + TILE_H)
    pts = ([(x, y0 - 1) for x in xs] + [(x, y0 + TILE_H) for x in xs] + [(x0 - 1, y) for y in ys] + [(x0 + TILE_W, y) for y in ys])
    return np.array([x + y * hiW for x, y in pts if 0 <= x < hiW and 0 <= y < hiH], dtype=np.int64)


def pixel_owner(px: np.ndarray, hiW: int, hiH: int, world: int) -> np.ndarray:
    tx, _, _ = tile_grid(hiW, hiH)
    x, y = px % hiW, px // hiW
    return ((y // TILE_H) * tx + x // TILE_W) % world


def halo_lists(rank: int, world: int, hiW: int, hiH: int):
    """(send_px per destination rank, recv_px per source rank): lists of pixel indices; send[r][k] on rank q IS recv[q][k] on rank r"""
    _, _, n = tile_grid(hiW, hiH)
    recv = [[] for _ in range(world)]
    for t in owned_tiles(rank, world, n):
        ring = halo_ring(int(t), hiW, hiH)
        own = pixel_owner(ring, hiW, hiH, world)
        for q in range(world):
            if q != rank:
                recv[q].extend(ring[own == q].tolist())
    send = [[] for _ in range(world)]
    for r in range(world):
        if r == rank:
            continue
        for t in owned_tiles(r, world, n):
            ring = halo_ring(int(t), hiW, hiH)
            send[r].extend(ring[pixel_owner(ring, hiW, hiH, world) == rank].tolist())
    return [np.array(v, dtype=np.int64) for v in send], [np.array(v, dtype=np.int64) for v in recv]


def history_slab_floats(world: int, n_tiles: int) -> int:
    return tiles_per_rank_padded(world, n_tiles) * 256 * 3
