"""Reading a dump of tools/ReferenceDump (the REFERENCE's own buffers) and holding a renderer - oracle or HIP path - to it.

Test infrastructure.  The file names are the ones tools/ReferenceDump/Program.cs writes (tests/test_reference_dump_tools.py keeps the two
lists in step); `write_dump` produces the same files from any renderer of this repository, so that the harness itself is tested without
a .NET runtime (a dump made from the oracle must compare clean, a perturbed one must not).
"""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np

from yetanotherconsolegameengine_amd import abi

ROOT = Path(__file__).resolve().parents[1]
NODE_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("start", "<i4"), ("count", "<i4")])

# name in the dump -> (ycge_buffer, dtype, elements per pixel)
FRAME_FILES = {"rays.f32": (abi.BUF_RAYS, "<f4", 6), "current_hdr.f32": (abi.BUF_CURRENT_HDR, "<f4", 3), "g_albedo.f32": (abi.BUF_G_ALBEDO, "<f4", 3),
               "g_normal.f32": (abi.BUF_G_NORMAL, "<f4", 3), "g_depth.f32": (abi.BUF_G_DEPTH, "<f4", 1), "sky.u8": (abi.BUF_SKY_MASK, "u1", 1),
               "taa_history.f32": (abi.BUF_TAA_HISTORY, "<f4", 3)}
SDR_FILE = "sdr.f32"          # 6 per chexel
RMS_TOL = 1e-4                # north_star: radiance within 1e-4 RMS of the reference
RAY_RMS_TOL = 1e-6            # rays when the two hosts' sinf / cosf / tanf differ in the last place


def dump_dirs():
    """every directory that holds a dump: tests/golden/reference/*/ (committed, small) and $YCGE_REFERENCE_GOLDENS/*/ (large ones)"""
    roots = [ROOT / "tests" / "golden" / "reference"]
    if os.environ.get("YCGE_REFERENCE_GOLDENS"):
        roots.append(Path(os.environ["YCGE_REFERENCE_GOLDENS"]))
    out = []
    for r in roots:
        if r.is_dir():
            out += sorted(d for d in r.iterdir() if (d / "meta.json").exists() and (d / "scene.ysc").exists())
    return out


class Dump:
    def __init__(self, path):
        self.path = Path(path)
        self.meta = json.loads((self.path / "meta.json").read_text())
        self.hiW, self.hiH, self.frames = self.meta["hi_w"], self.meta["hi_h"], self.meta["frames"]
        self.fbW, self.fbH = self.meta["fb_width"], self.meta["fb_height"]

    def frame(self, k, name):
        _, dt, n = FRAME_FILES[name]
        a = np.fromfile(self.path / f"f{k}_{name}", dtype=dt)
        return a.reshape((self.hiH, self.hiW, n) if n > 1 else (self.hiH, self.hiW))

    def sdr(self, k):
        return np.fromfile(self.path / f"f{k}_{SDR_FILE}", dtype="<f4").reshape(self.fbH, self.fbW, 2, 3)

    def accel(self, name):
        """('scene' | 'mesh<i>') -> (nodes, leaf index)"""
        return (np.fromfile(self.path / f"accel_{name}_nodes.bin", dtype=NODE_DTYPE), np.fromfile(self.path / f"accel_{name}_leaf.i32", dtype="<i4"))


def write_dump(path, renderer, frames, render_frame, scene_file=None, n_meshes=0, runtime="this repository's oracle (harness self-test)"):
    """The files Program.cs writes, from a renderer of this repository (oracle or HIP: .read / .accel / fbW ...).  render_frame(k) renders
    frame k and returns its SDR array (fbH, fbW, 2, 3)."""
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    nodes, leaf = renderer.accel(abi.ACCEL_SCENE_NODES), renderer.accel(abi.ACCEL_SCENE_LEAF_INDEX)
    nodes.tofile(path / "accel_scene_nodes.bin"); leaf.tofile(path / "accel_scene_leaf.i32")
    for i in range(n_meshes):
        renderer.accel(abi.ACCEL_MESH_NODES, i).tofile(path / f"accel_mesh{i}_nodes.bin")
        renderer.accel(abi.ACCEL_MESH_LEAF_INDEX, i).tofile(path / f"accel_mesh{i}_leaf.i32")
    for k in range(1, frames + 1):
        sdr = render_frame(k)
        for name, (which, dt, n) in FRAME_FILES.items():
            renderer.read(which).astype(dt).tofile(path / f"f{k}_{name}")
        np.ascontiguousarray(sdr, dtype="<f4").tofile(path / f"f{k}_{SDR_FILE}")
    (path / "meta.json").write_text(json.dumps({"fb_width": renderer.fbW, "fb_height": renderer.fbH, "super_sample": renderer.hiW // renderer.fbW, "hi_w": renderer.hiW,
                                                "hi_h": renderer.hiH, "frames": frames, "n_meshes": n_meshes, "scene_file": scene_file or "scene.ysc", "runtime": runtime}) + "\n")


def _rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


def _bits_differ(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        return int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))
    return int(np.count_nonzero(a != b))


def compare_accel(dump, renderer, n_meshes):
    """builders: bit-exact, no tolerance (pure fp32 arithmetic, comparisons and .NET's introsort).  Returns a list of complaints."""
    bad = []
    for name, wn, wl, idx in [("scene", abi.ACCEL_SCENE_NODES, abi.ACCEL_SCENE_LEAF_INDEX, 0)] + [(f"mesh{i}", abi.ACCEL_MESH_NODES, abi.ACCEL_MESH_LEAF_INDEX, i) for i in range(n_meshes)]:
        want_nodes, want_leaf = dump.accel(name)
        got_nodes, got_leaf = renderer.accel(wn, idx), renderer.accel(wl, idx)
        if want_nodes.shape != got_nodes.shape or want_nodes.tobytes() != got_nodes.tobytes():
            bad.append(f"{name}: nodes differ ({len(want_nodes)} in the dump, {len(got_nodes)} built here)")
        if want_leaf.shape != got_leaf.shape or not np.array_equal(want_leaf, got_leaf):
            bad.append(f"{name}: leaf index differs")
    return bad


def compare_frame(dump, k, renderer, sdr=None):
    """One frame of `renderer` (already rendered) against frame k of the dump.  Returns (report, complaints): geometry buffers bit-exact
    wherever the primary ray is bit-exact; radiance, history and SDR within 1e-4 RMS."""
    rep, bad = {}, []
    want_rays, got_rays = dump.frame(k, "rays.f32"), renderer.read(abi.BUF_RAYS)
    ray_same = (want_rays.view(np.uint32) == got_rays.view(np.uint32)).all(axis=2)
    rep["rays_bit_exact"] = float(ray_same.mean())
    rep["rays_rms"] = _rms(want_rays, got_rays)
    if rep["rays_rms"] > RAY_RMS_TOL:
        bad.append(f"frame {k}: rays differ by {rep['rays_rms']:.3g} RMS")
    for name in ("g_albedo.f32", "g_normal.f32", "g_depth.f32", "sky.u8"):
        which = FRAME_FILES[name][0]
        want, got = dump.frame(k, name), renderer.read(which)
        same_px = ray_same if want.ndim == 2 else ray_same[..., None]
        n = _bits_differ(np.where(same_px, want, 0), np.where(same_px, got, 0))
        rep[name + "_mismatch_where_rays_agree"] = n
        if n:
            bad.append(f"frame {k}: {name} differs on {n} values of pixels whose primary ray is bit-identical")
    for name in ("current_hdr.f32", "taa_history.f32"):
        want, got = dump.frame(k, name), renderer.read(FRAME_FILES[name][0])
        rep[name + "_rms"] = _rms(want, got); rep[name + "_bit_exact"] = float((want.view(np.uint32) == got.view(np.uint32)).mean())
        if rep[name + "_rms"] > RMS_TOL:
            bad.append(f"frame {k}: {name} RMS {rep[name + '_rms']:.3g} > {RMS_TOL}")
    if sdr is not None:
        want = dump.sdr(k)
        rep["sdr_rms"] = _rms(want, sdr); rep["sdr_bit_exact"] = float((want.view(np.uint32) == np.ascontiguousarray(sdr, dtype='<f4').view(np.uint32)).mean())
        if rep["sdr_rms"] > RMS_TOL:
            bad.append(f"frame {k}: SDR chexels RMS {rep['sdr_rms']:.3g} > {RMS_TOL}")
    return rep, bad
