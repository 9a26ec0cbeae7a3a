"""The recipe that will pin the oracle (tools/ReferenceDump, tools/scene_file.py, tests/reference_dump.py), checked as far as it can be
without a .NET runtime: the scene file is lossless, the oracle renders a scene read back from it exactly as it renders the scene itself,
the comparison harness accepts a dump and rejects a perturbed one, and the C# dump tool writes the file names the harness reads."""
from __future__ import annotations

import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import reference_dump as rd  # noqa: E402
import scene_file  # noqa: E402
from yetanotherconsolegameengine_amd import abi, scenes  # noqa: E402
from yetanotherconsolegameengine_amd.scene import (AmbientLight, Box, Material, PointLight, Scene, Sphere, Texture, VolumeGrid, XZRect, flatten, vec3)  # noqa: E402


def _records_equal(a, b, n):
    return all(bytes(a[i]) == bytes(b[i]) for i in range(n))


def _mixed_scene():
    """every kind of record the file holds: analytic objects, a checker, a textured material, a mesh with its triangles, a voxel grid with its lookup"""
    rng = np.random.default_rng(3)
    tex = Texture(rng.integers(0, 256, (5, 7, 4), dtype=np.uint8))
    s = Scene()
    s.Ambient = AmbientLight(vec3(1, 1, 1), 0.2)
    s.Add(XZRect(-4.0, 4.0, -8.0, 1.0, 0.0, scene_checker(), 0.02, 0.0))
    s.Add(Sphere(vec3(-1.0, 0.8, -3.0), 0.8, Material(vec3(0.9, 0.2, 0.2), DiffuseTexture=tex, TextureWeight=0.6, UVScale=2.0)))
    s.Add(Box(vec3(0.5, 0.0, -4.0), vec3(1.5, 1.0, -3.0), Material(vec3(0.2, 0.9, 0.3)), 0.1, 0.0))
    pos, faces = scenes.make_torus_knot(24, 8)
    from yetanotherconsolegameengine_amd.scene import Mesh
    s.Add(Mesh((pos[faces] * np.float32(0.3) + np.float32([0.0, 1.5, -4.0])).astype(np.float32), Material(vec3(0.3, 0.3, 0.9))))
    cells = np.zeros((6, 5, 4, 2), dtype=np.int32)
    cells[1:5, 0:2, 1:3, 0] = 2; cells[2, 2, 2] = (3, 1)
    palette = {(2, 0): Material(vec3(0.4, 0.3, 0.2)), (3, 1): Material(vec3(0.1, 0.6, 0.1))}
    s.Add(VolumeGrid(cells, vec3(-3.0, 0.0, -6.0), vec3(0.5, 0.5, 0.5), lambda a, b: palette[(a, b)]))
    s.Lights.append(PointLight(vec3(0.0, 5.0, -1.0), vec3(1.0, 0.9, 0.8), 60.0))
    return s


def scene_checker():
    from yetanotherconsolegameengine_amd.scene import Checker
    return Checker(vec3(0.9, 0.9, 0.9), vec3(0.1, 0.1, 0.1), 0.75)


def test_scene_file_is_lossless(tmp_path):
    flat = flatten(_mixed_scene())
    pose = dict(pos=(0.25, 1.5, 2.0), yaw=0.1, pitch=-0.2, fov=50.0)
    f = tmp_path / "mixed.ysc"
    scene_file.write_ysc(f, flat, 64, 18, 2, pose)
    back = scene_file.LoadedScene(f)
    a, b = flat.struct, back.struct
    assert (back.fb_width, back.fb_height, back.super_sample) == (64, 18, 2)
    assert back.pose["fov"] == 50.0 and np.float32(back.pose["yaw"]) == np.float32(0.1) and tuple(np.float32(back.pose["pos"])) == tuple(np.float32(pose["pos"]))
    for k in ("n_materials", "n_prims", "n_meshes", "n_grids", "n_lights", "n_textures", "is_volume_scene", "has_dynamic_textures", "ambient_intensity"):
        assert getattr(a, k) == getattr(b, k), k
    for k in ("ambient_color", "background_top", "background_bottom"):
        assert bytes(getattr(a, k)) == bytes(getattr(b, k))
    assert a.n_meshes == 1 and a.n_grids == 1 and a.n_textures == 1 and a.n_prims == 5
    assert _records_equal(flat.materials, back.materials, a.n_materials)
    assert _records_equal(flat.prims, back.prims, a.n_prims) and _records_equal(flat.lights, back.lights, a.n_lights)
    m0, m1 = flat.meshes[0], back.meshes[0]
    assert m0.n_triangles == m1.n_triangles and m0.material == m1.material
    assert np.array_equal(np.ctypeslib.as_array(m0.triangles, shape=(9 * m0.n_triangles,)).view(np.uint32), np.ctypeslib.as_array(m1.triangles, shape=(9 * m1.n_triangles,)).view(np.uint32))
    g0, g1 = flat.grids[0], back.grids[0]
    assert (g0.nx, g0.ny, g0.nz, g0.n_lookup, g0.wireframe) == (g1.nx, g1.ny, g1.nz, g1.n_lookup, g1.wireframe) and bytes(g0.min_corner) == bytes(g1.min_corner)
    n = 2 * g0.nx * g0.ny * g0.nz
    assert np.array_equal(np.ctypeslib.as_array(g0.cells, shape=(n,)), np.ctypeslib.as_array(g1.cells, shape=(n,)))
    assert all(bytes(g0.lookup[i]) == bytes(g1.lookup[i]) for i in range(g0.n_lookup))
    t0, t1 = flat.textures[0], back.textures[0]
    assert (t0.width, t0.height) == (t1.width, t1.height) == (7, 5)
    assert np.array_equal(np.ctypeslib.as_array(t0.pixels, shape=(35,)), np.ctypeslib.as_array(t1.pixels, shape=(35,)))
    # the library's own argument checks accept what came back (pure host code: no GPU)
    L = abi.load_library()
    msg = C.create_string_buffer(256)
    assert L.ycge_validate_scene(back.byref(), msg, 256) == abi.YCGE_OK, msg.value
    # a truncated file is refused, not half-read
    f.write_bytes(f.read_bytes()[:-3])
    with pytest.raises(Exception):
        scene_file.LoadedScene(f)


def test_the_oracle_renders_a_scene_file_like_the_scene(tmp_path, oracle):
    """config 1 through the file and directly: the same frames, bit for bit (what ReferenceDump's input means is what the tests mean)"""
    sc, w, h, ss, pose = scenes.config_scene(1)
    f = tmp_path / "c1.ysc"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "scene_file.py"), "1", str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    back = scene_file.LoadedScene(f)
    assert (back.fb_width, back.fb_height, back.super_sample) == (w, h, ss)
    a = oracle.OracleRenderer(sc, w, h, ss, pose)
    b = oracle.OracleRenderer(None, back.fb_width, back.fb_height, back.super_sample, back.pose, flat=back)
    for _ in range(2):
        a.render(stages=1, threads=4); b.render(stages=1, threads=4)
    for which in (abi.BUF_CURRENT_HDR, abi.BUF_G_NORMAL, abi.BUF_TAA_HISTORY):
        assert a.read(which).tobytes() == b.read(which).tobytes()
    a.close(); b.close()


def test_the_harness_accepts_a_dump_and_rejects_a_perturbed_one(tmp_path, oracle):
    """tests/reference_dump.py on a dump written from the oracle itself (the layout Program.cs writes): clean against the oracle; one flipped
    bit in a node, one nudged ray, a brightened history - each is reported"""
    sc, w, h, ss, pose = scenes.config_scene(3)
    w, h = 96, 27
    flat = flatten(sc)
    o = oracle.OracleRenderer(sc, w, h, ss, pose, flat=flat)
    d = tmp_path / "selftest"
    rd.write_dump(d, o, 2, lambda k: o.render(stages=2, threads=4, want_sdr=True), n_meshes=1)
    scene_file.write_ysc(d / "scene.ysc", flat, w, h, ss, pose)
    o.close()
    dump = rd.Dump(d)
    assert (dump.hiW, dump.hiH, dump.frames) == (96, 54, 2)
    back = scene_file.LoadedScene(d / "scene.ysc")
    o2 = oracle.OracleRenderer(None, back.fb_width, back.fb_height, back.super_sample, back.pose, flat=back)
    assert rd.compare_accel(dump, o2, 1) == []
    for k in (1, 2):
        sdr = o2.render(stages=2, threads=4, want_sdr=True)
        rep, bad = rd.compare_frame(dump, k, o2, sdr)
        assert bad == [] and rep["rays_bit_exact"] == 1.0 and rep["sdr_bit_exact"] == 1.0 and rep["taa_history.f32_rms"] == 0.0, (rep, bad)
    # perturbations (frame 2 is the renderer's current frame)
    nodes = np.fromfile(d / "accel_mesh0_nodes.bin", dtype=rd.NODE_DTYPE); nodes["left"][1] ^= 1; nodes.tofile(d / "accel_mesh0_nodes.bin")
    assert any("mesh0: nodes differ" in m for m in rd.compare_accel(rd.Dump(d), o2, 1))
    hist = np.fromfile(d / "f2_taa_history.f32", dtype="<f4"); (hist * np.float32(1.01) + np.float32(0.001)).astype("<f4").tofile(d / "f2_taa_history.f32")
    nrm = np.fromfile(d / "f2_g_normal.f32", dtype="<f4"); nrm[100] = -nrm[100] if nrm[100] != 0 else 1.0; nrm.tofile(d / "f2_g_normal.f32")
    rep, bad = rd.compare_frame(rd.Dump(d), 2, o2, sdr)
    assert any("taa_history" in m for m in bad) and any("g_normal" in m for m in bad)
    rays = np.fromfile(d / "f2_rays.f32", dtype="<f4"); rays[3::6] += np.float32(1e-3); rays.tofile(d / "f2_rays.f32")
    rep, bad = rd.compare_frame(rd.Dump(d), 2, o2, sdr)
    assert any("rays differ" in m for m in bad) and rep["rays_bit_exact"] < 0.01
    o2.close()


def test_the_csharp_dump_tool_writes_what_the_harness_reads():
    prog = (ROOT / "tools" / "ReferenceDump" / "Program.cs").read_text()
    for name in list(rd.FRAME_FILES) + [rd.SDR_FILE]:
        assert '"' + name + '"' in prog, name
    for name in ('"accel_scene"', '"accel_mesh"', '"_nodes.bin"', '"_leaf.i32"', '"meta.json"'):
        assert name in prog, name
    for key in ("fb_width", "fb_height", "super_sample", "hi_w", "hi_h", "frames", "n_meshes", "runtime"):
        assert '\\"' + key + '\\"' in prog, key
    # the private members it reads exist under those names in the survey's citations (RaytraceRenderer.cs:63-88, BVH.cs:11-25, MeshBVH.cs:18-39)
    for member in ("rays", "currentHdr", "gAlbedo", "gNormal", "gDepth", "skyMask", "taaHistory", "nodeMinX", "nodeCountUsed", "leafObjIndex", "leafTriIndex"):
        assert '"' + member + '"' in prog, member
    sf = (ROOT / "tools" / "ReferenceDump" / "SceneFile.cs").read_text()
    assert "0x31435359u" in sf and int.from_bytes(scene_file.MAGIC, "little") == 0x31435359
    for t in re.findall(r"r\.Take<(Y\w+)>", sf):
        assert t in ("YScene", "YMaterial", "YPrim", "YLight", "YMesh", "YGrid", "YVoxelLookup", "YTexture"), t
    csproj = (ROOT / "tools" / "ReferenceDump" / "ReferenceDump.csproj").read_text()
    assert "bindings/csharp/Ycge.cs" in csproj and "ConsoleGame.csproj" in csproj and "net8.0" in csproj


def test_committed_scene_files_and_the_one_command_recipe(tmp_path, oracle):
    """the scene files that are small live under tests/golden/reference/<name>/scene.ysc (gzip; sha256 in the
    README there), tools/ReferenceDump/run_all.sh renders them with the reference and ends in tools/compare_dump.py - numpy + g++ only -
    which holds a dump to the ORACLE and says PINNED or not.  Here, without .NET: every committed file is the one the README names and
    loads; a dump made by the oracle compares clean through the CLI (exit 0), a perturbed one does not (exit 1)."""
    import hashlib
    import shutil
    import subprocess
    ref = ROOT / "tests" / "golden" / "reference"
    readme = (ref / "README.md").read_text()
    names = sorted(p.parent.name for p in ref.glob("*/scene.ysc"))
    assert names == ["config1", "config2", "config3_320x90", "config4_reduced_320x90", "config5_reduced_96x27", "config5_reduced_noon_96x27"]
    for n in names:
        f = ref / n / "scene.ysc"
        assert f.stat().st_size < 2_000_000, n
        assert hashlib.sha256(f.read_bytes()).hexdigest() in readme, f"{n}: sha256 not in the README"
        back = scene_file.LoadedScene(f)
        assert back.fb_width > 0 and back.struct.n_prims > 0, n
    sh = (ROOT / "tools" / "ReferenceDump" / "run_all.sh").read_text()
    assert "tests/golden/reference" in sh and "compare_dump.py" in sh and "dotnet run" in sh and "TryFlipAndBlit" in sh
    d = tmp_path / "config1"
    d.mkdir()
    shutil.copy(ref / "config1" / "scene.ysc", d / "scene.ysc")
    back = scene_file.LoadedScene(d / "scene.ysc")
    o = oracle.OracleRenderer(None, back.fb_width, back.fb_height, back.super_sample, back.pose, flat=back)
    rd.write_dump(d, o, 2, lambda k: o.render(stages=2, threads=4, want_sdr=True), n_meshes=0)
    o.close()
    cli = [sys.executable, str(ROOT / "tools" / "compare_dump.py"), str(d)]
    r = subprocess.run(cli, capture_output=True, text=True)
    assert r.returncode == 0 and "PINNED" in r.stdout and "bit-exact" in r.stdout, r.stdout + r.stderr
    hdr = np.fromfile(d / "f2_current_hdr.f32", dtype="<f4"); (hdr + np.float32(0.01)).astype("<f4").tofile(d / "f2_current_hdr.f32")
    r = subprocess.run(cli, capture_output=True, text=True)
    assert r.returncode == 1 and "NOT pinned" in r.stdout and "current_hdr" in r.stdout, r.stdout + r.stderr
