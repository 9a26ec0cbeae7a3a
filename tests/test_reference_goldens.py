"""Oracle and HIP path held to buffers the REFERENCE ITSELF produced (tools/ReferenceDump on a machine with the .NET 8 SDK).

Skipped while no dump is present - the reference is C#/.NET 8, this image has no dotnet, so none can be made here and parity stays
"unpinned" (DESIGN.md section 2).  A dump directory holds `scene.ysc` (tools/scene_file.py), `meta.json` and the files of
tools/ReferenceDump/Program.cs; small ones are committed under tests/golden/reference/<name>/, large ones are found through
$YCGE_REFERENCE_GOLDENS.  Bars (tools/ReferenceDump/README.md): builders bit-exact; rays bit-exact where the hosts' sinf/cosf/tanf agree
(RMS <= 1e-6 otherwise); G-buffer bit-exact wherever the primary ray is; radiance, TAA history and SDR chexels within 1e-4 RMS
(RaytraceRenderer.cs:157-267 end to end).
"""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import reference_dump as rd  # noqa: E402

DUMPS = rd.dump_dirs()
if not DUMPS:
    pytest.skip("no reference dump present (tools/ReferenceDump needs the .NET 8 SDK; see its README): parity stays unpinned", allow_module_level=True)

import scene_file  # noqa: E402
from yetanotherconsolegameengine_amd import abi  # noqa: E402


def _load(d):
    dump = rd.Dump(d)
    back = scene_file.LoadedScene(d / "scene.ysc")
    assert (back.fb_width, back.fb_height) == (dump.fbW, dump.fbH), "scene.ysc is not the file the dump was made from"
    return dump, back


@pytest.mark.parametrize("d", DUMPS, ids=[p.name for p in DUMPS])
def test_oracle_against_the_reference(d, oracle):
    dump, back = _load(d)
    o = oracle.OracleRenderer(None, back.fb_width, back.fb_height, back.super_sample, back.pose, flat=back)
    bad = rd.compare_accel(dump, o, dump.meta["n_meshes"])
    for k in range(1, dump.frames + 1):
        sdr = o.render(stages=2, threads=8, want_sdr=True)
        rep, b = rd.compare_frame(dump, k, o, sdr)
        print(d.name, "oracle frame", k, rep)
        bad += b
    o.close()
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("d", DUMPS, ids=[p.name for p in DUMPS])
def test_hip_path_against_the_reference(d, product_lib):
    from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
    dump, back = _load(d)
    g = RaytraceRenderer(back, back.fb_width, back.fb_height, back.pose["fov"], back.super_sample, capture_debug=True)
    g.SetCamera(back.pose["pos"], back.pose["yaw"], back.pose["pitch"])
    bad = rd.compare_accel(dump, g, dump.meta["n_meshes"])
    for k in range(1, dump.frames + 1):
        sdr = g.TryFlipAndBlit(want_sdr=True)
        rep, b = rd.compare_frame(dump, k, g, sdr)
        print(d.name, "HIP frame", k, rep)
        bad += b
    g.close()
    assert not bad, bad
