"""Regenerates the committed fixtures under tests/golden/ and the bunny asset of the package (yetanotherconsolegameengine_amd/assets/).

Run in the authoring container only (needs /root/reference for the bunny asset):
    python tests/golden/make_fixtures.py bunny      # parse the reference's OBJ asset -> npz (data, not source)
    python tests/golden/make_fixtures.py goldens    # oracle-rendered golden buffers for config 1 (needs oracle/liborc_oracle.so)
    python tests/golden/make_fixtures.py post       # same scene through denoise / exposure / tonemap (frames 1-3)

The bunny fixture is the reference's own data file (ConsoleGame/assets/stanford-bunny.obj, the
Stanford bunny as exported by MeshLab) stored as parsed float32 positions + int32 fan-triangulated
faces, exactly what MeshLoader.cs:23-55 produces from it.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
HERE = Path(__file__).resolve().parent


def bunny():
    from yetanotherconsolegameengine_amd import mesh_loader
    pos, faces = mesh_loader.load_obj("/root/reference/ConsoleGame/assets/stanford-bunny.obj")
    assert faces.shape[0] == 69451, faces.shape
    from yetanotherconsolegameengine_amd import scenes
    scenes.ASSETS_DIR.mkdir(exist_ok=True)
    np.savez_compressed(scenes.BUNNY_FIXTURE, positions=pos, faces=faces)
    print("bunny:", pos.shape, faces.shape)


def goldens():
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob
    from yetanotherconsolegameengine_amd import abi, scenes
    sc, fbw, fbh, ss, pose = scenes.config_scene(1)
    with ob.OracleRenderer(sc, fbw, fbh, ss, pose) as r:
        out = {}
        for frame in (1, 2, 3):
            r.render(stages=1)
            if frame == 1:
                for name, which in (("rays", abi.BUF_RAYS), ("prim_id", abi.BUF_PRIM_ID), ("sub_id", abi.BUF_SUB_ID),
                                    ("hit_t", abi.BUF_HIT_T), ("current_hdr", abi.BUF_CURRENT_HDR),
                                    ("rng_state", abi.BUF_RNG_STATE)):
                    out[f"f1_{name}"] = r.read(which)
            out[f"f{frame}_taa_history"] = r.read(abi.BUF_TAA_HISTORY)
        np.savez_compressed(HERE / "cornell_80x45_frames123.npz", **out)
        print("goldens:", {k: v.shape for k, v in out.items()})


def post():
    """frames 1-3 of config 1 through the WHOLE of TryFlipAndBlit (stages = 2): exposure of every frame, SDR of every frame,
    denoised HDR of frame 3"""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob
    from yetanotherconsolegameengine_amd import abi, scenes
    sc, fbw, fbh, ss, pose = scenes.config_scene(1)
    with ob.OracleRenderer(sc, fbw, fbh, ss, pose) as r:
        out = {}
        for frame in (1, 2, 3):
            out[f"f{frame}_sdr"] = r.render(stages=2, want_sdr=True)
            out[f"f{frame}_exposure"] = np.float32(r.stats.exposure)
        out["f3_denoised"] = r.read(abi.BUF_DENOISED)
        np.savez_compressed(HERE / "cornell_80x45_post.npz", **out)
        print("post:", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    for what in sys.argv[1:] or ["bunny", "goldens", "post"]:
        {"bunny": bunny, "goldens": goldens, "post": post}[what]()
