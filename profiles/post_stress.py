"""The persistent in-place A-trous launch (k_atrous_stream) against the launch-per-group form over many frames and a moving camera:
every denoised frame and SDR frame must be bit-identical (a race shows up as a differing hash).  python profiles/post_stress.py [frames]"""
import hashlib, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
res = {}
for cfg, (fw, fh) in ((4, (None, None)), (3, (640, 360)), (5, (None, None))):
    for mode in ("2", "0"):
        os.environ["YCGE_POST_MODE"] = mode
        sc, w, h, ss, pose = scenes.config_scene(cfg)
        w, h = (fw or w), (fh or h)
        r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
        hs = []
        ms = []
        for f in range(frames if cfg == 4 else max(8, frames // 6)):
            r.SetCamera((pose["pos"][0] + 0.01 * f, pose["pos"][1], pose["pos"][2] - 0.005 * f), pose["yaw"] + 0.002 * f, pose["pitch"])
            sdr = r.TryFlipAndBlit(want_sdr=True)
            hs.append(hashlib.sha1(r.read(abi.BUF_DENOISED).tobytes() + sdr.tobytes()).hexdigest())
            ms.append(r.stats.post_ms)
        res[(cfg, mode)] = hs
        print(f"config {cfg} {w}x{h} mode {mode}: {len(hs)} frames, post median {np.median(ms[2:]):.3f} ms")
        r.close()
    same = res[(cfg, "2")] == res[(cfg, "0")]
    print(f"config {cfg}: persistent == launch form on every frame: {same}")
    assert same
print("ok")
