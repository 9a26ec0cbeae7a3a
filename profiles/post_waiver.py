"""The post stage with the reference's in-place iteration (config.atrous_inplace_exact = 1, bit-exact) and with the quirk waived (0: plain
ping-pong), configs 4 and 5: post_ms and the delivered SDR frame, medians of 9 frames."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
for c in (4, 3, 5):
    sc, w, h, ss, pose = scenes.config_scene(c)
    flat = flatten(sc)
    for exact in (1, 0):
        cfg = abi.default_config(); cfg.atrous_inplace_exact = exact
        r = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=cfg)
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        ms = []
        for i in range(12):
            r.TryFlipAndBlit(want_sdr=True, copy=False); ms.append((r.stats.post_ms, r.stats.total_ms))
        a = np.array(ms[3:])
        print(f"config {c} atrous_inplace_exact={exact}: post_ms median {np.median(a[:,0]):.3f}, frame with SDR read-back {np.median(a[:,1]):.3f} ms")
        r.close()
