"""Denoised buffer of a few frames as hashes: run under YCGE_POST_PERSISTENT=0 and =1 and compare the lines (a race shows as a differing hash)."""
import hashlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc, w, h, ss, pose = scenes.config_scene(cfg)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for f in range(frames):
    sdr = r.TryFlipAndBlit(want_sdr=True)
    den = r.read(abi.BUF_DENOISED)
    print(f"frame {f+1} denoised {hashlib.sha1(den.tobytes()).hexdigest()[:16]} sdr {hashlib.sha1(sdr.tobytes()).hexdigest()[:16]} post_ms {r.stats.post_ms:.3f}")
