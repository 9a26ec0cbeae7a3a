"""The one-call frame with config.multi_device_exchange = YCGE_EXCHANGE_RCCL as a world of ONE (pack -> ncclAllGather -> un-permute -> TAA on devices[0])
against the plain frame: what the slab path costs a frame on one GPU (the only thing a one-GPU box can time of it).
    python profiles/exchange_ms.py <config> [frames]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch  # noqa: F401  (brings its librccl into the process first, as bench.py does)
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sc, w, h, ss, pose = scenes.config_scene(cfg_n)
flat = flatten(sc)
for label, kw in (("plain frame", {}), ("RCCL world of one", "rccl"), ("plain frame", {})):
    if kw == "rccl":
        c = abi.default_config(); c.multi_device_exchange = abi.EXCHANGE_RCCL
        kw = dict(cfg=c, devices=[0])
    r = RaytraceRenderer(flat, w, h, pose["fov"], ss, **kw)
    r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for _ in range(12): r.TryFlipAndBlit()
    fr = []
    for _ in range(N):
        t0 = time.perf_counter(); r.TryFlipAndBlit(); fr.append((time.perf_counter() - t0) * 1e3)
    fr = np.array(fr)
    print(f"config {cfg_n} {label}: frame ms median {np.median(fr):.4f} min {fr.min():.4f} p99 {np.percentile(fr, 99):.4f}; trace_ms (until the frame is assembled) {r.stats.trace_ms:.4f}")
    r.close()
