"""Frames in flight against the synchronous call: ms per frame over N frames, trace launch durations of both."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sc, w, h, ss, pose = scenes.config_scene(cfg)
r = RaytraceRenderer(flatten(sc), w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for rep in range(2):
    for _ in range(8):
        r.TryFlipAndBlit()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr = []
    for _ in range(N):
        r.TryFlipAndBlit(); tr.append(r.stats.trace_ms)
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / N * 1e3
    for _ in range(8):
        r.RenderAsync()
    r.async_trace_ms()
    t0 = time.perf_counter()
    for _ in range(N):
        r.RenderAsync()
    r.Wait()
    ta = (time.perf_counter() - t0) / N * 1e3
    at = r.async_trace_ms()
    tail = f"(trace median {np.median(at):.4f}, min {at.min():.4f}, n {len(at)})" if len(at) else "(no timing events)"
    print(f"config {cfg}: synchronous {ts:.4f} ms/frame (trace median {np.median(tr):.4f}); in flight {ta:.4f} ms/frame {tail}")

# the frame the wrapper asks for (SDR out): synchronous against in flight with the post stage (three page-locked arrays in turn)
M = max(30, N // 5)
for _ in range(4): r.TryFlipAndBlit(want_sdr=True, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(M): r.TryFlipAndBlit(want_sdr=True, copy=False)
ts = (time.perf_counter() - t0) / M * 1e3
for i in range(6): r.RenderAsync(sdr_slot=i % 3)
r.Wait()
t0 = time.perf_counter()
for i in range(M): r.RenderAsync(sdr_slot=i % 3)
r.Wait()
ta = (time.perf_counter() - t0) / M * 1e3
print(f"config {cfg} with the post stage and the SDR read-back: synchronous {ts:.4f} ms/frame; in flight {ta:.4f} ms/frame")
