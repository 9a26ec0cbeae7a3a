"""A few frames of config 4 WITH the post stage (denoise / exposure / tonemap), for rocprofv3 --kernel-trace --stats."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc, w, h, ss, pose = scenes.config_scene(cfg)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for i in range(4):
    t = time.time(); r.TryFlipAndBlit(want_sdr=True); dt = time.time() - t
    print(f"frame {i+1}: trace {r.stats.trace_ms:.3f} taa {r.stats.taa_ms:.3f} post {r.stats.post_ms:.3f} ms, wall {dt*1e3:.1f} ms, exposure {r.stats.exposure:.5f} serial chunks {r.stats.exposure_serial_chunks:.0f}")
