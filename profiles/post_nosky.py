"""Post stage on a frame without a sky pixel (ground plane seen from above) at 1920x1080 and 3840x2160: time per level of the
in-place A-trous iteration when every band is busy (levels = W/2 + 3 H/2)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import Plane, PointLight, Scene, Solid, Sphere, vec3
s = Scene()
s.Add(Plane(vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), Solid(vec3(0.6, 0.6, 0.55)), 0.05, 0.0))
s.Add(Sphere(vec3(0.0, 0.5, -2.0), 0.5, Solid(vec3(0.8, 0.2, 0.2))))
s.Lights.append(PointLight(vec3(1.0, 3.0, -1.0), vec3(1, 1, 1), 30.0))
for (w, h, ss) in ((640, 180, 1), (1280, 360, 1), (1920, 540, 1), (1920, 540, 2)):
    r = RaytraceRenderer(s, w, h, 60.0, ss)
    r.SetCamera((0.0, 1.5, 0.0), 0.0, -1.2)
    for f in range(4):
        r.TryFlipAndBlit(want_sdr=True)
    W, H = w * ss, h * 2 * ss
    levels = W // 2 + 3 * (H // 2)
    print(f"{W}x{H}: post {r.stats.post_ms:.3f} ms, {levels} levels -> {r.stats.post_ms * 1e3 / levels:.3f} us per level (whole stage)")
    r.close()
