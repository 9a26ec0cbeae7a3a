#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s -k "denoise or post_stage" > gpurun_out/pytest_post.log 2>&1; echo "pytest(post) rc=$?"; grep -v "^$" gpurun_out/pytest_post.log | tail -25
python - <<'PY'
import time, numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
sc, w, h, ss, pose = scenes.config_scene(4)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for i in range(4):
    t = time.time(); r.TryFlipAndBlit(want_sdr=True); dt = time.time() - t
    print(f"frame {i+1}: trace {r.stats.trace_ms:.3f} taa {r.stats.taa_ms:.3f} post {r.stats.post_ms:.3f} ms, wall {dt*1e3:.1f} ms, exposure {r.stats.exposure:.5f}")
PY
