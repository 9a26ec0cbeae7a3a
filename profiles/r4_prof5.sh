#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
echo "== host issue time of the resident loop"; timeout 300 python profiles/rank_flight.py 4 8 resident 4 0,3 2>&1 | tail -1; timeout 300 python profiles/rank_flight.py 2 8 resident 4 0 2>&1 | tail -1
echo "== lit config 5 profile"; bash profiles/run_profiles.sh r4a_c5lit --config 5 --t01 0.5 > gpurun_out/prof_r4a_c5lit.log 2>&1; echo "profiles rc=$?"; head -40 gpurun_out/prof_r4a_c5lit/summary.txt
echo "== post stage: exact (Estrin exponential) and waived"; for c in 4 5; do timeout 300 python - <<PY
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten
sc, w, h, ss, pose = scenes.config_scene($c, t01=0.5)
flat = flatten(sc)
for exact in (1, 0):
    cfg = abi.default_config(); cfg.atrous_inplace_exact = exact
    r = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=cfg)
    r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    ms = []
    for i in range(12):
        r.TryFlipAndBlit(want_sdr=True, copy=False); ms.append((r.stats.post_ms, r.stats.total_ms))
    a = np.array(ms[3:])
    print(f"config $c atrous_inplace_exact={exact}: post_ms median {np.median(a[:,0]):.3f} frame with SDR read-back {np.median(a[:,1]):.3f}")
    r.close()
PY
done
