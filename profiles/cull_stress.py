"""Timed (non-counting) kernels of a voxel world against the oracle from random camera poses - inside chunks' air, below ground,
far outside, every direction: the grid cull (solid-voxel boxes) must never change a pixel.  python profiles/cull_stress.py [poses]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import parity_util as pu
import oracle_binding as ob
from yetanotherconsolegameengine_amd import build, scenes
build.build_library()
oracle = ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
sc, w, h, ss, pose = scenes.config_scene(5, small=True)
bad = 0
for k in range(n):
    p = pose["pos"]
    ps = dict(pose, pos=(p[0] + float(rng.uniform(-120, 120)), p[1] + float(rng.uniform(-60, 60)), p[2] + float(rng.uniform(-120, 120))),
              yaw=float(rng.uniform(-3.2, 3.2)), pitch=float(rng.uniform(-1.5, 1.5)))
    o, g = pu.run_pair(oracle, sc, 64, 18, 2, ps, frames=1, oracle_threads=16, count=False)
    st = pu.compare_frame(o, g, check_counters=False)
    mism = {key: v for key, v in st.items() if key.endswith("_mismatch") and v}
    if mism: bad += 1; print("pose", k, ps, mism)
    o.close(); g.close()
print(f"{n} poses, {bad} with a difference")
sys.exit(1 if bad else 0)
