"""Profiling build (-DYCGE_DBG_VOXSTAT): what the continuation rays of a voxel world do in k_wf_trace_p - scene-tree steps by kind, objects culled
by the box of their solid voxels, grids entered, cell steps and cell fetches, lanes busy per round, clocks per phase.
    python profiles/build_variant.py voxstat -DYCGE_DBG_VOXSTAT=1 ; YCGE_LIB=.../var_voxstat.so python profiles/vox_stats.py [t01]"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
t01 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
sc, w, h, ss, pose = scenes.config_scene(5, t01=t01)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
f = r.L.ycge_debug_read_batch_stats; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p]
def read():
    a = (C.c_uint64 * 64)(); assert f(r.ctx, a) == 0; return np.array(list(a), dtype=np.float64).reshape(8, 8)
for _ in range(4): r.TryFlipAndBlit()
a = read(); n = 6; ms = []
for _ in range(n): r.TryFlipAndBlit(); ms.append(r.stats.trace_ms)
d = (read() - a) / n
L, W = d[2], d[3]
print(f"config 5 t01 {t01}: trace {np.median(ms):.3f} ms (profiling build); k_wf_trace_p per frame:")
print(f"  scene-tree steps: {L[0]/1e6:.1f} M node, {L[1]/1e6:.1f} M leaf, {L[2]/1e6:.1f} M object; objects culled by their solid box {L[3]/1e6:.1f} M; grids asked {L[4]/1e6:.1f} M, entered {L[5]/1e6:.1f} M")
print(f"  cell steps {L[6]/1e6:.1f} M of which {L[7]/1e6:.1f} M fetch a cell")
print(f"  wavefronts {W[7]:.0f}, rounds {W[0]/1e6:.2f} M; lanes with a ray per round {W[1]/W[0]:.1f}, in the tree {W[2]/W[0]:.1f}, in a grid {W[3]/W[0]:.1f}")
tt = W[4] + W[5] + W[6]
print(f"  wavefront clocks (100 MHz ticks): refill {100*W[4]/tt:.0f} %, tree phase {100*W[5]/tt:.0f} %, cell phase {100*W[6]/tt:.0f} %; per round {tt/W[0]:.1f} ticks = {tt/W[0]*24:.0f} shader clocks at 2.4 GHz")
