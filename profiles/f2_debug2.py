"""Device BVH kernel (csrc/ycge_bvh_build.hip) on random boxes through ycge_debug_device_bvh, against the host builder; per-node split
diagnostics of the device tree when they differ.  python profiles/f2_debug2.py 17 64 2560"""
import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import build
L = C.CDLL(str(build.LIB))
L.ycge_host_build_tree.restype = C.c_int
L.ycge_host_build_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
L.ycge_debug_device_bvh.restype = C.c_int
L.ycge_debug_device_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [17]:
    cen = rng.uniform((-12, 0, -12), (12, 6, 12), (n, 3)).astype(np.float32); r = rng.uniform(0.05, 0.3, (n, 1)).astype(np.float32)
    b = np.concatenate([cen - r, cen + r], 1).astype(np.float32); c = (np.float32(0.5) * (b[:, :3] + b[:, 3:])).astype(np.float32)
    hn = np.zeros((2 * n, 10), np.float32); hl = np.zeros(n, np.int32); st = np.zeros(3, np.int32)
    k = L.ycge_host_build_tree(b.ctypes.data, c.ctypes.data, n, 0, hn.ctypes.data, hl.ctypes.data, st.ctypes.data)
    dn = np.zeros((2 * n, 10), np.float32); dl = np.zeros(n, np.int32); res = np.zeros(16, np.uint32); bn = np.zeros((2 * n, 16), np.int32)
    kd = L.ycge_debug_device_bvh(b.ctypes.data, c.ctypes.data, n, dn.ctypes.data, dl.ctypes.data, res.ctypes.data, bn.ctypes.data)
    same = kd == k and hn[:k].tobytes() == dn[:k].tobytes() and (hl == dl).all()
    print(f"n={n}: host {k} nodes, device {kd}, equal={same}")
    if not same and n <= 64:
        for i in range(min(2 * n, max(kd, 0))):
            s_, cnt, depth, left, inner, pre, ipre, pad = bn[i, :8]
            if left >= 0:
                print(f"  node {i}: start {s_} count {cnt} depth {depth} -> split bin {pad & 255} axis {(pad >> 8) & 255} n_left {pad >> 16} "
                      f"cost {bn[i, 14:15].view(np.float32)[0]} inv_extent {bn[i, 15:16].view(np.float32)[0]}")
        print("  host leaf", hl.tolist()); print("  dev  leaf", dl.tolist())
