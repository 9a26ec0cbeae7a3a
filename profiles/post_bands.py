"""Per-band begin / end of the persistent in-place A-trous launch (k_atrous_stream): chain per pass, lag between neighbouring bands."""
import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from yetanotherconsolegameengine_amd import scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
import os
sc, w, h, ss, pose = scenes.config_scene(int(os.environ.get('CFG', '4')))
ss = int(os.environ.get('SS', ss))          # SS=2: the 3840x2160 trace grid of config 5 (NB=540 half-bands)
r = RaytraceRenderer(sc, w, h, pose["fov"], ss)
r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
import time
for i in range(4):
    t_wall = time.time()
    r.TryFlipAndBlit(want_sdr=True)
    print(f"frame {i}: post {r.stats.post_ms:.3f} ms, wall {time.time() - t_wall:.3f} s", flush=True)
print(f"post {r.stats.post_ms:.3f} ms")
import os
nb = int(os.environ.get('NB', '135'))
buf = np.zeros(nb * 32 + 8000, np.uint32)
r.L.ycge_debug_read_post_progress.restype = C.c_int
r.L.ycge_debug_read_post_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert r.L.ycge_debug_read_post_progress(r.ctx, buf.ctypes.data, buf.size) == 0
rec = buf[:nb * 32].reshape(nb, 32)
tl = buf[nb * 32:].reshape(2, 1000, 4)
t = rec[:, 4:8].copy().view(np.uint64).reshape(nb, 2).astype(np.float64) * 0.01     # us
passes = rec[:, 8].astype(np.float64)
t0 = t[:, 0].min()
dur = t[:, 1] - t[:, 0]
print(f"launch span {t[:, 1].max() - t0:.1f} us; band 0: {dur[0]:.1f} us for {passes[0]:.0f} passes = {dur[0] / passes[0]:.3f} us per pass")
print("per pass (own duration / passes), bands 0, 1, 2, 10, 60, last:", [round(dur[b] / passes[b], 3) for b in (0, 1, 2, 10, 60, nb - 1)])
end_lag = np.diff(t[:, 1])
print(f"end(b) - end(b-1): mean {end_lag.mean():.2f} us, median {np.median(end_lag):.2f}, max {end_lag.max():.2f}")
beg_lag = np.diff(t[:, 0])
print(f"begin(b) - begin(b-1): mean {beg_lag.mean():.2f} us (dispatch)")

# hand-over of level 1400 (bands that have it: 12 b <= 1400 < 12 b + 970)
pub = rec[:, 10:12].copy().view(np.uint64).ravel().astype(np.float64) * 0.01       # this band published "1400 complete"
fet = rec[:, 12:14].copy().view(np.uint64).ravel().astype(np.float64) * 0.01       # this band was allowed to fetch level 1400
cmp_ = rec[:, 14:16].copy().view(np.uint64).ravel().astype(np.float64) * 0.01      # this band began computing level 1400
act = [b for b in range(1, nb) if pub[b] > 0 and pub[b - 1] > 0 and fet[b] > 0 and cmp_[b] > 0]
if act:
    a = np.array(act)
    print(f"level 1400 over {len(a)} band pairs (us): published(b) - published(b-1) median {np.median(pub[a] - pub[a-1]):.2f}")
    print(f"  upstream computes 1399 -> ... this band may fetch 1400: fetch(b) - compute(b-1, 1400) median {np.median(fet[a] - cmp_[a-1]):.2f}")
    print(f"  fetch -> begin computing: {np.median(cmp_[a] - fet[a]):.2f};  begin computing -> published complete: {np.median(pub[a] - cmp_[a]):.2f}")
    print(f"  compute(b,1400) - compute(b-1,1400): {np.median(cmp_[a] - cmp_[a-1]):.2f}")

import os
PB = int(os.environ.get('YCGE_POST_PROBE_BAND', '0'))
for k in (0, 1):
    tt = (tl[k, :, 0].astype(np.uint64) | (tl[k, :, 1].astype(np.uint64) << np.uint64(32))).astype(np.float64) * 0.01
    n = int((tt > 0).sum())
    dt = np.diff(tt[:n])
    sp = tl[k, :n, 2]
    gaps = [(round(float(tt[j] - t0), 0), round(float(dt[j]), 1)) for j in range(n - 1) if dt[j] > 5]
    print(f"band {PB + k}: gaps above 5 us (time since launch, gap):", gaps[:12])
    print(f"band {PB + k}: {n} passes; pass-to-pass median {np.median(dt):.2f} us, mean {dt.mean():.2f}, p90 {np.percentile(dt, 90):.2f}; passes that waited {int((sp > 0).sum())}, mean spins when waiting {sp[sp > 0].mean() if (sp > 0).any() else 0:.1f}")
    print("   first 12 pass gaps:", np.round(dt[:12], 2).tolist(), " middle:", np.round(dt[400:412], 2).tolist())
t60 = (tl[0, :, 0].astype(np.uint64) | (tl[0, :, 1].astype(np.uint64) << np.uint64(32))).astype(np.float64) * 0.01
t61 = (tl[1, :, 0].astype(np.uint64) | (tl[1, :, 1].astype(np.uint64) << np.uint64(32))).astype(np.float64) * 0.01
l60, l61 = tl[0, :, 3].astype(int), tl[1, :, 3].astype(int)
# same-level offset: when band 61 starts level L vs band 60
m60 = {int(l): t for l, t in zip(l60, t60) if t > 0}
offs = [t - m60[int(l)] for l, t in zip(l61, t61) if t > 0 and int(l) in m60]
print(f"band {PB+1} starts a level {np.median(offs):.2f} us (median) after band {PB} starts the same level; min {np.min(offs):.2f} max {np.max(offs):.2f}")
big = [(int(bb + 1), round(float(x), 1)) for bb, x in enumerate(end_lag) if x > 40]
print("end lags above 40 us (band, us):", big)
print("end lag by band, every 8th:", [round(float(x), 1) for x in end_lag[::8]])
