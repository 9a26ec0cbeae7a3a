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
ran_ = rec[:, 8] > 0
t0 = t[ran_, 0].min()
dur = t[:, 1] - t[:, 0]
print(f"launch span {t[ran_, 1].max() - t0:.1f} us over the {int(ran_.sum())} bands that ran passes; band 0: {dur[0]:.1f} us for {passes[0]:.0f} passes = {dur[0] / passes[0]:.3f} us per pass")
print("per pass (own duration / passes), bands 0, 1, 2, 10, 60, last:", [round(dur[b] / passes[b], 3) for b in (0, 1, 2, 10, 60, nb - 1)])
# the two-set form records when a band's first level was allowed: skew between chain neighbours and the pace once running
tf = rec[:, 18:20].copy().view(np.uint64).ravel().astype(np.float64) * 0.01
if (tf[ran_] > 0).all() and nb > 8:
    chunks = (nb - 2) // 2
    for name, lo in (("even", 1), ("odd", 1 + chunks)):
        idx = np.arange(lo, lo + chunks)
        idx = idx[ran_[idx]]
        sk = np.diff(tf[idx])
        pace = (t[idx, 1] - tf[idx]) / passes[idx]
        print(f"{name} chain: {len(idx)} bands; first-level skew between neighbours median {np.median(sk):.2f} us (p10 {np.percentile(sk, 10):.2f}, p90 {np.percentile(sk, 90):.2f}); "
              f"pace once running: median {np.median(pace):.3f} us per pass (p10 {np.percentile(pace, 10):.3f}, p90 {np.percentile(pace, 90):.3f}); passes per band median {np.median(passes[idx]):.0f}")
pf = rec[:, 20:30].copy().view(np.uint64).reshape(nb, 5).astype(np.float64)        # profiling instantiation: set 0's shader clocks
if pf.sum() > 0:
    per = pf / np.maximum(passes, 1)[:, None] * 2.0      # set 0 computes every second pass and fetches in the others: x2 = per pass of its kind
    sel = ran_ & (passes > 100)
    names = ("head wait (vmcnt 0)", "compute incl. barrier", "wait for the band above", "fetch issue", "idle at the barrier after fetching")
    print("set 0, shader clocks per pass of that kind, median over bands that ran (band 0 in brackets):")
    for k, nm in enumerate(names):
        print(f"   {nm:36s} {np.median(per[sel, k]):8.0f}   [{per[0, k]:8.0f}]")
    tot = per[sel].sum(axis=1)
    print(f"   compute-kind pass = head + compute: {np.median(per[sel, 0] + per[sel, 1]):.0f}; fetch-kind pass = wait + fetch + idle: {np.median(per[sel, 2] + per[sel, 3] + per[sel, 4]):.0f}")
hw = rec[:, 9]
ran = passes > 0
place = ((hw >> 28) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 15)      # xcc, se, sh, cu
cnt = np.bincount(np.unique(place[ran], return_inverse=True)[1])
print(f"bands that ran passes: {int(ran.sum())} on {len(cnt)} distinct CUs; CUs holding 1/2/3/4+ bands: {[int((cnt == k).sum()) for k in (1, 2, 3)] + [int((cnt >= 4).sum())]}")
# co-residency proper: for every band, how many other bands on its CU overlap it in time for more than half its own duration
ov = np.zeros(nb, int)
for b in np.nonzero(ran)[0]:
    same = np.nonzero(ran & (place == place[b]))[0]
    for o_ in same:
        if o_ != b and min(t[b, 1], t[o_, 1]) - max(t[b, 0], t[o_, 0]) > 0.5 * dur[b]:
            ov[b] += 1
print("bands by number of co-resident bands (overlapping > half their life) 0/1/2:", [int((ov[ran] == k).sum()) for k in (0, 1, 2)])
print("per XCC:", np.bincount(((hw >> 28) & 15)[ran], minlength=8).tolist())
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
    if n < 3:
        continue
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
if offs:          # (only the profiling instantiation records a timeline: YCGE_POST_PROBE_BAND)
    print(f"band {PB+1} starts a level {np.median(offs):.2f} us (median) after band {PB} starts the same level; min {np.min(offs):.2f} max {np.max(offs):.2f}")
big = [(int(bb + 1), round(float(x), 1)) for bb, x in enumerate(end_lag) if x > 40]
print("end lags above 40 us (band, us):", big)
print("end lag by band, every 8th:", [round(float(x), 1) for x in end_lag[::8]])
