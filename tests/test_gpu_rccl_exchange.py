"""-m gpu: the collective BEHIND the C-ABI (missing 3; north_star: "the framebuffer is tiled across the GPUs of one node with an
RCCL all-gather over xGMI to reassemble the frame", called "via a thin P/Invoke C-ABI" by a host that makes ONE TryFlipAndBlit call,
RaytraceEntity.cs:230).  config.multi_device_exchange = YCGE_EXCHANGE_RCCL (ABI 9): ycge_render_frame traces every device's tiles into a
slab, queues ONE ncclAllGather over in-process communicators (ncclCommInitAll; librccl.so dlopen'ed), un-permutes the gathered frame on
devices[0] and runs TAA and the post stage there.  A one-GPU box can run it as a world of ONE (RCCL refuses two ranks on one device): the
frame goes through pack -> all-gather -> un-permute and must equal the plain frame bit for bit.  Without librccl.so the context falls back
to the peer push and says so (ycge_exchange_query)."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _query(r):
    mode, world = C.c_int32(-1), C.c_int32(-1)
    r._check(r.L.ycge_exchange_query(r.ctx, C.byref(mode), C.byref(world)))
    return mode.value, world.value


@pytest.mark.parametrize("cfg_n", [1, 3])
def test_all_gather_behind_the_one_call_equals_the_plain_frame(product_lib, cfg_n):
    sc, w, h, ss, pose = scenes.config_scene(cfg_n)
    if cfg_n == 3: w, h = 320, 90
    flat = flatten(sc)
    cfg = abi.default_config()
    cfg.multi_device_exchange = abi.EXCHANGE_RCCL
    g = RaytraceRenderer(flat, w, h, pose["fov"], ss, cfg=cfg, devices=[0], capture_debug=True, count_work=True)
    one = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, count_work=True)
    assert _query(g) == (abi.EXCHANGE_RCCL, 1), "librccl.so not found or its communicator did not come up"
    assert _query(one) == (abi.EXCHANGE_PEER_PUSH, 1)
    for r in (g, one):
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    for f in range(3):
        sg = g.TryFlipAndBlit(want_sdr=True); s1 = one.TryFlipAndBlit(want_sdr=True)
        assert pu.bits_equal(sg, s1), f
        for b in (abi.BUF_CURRENT_HDR, abi.BUF_G_ALBEDO, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_SKY_MASK, abi.BUF_TAA_HISTORY, abi.BUF_DENOISED):
            assert pu.bits_equal(g.read(b), one.read(b)), (f, b)
        for k in ("n_rays", "n_box", "n_tri", "n_prim"):
            assert getattr(g.stats, k) == getattr(one.stats, k), (f, k)
    g.Resize(96, 27, 1); one.Resize(96, 27, 1)          # the slabs are per size
    assert pu.bits_equal(g.TryFlipAndBlit(want_sdr=True), one.TryFlipAndBlit(want_sdr=True))
    g.close(); one.close()


def test_rccl_refuses_two_ranks_on_one_device_and_says_so(product_lib):
    cfg = abi.default_config()
    cfg.multi_device_exchange = abi.EXCHANGE_RCCL
    with pytest.raises(abi.YcgeError) as e:
        RaytraceRenderer(None, 96, 27, 45.0, 1, cfg=cfg, devices=[0, 0])
    assert e.value.status == abi.YCGE_ERR_INVALID_ARG and "distinct devices" in str(e.value)


def test_without_librccl_the_peers_push_their_tiles():
    """YCGE_RCCL_LIB names what to dlopen (tests): a name that does not exist = a host without RCCL.  The same request then gives a context that
    pushes tiles between its devices as before, says so, and renders the same frames.  (A process of its own: the loader looks once.)"""
    code = r'''
import ctypes as C, sys
sys.path[:0] = [%r, %r]
import numpy as np
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
sc, w, h, ss, pose = scenes.config_scene(1)
cfg = abi.default_config(); cfg.multi_device_exchange = abi.EXCHANGE_RCCL
g = RaytraceRenderer(sc, w, h, pose["fov"], ss, cfg=cfg, devices=[0, 0]); one = RaytraceRenderer(sc, w, h, pose["fov"], ss)
mode, world = C.c_int32(-1), C.c_int32(-1)
assert g.L.ycge_exchange_query(g.ctx, C.byref(mode), C.byref(world)) == 0 and (mode.value, world.value) == (abi.EXCHANGE_PEER_PUSH, 2), (mode.value, world.value)
for r in (g, one): r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
for f in range(2):
    a, b = g.TryFlipAndBlit(want_sdr=True), one.TryFlipAndBlit(want_sdr=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f
print("fallback ok")
''' % (str(ROOT), str(ROOT / "tests"))
    import os
    env = dict(os.environ, YCGE_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout + r.stderr
