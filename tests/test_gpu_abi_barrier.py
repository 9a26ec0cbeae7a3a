"""-m gpu: the exception barrier of the C-ABI under allocation failure ("no exceptions/longjmp across
the ABI" - the C# side converts codes to exceptions, RaytraceEntity.cs has no native-exception story).

lib/var_faultinject.so is the product's sources with -DYCGE_FAULT_INJECTION=1: the library's own (hidden) operator new throws std::bad_alloc
on the n-th allocation after ycge_debug_fail_allocation(n).  n walks through ycge_scene_upload and ycge_create(n_devices = 2): every call
returns YCGE_OK or YCGE_ERR_OUT_OF_MEMORY - nothing unwinds into the caller, nothing is leaked that a later call trips over - and the
context that survived a failed upload renders the same frames as a fresh one."""
import ctypes as C

import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, build, scenes
from yetanotherconsolegameengine_amd.renderer import RaytraceRenderer
from yetanotherconsolegameengine_amd.scene import flatten

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fi_lib(torch_hip_first):
    L = abi.load_library(build.build_variant("faultinject"))
    L.ycge_debug_fail_allocation.restype = C.c_int
    L.ycge_debug_fail_allocation.argtypes = [C.c_int64]
    return L


@pytest.mark.parametrize("cfg_n", [1, 3])
def test_the_nth_allocation_of_a_scene_upload_fails(fi_lib, cfg_n):
    L = fi_lib
    sc, w, h, ss, pose = scenes.config_scene(cfg_n)
    if cfg_n == 3: w, h = 160, 45
    flat = flatten(sc)
    fresh = RaytraceRenderer(flat, w, h, pose["fov"], ss, capture_debug=True, lib=L)
    fresh.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    r = RaytraceRenderer(None, w, h, pose["fov"], ss, capture_debug=True, lib=L)
    r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
    failed, n = 0, 0
    # (every n up to 40, then strides: a mesh upload makes thousands of small allocations while it builds its tree)
    while True:
        L.ycge_debug_fail_allocation(n)
        rc = L.ycge_scene_upload(r.ctx, flat.byref())
        left = L.ycge_debug_fail_allocation(-1)
        assert rc in (abi.YCGE_OK, abi.YCGE_ERR_OUT_OF_MEMORY), (n, rc, L.ycge_last_error(r.ctx))
        if rc == abi.YCGE_OK:
            assert left >= 0, n          # the countdown never reached zero: this upload made fewer than n allocations
            break
        failed += 1
        assert b"bad_alloc" in L.ycge_last_error(r.ctx)
        # a failed upload leaves a context without a scene, not one with half of one
        assert L.ycge_render_frame(r.ctx, None, None) == abi.YCGE_ERR_NO_SCENE, n
        n += 1 if n < 40 else max(1, n // 3)
    assert failed >= 10, failed
    r.flat = flat
    fresh.set_frame_counter(0); r.set_frame_counter(0)          # (a call on a context without a scene counts a frame, as the reference's TryFlipAndBlit does before it throws, RaytraceRenderer.cs:175)
    for f in range(2):
        fresh.TryFlipAndBlit(); r.TryFlipAndBlit()
        for which in (abi.BUF_CURRENT_HDR, abi.BUF_TAA_HISTORY, abi.BUF_G_NORMAL, abi.BUF_G_DEPTH, abi.BUF_PRIM_ID, abi.BUF_RNG_STATE):
            assert pu.bits_equal(fresh.read(which), r.read(which)), (f, which)
    fresh.close(); r.close()


def test_the_nth_allocation_of_a_two_device_create_fails(fi_lib):
    L = fi_lib
    c = abi.default_config()
    c.fb_width, c.fb_height, c.super_sample = 96, 27, 1
    c.n_devices = 2
    c.devices[0] = c.devices[1] = 0
    failed = 0
    for n in list(range(0, 48)) + list(range(48, 400, 7)):
        ctx = C.c_void_p()
        L.ycge_debug_fail_allocation(n)
        rc = L.ycge_create(C.byref(c), C.byref(ctx))
        left = L.ycge_debug_fail_allocation(-1)
        assert rc in (abi.YCGE_OK, abi.YCGE_ERR_OUT_OF_MEMORY), (n, rc, L.ycge_last_error(None))
        if rc == abi.YCGE_OK:
            assert ctx.value
            L.ycge_destroy(ctx)
            if left >= 0:
                break
        else:
            failed += 1
            assert not ctx.value and b"bad_alloc" in L.ycge_last_error(None), n
    assert failed >= 5, failed
    # the library is still whole: a context made now traces a frame
    sc, w, h, ss, pose = scenes.config_scene(1)
    with RaytraceRenderer(sc, w, h, pose["fov"], ss, lib=L, devices=[0, 0]) as r:
        r.SetCamera(pose["pos"], pose["yaw"], pose["pitch"])
        r.TryFlipAndBlit()
