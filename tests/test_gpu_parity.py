"""-m gpu: the HIP path against the oracle through the C-ABI, on the same seeded inputs.

Bars (north_star / SURVEY 8d): primary hit index and t bit-exact, integer RNG state bit-exact,
traversal counters equal, radiance RMS <= 1e-4 (we additionally report exact-bit mismatch counts).
"""
import numpy as np
import pytest

import parity_util as pu
from yetanotherconsolegameengine_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _assert_parity(stats, label):
    print(label, stats)
    for k in ("rays", "prim_id", "sub_id", "hit_t", "rng_state", "sky", "g_depth"):
        assert stats[k + "_mismatch"] == 0, f"{label}: {k} differs in {stats[k + '_mismatch']} elements"
    for k in ("current_hdr", "taa_history", "g_albedo", "g_normal"):
        assert stats[k + "_rms"] <= pu.RMS_TOL, f"{label}: {k} RMS {stats[k + '_rms']}"
    for k in ("n_rays", "n_box", "n_tri", "n_prim", "n_vox"):
        assert stats[k][0] == stats[k][1], f"{label}: counter {k} oracle {stats[k][0]} != hip {stats[k][1]}"


@pytest.mark.parametrize("cfg_n", [1, 2])
def test_analytic_scenes_three_frames(product_lib, oracle, cfg_n):
    sc, w, h, ss, pose = scenes.config_scene(cfg_n)
    o, g = pu.run_pair(oracle, sc, w, h, ss, pose, frames=1)
    _assert_parity(pu.compare_frame(o, g), f"cfg{cfg_n} frame1")
    for f in (2, 3):
        o.render(stages=1, threads=8); g.TryFlipAndBlit()
        _assert_parity(pu.compare_frame(o, g), f"cfg{cfg_n} frame{f}")
    o.close(); g.close()


def test_bunny_reduced_resolution(product_lib, oracle):
    sc, w, h, ss, pose = scenes.config_scene(3)
    o, g = pu.run_pair(oracle, sc, 320, 90, 1, pose, frames=2)
    _assert_parity(pu.compare_frame(o, g), "cfg3 320x180")
    assert pu.bits_equal(o.accel(abi.ACCEL_MESH_NODES), g.accel(abi.ACCEL_MESH_NODES))
    assert pu.bits_equal(o.accel(abi.ACCEL_MESH_LEAF_INDEX), g.accel(abi.ACCEL_MESH_LEAF_INDEX))
    o.close(); g.close()


def test_dragon_standin_small(product_lib, oracle):
    sc, w, h, ss, pose = scenes.config_scene(4, small=True)
    o, g = pu.run_pair(oracle, sc, 256, 72, 1, pose, frames=2)
    _assert_parity(pu.compare_frame(o, g), "cfg4-small")
    o.close(); g.close()


def test_voxel_world_small(product_lib, oracle):
    sc, w, h, ss, pose = scenes.config_scene(5, small=True)
    o, g = pu.run_pair(oracle, sc, 160, 45, 2, pose, frames=2)
    _assert_parity(pu.compare_frame(o, g), "cfg5-small ss2")
    o.close(); g.close()
