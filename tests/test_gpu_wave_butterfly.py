"""The xor butterflies behind every block reduction, bounding box and argmax exchange lanes with v_permlane32_swap / v_permlane16_swap
and DPP row operations instead of __shfl_xor's LDS crossbar.  The device checks them against __shfl_xor lane by lane (gms_debug_f32
op 3): the sums of SLAM.update (SLAM.java:100-115) keep their fixed shape only if every step delivers exactly lane ^ O's value."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap

pytestmark = pytest.mark.gpu


def test_wave_xor_delivers_the_partner_lane():
    m = GridMap(3.2, 3.2, 0.05, (-1.6, -1.6))
    rng = np.random.default_rng(5)
    a = rng.standard_normal(64 * 1024 * 8).astype(np.float32)            # distinct bit patterns in every lane
    code = m.debug_f32(3, a).astype(np.int64)
    assert not (code & 63).any(), f"steps that do not match __shfl_xor: {np.unique(code & 63)}"
    m.close()
