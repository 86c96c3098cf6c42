"""Hand-derived known answers for the parts of the path that had none: the likelihood blur, probabilityOf, resample,
getWeightedPose.  PARITY UNPINNED still holds (only a JVM run could pin it) -- these narrow what an unpinned oracle
can hide: the expected values below come from NEITHER restatement.

How the answers are derived
  * exact rational arithmetic (fractions.Fraction): inputs are chosen so that every intermediate of the reference's
    fp64 code is exactly representable (dyadic taps and weights), hence the sequential fp64 result == the exact
    rational result whatever the order of additions -- the expected value is the mathematical definition, typed out;
  * where an intermediate is NOT representable (0.9 * val, 1 - 0.9, / 10f ...) every Java operation is one exact
    rational operation followed by one round-to-nearest-even to double, `rn()` below (float(Fraction) is correctly
    rounded in CPython), in the order the Java source states.

Each case runs against the C oracle and the numpy oracle here (CPU) and against the HIP path through the C-ABI (-m gpu).
References: J/app/Util.java:378-426, J/slam/GridMap.java:233-250,259-294, J/slam/SLAM.java:133-153,165-178,
J/math/MathUtil.java:65-72, J/math/Transform.java:13-32.
"""
import math
from fractions import Fraction as F

import numpy as np
import pytest

from oracle import np_oracle as npo
from oracle import oracle as orc


def rn(x) -> float:
    """round-to-nearest-even of an exact rational to double"""
    return float(F(x))


def f32(x) -> float:
    return float(np.float32(x))


# ----------------------------------------------------------------------------------------------------------------
# (a) threshold + separable blur on a 7 x 7 map, dyadic taps, borders skipped (Util.java:393-401,415-422)
# ----------------------------------------------------------------------------------------------------------------
TAPS3 = [F(1, 4), F(1, 2), F(1, 4)]
TAPS7 = [F(1, 64), F(6, 64), F(15, 64), F(20, 64), F(15, 64), F(6, 64), F(1, 64)]
# log-odds signs of the 7 x 7 map, row y = 0 first: '+' occupied (> 0 -> 1), '-' free (< 0 -> 0), '.' unexplored (== 0 -> 0.5)
MAP7 = ["+..-...",
        ".+.-..-",
        "..+....",
        "---+---",
        "....+..",
        "-....+.",
        "......+"]


def thresholded():
    code = {"+": F(1), "-": F(0), ".": F(1, 2)}                        # GridMap.java:239-244
    return [[code[ch] for ch in row] for row in MAP7]


def blur_exact(p, taps):
    """the mathematical definition: out[y][x] = sum_j taps[j] * (sum_i taps[i] * p[y+j-k][x+i-k]), taps that fall outside
    the map contribute nothing and the rest are NOT renormalised"""
    H, W, k = len(p), len(p[0]), (len(taps) - 1) // 2
    hz = [[sum((taps[i + k] * p[y][x + i] for i in range(-k, k + 1) if 0 <= x + i < W), F(0)) for x in range(W)] for y in range(H)]
    return [[sum((taps[j + k] * hz[y + j][x] for j in range(-k, k + 1) if 0 <= y + j < H), F(0)) for x in range(W)] for y in range(H)]


def log_of_map():
    val = {"+": 2.5, "-": -0.75, ".": 0.0}
    return np.array([[val[ch] for ch in row] for row in MAP7], dtype=np.float64).reshape(-1)


def expected_field(taps):
    e = blur_exact(thresholded(), taps)
    out = np.array([[float(v) for v in row] for row in e], dtype=np.float64)
    assert all(F(out[y, x]) == e[y][x] for y in range(7) for x in range(7))       # every value is a double: exact
    return out.reshape(-1)


# three spot values typed out in full (3 taps), so that the helper above is itself pinned by hand arithmetic:
#   cell (0,0): p = 1, right 0.5, below 0.5, diagonal 1.   H(0,0) = 1/2*1 + 1/4*1/2 = 5/8; H(0,1)(x=0,y=1) = 1/2*1/2 + 1/4*1 = 1/2
#               V = 1/2*5/8 + 1/4*1/2 = 7/16                      (the two missing taps are skipped, not renormalised)
#   cell (3,3): row 3 = 0 0 0 1 0 0 0 -> H(3,3) = 1/2; row 2 = .5 .5 1 .5 .5 .5 .5 -> H(3,2) = 1/4*1 + 1/2*1/2 + 1/4*1/2 = 5/8;
#               row 4 = .5 .5 .5 .5 1 .5 .5 -> H(3,4) = 1/4*1/2 + 1/2*1/2 + 1/4*1 = 5/8;  V = 1/4*5/8 + 1/2*1/2 + 1/4*5/8 = 9/16
#   cell (6,6): p = 1, left .5, above .5, diagonal 1 -> same as (0,0) by symmetry = 7/16
def test_blur_helper_agrees_with_the_typed_out_values():
    e = blur_exact(thresholded(), TAPS3)
    assert e[0][0] == F(7, 16) and e[3][3] == F(9, 16) and e[6][6] == F(7, 16)


def _grid7():
    g = orc.Grid(0.35, 0.35, 0.05, 0.0, 0.0)
    assert (g.W, g.H) == (7, 7)
    return g


@pytest.mark.parametrize("taps", [TAPS3, TAPS7], ids=["3tap", "7tap"])
def test_blur_kat_c_oracle_and_numpy_oracle(taps):
    want = expected_field(taps)
    k = np.array([float(t) for t in taps])
    g = _grid7()
    g.set_kernel(k)
    assert np.array_equal(g.build_likelihood(log_of_map()), want)
    n = npo.NpGrid(0.35, 0.35, 0.05, 0.0, 0.0)
    n.kernel = k
    assert np.array_equal(n.build_likelihood(log_of_map()), want)


@pytest.mark.gpu
@pytest.mark.parametrize("taps", [TAPS3, TAPS7], ids=["3tap", "7tap"])
def test_blur_kat_hip(taps):
    from gridmap_slam_robot_amd import GridMap
    m = GridMap(0.35, 0.35, 0.05, (0.0, 0.0), kernel=[float(t) for t in taps])
    assert (m.W, m.H) == (7, 7)
    m.upload_log(log_of_map())
    m.compute_likelihood_map()
    assert np.array_equal(m.download_likelihood().reshape(-1), expected_field(taps))


# ----------------------------------------------------------------------------------------------------------------
# (b) probabilityOf on a hand-made likelihood field (GridMap.java:259-294)
# ----------------------------------------------------------------------------------------------------------------
# 10 x 10 map at 0.05 m, position (-0.25, -0.25); pose (0, 0, theta = 0): cos = 1, sin = 0 exactly, so
# X = localX * 1 - localY * 0 + 0 = localX (exact), and cell = (int)((X - (-0.25)) / (double)0.05f).
RES = f32(0.05)                                                       # 0.05000000074505806
LIK = {(7, 5): 0.75, (2, 5): 0.5, (0, 0): 0.125, (9, 9): 1.0}        # every other cell 0.25


def lik_field():
    a = np.full((10, 10), 0.25)
    for (x, y), v in LIK.items():
        a[y, x] = v
    return a.reshape(-1)


def cell_of(local):
    """(int)((X - position) / resolution), every step one rational operation + one rounding (GridMap.java:273)"""
    x = rn(F(local) * 1 - 0)                  # localX*cos - localY*sin with cos = 1, sin = 0: exact
    x = rn(F(x) + 0)                          # + p.x = 0
    d = rn(F(x) - F(-0.25))                   # - position.getX()
    q = rn(F(d) / F(RES))                     # / resolution (float widened to double)
    return math.trunc(q)                      # (int): toward zero


BEAMS_B = [  # (localX, localY, wasHit) -> what the reference does with it
    (0.11, 0.01, True),     # cell (7,5)  val 0.75            -> zHit*val + zRandom*1.0/10f
    (-0.13, 0.02, True),    # cell (2,5)  val == 0.5          -> 1.0 / 10f            (the `== 0.5` branch)
    (-0.26, -0.27, True),   # q in (-1, 0) on both axes       -> (int) truncates to cell (0,0): IN the map, val 0.125
    (0.40, 0.0, True),      # 0.65 / (double)0.05f = 12.9999998 -> cell x = 12 (not 13: the float resolution) >= 10 -> skipped (GridMap.java:276)
    (0.11, 0.01, False),    # wasHit == false                 -> skipped (:269)
    (0.249, 0.249, True),   # cell (9,9)  val 1.0
    (0.01, 0.01, True),     # cell (5,5)  val 0.25 (background).  NB (0, 0) would be cell (4,4): 0.25 / (double)0.05f = 4.99999993
]


def expected_probability():
    assert [cell_of(b[0]) for b in BEAMS_B] == [7, 2, 0, 12, 7, 9, 5]
    assert [cell_of(b[1]) for b in BEAMS_B] == [5, 5, 0, 4, 5, 9, 5]
    z_hit = 0.9
    z_random = rn(F(1) - F(z_hit))                                   # GridMap.java:259: 1 - zHit
    max_range = f32(10.0)
    c_rand = rn(F(rn(F(z_random) * 1)) / F(max_range))               # zRandom * 1.0 / SENSOR_MAX_RANGE, left to right
    uniform = rn(F(1) / F(max_range))                                # 1.0 / SENSOR_MAX_RANGE
    product = 1.0
    for val in (0.75, 0.5, 0.125, 1.0, 0.25):                        # the five beams that count, in order
        if val == 0.5:
            f = uniform
        else:
            f = rn(F(rn(F(z_hit) * F(val))) + F(c_rand))             # zHit * val  then  + ...
        product = rn(F(product) * F(f))
    return product


def beams_b():
    lx = np.array([b[0] for b in BEAMS_B]); ly = np.array([b[1] for b in BEAMS_B])
    return orc.make_beams(lx, ly, np.sqrt(lx * lx + ly * ly), [b[2] for b in BEAMS_B])


def test_probability_kat_value_is_what_hand_arithmetic_says():
    # 0.9*0.75+0.01 = 0.685, 0.1, 0.9*0.125+0.01 = 0.1225, 0.91, 0.9*0.25+0.01 = 0.235 -> product ~ 0.0017945...
    assert abs(expected_probability() - 0.685 * 0.1 * 0.1225 * 0.91 * 0.235) < 1e-15


def test_probability_kat_c_oracle_and_numpy_oracle():
    want = expected_probability()
    pose = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    g = orc.Grid(0.5, 0.5, 0.05, -0.25, -0.25)
    assert (g.W, g.H) == (10, 10)
    assert g.probability_of(lik_field(), beams_b(), pose) == want
    assert g.score(lik_field(), beams_b(), pose[None])[0] == want
    n = npo.NpGrid(0.5, 0.5, 0.05, -0.25, -0.25)
    assert n.score(lik_field(), beams_b(), pose[None])[0] == want
    # the other side of the `== 0.5` branch: one ulp away from 0.5 takes the zHit formula
    lik2 = lik_field(); lik2[2 + 5 * 10] = np.nextafter(0.5, 1.0)
    want2 = _with_val(np.nextafter(0.5, 1.0))
    assert want2 != want
    assert g.probability_of(lik2, beams_b(), pose) == want2
    assert n.score(lik2, beams_b(), pose[None])[0] == want2


def _with_val(v):
    z_hit = 0.9
    c_rand = rn(F(rn(F(rn(F(1) - F(z_hit))) * 1)) / F(f32(10.0)))
    product = 1.0
    for val in (0.75, v, 0.125, 1.0, 0.25):
        f = rn(F(1) / F(f32(10.0))) if val == 0.5 else rn(F(rn(F(z_hit) * F(val))) + F(c_rand))
        product = rn(F(product) * F(f))
    return product


@pytest.mark.gpu
def test_probability_kat_hip():
    from gridmap_slam_robot_amd import GridMap, ParticleFilter
    m = GridMap(0.5, 0.5, 0.05, (-0.25, -0.25))
    assert (m.W, m.H) == (10, 10)
    m.upload_likelihood(lik_field())
    pose = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    assert m.probability_of(beams_b(), pose) == expected_probability()
    lik2 = lik_field(); lik2[2 + 5 * 10] = np.nextafter(0.5, 1.0)
    m.upload_likelihood(lik2)
    assert m.probability_of(beams_b(), pose) == _with_val(np.nextafter(0.5, 1.0))
    # several particles, every scoring kernel variant's lane layout: the same pose scores the same
    pf = ParticleFilter(m, 70)
    pf.set_poses(np.tile(pose, (70, 1)))
    pf.score(beams_b())
    assert (pf.get_weights() == _with_val(np.nextafter(0.5, 1.0))).all()


# ----------------------------------------------------------------------------------------------------------------
# (c) resample with U exactly ON cumulative boundaries (SLAM.java:133-153): `while (U > c)` is strict
# ----------------------------------------------------------------------------------------------------------------
W8 = [0.125, 0.25, 0.125, 0.0, 0.25, 0.125, 0.125, 0.0]              # cumulative: .125 .375 .5 .5 .75 .875 1 1
# rand = 0:   U = 0, .125, .25, .375, .5, .625, .75, .875 (all exact in binary)
#   U=0 -> 0 | .125 !> .125 -> 0 | .25 > .125 -> 1 (c=.375) | .375 !> .375 -> 1 | .5 > .375 -> 2 (c=.5) | .625 > .5 -> 3 (c=.5,
#   zero weight) -> 4 (c=.75) | .75 !> .75 -> 4 | .875 > .75 -> 5 (c=.875)
IDX_RAND_0 = [0, 0, 1, 1, 2, 4, 4, 5]
# rand = 0.5: r = .0625, U = .0625 .1875 .3125 .4375 .5625 .6875 .8125 .9375
#   -> 0 | 1 | 1 | 2 (c=.5) | 3 -> 4 (c=.75) | 4 | 5 (c=.875) | 6 (c=1)
IDX_RAND_HALF = [0, 1, 1, 2, 4, 4, 5, 6]


def test_resample_kat_c_oracle_and_numpy_oracle():
    w = np.array(W8)
    for r01, want in ((0.0, IDX_RAND_0), (0.5, IDX_RAND_HALF)):
        got, clamped = orc.resample_indices(w.copy(), r01)
        assert got.tolist() == want and clamped == 0
        assert npo.resample_indices(w, r01).tolist() == want


@pytest.mark.gpu
def test_resample_kat_hip():
    from gridmap_slam_robot_amd import GridMap, ParticleFilter
    m = GridMap(0.5, 0.5, 0.05, (-0.25, -0.25))
    for r01, want in ((0.0, IDX_RAND_0), (0.5, IDX_RAND_HALF)):
        pf = ParticleFilter(m, 8)
        poses = np.zeros((8, 3), dtype=np.float32); poses[:, 0] = np.arange(8) * 0.01
        pf.set_poses(poses)
        pf.set_weights(np.array(W8))
        idx, amb = pf.resample(r01, want_indices=True)
        # on a boundary the device flags the slot as ambiguous (a blocked scan could round differently); here every
        # partial sum is exact, so the indices must be the sequential ones all the same
        assert idx.tolist() == want
        assert np.array_equal(pf.get_poses()[:, 0], poses[want, 0])
        assert np.array_equal(pf.get_weights(), np.array(W8)[want])            # copies keep their weight (SLAM.java:42)
        pf.close()


# ----------------------------------------------------------------------------------------------------------------
# (d) getWeightedPose with angleConstrain wraps (SLAM.java:165-178, MathUtil.java:65-72)
# ----------------------------------------------------------------------------------------------------------------
POSES_D = np.array([[1.0, -2.0, 4.5],          # theta > pi: 4.5 - 2pi
                    [0.5, 0.25, -4.0],         # theta < -pi: (-4 + 2pi) < pi -> + 2pi again -> then - 2pi
                    [-1.5, 3.0, 0.3],          # inside (-pi, pi): the loops add 2pi, then take it off again
                    [2.0, 1.0, 3.5]], dtype=np.float32)
W_D = [0.5, 0.25, 0.125, 0.125]


def angle_constrain_exact(a: float) -> float:
    two_pi = rn(F(math.pi) * 2)
    while a < math.pi:
        a = rn(F(a) + F(two_pi))
    while a > math.pi:
        a = rn(F(a) - F(two_pi))
    return a


def expected_weighted_pose():
    xs = ys = ts = ws = 0.0
    for p, w in zip(POSES_D, W_D):
        xs = rn(F(xs) + F(rn(F(float(p[0])) * F(w))))
        ys = rn(F(ys) + F(rn(F(float(p[1])) * F(w))))
        ts = rn(F(ts) + F(rn(F(angle_constrain_exact(float(p[2]))) * F(w))))
        ws = rn(F(ws) + F(w))
    return np.array([np.float32(rn(F(xs) / F(ws))), np.float32(rn(F(ys) / F(ws))), np.float32(rn(F(ts) / F(ws)))], dtype=np.float32)


def test_weighted_pose_kat_hand_values():
    # x: 1*.5 + .5*.25 - 1.5*.125 + 2*.125 = 0.6875; y: -1 + .0625 + .375 + .125 = -0.4375 (both exact)
    e = expected_weighted_pose()
    assert e[0] == np.float32(0.6875) and e[1] == np.float32(-0.4375)
    # theta: 0.5*(4.5-2pi) + 0.25*(-4+2pi) + 0.125*0.3 + 0.125*(3.5-2pi)
    approx = 0.5 * (4.5 - 2 * math.pi) + 0.25 * (-4 + 2 * math.pi) + 0.125 * 0.3 + 0.125 * (3.5 - 2 * math.pi)
    assert abs(float(e[2]) - approx) < 1e-6
    # an in-range float angle survives the + 2pi, - 2pi round trip exactly (24-bit mantissa, 53-bit arithmetic)
    a = f32(0.3)
    assert angle_constrain_exact(a) == a


def test_weighted_pose_kat_c_oracle_and_numpy_oracle():
    want = expected_weighted_pose()
    assert np.array_equal(orc.weighted_pose(POSES_D, np.array(W_D)), want)
    assert np.array_equal(npo.weighted_pose(POSES_D, np.array(W_D)), want)


@pytest.mark.gpu
def test_weighted_pose_kat_hip():
    from gridmap_slam_robot_amd import GridMap, ParticleFilter
    m = GridMap(0.5, 0.5, 0.05, (-0.25, -0.25))
    pf = ParticleFilter(m, 4)
    pf.set_poses(POSES_D)
    pf.set_weights(np.array(W_D))
    # blocked reduction instead of the sequential sum: identical here because x and y sums are exact, theta within an
    # ulp of float
    got = pf.weighted_pose()
    want = expected_weighted_pose()
    assert got[0] == want[0] and got[1] == want[1]
    assert abs(float(got[2]) - float(want[2])) <= 2.4e-7
