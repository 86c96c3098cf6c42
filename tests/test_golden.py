"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).

CPU part: the C oracle still reproduces them (pins it against compiler / libm drift).
GPU part: the HIP path reproduces them through the C-ABI, without importing the oracle.
The vectors come from the restatement, not from the reference (no JVM here, no fixtures there):
PARITY UNPINNED by the reference itself.
"""
import glob
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import BEAM_DTYPE

from _checks import assert_resample_indices

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
SCANS = sorted(glob.glob(os.path.join(G, "scan*.npz")))
RAYS = sorted(glob.glob(os.path.join(G, "rays_seed*.npz")))


def test_fixtures_present():
    assert len(SCANS) >= 5 and len(RAYS) == 3


# ------------------------------------------------------------------ CPU: oracle vs fixtures
@pytest.mark.parametrize("path", RAYS, ids=os.path.basename)
def test_oracle_reproduces_ray_fixtures(path):
    from oracle import oracle as orc
    d = np.load(path)
    g = orc.Grid(5.0, 5.0, 0.05, 0.0, 0.0)
    for i, r in enumerate(d["rays"]):
        want = d["cells"][d["offsets"][i]:d["offsets"][i + 1]]
        assert np.array_equal(g.trace_ray(*r, 2), want)


@pytest.mark.parametrize("path", SCANS, ids=os.path.basename)
def test_oracle_reproduces_scan_fixtures(path):
    from oracle import oracle as orc
    d = np.load(path)
    ext, res = float(d["extent"]), float(d["res"])
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    assert (g.W, g.H) == (int(d["W"]), int(d["H"]))
    assert np.array_equal(g.kernel, d["kernel"]) and g.l_free == float(d["l_free"]) and g.l_occ == float(d["l_occ"])
    scans = d["scans"].view(BEAM_DTYPE).reshape(len(d["poses"]), -1)
    log = g.new_log()
    for t in range(len(scans) - 1):
        g.integrate(log, scans[t], d["poses"][t])
    assert np.array_equal(log, d["log"])
    lik = g.build_likelihood(log)
    assert np.array_equal(lik, d["lik"])
    w = g.score(lik, scans[-1], d["particles"])
    assert np.array_equal(w, d["w_raw"])
    ws, strongest = orc.normalize(w)
    assert ws == float(d["weight_sum"]) and strongest == int(d["strongest"]) and np.array_equal(w, d["w_norm"])
    assert orc.neff(w) == float(d["neff"])
    assert np.array_equal(orc.weighted_pose(d["particles"], w), d["weighted_pose"])
    assert np.array_equal(orc.resample_indices(w, float(d["r01"]))[0], d["resample_idx"])


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("path", RAYS, ids=os.path.basename)
def test_hip_reproduces_ray_fixtures(path):
    from gridmap_slam_robot_amd import GridMap
    d = np.load(path)
    m = GridMap(5.0, 5.0, 0.05, (0.0, 0.0))
    for i, r in enumerate(d["rays"]):
        want = d["cells"][d["offsets"][i]:d["offsets"][i + 1]]
        assert np.array_equal(m.trace_ray(*r, 2), want), i


@pytest.mark.gpu
@pytest.mark.parametrize("path", SCANS, ids=os.path.basename)
def test_hip_reproduces_scan_fixtures(path):
    from gridmap_slam_robot_amd import GridMap, ParticleFilter
    d = np.load(path)
    ext, res = float(d["extent"]), float(d["res"])
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), kernel=d["kernel"], l_free=float(d["l_free"]), l_occ=float(d["l_occ"]))
    scans = d["scans"].view(BEAM_DTYPE).reshape(len(d["poses"]), -1)
    for t in range(len(scans) - 1):
        m.update(scans[t], d["poses"][t])
    log = m.download_log().reshape(-1)
    assert np.array_equal(log != 0, d["log"] != 0)                        # same cells touched
    nz = d["log"] != 0
    assert np.max(np.abs(log[nz] - d["log"][nz]) / np.abs(d["log"][nz])) <= 1e-13      # bar: 1e-5
    m.upload_log(d["log"])                                              # identical map from here on
    m.compute_likelihood_map()
    assert np.array_equal(m.download_likelihood().reshape(-1), d["lik"])
    N = len(d["particles"])
    pf = ParticleFilter(m, N)
    pf.set_poses(d["particles"])
    pf.score(scans[-1])
    w = pf.get_weights()
    ok = d["w_raw"] > 1e-290
    assert np.max(np.abs(w[ok] - d["w_raw"][ok]) / d["w_raw"][ok]) <= 1e-11            # bar: 1e-5
    st = pf.normalize()
    assert st["strongest"] == int(d["strongest"])
    assert abs(st["weight_sum"] - float(d["weight_sum"])) <= 1e-11 * float(d["weight_sum"])
    assert abs(st["neff"] - float(d["neff"])) <= 1e-9 * float(d["neff"])
    wn = pf.get_weights()
    assert np.max(np.abs(wn[ok] - d["w_norm"][ok]) / d["w_norm"][ok]) <= 1e-11
    assert np.allclose(pf.weighted_pose(), d["weighted_pose"], rtol=0, atol=2e-6)
    idx, amb = pf.resample(float(d["r01"]), want_indices=True)
    assert_resample_indices(idx, d["resample_idx"], amb)
