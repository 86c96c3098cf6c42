"""likelihoodData on demand: the scan steps' dirty-tile rebuilds write the factor table only and leave likelihoodData to a later pass
(gms_ensure_lik).  Whatever the order of calls, a reader must see the field of the LAST computeLikelihoodMap / scan step -- in
particular after an integrateObservation that was not followed by a rebuild (GridMap.java:173-191 touches logData only)."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lazy", ["1", "0"])
def test_likelihood_data_is_the_last_rebuilds_field(monkeypatch, lazy):
    monkeypatch.setenv("GMS_LIK_LAZY", lazy)
    ext, res, B, N = 6.4, 0.05, 120, 400
    tr = synth.make_trace(ext, res, B, T=16, seed=9, n_scans=12)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    pf = ParticleFilter(m, N)
    log = g.new_log()
    for t in range(3):
        m.update(tr.scans[t], tr.poses[t]); g.integrate(log, tr.scans[t], tr.poses[t])
    for t in range(3, 9):
        P = synth.make_particles(tr.poses[t], N, seed=t, sigma_xy=0.02, sigma_theta_deg=0.5)
        pf.slam_update(P, tr.scans[t], 0.37, 0.5, True)                       # the hot path: factor table only
        g.integrate(log, tr.scans[t], pf.last_step()["weighted_pose"])
        lik = g.build_likelihood(log)
        if t % 2:
            assert np.array_equal(m.download_likelihood().reshape(-1), lik)   # ... materialised on demand, deferred apply pass or not
        if t == 5:
            # integrateObservation WITHOUT a rebuild: logData moves on, likelihoodData stays the field of the last step
            m.integrate_observation(tr.scans[9], tr.poses[9]); g.integrate(log, tr.scans[9], tr.poses[9])
            assert np.array_equal(m.download_likelihood().reshape(-1), lik)
            cx, cy = 70, 60                                                   # the middle of a cell: no rounding question in the index
            assert m.get_likelihood(((cx + 0.5) * res - ext / 2, (cy + 0.5) * res - ext / 2)) == lik.reshape(m.H, m.W)[cy, cx]
        if t == 7:
            m.reset()                                                         # reset touches logData only (GridMap.java:129-132)
            assert np.array_equal(m.download_likelihood().reshape(-1), lik)
            log = g.new_log()
            m.compute_likelihood_map()
            assert np.array_equal(m.download_likelihood().reshape(-1), g.build_likelihood(log))
    # a copy carries the field along; the scores that follow come from the same factor table either way
    m2 = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    m2.copy_from(m)
    assert np.array_equal(m2.download_likelihood(), m.download_likelihood())
    pf.close(); m.close(); m2.close()
