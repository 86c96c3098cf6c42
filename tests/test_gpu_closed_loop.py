"""The whole path as a loop, against the oracle scan after scan: motion-model samples -> fused scan step (weights,
bookkeeping, conditional resample, map update at the weighted pose, likelihood rebuild; SLAM.java:80-131 +
GridMapApp.java:185-186) -> next scan.  Every step is checked against the oracle's restatement of the same stage, fed with
the device's state of the step before (so a difference cannot hide behind an earlier one): weight sum, Neff, strongest
particle, the resampling decision and the surviving particles bit for bit, the weighted pose, the log-odds map and the
likelihood field.  Runs with the caller's particle order and with the locality order forced on (k_order)."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from oracle import oracle as orc

from _checks import assert_resample_indices, near_boundary_slots

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order_mode", [-1, 1])
def test_thirty_scan_steps_against_the_oracle(monkeypatch, order_mode):
    ext, res, B, N, T = 12.8, 0.05, 180, 2500, 30                  # 2500: three scoring workgroups, the last one ragged
    tr = synth.make_trace(ext, res, B, T=T + 4, seed=21, n_scans=T + 4)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    monkeypatch.setenv("GMS_SCORE_ORDER", str(order_mode))
    pf = ParticleFilter(m, N)
    log = g.new_log()
    for t in range(3):                                            # a map to localise in
        g.integrate(log, tr.scans[t], tr.poses[t])
        m.update(tr.scans[t], tr.poses[t])
    log = m.download_log().reshape(-1).copy()                     # the oracle's copy of the map, carried along from here
    lik = g.build_likelihood(log)
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)

    particles = synth.make_particles(tr.poses[3], N, seed=5, sigma_xy=0.03, sigma_theta_deg=1.0)
    rng = np.random.default_rng(11)
    resampled = 0
    for t in range(3, 3 + T):
        # the host's side of SLAM.update: motion-model samples of the current particles (Odometry.java:60-96, the oracle's
        # Philox form; a fifth of its heading noise, so that the cloud stays where the map is informative)
        step = tr.poses[t] - tr.poses[t - 1]
        S = orc.sample_motion(particles, float(np.hypot(step[0], step[1])), float(step[2]), seed=77, sequence=t)
        P = S.copy()
        P[:, 2] = particles[:, 2] + np.float32(0.2) * (S[:, 2] - particles[:, 2])
        P = np.ascontiguousarray(P, dtype=np.float32)
        r01 = float(rng.random())

        st = pf.slam_update(P, tr.scans[t], r01, 0.5, fetch=True)
        last = pf.last_step()

        # ---- the oracle's step on the same inputs
        w = g.score(lik, tr.scans[t], P)                           # GridMap.java:260-291
        wn = w.copy()
        ws, strongest = orc.normalize(wn)                          # SLAM.java:100-124
        assert ws > 0 and st["strongest"] == strongest
        assert abs(st["weight_sum"] - ws) <= 1e-11 * ws
        neff = orc.neff(wn)                                        # :180-190
        assert abs(st["neff"] - neff) <= 1e-9 * neff
        assert st["n_zero"] == int((w == 0).sum())
        assert np.allclose(last["weighted_pose"], orc.weighted_pose(P, wn), rtol=0, atol=2e-6)      # :165-178
        assert np.array_equal(last["strongest_pose"], P[strongest])
        want_resample = neff < 0.5 * N                             # GridMapApp.java:185-186
        if abs(neff - 0.5 * N) > 1e-6 * N:
            assert last["did_resample"] == want_resample
        got = pf.get_poses()
        if last["did_resample"]:
            resampled += 1
            idx, _ = orc.resample_indices(np.ascontiguousarray(wn), r01)              # SLAM.java:133-153
            # The oracle scans ITS normalised weights, equal to the device's to 1e-11: besides the slots the device flags
            # (blocked scan vs running sum), only a threshold within that distance of a boundary may pick a neighbour.
            got_idx = pf.last_resample_indices()
            assert_resample_indices(got_idx, idx, last["n_ambiguous"] + near_boundary_slots(wn, r01))
            assert np.array_equal(got, P[got_idx])                                    # copies of the chosen particles, bit for bit
            assert np.allclose(pf.get_weights(), wn[got_idx], rtol=1e-10, atol=0)     # a child keeps its parent's weight (:141-148)
        else:
            assert np.array_equal(got, P)
        particles = got

        # ---- the map branch at the device's weighted pose (GridMap.java:173-250).  The likelihood field is read back every
        # step (the next step's scores depend on it); the log-odds every fourth step only: reading them makes the library
        # run the apply pass it would otherwise defer to the next step's launches, and both ways must be covered.
        g.integrate(log, tr.scans[t], last["weighted_pose"])
        lik = g.build_likelihood(log)
        assert np.array_equal(m.download_likelihood().reshape(-1), lik)               # dirty-tile rebuilds == the full rebuild
        if t % 4 == 0 or t == 2 + T:
            got_log = m.download_log().reshape(-1)
            assert np.array_equal(got_log != 0, log != 0)
            nz = log != 0
            assert np.max(np.abs(got_log[nz] - log[nz]) / np.abs(log[nz])) <= 1e-13
    assert 0 < resampled <= T
    pf.close(); m.close()
