"""Opt-in log-normalisation (gms_pf_set_log_normalize; SURVEY.md section 9.6 -- beyond the reference's arithmetic, default off):
weight = exp(logw - max logw) / sum instead of the plain product of up to 720 factors, which underflows for nearly every
particle of a wide cloud.  Checked against (1) a sequential numpy restatement over the device's own log-weights (the
normalisation arithmetic alone: 1e-12), (2) the oracle's sum of log factors (orc_score_log: 1e-9), and through the fused scan
step (statistics, weighted pose, resampling on the rescaled weights).  Off, the filter's outputs are the reference's product --
that is what every other test asserts."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from gridmap_slam_robot_amd._lib import GMS_ERR_STATE, GmsError
from oracle import oracle as orc

from _checks import assert_resample_indices

pytestmark = pytest.mark.gpu

EXT, RES, B, N = 10.24, 0.02, 720, 4096


def _world():
    tr = synth.make_trace(EXT, RES, B, T=12, seed=31)
    g = orc.Grid(EXT, EXT, RES, -EXT / 2, -EXT / 2)
    m = GridMap(EXT, EXT, RES, (-EXT / 2, -EXT / 2))
    log = g.new_log()
    for t in range(6):
        g.integrate(log, tr.scans[t], tr.poses[t])
        m.update(tr.scans[t], tr.poses[t])
    return tr, g, m, log, g.build_likelihood(log)


def _reference(lw, P):
    """exp(lw - max) / sum, sequentially, and the statistics that follow from it"""
    M = lw.max()
    v = np.exp(lw - M)
    S = 0.0
    for x in v:
        S += x
    wn = v / S
    return M, v, S, wn


def test_normalise_from_log_weights_instead_of_the_underflowing_product():
    tr, g, m, log, lik = _world()
    P = synth.make_particles(tr.poses[6], N, seed=4)                  # sigma 0.10 m / 5 degrees: the headline cloud
    scan = tr.scans[6]
    plain = ParticleFilter(m, N)
    plain.set_poses(P)
    plain.score(scan)
    st_plain = plain.normalize()
    pf = ParticleFilter(m, N)
    pf.set_log_normalize(True)
    pf.set_poses(P)
    pf.score(scan)
    st = pf.normalize()
    lw = pf.get_log_weights()
    assert np.array_equal(lw, plain.get_log_weights())               # the scoring pass itself is the same
    w = pf.get_weights()
    M, v, S, wn = _reference(lw, P)
    assert st["max_log_weight"] == M
    assert abs(st["weight_sum"] - S) <= 1e-12 * S
    big = wn > 1e-280
    assert np.max(np.abs(w[big] - wn[big]) / wn[big]) <= 1e-12
    assert (w[~big] <= 1e-279).all()
    assert st["strongest"] == int(np.argmax(lw))                      # first maximum (SLAM.java:110-115)
    assert abs(st["neff"] - 1.0 / float((wn * wn).sum())) <= 1e-9 * st["neff"]
    assert np.max(np.abs(pf.weighted_pose() - orc.weighted_pose(P, wn))) < 1e-5
    # the oracle's sum of log factors (an independent route to the same log-weights)
    lw_o = g.score_log(lik, scan, P)
    _, _, _, wn_o = _reference(lw_o, P)
    big = wn_o > 1e-30
    assert np.max(np.abs(w[big] - wn_o[big]) / wn_o[big]) <= 1e-9
    # what the option is for: the plain product leaves far fewer particles with any weight at all
    assert st["n_zero"] < st_plain["n_zero"] and st["neff"] >= st_plain["neff"] * 0.999
    assert (w > 0).sum() > (plain.get_weights() > 0).sum()
    # resampling runs on the rescaled weights
    idx, amb = pf.resample(0.37, want_indices=True)
    want, _ = orc.resample_indices(w.copy(), 0.37)
    assert_resample_indices(idx, want, amb)
    assert np.array_equal(pf.get_poses(), P[idx])


def test_the_fused_scan_step_with_log_normalisation_and_switching_it_off_again():
    import torch
    tr, g, m, log, lik = _world()
    dev = torch.device("cuda", 0)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    P = synth.make_particles(tr.poses[6], N, seed=9)
    Pd = torch.from_numpy(P).to(dev)
    beams = torch.from_numpy(tr.scans[6].view(np.uint8).copy()).to(dev)
    pf = ParticleFilter(m, N)
    pf.set_log_normalize(True)
    pf.slam_update_dev(Pd.data_ptr(), beams.data_ptr(), B, 0.61, -1.0, True)        # no resample: the weights stay comparable
    torch.cuda.synchronize()
    lw = pf.get_log_weights()
    M, v, S, wn = _reference(lw, P)
    w = pf.get_weights()
    big = wn > 1e-280
    assert np.max(np.abs(w[big] - wn[big]) / wn[big]) <= 1e-12
    st = pf.stats()
    assert st["max_log_weight"] == M and abs(st["weight_sum"] - S) <= 1e-12 * S
    wp = pf.last_step()["weighted_pose"]
    assert np.max(np.abs(np.asarray(wp).reshape(-1)[:3] - orc.weighted_pose(P, wn))) < 1e-5
    # the map was updated at that pose
    g.integrate(log, tr.scans[6], np.asarray(wp, dtype=np.float32).reshape(-1)[:3])
    assert np.max(np.abs(m.download_log().reshape(-1) - log)) < 1e-10
    # with the conditional resample inside the step
    P2 = synth.make_particles(tr.poses[7], N, seed=10)
    P2d = torch.from_numpy(P2).to(dev)
    beams2 = torch.from_numpy(tr.scans[7].view(np.uint8).copy()).to(dev)
    # the same scan through the separate entry points first (the map is the one the step will score against)
    sep = ParticleFilter(m, N)
    sep.set_log_normalize(True)
    sep.set_poses(P2)
    sep.score(tr.scans[7])
    st_sep = sep.normalize()
    wn_sep = sep.get_weights()
    idx_sep, amb = sep.resample(0.25, want_indices=True)
    want, _ = orc.resample_indices(wn_sep.copy(), 0.25)
    assert_resample_indices(idx_sep, want, amb)
    pf.slam_update_dev(P2d.data_ptr(), beams2.data_ptr(), B, 0.25, 2.0, True)       # fraction 2: neff < 2 N always -> resample
    torch.cuda.synchronize()
    last = pf.last_step()
    assert last["did_resample"]
    idx = pf.last_resample_indices().reshape(-1)
    assert np.array_equal(idx, idx_sep.reshape(-1))                  # the fused step is the separate calls, bit for bit
    assert np.array_equal(pf.get_poses(), P2[idx]) and (np.diff(idx) >= 0).all()
    assert pf.stats()["weight_sum"] == st_sep["weight_sum"] and np.isfinite(st_sep["weight_sum"]) and st_sep["weight_sum"] >= 1.0
    # off again: the reference's product
    pf.set_log_normalize(False)
    ref = ParticleFilter(m, N)
    P3 = synth.make_particles(tr.poses[8], N, seed=12, sigma_xy=0.01, sigma_theta_deg=0.2)     # tight: the plain product stays finite
    for f in (pf, ref):
        f.set_poses(P3)
        f.score(tr.scans[8])
    st_a, st_b = pf.normalize(), ref.normalize()
    assert st_a == st_b and st_a["weight_sum"] > 0 and np.isfinite(st_a["neff"]) and np.array_equal(pf.get_weights(), ref.get_weights())


def test_weights_the_caller_sets_are_never_rescaled_and_shards_refuse_the_option():
    tr, g, m, log, lik = _world()
    pf = ParticleFilter(m, 512)
    pf.set_log_normalize(True)
    w = np.random.default_rng(0).uniform(0.1, 1.0, 512)
    pf.set_poses(np.zeros((512, 3), dtype=np.float32))
    pf.set_weights(w)
    st = pf.normalize()                                               # not a scoring pass: plain normalisation of what was set
    wn = w.copy()
    ws, strongest = orc.normalize(wn)
    assert abs(st["weight_sum"] - ws) <= 1e-12 * ws and st["strongest"] == strongest
    assert np.max(np.abs(pf.get_weights() - wn) / wn) <= 1e-12
    sh = ParticleFilter(m, 256)
    sh.set_shard(256, 1024)
    with pytest.raises(GmsError) as e:
        sh.set_log_normalize(True)
    assert e.value.code == GMS_ERR_STATE
    sh.set_log_normalize(False)                                       # turning it off is always allowed


@pytest.mark.parametrize("beams", [96, 400])
def test_batched_maps_and_short_scans(beams):
    """Three maps in one handle (every kernel carries a map index; the scale is per map) and a scan short enough for a single
    scoring segment (96 beams: the weights come out of the scoring launch directly, no segment products to combine)."""
    M, ext, res, n = 3, 6.4, 0.05, 700
    traces = [synth.make_trace(ext, res, beams, T=8, seed=70 + i) for i in range(M)]
    mb = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=beams)
    for t in range(4):
        mb.update(np.stack([tr.scans[t] for tr in traces]), np.stack([tr.poses[t] for tr in traces]))
    pf = ParticleFilter(mb, n)
    pf.set_log_normalize(True)
    P = np.stack([synth.make_particles(tr.poses[4], n, seed=5 + i) for i, tr in enumerate(traces)])
    pf.set_poses(P)
    pf.score(np.stack([tr.scans[4] for tr in traces]))
    st = pf.normalize()
    lw = pf.get_log_weights().reshape(M, n)
    w = pf.get_weights().reshape(M, n)
    for i in range(M):
        Mx, v, S, wn = _reference(lw[i], P[i])
        assert st[i]["max_log_weight"] == Mx and abs(st[i]["weight_sum"] - S) <= 1e-12 * S
        big = wn > 1e-280
        assert np.max(np.abs(w[i][big] - wn[big]) / wn[big]) <= 1e-12
        assert st[i]["strongest"] == int(np.argmax(lw[i]))
    # the fused batched step takes the same route
    pf2 = ParticleFilter(mb, n)
    pf2.set_log_normalize(True)
    pf2.slam_update(P, np.stack([tr.scans[4] for tr in traces]), np.full(M, 0.4), -1.0, False)
    assert np.array_equal(pf2.get_weights(), pf.get_weights())


def test_only_an_unconsumed_scoring_pass_is_rescaled():
    """(round-4 advisor finding) weights the caller sets AFTER a scoring pass, and the copies a resampling leaves, are not the
    log-weights' any more: the next normalise / getWeightedPose must take them as they are, not rebuild them from the stale log-weights."""
    tr, g, m, log, lik = _world()
    n = 1024
    P = synth.make_particles(tr.poses[6], n, seed=21)
    pf = ParticleFilter(m, n)
    pf.set_log_normalize(True)
    pf.set_poses(P)
    pf.score(tr.scans[6])                                             # a scoring pass nobody consumes ...
    w = np.random.default_rng(1).uniform(0.1, 1.0, n)
    pf.set_weights(w)                                                 # ... because the caller replaces its weights
    st = pf.normalize()
    wn = w.copy()
    ws, strongest = orc.normalize(wn)
    assert abs(st["weight_sum"] - ws) <= 1e-12 * ws and st["strongest"] == strongest
    assert np.max(np.abs(pf.get_weights() - wn) / wn) <= 1e-12
    # score -> normalise -> resample -> getWeightedPose (GridMapApp.java:186-192): the pose of the COPIES with the weights they kept
    pf.set_poses(P)
    pf.score(tr.scans[6])
    pf.normalize()
    wn = pf.get_weights()
    idx, amb = pf.resample(0.61, want_indices=True)
    want, _ = orc.resample_indices(wn.copy(), 0.61)
    assert_resample_indices(idx, want, amb)
    got = pf.weighted_pose()
    ref = orc.weighted_pose(P[idx], wn[idx])
    assert np.allclose(got, ref, rtol=0, atol=2e-6), (got, ref)
    assert np.max(np.abs(pf.get_weights() - wn[idx])) == 0           # the copies' weights are untouched by the read-out
