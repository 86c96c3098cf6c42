"""The near-field workgroups of the single-map ray cast (raycast_near_body: the first 64 steps of every ray counted in an LDS tile
per 64-beam wedge) against the direct-atomic form (GMS_RAYCAST_NEAR=0) and the oracle: same cells, same counts, bit for bit."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import BEAM_DTYPE, GridMap, ParticleFilter, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _scan(rng, B, max_range, short=False):
    a = np.sort(rng.uniform(-np.pi, np.pi, B))
    d = rng.uniform(0.0, 0.3 if short else max_range, B)
    d[rng.random(B) < 0.1] = 0.0                                   # zero-length rays: the start cell three times over
    hit = rng.random(B) < 0.7
    s = np.zeros(B, dtype=BEAM_DTYPE)
    s["local_x"], s["local_y"], s["distance"], s["hit"] = d * np.cos(a), d * np.sin(a), d, hit
    return s


@pytest.mark.parametrize("B", [31, 32, 64, 65, 360, 721])
def test_near_field_equals_direct_atomics_and_the_oracle(monkeypatch, B):
    ext, res = 12.8, 0.05
    rng = np.random.default_rng(B)
    maps = {}
    for near in ("1", "0"):
        monkeypatch.setenv("GMS_RAYCAST_NEAR", near)
        maps[near] = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=1024)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    log = g.new_log()
    poses = [np.array([0.0, 0.0, 0.3], np.float32), np.array([-6.39, 6.39, -2.0], np.float32),       # a corner: wedges leave the map at once
             np.array([6.2, -0.01, 3.1], np.float32), np.array([-7.0, 0.0, 0.0], np.float32),          # the last one starts outside the map
             np.array([1.234, -2.345, 1.0], np.float32)]
    for i, pose in enumerate(poses):
        scan = _scan(rng, B, 10.0, short=(i == 4))
        for m in maps.values():
            m.update(scan, pose)                                   # the two-launch map update: [ray cast | previous apply] -> likelihood
        g.integrate(log, scan, pose)
        a, b = maps["1"].download_log().reshape(-1), maps["0"].download_log().reshape(-1)
        assert np.array_equal(a, b)
        assert np.array_equal(a != 0, log != 0)
        nz = log != 0
        assert np.max(np.abs(a[nz] - log[nz]) / np.abs(log[nz])) <= 1e-13
        assert np.array_equal(maps["1"].download_likelihood(), maps["0"].download_likelihood())
    # the immediate protocol (integrate, then a full rebuild) and the fused scan step take the same near-field route
    scan = _scan(rng, B, 10.0)
    for m in maps.values():
        m.integrate_observation(scan, poses[0])
        m.compute_likelihood_map()
    g.integrate(log, scan, poses[0])
    assert np.array_equal(maps["1"].download_log(), maps["0"].download_log())
    assert np.array_equal(maps["1"].download_likelihood().reshape(-1), g.build_likelihood(maps["1"].download_log().reshape(-1)))
    N = 300
    P = synth.make_particles(poses[0], N, seed=2, sigma_xy=0.02, sigma_theta_deg=0.5)
    pfs = {k: ParticleFilter(m, N) for k, m in maps.items()}
    for t in range(3):
        scan = _scan(rng, B, 6.0)
        for pf in pfs.values():
            pf.slam_update(P, scan, 0.4, 0.5, True)
    assert np.array_equal(maps["1"].download_log(), maps["0"].download_log())
    assert np.array_equal(pfs["1"].get_poses(), pfs["0"].get_poses())
    for pf in pfs.values():
        pf.close()
    for m in maps.values():
        m.close()


def test_tiled_batched_ray_cast_equals_direct_atomics(monkeypatch):
    """batched handles count in LDS tiles per 64-beam wedge (k_raycast_tile); GMS_RAYCAST_TILE=0 sends every count straight to the
    grid: same maps, same fields, through map updates and fused steps"""
    ext, res, B, M, N = 12.8, 0.05, 720, 8, 256
    rng = np.random.default_rng(3)
    maps = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("GMS_RAYCAST_TILE", sw)
        maps[sw] = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=1024)
    poses = np.stack([np.array([0.3 * i - 1.0, 0.2 * i - 0.5, 0.4 * i], np.float32) for i in range(M)])
    for t in range(3):
        scans = np.stack([_scan(rng, B, 9.0) for _ in range(M)])
        for m in maps.values():
            m.update(scans, poses)
        assert np.array_equal(maps["1"].download_log(), maps["0"].download_log())
        assert np.array_equal(maps["1"].download_likelihood(), maps["0"].download_likelihood())
    pfs = {k: ParticleFilter(m, N) for k, m in maps.items()}
    P = np.stack([synth.make_particles(poses[i], N, seed=i, sigma_xy=0.02, sigma_theta_deg=0.5) for i in range(M)])
    for t in range(3):
        scans = np.stack([_scan(rng, B, 6.0) for _ in range(M)])
        for pf in pfs.values():
            pf.slam_update(P, scans, np.full(M, 0.4), 0.5, True)
    assert np.array_equal(maps["1"].download_log(), maps["0"].download_log())
    assert np.array_equal(maps["1"].download_likelihood(), maps["0"].download_likelihood())
    assert np.array_equal(pfs["1"].get_poses(), pfs["0"].get_poses())
    for pf in pfs.values():
        pf.close()
    for m in maps.values():
        m.close()
