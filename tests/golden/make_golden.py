#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the C oracle (oracle/gms_oracle.c),
cross-checking every one of them against the independent numpy restatement (oracle/np_oracle.py) before
writing.  The reference itself holds no fixtures and cannot run here (no JVM): PARITY UNPINNED by the
reference; these vectors pin the oracle against drift (compiler, libm) and give the GPU path fixed
expected outputs that do not depend on the oracle being importable.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gridmap_slam_robot_amd import synth  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def rays_fixture(seed):
    g = orc.Grid(5.0, 5.0, 0.05, 0.0, 0.0)          # 100 x 100
    n = npo.NpGrid(5.0, 5.0, 0.05, 0.0, 0.0)
    rng = np.random.default_rng(seed)
    rays = rng.uniform(-10, 110, (200, 4)).astype(np.float32)
    rays[::7, 2] = rays[::7, 0]                      # vertical
    rays[::11, 3] = rays[::11, 1]                    # horizontal
    rays[::13, 0] = np.floor(rays[::13, 0])          # start on a cell edge
    rays[5] = rays[5, [0, 1, 0, 1]]                  # zero length
    cells, offs = [], [0]
    for r in rays:
        c = g.trace_ray(*r, 2)
        c2 = np.array(list(n.ray_cells(*r, 2)), dtype=np.int32).reshape(-1, 2)
        assert np.array_equal(c, c2)
        cells.append(c)
        offs.append(offs[-1] + len(c))
    return dict(W=100, H=100, rays=rays, cells=np.concatenate(cells).astype(np.int32), offsets=np.array(offs, dtype=np.int64))


def scan_fixture(extent, res, B, n_scans, N, seed, cross_check=True):
    tr = synth.make_trace(extent, res, B, T=2 * n_scans, seed=seed)
    g = orc.Grid(extent, extent, res, -extent / 2, -extent / 2)
    log = g.new_log()
    for t in range(n_scans):
        g.integrate(log, tr.scans[t], tr.poses[t])
    lik = g.build_likelihood(log)
    P = synth.make_particles(tr.poses[n_scans], N, seed=seed, sigma_xy=res, sigma_theta_deg=0.5)
    w_raw = g.score(lik, tr.scans[n_scans], P)
    if cross_check:
        n = npo.NpGrid(extent, extent, res, -extent / 2, -extent / 2)
        log_n = np.zeros(n.W * n.H)
        for t in range(n_scans):
            n.integrate(log_n, tr.scans[t], tr.poses[t])
        assert np.array_equal(log, log_n)
        assert np.array_equal(lik, n.build_likelihood(log_n))
        assert np.array_equal(w_raw, n.score(lik, tr.scans[n_scans], P))
    wn = w_raw.copy()
    ws, strongest = orc.normalize(wn)
    r01 = 0.25
    idx, clamped = orc.resample_indices(wn, r01)
    return dict(extent=extent, res=res, W=g.W, H=g.H, scans=tr.scans[: n_scans + 1].view(np.uint8), poses=tr.poses[: n_scans + 1],
                log=log, lik=lik, particles=P, w_raw=w_raw, w_norm=wn, weight_sum=ws, strongest=strongest, neff=orc.neff(wn),
                weighted_pose=orc.weighted_pose(P, wn), r01=r01, resample_idx=idx, kernel=g.kernel,
                l_free=g.l_free, l_occ=g.l_occ)


def main():
    for seed in (1, 2, 3):
        np.savez_compressed(os.path.join(HERE, f"rays_seed{seed}.npz"), **rays_fixture(seed))
        np.savez_compressed(os.path.join(HERE, f"scan64_seed{seed}.npz"), **scan_fixture(3.2, 0.05, 72, 6, 128, seed))
    # one BASELINE C1-sized case (512 x 512 @ 5 cm, 360 beams); numpy cross-check on the map only (minutes otherwise)
    np.savez_compressed(os.path.join(HERE, "scan512_seed1.npz"), **scan_fixture(25.6, 0.05, 360, 8, 256, 1, cross_check=False))
    # and one 2 cm case (11-tap kernel whose taps do NOT sum to 1: the `== 0.5` branch goes the other way)
    np.savez_compressed(os.path.join(HERE, "scan2cm_seed2.npz"), **scan_fixture(5.12, 0.02, 180, 6, 128, 2))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
