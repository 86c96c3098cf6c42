#!/usr/bin/env python3
"""Timings of the SURVEY section 8(f) rows on one GPU: findBestPose lattice search (gms_pf_refine_poses), motion-model
sampling (gms_pf_sample_motion), scan de-skew (gms_map_deskew), combined map (gms_map_combine)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmap_slam_robot_amd import GridMap, Observation, ParticleFilter, synth


def timeit(fn, sync, iters=20):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    sync()
    return (time.perf_counter() - t0) / iters


def main():
    out = {}
    for name, ext, res, B, N in (("reference app (6 m, 5 cm, 360 beams, 500 particles)", 6.0, 0.05, 360, 500),
                                 ("C2 (51.2 m, 5 cm, 360 beams, 1024 particles)", 51.2, 0.05, 360, 1024)):
        tr = synth.make_trace(ext, res, B, T=12, seed=5)
        m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
        for t in range(6):
            m.update(tr.scans[t], tr.poses[t])
        pf = ParticleFilter(m, N)
        P = synth.make_particles(tr.poses[6], N, seed=1, sigma_xy=0.05, sigma_theta_deg=3.0)
        obs = Observation(tr.scans[6])
        def refine():
            pf.set_poses(P); pf.refine_poses(obs)
        dt = timeit(refine, m.synchronize, iters=5)
        n_hit = int(tr.scans[6]["hit"].sum())
        out[name] = {"refine_ms": dt * 1e3, "lattice_poses_per_particle": 1210,
                     "beam_evals_per_s": N * 1210 * n_hit / dt}
        dt = timeit(lambda: pf.sample_motion(0.05, 0.02, 7, 3), m.synchronize, iters=50)
        out[name]["sample_motion_us"] = dt * 1e6
        ang = np.linspace(0, 2 * np.pi, B, endpoint=False); dist = np.full(B, 2.0); hit = np.ones(B, dtype=np.uint8)
        dt = timeit(lambda: m.deskew(ang, dist, hit, 0.05, 0.02), m.synchronize, iters=20)
        out[name]["deskew_host_roundtrip_us"] = dt * 1e6
    batch = GridMap(51.2, 51.2, 0.05, (-25.6, -25.6), n_maps=16)
    one = GridMap(51.2, 51.2, 0.05, (-25.6, -25.6))
    dt = timeit(lambda: one.combine_from(batch), one.synchronize, iters=20)
    out["combined map, 16 x 1024^2"] = {"combine_ms": dt * 1e3, "GB_per_s": 17 * 1024 * 1024 * 8 / dt / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
