#!/usr/bin/env python3
"""The per-particle-map SLAM.update with and without the pose refinement (gms_slam_set_refine: findBestPose of every particle
against its own field, SLAM.java:96): milliseconds per update un-bracketed, then microseconds per kernel class under event
brackets; look-ups per second of the refinement launch (lattice poses x hit beams x particles)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gridmap_slam_robot_amd import SLAMParticleMaps, synth


def run(N, ext, B, steps=40):
    T = 48
    frames, _ = synth.make_recording(ext, B, T=T, seed=77)
    start = synth.true_pose(synth.make_world(ext, 77), -1, T)
    dev = torch.device("cuda", 0)
    s = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=max(128, B))
    s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
    s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
    scans, odo, hits = [], [], []
    for f in frames:
        obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        hits.append(int(obs.beams["hit"].astype(bool).sum()))
        scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev))
        odo.append((f.d_center, f.d_theta))
    out = {"particles": N, "grid": [s.W, s.H], "beams": B, "mean_hits": float(np.mean(hits))}
    for refine in (False, True):
        s.reset(); s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
        s.set_refine(refine)
        for i in range(60):
            s.update_dev(scans[i % T].data_ptr(), B, odo[i % T], seed=11, sequence=i)
            if i % 4 == 3: s.resample(0.3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            s.update_dev(scans[(12 + i) % T].data_ptr(), B, odo[(12 + i) % T], seed=11, sequence=100 + i)
        torch.cuda.synchronize()
        upd = (time.perf_counter() - t0) / steps
        s.grid_map.profile(True); s.grid_map.profile_reset()
        for i in range(steps):
            s.update_dev(scans[(12 + i) % T].data_ptr(), B, odo[(12 + i) % T], seed=11, sequence=300 + i)
        torch.cuda.synchronize()
        p = s.grid_map.profile_get(); s.grid_map.profile(False)
        kern = {k: round(ms / n * 1e3, 2) for k, (ms, n) in p.items() if n}
        out["refine" if refine else "plain"] = {"update_ms": upd * 1e3, "kernel_us_bracketed": kern}
        if refine and "refine" in kern:
            look = 1210.0 * np.mean(hits) * N
            out["refine"]["lookups_per_launch"] = look
            out["refine"]["lookups_per_s"] = look / (kern["refine"] * 1e-6)
    s.close()
    return out


if __name__ == "__main__":
    res = []
    for N, ext, B in ((500, 6.0, 90), (500, 6.0, 180), (4096, 12.8, 180))[:int(os.environ.get("PM_CASES", "3"))]:
        r = run(N, ext, B, steps=20 if N > 1000 else 40)
        print(json.dumps(r), flush=True)
        res.append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pm_refine_probe.json"), "w"), indent=1)
