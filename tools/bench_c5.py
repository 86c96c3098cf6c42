#!/usr/bin/env python3
"""BASELINE config 5 ("throughput mode"): M independent 1024x1024 maps x 4096 particles x 1080 beams in one
batched handle (n_maps = M), every kernel carrying the map index.  Prints one JSON line (not the bench.py
contract line: that one is quoted on C3)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", type=int, default=64)
    ap.add_argument("--particles", type=int, default=4096)
    ap.add_argument("--beams", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import torch
    from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
    M, N, B = args.maps, args.particles, args.beams
    ext, res, T = 51.2, 0.05, 16
    dev = torch.device("cuda", 0)
    traces = [synth.make_trace(ext, res, B, T=T, seed=100 + (i % 8)) for i in range(min(M, 8))]
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=2048)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(T // 2):
        m.update(np.stack([traces[i % 8].scans[t] for i in range(M)]), np.stack([traces[i % 8].poses[t] for i in range(M)]))
    scans_dev = [torch.from_numpy(np.stack([traces[i % 8].scans[t] for i in range(M)]).view(np.uint8).copy()).to(dev) for t in range(T)]
    poses_dev = []
    for s in range(4):
        t = T // 2 + s
        P = np.stack([synth.make_particles(traces[i % 8].poses[t], N, seed=7 + i + 64 * s) for i in range(M)])
        poses_dev.append(torch.from_numpy(P).to(dev))
    pf = ParticleFilter(m, N)
    r01 = np.random.default_rng(1).random((256, M))
    def step(i):
        s = i % 4
        pf.slam_update_dev(poses_dev[s].data_ptr(), scans_dev[T // 2 + s].data_ptr(), B, r01[i % 256], 0.5, True)
    for i in range(args.warmup): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps): step(args.warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.profile(True); m.profile_reset()
    for i in range(10): step(i)
    torch.cuda.synchronize()
    prof = {k: round(v[0] / 10, 4) for k, v in m.profile_get().items() if v[1]}
    m.profile(False)
    st = pf.stats()
    n_hit = int(traces[0].scans[T // 2]["hit"].sum())
    print(json.dumps({"config": f"C5: {M} maps x 1024^2 @ 5 cm x {N} particles x {B} beams", "ms_per_step": el / args.steps * 1e3,
                      "particle_scan_evals_per_s": M * N * args.steps / el, "beam_evals_per_s": M * N * n_hit * args.steps / el,
                      "kernel_ms_per_step": prof, "neff_map0": st[0]["neff"], "n_zero_map0": st[0]["n_zero"]}))

if __name__ == "__main__":
    main()
