"""How the per-rank tail of a sharded scan step grows with the GLOBAL population (weak scaling), measured on one
GPU: the shard stays at 16384 particles, n_global = world x 16384, the other ranks' slots of the gathered buffer
hold a copy of this rank's packed particles (their content does not matter for the timing).  No collectives run:
this isolates the redundant per-rank work (chunk sums over the global population, resample source search)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth

def main():
    cfg = synth.CONFIGS["C3"]
    B, ext, res, n = cfg["beams"], cfg["extent"], cfg["resolution"], cfg["particles"]
    dev = torch.device("cuda", 0)
    tr = synth.make_trace(ext, res, B, T=40, seed=1234)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=2048)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(32):
        m.update(tr.scans[t], tr.poses[t])
    beams = torch.from_numpy(tr.scans[32].view(np.uint8).copy()).to(dev)
    P = torch.from_numpy(synth.make_particles(tr.poses[32], n, seed=99)).to(dev)
    for world in (1, 2, 4, 8):
        pf = ParticleFilter(m, n)
        pf.set_shard(0, n * world)
        partials = torch.zeros(pf.partials_len(), dtype=torch.float64, device=dev)
        glob = torch.zeros(3 * n * world, dtype=torch.float64, device=dev)
        def step(i):
            pf.set_poses_dev(P.data_ptr())
            pf.score_dev(beams.data_ptr(), B)
            pf.local_partials(partials.data_ptr())
            pf.apply_partials(partials.data_ptr(), glob.data_ptr())
            m.update_at_dev(beams.data_ptr(), B, pf)
            pf.import_global(glob.data_ptr())
            pf.resample_if(0.37, 0.5)
        step(0); torch.cuda.synchronize()
        for r in range(1, world):                       # the other ranks' slots: copies of this rank's
            glob[3 * n * r:3 * n * (r + 1)] = glob[:3 * n]
        for i in range(20): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 300
        for i in range(K): step(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print(f"world {world}: n_global {n * world:7d}  {dt * 1e6:7.1f} us/step", flush=True)
        pf.close()

if __name__ == "__main__":
    main()
