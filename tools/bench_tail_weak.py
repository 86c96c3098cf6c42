"""How the per-rank tail of a sharded scan step grows with the GLOBAL population (weak scaling), measured on one
GPU: the shard stays at 16384 particles, n_global = world x 16384, the other ranks' slots of the two gather buffers
hold copies of this rank's payloads (their content does not matter for the timing).  No collective runs: this
isolates the redundant per-rank work of gms_slam_update_sharded_begin_dev / _end_dev (fold of the longer partial
vector, cumulative sums over the global population, resample source search)."""
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
    import ctypes as C
    hip = C.CDLL(None)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for world in (1, 2, 4, 8):
        pf = ParticleFilter(m, n)
        pf.set_shard(0, n * world)
        def step(i):
            pf.slam_update_sharded_begin_dev(P.data_ptr(), beams.data_ptr(), B)
            # (the all-gather would run here: the other ranks' slots hold copies of this rank's payloads)
            pf.slam_update_sharded_end_dev(beams.data_ptr(), B, 0.37, 0.5, True)
        pf.slam_update_sharded_begin_dev(P.data_ptr(), beams.data_ptr(), B)
        torch.cuda.synchronize(); m.synchronize()
        pk, nb, pt, nd = pf.gather_buffers()
        for r in range(1, world):
            assert hip.hipMemcpy(pk + r * nb, pk, nb, 3) == 0 and hip.hipMemcpy(pt + r * nd * 8, pt, nd * 8, 3) == 0
        pf.slam_update_sharded_end_dev(beams.data_ptr(), B, 0.37, 0.5, True)
        for i in range(20): step(i)
        torch.cuda.synchronize(); m.synchronize(); t0 = time.perf_counter()
        K = 300
        for i in range(K): step(i)
        torch.cuda.synchronize(); m.synchronize(); dt = (time.perf_counter() - t0) / K
        print(f"world {world}: n_global {n * world:7d}  {dt * 1e6:7.1f} us/step", flush=True)
        pf.close()

if __name__ == "__main__":
    main()
