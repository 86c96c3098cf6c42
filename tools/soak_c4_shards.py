"""Soak: the full-size C4 scenario of tests/test_gpu_configs.py (8 shards x 8192 on one GPU, nine 2048^2 maps on nine streams)
repeated N times; reports where a shard's statistics, log-odds or likelihood field differ from the stand-alone filter's.
This is what found the tile-state race of the likelihood kernel (a wavefront skipping its rows of a uniform tile, ~1 run in 6
here, invisible in single-map tests): 80 clean runs since the fix.  usage: soak_c4_shards.py [runs]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth

def dtod(dst, src, n):
    hip = C.CDLL(None)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n), 3) == 0

def run(rep):
    dev = torch.device("cuda", 0)
    c = synth.CONFIGS["C4"]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], c["particles"]
    world, n = 8, N // 8
    tr = synth.make_trace(ext, res, B, T=16, seed=1234, n_scans=7)
    ref_map = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    maps = [GridMap(ext, ext, res, (-ext / 2, -ext / 2)) for _ in range(world)]
    for t in range(4):
        for m in [ref_map] + maps:
            m.update(tr.scans[t], tr.poses[t])
    ref = ParticleFilter(ref_map, N)
    pfs = []
    for r, m in enumerate(maps):
        pf = ParticleFilter(m, n); pf.set_shard(r * n, N); pfs.append(pf)
    bad = False
    for t, (frac, sig_xy, sig_th) in ((4, (0.9, res, 0.3)), (5, (0.9, 2 * res, 0.5)), (6, (-1.0, res, 0.3))):
        Ph = synth.make_particles(tr.poses[t], N, seed=300 + t, sigma_xy=sig_xy, sigma_theta_deg=sig_th)
        P = torch.from_numpy(Ph).to(dev)
        beams = torch.from_numpy(tr.scans[t].view(np.uint8).copy()).to(dev)
        r01 = 0.2718 + 0.1 * t
        if frac >= 0:
            ref.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, frac, True)
        else:
            ref.set_poses_dev(P.data_ptr()); ref.score_dev(beams.data_ptr(), B); ref.normalize(fetch=False)
            ref_map.update_at_dev(beams.data_ptr(), B, ref)
        for r, pf in enumerate(pfs):
            pf.slam_update_sharded_begin_dev(P[r * n:(r + 1) * n].data_ptr(), beams.data_ptr(), B)
        for m in maps:
            m.synchronize()
        bufs = [pf.gather_buffers() for pf in pfs]
        for r in range(world):
            for q in range(world):
                if q != r:
                    pk, nb, pt, nd = bufs[r]
                    dtod(pk + q * nb, bufs[q][0] + q * nb, nb)
                    dtod(pt + q * nd * 8, bufs[q][2] + q * nd * 8, nd * 8)
        if os.environ.get("SYNC_AFTER_COPY"):
            torch.cuda.synchronize()
        for pf in pfs:
            pf.slam_update_sharded_end_dev(beams.data_ptr(), B, r01, frac, True)
        ref_lik = ref_map.download_likelihood(); ref_log = ref_map.download_log()
        st = ref.stats()
        for r, (pf, m) in enumerate(zip(pfs, maps)):
            lg, lk = m.download_log(), m.download_likelihood()
            dl, dk = (lg != ref_log), (lk != ref_lik)
            if pf.stats() != st or dl.any() or dk.any():
                bad = True
                ys, xs = np.nonzero(dk)
                print(f"rep {rep} t {t} shard {r}: stats_equal={pf.stats() == st} log_diff={int(dl.sum())} lik_diff={int(dk.sum())}",
                      (f"lik box x[{xs.min()},{xs.max()}] y[{ys.min()},{ys.max()}]" if dk.any() else ""), flush=True)
    for pf in pfs + [ref]:
        pf.close()
    return bad

if __name__ == "__main__":
    nbad = 0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        nbad += run(rep)
    print("bad runs:", nbad)
