"""Two PROCESSES, one GPU: the single-exchange sharded scan step through distributed.ShardedParticleFilter.scan_step with a
gloo process group (the all-gathers of the two device-resident gather buffers go through torch.distributed), against the
stand-alone filter in the parent.  What a multi-GPU run does per rank, minus RCCL: separate HIP contexts, real ranks,
real offsets, the library's buffers aliased by torch tensors."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EXT, RES, B, N_LOCAL, STEPS = 6.4, 0.05, 160, 512, 5


def _inputs(world):
    from gridmap_slam_robot_amd import synth
    tr = synth.make_trace(EXT, RES, B, T=12, seed=23)
    N = N_LOCAL * world
    sets = [synth.make_particles(tr.poses[3 + t], N, seed=70 + t, sigma_xy=0.04, sigma_theta_deg=2.0) for t in range(STEPS)]
    r01 = np.random.default_rng(4).random(STEPS)
    return tr, sets, r01


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gridmap_slam_robot_amd import GridMap
        from gridmap_slam_robot_amd.distributed import HipShardOps, ShardedParticleFilter
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        tr, sets, r01 = _inputs(world)
        m = GridMap(EXT, EXT, RES, (-EXT / 2, -EXT / 2))
        for t in range(3):
            m.update(tr.scans[t], tr.poses[t])
        ops = HipShardOps(m, N_LOCAL, rank * N_LOCAL, N_LOCAL * world)
        spf = ShardedParticleFilter(N_LOCAL * world, ops)
        assert (spf.rank, spf.world, spf.offset) == (rank, world, rank * N_LOCAL)
        for t in range(STEPS):
            P = torch.from_numpy(sets[t][rank * N_LOCAL:(rank + 1) * N_LOCAL].copy()).to(dev)
            beams = torch.from_numpy(tr.scans[3 + t].view(np.uint8).copy()).to(dev)
            spf.scan_step((P.data_ptr(), beams.data_ptr(), B, True), float(r01[t]), 0.9)
            torch.cuda.synchronize()
        ret[rank] = dict(poses=ops.pf.get_poses(), weights=ops.pf.get_weights(), stats=ops.pf.stats(), log=m.download_log(),
                         lik=m.download_likelihood())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_scan_step_equals_standalone():
    import torch
    import torch.multiprocessing as mp
    from gridmap_slam_robot_amd import GridMap, ParticleFilter
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    dev = torch.device("cuda", 0)
    tr, sets, r01 = _inputs(world)
    m = GridMap(EXT, EXT, RES, (-EXT / 2, -EXT / 2))
    for t in range(3):
        m.update(tr.scans[t], tr.poses[t])
    ref = ParticleFilter(m, N_LOCAL * world)
    for t in range(STEPS):
        P = torch.from_numpy(sets[t]).to(dev)
        beams = torch.from_numpy(tr.scans[3 + t].view(np.uint8).copy()).to(dev)
        ref.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, float(r01[t]), 0.9, True)
    poses, weights, st = ref.get_poses(), ref.get_weights(), ref.stats()
    for r in range(world):
        out = ret[r]
        assert out["stats"] == st
        assert np.array_equal(out["poses"], poses[r * N_LOCAL:(r + 1) * N_LOCAL])
        assert np.array_equal(out["weights"], weights[r * N_LOCAL:(r + 1) * N_LOCAL])
        assert np.array_equal(out["log"], m.download_log()) and np.array_equal(out["lik"], m.download_likelihood())
