"""Config 4's shard sizes through the in-library RCCL route with a ONE-rank communicator (what a single-GPU box can run of it): the
whole population of 65 536 as one shard and the 8-GPU shard size 8192, the grouped all-gather and -- GMS_EXCHANGE=p2p -- the grouped
send / receive form of the same exchange; every particle, weight, statistic and map cell equal to the stand-alone filter's, bit for bit.
(RCCL refuses two ranks on one device; tests/test_gpu_two_gpus_rccl.py is the multi-rank run and switches itself on where two or
more devices are visible.)"""
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exchange", ["allgather", "p2p"])
def test_c4_shard_sizes_through_a_one_rank_communicator_equal_the_standalone_filter(exchange, monkeypatch):
    import torch
    from gridmap_slam_robot_amd.distributed import RcclComm
    c = synth.CONFIGS["C4"]
    ext, res, B = c["extent"], c["resolution"], c["beams"]
    T = 64
    tr = synth.make_trace(ext, res, B, T=T, seed=1234, n_scans=T // 2 + 3)
    dev = torch.device("cuda", 0)
    maps = []
    for k in range(2):
        m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=2048)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for t in range(T // 2):
            m.update(tr.scans[t], tr.poses[t])
        maps.append(m)
    if exchange == "p2p":
        monkeypatch.setenv("GMS_EXCHANGE", "p2p")
    comm = RcclComm()
    assert (comm.rank, comm.world) == (0, 1)                          # rccl_ranks == 1
    rng = np.random.default_rng(17)
    for N in (8192, 65536):
        alone, shard = ParticleFilter(maps[0], N), ParticleFilter(maps[1], N)
        shard.set_shard(0, N)
        for s in range(2):
            t = T // 2 + s
            P = torch.from_numpy(synth.make_particles(tr.poses[t], N, seed=99 + s)).to(dev)
            beams = torch.from_numpy(tr.scans[t].view(np.uint8).copy()).to(dev)
            r01 = float(rng.random())
            alone.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, 0.5, True)
            shard.slam_update_sharded_dev(comm, P.data_ptr(), beams.data_ptr(), B, r01, 0.5, True)
            torch.cuda.synchronize()
            assert alone.stats() == shard.stats()
            la, ls = alone.last_step(), shard.last_step()
            assert all(np.array_equal(np.asarray(la[k]), np.asarray(ls[k])) for k in la), (la, ls)
            assert np.array_equal(alone.get_poses(), shard.get_poses()) and np.array_equal(alone.get_weights(), shard.get_weights())
            assert np.array_equal(alone.last_resample_indices(), shard.last_resample_indices())
            assert np.array_equal(maps[0].download_log(), maps[1].download_log())
            assert np.array_equal(maps[0].download_likelihood(), maps[1].download_likelihood())
        alone.close(); shard.close()
    comm.close()
