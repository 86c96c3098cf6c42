"""The particle-sharding protocol (gridmap_slam_robot_amd.distributed) on CPU: world_size-2 gloo
process groups, with a numpy stand-in for the shard-local kernels (the product runs those in HIP; the
stand-in lives here, in tests/, and is checked against the oracle).  What is exercised is the
collective logic: the sparse block-partial all-reduce, the packed all-gather, shard arithmetic, and
that k ranks reproduce 1 rank bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gridmap_slam_robot_amd import _lib
from gridmap_slam_robot_amd.distributed import ShardedParticleFilter
from oracle import oracle as orc

BLOCK = _lib.GMS_BLOCK
STRIDE = _lib.GMS_PARTIAL_STRIDE


class NumpyShardOps:
    """CPU stand-in with the same interface and the same data layout as HipShardOps."""

    def __init__(self, poses, weights, offset, n_global):
        self.pose = np.array(poses, dtype=np.float32)
        self.w = np.array(weights, dtype=np.float64)
        self.n, self.offset, self.n_global = len(self.w), offset, n_global
        self.nblk = (n_global + BLOCK - 1) // BLOCK
        self.st = {}
        self.glob = None

    def new_buffer(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def partials_len(self):
        return self.nblk * STRIDE

    def local_partials(self, t):
        p = np.zeros((self.nblk, STRIDE))
        th = np.array([orc.weighted_pose(np.array([[0, 0, a]], dtype=np.float32), np.ones(1))[2] for a in self.pose[:, 2]],
                      dtype=np.float64)      # angleConstrain through the oracle, float-rounded: fine for a stand-in
        for lb in range((self.n + BLOCK - 1) // BLOCK):
            sl = slice(lb * BLOCK, min(self.n, (lb + 1) * BLOCK))
            w = self.w[sl]
            gb = self.offset // BLOCK + lb
            p[gb] = [w.sum(), w.max(), self.offset + sl.start + int(np.argmax(w)), (w == 0).sum(), 0.0,
                     (w * w).sum(), (self.pose[sl, 0] * w).sum(), (self.pose[sl, 1] * w).sum(), (th[sl] * w).sum()]
        t.copy_(torch.from_numpy(p.reshape(-1)))

    def _fold(self, t):
        p = t.numpy().reshape(self.nblk, STRIDE)
        s = p[:, 0].sum()
        best = np.lexsort((p[:, 2], -p[:, 1]))[0]
        self.st = dict(weight_sum=s, strongest=int(p[best, 2]), n_zero=int(p[:, 3].sum()), neff=s * s / p[:, 5].sum(),
                       wpose=np.array([p[:, 6].sum() / s, p[:, 7].sum() / s, p[:, 8].sum() / s], dtype=np.float32))
        return s

    def apply_partials(self, t, packed):
        s = self._fold(t)
        self.w = self.w / s
        a = np.zeros(self.n, dtype=_lib.PACKED_DTYPE)
        a["w"], a["x"], a["y"], a["theta"] = self.w, self.pose[:, 0], self.pose[:, 1], self.pose[:, 2]
        packed.copy_(torch.from_numpy(a.view(np.float64)))

    def stats_from_partials(self, t):
        self._fold(t)

    def import_global(self, packed_global):
        self.glob = packed_global.numpy().view(_lib.PACKED_DTYPE).copy()

    def resample(self, r01, fraction):
        if fraction is not None and not (self.st["neff"] < fraction * self.n_global):
            return
        idx, _ = orc.resample_indices(np.ascontiguousarray(self.glob["w"]), r01)
        mine = idx[self.offset:self.offset + self.n]
        self.idx = mine
        self.w = self.glob["w"][mine].copy()
        self.pose = np.stack([self.glob["x"][mine], self.glob["y"][mine], self.glob["theta"][mine]], axis=1)

    def stats(self):
        return self.st

    def weighted_pose(self):
        return self.st["wpose"]

    # -- single-exchange protocol (ShardedParticleFilter.scan_step): raw payloads, everything else local
    def exchange_begin(self, inputs):
        self.t_glob = torch.zeros(self.nblk * STRIDE, dtype=torch.float64)
        self.local_partials(self.t_glob)                      # own slots filled, the gather brings the rest
        self.p_glob = torch.zeros(3 * self.n_global, dtype=torch.float64)
        a = np.zeros(self.n, dtype=_lib.PACKED_DTYPE)
        a["w"], a["x"], a["y"], a["theta"] = self.w, self.pose[:, 0], self.pose[:, 1], self.pose[:, 2]      # RAW weights
        self.p_glob[3 * self.offset:3 * (self.offset + self.n)] = torch.from_numpy(a.view(np.float64))

    def gather_views(self):
        lo, hi = self.offset // BLOCK, (self.offset + self.n) // BLOCK
        return (self.p_glob, self.p_glob[3 * self.offset:3 * (self.offset + self.n)],
                self.t_glob, self.t_glob[lo * STRIDE:hi * STRIDE])

    def exchange_end(self, inputs, r01, fraction):
        s = self._fold(self.t_glob)
        self.w = self.w / s                                   # weight /= weightSum for the own particles
        self.glob = self.p_glob.numpy().view(_lib.PACKED_DTYPE).copy()
        self.glob["w"] = self.glob["w"] / s                   # ... and the same division on the gathered raw copies
        self.resample(r01, fraction)


def make_inputs(n_global, seed=3):
    rng = np.random.default_rng(seed)
    w = rng.uniform(0, 1, n_global) ** 6
    w[rng.integers(0, n_global, n_global // 9)] = 0.0
    poses = rng.normal(0, 1, (n_global, 3)).astype(np.float32)
    return poses, w


def run_filter(rank, world, n_global, r01):
    poses, w = make_inputs(n_global)
    n, off = ShardedParticleFilter.shard_of(n_global, world, rank)
    ops = NumpyShardOps(poses[off:off + n], w[off:off + n], off, n_global)
    spf = ShardedParticleFilter(n_global, ops)
    spf.normalize()
    st = dict(spf.stats())
    # only this rank's slots of the partial vector were non-zero before the all-reduce
    mine = torch.zeros_like(spf.partials)
    ops.w = w[off:off + n].copy()
    ops.local_partials(mine)
    ops.w = ops.glob["w"][off:off + n].copy()
    own = mine.numpy().reshape(-1, STRIDE)
    lo, hi = off // BLOCK, (off + n + BLOCK - 1) // BLOCK
    assert not own[:lo].any() and not own[hi:].any()
    wn = ops.w.copy()
    spf.resample(r01, None)
    spf.refresh_stats()
    return dict(st=st, wn=wn, idx=ops.idx.copy(), pose=ops.pose.copy(), wpose_after=spf.weighted_pose().copy(),
                glob_w=ops.glob["w"].copy())


def _worker(rank, world, port, n_global, r01, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = run_filter(rank, world, n_global, r01)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2])
def test_two_ranks_reproduce_one_rank(world):
    n_global, r01 = 4 * BLOCK, 0.371
    single = run_filter(0, 1, n_global, r01)             # no process group: world 1
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_global, r01, ret), nprocs=world, join=True)
    n = n_global // world
    for r in range(world):
        out = ret[r]
        assert out["st"]["weight_sum"] == single["st"]["weight_sum"]
        assert out["st"]["strongest"] == single["st"]["strongest"]
        assert out["st"]["n_zero"] == single["st"]["n_zero"]
        assert out["st"]["neff"] == single["st"]["neff"]
        assert np.array_equal(out["st"]["wpose"], single["st"]["wpose"])
        assert np.array_equal(out["glob_w"], single["glob_w"])                       # all-gather layout
        assert np.array_equal(out["wn"], single["wn"][r * n:(r + 1) * n])
        assert np.array_equal(out["idx"], single["idx"][r * n:(r + 1) * n])          # each rank fills its own slots
        assert np.array_equal(out["pose"], single["pose"][r * n:(r + 1) * n])
        assert np.array_equal(out["wpose_after"], single["wpose_after"])
    # and the single-rank run agrees with the sequential oracle
    poses, w = make_inputs(n_global)
    wo = w.copy()
    ws, strongest = orc.normalize(wo)
    assert single["st"]["strongest"] == strongest
    assert abs(single["st"]["weight_sum"] - ws) <= 1e-12 * ws
    assert abs(single["st"]["neff"] - orc.neff(wo)) <= 1e-9 * orc.neff(wo)
    assert np.allclose(single["wn"], wo, rtol=1e-12, atol=0)
    idx_o, _ = orc.resample_indices(np.ascontiguousarray(single["glob_w"]), r01)
    assert np.array_equal(single["idx"], idx_o)


def test_shard_arithmetic():
    assert ShardedParticleFilter.shard_of(8 * BLOCK, 8, 3) == (BLOCK, 3 * BLOCK)
    assert ShardedParticleFilter.shard_of(1000, 1, 0) == (1000, 0)          # a single rank may hold any count
    with pytest.raises(ValueError):
        ShardedParticleFilter.shard_of(8 * BLOCK + 8, 8, 0)                  # shard not a multiple of GMS_BLOCK
    with pytest.raises(ValueError):
        ShardedParticleFilter.shard_of(1001, 2, 0)


def test_hip_shard_ops_refuse_to_run_without_a_gpu(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    from gridmap_slam_robot_amd.distributed import HipShardOps
    with pytest.raises(RuntimeError):
        HipShardOps(None, BLOCK, 0, BLOCK)


def run_single_exchange(rank, world, n_global, r01):
    poses, w = make_inputs(n_global)
    n, off = ShardedParticleFilter.shard_of(n_global, world, rank)
    ops = NumpyShardOps(poses[off:off + n], w[off:off + n], off, n_global)
    spf = ShardedParticleFilter(n_global, ops)
    spf.scan_step(None, r01, None)
    return dict(st=dict(ops.st), idx=ops.idx.copy(), pose=ops.pose.copy(), w=ops.w.copy(), glob_w=ops.glob["w"].copy())


def _worker_single(rank, world, port, n_global, r01, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = run_single_exchange(rank, world, n_global, r01)
    finally:
        dist.destroy_process_group()


def test_single_exchange_step_two_ranks_reproduce_one_rank_and_the_two_collective_protocol():
    """ShardedParticleFilter.scan_step: raw particles + block partials all-gathered, everything after it local.
    Two ranks == one rank bit for bit, and both == the all-reduce / all-gather protocol above."""
    n_global, r01, world = 4 * BLOCK, 0.371, 2
    single = run_single_exchange(0, 1, n_global, r01)
    old = run_filter(0, 1, n_global, r01)
    assert single["st"]["weight_sum"] == old["st"]["weight_sum"] and single["st"]["strongest"] == old["st"]["strongest"]
    assert np.array_equal(single["glob_w"], old["glob_w"]) and np.array_equal(single["idx"], old["idx"])
    assert np.array_equal(single["pose"], old["pose"])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_single, args=(world, _free_port(), n_global, r01, ret), nprocs=world, join=True)
    n = n_global // world
    for r in range(world):
        out = ret[r]
        for k in ("weight_sum", "strongest", "n_zero", "neff"):
            assert out["st"][k] == single["st"][k]
        assert np.array_equal(out["st"]["wpose"], single["st"]["wpose"])
        assert np.array_equal(out["glob_w"], single["glob_w"])
        assert np.array_equal(out["idx"], single["idx"][r * n:(r + 1) * n])
        assert np.array_equal(out["pose"], single["pose"][r * n:(r + 1) * n])
        assert np.array_equal(out["w"], single["w"][r * n:(r + 1) * n])
