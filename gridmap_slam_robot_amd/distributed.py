"""Particles sharded over the GPUs of one node: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI) for the two exchange steps the path really has.

The reference has no distributed code at all; this is the sharding BASELINE.json's
north_star asks for.  Rank r holds the contiguous particle block [r*n, (r+1)*n) and a full replica of
the map (the deterministic integer-count map update is run redundantly: 720 rays are cheaper than
broadcasting 32 MiB).  Per scan:

  1. all-reduce(SUM) of the block-partial vector (GMS_PARTIAL_STRIDE doubles per 256-particle block,
     each rank's own blocks filled, zero elsewhere).  Adding zeros is exact, so every rank ends up
     with the same partials whatever the rank count, and folds them in block order: weightSum,
     strongest, n_zero are bit-identical for 1, 2, 4, 8 GPUs (SLAM.java:87-121).
     The same vector carries sum w^2 and sum x*w, y*w, theta*w, so Neff and the weighted pose
     (SLAM.java:165-190) need no further exchange.
  2. all-gather of the packed normalised particles {w, x, y, theta} (24 B each): every rank then
     fills its own slots of the systematic resample (SLAM.java:133-153) from the same global array.

Both messages are small (18 KiB and 1.5 MiB at 65 536 particles): latency-bound, one RCCL call each.

`ShardedParticleFilter.scan_step` is the protocol the product actually runs per scan, with ONE exchange: the ranks
all-gather their RAW particles {w, x, y, theta} and their block partials (the gathered partial vector is what the
all-reduce above assembles), both gathers issued together, and normalisation, statistics, map update and resample
are local afterwards (the normalised weight of another rank's particle is the owner's division repeated on the
gathered copy).  libgridmapslam.so does the same with RCCL inside (`gms_slam_update_sharded_dev` over an `RcclComm`:
one C-ABI call per scan, no Python collective wrappers on the critical path); `scan_step` is the torch.distributed
route and the one the gloo tests exercise.

The collective logic is independent of where the shard kernels run: `ops` is the object that performs
them.  The product uses HipShardOps (libgridmapslam.so, device pointers of torch CUDA tensors, the
library running on torch's current stream).  There is no CPU implementation in this package; the gloo
tests under tests/ inject their own stand-in to exercise the protocol.
"""
from __future__ import annotations

from typing import Optional

import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .gridmap import GridMap, ParticleFilter


class _DeviceBuffer:
    """A device allocation of the library as a 1-D float64 array for torch.as_tensor (no copy, no ownership)."""

    def __init__(self, ptr: int, n_doubles: int):
        self.__cuda_array_interface__ = {"shape": (int(n_doubles),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HipShardOps:
    """Shard-local kernels through the C-ABI on the current CUDA(HIP) device."""

    def __init__(self, grid_map: GridMap, n_local: int, offset: int, n_global: int):
        if not torch.cuda.is_available():
            raise RuntimeError("HipShardOps needs a HIP device (there is no CPU path)")
        self.map = grid_map
        self.pf = ParticleFilter(grid_map, n_local)
        self.pf.set_shard(offset, n_global)
        self.device = torch.device("cuda", torch.cuda.current_device())
        # The library and the torch.distributed collectives must share ONE real stream: c10d orders a collective
        # against torch's current stream, and the default stream's handle is 0, which the library reads as "use your
        # own stream" -- two unrelated streams then, and a collective can start before the kernels that fill its
        # buffer have run (seen as diverging filters after a few scans).  So: the current stream if it is a real one,
        # else a stream of our own, made current around every collective (ShardedParticleFilter._on_stream).
        cur = torch.cuda.current_stream()
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device=self.device)
        grid_map.set_stream(self.stream.cuda_stream)

    def new_buffer(self, n_doubles: int) -> torch.Tensor:
        return torch.zeros(n_doubles, dtype=torch.float64, device=self.device)

    def partials_len(self) -> int:
        return self.pf.partials_len()

    def local_partials(self, partials: torch.Tensor):
        self.pf.local_partials(partials.data_ptr())

    def apply_partials(self, partials: torch.Tensor, packed_local: torch.Tensor):
        self.pf.apply_partials(partials.data_ptr(), packed_local.data_ptr())

    def stats_from_partials(self, partials: torch.Tensor):
        self.pf.stats_from_partials(partials.data_ptr())

    def import_global(self, packed_global: torch.Tensor):
        self.pf.import_global(packed_global.data_ptr())

    def resample(self, r01: float, fraction: Optional[float]):
        if fraction is None:
            self.pf.resample(r01)
        else:
            self.pf.resample_if(r01, fraction)

    # -- single-exchange scan step (ShardedParticleFilter.scan_step): inputs = (dev_poses or 0, dev_beams, B, integrate)
    def exchange_begin(self, inputs):
        dev_poses, dev_beams, B, _ = inputs
        self.pf.slam_update_sharded_begin_dev(dev_poses, dev_beams, B)

    def gather_views(self):
        """(packed_global, packed_local, partials_global, partials_local): torch tensors that ALIAS the handle's two
        gather buffers (no copy); the local ones are this rank's slices, so the all-gathers run in place."""
        if getattr(self, "_views", None) is None:
            pk, nb, pt, nd = self.pf.gather_buffers()
            world = self.pf.n_global // self.pf.n
            rank = self.pf.offset // self.pf.n
            g = torch.as_tensor(_DeviceBuffer(pk, world * nb // 8), device=self.device)
            t = torch.as_tensor(_DeviceBuffer(pt, world * nd), device=self.device)
            self._views = (g, g[rank * nb // 8:(rank + 1) * nb // 8], t, t[rank * nd:(rank + 1) * nd])
        return self._views

    def exchange_end(self, inputs, r01: float, fraction: Optional[float]):
        _, dev_beams, B, integrate = inputs
        self.pf.slam_update_sharded_end_dev(dev_beams, B, r01, -1.0 if fraction is None else fraction, integrate)

    def stats(self) -> dict:
        return self.pf.stats()

    def weighted_pose(self) -> np.ndarray:
        return self.pf.weighted_pose()


class RcclComm:
    """One rank's RCCL communicator owned by libgridmapslam.so (gms_comm_*): the two exchanges of a sharded
    scan step are enqueued by the library itself, so a step is ONE C-ABI call and the host spends no time in
    a Python collective wrapper.  torch.distributed is only the out-of-band channel for the 128-byte id."""

    def __init__(self, device: Optional[int] = None, group=None):
        import ctypes as C
        lib = _lib.load()
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(lib.gms_comm_load(rccl.encode() if os.path.exists(rccl) else None))     # the copy torch already holds
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.gms_comm_unique_id(uid))
        if world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = C.create_string_buffer(box[0], 128)
        self.rank, self.world = rank, world
        self.device = torch.cuda.current_device() if device is None else device
        h = C.c_void_p()
        _lib.check(lib.gms_comm_create(C.byref(h), uid, rank, world, self.device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().gms_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the module globals may already be gone
            pass


class ShardedParticleFilter:
    """ParticleFilter whose particles are split over the ranks of a process group."""

    def __init__(self, n_global: int, ops, group=None, coll=None):
        """coll (optional): an object with rank, world, all_reduce_sum(tensor) and all_gather_into(out, mine) that performs the two
        collectives instead of torch.distributed (tests run several shards as threads of one process on one GPU with it)"""
        self.group = group
        self.coll = coll
        self.world = coll.world if coll is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        # measurement aid: run the collectives even on a one-rank group (GMS_FORCE_COLLECTIVES=1)
        self.force = coll is None and dist.is_initialized() and os.environ.get("GMS_FORCE_COLLECTIVES") == "1"
        self.rank = coll.rank if coll is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.n_global = n_global
        self.n_local, self.offset = self.shard_of(n_global, self.world, self.rank)
        self.ops = ops
        self.partials = ops.new_buffer(ops.partials_len())
        self.packed_local = ops.new_buffer(3 * self.n_local)            # 24 B per particle
        self.packed_global = ops.new_buffer(3 * n_global)

    @staticmethod
    def shard_of(n_global: int, world: int, rank: int):
        """Contiguous equal blocks; shard boundaries must fall on GMS_BLOCK so that the reduction
        blocks never straddle ranks."""
        if n_global % world:
            raise ValueError(f"{n_global} particles do not split evenly over {world} ranks")
        n = n_global // world
        if world > 1 and n % _lib.GMS_BLOCK:
            raise ValueError(f"shard size {n} must be a multiple of GMS_BLOCK={_lib.GMS_BLOCK}")
        return n, rank * n

    # ------------------------------------------------------------------------------------------
    def _on_stream(self):
        """Context in which the collectives are issued: the shard ops' stream made current (so that c10d orders them
        after the library's kernels and the library's next kernels after them); inputs produced on the stream that
        was current before are waited for first.  A no-op for CPU stand-ins."""
        st = getattr(self.ops, "stream", None)
        if st is None:
            import contextlib
            return contextlib.nullcontext()
        prev = torch.cuda.current_stream()
        if prev.cuda_stream != st.cuda_stream:
            st.wait_stream(prev)
        return torch.cuda.stream(st)

    def _all_gather_start(self):
        """Start the all-gather of the packed particles; returns a work handle (None = already complete)."""
        if self.world == 1 and not self.force:
            self.packed_global.copy_(self.packed_local)
            return None
        if self.coll is not None:
            self.coll.all_gather_into(self.packed_global, self.packed_local)
            return None
        try:
            return dist.all_gather_into_tensor(self.packed_global, self.packed_local, group=self.group, async_op=True)
        except (RuntimeError, NotImplementedError, TypeError):
            parts = list(self.packed_global.chunk(self.world))
            return dist.all_gather(parts, self.packed_local, group=self.group, async_op=True)

    def normalize_begin(self):
        """SLAM.update's bookkeeping over all ranks (SLAM.java:87-129), first half: block partials,
        all-reduce, weight /= weightSum (the statistics -- weight sum, strongest, weighted pose -- are final
        after this), and the START of the all-gather of the packed normalised particles.  Work that needs
        only the weighted pose (the map update) can be enqueued between normalize_begin and normalize_end:
        RCCL runs the all-gather on its own stream beside it."""
        with self._on_stream():
            self.ops.local_partials(self.partials)
            if self.coll is not None:
                if self.world > 1:
                    self.coll.all_reduce_sum(self.partials)
            elif self.world > 1 or self.force:
                dist.all_reduce(self.partials, op=dist.ReduceOp.SUM, group=self.group)
            self.ops.apply_partials(self.partials, self.packed_local)
            self._pending = self._all_gather_start()

    def normalize_end(self):
        """Second half: wait for the all-gather (the current stream waits, not the host) and hand the global
        population to the shard (resampling source)."""
        with self._on_stream():
            work = getattr(self, "_pending", None)
            if work is not None:
                work.wait()
            self._pending = None
            self.ops.import_global(self.packed_global)

    def normalize(self):
        self.normalize_begin()
        self.normalize_end()

    def scan_step(self, inputs, r01: float, fraction: Optional[float] = 0.5):
        """One scan (SLAM.update + conditional resample) with ONE exchange: every rank scores its shard and leaves its
        block partials and its RAW particles {w, x, y, theta} in its slots of the two gather buffers; both buffers are
        all-gathered (issued together); statistics, normalisation, map update and resample are local after that.
        `inputs` is whatever the shard ops need (HipShardOps: (dev_poses, dev_beams, B, integrate)).  Every rank
        passes the same scan and r01.  (libgridmapslam's gms_slam_update_sharded_dev is this with RCCL inside.)"""
        with self._on_stream():
            self.ops.exchange_begin(inputs)
            if self.world > 1 or self.force:
                pg, pl, tg, tl = self.ops.gather_views()
                works = []
                for out, mine in ((pg, pl), (tg, tl)):
                    try:
                        works.append(dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True))
                    except (RuntimeError, NotImplementedError, TypeError):
                        mine_c = mine.clone()             # list form (gloo): the input must not alias an output chunk
                        works.append(dist.all_gather(list(out.chunk(self.world)), mine_c, group=self.group, async_op=True))
                for wk in works:
                    wk.wait()
            self.ops.exchange_end(inputs, r01, fraction)

    def resample(self, r01: float, fraction: Optional[float] = None):
        """SLAM.resample (SLAM.java:133-153); with `fraction`, only if neff < fraction*N
        (J/app/GridMapApp.java:185-186).  Every rank must pass the same r01."""
        self.ops.resample(r01, fraction)

    def refresh_stats(self):
        """Weighted pose / Neff of the CURRENT (e.g. resampled) particles: one more all-reduce of the
        partial vector, nothing rewritten (getWeightedPose after resample, J/app/GridMapApp.java:192)."""
        with self._on_stream():
            self.ops.local_partials(self.partials)
            if self.world > 1:
                if self.coll is not None:
                    self.coll.all_reduce_sum(self.partials)
                else:
                    dist.all_reduce(self.partials, op=dist.ReduceOp.SUM, group=self.group)
            self.ops.stats_from_partials(self.partials)

    def stats(self) -> dict:
        return self.ops.stats()

    def weighted_pose(self) -> np.ndarray:
        return self.ops.weighted_pose()


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's own filter shape over several GPUs: particles WITH their maps (SLAM.java: one GridMapData per particle), no replica.
# ---------------------------------------------------------------------------------------------------------------------------------
class TorchCollectives:
    """the collectives of ShardedSlamParticleMaps over torch.distributed (backend "nccl" = RCCL on the GPUs of a node; gloo in the
    CPU tests).  Variable-size exchanges are all_to_all_single where the backend has it, pairwise isend / irecv otherwise."""

    def __init__(self, group=None, force: bool = False):
        """force: run the collectives even on a one-rank group (the RCCL calls themselves, where only one GPU exists)"""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.force = bool(force) and dist.is_initialized()
        # device tensors travel as they are over RCCL; any other backend (gloo: two test ranks on one GPU) gets host copies
        self.device_ok = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def _host(self, t: torch.Tensor) -> torch.Tensor:
        return t if (self.device_ok or not t.is_cuda) else t.cpu()

    def all_reduce_sum(self, t: torch.Tensor):
        if self.world > 1 or self.force:
            h = self._host(t)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            if h is not t:
                t.copy_(h)

    def all_gather_into(self, out: torch.Tensor, mine: torch.Tensor):
        if self.world == 1 and not self.force:
            out.copy_(mine)
            return
        ho, hm = self._host(out), self._host(mine)
        try:
            dist.all_gather_into_tensor(ho, hm, group=self.group)
        except (RuntimeError, NotImplementedError, TypeError):
            dist.all_gather(list(ho.chunk(self.world)), hm.clone(), group=self.group)
        if ho is not out:
            out.copy_(ho)

    def all_gather_host(self, a: np.ndarray) -> np.ndarray:
        """[world][len(a)] of a small host array (the resampling sources: 4 bytes per particle)"""
        if self.world == 1:
            return a[None].copy()
        t = torch.from_numpy(np.ascontiguousarray(a))
        if self.device_ok:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def exchange(self, send: list, recv_counts: list, rec: int, like: torch.Tensor) -> list:
        """send[q]: [k_q][rec] tensor for rank q (k_q may be 0); returns recv[q]: [recv_counts[q]][rec] from rank q"""
        recv = [like.new_empty((int(c), rec)) for c in recv_counts]
        if self.world == 1:
            return recv
        staged = not self.device_ok and like.is_cuda
        hrecv = [r.cpu() if staged else r for r in recv]
        ops = []
        for q in range(self.world):
            if q == self.rank:
                continue
            peer = dist.get_global_rank(self.group, q) if self.group is not None else q
            if send[q] is not None and send[q].numel():
                ops.append(dist.P2POp(dist.isend, self._host(send[q]).contiguous(), peer, self.group))
            if hrecv[q].numel():
                ops.append(dist.P2POp(dist.irecv, hrecv[q], peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if staged:
            for r, h in zip(recv, hrecv):
                r.copy_(h)
        return recv


def plan_map_exchange(all_sources: np.ndarray, rank: int, n_local: int):
    """What one rank sends and receives in a sharded resample().  all_sources [world][n_local]: the global source index of every slot
    of every rank (non-decreasing: systematic resampling keeps the order).  Returns (send_lists, recv_counts, src_local, recv_pos):
    send_lists[q] = this rank's local particle indices whose records rank q needs (ascending, each once); recv_counts[q] = records
    that arrive from rank q (in ascending order of their global index); src_local[m] = the local source of this rank's slot m or -1;
    recv_pos[m] = for a remote source, its position in the concatenation of the received buffers in rank order."""
    world = all_sources.shape[0]
    lo = rank * n_local
    send_lists, recv_counts = [], []
    mine = all_sources[rank]
    owner = mine // n_local
    src_local = np.where(owner == rank, mine - lo, -1).astype(np.int32)
    recv_pos = np.full(n_local, -1, dtype=np.int32)
    base = 0
    for q in range(world):
        if q == rank:
            send_lists.append(np.zeros(0, np.int32)); recv_counts.append(0)
            continue
        wanted_by_q = np.unique(all_sources[q][(all_sources[q] // n_local) == rank])        # ascending, each once
        send_lists.append((wanted_by_q - lo).astype(np.int32))
        from_q = np.unique(mine[owner == q])
        recv_counts.append(int(from_q.size))
        if from_q.size:
            sel = owner == q
            recv_pos[sel] = base + np.searchsorted(from_q, mine[sel]).astype(np.int32)
        base += int(from_q.size)
    return send_lists, recv_counts, src_local, recv_pos


class SlamShardOps:
    """One rank's block of the particles WITH their maps, through the C-ABI (gms_slam_create_shard): the shard-local half of
    SLAM.update / SLAM.resample plus the weight-exchange hooks ShardedParticleFilter drives.  No CPU path."""

    def __init__(self, width, height, resolution, position, n_local: int, offset: int, n_global: int, device: Optional[int] = None, max_beams: int = 0):
        import ctypes as C
        if not torch.cuda.is_available():
            raise RuntimeError("SlamShardOps needs a HIP device (there is no CPU path)")
        from .gridmap import SLAMParticleMaps
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.slam = SLAMParticleMaps.__new__(SLAMParticleMaps)
        self.slam._init_shard(width, height, resolution, position, n_local, offset, n_global, self.device.index, max_beams)
        self.pf = self.slam.pf
        self.n, self.offset, self.n_global = n_local, offset, n_global
        cur = torch.cuda.current_stream(self.device)
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device=self.device)
        self.slam.grid_map.set_stream(self.stream.cuda_stream)
        v = C.c_int64(0)
        _lib.check(_lib.load().gms_slam_record_doubles(self.slam._h, C.byref(v)))
        self.record_doubles = int(v.value)

    # -- the weight exchange (ShardedParticleFilter's ops interface)
    def new_buffer(self, n_doubles: int) -> torch.Tensor:
        return torch.zeros(n_doubles, dtype=torch.float64, device=self.device)

    def partials_len(self) -> int:
        return self.pf.partials_len()

    def local_partials(self, partials):
        self.pf.local_partials(partials.data_ptr())

    def apply_partials(self, partials, packed_local):
        self.pf.apply_partials(partials.data_ptr(), packed_local.data_ptr())

    def stats_from_partials(self, partials):
        self.pf.stats_from_partials(partials.data_ptr())

    def import_global(self, packed_global):
        self.pf.import_global(packed_global.data_ptr())

    def stats(self) -> dict:
        return self.pf.stats()

    def weighted_pose(self) -> np.ndarray:
        return self.pf.weighted_pose()

    # -- the shard-local halves
    def update_local(self, z, odometry, seed: int, sequence: int, sample_motion: bool = True):
        from .gridmap import _beams_of
        b = _beams_of(z)
        have = odometry is not None and sample_motion
        dc, dt = odometry if odometry is not None else (0.0, 0.0)
        _lib.check(_lib.load().gms_slam_update_local(self.slam._h, _lib.ptr(b), len(b), int(have), float(dc), float(dt), int(seed), int(sequence)))

    def draw(self, r01: float, fraction: Optional[float]):
        import ctypes as C
        did = C.c_int32(0)
        src = np.empty(self.n, dtype=np.int32)
        _lib.check(_lib.load().gms_slam_shard_draw(self.slam._h, float(r01), -1.0 if fraction is None else float(fraction), C.byref(did), _lib.ptr(src)))
        return bool(did.value), src

    def export(self, local_indices: np.ndarray) -> torch.Tensor:
        idx = np.ascontiguousarray(local_indices, dtype=np.int32)
        out = torch.empty((idx.size, self.record_doubles), dtype=torch.float64, device=self.device)
        if idx.size:
            import ctypes as C
            _lib.check(_lib.load().gms_slam_shard_export(self.slam._h, _lib.ptr(idx), int(idx.size), C.c_void_p(out.data_ptr())))
        return out

    def gather(self, src_local: np.ndarray, recv_pos: np.ndarray, recv: Optional[torch.Tensor]):
        import ctypes as C
        a, b = np.ascontiguousarray(src_local, dtype=np.int32), np.ascontiguousarray(recv_pos, dtype=np.int32)
        _lib.check(_lib.load().gms_slam_shard_gather(self.slam._h, _lib.ptr(a), _lib.ptr(b), C.c_void_p(recv.data_ptr()) if recv is not None and recv.numel() else None))

    def like(self) -> torch.Tensor:
        return torch.empty(0, dtype=torch.float64, device=self.device)

    def synchronize(self):
        self.slam.grid_map.synchronize()

    # -- the same two steps with the exchanges inside the library (RCCL): one C-ABI call each, no Python collective on the path
    def update_rccl(self, comm: "RcclComm", z, odometry, seed: int, sequence: int, sample_motion: bool = True) -> dict:
        """SLAM.update(z, u) over all ranks through gms_slam_update_sharded_maps; returns the statistics (every rank: the same)"""
        from .gridmap import _beams_of
        from ._lib import GmsPfStats
        import ctypes as C
        b = _beams_of(z)
        have = odometry is not None and sample_motion
        dc, dt = odometry if odometry is not None else (0.0, 0.0)
        st = GmsPfStats()
        _lib.check(_lib.load().gms_slam_update_sharded_maps(self.slam._h, comm._h, _lib.ptr(b), len(b), int(have), float(dc), float(dt), int(seed),
                                                            int(sequence), C.byref(st)))
        return {"weight_sum": st.weight_sum, "neff": st.neff, "strongest": st.strongest, "n_zero": st.n_zero}

    def resample_rccl(self, comm: "RcclComm", r01: float, fraction: Optional[float] = None) -> bool:
        """SLAM.resample() over all ranks through gms_slam_resample_sharded_maps; returns whether it drew"""
        import ctypes as C
        did = C.c_int32(0)
        _lib.check(_lib.load().gms_slam_resample_sharded_maps(self.slam._h, comm._h, float(r01), -1.0 if fraction is None else float(fraction), C.byref(did)))
        return bool(did.value)


class ShardedSlamParticleMaps:
    """SLAM (J/slam/SLAM.java) with one GridMapData per particle, the particles AND their maps split over the ranks: rank r holds the
    block [r n, (r + 1) n) and no replica of anything.  update(): every rank runs the per-particle body for its block, then the two
    small collectives of ShardedParticleFilter (block partials all-reduced, packed particles all-gathered): weightSum, strongest, Neff,
    weighted pose as on one GPU.  resample(): every rank draws its own slots from the gathered population; the ranks all-gather the
    slots' sources (4 bytes per particle) and exchange the RECORDS (logData + class planes) of exactly the particles that crossed a
    rank boundary -- systematic resampling keeps the particle order, so these are the blocks' edges unless the weights have collapsed --
    and every rank makes its copies from its own previous generation and the received records.  Poses, weights and maps equal the
    one-GPU filter's for any number of ranks (same block shape of every reduction, same draw, bit-identical copies).
    `ops`: SlamShardOps (HIP) -- or a stand-in with the same methods (the gloo tests); `coll`: TorchCollectives or a stand-in."""

    def __init__(self, n_global: int, ops, coll=None, group=None):
        self.coll = coll if coll is not None else TorchCollectives(group)
        self.ops = ops
        self.n_global = n_global
        self.world, self.rank = self.coll.world, self.coll.rank
        self.n_local, self.offset = ShardedParticleFilter.shard_of(n_global, self.world, self.rank)
        self.weights = ShardedParticleFilter(n_global, ops, group=group, coll=self.coll)
        self.records_sent = 0
        self.records_received = 0
        self.resamples = 0

    def update(self, z, odometry=None, seed: int = 0, sequence: int = 0, sample_motion: bool = True) -> float:
        """SLAM.update(z, u) (SLAM.java:80-131); returns Neff (every rank: the same)"""
        self.ops.update_local(z, odometry, seed, sequence, sample_motion)
        self.weights.normalize()
        return self.weights.stats()["neff"]

    def resample(self, r01: float, fraction: Optional[float] = None) -> bool:
        """SLAM.resample() (SLAM.java:133-153); with `fraction`: only if Neff < fraction * N (GridMapApp.java:185-186), decided alike on
        every rank from the same statistics.  Every rank passes the same r01.  Returns whether it drew."""
        did, src = self.ops.draw(r01, fraction)
        flags = self.coll.all_gather_host(np.array([int(did)], dtype=np.int32))
        assert (flags == int(did)).all(), "the ranks disagree on the resampling rule: their statistics differ"
        if not did:
            return False
        all_src = self.coll.all_gather_host(src)                                   # [world][n_local]
        send_lists, recv_counts, src_local, recv_pos = plan_map_exchange(all_src, self.rank, self.n_local)
        send = [self.ops.export(l) if l.size else None for l in send_lists]
        getattr(self.ops, "synchronize", lambda: None)()                           # the records are complete before they travel
        recv = self.coll.exchange(send, recv_counts, self.ops.record_doubles, self.ops.like())
        buf = torch.cat([r for r in recv if r.numel()]) if any(r.numel() for r in recv) else None
        self.ops.gather(src_local, recv_pos, buf)
        getattr(self.ops, "synchronize", lambda: None)()                           # (buf may be freed on return)
        self.records_sent += int(sum(l.size for l in send_lists))
        self.records_received += int(sum(recv_counts))
        self.resamples += 1
        return True

    def stats(self) -> dict:
        return self.weights.stats()

    def weighted_pose(self) -> np.ndarray:
        return self.weights.weighted_pose()
