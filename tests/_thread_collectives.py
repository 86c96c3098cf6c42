"""Collectives among the THREADS of one process (tests): several shards of a sharded filter run as threads on one GPU (or on the CPU),
device copies standing in for the transfers.  Same interface as gridmap_slam_robot_amd.distributed.TorchCollectives."""
import threading

import numpy as np
import torch


class ThreadWorld:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank: int) -> "ThreadCollectives":
        return ThreadCollectives(self, rank)


class ThreadCollectives:
    def __init__(self, tw: ThreadWorld, rank: int):
        self.tw, self.rank, self.world = tw, rank, tw.world

    def _swap(self, mine):
        """every rank deposits an object; returns the list of all of them"""
        self.tw.slots[self.rank] = mine
        self.tw.barrier.wait()
        out = list(self.tw.slots)
        self.tw.barrier.wait()
        return out

    def _sync(self, t):
        if isinstance(t, torch.Tensor) and t.is_cuda:
            torch.cuda.synchronize(t.device)

    def all_reduce_sum(self, t: torch.Tensor):
        self._sync(t)
        parts = self._swap(t.clone())
        acc = parts[0].clone()
        for p in parts[1:]:
            acc += p                          # rank order, as a ring would not guarantee -- adding zeros is exact (distributed.py)
        t.copy_(acc)
        self._sync(t)
        self.tw.barrier.wait()

    def all_gather_into(self, out: torch.Tensor, mine: torch.Tensor):
        self._sync(mine)
        parts = self._swap(mine.clone())
        out.copy_(torch.cat(parts))
        self._sync(out)
        self.tw.barrier.wait()

    def all_gather_host(self, a: np.ndarray) -> np.ndarray:
        return np.stack(self._swap(np.array(a, copy=True)))

    def exchange(self, send, recv_counts, rec, like):
        for s in send:
            self._sync(s)
        boxes = self._swap(send)                                   # boxes[q][r]: what q sends to r
        recv = []
        for q in range(self.world):
            t = boxes[q][self.rank] if q != self.rank else None
            if t is None or t.numel() == 0:
                assert recv_counts[q] == 0
                recv.append(like.new_empty((0, rec)))
            else:
                assert t.shape == (recv_counts[q], rec), (t.shape, recv_counts[q], rec)
                recv.append(t.clone())                             # the device copy that stands for the transfer
        for r in recv:
            self._sync(r)
        self.tw.barrier.wait()
        return recv
