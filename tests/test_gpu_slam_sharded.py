"""The reference-shape filter over several ranks -- particles WITH their maps, no replica (gridmap_slam_robot_amd.distributed.
ShardedSlamParticleMaps over gms_slam_create_shard): 2 and 4 shards run as threads of this process on ONE GPU, device copies standing
in for the transfers (the pool has one GPU per box), against the stand-alone gms_slam of the same population over ten frames of a
drive with the caller's resampling rule: every pose, every weight, Neff, the weighted pose and EVERY MAP (logData and likelihoodData)
equal, bit for bit, whatever the number of shards; and what crossed a shard boundary at each resampling step is counted."""
import os
import threading

import numpy as np
import pytest
import torch

from gridmap_slam_robot_amd import SLAMParticleMaps, synth
from gridmap_slam_robot_amd._lib import GMS_BLOCK, GMS_ERR_STATE, GmsError
from gridmap_slam_robot_amd.distributed import ShardedSlamParticleMaps, SlamShardOps, plan_map_exchange
from oracle import oracle as orc

from _thread_collectives import ThreadWorld

pytestmark = pytest.mark.gpu


def _scans(ext, B, T):
    frames, _ = synth.make_recording(ext, B, T=48, seed=77, n_frames=T)
    start = synth.true_pose(synth.make_world(ext, 77), -1, 48)
    return [(orc.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta), (f.d_center, f.d_theta)) for f in frames], start


def _stand_alone(ext, res, N, scans, start, r01s, fractions, refine=False):
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    dev.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
    dev.set_refine(refine)                            # (SLAM.java:96: the motion draw then happens in the refinement launch, keyed by the global index alike)
    out = []
    for k, (z, u) in enumerate(scans):
        neff = dev.update(z, u, seed=5, sequence=k)
        did = neff < fractions[k] * N
        P, w = dev.get_particles()
        rec = {"neff": neff, "did": did, "wpose": dev.get_weighted_pose().copy(), "strongest": dev.strongest, "poses_before": P, "weights_before": w}
        if did:
            dev.resample(r01s[k])
        rec["poses"], rec["weights"] = dev.get_particles()
        out.append(rec)
    logs, liks = dev.maps().copy(), dev.maps(likelihood=True).copy()
    dev.close()
    return out, logs, liks


@pytest.mark.parametrize("world,refine", [(2, False), (4, False), (2, True)])
def test_shards_on_one_gpu_equal_the_stand_alone_filter(world, refine):
    ext, res, B, N, T = 6.0, 0.05, 90, 4 * GMS_BLOCK, (10 if not refine else 5)
    # (the plain product's Neff collapses on every frame: every third frame runs the rule with a threshold nothing falls below)
    fractions = [1e-9 if k % 3 == 2 else 0.5 for k in range(T)]
    scans, start = _scans(ext, B, T)
    r01s = np.random.default_rng(9).random(T)
    want, want_logs, want_liks = _stand_alone(ext, res, N, scans, start, r01s, fractions, refine)
    assert any(r["did"] for r in want) and not all(r["did"] for r in want), "the drive must exercise both sides of the resampling rule"
    n = N // world
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                ops = SlamShardOps(ext, ext, res, (-ext / 2, -ext / 2), n, rank * n, N, max_beams=128)
                ops.slam.set_poses(np.tile(np.asarray(start, np.float32), (n, 1)))
                ops.slam.set_refine(refine)
                f = ShardedSlamParticleMaps(N, ops, coll=tw.comm(rank))
                per_frame = []
                for k, (z, u) in enumerate(scans):
                    neff = f.update(z, u, seed=5, sequence=k)
                    st = dict(f.stats())
                    P, w = ops.slam.get_particles()
                    rec = {"neff": neff, "wpose": f.weighted_pose().copy(), "strongest": st["strongest"], "poses_before": P, "weights_before": w}
                    rec["did"] = f.resample(float(r01s[k]), fractions[k])
                    rec["poses"], rec["weights"] = ops.slam.get_particles()
                    per_frame.append(rec)
                results[rank] = dict(frames=per_frame, logs=ops.slam.maps().copy(), liks=ops.slam.maps(likelihood=True).copy(),
                                     sent=f.records_sent, received=f.records_received, resamples=f.resamples)
                # a shard refuses the stand-alone entry points
                with pytest.raises(GmsError) as e:
                    ops.slam.resample(0.5)
                assert e.value.code == GMS_ERR_STATE
                ops.slam.close()
        except BaseException as e:      # noqa: BLE001 -- a failing rank must not leave the others waiting at a barrier
            errors.append((rank, e))
            tw.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    moved = 0
    for r in range(world):
        sl = slice(r * n, (r + 1) * n)
        for k in range(T):
            got, ref = results[r]["frames"][k], want[k]
            assert got["neff"] == ref["neff"] and got["strongest"] == ref["strongest"] and np.array_equal(got["wpose"], ref["wpose"]), (r, k)
            assert np.array_equal(got["poses_before"], ref["poses_before"][sl]) and np.array_equal(got["weights_before"], ref["weights_before"][sl]), (r, k)
            assert got["did"] == ref["did"], (r, k)
            assert np.array_equal(got["poses"], ref["poses"][sl]) and np.array_equal(got["weights"], ref["weights"][sl]), (r, k)
        assert np.array_equal(results[r]["logs"], want_logs[sl]), f"rank {r}: logData differs"
        assert np.array_equal(results[r]["liks"], want_liks[sl]), f"rank {r}: likelihoodData differs"
        moved += results[r]["received"]
    assert sum(x["sent"] for x in results) == sum(x["received"] for x in results)
    assert moved > 0, "at least one map must have crossed a shard boundary"
    n_res = results[0]["resamples"]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, f"slam_sharded_{world}{'_refine' if refine else ''}.json"), "w") as fh:
            json.dump({"what": "maps that crossed a shard boundary in resample(), shards as threads on one GPU", "shards": world, "particles": N,
                       "frames": T, "resampling_steps": n_res, "records_moved": moved, "records_moved_per_step": moved / max(1, n_res),
                       "fraction_of_maps_moved": moved / max(1, n_res) / N}, fh, indent=1)


def test_the_exchange_plan():
    """plan_map_exchange: every remote source is sent once per destination rank, positions index the concatenated receive buffer"""
    rng = np.random.default_rng(1)
    for world, n in ((2, 8), (4, 16), (3, 5)):
        N = world * n
        src = np.sort(rng.integers(0, N, N)).reshape(world, n)                # non-decreasing, as a systematic draw
        plans = [plan_map_exchange(src, r, n) for r in range(world)]
        for r in range(world):
            send, counts, src_local, pos = plans[r]
            assert send[r].size == 0 and counts[r] == 0
            for q in range(world):
                if q != r:
                    assert counts[q] == plans[q][0][r].size                    # what q sends to r is what r expects from q
            # replay: build the receive buffer from the senders' lists and check every slot gets its source
            recv = np.concatenate([plans[q][0][r] + q * n for q in range(world)]) if world > 1 else np.zeros(0, int)
            for m in range(n):
                g = src[r][m]
                assert (src_local[m] + r * n == g) if src_local[m] >= 0 else (recv[pos[m]] == g)


def test_one_rank_over_real_rccl_collectives():
    """The pool has one GPU per box: the torch.distributed route of the sharded filter with backend nccl (= RCCL) and a group of ONE
    rank, the collectives forced to run (TorchCollectives(force=True)) -- RCCL's all-reduce and all-gather on the library's own
    buffers and stream, the plan and the gather kernel with no peer -- against the stand-alone handle."""
    import socket
    import torch.distributed as dist
    from gridmap_slam_robot_amd.distributed import TorchCollectives
    ext, res, B, N, T = 4.0, 0.05, 72, 2 * GMS_BLOCK, 5
    scans, start = _scans(ext, B, T)
    r01s = np.random.default_rng(2).random(T)
    fractions = [0.5] * T
    want, want_logs, want_liks = _stand_alone(ext, res, N, scans, start, r01s, fractions)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ops = SlamShardOps(ext, ext, res, (-ext / 2, -ext / 2), N, 0, N, max_beams=128)
        ops.slam.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
        f = ShardedSlamParticleMaps(N, ops, coll=TorchCollectives(force=True))
        for k, (z, u) in enumerate(scans):
            neff = f.update(z, u, seed=5, sequence=k)
            assert neff == want[k]["neff"] and np.array_equal(f.weighted_pose(), want[k]["wpose"]), k
            assert f.resample(float(r01s[k]), fractions[k]) == want[k]["did"]
            P, w = ops.slam.get_particles()
            assert np.array_equal(P, want[k]["poses"]) and np.array_equal(w, want[k]["weights"]), k
        assert np.array_equal(ops.slam.maps(), want_logs) and np.array_equal(ops.slam.maps(likelihood=True), want_liks)
        ops.slam.close()
    finally:
        dist.destroy_process_group()


def test_one_rank_through_the_library_route():
    """gms_slam_update_sharded_maps / gms_slam_resample_sharded_maps: the exchanges inside the library (RCCL all-reduce and all-gathers on a
    communicator of ONE rank -- what one GPU can run; the grouped ncclSend / ncclRecv of the records has no peer here and has never
    executed) against the stand-alone handle, and a shard whose block is not the whole population refused by the one-rank communicator."""
    from gridmap_slam_robot_amd.distributed import RcclComm
    ext, res, B, N, T = 4.0, 0.05, 72, 2 * GMS_BLOCK, 5
    scans, start = _scans(ext, B, T)
    r01s = np.random.default_rng(2).random(T)
    fractions = [0.5, 1e-9, 0.5, 0.5, 1e-9]
    want, want_logs, want_liks = _stand_alone(ext, res, N, scans, start, r01s, fractions)
    torch.cuda.set_device(0)
    comm = RcclComm()
    assert comm.world == 1
    ops = SlamShardOps(ext, ext, res, (-ext / 2, -ext / 2), N, 0, N, max_beams=128)
    ops.slam.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
    for k, (z, u) in enumerate(scans):
        st = ops.update_rccl(comm, z, u, seed=5, sequence=k)
        assert st["neff"] == want[k]["neff"] and st["strongest"] == want[k]["strongest"], k
        assert ops.resample_rccl(comm, float(r01s[k]), fractions[k]) == want[k]["did"]
        P, w = ops.slam.get_particles()
        assert np.array_equal(P, want[k]["poses"]) and np.array_equal(w, want[k]["weights"]), k
    assert np.array_equal(ops.slam.maps(), want_logs) and np.array_equal(ops.slam.maps(likelihood=True), want_liks)
    half = SlamShardOps(ext, ext, res, (-ext / 2, -ext / 2), GMS_BLOCK, 0, N, max_beams=128)
    with pytest.raises(GmsError):
        half.update_rccl(comm, scans[0][0], scans[0][1], seed=5, sequence=0)      # half the population on a communicator of one rank
    half.slam.close(); ops.slam.close(); comm.close()
