"""RCCL with more than one rank: runs only where gms_device_count() >= 2 (the 1-GPU boxes of this pool skip it; a multi-GPU
node runs it as part of `pytest -m gpu`).  bench.py launched exactly as the driver launches it -- torch.distributed.run, one rank
per GPU, backend nccl (= RCCL) -- with its default configuration at N > 1, BASELINE.json configs[3]: 65 536 particles split over
the ranks, one grouped all-gather per scan inside the library.  The run verifies itself against a stand-alone filter of the
whole population before it times anything; here that verdict, the route and the communicator's size are asserted."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    from gridmap_slam_robot_amd import _lib
    return int(_lib.load().gms_device_count())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_over_rccl_shards_the_fixed_population_and_equals_the_standalone_filter(world, tmp_path):
    n = _n_devices()
    if n < world:
        pytest.skip(f"{n} HIP device(s) visible: a {world}-rank RCCL communicator needs {world} (one rank per device)")
    rep = str(tmp_path / "report.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GMS_BENCH_DIST_BACKEND", None)
    env.pop("GMS_BENCH_SHARE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "3",
           "--no-secondary", "--report", rep]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, out.stdout[-2000:]
    d = json.loads(lines[0])
    full = json.load(open(rep))
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["particles_total"] == 65536 and d["config"]["particles_per_gpu"] == 65536 // world
    assert d["rccl_ranks"] == world, full.get("verify")                               # the in-library communicator spans every rank
    assert d["config"]["exchange"].startswith("in-library RCCL")
    assert d["sharded_equals_standalone"] is True, full.get("verify")
    assert full["verify"]["population"] == 65536 and full["verify"]["mismatches"] is None
    assert len(d["per_rank_ms_per_step"]) == world and d["value"] > 0
    assert d.get("exchange_latency_us") is not None and d["exchange_latency_us"] > 0      # the grouped all-gather, event-bracketed
