"""bench.py's N > 1 control flow on ONE GPU: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run ... bench.py --gpus 2), sharing device 0, collectives over gloo (test hooks
GMS_BENCH_DIST_BACKEND / GMS_BENCH_SHARE_DEVICE).  RCCL refuses two ranks on one device, so the in-library route must FAIL
on both ranks, every rank must fall back together to the torch.distributed exchange, the self-verification must run
(sharded == stand-alone on rank 0, bit for bit) and rank 0 must print one well-formed JSON line.  The first real RCCL
world > 1 run happens on the driver's multi-GPU node; this is everything around it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line_and_report(out, report):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["report"] and os.path.samefile(os.path.join(ROOT, d["report"]), report)
    return d, json.load(open(report))


def _run(cmd_tail, timeout):
    """torch.distributed.run on a port that was free a moment ago; the rendezvous can still lose the port to another process
    (it then waits out its own ten-minute time-out): one more attempt on a fresh port before the test gives up."""
    env = dict(os.environ, GMS_BENCH_DIST_BACKEND="gloo", GMS_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = None
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + cmd_tail
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        except subprocess.TimeoutExpired as e:
            print(f"attempt {attempt}: timed out after {timeout} s\n{(e.stderr or b'')[-2000:]}", file=sys.stderr)
            continue
        if out.returncode == 0:
            return out
        print(f"attempt {attempt}: exit code {out.returncode}\n{out.stderr[-3000:]}", file=sys.stderr)
    assert out is not None and out.returncode == 0, "bench.py --gpus 2 failed twice" + ("" if out is None else out.stderr[-3000:])
    return out


def test_bench_with_two_ranks_on_one_gpu_falls_back_verifies_and_reports(tmp_path):
    rep = str(tmp_path / "report.json")
    out = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "C2", "--particles", "1024", "--report", rep], 240)
    d, full = _line_and_report(out, rep)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["particles_total"] == 2048 and "sharded x2" in d["config"]["parallelism"]
    assert d["config"]["exchange"].startswith("torch.distributed")          # the in-library route cannot run here
    assert d["sharded_equals_standalone"] is True, full.get("verify")
    assert full["verify"]["population"] == 2048 and full["verify"]["mismatches"] is None
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert d["value"] > 0 and d["roofline"]["launches_timed"] >= 1 and d["cpu_baseline"] is None


def test_bench_default_configuration_with_two_ranks_is_the_fixed_population_with_the_weak_series_beside_it(tmp_path):
    """No --config at N > 1 = BASELINE configs[3]: ONE population split over the ranks ("strong"); the weak series rides along.
    (--particles would switch the companion run off, so the sizes here are the real ones: 2 x 32768 and 2 x 16384 on one GPU.)"""
    rep = str(tmp_path / "report.json")
    out = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--report", rep], 420)
    d, full = _line_and_report(out, rep)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["particles_total"] == 65536 and d["config"]["particles_per_gpu"] == 32768
    assert d["config"]["workload"].startswith("C4")
    assert d["sharded_equals_standalone"] is True, full.get("verify")
    weak = full["secondary"]["weak"]
    assert weak["scaling"] == "weak" and weak["config"]["particles_total"] == 32768 and weak["config"]["particles_per_gpu"] == 16384
    assert weak["sharded_equals_standalone"] is True and weak["value"] > 0
    assert d["secondary_ms_per_step"]["weak"] == pytest.approx(weak["ms_per_step"], rel=1e-4)


def test_bench_config5_with_two_ranks_shards_by_map_without_a_collective(tmp_path):
    """Config 5 (64 independent maps) at N > 1: the job's 64 maps are SPLIT over the ranks -- 32 per rank here, 8 per GPU on an
    8-GPU node -- nothing is exchanged on the data path, the job is fixed ("strong"); the line carries the aggregate
    over both ranks and each rank's own time."""
    rep = str(tmp_path / "report.json")
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "C5", "--particles", "256", "--report", rep], 300)
    d, full = _line_and_report(out, rep)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["maps"] == 32 and d["config"]["maps_total"] == 64 and d["config"]["particles_total"] == 64 * 256
    assert "no collective" in d["config"]["parallelism"] and d["config"].get("exchange") is None and full["config"]["exchange"] is None
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    # value = all ranks' particles x steps / the slowest rank's time
    assert abs(d["value"] - 64 * 256 * 3 / (max(d["per_rank_ms_per_step"]) * 1e-3 * 3)) <= 1e-3 * d["value"]      # (per-rank times are printed to five decimals)
    assert d.get("sharded_equals_standalone") is None          # nothing is sharded: there is nothing to verify against


def test_plain_python_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts the two ranks itself (a child torch.distributed.run,
    before this process touches the GPU) and hands rank 0's line and the exit code through."""
    rep = str(tmp_path / "report.json")
    env = dict(os.environ, GMS_BENCH_DIST_BACKEND="gloo", GMS_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--config", "C2",
                          "--particles", "1024", "--report", rep], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d, full = _line_and_report(out, rep)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["config"]["particles_total"] == 2048
    assert d["sharded_equals_standalone"] is True and len(d["per_rank_ms_per_step"]) == 2
    assert "rccl_ranks" in d or d["config"]["exchange"].startswith("torch.distributed")     # (gloo here: no RCCL communicator spans two ranks on one GPU)


def test_bench_particle_maps_sharded_over_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2 --particle-maps N,EXT,B`: the reference-shape filter sharded -- particles with their maps, no replica -- as two
    ranks sharing device 0 with gloo collectives (records staged through the host): one well-formed line from rank 0, the population
    fixed (strong scaling), maps crossing the rank boundary counted."""
    rep = str(tmp_path / "report.json")
    out = _run(["--gpus", "2", "--particle-maps", "512,4,72", "--steps", "4", "--report", rep], 240)
    d, full = _line_and_report(out, rep)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    pm = full["per_particle_maps_sharded"]
    assert pm["particles"] == 512 and pm["particles_per_rank"] == 256 and pm["ranks"] == 2
    assert pm["update_ms"] > 0 and pm["resampling_steps_timed"] >= 1 and pm["records_moved_per_resample"] > 0
