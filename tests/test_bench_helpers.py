"""bench.py's host-side arithmetic (no GPU): the algorithmic bytes per launch follow SURVEY.md section 8(d), the PMC traffic
lookup reads the committed per-configuration summaries, and the committed bench lines carry the contract's fields."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_the_survey():
    b = _bench()
    # C3 scoring: 8 B per beam-eval + 20 B per particle + 17 B per beam (94.4 MB + 0.33 MB + 12 KB at 720 hits)
    assert b.algorithmic_bytes("score", n_particles=16384, n_hit=720, n_beams=720) == 8 * 16384 * 720 + 20 * 16384 + 17 * 720
    assert b.algorithmic_bytes("likelihood", cells=2048 * 2048, full_rebuild=True, paired=False) == 16 * 2048 * 2048      # 64 MiB
    assert b.algorithmic_bytes("raycast", visits=1000, n_particles=256, paired=False) == 16 * 1000
    assert b.algorithmic_bytes("raycast", visits=1000, n_particles=256, paired=True) == 16 * 1000 + 16 * 256 + 16 * 1000     # + normalise + previous apply
    assert b.algorithmic_bytes("reduce", visits=1000, n_particles=256, paired=True) == 16 * 256
    assert b.algorithmic_bytes("likelihood", dirty_cells=4096, n_particles=256, paired=True) == 16 * 4096 + 32 * 256     # + resample
    assert b.algorithmic_bytes("score", n_particles=4096, n_hit=1000, n_beams=1080, n_maps=64) == 64 * (8 * 4096 * 1000 + 20 * 4096 + 17 * 1080)


def test_pmc_traffic_reads_the_committed_summaries():
    b = _bench()
    files = glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic*.json"))
    assert files, "profiles/*/pmc_traffic.json must be committed"
    t = b.pmc_traffic("C3", "score")
    assert t is not None and 1e6 < t < 92e6          # fabric-side bytes of the scoring launch: a fraction of the 91.7 MB algorithmic
    assert b.pmc_traffic("C5", "raycast") is not None
    assert b.pmc_traffic("C3", "no-such-class") is None and b.pmc_traffic("C9", "score") is None


def test_committed_bench_lines_carry_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02", "bench*.json")))
    assert lines
    for f in lines:
        d = json.load(open(f))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in d, (f, k)
        assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (f, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    main = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench.json")))
    cb = main["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "map_update_ms_per_scan"):
        assert k in cb
    assert cb["kind"] == "port" and cb["cores"] == 1
    assert set(main["secondary"]) == {"C5", "C2"}


def _canned_full_report():
    """the full report of a default 1-GPU run, as large as round 3's line was (six secondary blocks)"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r03", "bench.json")))
    assert len(json.dumps(full)) > 20000           # the 22 KB line the driver could not take
    return full


def test_the_stdout_line_is_compact_strict_json_with_the_contract_fields():
    b = _bench()
    full = _canned_full_report()
    full["secondary"]["bad"] = {"error": "x"}
    full["roofline"]["lookup_ceiling_frac"] = 0.72
    full["filter"]["neff"] = float("nan")           # NaN may not reach the line (strict JSON)
    text = b.compact_line(b._json_safe(full), "bench_report.json")
    assert "\n" not in text and len(text.encode()) < b.LINE_LIMIT <= 4096 < 8192
    d = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))        # NaN / Infinity would raise
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "report"):
        assert k in d, k
    assert d["report"] == "bench_report.json" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "step"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    assert r["lookup_ceiling_frac"] == 0.72 and "frac" in r["step"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "seconds"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1
    assert d["secondary_ms_per_step"]["bad"] is None and d["secondary_ms_per_step"]["C5"] > 0
    # value and ms_per_step survive the rounding to six significant figures
    assert abs(d["value"] - full["value"]) < 1e-5 * full["value"] and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-5 * full["ms_per_step"]


def test_an_oversized_line_sheds_its_optional_blocks_not_the_contract(tmp_path):
    b = _bench()
    full = _canned_full_report()
    full["secondary"] = {f"run{i}_{'x' * 40}": {"ms_per_step": 0.1 * i} for i in range(200)}
    text = b.compact_line(b._json_safe(full), "r.json")
    d = json.loads(text)
    assert len(text) <= b.LINE_LIMIT and "secondary_ms_per_step" not in d and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    # emit(): the report file holds everything, the line names it
    rd, wr = os.pipe()
    rep = tmp_path / "sub" / "report.json"
    b.emit(full, wr, str(rep))
    os.close(wr)
    line = os.read(rd, 1 << 16).decode()
    os.close(rd)
    assert line.endswith("\n") and line.count("\n") == 1
    assert len(json.load(open(rep))["secondary"]) == 200 and json.loads(line)["report"].endswith("report.json")


def test_plain_gpus_n_launches_the_ranks_itself(tmp_path):
    """`python bench.py --gpus N` with no launcher around it (the form the driver uses for its 1-GPU line) must not exit with rc 2:
    it starts N ranks as a child `python -m torch.distributed.run ... bench.py <same arguments>` before touching the GPU.  Here the
    dry run: the command it would start (GMS_BENCH_LAUNCH_DRYRUN=1; the real thing runs in tests/test_gpu_bench_two_ranks.py)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, GMS_BENCH_LAUNCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"], capture_output=True,
                         text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # under a launcher (WORLD_SIZE set) nothing is re-launched: the mismatch is reported instead
    env2 = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", GMS_BENCH_LAUNCH_DRYRUN="1")
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('return self_launch(args.gpus, sys.argv[1:])') < src.index("import torch\n    import torch.distributed as dist")
