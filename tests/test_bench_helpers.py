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
