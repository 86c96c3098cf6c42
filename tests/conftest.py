import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_probe():
    """(device count, reason when 0).  The library is the product: a missing or unloadable .so is reported as such."""
    try:
        from gridmap_slam_robot_amd import _lib
        L = _lib.load()
    except Exception as e:                      # missing .so, unresolved symbol, wrong arch ...
        return 0, f"libgridmapslam.so did not load: {e!r}"
    try:
        n = int(L.gms_device_count())
    except Exception as e:
        return 0, f"gms_device_count failed: {e!r}"
    return n, ("no HIP device visible" if n <= 0 else "")


def _gpu_run_requested(config) -> bool:
    """`-m gpu` (or any mark expression that selects gpu tests), or GMS_REQUIRE_GPU=1."""
    if os.environ.get("GMS_REQUIRE_GPU", "") not in ("", "0"):
        return True
    expr = (getattr(config.option, "markexpr", "") or "").replace("(", " ").replace(")", " ")
    toks = expr.split()
    return any(t == "gpu" and (i == 0 or toks[i - 1] != "not") for i, t in enumerate(toks))


@pytest.fixture(scope="session")
def have_gpu():
    return _gpu_probe()[0] > 0


def pytest_collection_modifyitems(config, items):
    n, why = _gpu_probe()
    if n > 0:
        return
    if _gpu_run_requested(config):
        # The GPU tests were asked for and cannot run: that is a failure of the run, not a set of skips -- a broken .so
        # on the GPU box must not come back as rc 0 with everything skipped.
        pytest.exit(f"-m gpu selected but the GPU tests cannot run: {why}", returncode=3)
    skip = pytest.mark.skip(reason=f"{why} (GPU parity tests run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Resample agreement as numbers ("indices equal for fixed r"; the device's blocked scan may pick a neighbour
    where U lies within rounding distance of a boundary): every checked resampling step's slots, differing slots and flagged
    slots, summed per test, printed and -- on the GPU box -- left in gpurun_out/resample_agreement.json."""
    try:
        import _checks
    except Exception:
        return
    if not _checks.AGREEMENT:
        return
    per = {}
    for r in _checks.AGREEMENT:
        e = per.setdefault(r["where"], {"steps": 0, "slots": 0, "slots_differing": 0, "n_ambiguous": 0, "largest_population": 0})
        e["steps"] += 1
        e["slots"] += r["slots"]
        e["slots_differing"] += r["slots_differing"]
        e["n_ambiguous"] += r["n_ambiguous"]
        e["largest_population"] = max(e["largest_population"], r["slots"])
    tot = {k: sum(e[k] for e in per.values()) for k in ("steps", "slots", "slots_differing", "n_ambiguous")}
    tr = session.config.pluginmanager.get_plugin("terminalreporter")
    line = (f"resample agreement with the sequential oracle: {tot['slots_differing']} of {tot['slots']} slots differ over {tot['steps']} "
            f"resampling steps ({tot['n_ambiguous']} slots flagged ambiguous by the device)")
    if tr is not None:
        tr.write_line(line)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "resample_agreement.json"), "w") as f:
            json.dump({"total": tot, "per_test": per}, f, indent=1)
