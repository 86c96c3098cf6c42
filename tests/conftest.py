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
