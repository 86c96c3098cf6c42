import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count() -> int:
    try:
        from gridmap_slam_robot_amd import _lib
        return _lib.load().gms_device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def have_gpu():
    return _gpu_count() > 0


def pytest_collection_modifyitems(config, items):
    # a GPU test on a box without a device is an environment error, not a pass: skip loudly
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (GPU parity tests run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
