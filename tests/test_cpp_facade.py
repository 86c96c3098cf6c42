"""include/gridmapslam.hpp (the C++ mirror of the reference's Java classes) compiles everywhere and, on a
GPU box, drives a SLAM step through the C-ABI from a plain g++ program."""
import os
import subprocess

import pytest

from gridmap_slam_robot_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp")
LIBDIR = os.path.dirname(_lib.LIB_PATH)


def _build(tmp_path):
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lgridmapslam", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_facade_links_against_the_c_abi(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)


def test_facade_reports_missing_gpu(tmp_path, have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 2 and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_facade_runs_a_slam_step(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ")
