"""Builds libgridmapslam.so (the HIP C-ABI library) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgridmapslam.so")
# gms_fused_kernels.hip is the device translation unit: it includes gms_map_kernels.hip and gms_pf_kernels.hip
SOURCES = ["gms_host.hip", "gms_fused_kernels.hip"]
HEADERS = [os.path.join(CSRC, "gms_internal.h"), os.path.join(CSRC, "gms_device.h"),
           os.path.join(CSRC, "gms_map_kernels.hip"), os.path.join(CSRC, "gms_pf_kernels.hip"),
           os.path.join(ROOT, "include", "gridmapslam.h")]
# -ffp-contract=off: the reference (JVM) never fuses a multiply with an add; parity depends on it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("GMS_EXTRA_FLAGS", "").split()      # experiments only (e.g. -DRC_RAYS=4)
    cmd = [hipcc()] + FLAGS + extra + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
