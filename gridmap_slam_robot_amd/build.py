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
# gms_fused_kernels.hip is the device translation unit: it includes gms_map_kernels.hip, gms_pf_kernels.hip and gms_slam_kernels.hip
SOURCES = ["gms_host.hip", "gms_slam_host.hip", "gms_fused_kernels.hip"]
# -ffp-contract=off: the reference (JVM) never fuses a multiply with an add; parity depends on it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def extra_flags() -> list:
    return os.environ.get("GMS_EXTRA_FLAGS", "").split()      # experiments only (e.g. -DRC_RAYS=4, -DGMS_STAMPS)


def source_hash() -> str:
    """sha256 (16 hex digits) over the library's sources -- csrc/* and the public header, in name order -- and, when
    GMS_EXTRA_FLAGS is set, over those flags as well: an instrumented or experimental build carries another hash than the
    product build of the same sources, so neither is ever taken for the other (gms_build_info(), the up-to-date check)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(ROOT, "include", "gridmapslam.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    extra = extra_flags()
    if extra:
        h.update(b"\0flags\0" + " ".join(extra).encode())
    return h.hexdigest()[:16]


def built_hash(lib: str = LIB) -> str | None:
    """the source hash a built library carries (the string gms_build_info() returns), read from the file itself"""
    try:
        blob = open(lib, "rb").read()
    except OSError:
        return None
    i = blob.find(b"GMS_SOURCE_HASH=")
    return blob[i + 16:i + 32].decode("ascii", "replace") if i >= 0 else None


def needs_build() -> bool:
    return built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles the library unless the in-tree one was built from exactly these sources; says which on stderr, with the hash
    (the same string gms_build_info() returns at run time, so a log shows which binary ran)."""
    want = source_hash()
    if not force and built_hash() == want:
        print(f"libgridmapslam: reused {want} (in-tree build of these sources)", file=sys.stderr)
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = extra_flags()
    cmd = [hipcc()] + FLAGS + extra + [f'-DGMS_SOURCE_HASH="{want}"', "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    print(f"libgridmapslam: built {want}" + (f" (extra flags: {' '.join(extra)})" if extra else ""), file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
