"""Builds jni/gms_jni.c + tests/jni_stub/jni_env.c into tests/jni_stub/_build/libgms_jni_stub.so with gcc, against the tests-local jni.h and
the C-ABI library: the JNI shim as a loadable object whose natives can be CALLED without a JVM (test infrastructure)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libgms_jni_stub.so")


def build() -> str:
    from gridmap_slam_robot_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    srcs = [os.path.join(ROOT, "jni", "gms_jni.c"), os.path.join(HERE, "jni_env.c")]
    newest = max(os.path.getmtime(f) for f in srcs + [os.path.join(HERE, "jni.h"), os.path.join(ROOT, "include", "gridmapslam.h"), _lib.LIB_PATH])
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D_GNU_SOURCE",
           "-I", HERE, "-I", os.path.join(ROOT, "include")] + srcs + ["-o", OUT, "-L", libdir, "-lgridmapslam", f"-Wl,-rpath,{libdir}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return OUT
