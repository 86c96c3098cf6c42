"""The JNI shim cannot be built in this image (no JDK, no jni.h).  What CAN be checked here: that jni/gms_jni.c is valid C
against the C-ABI header and the JNI signatures it uses -- `gcc -fsyntax-only` with the minimal tests/jni_stub/jni.h, a
SYNTAX check only (nothing is linked or run) -- and that every native the Java facade declares has its C definition and
vice versa."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jni_shim_is_valid_c_against_the_c_abi():
    cmd = ["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "gms_jni.c")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_java_natives_and_c_definitions_match():
    java = open(os.path.join(ROOT, "jni", "java", "com", "fmsz", "gridmapgl", "slam", "NativeSlam.java")).read()
    c = open(os.path.join(ROOT, "jni", "gms_jni.c")).read()
    declared = set(re.findall(r"static native \S+ (\w+)\(", java))
    defined = set(re.findall(r"CLS\((\w+)\)\(JNIEnv", c))
    assert declared == defined, (sorted(declared - defined), sorted(defined - declared))
    assert len(declared) >= 27
    # argument counts agree (JNIEnv*, jclass + the Java parameters)
    for name in declared:
        jargs = re.search(r"static native \S+ %s\(([^)]*)\)" % name, java).group(1)
        cargs = re.search(r"CLS\(%s\)\(([^)]*)\)" % name, c, re.S).group(1)
        nj = len([a for a in jargs.split(",") if a.strip()])
        nc = len([a for a in cargs.split(",") if a.strip()])
        assert nc == nj + 2, (name, jargs, cargs)


def test_no_jni_critical_region_brackets_a_library_call():
    """between GetPrimitiveArrayCritical and its Release there may be no gms_* call (they can synchronise a stream)"""
    c = open(os.path.join(ROOT, "jni", "gms_jni.c")).read()
    c = re.sub(r"/\*.*?\*/", "", c, flags=re.S)                       # code only
    regions = list(re.finditer(r"->GetPrimitiveArrayCritical\((.*?)->ReleasePrimitiveArrayCritical\(", c, re.S))
    assert regions
    for m in regions:
        assert not re.search(r"\bgms_\w+\(", m.group(1)), m.group(1)[:200]


def test_facade_overrides_the_reference_surface():
    g = open(os.path.join(ROOT, "jni", "java", "com", "fmsz", "gridmapgl", "slam", "GridMapGpu.java")).read()
    assert "class GridMapGpu extends GridMap" in g
    for sig in ["createMapData(GridMapData other)", "reset(GridMapData map)", "getRawAt(GridMapData map, int x, int y)",
                "getProbAt(GridMapData map, int x, int y)", "getRawAt(GridMapData map, Vec2 point)", "getLikelihood(GridMapData map, Vec2 point)",
                "integrateObservation(GridMapData map, Observation obs, Pose p)",
                "applyMeasurement(GridMapData map, float startX, float startY, float endX, float endY, float measuredDistance, boolean wasHit)",
                "computeLikelihoodMap(GridMapData map)", "probabilityOf(GridMapData map, Observation obs, Pose p)",
                "findBestPose(GridMapData map, Observation obs, Pose startPose)"]:
        assert sig in g, sig
    p = open(os.path.join(ROOT, "jni", "java", "com", "fmsz", "gridmapgl", "slam", "ParticleFilterGpu.java")).read()
    assert "class ParticleFilterGpu extends ParticleFilter" in p and "public double update(Observation z" in p
