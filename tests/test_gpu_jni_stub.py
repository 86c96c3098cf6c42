"""jni/gms_jni.c EXECUTED: every native of NativeSlam.java called through a minimal JNIEnv (tests/jni_stub/jni_env.c: the twelve
JNIEnv entries the shim uses, over C buffers standing for Java arrays) and compared with the ctypes path on the same inputs.  No JVM
exists in this image, so this is as far as the shim can be run here: what stays unverified is only that a JVM loads it
(System.loadLibrary, the mangled names against real class files).  The stub also checks the two JNI rules the shim must keep: array
regions inside their bounds, and no JNI call between GetPrimitiveArrayCritical and its Release."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, SLAMParticleMaps, synth
from gridmap_slam_robot_amd._lib import GmsParams, load

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "jni_stub"))
import build_stub  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, D, I, L, Z, P = C.c_float, C.c_double, C.c_int32, C.c_int64, C.c_uint8, C.c_void_p
SIG = {  # native -> (return type, argument types after JNIEnv*, jclass); P = a Java array or null
    "mapCreate": (L, [F, F, F, F, F, D, D, P, I, I]), "mapDestroy": (None, [L]), "mapReset": (None, [L]),
    "mapIntegrate": (None, [L, P, I, F, F, F]), "mapApplyRay": (None, [L, F, F, F, F, F, Z]), "mapBuildLikelihood": (None, [L]),
    "mapDownload": (None, [L, P, P]), "mapUpload": (None, [L, P, P]), "mapGetAtPoint": (None, [L, F, F, P]), "mapUpdateAt": (None, [L, P, I, L]),
    "pfCreate": (L, [L, I]), "pfDestroy": (None, [L]), "pfSetShard": (None, [L, L, L]), "pfSetRefine": (None, [L, Z]),
    "pfSetLogNormalize": (None, [L, Z]), "pfSetPoses": (None, [L, P, I]), "pfGetParticles": (None, [L, P, P, I]), "pfScore": (None, [L, P, I]),
    "pfProbabilityOf": (D, [L, P, I, F, F, F]), "pfFindBestPose": (None, [L, P, I, F, F, F, P]), "pfNormalize": (None, [L, P]),
    "pfResample": (None, [L, D]), "pfWeightedPose": (None, [L, P]), "slamUpdate": (None, [L, P, I, P, I, D, D, Z, P]),
    "commUniqueId": (None, [P]), "commCreate": (L, [P, I, I, I]), "commDestroy": (None, [L]),
    "slamUpdateSharded": (None, [L, L, P, I, P, I, D, D, Z, P]),
    "pmCreateShard": (L, [F, F, F, F, F, D, D, P, I, I, I, L, L]), "pmUpdateSharded": (None, [L, L, P, I, Z, D, D, L, L, P]),
    "pmResampleSharded": (Z, [L, L, D, D]),
    "pmCreate": (L, [F, F, F, F, F, D, D, P, I, I, I]), "pmDestroy": (None, [L]), "pmReset": (None, [L]), "pmSetRefine": (None, [L, Z]),
    "pmUpdate": (None, [L, P, I, Z, D, D, L, L, P]), "pmResample": (None, [L, D]), "pmResampleIf": (None, [L, D, D]),
    "pmGetParticles": (None, [L, P, P, I]), "pmWeightedPose": (None, [L, P]), "pmDownloadMap": (None, [L, I, P, P]), "pmCombined": (None, [L, P, P]),
}


class Jni:
    """the natives behind a stub JNIEnv; arrays are numpy-backed"""

    def __init__(self):
        self.lib = C.CDLL(build_stub.build())
        self.lib.stub_env.restype = P
        self.lib.stub_new_array.restype = P; self.lib.stub_new_array.argtypes = [I, I]
        self.lib.stub_array_data.restype = P; self.lib.stub_array_data.argtypes = [P]
        self.lib.stub_free_array.argtypes = [P]
        self.lib.stub_take_exception.argtypes = [C.c_char_p, C.c_char_p, I]
        self.env = self.lib.stub_env()
        self.called = set()
        for name, (res, args) in SIG.items():
            f = getattr(self.lib, "Java_com_fmsz_gridmapgl_slam_NativeSlam_" + name)
            f.restype = res
            f.argtypes = [P, P] + args
            setattr(self, "_" + name, f)

    def array(self, values, kind):
        dt = {1: np.float64, 2: np.float32, 3: np.int8}[kind]
        v = np.ascontiguousarray(values, dtype=dt).reshape(-1)
        a = self.lib.stub_new_array(kind, v.size)
        self.view(a, kind, v.size)[:] = v
        return a

    def view(self, a, kind, n):
        dt = {1: C.c_double, 2: C.c_float, 3: C.c_int8}[kind]
        return np.ctypeslib.as_array(C.cast(self.lib.stub_array_data(a), C.POINTER(dt)), shape=(n,))

    def call(self, name, *args, expect=None):
        self.called.add(name)
        r = getattr(self, "_" + name)(self.env, None, *args)
        cls, msg = C.create_string_buffer(200), C.create_string_buffer(600)
        had = self.lib.stub_take_exception(cls, msg, 200)
        if expect is None:
            assert not had, f"{name}: {cls.value.decode()}: {msg.value.decode()}"
        else:
            assert had and cls.value.decode() == expect, f"{name}: expected {expect}, got {cls.value.decode()!r} {msg.value.decode()!r}"
        return r


def flat(z):
    """NativeSlam.flatten(Observation): double[4 * B] = localX, localY, distance, wasHit"""
    out = np.empty((len(z), 4))
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = z["local_x"], z["local_y"], z["distance"], z["hit"] != 0
    return out.reshape(-1)


@pytest.mark.gpu
def test_every_native_called_once_against_the_ctypes_path():
    j = Jni()
    ext, res, B, N = 6.4, 0.05, 120, 256
    tr = synth.make_trace(ext, res, B, T=10, seed=3)
    prm = GmsParams()
    assert load().gms_params_default(C.byref(prm), ext, ext, res, -ext / 2, -ext / 2) == 0
    kernel = np.array([prm.kernel[i] for i in range(prm.ktaps)])
    karr = j.array(kernel, 1)
    # ---- GridMap
    m = j.call("mapCreate", ext, ext, res, -ext / 2, -ext / 2, prm.l_free, prm.l_occ, karr, 256, 0)
    assert m
    ref = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=256)
    W, H = ref.W, ref.H
    scans = [j.array(flat(tr.scans[t]), 1) for t in range(8)]
    for t in range(3):
        j.call("mapIntegrate", m, scans[t], B, *[float(v) for v in tr.poses[t]])
        ref.integrate_observation(tr.scans[t], tr.poses[t])
    j.call("mapApplyRay", m, 20.0, 20.0, 60.0, 31.0, 44.0, 1)
    ref.apply_measurement(20.0, 20.0, 60.0, 31.0, 44.0, True)
    j.call("mapBuildLikelihood", m)
    ref.compute_likelihood_map()
    la, ka = j.array(np.zeros(W * H), 1), j.array(np.zeros(W * H), 1)
    j.call("mapDownload", m, la, ka)
    assert np.array_equal(j.view(la, 1, W * H), ref.download_log().reshape(-1))
    assert np.array_equal(j.view(ka, 1, W * H), ref.download_likelihood().reshape(-1))
    out2 = j.array(np.zeros(2), 1)
    j.call("mapGetAtPoint", m, 0.4, -0.3, out2)
    i = int((0.4 + ext / 2) / res) + int((-0.3 + ext / 2) / res) * W
    assert np.array_equal(j.view(out2, 1, 2), [j.view(la, 1, W * H)[i], j.view(ka, 1, W * H)[i]])
    j.call("mapGetAtPoint", m, 1e6, 0.0, out2, expect="java/lang/ArrayIndexOutOfBoundsException")      # as the Java array access would
    j.call("mapReset", m)
    z0 = j.array(np.zeros(W * H), 1)
    j.call("mapDownload", m, z0, None)
    assert not j.view(z0, 1, W * H).any()
    j.call("mapUpload", m, la, ka)                                  # back to the built map
    short = j.array(np.zeros(10), 1)
    j.call("mapDownload", m, short, None, expect="java/lang/IllegalArgumentException")
    # ---- ParticleFilter
    pf = j.call("pfCreate", m, N)
    rpf = ParticleFilter(ref, N)
    Pn = synth.make_particles(tr.poses[3], N, seed=5, sigma_xy=0.03, sigma_theta_deg=1.0)
    parr = j.array(Pn, 2)
    j.call("pfSetPoses", pf, parr, N)
    j.call("pfSetPoses", pf, parr, N - 1, expect="java/lang/IllegalArgumentException")
    j.call("pfScore", pf, scans[3], B)
    st = j.array(np.zeros(3), 1)
    j.call("pfNormalize", pf, st)
    rpf.set_poses(Pn); rpf.score(tr.scans[3]); rst = rpf.normalize()
    assert np.array_equal(j.view(st, 1, 3), [rst["weight_sum"], rst["neff"], rst["strongest"]])
    xa, wa = j.array(np.zeros(3 * N), 2), j.array(np.zeros(N), 1)
    j.call("pfGetParticles", pf, xa, wa, N)
    assert np.array_equal(j.view(xa, 2, 3 * N).reshape(N, 3), rpf.get_poses()) and np.array_equal(j.view(wa, 1, N), rpf.get_weights())
    o3 = j.array(np.zeros(3), 2)
    j.call("pfWeightedPose", pf, o3)
    assert np.array_equal(j.view(o3, 2, 3), rpf.weighted_pose())
    j.call("pfResample", pf, 0.37)
    rpf.resample(0.37)
    j.call("pfGetParticles", pf, xa, wa, N)
    assert np.array_equal(j.view(xa, 2, 3 * N).reshape(N, 3), rpf.get_poses())
    # one pose: probabilityOf / findBestPose through a one-particle filter (GridMapGpu)
    pf1 = j.call("pfCreate", m, 1)
    r1 = ParticleFilter(ref, 1)
    pose = [float(v) for v in tr.poses[4]]
    got = j.call("pfProbabilityOf", pf1, scans[4], B, *pose)
    r1.set_poses(np.asarray([pose], np.float32)); r1.score(tr.scans[4])
    assert got == r1.get_weights()[0] and got > 0
    j.call("pfFindBestPose", pf1, scans[4], B, *pose, o3)
    r1.set_poses(np.asarray([pose], np.float32)); r1.refine_poses(tr.scans[4])
    assert np.array_equal(j.view(o3, 2, 3), r1.get_poses().reshape(-1))
    j.call("pfProbabilityOf", pf, scans[4], B, *pose, expect="java/lang/IllegalArgumentException")       # not a one-particle filter
    # options
    j.call("pfSetRefine", pf, 1); j.call("pfSetRefine", pf, 0)
    j.call("pfSetLogNormalize", pf, 1); j.call("pfSetLogNormalize", pf, 0)
    j.call("pfSetShard", pf, 0, N)                                   # a "shard" that is the whole population
    j.call("pfSetShard", pf, 3, N, expect="java/lang/IllegalArgumentException")
    # ---- the scan step (SLAMGpu's update): statistics out, map updated at the weighted pose
    P5 = synth.make_particles(tr.poses[5], N, seed=6, sigma_xy=0.03, sigma_theta_deg=1.0)
    j.call("slamUpdate", pf, j.array(P5, 2), N, scans[5], B, 0.25, -1.0, 1, st)
    rst = rpf.slam_update(P5, tr.scans[5], 0.25, -1.0, True, fetch=True)
    assert np.array_equal(j.view(st, 1, 3), [rst["weight_sum"], rst["neff"], rst["strongest"]])
    j.call("mapUpdateAt", m, scans[6], B, pf)
    ref.update_at(tr.scans[6], rpf)
    j.call("mapDownload", m, la, ka)
    assert np.array_equal(j.view(la, 1, W * H), ref.download_log().reshape(-1)) and np.array_equal(j.view(ka, 1, W * H), ref.download_likelihood().reshape(-1))
    # ---- the exchange inside the library: a communicator of ONE rank (the pool has one GPU per box)
    ida = j.array(np.zeros(128), 3)
    j.call("commUniqueId", ida)
    assert j.view(ida, 3, 128).any()
    cm = j.call("commCreate", ida, 0, 1, 0)
    assert cm
    P7 = synth.make_particles(tr.poses[7], N, seed=7, sigma_xy=0.03, sigma_theta_deg=1.0)
    j.call("slamUpdateSharded", pf, cm, j.array(P7, 2), N, scans[7], B, 0.5, -1.0, 1, st)
    rst = rpf.slam_update(P7, tr.scans[7], 0.5, -1.0, True, fetch=True)
    assert np.array_equal(j.view(st, 1, 3), [rst["weight_sum"], rst["neff"], rst["strongest"]])
    j.call("commDestroy", cm)
    j.call("pfDestroy", pf1); j.call("pfDestroy", pf)
    j.call("mapDestroy", m)
    rpf.close(); r1.close(); ref.close()
    # ---- SLAM as the reference has it: one GridMapData per particle (SLAMGpu)
    n = 40
    s = j.call("pmCreate", ext, ext, res, -ext / 2, -ext / 2, prm.l_free, prm.l_occ, karr, 256, 0, n)
    rs = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=n, max_beams=256)
    for refine in (0, 1):
        j.call("pmSetRefine", s, refine); rs.set_refine(bool(refine))
        for t in (1, 2):
            j.call("pmUpdate", s, scans[t], B, 1, 0.05, 0.02, 77, 10 * refine + t, st)
            rneff = rs.update(tr.scans[t], (0.05, 0.02), seed=77, sequence=10 * refine + t)
            assert j.view(st, 1, 3)[1] == rneff and j.view(st, 1, 3)[2] == rs.strongest
    j.call("pmResample", s, 0.41); rs.resample(0.41)
    j.call("pmResampleIf", s, 0.13, 0.5); rs.resample_if(0.13, 0.5)
    xa, wa = j.array(np.zeros(3 * n), 2), j.array(np.zeros(n), 1)
    j.call("pmGetParticles", s, xa, wa, n)
    Pr, wr = rs.get_particles()
    assert np.array_equal(j.view(xa, 2, 3 * n).reshape(n, 3), Pr) and np.array_equal(j.view(wa, 1, n), wr)
    j.call("pmWeightedPose", s, o3)
    assert np.array_equal(j.view(o3, 2, 3), rs.get_weighted_pose())
    j.call("pmDownloadMap", s, 7, la, ka)
    assert np.array_equal(j.view(la, 1, W * H), rs.map_of(7).reshape(-1)) and np.array_equal(j.view(ka, 1, W * H), rs.map_of(7, likelihood=True).reshape(-1))
    j.call("pmCombined", s, la, ka)
    comb = rs.calculate_combined().reshape(-1)
    got = j.view(la, 1, W * H)
    assert np.array_equal(np.isfinite(got), np.isfinite(comb)) and np.array_equal(got[np.isfinite(comb)], comb[np.isfinite(comb)])
    j.call("pmReset", s)
    j.call("pmDownloadMap", s, 0, la, None)
    assert not j.view(la, 1, W * H).any()
    j.call("pmDestroy", s)
    rs.close()
    # ---- ... sharded with its maps: one rank's block (here: the whole population on a communicator of one rank) through the library's own exchanges
    from gridmap_slam_robot_amd._lib import GMS_BLOCK
    n = GMS_BLOCK
    j.call("commUniqueId", ida)
    cm = j.call("commCreate", ida, 0, 1, 0)
    s = j.call("pmCreateShard", ext, ext, res, -ext / 2, -ext / 2, prm.l_free, prm.l_occ, karr, 256, 0, n, 0, n)
    rs = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=n, max_beams=256)
    for t in (1, 2):
        j.call("pmUpdateSharded", s, cm, scans[t], B, 1, 0.05, 0.02, 77, t, st)
        rneff = rs.update(tr.scans[t], (0.05, 0.02), seed=77, sequence=t)
        assert j.view(st, 1, 3)[1] == rneff and j.view(st, 1, 3)[2] == rs.strongest
    assert j.call("pmResampleSharded", s, cm, 0.41, -1.0) == 1
    rs.resample(0.41)
    assert j.call("pmResampleSharded", s, cm, 0.41, 1e-9) == 0                      # the rule says no: nothing is drawn
    xa, wa = j.array(np.zeros(3 * n), 2), j.array(np.zeros(n), 1)
    j.call("pmGetParticles", s, xa, wa, n)
    Pr, wr = rs.get_particles()
    assert np.array_equal(j.view(xa, 2, 3 * n).reshape(n, 3), Pr) and np.array_equal(j.view(wa, 1, n), wr)
    j.call("pmDownloadMap", s, 5, la, ka)
    assert np.array_equal(j.view(la, 1, W * H), rs.map_of(5).reshape(-1)) and np.array_equal(j.view(ka, 1, W * H), rs.map_of(5, likelihood=True).reshape(-1))
    j.call("pmCreateShard", ext, ext, res, -ext / 2, -ext / 2, prm.l_free, prm.l_occ, karr, 256, 0, 100, 0, 200, expect="java/lang/IllegalArgumentException")
    j.call("pmDestroy", s); j.call("commDestroy", cm)
    rs.close()
    # every native NativeSlam.java declares has been through the stub, and the shim kept the JNI rules
    java = open(os.path.join(ROOT, "jni", "java", "com", "fmsz", "gridmapgl", "slam", "NativeSlam.java")).read()
    declared = set(re.findall(r"static native \S+ (\w+)\(", java))
    assert declared == set(SIG) == j.called, (sorted(declared - j.called), sorted(j.called - declared))
    assert j.lib.stub_violations() == 0


def test_the_shim_builds_against_the_stub_and_exports_every_native():
    """(no GPU) gcc -Werror over jni/gms_jni.c + the stub JNIEnv, linked against the C-ABI library; every declared native is exported"""
    lib = C.CDLL(build_stub.build())
    java = open(os.path.join(ROOT, "jni", "java", "com", "fmsz", "gridmapgl", "slam", "NativeSlam.java")).read()
    declared = set(re.findall(r"static native \S+ (\w+)\(", java))
    assert declared == set(SIG)
    for name in declared:
        assert hasattr(lib, "Java_com_fmsz_gridmapgl_slam_NativeSlam_" + name), name
