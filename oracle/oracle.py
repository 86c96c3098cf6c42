"""ctypes binding of oracle/liboracle.so (the C restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- see oracle/gms_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  PARITY UNPINNED (no runnable reference, no
reference fixtures): see the header of gms_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

BEAM_DTYPE = np.dtype(
    [("local_x", "<f8"), ("local_y", "<f8"), ("distance", "<f8"), ("hit", "u1"), ("pad_", "u1", (7,))]
)
ORC_MAX_TAPS = 129


class OrcGrid(C.Structure):
    _fields_ = [
        ("W", C.c_int32), ("H", C.c_int32),
        ("resolution", C.c_float), ("pos_x", C.c_float), ("pos_y", C.c_float),
        ("l_free", C.c_double), ("l_prior", C.c_double), ("l_occ", C.c_double),
        ("ktaps", C.c_int32),
        ("kernel", C.c_double * ORC_MAX_TAPS),
        ("extra_steps", C.c_int32),
        ("hit_tolerance", C.c_float),
        ("z_hit", C.c_double), ("z_random", C.c_double),
        ("max_range", C.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gms_oracle.c")
    hdr = os.path.join(_HERE, "gms_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        gp = C.POINTER(OrcGrid)
        vp = C.c_void_p
        L.orc_log_odds.restype = C.c_double
        L.orc_log_odds.argtypes = [C.c_double]
        L.orc_inv_log_odds.restype = C.c_double
        L.orc_inv_log_odds.argtypes = [C.c_double]
        L.orc_generate_gaussian_kernel.restype = None
        L.orc_generate_gaussian_kernel.argtypes = [C.c_double, C.c_int, dp]
        L.orc_grid_init.restype = None
        L.orc_grid_init.argtypes = [gp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_trace_ray.restype = C.c_int32
        L.orc_trace_ray.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_int32, ip, C.c_int32]
        L.orc_apply_measurement.restype = C.c_int32
        L.orc_apply_measurement.argtypes = [gp, dp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_int, ip, C.POINTER(C.c_uint8), C.c_int32]
        L.orc_scan_rays.restype = None
        L.orc_scan_rays.argtypes = [gp, vp, C.c_int32, fp, fp]
        L.orc_integrate.restype = C.c_int64
        L.orc_integrate.argtypes = [gp, dp, vp, C.c_int32, fp]
        L.orc_scan_counts.restype = C.c_int64
        L.orc_scan_counts.argtypes = [gp, vp, C.c_int32, fp, vp]
        L.orc_build_likelihood.restype = None
        L.orc_build_likelihood.argtypes = [gp, dp, dp, dp]
        L.orc_probability_of.restype = C.c_double
        L.orc_probability_of.argtypes = [gp, dp, vp, C.c_int32, fp]
        L.orc_score.restype = None
        L.orc_score.argtypes = [gp, dp, vp, C.c_int32, fp, C.c_int32, dp]
        L.orc_score_mt.restype = None
        L.orc_score_mt.argtypes = [gp, dp, vp, C.c_int32, fp, C.c_int32, dp, C.c_int32]
        L.orc_score_log.restype = None
        L.orc_score_log.argtypes = [gp, dp, vp, C.c_int32, fp, C.c_int32, dp]
        L.orc_normalize.restype = C.c_double
        L.orc_normalize.argtypes = [dp, C.c_int32, ip]
        L.orc_neff.restype = C.c_double
        L.orc_neff.argtypes = [dp, C.c_int32]
        L.orc_weighted_pose.restype = None
        L.orc_weighted_pose.argtypes = [fp, dp, C.c_int32, fp]
        L.orc_resample_indices.restype = C.c_int32
        L.orc_resample_indices.argtypes = [dp, C.c_int32, C.c_double, ip]
        L.orc_find_best_pose.restype = C.c_double
        L.orc_find_best_pose.argtypes = [gp, dp, vp, C.c_int32, fp, fp, ip]
        L.orc_sample_motion.restype = None
        L.orc_sample_motion.argtypes = [fp, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64]
        L.orc_combine_maps.restype = None
        L.orc_combine_maps.argtypes = [dp, C.c_int32, C.c_int64, dp]
        L.orc_deskew.restype = None
        L.orc_deskew.argtypes = [dp, dp, C.POINTER(C.c_uint8), C.c_int32, C.c_double, C.c_double, vp]
        L.orc_philox4x32.restype = None
        L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox_normals.restype = None
        L.orc_philox_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, dp, dp]
        L.orc_point_index.restype = C.c_int32
        L.orc_point_index.argtypes = [C.POINTER(OrcGrid), C.c_float, C.c_float]
        L.orc_pose_trig.restype = None
        L.orc_pose_trig.argtypes = [C.c_float, dp, dp]
        L.orc_count_trig_mismatches.restype = C.c_int64
        L.orc_count_trig_mismatches.argtypes = [fp, fp, fp, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        L.orc_trig_near_float_boundary.restype = C.c_int64
        L.orc_trig_near_float_boundary.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, fp, ip, dp, C.c_int64, C.c_int32]
        L.orc_count_sqrt_mismatches.restype = C.c_int64
        L.orc_count_sqrt_mismatches.argtypes = [fp, fp, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        L.orc_slam_new.restype = vp
        L.orc_slam_new.argtypes = [gp, C.c_int32]
        L.orc_slam_free.restype = None
        L.orc_slam_free.argtypes = [vp]
        L.orc_slam_reset.restype = None
        L.orc_slam_reset.argtypes = [vp]
        L.orc_slam_count.restype = C.c_int32
        L.orc_slam_count.argtypes = [vp]
        L.orc_slam_poses.restype = fp
        L.orc_slam_poses.argtypes = [vp]
        L.orc_slam_weights.restype = dp
        L.orc_slam_weights.argtypes = [vp]
        L.orc_slam_log.restype = dp
        L.orc_slam_log.argtypes = [vp, C.c_int32]
        L.orc_slam_lik.restype = dp
        L.orc_slam_lik.argtypes = [vp, C.c_int32]
        L.orc_slam_strongest.restype = C.c_int32
        L.orc_slam_strongest.argtypes = [vp]
        L.orc_slam_update.restype = C.c_double
        L.orc_slam_update.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int32]
        L.orc_slam_update_mt.restype = C.c_double
        L.orc_slam_update_mt.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]
        L.orc_set_threads.restype = C.c_int32
        L.orc_set_threads.argtypes = [C.c_int32]
        L.orc_slam_resample.restype = C.c_int32
        L.orc_slam_resample.argtypes = [vp, C.c_double, ip]
        L.orc_slam_neff.restype = C.c_double
        L.orc_slam_neff.argtypes = [vp]
        L.orc_slam_weighted_pose.restype = None
        L.orc_slam_weighted_pose.argtypes = [vp, fp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def make_beams(local_x, local_y, distance, hit) -> np.ndarray:
    b = np.zeros(len(local_x), dtype=BEAM_DTYPE)
    b["local_x"] = local_x
    b["local_y"] = local_y
    b["distance"] = distance
    b["hit"] = np.asarray(hit).astype(np.uint8)
    return b


class Grid:
    """GridMap geometry/constants (J/slam/GridMap.java:80-100) over the C oracle."""

    def __init__(self, width: float, height: float, resolution: float, pos_x: float, pos_y: float):
        self.g = OrcGrid()
        lib().orc_grid_init(C.byref(self.g), width, height, resolution, pos_x, pos_y)
        if self.g.ktaps == 0:
            raise ValueError("likelihood kernel too wide for the oracle")

    # -- geometry -----------------------------------------------------------------------------
    @property
    def W(self): return self.g.W
    @property
    def H(self): return self.g.H
    @property
    def resolution(self): return self.g.resolution
    @property
    def pos(self): return (self.g.pos_x, self.g.pos_y)
    @property
    def kernel(self) -> np.ndarray:
        return np.array(self.g.kernel[: self.g.ktaps], dtype=np.float64)

    def point_index(self, px: float, py: float) -> int:
        """flat index of getRawAt(map, Vec2) / getLikelihood(map, Vec2) (GridMap.java:142-156)"""
        return int(lib().orc_point_index(C.byref(self.g), np.float32(px), np.float32(py)))

    def set_kernel(self, taps):
        taps = np.asarray(taps, dtype=np.float64)
        assert taps.size % 2 == 1 and taps.size <= ORC_MAX_TAPS
        self.g.ktaps = taps.size
        for i, t in enumerate(taps):
            self.g.kernel[i] = float(t)

    @property
    def l_free(self): return self.g.l_free
    @property
    def l_occ(self): return self.g.l_occ

    def new_log(self) -> np.ndarray:
        return np.full(self.W * self.H, self.g.l_prior, dtype=np.float64)

    # -- path functions -----------------------------------------------------------------------
    def trace_ray(self, x0, y0, x1, y1, extra=2, cap=1 << 16) -> np.ndarray:
        cells = np.empty((cap, 2), dtype=np.int32)
        n = lib().orc_trace_ray(self.W, self.H, x0, y0, x1, y1, extra, _ip(cells), cap)
        assert n <= cap
        return cells[:n].copy()

    def apply_measurement(self, log, sx, sy, ex, ey, measured, hit, cap=1 << 16):
        cells = np.empty((cap, 2), dtype=np.int32)
        cls = np.empty(cap, dtype=np.uint8)
        n = lib().orc_apply_measurement(C.byref(self.g), _dp(log) if log is not None else None,
                                        sx, sy, ex, ey, measured, int(bool(hit)), _ip(cells),
                                        cls.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
        assert n <= cap
        return cells[:n].copy(), cls[:n].copy()

    def scan_rays(self, beams: np.ndarray, pose) -> np.ndarray:
        pose = np.asarray(pose, dtype=np.float32)
        out = np.empty((len(beams), 6), dtype=np.float32)
        lib().orc_scan_rays(C.byref(self.g), beams.ctypes.data, len(beams), _fp(pose), _fp(out))
        return out

    def scan_counts(self, beams: np.ndarray, pose) -> np.ndarray:
        """[H * W][3] visits of every cell by the scan's rays per sensor class (free, prior, occupied): integrateObservation's integers"""
        pose = np.asarray(pose, dtype=np.float32)
        out = np.empty((self.W * self.H, 3), dtype=np.uint32)
        lib().orc_scan_counts(C.byref(self.g), beams.ctypes.data, len(beams), _fp(pose), out.ctypes.data)
        return out

    def integrate(self, log: np.ndarray, beams: np.ndarray, pose) -> int:
        pose = np.asarray(pose, dtype=np.float32)
        return lib().orc_integrate(C.byref(self.g), _dp(log), beams.ctypes.data, len(beams), _fp(pose))

    def build_likelihood(self, log: np.ndarray) -> np.ndarray:
        lik = np.zeros(self.W * self.H, dtype=np.float64)
        scratch = np.empty(2 * self.W * self.H, dtype=np.float64)
        lib().orc_build_likelihood(C.byref(self.g), _dp(log), _dp(lik), _dp(scratch))
        return lik

    def probability_of(self, lik, beams, pose) -> float:
        pose = np.asarray(pose, dtype=np.float32)
        return lib().orc_probability_of(C.byref(self.g), _dp(lik), beams.ctypes.data, len(beams), _fp(pose))

    def score(self, lik, beams, poses) -> np.ndarray:
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        w = np.empty(len(poses), dtype=np.float64)
        lib().orc_score(C.byref(self.g), _dp(lik), beams.ctypes.data, len(beams), _fp(poses), len(poses), _dp(w))
        return w

    def score_mt(self, lik, beams, poses, threads: int) -> np.ndarray:
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        w = np.empty(len(poses), dtype=np.float64)
        lib().orc_score_mt(C.byref(self.g), _dp(lik), beams.ctypes.data, len(beams), _fp(poses), len(poses), _dp(w), int(threads))
        return w

    def score_log(self, lik, beams, poses) -> np.ndarray:
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
        w = np.empty(len(poses), dtype=np.float64)
        lib().orc_score_log(C.byref(self.g), _dp(lik), beams.ctypes.data, len(beams), _fp(poses), len(poses), _dp(w))
        return w

    def find_best_pose(self, lik, beams, start):
        start = np.asarray(start, dtype=np.float32)
        best = np.empty(3, dtype=np.float32)
        n = C.c_int32(0)
        p = lib().orc_find_best_pose(C.byref(self.g), _dp(lik), beams.ctypes.data, len(beams), _fp(start),
                                     _fp(best), C.byref(n))
        return best, p, n.value


def normalize(weights: np.ndarray):
    """In place; returns (weight_sum, strongest)."""
    s = C.c_int32(-1)
    ws = lib().orc_normalize(_dp(weights), len(weights), C.byref(s))
    return ws, s.value


def neff(weights: np.ndarray) -> float:
    return lib().orc_neff(_dp(weights), len(weights))


def weighted_pose(poses: np.ndarray, weights: np.ndarray) -> np.ndarray:
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3)
    out = np.empty(3, dtype=np.float32)
    lib().orc_weighted_pose(_fp(poses), _dp(weights), len(weights), _fp(out))
    return out


def resample_indices(weights: np.ndarray, r01: float):
    idx = np.empty(len(weights), dtype=np.int32)
    clamped = lib().orc_resample_indices(_dp(weights), len(weights), r01, _ip(idx))
    return idx, clamped


def pose_trig(theta):
    c = C.c_double()
    s = C.c_double()
    lib().orc_pose_trig(np.float32(theta), C.byref(c), C.byref(s))
    return c.value, s.value


def count_trig_mismatches(theta: np.ndarray, got_cos: np.ndarray, got_sin: np.ndarray, threads: int = 1):
    """(mismatches, index of the first one or -1): got_* against (float)cos/sin((double)theta), glibc, element by element"""
    first = C.c_int64(-1)
    n = lib().orc_count_trig_mismatches(_fp(theta), _fp(got_cos), _fp(got_sin), len(theta), threads, C.byref(first))
    return int(n), int(first.value)


def trig_near_float_boundary(lo_bits: int, hi_bits: int, negative: bool, window_ulps: float, threads: int = 1, cap: int = 1 << 16):
    """[(theta, 'cos' | 'sin', distance in ulps)] for the float bit patterns lo..hi whose glibc cos / sin lies within window_ulps of a
    rounding boundary of the float it is narrowed to (orc_trig_near_float_boundary)."""
    th = np.empty(cap, dtype=np.float32)
    wh = np.empty(cap, dtype=np.int32)
    ds = np.empty(cap, dtype=np.float64)
    n = lib().orc_trig_near_float_boundary(lo_bits, hi_bits, 0x80000000 if negative else 0, window_ulps, _fp(th), _ip(wh), _dp(ds), cap, threads)
    if n > cap:
        raise RuntimeError(f"{n} candidates exceed the buffer of {cap}")
    order = np.lexsort((wh[:n], th[:n].view(np.uint32)))
    return [(float(th[i]), "sin" if wh[i] else "cos", float(ds[i])) for i in order]


def count_sqrt_mismatches(a: np.ndarray, got: np.ndarray, threads: int = 1):
    first = C.c_int64(-1)
    n = lib().orc_count_sqrt_mismatches(_fp(a), _fp(got), len(a), threads, C.byref(first))
    return int(n), int(first.value)


def set_threads(n: int) -> int:
    """OpenMP threads of the oracle's parallel loops from now on (orc_set_threads); returns the previous maximum"""
    return int(lib().orc_set_threads(int(n)))


def gaussian_kernel(sigma: float, size: int) -> np.ndarray:
    out = np.empty(2 * size + 1, dtype=np.float64)
    lib().orc_generate_gaussian_kernel(sigma, size, _dp(out))
    return out


def log_odds(p: float) -> float:
    return lib().orc_log_odds(p)


def sample_motion(poses: np.ndarray, d_center: float, d_theta: float, seed: int, sequence: int, index_offset: int = 0) -> np.ndarray:
    """Odometry.apply on every pose (Odometry.java:77-96) with Philox-generated normals; returns new poses."""
    out = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 3).copy()
    lib().orc_sample_motion(_fp(out), len(out), index_offset, d_center, d_theta, seed, sequence)
    return out


def philox_normals(seed: int, sequence: int, index: int):
    a, b = C.c_double(), C.c_double()
    lib().orc_philox_normals(seed, sequence, index, C.byref(a), C.byref(b))
    return a.value, b.value


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return [int(x) for x in o]


def combine_maps(logs: np.ndarray) -> np.ndarray:
    """calculateCombined (GridMapApp.java:439-458) over logs[n_maps][cells]."""
    logs = np.ascontiguousarray(logs, dtype=np.float64)
    M = logs.shape[0]
    cells = logs.size // M
    out = np.empty(cells, dtype=np.float64)
    lib().orc_combine_maps(_dp(logs), M, cells, _dp(out))
    return out


def deskew(angle, distance, hit, d_center: float, d_theta: float) -> np.ndarray:
    """The de-skew loop of GridMapApp.onHandleData (GridMapApp.java:143-175) -> beams."""
    angle = np.ascontiguousarray(angle, dtype=np.float64)
    distance = np.ascontiguousarray(distance, dtype=np.float64)
    hit = np.ascontiguousarray(hit, dtype=np.uint8)
    out = np.zeros(len(angle), dtype=BEAM_DTYPE)
    lib().orc_deskew(_dp(angle), _dp(distance), hit.ctypes.data_as(C.POINTER(C.c_uint8)), len(angle), d_center, d_theta,
                     out.ctypes.data)
    return out


class Slam:
    """SLAM (J/slam/SLAM.java): n particles, each with a pose, a weight and its own GridMapData -- the reference's filter shape
    (orc_slam_* in gms_oracle.c).  poses / weights / log(i) / lik(i) are COPIES of the oracle's state."""

    def __init__(self, grid: "Grid", n: int):
        self.grid = grid
        self.n = int(n)
        self._h = lib().orc_slam_new(C.byref(grid.g), self.n)
        if not self._h:
            raise MemoryError("orc_slam_new")
        self.cells = grid.W * grid.H

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_slam_free(self._h)
            self._h = None

    def reset(self):
        lib().orc_slam_reset(self._h)

    @property
    def poses(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_slam_poses(self._h), shape=(self.n, 3)).copy()

    def set_poses(self, poses):
        np.ctypeslib.as_array(lib().orc_slam_poses(self._h), shape=(self.n, 3))[:] = np.asarray(poses, dtype=np.float32).reshape(self.n, 3)

    @property
    def weights(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_slam_weights(self._h), shape=(self.n,)).copy()

    def set_weights(self, w):
        np.ctypeslib.as_array(lib().orc_slam_weights(self._h), shape=(self.n,))[:] = np.asarray(w, dtype=np.float64)

    def log(self, i: int) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_slam_log(self._h, int(i)), shape=(self.cells,)).copy()

    def lik(self, i: int) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_slam_lik(self._h, int(i)), shape=(self.cells,)).copy()

    def set_log(self, i: int, log):
        np.ctypeslib.as_array(lib().orc_slam_log(self._h, int(i)), shape=(self.cells,))[:] = np.asarray(log, dtype=np.float64).reshape(-1)

    def set_lik(self, i: int, lik):
        np.ctypeslib.as_array(lib().orc_slam_lik(self._h, int(i)), shape=(self.cells,))[:] = np.asarray(lik, dtype=np.float64).reshape(-1)

    def logs(self) -> np.ndarray:
        return np.stack([self.log(i) for i in range(self.n)])

    def liks(self) -> np.ndarray:
        return np.stack([self.lik(i) for i in range(self.n)])

    @property
    def strongest(self) -> int:
        return int(lib().orc_slam_strongest(self._h))

    def update(self, beams: np.ndarray, odometry=None, seed: int = 0, sequence: int = 0, refine: bool = False,
               sample_motion: bool = True, threads: int = 1) -> float:
        """SLAM.update(z, u) (SLAM.java:80-131); odometry = (dCenter, dTheta) or None (= (0, 0), no motion sample); sample_motion =
        False keeps the poses as they are (the caller has set the samples) while dTheta still decides skipUpdate (:82); returns Neff"""
        have = odometry is not None
        dc, dt = (odometry if have else (0.0, 0.0))
        return float(lib().orc_slam_update_mt(self._h, beams.ctypes.data, len(beams), int(have and sample_motion), float(dc), float(dt),
                                              int(seed), int(sequence), int(bool(refine)), int(threads)))

    def resample(self, r01: float):
        """SLAM.resample() (SLAM.java:133-153); returns (source index per slot, clamped slots)"""
        idx = np.empty(self.n, dtype=np.int32)
        clamped = lib().orc_slam_resample(self._h, float(r01), _ip(idx))
        return idx, int(clamped)

    def neff(self) -> float:
        return float(lib().orc_slam_neff(self._h))

    def weighted_pose(self) -> np.ndarray:
        out = np.empty(3, dtype=np.float32)
        lib().orc_slam_weighted_pose(self._h, _fp(out))
        return out
