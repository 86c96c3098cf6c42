"""Host-side mirror of the reference's GridMap / ParticleFilter / SLAM class surface, over the C-ABI.

Method names and argument meaning follow the Java classes (J/ = java/GridMapGL/src/main/java/com/
fmsz/gridmapgl/ in the reference tree):

  GridMap          J/slam/GridMap.java:47-432   (geometry + GridMapData; one handle holds both)
  Observation      J/slam/Observation.java:29-106
  ParticleFilter   J/slam/ParticleFilter.java:19-84  (resample semantics of SLAM.resample)
  SLAM             J/slam/SLAM.java:26-204

Everything that computes runs in libgridmapslam.so on the GPU; this file only marshals.
snake_case names are primary, the Java camelCase names are kept as aliases.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import BEAM_DTYPE, GmsParams, GmsPfStats, check, load, ptr

__all__ = ["GridMap", "Observation", "ParticleFilter", "SLAM", "Pose"]


def Pose(x: float, y: float, theta: float) -> np.ndarray:
    """Pose{float x, y, theta} (J/slam/Pose.java:21-35)."""
    return np.array([x, y, theta], dtype=np.float32)


class Observation:
    """One LIDAR revolution (J/slam/Observation.java)."""

    def __init__(self, beams: Optional[np.ndarray] = None):
        self.beams = np.zeros(0, dtype=BEAM_DTYPE) if beams is None else np.ascontiguousarray(beams, dtype=BEAM_DTYPE)

    @staticmethod
    def from_polar(angles, distances, hits) -> "Observation":
        """Measurement(angle, distance, wasHit): localX = distance * cos(angle) (Observation.java:44-51)."""
        angles = np.asarray(angles, dtype=np.float64)
        distances = np.asarray(distances, dtype=np.float64)
        b = np.zeros(angles.shape, dtype=BEAM_DTYPE)
        # math.cos per element: the same libm the oracle uses (numpy's SIMD cos may differ by an ulp)
        b["local_x"] = distances * np.array([math.cos(a) for a in angles.ravel()]).reshape(angles.shape)
        b["local_y"] = distances * np.array([math.sin(a) for a in angles.ravel()]).reshape(angles.shape)
        b["distance"] = distances
        b["hit"] = np.asarray(hits).astype(np.uint8)
        return Observation(b)

    @staticmethod
    def from_local(x, y, hits) -> "Observation":
        """Measurement(x, y, wasHit, dummy) (Observation.java:69-76)."""
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        b = np.zeros(x.shape, dtype=BEAM_DTYPE)
        b["local_x"] = x
        b["local_y"] = y
        b["distance"] = np.sqrt(x * x + y * y)
        b["hit"] = np.asarray(hits).astype(np.uint8)
        return Observation(b)

    def getNumberOfMeasurements(self) -> int:
        return int(self.beams.shape[-1])

    def __len__(self) -> int:
        return self.getNumberOfMeasurements()


def _beams_of(obs) -> np.ndarray:
    b = obs.beams if isinstance(obs, Observation) else obs
    return np.ascontiguousarray(b, dtype=BEAM_DTYPE)


class GridMap:
    """GridMap(width, height, resolution, position) + its GridMapData (GridMap.java:80-132)."""

    def __init__(self, width: float, height: float, resolution: float, position: Sequence[float],
                 n_maps: int = 1, device: int = 0, max_beams: int = 0, kernel=None,
                 l_free: Optional[float] = None, l_occ: Optional[float] = None):
        L = load()
        p = GmsParams()
        check(L.gms_params_default(C.byref(p), width, height, resolution, position[0], position[1]))
        p.n_maps = n_maps
        p.device = device
        p.max_beams = max_beams
        if kernel is not None:
            k = np.asarray(kernel, dtype=np.float64)
            if k.size % 2 != 1 or k.size > _lib.GMS_MAX_TAPS:
                raise ValueError("kernel must have an odd number of taps <= GMS_MAX_TAPS")
            p.ktaps = k.size
            for i, t in enumerate(k):
                p.kernel[i] = float(t)
        if l_free is not None:
            p.l_free = l_free
        if l_occ is not None:
            p.l_occ = l_occ
        self.params = p
        self._h = C.c_void_p()
        check(L.gms_map_create(C.byref(p), C.byref(self._h)))
        W, H, M = C.c_int32(), C.c_int32(), C.c_int32()
        check(L.gms_map_get_size(self._h, C.byref(W), C.byref(H), C.byref(M)))
        self.W, self.H, self.n_maps = W.value, H.value, M.value
        self._filters = weakref.WeakSet()         # ParticleFilters bound to this map: they are closed before it

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for pf in list(getattr(self, "_filters", ())):      # gms_map_destroy refuses while filters are alive
                pf.close()
            check(load().gms_map_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- geometry (GridMap.java:422-432) ---------------------------------------------------------
    @property
    def resolution(self) -> float:
        return float(self.params.resolution)

    @property
    def position(self):
        return (float(self.params.pos_x), float(self.params.pos_y))

    @property
    def kernel(self) -> np.ndarray:
        return np.array(self.params.kernel[: self.params.ktaps], dtype=np.float64)

    def getResolution(self): return self.resolution
    def getPosition(self): return self.position
    def getWorldSize(self):
        # gridSize * resolution in float (GridMap.java:88)
        return (float(np.float32(self.W) * np.float32(self.resolution)), float(np.float32(self.H) * np.float32(self.resolution)))

    def point_in_map(self, point) -> bool:
        """pointInMap (GridMap.java:164-170): float arithmetic."""
        tx = (np.float32(point[0]) - np.float32(self.params.pos_x)) / np.float32(self.params.resolution)
        ty = (np.float32(point[1]) - np.float32(self.params.pos_y)) / np.float32(self.params.resolution)
        return not (tx < 0 or ty < 0 or tx >= self.W or ty >= self.H)

    # -- stream / sync ---------------------------------------------------------------------------
    def set_stream(self, hip_stream: Optional[int]):
        """Run on an existing HIP stream (e.g. torch.cuda.Stream().cuda_stream).  None / 0 = the handle's own stream --
        note that torch's DEFAULT stream has handle 0, so passing it does not put the library on torch's stream."""
        check(load().gms_map_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        check(load().gms_map_synchronize(self._h))

    # -- GridMapData ------------------------------------------------------------------------------
    def _shape(self):
        return (self.H, self.W) if self.n_maps == 1 else (self.n_maps, self.H, self.W)

    def reset(self):
        check(load().gms_map_reset(self._h))

    def upload_log(self, log):
        a = np.ascontiguousarray(log, dtype=np.float64)
        assert a.size == self.n_maps * self.W * self.H
        check(load().gms_map_upload_log(self._h, ptr(a)))

    def download_log(self) -> np.ndarray:
        out = np.empty(self._shape(), dtype=np.float64)
        check(load().gms_map_download_log(self._h, ptr(out)))
        return out

    def upload_likelihood(self, lik):
        a = np.ascontiguousarray(lik, dtype=np.float64)
        assert a.size == self.n_maps * self.W * self.H
        check(load().gms_map_upload_likelihood(self._h, ptr(a)))

    def download_likelihood(self) -> np.ndarray:
        out = np.empty(self._shape(), dtype=np.float64)
        check(load().gms_map_download_likelihood(self._h, ptr(out)))
        return out

    def copy_from(self, other: "GridMap"):
        """createMapData(other) (GridMap.java:106-124)."""
        check(load().gms_map_copy(self._h, other._h))

    def combine_from(self, batch: "GridMap"):
        """calculateCombined (J/app/GridMapApp.java:439-458): this (single) map := combination of batch's maps."""
        check(load().gms_map_combine(self._h, batch._h))

    def deskew(self, angles, distances, hits, d_center: float, d_theta: float) -> "Observation":
        """The de-skew loop of GridMapApp.onHandleData (J/app/GridMapApp.java:143-175), on the device."""
        a = np.ascontiguousarray(angles, dtype=np.float64)
        d = np.ascontiguousarray(distances, dtype=np.float64)
        h = np.ascontiguousarray(hits, dtype=np.uint8)
        out = np.zeros(len(a), dtype=BEAM_DTYPE)
        check(load().gms_map_deskew(self._h, ptr(a), ptr(d), ptr(h), len(a), d_center, d_theta, ptr(out), None))
        return Observation(out)

    def get_raw_at(self, x: int, y: int, mi: int = 0) -> float:
        raw = C.c_double()
        check(load().gms_map_get_raw_at(self._h, mi, x, y, C.byref(raw), None))
        return raw.value

    def get_prob_at(self, x: int, y: int, mi: int = 0) -> float:
        prob = C.c_double()
        check(load().gms_map_get_raw_at(self._h, mi, x, y, None, C.byref(prob)))
        return prob.value

    # -- the hot path -----------------------------------------------------------------------------
    def get_raw_at_point(self, point, mi: int = 0) -> float:
        """getRawAt(map, Vec2 point) (GridMap.java:142-148)."""
        raw = C.c_double()
        check(load().gms_map_get_at_point(self._h, mi, float(point[0]), float(point[1]), C.byref(raw), None))
        return raw.value

    def get_likelihood(self, point, mi: int = 0) -> float:
        """getLikelihood(map, Vec2 point) (GridMap.java:150-156)."""
        lik = C.c_double()
        check(load().gms_map_get_at_point(self._h, mi, float(point[0]), float(point[1]), None, C.byref(lik)))
        return lik.value

    getLikelihood = get_likelihood

    def _beam_args(self, obs):
        b = _beams_of(obs)
        if self.n_maps > 1:
            assert b.ndim == 2 and b.shape[0] == self.n_maps, "beams must be [n_maps][B]"
        B = b.shape[-1]
        return b, B

    def _pose_args(self, pose):
        p = np.ascontiguousarray(pose, dtype=np.float32)
        assert p.size == 3 * self.n_maps
        return p

    def integrate_observation(self, obs, pose):
        """integrateObservation(map, obs, pose) (GridMap.java:173-191)."""
        b, B = self._beam_args(obs)
        p = self._pose_args(pose)
        check(load().gms_map_integrate(self._h, ptr(b), B, ptr(p)))

    def integrate_at(self, obs, pf: "ParticleFilter", strongest: bool = False):
        b, B = self._beam_args(obs)
        check(load().gms_map_integrate_at(self._h, ptr(b), B, pf._h, 1 if strongest else 0))

    def apply_measurement(self, start_x, start_y, end_x, end_y, measured_distance, was_hit):
        """applyMeasurement (GridMap.java:194-228), grid coordinates, map 0."""
        check(load().gms_map_apply_ray(self._h, start_x, start_y, end_x, end_y, measured_distance, int(bool(was_hit))))

    def trace_ray(self, x0, y0, x1, y1, extra=2, cap=4096) -> np.ndarray:
        """RayIterator.init + iteration (J/slam/RayIterator.java:65-130) -> [n][2] cells, in order."""
        cells = np.empty((cap, 2), dtype=np.int32)
        n = C.c_int32()
        check(load().gms_map_trace_ray(self._h, x0, y0, x1, y1, extra, ptr(cells), cap, C.byref(n)))
        if n.value > cap:
            return self.trace_ray(x0, y0, x1, y1, extra, cap=n.value)
        return cells[: n.value].copy()

    def trace_scan(self, obs, pose, cap=2048):
        """Cells and sensor-model classes each beam of integrateObservation visits (map untouched)."""
        b = _beams_of(obs)
        B = b.shape[-1]
        p = np.ascontiguousarray(pose, dtype=np.float32)
        cells = np.empty((B, cap, 2), dtype=np.int32)
        cls = np.empty((B, cap), dtype=np.uint8)
        counts = np.empty(B, dtype=np.int32)
        check(load().gms_map_trace_scan(self._h, ptr(b), B, ptr(p), ptr(cells), ptr(cls), cap, ptr(counts)))
        if counts.max(initial=0) > cap:
            return self.trace_scan(obs, pose, cap=int(counts.max()))
        return cells, cls, counts

    def compute_likelihood_map(self):
        """computeLikelihoodMap(map) (GridMap.java:233-250)."""
        check(load().gms_map_build_likelihood(self._h))

    def update(self, obs, pose):
        """integrateObservation + computeLikelihoodMap restricted to what the scan changed."""
        b, B = self._beam_args(obs)
        p = self._pose_args(pose)
        check(load().gms_map_update(self._h, ptr(b), B, ptr(p)))

    def update_at(self, obs, pf: "ParticleFilter", strongest: bool = False):
        b, B = self._beam_args(obs)
        check(load().gms_map_update_at(self._h, ptr(b), B, pf._h, 1 if strongest else 0))

    def probability_of(self, obs, pose) -> float:
        """probabilityOf(map, obs, pose) (GridMap.java:261-294) for one pose (map 0)."""
        assert self.n_maps == 1
        pf = ParticleFilter(self, 1)
        try:
            pf.set_poses(np.asarray(pose, dtype=np.float32).reshape(1, 3))
            pf.score(obs)
            return float(pf.get_weights()[0])
        finally:
            pf.close()

    def find_best_pose(self, obs, start_pose) -> np.ndarray:
        """findBestPose(map, obs, startPose) (GridMap.java:319-346)."""
        assert self.n_maps == 1
        pf = ParticleFilter(self, 1)
        try:
            pf.set_poses(np.asarray(start_pose, dtype=np.float32).reshape(1, 3))
            pf.refine_poses(obs)
            return pf.get_poses()[0]
        finally:
            pf.close()

    # -- measurement ------------------------------------------------------------------------------
    def profile(self, on=True):
        """on: True/False, or a bitmask of kernel classes (bit k = _lib.K_*)."""
        mask = ((1 << len(_lib.KERNEL_NAMES)) - 1 if on else 0) if isinstance(on, bool) else int(on)
        check(load().gms_profile_enable(self._h, mask))

    def deskew_dev(self, angles, distances, hits, d_center: float, d_theta: float):
        """deskew() without the read-back: the beams stay in the map's device staging buffer; returns (device address, count).
        Nothing is synchronised: the next launches on the handle's stream see them."""
        a = np.ascontiguousarray(angles, dtype=np.float64)
        d = np.ascontiguousarray(distances, dtype=np.float64)
        h = np.ascontiguousarray(hits, dtype=np.uint8)
        dev = C.c_void_p()
        check(load().gms_map_deskew(self._h, ptr(a), ptr(d), ptr(h), len(a), d_center, d_theta, None, C.byref(dev)))
        return dev.value, len(a)

    # -- device-resident inputs (raw device pointers, e.g. torch tensor .data_ptr()) -------------------
    def update_dev(self, dev_beams: int, B: int, dev_poses: int):
        check(load().gms_map_update_dev(self._h, C.c_void_p(dev_beams), B, C.c_void_p(dev_poses)))

    def integrate_dev(self, dev_beams: int, B: int, dev_poses: int):
        check(load().gms_map_integrate_dev(self._h, C.c_void_p(dev_beams), B, C.c_void_p(dev_poses)))

    def update_at_dev(self, dev_beams: int, B: int, pf: "ParticleFilter", strongest: bool = False):
        check(load().gms_map_update_at_dev(self._h, C.c_void_p(dev_beams), B, pf._h, 1 if strongest else 0))

    def integrate_at_dev(self, dev_beams: int, B: int, pf: "ParticleFilter", strongest: bool = False):
        check(load().gms_map_integrate_at_dev(self._h, C.c_void_p(dev_beams), B, pf._h, 1 if strongest else 0))

    def profile_sample(self, stride: int = 1):
        """Bracket only every stride-th launch of the enabled kernel classes."""
        check(load().gms_profile_sample(self._h, stride))

    def profile_calibrate(self, reps: int = 200) -> float:
        """Mean event-bracket time around an empty kernel, in milliseconds (what a bracket costs by itself)."""
        v = C.c_double()
        check(load().gms_profile_calibrate(self._h, reps, C.byref(v)))
        return v.value

    def profile_calibrate2(self, reps: int = 200):
        """(event bracket around an empty kernel, one empty kernel in a back-to-back queue), both in milliseconds; their
        difference is what the two event markers add to a bracketed launch."""
        a, b = C.c_double(), C.c_double()
        check(load().gms_profile_calibrate2(self._h, reps, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_reset(self):
        check(load().gms_profile_reset(self._h))

    def profile_get(self) -> dict:
        out = {}
        for k, name in enumerate(_lib.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_int64()
            check(load().gms_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def tile_stats(self, enable: bool = True, fetch: bool = True):
        """Census of the likelihood tiles the rebuilds walked since the last call (gms_map_tile_stats): a dict
        {left_alone, constants_kept, constants_written, blurred}; reads and clears the counters, `enable` keeps counting."""
        out = (C.c_int64 * 4)()
        check(load().gms_map_tile_stats(self._h, 1 if enable else 0, out if fetch else None))
        return dict(zip(("left_alone", "constants_kept", "constants_written", "blurred"), [int(v) for v in out])) if fetch else None

    def debug_f32(self, op: int, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        check(load().gms_debug_f32(self._h, op, ptr(a), ptr(out), a.size))
        return out

    # Java names
    integrateObservation = integrate_observation
    applyMeasurement = apply_measurement
    computeLikelihoodMap = compute_likelihood_map
    probabilityOf = probability_of
    findBestPose = find_best_pose
    getRawAt = get_raw_at
    getProbAt = get_prob_at
    pointInMap = point_in_map


class ParticleFilter:
    """ParticleFilter(numberOfParticles) (J/slam/ParticleFilter.java:43) bound to a GridMap."""

    def __init__(self, grid_map: GridMap, n: int):
        self.map = grid_map
        self.n = n
        self.n_maps = grid_map.n_maps
        self._h = C.c_void_p()
        check(load().gms_pf_create(grid_map._h, n, C.byref(self._h)))
        self.offset, self.n_global = 0, n
        grid_map._filters.add(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().gms_pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pshape(self, last=()):
        return ((self.n,) if self.n_maps == 1 else (self.n_maps, self.n)) + tuple(last)

    def set_shard(self, offset: int, n_global: int):
        check(load().gms_pf_set_shard(self._h, offset, n_global))
        self.offset, self.n_global = offset, n_global

    # -- getParticles() ---------------------------------------------------------------------------
    def set_poses(self, xytheta):
        a = np.ascontiguousarray(xytheta, dtype=np.float32)
        assert a.size == self.n_maps * self.n * 3
        check(load().gms_pf_set_poses(self._h, ptr(a)))

    def get_poses(self) -> np.ndarray:
        out = np.empty(self._pshape((3,)), dtype=np.float32)
        check(load().gms_pf_get_poses(self._h, ptr(out)))
        return out

    def set_weights(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        assert a.size == self.n_maps * self.n
        check(load().gms_pf_set_weights(self._h, ptr(a)))

    def get_weights(self) -> np.ndarray:
        out = np.empty(self._pshape(), dtype=np.float64)
        check(load().gms_pf_get_weights(self._h, ptr(out)))
        return out

    def get_log_weights(self) -> np.ndarray:
        out = np.empty(self._pshape(), dtype=np.float64)
        check(load().gms_pf_get_log_weights(self._h, ptr(out)))
        return out

    def get_particles(self):
        """(poses [n][3], weights [n]) -- the fields of ParticleFilter.Particle (ParticleFilter.java:21-38)."""
        return self.get_poses(), self.get_weights()

    # -- SLAM.update pieces ------------------------------------------------------------------------
    def score(self, obs):
        """weight[i] = probabilityOf(map, obs, pose[i]) (SLAM.java:99)."""
        b, B = self.map._beam_args(obs)
        check(load().gms_pf_score(self._h, ptr(b), B))

    def set_poses_dev(self, dev_xytheta: int):
        check(load().gms_pf_set_poses_dev(self._h, C.c_void_p(dev_xytheta)))

    def score_dev(self, dev_beams: int, B: int):
        check(load().gms_pf_score_dev(self._h, C.c_void_p(dev_beams), B))

    def _r01(self, r01):
        """the draws as a contiguous float64 [n_maps] (an array of that shape passes through untouched: the hot loop's case)"""
        if isinstance(r01, np.ndarray) and r01.dtype == np.float64 and r01.shape == (self.n_maps,) and r01.flags.c_contiguous:
            return r01
        return np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))

    def slam_update_dev(self, dev_xytheta: int, dev_beams: int, B: int, r01, fraction: float = 0.5, integrate: bool = True):
        """SLAM.update + `if (neff < fraction*N) resample()` in one call on device-resident inputs."""
        r = self._r01(r01)
        check(load().gms_slam_update_dev(self._h, C.c_void_p(dev_xytheta or 0), C.c_void_p(dev_beams), B, ptr(r), fraction,
                                         1 if integrate else 0))

    def slam_update_u_dev(self, d_center: float, d_theta: float, seed: int, sequence: int, dev_beams: int, B: int, r01,
                          fraction: float = 0.5, integrate: bool = True):
        """SLAM.update(z, u) with the motion-model sample inside the scoring launch (= sample_motion + slam_update_dev(0, ...))."""
        r = self._r01(r01)
        check(load().gms_slam_update_u_dev(self._h, d_center, d_theta, seed, sequence, C.c_void_p(dev_beams), B, ptr(r), fraction,
                                           1 if integrate else 0))

    def slam_frame(self, angles, distances, hits, d_center: float, d_theta: float, seed: int, sequence: int, r01,
                   fraction: float = 0.5, integrate: bool = True):
        """One recorded revolution (GridMapApp.java:133-192) in one call: de-skew, motion-model sample, scan step."""
        a = np.ascontiguousarray(angles, dtype=np.float64)
        d = np.ascontiguousarray(distances, dtype=np.float64)
        h = np.ascontiguousarray(hits, dtype=np.uint8)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        check(load().gms_slam_frame(self._h, ptr(a), ptr(d), ptr(h), a.size, d_center, d_theta, seed, sequence, ptr(r), fraction,
                                    1 if integrate else 0))

    def slam_update(self, poses, obs, r01, fraction: float = 0.5, integrate: bool = True, fetch: bool = False):
        """SLAM.update + conditional resample with HOST inputs (poses may be None)."""
        b, B = self.map._beam_args(obs)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        p = None if poses is None else np.ascontiguousarray(poses, dtype=np.float32)
        arr = (GmsPfStats * self.n_maps)() if fetch else None
        check(load().gms_slam_update(self._h, None if p is None else ptr(p), ptr(b), B, ptr(r), fraction, 1 if integrate else 0, arr))
        return self._stats(arr) if fetch else None

    def _stats(self, arr):
        out = [dict(weight_sum=s.weight_sum, neff=s.neff, strongest=s.strongest, n_zero=s.n_zero,
                    max_log_weight=s.max_log_weight) for s in arr]
        return out[0] if self.n_maps == 1 else out

    def normalize(self, fetch: bool = True):
        """weightSum / strongest / weight /= weightSum / Neff (SLAM.java:87-129)."""
        if not fetch:
            check(load().gms_pf_normalize(self._h, None))
            return None
        arr = (GmsPfStats * self.n_maps)()
        check(load().gms_pf_normalize(self._h, arr))
        return self._stats(arr)

    def stats(self):
        arr = (GmsPfStats * self.n_maps)()
        check(load().gms_pf_get_stats(self._h, arr))
        return self._stats(arr)

    def weighted_pose(self) -> np.ndarray:
        """getWeightedPose() (SLAM.java:165-178)."""
        out = np.empty((self.n_maps, 3), dtype=np.float32)
        check(load().gms_pf_weighted_pose(self._h, ptr(out)))
        return out[0] if self.n_maps == 1 else out

    def set_log_normalize(self, on: bool = True):
        """Opt-in (not in the reference): normalise from the log-weights, weight = exp(logw - max logw), instead of the plain
        product that underflows at hundreds of beams (gms_pf_set_log_normalize)."""
        check(load().gms_pf_set_log_normalize(self._h, 1 if on else 0))

    def set_reference_order(self, on: bool = True):
        """the audit path (gms_pf_set_reference_order): the scan's product, weightSum and the cumulative weights each as ONE chain in
        the reference's order (tests; slow)"""
        check(load().gms_pf_set_reference_order(self._h, 1 if on else 0))

    def resample(self, r01=None, want_indices: bool = False):
        """resample() (SLAM.java:133-153); r01 stands for Math.random()."""
        if r01 is None:
            r01 = np.random.random(self.n_maps)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        if not want_indices:
            check(load().gms_pf_resample(self._h, ptr(r), None, None))
            return None
        idx = np.empty(self._pshape(), dtype=np.int32)
        amb = np.empty(self.n_maps, dtype=np.int32)
        check(load().gms_pf_resample(self._h, ptr(r), ptr(idx), ptr(amb)))
        return idx, (int(amb[0]) if self.n_maps == 1 else amb)

    def resample_if(self, r01, fraction: float = 0.5):
        """if (neff < fraction * N) resample()  (J/app/GridMapApp.java:185-186), decided on the device."""
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        check(load().gms_pf_resample_if(self._h, ptr(r), fraction))

    def last_resample_indices(self) -> np.ndarray:
        """source slot of every particle after the last resampling step (also the one inside a scan step)."""
        idx = np.empty(self._pshape(), dtype=np.int32)
        check(load().gms_pf_last_resample_indices(self._h, ptr(idx)))
        return idx

    def did_resample(self):
        f = np.empty(self.n_maps, dtype=np.int32)
        check(load().gms_pf_did_resample(self._h, ptr(f)))
        return bool(f[0]) if self.n_maps == 1 else f.astype(bool)

    def last_step(self) -> dict:
        """Device-side record of the last normalise / scan step: the weighted pose of the SCORED population (what a
        fused scan step integrated the scan at), the strongest particle's pose, did_resample, n_ambiguous."""
        wp = np.empty((self.n_maps, 3), dtype=np.float32)
        sp = np.empty((self.n_maps, 3), dtype=np.float32)
        did = np.empty(self.n_maps, dtype=np.int32)
        amb = np.empty(self.n_maps, dtype=np.int32)
        check(load().gms_pf_last_step(self._h, ptr(wp), ptr(sp), ptr(did), ptr(amb)))
        if self.n_maps == 1:
            return dict(weighted_pose=wp[0], strongest_pose=sp[0], did_resample=bool(did[0]), n_ambiguous=int(amb[0]))
        return dict(weighted_pose=wp, strongest_pose=sp, did_resample=did.astype(bool), n_ambiguous=amb)

    def sample_motion(self, d_center: float, d_theta: float, seed: int, sequence: int):
        """pose[i] = sampleMotionModel(pose[i], u) (SLAM.java:155-163 -> Odometry.apply, Odometry.java:77-96)."""
        check(load().gms_pf_sample_motion(self._h, d_center, d_theta, seed, sequence))

    def set_refine(self, on: bool = True):
        """scan steps (slam_update*) run findBestPose on every particle before weighting it (SLAM.java:96-97)."""
        check(load().gms_pf_set_refine(self._h, 1 if on else 0))

    def refine_poses(self, obs):
        """pose[i] = findBestPose(map, obs, pose[i]) (GridMap.java:319-346)."""
        b, B = self.map._beam_args(obs)
        check(load().gms_pf_refine_poses(self._h, ptr(b), B))

    # -- multi-GPU plumbing (device pointers; see distributed.py) ----------------------------------
    def partials_len(self) -> int:
        n = C.c_int64()
        check(load().gms_pf_partials_len(self._h, C.byref(n)))
        return n.value

    def local_partials(self, dev_ptr: int):
        check(load().gms_pf_local_partials(self._h, C.c_void_p(dev_ptr)))

    def apply_partials(self, dev_partials: int, dev_packed: int):
        check(load().gms_pf_apply_partials(self._h, C.c_void_p(dev_partials), C.c_void_p(dev_packed)))

    def stats_from_partials(self, dev_partials: int):
        check(load().gms_pf_stats_from_partials(self._h, C.c_void_p(dev_partials)))

    def pack(self, dev_packed: int):
        check(load().gms_pf_pack(self._h, C.c_void_p(dev_packed)))

    def import_global(self, dev_packed_global: int):
        check(load().gms_pf_import_global(self._h, C.c_void_p(dev_packed_global)))

    # -- sharded filter with the exchanges inside the library (RCCL; see distributed.RcclComm) --------
    def normalize_sharded_begin(self, comm):
        check(load().gms_pf_normalize_sharded_begin(self._h, comm._h))

    def normalize_sharded_end(self, comm):
        check(load().gms_pf_normalize_sharded_end(self._h, comm._h))

    def slam_update_sharded_dev(self, comm, dev_xytheta: int, dev_beams: int, B: int, r01, fraction: float = 0.5,
                                integrate: bool = True):
        """slam_update_dev for a sharded filter: both collectives happen inside the call."""
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        check(load().gms_slam_update_sharded_dev(self._h, comm._h, C.c_void_p(dev_xytheta or 0), C.c_void_p(dev_beams), B,
                                                 ptr(r), fraction, 1 if integrate else 0))

    def slam_update_sharded_begin_dev(self, dev_xytheta: int, dev_beams: int, B: int):
        check(load().gms_slam_update_sharded_begin_dev(self._h, C.c_void_p(dev_xytheta or 0), C.c_void_p(dev_beams), B))

    def gather_buffers(self):
        """(packed_global ptr, bytes per rank, partials_global ptr, doubles per rank): both are all-gathered in place."""
        a, b = C.c_void_p(), C.c_void_p()
        na, nb = C.c_int64(), C.c_int64()
        check(load().gms_pf_gather_buffers(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return a.value, na.value, b.value, nb.value

    def slam_update_sharded_end_dev(self, dev_beams: int, B: int, r01, fraction: float = 0.5, integrate: bool = True):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(r01, dtype=np.float64), (self.n_maps,)))
        check(load().gms_slam_update_sharded_end_dev(self._h, C.c_void_p(dev_beams), B, ptr(r), fraction, 1 if integrate else 0))

    getParticles = get_particles
    getWeightedPose = weighted_pose


class SLAM:
    """The particle filter the reference actually runs (J/slam/SLAM.java), with ONE shared map and N
    poses scored against it, the pose proposal being an input."""

    def __init__(self, width=6.0, height=6.0, resolution=0.05, position=(-3.0, -3.0), num_particles=500,
                 device: int = 0, max_beams: int = 0):
        self.grid_map = GridMap(width, height, resolution, position, device=device, max_beams=max_beams)   # SLAM.java:57
        self.num_particles = num_particles                                                                  # SLAM.java:50
        self.pf = ParticleFilter(self.grid_map, num_particles)
        self.strongest = 0
        self.neff = float(num_particles)
        self.grid_map.compute_likelihood_map()

    def reset(self):
        """reset() (SLAM.java:65-77)."""
        self.grid_map.reset()
        self.grid_map.compute_likelihood_map()
        self.pf.close()
        self.pf = ParticleFilter(self.grid_map, self.num_particles)

    def update(self, z: Observation, poses=None, d_theta: float = 0.0, refine: bool = False) -> float:
        """update(z, u) (SLAM.java:80-131).  `poses` are the motion-model samples (an input here);
        d_theta is u.dTheta for the skip-update rule (:82)."""
        skip_update = abs(d_theta) > math.radians(30)                   # :82
        if poses is not None:
            self.pf.set_poses(poses)                                    # :90
        if refine:
            self.pf.refine_poses(z)                                     # :96
        self.pf.score(z)                                                # :99
        st = self.pf.normalize()                                        # :100-124
        self.strongest, self.neff = st["strongest"], st["neff"]
        if not skip_update:
            self.grid_map.update_at(z, self.pf, strongest=False)        # :105 + :93 for the next scan
        return self.neff

    def resample(self, r01=None):
        self.pf.resample(r01)                                           # :133-153

    def get_weighted_pose(self):
        return self.pf.weighted_pose()                                  # :165-178

    def calculate_neff(self) -> float:
        return self.pf.stats()["neff"]                                  # :180-190

    def get_particles(self):
        return self.pf.get_particles()

    def get_strongest_particle(self) -> int:
        return self.strongest

    def get_grid_map(self) -> GridMap:
        return self.grid_map

    getWeightedPose = get_weighted_pose
    calculateNeff = calculate_neff
    getParticles = get_particles
    getStrongestParticle = get_strongest_particle
    getGridMap = get_grid_map


class _Borrowed:
    """a handle owned by another object (never destroyed from here)"""

    def close(self):
        self._h = C.c_void_p()


class _BorrowedMap(_Borrowed, GridMap):
    def __init__(self, handle, params):
        self.params = params
        self._h = handle
        W, H, M = C.c_int32(), C.c_int32(), C.c_int32()
        check(load().gms_map_get_size(self._h, C.byref(W), C.byref(H), C.byref(M)))
        self.W, self.H, self.n_maps = W.value, H.value, M.value
        self._filters = weakref.WeakSet()


class _BorrowedFilter(_Borrowed, ParticleFilter):
    def __init__(self, handle, grid_map, n):
        self.map = grid_map
        self.n = n
        self.n_maps = 1
        self._h = handle
        self.offset, self.n_global = 0, n


class SLAMParticleMaps:
    """SLAM as the reference has it (J/slam/SLAM.java:26-204): num_particles particles, each with its own pose, weight AND
    GridMapData -- update() scores a particle against its own likelihood field and integrates the scan into its own map at its own
    pose (:88-107), resample() deep-copies the surviving particles' maps (:41-45).  (`SLAM` above is the shared-map filter that
    BASELINE's configurations need.)  findBestPoseOptim (:97) is left out; the motion-model draw is Philox(seed; particle, sequence)."""

    def __init__(self, width=6.0, height=6.0, resolution=0.05, position=(-3.0, -3.0), num_particles=500, device: int = 0,
                 max_beams: int = 0, kernel=None):
        L = load()
        p = GmsParams()
        check(L.gms_params_default(C.byref(p), width, height, resolution, position[0], position[1]))   # SLAM.java:57
        p.device = device
        p.max_beams = max_beams
        if kernel is not None:
            k = np.asarray(kernel, dtype=np.float64)
            p.ktaps = k.size
            for i, t in enumerate(k):
                p.kernel[i] = float(t)
        self.params = p
        self.num_particles = int(num_particles)                                                          # :50
        self._h = C.c_void_p()
        check(L.gms_slam_create(C.byref(p), self.num_particles, C.byref(self._h)))
        mh, ph = C.c_void_p(), C.c_void_p()
        check(L.gms_slam_handles(self._h, C.byref(mh), C.byref(ph)))
        self.grid_map = _BorrowedMap(mh, p)                     # getGridMap() (:200); its GridMapData receives calculate_combined()
        self.pf = _BorrowedFilter(ph, self.grid_map, self.num_particles)
        self.W, self.H = self.grid_map.W, self.grid_map.H
        self.strongest = 0
        self.neff = float(num_particles)
        self.sequence = 0

    def _init_shard(self, width, height, resolution, position, n_local, offset, n_global, device=0, max_beams=0):
        """one rank's block of a sharded filter (gms_slam_create_shard): distributed.SlamShardOps"""
        L = load()
        p = GmsParams()
        check(L.gms_params_default(C.byref(p), width, height, resolution, position[0], position[1]))
        p.device = device or 0
        p.max_beams = max_beams
        self.params = p
        self.num_particles = int(n_local)
        self._h = C.c_void_p()
        check(L.gms_slam_create_shard(C.byref(p), int(n_local), int(offset), int(n_global), C.byref(self._h)))
        mh, ph = C.c_void_p(), C.c_void_p()
        check(L.gms_slam_handles(self._h, C.byref(mh), C.byref(ph)))
        self.grid_map = _BorrowedMap(mh, p)
        self.pf = _BorrowedFilter(ph, self.grid_map, self.num_particles)
        self.pf.offset, self.pf.n_global = int(offset), int(n_global)
        self.W, self.H = self.grid_map.W, self.grid_map.H
        self.strongest, self.neff, self.sequence = 0, float(n_global), 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.grid_map.close(); self.pf.close()
            check(load().gms_slam_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(load().gms_slam_reset(self._h))                                                            # :65-77

    def set_refine(self, on: bool = True):
        """update() runs GridMap.findBestPose (J/slam/GridMap.java:319-346) for every particle against its own likelihood field
        before weighting it (SLAM.java:96; the reference calls findBestPoseOptim, :97, and keeps this search commented out beside it)"""
        check(load().gms_slam_set_refine(self._h, int(bool(on))))

    def update(self, z, odometry=None, seed: int = 0, sequence: Optional[int] = None, fetch: bool = True, sample_motion: bool = True):
        """update(z, u) (:80-131); odometry = (dCenter, dTheta) or None (= (0, 0) and no motion sample: `u == null` in
        sampleMotionModel, :159); sample_motion = False keeps the poses (dTheta still decides skipUpdate, :82); returns Neff"""
        b = _beams_of(z)
        have = odometry is not None and sample_motion
        dc, dt = (odometry if odometry is not None else (0.0, 0.0))
        if sequence is None:
            sequence = self.sequence
            self.sequence += 1
        st = GmsPfStats()
        check(load().gms_slam_update_per_particle(self._h, ptr(b), len(b), int(have), float(dc), float(dt), int(seed), int(sequence),
                                                  C.byref(st) if fetch else None))
        if fetch:
            self.strongest, self.neff = st.strongest, st.neff
            self.last_stats = {"weight_sum": st.weight_sum, "neff": st.neff, "strongest": st.strongest, "n_zero": st.n_zero,
                               "max_log_weight": st.max_log_weight}
            return st.neff
        return None

    def update_dev(self, dev_beams: int, B: int, odometry=None, seed: int = 0, sequence: int = 0, sample_motion: bool = True):
        have = odometry is not None and sample_motion
        dc, dt = (odometry if odometry is not None else (0.0, 0.0))
        check(load().gms_slam_update_per_particle_dev(self._h, C.c_void_p(dev_beams), B, int(have), float(dc), float(dt), int(seed),
                                                      int(sequence), None))

    def resample(self, r01: Optional[float] = None, want_indices: bool = False):
        """resample() (:133-153): r01 stands for Math.random()"""
        r = float(np.random.random() if r01 is None else r01)
        idx = np.empty(self.num_particles, dtype=np.int32) if want_indices else None
        amb = C.c_int32(0)
        check(load().gms_slam_resample_maps(self._h, r, ptr(idx) if want_indices else None, C.byref(amb) if want_indices else None))
        return (idx, amb.value) if want_indices else None

    def resample_if(self, r01: Optional[float] = None, fraction: float = 0.5):
        """`if (neff < fraction * n) resample()` (GridMapApp.java:185-186) decided on the device from the last update's Neff: no host
        round trip (update_dev + resample_if is one revolution); pf.last_resample_indices() tells afterwards what happened"""
        r = float(np.random.random() if r01 is None else r01)
        check(load().gms_slam_resample_maps_if(self._h, r, float(fraction)))

    def get_weighted_pose(self) -> np.ndarray:
        return self.pf.weighted_pose()                                                                   # :165-178

    def calculate_neff(self) -> float:
        return self.pf.stats()["neff"]                                                                   # :180-190

    def get_particles(self):
        """(poses [n][3], weights [n]); the maps: map_of(i) / maps()"""
        return self.pf.get_particles()                                                                   # :192

    def set_poses(self, xytheta):
        self.pf.set_poses(xytheta)

    def map_of(self, i: int, likelihood: bool = False) -> np.ndarray:
        """Particle i's logData (or likelihoodData) as [H][W] (Particle.m, :33)"""
        out = np.empty((self.H, self.W), dtype=np.float64)
        check(load().gms_slam_download_map(self._h, int(i), None if likelihood else ptr(out), ptr(out) if likelihood else None))
        return out

    def maps(self, likelihood: bool = False) -> np.ndarray:
        out = np.empty((self.num_particles, self.H, self.W), dtype=np.float64)
        check(load().gms_slam_download_maps(self._h, None if likelihood else ptr(out), ptr(out) if likelihood else None))
        return out

    def set_map(self, i: int, log=None, lik=None):
        lg = None if log is None else np.ascontiguousarray(log, dtype=np.float64)
        lk = None if lik is None else np.ascontiguousarray(lik, dtype=np.float64)
        for name, a in (("log", lg), ("lik", lk)):          # (the library copies W * H doubles from the pointer it is given)
            if a is not None and a.size != self.W * self.H:
                raise ValueError(f"set_map: {name} has {a.size} values, the map has {self.W} x {self.H} cells")
        check(load().gms_slam_upload_map(self._h, int(i), None if lg is None else ptr(lg), None if lk is None else ptr(lk)))

    def calculate_combined(self) -> np.ndarray:
        """GridMapApp.calculateCombined (J/app/GridMapApp.java:439-458): the combined logData [H][W]; the likelihood field of it is
        grid_map.download_likelihood()"""
        check(load().gms_slam_combined(self._h))
        return self.grid_map.download_log()

    def trace_scan(self, i: int, z, cap: int = 0):
        """(cells [B][cap][2], classes [B][cap], counts [B]): the cell walk of integrateObservation for particle i at its current pose
        as the update kernel walks and classifies it (prior-class visits included), in walk order; nothing is written to a map"""
        b = _beams_of(z)
        cap = int(cap) if cap > 0 else self.W + self.H + 8
        cells = np.zeros((len(b), cap, 2), dtype=np.int32)
        cls = np.zeros((len(b), cap), dtype=np.uint8)
        counts = np.zeros(len(b), dtype=np.int32)
        check(load().gms_slam_trace_scan(self._h, int(i), ptr(b), len(b), ptr(cells), ptr(cls), cap, ptr(counts)))
        return cells, cls, counts

    def maps_copied(self) -> int:
        v = C.c_int64(0)
        check(load().gms_slam_copies(self._h, C.byref(v)))
        return int(v.value)

    def get_strongest_particle(self) -> int:
        return self.strongest                                                                            # :196

    def get_grid_map(self) -> GridMap:
        return self.grid_map                                                                             # :200

    getWeightedPose = get_weighted_pose
    calculateNeff = calculate_neff
    getParticles = get_particles
    getStrongestParticle = get_strongest_particle
    getGridMap = get_grid_map
