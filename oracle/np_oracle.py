"""Second, independently written restatement of the reference hot path (numpy / plain Python).

TEST INFRASTRUCTURE ONLY (same rules as oracle/gms_oracle.h).  PARITY UNPINNED: there is no
runnable reference and the reference has no fixtures; this file exists so that two separate
readings of the Java sources can be compared bit for bit (tests/test_oracle_cross.py).  It was
written from the Java, not from gms_oracle.c, and is structured differently on purpose
(generator-based ray walk, vectorised blur and scoring).

Citations: J/ = /root/reference/java/GridMapGL/src/main/java/com/fmsz/gridmapgl/.
Scalars are np.float32 / np.float64 so every operation rounds as the JVM's would; no fused
multiply-add exists in numpy scalar arithmetic.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
f64 = np.float64
INT_MAX = 2**31 - 1
INT_MIN = -(2**31)


def jint(d) -> int:
    """Java (int) narrowing of a double (JLS 5.1.3)."""
    d = float(d)
    if math.isnan(d):
        return 0
    if d >= INT_MAX:
        return INT_MAX
    if d <= INT_MIN:
        return INT_MIN
    return int(d)  # truncation toward zero


def wrap32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


# ---------------------------------------------------------------- J/app/Util.java
def log_odds(p: float) -> float:
    # Util.java:35-37 -- Math.log(odds / (1.0f - odds))
    return math.log(p / (1.0 - p))


def gaussian_kernel(sigma: float, size: int) -> np.ndarray:
    # Util.java:428-455
    vals = []
    norm = 1.0 / (math.sqrt(2 * math.pi) * sigma)
    coeff = 2 * sigma * sigma
    total = 0.0
    for x in range(-size, size + 1):
        g = norm * math.exp((-x * x) / coeff)
        vals.append(g)
        total += g
    return np.array([v / total for v in vals], dtype=np.float64)


P_FREE = float(f32(0.30))   # SensorModel.java:23
P_OCC = float(f32(0.9))     # :24
P_PRIOR = float(f32(0.5))   # :25


class NpGrid:
    # GridMap ctor, GridMap.java:80-100
    def __init__(self, width, height, resolution, pos_x, pos_y):
        self.res = f32(resolution)
        self.px = f32(pos_x)
        self.py = f32(pos_y)
        self.W = jint(math.ceil(float(f32(width) / self.res)))
        self.H = jint(math.ceil(float(f32(height) / self.res)))
        sigma = math.sqrt(0.05 / float(self.res))
        self.kernel = gaussian_kernel(sigma, jint(math.ceil(sigma * 3)))
        self.l = (log_odds(P_FREE), log_odds(P_PRIOR), log_odds(P_OCC))
        self.z_hit = 0.9
        self.z_random = 1 - self.z_hit
        self.max_range = f32(10.0)

    # ------------------------------------------------------------ RayIterator.java:65-130
    def ray_cells(self, x0, y0, x1, y1, extra=2):
        """Generator over the visited cells, in order."""
        x0, y0, x1, y1 = f32(x0), f32(y0), f32(x1), f32(y1)
        dx = abs(f32(x1 - x0))
        dy = abs(f32(y1 - y0))
        x = jint(math.floor(float(x0)))
        y = jint(math.floor(float(y0)))
        n = 1 + extra
        with np.errstate(invalid="ignore", over="ignore"):
            if dx == 0:
                x_inc = 0
                error = f32(np.inf)
            elif x1 > x0:
                x_inc = 1
                n = wrap32(n + jint(math.floor(float(x1)) - x))
                error = f32((math.floor(float(x0)) + 1 - float(x0)) * float(dy))
            else:
                x_inc = -1
                n = wrap32(n + wrap32(x - jint(math.floor(float(x1)))))
                error = f32((float(x0) - math.floor(float(x0))) * float(dy))
            if dy == 0:
                y_inc = 0
                error = f32(error - f32(np.inf))
            elif y1 > y0:
                y_inc = 1
                n = wrap32(n + wrap32(jint(math.floor(float(y1))) - y))
                error = f32(float(error) - (math.floor(float(y0)) + 1 - float(y0)) * float(dx))
            else:
                y_inc = -1
                n = wrap32(n + wrap32(y - jint(math.floor(float(y1)))))
                error = f32(float(error) - (float(y0) - math.floor(float(y0))) * float(dx))
            while n > 0 and 0 <= x < self.W and 0 <= y < self.H:
                yield x, y
                if error > 0:
                    y = wrap32(y + y_inc)
                    error = f32(error - dx)
                else:
                    x = wrap32(x + x_inc)
                    error = f32(error + dy)
                n -= 1

    # ------------------------------------------------------------ SensorModel.java:31-41
    @staticmethod
    def sensor_class(cur, measured, hit, tol=f32(2.0)):
        half = f32(tol / f32(2))
        if not hit:
            return 0 if cur < measured else 1
        if cur < f32(measured - half):
            return 0
        if cur > f32(measured + half):
            return 1
        return 2

    # ------------------------------------------------------------ Transform.java:13-32
    @staticmethod
    def trig(theta):
        t = float(f32(theta))
        return float(f32(math.cos(t))), float(f32(math.sin(t)))

    # ------------------------------------------------------------ GridMap.java:173-228
    def scan_rays(self, beams, pose):
        c, s = self.trig(pose[2])
        px, py = float(f32(pose[0])), float(f32(pose[1]))
        res = float(self.res)
        with np.errstate(invalid="ignore"):
            sx = f32(((0.0 * c - 0.0 * s + px) - float(self.px)) / res)
            sy = f32(((0.0 * s + 0.0 * c + py) - float(self.py)) / res)
        rays = []
        for b in beams:
            lx, ly = float(b["local_x"]), float(b["local_y"])
            ex = f32(((lx * c - ly * s + px) - float(self.px)) / res)
            ey = f32(((lx * s + ly * c + py) - float(self.py)) / res)
            measured = f32(f32(b["distance"]) / self.res)
            rays.append((sx, sy, ex, ey, measured, bool(b["hit"])))
        return rays

    def apply_measurement(self, log, sx, sy, ex, ey, measured, hit):
        half = f32(0.5)
        cells, classes = [], []
        for (cx, cy) in self.ray_cells(f32(sx + half), f32(sy + half), f32(ex + half), f32(ey + half), 2):
            dX = f32(sx - f32(f32(cx) + half))
            dY = f32(sy - f32(f32(cy) + half))
            dist = f32(math.sqrt(float(f32(f32(dX * dX) + f32(dY * dY)))))
            k = self.sensor_class(dist, measured, hit)
            if log is not None:
                log[cx + cy * self.W] += self.l[k]
            cells.append((cx, cy))
            classes.append(k)
        return cells, classes

    def integrate(self, log, beams, pose):
        visits = 0
        for (sx, sy, ex, ey, measured, hit) in self.scan_rays(beams, pose):
            c, _ = self.apply_measurement(log, sx, sy, ex, ey, measured, hit)
            visits += len(c)
        return visits

    # ------------------------------------------------------------ GridMap.java:233-250, Util.java:378-426
    def build_likelihood(self, log):
        W, H = self.W, self.H
        lg = np.asarray(log, dtype=np.float64).reshape(H, W)
        thr = self.l[1]
        prob = np.where(lg > thr, 1.0, np.where(lg < thr, 0.0, 0.5))
        k = (len(self.kernel) - 1) // 2
        hz = np.zeros((H, W))
        for i in range(-k, k + 1):           # taps in order; out-of-range taps skipped
            lo, hi = max(0, -i), min(W, W - i)
            if lo < hi:
                hz[:, lo:hi] = hz[:, lo:hi] + self.kernel[i + k] * prob[:, lo + i:hi + i]
        out = np.zeros((H, W))
        for i in range(-k, k + 1):
            lo, hi = max(0, -i), min(H, H - i)
            if lo < hi:
                out[lo:hi, :] = out[lo:hi, :] + self.kernel[i + k] * hz[lo + i:hi + i, :]
        return out.reshape(-1)

    # ------------------------------------------------------------ GridMap.java:259-294
    def score(self, lik, beams, poses):
        poses = np.asarray(poses, dtype=np.float32).reshape(-1, 3)
        N = len(poses)
        th = poses[:, 2].astype(np.float64)
        c = np.cos(th).astype(np.float32).astype(np.float64)
        s = np.sin(th).astype(np.float32).astype(np.float64)
        # numpy's vector cos/sin may differ from libm by an ulp; recompute with math.* to be safe
        c = np.array([float(f32(math.cos(float(t)))) for t in th])
        s = np.array([float(f32(math.sin(float(t)))) for t in th])
        px = poses[:, 0].astype(np.float64)
        py = poses[:, 1].astype(np.float64)
        res = float(self.res)
        gpx, gpy = float(self.px), float(self.py)
        prod = np.ones(N)
        inv_max = 1.0 / float(self.max_range)
        c0 = self.z_random * 1.0 / float(self.max_range)
        with np.errstate(invalid="ignore", over="ignore"):
            for b in beams:
                if not b["hit"]:
                    continue
                lx, ly = float(b["local_x"]), float(b["local_y"])
                qx = ((lx * c - ly * s + px) - gpx) / res
                qy = ((lx * s + ly * c + py) - gpy) / res
                gx = np.where(np.isnan(qx), 0.0, np.clip(np.trunc(qx), INT_MIN, INT_MAX)).astype(np.int64)
                gy = np.where(np.isnan(qy), 0.0, np.clip(np.trunc(qy), INT_MIN, INT_MAX)).astype(np.int64)
                inside = ~((gx < 0) | (gy < 0) | (gx >= self.W) | (gy >= self.H))
                idx = np.where(inside, gx + gy * self.W, 0)
                val = lik[idx]
                f = np.where(val == 0.5, inv_max, self.z_hit * val + c0)
                prod = prod * np.where(inside, f, 1.0)
        return prod


# ---------------------------------------------------------------- J/slam/SLAM.java
def normalize(weights):
    # SLAM.java:87-121
    s = 0.0
    best = None
    for i, w in enumerate(weights):
        s += float(w)
        if best is None or float(w) > float(weights[best]):
            best = i
    return np.array([float(w) / s for w in weights]) if s == s else None, s, best


def neff(weights):
    # SLAM.java:180-190
    s = 0.0
    for w in weights:
        s += float(w)
    q = 0.0
    for w in weights:
        q += (float(w) / s) * (float(w) / s)
    return 1.0 / q


def angle_constrain(a):
    # MathUtil.java:65-72
    while a < math.pi:
        a += math.pi * 2
    while a > math.pi:
        a -= math.pi * 2
    return a


def weighted_pose(poses, weights):
    # SLAM.java:165-178
    xs = ys = ts = ws = 0.0
    for p, w in zip(np.asarray(poses, dtype=np.float32).reshape(-1, 3), weights):
        w = float(w)
        xs += float(p[0]) * w
        ys += float(p[1]) * w
        ts += angle_constrain(float(p[2])) * w
        ws += w
    return np.array([f32(xs / ws), f32(ys / ws), f32(ts / ws)], dtype=np.float32)


def resample_indices(weights, r01):
    # SLAM.java:133-153
    N = len(weights)
    r = r01 * 1.0 / N
    c = float(weights[0])
    i = 0
    out = []
    for m in range(1, N + 1):
        U = r + (m - 1) * 1.0 / N
        while U > c and i < N - 1:
            i += 1
            c += float(weights[i])
        out.append(i)
    return np.array(out, dtype=np.int32)


# ---------------------------------------------------------------- J/slam/SLAM.java as a class, J/slam/Odometry.java:77-96
def odometry_apply(pose, d_center, d_theta, z0, z1):
    """Odometry.apply(p) (Odometry.java:77-96) given the two standard normal variates of its draws (the reference's unseeded
    Well1024a stream is not reproducible: the variates are an input, see orc_philox_normals)."""
    d_center_sd = (0.01 + abs(d_center) * 0.05) / 2                  # :63
    d_theta_sd = 5 * (math.pi / 180.0) + 0.1 * abs(d_theta)          # :64
    d = d_center + d_center_sd * z0                                  # :80
    theta = d_theta + d_theta_sd * z1                                # :81
    th = f32(angle_constrain(float(f32(pose[2])) + theta))           # :92
    c, s = NpGrid.trig(th)                                           # MathUtil.cos(float) / sin(float)
    x = f32(float(f32(pose[0])) + c * d)                             # :93 (float += double: widened, added, narrowed)
    y = f32(float(f32(pose[1])) + s * d)                             # :94
    return np.array([x, y, th], dtype=np.float32)


class NpSlam:
    """SLAM.java:26-204, one object per Particle as in the reference (a dict {weight, pose, log, lik}); findBestPoseOptim (:97) is
    left out (see gms_oracle.h).  normals(i, sequence) -> (z0, z1) supplies the motion model's variates."""

    def __init__(self, grid: NpGrid, n: int, normals=None):
        self.g = grid
        self.n = n
        self.normals = normals
        self.strongest = None
        self.reset()

    def _new_map(self, other=None):
        # GridMap.createMapData (GridMap.java:106-124)
        cells = self.g.W * self.g.H
        if other is None:
            return np.full(cells, self.g.l[1], dtype=np.float64), np.zeros(cells, dtype=np.float64)
        return other[0].copy(), other[1].copy()

    def reset(self):
        # :65-77
        self.particles = []
        for _ in range(self.n):
            log, lik = self._new_map()
            self.particles.append({"weight": 1.0 / self.n, "pose": np.zeros(3, dtype=np.float32), "log": log, "lik": lik})
        self.strongest = self.particles[0]

    def update(self, beams, odometry=None, sequence=0):
        # :80-131
        d_theta = odometry[1] if odometry is not None else 0.0
        skip_update = abs(d_theta) > (math.pi / 180.0) * 30          # :82
        self.strongest = None
        weight_sum = 0.0
        for i, p in enumerate(self.particles):
            if odometry is not None:                                 # :90, :155-163
                z0, z1 = self.normals(i, sequence)
                p["pose"] = odometry_apply(p["pose"], odometry[0], odometry[1], z0, z1)
            p["lik"] = self.g.build_likelihood(p["log"])             # :93
            p["weight"] = float(self.g.score(p["lik"], beams, p["pose"].reshape(1, 3))[0])     # :99
            weight_sum += p["weight"]                                # :100
            if not skip_update:
                self.g.integrate(p["log"], beams, p["pose"])         # :105
            if self.strongest is None or p["weight"] > self.strongest["weight"]:   # :110-115
                self.strongest = p
        for p in self.particles:                                     # :120-121
            p["weight"] /= weight_sum
        return self.neff()

    def resample(self, r01):
        # :133-153
        new = []
        r = r01 * 1.0 / self.n
        c = self.particles[0]["weight"]
        i = 0
        idx = []
        for m in range(1, self.n + 1):
            U = r + (m - 1) * 1.0 / self.n
            while U > c and i < self.n - 1:
                i += 1
                c += self.particles[i]["weight"]
            src = self.particles[i]
            log, lik = self._new_map((src["log"], src["lik"]))       # :41-45
            new.append({"weight": src["weight"], "pose": src["pose"].copy(), "log": log, "lik": lik})
            idx.append(i)
        self.particles = new
        return np.array(idx, dtype=np.int32)

    def neff(self):
        return neff([p["weight"] for p in self.particles])           # :180-190

    def weighted_pose(self):
        return weighted_pose(np.stack([p["pose"] for p in self.particles]), [p["weight"] for p in self.particles])   # :165-178
