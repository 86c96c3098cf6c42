"""Synthetic scan + odometry traces for tests and bench (no dataset or recording ships with the
reference: maps/* is git-ignored there, java/GridMapGL/.gitignore:4).

World: a rectangular room (thin wall boxes, one doorway gap) plus seeded axis-aligned boxes; the robot
drives a circle with tangent heading.  Beam b of B has robot-frame angle 2*pi*b/B - pi/2 (mirrors
Robot.SENSOR_ANGLE_OFFSET, J/slam/Robot.java:20); range = analytic ray/box intersection + N(0, 1 cm);
no return within SENSOR_MAX_RANGE (10 m, J/slam/SensorModel.java:20) => distance = 10, wasHit = false
(as J/conn/ConnectionThread.java:77-81).  localX/localY = distance * cos/sin(angle) in double
(J/slam/Observation.java:49-50).

Pure numpy host code; it produces inputs, it is not on the measured path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from ._lib import BEAM_DTYPE

MAX_RANGE = 10.0


@dataclass
class World:
    boxes: np.ndarray            # [K][4] xmin, ymin, xmax, ymax
    half: float                  # room half size
    radius: float                # trajectory radius


def make_world(extent_m: float, seed: int = 1234, n_boxes: int = 8) -> World:
    """Room half-size min(5.5 m, 0.4*extent) so that nearly every beam returns within 10 m."""
    rng = np.random.default_rng(seed)
    half = min(5.5, 0.4 * extent_m)
    radius = min(1.5, 0.27 * half)
    t = 0.02 * half + 0.02                      # wall thickness
    gap = 0.15 * half                           # doorway half-width in the east wall
    boxes = [
        (-half - t, -half - t, half + t, -half),            # south
        (-half - t, half, half + t, half + t),              # north
        (-half - t, -half, -half, half),                    # west
        (half, -half, half + t, -gap),                      # east, below the door
        (half, gap, half + t, half),                        # east, above the door
    ]
    for _ in range(n_boxes):
        for _try in range(100):
            w, h = rng.uniform(0.05 * half, 0.25 * half, size=2)
            cx, cy = rng.uniform(-0.85 * half, 0.85 * half, size=2)
            # keep the driving annulus free
            d = math.hypot(cx, cy)
            if abs(d - radius) > 0.6 * (w + h) + 0.12 * half:
                boxes.append((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2))
                break
    return World(np.array(boxes, dtype=np.float64), half, radius)


def true_pose(world: World, t: int, T: int) -> np.ndarray:
    a = 2 * math.pi * t / T
    return np.array([world.radius * math.cos(a), world.radius * math.sin(a), a + math.pi / 2], dtype=np.float32)


def cast(world: World, origin, angles: np.ndarray) -> np.ndarray:
    """Distance along each world-frame angle to the first box boundary (inf if none)."""
    ox, oy = float(origin[0]), float(origin[1])
    dx, dy = np.cos(angles)[:, None], np.sin(angles)[:, None]
    bx0, by0, bx1, by1 = (world.boxes[:, i][None, :] for i in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        tx0, tx1 = (bx0 - ox) / dx, (bx1 - ox) / dx
        ty0, ty1 = (by0 - oy) / dy, (by1 - oy) / dy
    txn, txf = np.minimum(tx0, tx1), np.maximum(tx0, tx1)
    tyn, tyf = np.minimum(ty0, ty1), np.maximum(ty0, ty1)
    # rays parallel to an axis: inside the slab => (-inf, inf), outside => no hit
    par_x = np.broadcast_to(dx == 0, txn.shape)
    in_x = (ox >= bx0) & (ox <= bx1)
    txn = np.where(par_x, np.where(in_x, -np.inf, np.inf), txn)
    txf = np.where(par_x, np.where(in_x, np.inf, -np.inf), txf)
    par_y = np.broadcast_to(dy == 0, tyn.shape)
    in_y = (oy >= by0) & (oy <= by1)
    tyn = np.where(par_y, np.where(in_y, -np.inf, np.inf), tyn)
    tyf = np.where(par_y, np.where(in_y, np.inf, -np.inf), tyf)
    tn, tf = np.maximum(txn, tyn), np.minimum(txf, tyf)
    hit = (tn <= tf) & (tf > 0)
    t = np.where(hit, np.where(tn > 0, tn, tf), np.inf)
    return t.min(axis=1)


def make_scan(world: World, pose, B: int, rng: np.random.Generator, noise: float = 0.01) -> np.ndarray:
    """One revolution seen from `pose` -> beams (BEAM_DTYPE [B])."""
    b = np.arange(B, dtype=np.float64)
    local = 2 * math.pi * b / B - math.pi / 2
    rng_ = cast(world, pose, local + float(pose[2]))
    rng_ = rng_ + rng.normal(0.0, noise, size=B)
    hit = rng_ < MAX_RANGE
    dist = np.where(hit, np.maximum(rng_, 0.02), MAX_RANGE)
    beams = np.zeros(B, dtype=BEAM_DTYPE)
    beams["local_x"] = dist * np.array([math.cos(a) for a in local])
    beams["local_y"] = dist * np.array([math.sin(a) for a in local])
    beams["distance"] = dist
    beams["hit"] = hit.astype(np.uint8)
    return beams


@dataclass
class Trace:
    world: World
    poses: np.ndarray            # [T][3] float32 true poses
    scans: np.ndarray            # [T][B] BEAM_DTYPE
    extent: float
    resolution: float
    position: tuple = field(default=(0.0, 0.0))


def make_trace(extent_m: float, resolution: float, B: int, T: int = 64, seed: int = 1234, n_scans=None) -> Trace:
    """Square map of `extent_m` centred on the origin; T poses on the circle, the first n_scans generated."""
    world = make_world(extent_m, seed)
    rng = np.random.default_rng(seed + 1)
    n = T if n_scans is None else n_scans
    poses = np.stack([true_pose(world, t, T) for t in range(n)])
    scans = np.stack([make_scan(world, poses[t], B, rng) for t in range(n)])
    return Trace(world, poses, scans, extent_m, resolution, (-extent_m / 2, -extent_m / 2))


def make_particles(pose, N: int, seed: int = 99, sigma_xy: float = 0.10, sigma_theta_deg: float = 5.0) -> np.ndarray:
    """pose_i = pose + N(0, sigma_xy, sigma_theta) as float32 [N][3]; particle 0 is the pose itself."""
    rng = np.random.default_rng(seed)
    p = np.empty((N, 3), dtype=np.float64)
    p[:, 0] = pose[0] + rng.normal(0, sigma_xy, N)
    p[:, 1] = pose[1] + rng.normal(0, sigma_xy, N)
    p[:, 2] = pose[2] + rng.normal(0, math.radians(sigma_theta_deg), N)
    p[0] = np.asarray(pose, dtype=np.float64)
    return p.astype(np.float32)


# BASELINE.json configs, made concrete
CONFIGS = {
    "C1": dict(particles=1, beams=360, extent=25.6, resolution=0.05, n_maps=1),
    "C2": dict(particles=1024, beams=360, extent=51.2, resolution=0.05, n_maps=1),
    "C3": dict(particles=16384, beams=720, extent=40.96, resolution=0.02, n_maps=1),
    "C4": dict(particles=65536, beams=720, extent=40.96, resolution=0.02, n_maps=1),
    "C5": dict(particles=4096, beams=1080, extent=51.2, resolution=0.05, n_maps=64),
}


def make_recording(extent_m: float, B: int, T: int = 64, seed: int = 1234, n_frames=None, noise: float = 0.01):
    """A synthetic RECORDING of the same drive: what DataRecorder would have saved (J/app/DataRecorder.java:381-400) -- per
    revolution the odometry u = (dCenter, dTheta) since the previous one and the RAW polar measurements {angle, distance,
    wasHit}, each taken from where the robot was at that instant of the revolution, so that the de-skew loop of
    GridMapApp.onHandleData (J/app/GridMapApp.java:143-175) has something to undo.  Measurement i of `length` is taken at the
    fraction d_i = -(length - i) / length of the motion before the frame's end (:150).  Returns (frames, end poses [n][3])."""
    from .trace import Frame
    world = make_world(extent_m, seed)
    rng = np.random.default_rng(seed + 2)
    n = T if n_frames is None else n_frames
    poses = np.stack([true_pose(world, t, T).astype(np.float64) for t in range(-1, n)])      # poses[k] = end pose of frame k - 1
    frames = []
    for k in range(n):
        p0, p1 = poses[k], poses[k + 1]
        d_center = float(math.hypot(p1[0] - p0[0], p1[1] - p0[1]))
        d_theta = float(p1[2] - p0[2])
        i = np.arange(B, dtype=np.float64)
        frac = -(B - i) / B                                                  # -1 .. -1/B
        # where the robot was when measurement i was taken: back along the motion of this frame
        th = p1[2] + d_theta * frac
        px = p1[0] + np.cos(th) * d_center * frac
        py = p1[1] + np.sin(th) * d_center * frac
        local = 2 * math.pi * i / B - math.pi / 2
        dist = np.array([cast(world, (px[j], py[j]), np.array([local[j] + th[j]]))[0] for j in range(B)])
        dist = dist + rng.normal(0.0, noise, size=B)
        hit = dist < MAX_RANGE
        dist = np.where(hit, np.maximum(dist, 0.02), MAX_RANGE)
        frames.append(Frame(float(np.float32(0.1 * k)), d_center, d_theta, local.copy(), dist, hit.astype(np.uint8)))
    return frames, poses[1:].astype(np.float32)


def dead_reckon(pose, d_center: float, d_theta: float) -> np.ndarray:
    """Odometry.apply without its noise (J/slam/Odometry.java:77-96): heading first, then the step along the new heading."""
    th = np.float32(np.float64(pose[2]) + d_theta)
    c, s = np.float32(math.cos(float(th))), np.float32(math.sin(float(th)))
    return np.array([np.float32(np.float64(pose[0]) + np.float64(c) * d_center), np.float32(np.float64(pose[1]) + np.float64(s) * d_center), th],
                    dtype=np.float32)
