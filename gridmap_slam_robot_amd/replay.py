"""Replaying a recorded trace through the device path, frame by frame: what GridMapApp.onHandleData does with every
revolution (J/app/GridMapApp.java:133-192) -- de-skew the raw measurements with the frame's odometry (:143-175), then
SLAM.update(z, u) (J/slam/SLAM.java:80-131: motion-model sample per particle, weight, map update at the filter's pose) and
`if (neff < N / 2) resample()` (:185-186).  Here: gms_slam_frame, one call per frame (= gms_map_deskew -> gms_pf_sample_motion ->
gms_slam_update_dev), the scan handed over as the recording's raw host arrays, everything else resident on the device."""
from __future__ import annotations

import math

import numpy as np

from .gridmap import GridMap, ParticleFilter
from .synth import dead_reckon
from .trace import Frame


class TraceReplay:
    def __init__(self, grid_map: GridMap, pf: ParticleFilter, start_pose, seed: int = 2024, resample_fraction: float = 0.5,
                 one_call: bool = True):
        self.m, self.pf, self.seed, self.fraction = grid_map, pf, seed, resample_fraction
        self.one_call = one_call        # gms_slam_frame per revolution; False: the three calls it stands for (the same bits)
        self.pose = np.asarray(start_pose, dtype=np.float32)          # dead-reckoned pose of the bootstrap frames
        self.frame_no = 0
        pf.set_poses(np.broadcast_to(self.pose, (pf.n, 3)).copy())    # SLAM.java:65-77: every particle starts at the same pose

    def bootstrap(self, f: Frame):
        """A mapping-only frame: the scan goes into the map at the dead-reckoned pose.  (The reference starts from an empty
        map, whose likelihood field scores every pose alike: 0.1 per beam, GridMap.java:285-286 -- 1e-360 at 360 beams.  A few
        frames of plain mapping give the filter something to localise against.)"""
        self.pose = dead_reckon(self.pose, f.d_center, f.d_theta)
        obs = self.m.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)                     # GridMapApp.java:143-175
        self.m.update(obs, self.pose)                                                              # GridMap.java:173-250
        self.pf.set_poses(np.broadcast_to(self.pose, (self.pf.n, 3)).copy())
        self.frame_no += 1

    def step(self, f: Frame, r01: float):
        """One recorded revolution through the filter (GridMapApp.java:143-192)."""
        integrate = abs(f.d_theta) <= math.radians(30)                                             # SLAM.java:82
        if self.one_call:
            self.pf.slam_frame(f.angle, f.distance, f.hit, f.d_center, f.d_theta, self.seed, self.frame_no, r01, self.fraction, integrate)
        else:
            dev, B = self.m.deskew_dev(f.angle, f.distance, f.hit, f.d_center, f.d_theta)          # :143-175
            self.pf.sample_motion(f.d_center, f.d_theta, self.seed, self.frame_no)                 # SLAM.java:90, 155-163
            self.pf.slam_update_dev(0, dev, B, r01, self.fraction, integrate)                      # :87-131 + GridMapApp.java:185-186
        self.frame_no += 1
