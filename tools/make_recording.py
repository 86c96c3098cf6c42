#!/usr/bin/env python3
"""Writes tests/golden/recording_360.bin: a synthetic recording in the reference's DataRecorder format
(J/app/DataRecorder.java:381-436; gridmap_slam_robot_amd/trace.py) -- the reference ships none (maps/* is git-ignored there,
java/GridMapGL/.gitignore:4).  64 revolutions of 360 raw measurements around the synthetic room of synth.make_world(25.6 m),
odometry per frame; `bench.py --trace` and tests/test_gpu_trace_replay.py replay it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gridmap_slam_robot_amd import synth
from gridmap_slam_robot_amd.trace import write_trace, read_trace

frames, poses = synth.make_recording(25.6, 360, T=64, seed=4321)
out = os.path.join(ROOT, "tests", "golden", "recording_360.bin")
write_trace(out, frames)
np.save(os.path.join(ROOT, "tests", "golden", "recording_360_poses.npy"), poses)
back = read_trace(out)
assert len(back) == len(frames) and all(np.array_equal(a.distance, b.distance) for a, b in zip(frames, back))
print(out, os.path.getsize(out), "bytes,", len(frames), "frames")
