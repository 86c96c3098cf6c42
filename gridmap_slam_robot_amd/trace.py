"""Reader / writer of the reference's recorded-trace files (the only on-disk format that carries scans).

Format (big-endian java.io.DataOutputStream; J/app/DataRecorder.java:381-436, J/app/ObjectSerializer.java:36-83):

    byte   0xFF                                   header
    short  number of frames
    per frame:
        float  timeStamp                          (DataRecorder.java:391)
        double dCenter, double dTheta             (ObjectSerializer.writeOdometry :36-40)
        short  number of measurements             (writeObservation :47-57)
        per measurement: double angle, double distance, byte wasHit      (writeMeasurement :72-76)

Host-side I/O only: it feeds recorded runs to the device path (GridMap.deskew -> score / update)."""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List

import numpy as np

_MEAS = np.dtype([("angle", ">f8"), ("distance", ">f8"), ("hit", "u1")])


@dataclass
class Frame:
    """RecordedTimeFrame: timestamp + TimeFrame(z, u) (J/app/DataRecorder.java, J/slam/TimeFrame.java)."""
    time_stamp: float
    d_center: float
    d_theta: float
    angle: np.ndarray       # float64 [n]
    distance: np.ndarray    # float64 [n]
    hit: np.ndarray         # uint8   [n]


def read_trace(path: str) -> List[Frame]:
    """DataRecorder.load (DataRecorder.java:403-436)."""
    buf = open(path, "rb").read()
    if not buf or buf[0] != 0xFF:
        # the reference throws IllegalStateException("... header byte is not correct! ...") (:415-418)
        raise ValueError(f"header byte is not correct! Wanted 255, got {buf[0] if buf else 'EOF'}")
    (n_frames,) = struct.unpack_from(">h", buf, 1)
    off = 3
    frames = []
    for _ in range(n_frames):
        ts, dc, dt, n = struct.unpack_from(">fddh", buf, off)
        off += 4 + 8 + 8 + 2
        m = np.frombuffer(buf, dtype=_MEAS, count=n, offset=off)
        off += n * _MEAS.itemsize
        frames.append(Frame(ts, dc, dt, m["angle"].astype(np.float64), m["distance"].astype(np.float64), (m["hit"] != 0).astype(np.uint8)))
    return frames


def write_trace(path: str, frames: List[Frame]) -> None:
    """DataRecorder.save (DataRecorder.java:381-400)."""
    out = bytearray(b"\xff")
    out += struct.pack(">h", len(frames))
    for f in frames:
        out += struct.pack(">fddh", f.time_stamp, f.d_center, f.d_theta, len(f.angle))
        m = np.zeros(len(f.angle), dtype=_MEAS)
        m["angle"], m["distance"], m["hit"] = f.angle, f.distance, np.asarray(f.hit).astype(np.uint8)
        out += m.tobytes()
    with open(path, "wb") as fh:
        fh.write(bytes(out))
