"""gridmap_slam_robot_amd -- MI355X-native occupancy-grid SLAM core.

One hot path of antbern/gridmap-slam-robot (log-odds ray-cast map update, likelihood field,
particle scan matcher) as hand-written HIP kernels for gfx950 behind the C-ABI of
include/gridmapslam.h.  This package is the thin host side: a ctypes binding (_lib), the mirror of
the reference's GridMap / ParticleFilter / SLAM class surface (gridmap), particle sharding over
torch.distributed (distributed) and the synthetic trace generator used by tests and bench (synth).
"""
from .gridmap import GridMap, Observation, ParticleFilter, Pose, SLAM, SLAMParticleMaps  # noqa: F401
from ._lib import BEAM_DTYPE, GmsError, load  # noqa: F401

__version__ = "0.1.0"
