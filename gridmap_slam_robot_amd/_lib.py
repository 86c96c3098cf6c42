"""ctypes binding of libgridmapslam.so -- the C-ABI declared in include/gridmapslam.h.

The library is the product: there is no Python or CPU fallback.  Loading fails loudly when the
shared object has not been built (`python -m gridmap_slam_robot_amd.build`), and every compute call
fails with GMS_ERR_NO_DEVICE when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgridmapslam.so")
if os.environ.get("GMS_LIBRARY"):          # experiments only (tools/ab_compare.sh alternates two builds on one box)
    LIB_PATH = os.environ["GMS_LIBRARY"]

GMS_MAX_TAPS = 129
GMS_BLOCK = 256
GMS_PARTIAL_STRIDE = 9
PACKED_BYTES = 24

GMS_OK, GMS_ERR_INVALID, GMS_ERR_NO_DEVICE, GMS_ERR_HIP, GMS_ERR_NOMEM, GMS_ERR_STATE = 0, -1, -2, -3, -4, -5
K_RAYCAST, K_APPLY, K_LIKELIHOOD, K_SCORE, K_REDUCE, K_RESAMPLE, K_REFINE, K_EXCHANGE, K_ORDER, K_MAPCOPY, K_COUNT = range(11)   # enum of gridmapslam.h (GMS_K_*)
KERNEL_NAMES = ["raycast", "apply", "likelihood", "score", "reduce", "resample", "refine", "exchange", "order", "mapcopy"]
assert len(KERNEL_NAMES) == K_COUNT

BEAM_DTYPE = np.dtype(
    [("local_x", "<f8"), ("local_y", "<f8"), ("distance", "<f8"), ("hit", "u1"), ("pad_", "u1", (7,))]
)
PACKED_DTYPE = np.dtype([("w", "<f8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("pad", "<u4")])


class GmsParams(C.Structure):
    _fields_ = [
        ("width_m", C.c_float), ("height_m", C.c_float), ("resolution", C.c_float),
        ("pos_x", C.c_float), ("pos_y", C.c_float),
        ("n_maps", C.c_int32), ("device", C.c_int32),
        ("l_free", C.c_double), ("l_occ", C.c_double),
        ("ktaps", C.c_int32),
        ("kernel", C.c_double * GMS_MAX_TAPS),
        ("extra_steps", C.c_int32), ("hit_tolerance", C.c_float),
        ("z_hit", C.c_double), ("z_random", C.c_double),
        ("max_range", C.c_float), ("max_beams", C.c_int32),
    ]


class GmsPfStats(C.Structure):
    _fields_ = [
        ("weight_sum", C.c_double), ("neff", C.c_double),
        ("strongest", C.c_int32), ("n_zero", C.c_int32),
        ("max_log_weight", C.c_double),
    ]


class GmsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libgridmapslam error {code}: {msg}")
        self.code = code


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (different SONAME from /opt/rocm's
    libamdhip64.so.7).  Two HIP runtimes in one process cannot share streams, and whichever initialises
    second may not see the GPU.  Promote torch's copy to the global symbol scope BEFORE our library is
    loaded, so that its hip* symbols bind to the runtime torch uses: one runtime, torch streams usable
    through gms_map_set_stream, RCCL ordering intact.  Without torch the system runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load() -> C.CDLL:
    """Load the in-tree HIP library; raise if it is missing (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gridmap_slam_robot_amd.build` "
            "(hipcc, --offload-arch=gfx950). gridmap_slam_robot_amd has no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
    pp = C.POINTER(GmsParams)
    sp = C.POINTER(GmsPfStats)

    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("gms_version", C.c_int)
    sig("gms_build_info", C.c_char_p)
    sig("gms_last_error", C.c_char_p)
    sig("gms_device_count", C.c_int)
    sig("gms_params_default", C.c_int, pp, f32, f32, f32, f32, f32)
    sig("gms_grid_size", C.c_int, pp, vp, vp)
    sig("gms_generate_gaussian_kernel", C.c_int, f64, i32, vp)
    sig("gms_log_odds", f64, f64)
    sig("gms_inv_log_odds", f64, f64)
    sig("gms_map_create", C.c_int, pp, C.POINTER(vp))
    sig("gms_map_destroy", C.c_int, vp)
    sig("gms_map_get_size", C.c_int, vp, vp, vp, vp)
    sig("gms_map_set_stream", C.c_int, vp, vp)
    sig("gms_map_synchronize", C.c_int, vp)
    sig("gms_map_reset", C.c_int, vp)
    sig("gms_map_upload_log", C.c_int, vp, vp)
    sig("gms_map_download_log", C.c_int, vp, vp)
    sig("gms_map_upload_likelihood", C.c_int, vp, vp)
    sig("gms_map_download_likelihood", C.c_int, vp, vp)
    sig("gms_map_copy", C.c_int, vp, vp)
    sig("gms_map_get_raw_at", C.c_int, vp, i32, i32, i32, vp, vp)
    sig("gms_map_get_at_point", C.c_int, vp, i32, C.c_float, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("gms_map_combine", C.c_int, vp, vp)
    sig("gms_map_deskew", C.c_int, vp, vp, vp, vp, i32, f64, f64, vp, vp)
    sig("gms_map_integrate", C.c_int, vp, vp, i32, vp)
    sig("gms_map_integrate_at", C.c_int, vp, vp, i32, vp, i32)
    sig("gms_map_apply_ray", C.c_int, vp, f32, f32, f32, f32, f32, i32)
    sig("gms_map_trace_ray", C.c_int, vp, f32, f32, f32, f32, i32, vp, i32, vp)
    sig("gms_map_trace_scan", C.c_int, vp, vp, i32, vp, vp, vp, i32, vp)
    sig("gms_map_build_likelihood", C.c_int, vp)
    sig("gms_map_update", C.c_int, vp, vp, i32, vp)
    sig("gms_map_update_at", C.c_int, vp, vp, i32, vp, i32)
    sig("gms_map_integrate_dev", C.c_int, vp, vp, i32, vp)
    sig("gms_map_integrate_at_dev", C.c_int, vp, vp, i32, vp, i32)
    sig("gms_map_update_dev", C.c_int, vp, vp, i32, vp)
    sig("gms_map_update_at_dev", C.c_int, vp, vp, i32, vp, i32)
    sig("gms_pf_set_poses_dev", C.c_int, vp, vp)
    sig("gms_pf_score_dev", C.c_int, vp, vp, i32)
    sig("gms_slam_update_dev", C.c_int, vp, vp, vp, i32, vp, f64, i32)
    sig("gms_slam_update_u_dev", C.c_int, vp, f64, f64, C.c_uint64, C.c_uint64, vp, i32, vp, f64, i32)
    sig("gms_slam_frame", C.c_int, vp, vp, vp, vp, i32, f64, f64, C.c_uint64, C.c_uint64, vp, f64, i32)
    sig("gms_slam_update", C.c_int, vp, vp, vp, i32, vp, f64, i32, sp)
    sig("gms_pf_create", C.c_int, vp, i32, C.POINTER(vp))
    sig("gms_pf_destroy", C.c_int, vp)
    sig("gms_pf_set_shard", C.c_int, vp, i64, i64)
    sig("gms_pf_set_poses", C.c_int, vp, vp)
    sig("gms_pf_get_poses", C.c_int, vp, vp)
    sig("gms_pf_set_weights", C.c_int, vp, vp)
    sig("gms_pf_get_weights", C.c_int, vp, vp)
    sig("gms_pf_get_log_weights", C.c_int, vp, vp)
    sig("gms_pf_score", C.c_int, vp, vp, i32)
    sig("gms_pf_normalize", C.c_int, vp, sp)
    sig("gms_pf_get_stats", C.c_int, vp, sp)
    sig("gms_pf_weighted_pose", C.c_int, vp, vp)
    sig("gms_pf_resample", C.c_int, vp, vp, vp, vp)
    sig("gms_pf_resample_if", C.c_int, vp, vp, f64)
    sig("gms_pf_last_resample_indices", C.c_int, vp, vp)
    sig("gms_pf_count", C.c_int, vp, vp, vp, vp)
    sig("gms_pf_did_resample", C.c_int, vp, vp)
    sig("gms_pf_refine_poses", C.c_int, vp, vp, i32)
    sig("gms_pf_last_step", C.c_int, vp, vp, vp, vp, vp)
    sig("gms_pf_set_refine", C.c_int, vp, i32)
    sig("gms_pf_sample_motion", C.c_int, vp, f64, f64, C.c_uint64, C.c_uint64)
    sig("gms_pf_partials_len", C.c_int, vp, vp)
    sig("gms_pf_local_partials", C.c_int, vp, vp)
    sig("gms_pf_apply_partials", C.c_int, vp, vp, vp)
    sig("gms_pf_pack", C.c_int, vp, vp)
    sig("gms_pf_stats_from_partials", C.c_int, vp, vp)
    sig("gms_pf_import_global", C.c_int, vp, vp)
    sig("gms_comm_load", C.c_int, C.c_char_p)
    sig("gms_comm_unique_id", C.c_int, vp)
    sig("gms_comm_create", C.c_int, C.POINTER(vp), vp, i32, i32, i32)
    sig("gms_comm_destroy", C.c_int, vp)
    sig("gms_comm_rank", C.c_int, vp, C.POINTER(i32), C.POINTER(i32))
    sig("gms_pf_normalize_sharded_begin", C.c_int, vp, vp)
    sig("gms_pf_normalize_sharded_end", C.c_int, vp, vp)
    sig("gms_slam_update_sharded_dev", C.c_int, vp, vp, vp, vp, i32, vp, f64, i32)
    sig("gms_slam_update_sharded", C.c_int, vp, vp, vp, vp, i32, vp, f64, i32, sp)
    sig("gms_slam_update_sharded_begin_dev", C.c_int, vp, vp, vp, i32)
    sig("gms_pf_gather_buffers", C.c_int, vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(vp), C.POINTER(C.c_int64))
    sig("gms_slam_update_sharded_end_dev", C.c_int, vp, vp, i32, vp, f64, i32)
    sig("gms_profile_enable", C.c_int, vp, i32)
    sig("gms_profile_reset", C.c_int, vp)
    sig("gms_profile_get", C.c_int, vp, i32, vp, vp)
    sig("gms_profile_sample", C.c_int, vp, i32)
    sig("gms_profile_calibrate", C.c_int, vp, i32, C.POINTER(C.c_double))
    sig("gms_profile_calibrate2", C.c_int, vp, i32, vp, vp)
    sig("gms_map_tile_stats", C.c_int, vp, i32, vp)
    sig("gms_pf_set_log_normalize", C.c_int, vp, i32)
    sig("gms_pf_set_reference_order", C.c_int, vp, i32)
    sig("gms_debug_f32", C.c_int, vp, i32, vp, vp, i64)
    sig("gms_slam_create", C.c_int, pp, i32, C.POINTER(vp))
    sig("gms_slam_destroy", C.c_int, vp)
    sig("gms_slam_reset", C.c_int, vp)
    sig("gms_slam_count", C.c_int, vp, vp, vp, vp)
    sig("gms_slam_handles", C.c_int, vp, C.POINTER(vp), C.POINTER(vp))
    sig("gms_slam_set_refine", C.c_int, vp, i32)
    sig("gms_slam_update_per_particle", C.c_int, vp, vp, i32, i32, f64, f64, C.c_uint64, C.c_uint64, sp)
    sig("gms_slam_update_per_particle_dev", C.c_int, vp, vp, i32, i32, f64, f64, C.c_uint64, C.c_uint64, sp)
    sig("gms_slam_resample_maps", C.c_int, vp, f64, vp, vp)
    sig("gms_slam_resample_maps_if", C.c_int, vp, f64, f64)
    sig("gms_slam_download_map", C.c_int, vp, i32, vp, vp)
    sig("gms_slam_upload_map", C.c_int, vp, i32, vp, vp)
    sig("gms_slam_download_maps", C.c_int, vp, vp, vp)
    sig("gms_slam_combined", C.c_int, vp)
    sig("gms_slam_copies", C.c_int, vp, C.POINTER(C.c_int64))
    sig("gms_slam_trace_scan", C.c_int, vp, i32, vp, i32, vp, vp, i32, vp)
    sig("gms_slam_create_shard", C.c_int, pp, i32, C.c_int64, C.c_int64, C.POINTER(vp))
    sig("gms_slam_update_local", C.c_int, vp, vp, i32, i32, f64, f64, C.c_uint64, C.c_uint64)
    sig("gms_slam_update_local_dev", C.c_int, vp, vp, i32, i32, f64, f64, C.c_uint64, C.c_uint64)
    sig("gms_slam_shard_draw", C.c_int, vp, f64, f64, C.POINTER(C.c_int32), vp)
    sig("gms_slam_record_doubles", C.c_int, vp, C.POINTER(C.c_int64))
    sig("gms_slam_shard_export", C.c_int, vp, vp, i32, vp)
    sig("gms_slam_shard_gather", C.c_int, vp, vp, vp, vp)
    sig("gms_slam_update_sharded_maps", C.c_int, vp, vp, vp, i32, i32, f64, f64, C.c_uint64, C.c_uint64, sp)
    sig("gms_slam_resample_sharded_maps", C.c_int, vp, vp, f64, f64, C.POINTER(C.c_int32))
    sig("gms_slam_plan_exchange", C.c_int, vp, i32, i32, i32, vp, vp, vp, vp, vp)
    sig("gms_debug_set_stamps", C.c_int, vp, vp)
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != GMS_OK:
        raise GmsError(rc, load().gms_last_error().decode("utf-8", "replace"))


def ptr(a: np.ndarray) -> int:
    return a.ctypes.data
