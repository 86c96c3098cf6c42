#!/usr/bin/env python3
"""Stage timeline of one C3 scan step from an instrumented build (development tool).

  GMS_EXTRA_FLAGS=-DGMS_STAMPS python -m gridmap_slam_robot_amd.build --force; cp .../libgridmapslam.so .../exp_stamps.so; rebuild the product
  GMS_LIBRARY=$PWD/gridmap_slam_robot_amd/lib/exp_stamps.so python tools/stamps.py [--config C3] [--steps 40]

Every instrumented kernel writes wall-clock stamps (100 MHz) of its stages per workgroup; the stamps of the LAST step are
printed relative to the first workgroup entering the scoring kernel: first / median / last workgroup per stage."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = {
    0: ("k_score_c", {0: "entered", 3: "pose loaded (thread 0)", 4: "trig done (thread 0)", 1: "beams compacted, trig done", 2: "products stored"}),
    1: ("k_partials", {0: "entered", 2: "weights combined, sums accumulated (thread 0)", 3: "butterflies done", 4: "waves combined", 1: "left"}),
    2: ("k_norm_raycast", {0: "entered", 1: "ray: pose folded", 2: "far: rays set up", 3: "far: recurrence done (producer)", 4: "far: first consumer done",
                           5: "far: box committed", 9: "near: tile cleared", 10: "near: 64 steps counted", 11: "near: tile flushed", 12: "near: box committed",
                           14: "normalise: left", 15: "apply: left"}),
    3: ("k_lik_resample", {0: "entered", 5: "resample: Neff folded", 6: "resample: chunk offsets scanned", 7: "resample: source found (thread 0)", 1: "resample: left", 3: "likelihood: first tile staged", 4: "likelihood: first tile H pass done (non-uniform)", 2: "likelihood: left"}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--loop", action="store_true", help="the closed loop (motion model and resampling on the device)")
    ap.add_argument("--log-normalize", action="store_true", help="with gms_pf_set_log_normalize on")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    from gridmap_slam_robot_amd import _lib
    args = argparse.Namespace(particles=0, exchange="auto", host_inputs=False, full_rebuild=False)
    wl = bench.Workload(a.config, args, torch, dist, 0, 1, 0, False, loop=a.loop)
    if a.log_normalize:
        wl.pf.set_log_normalize(True)
    buf = torch.zeros(4 * 1024 * 16, dtype=torch.int64, device=wl.dev)
    _lib.check(_lib.load().gms_debug_set_stamps(wl.m._h, C.c_void_p(buf.data_ptr())))
    for i in range(a.steps):
        wl.step(i)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(4, 1024, 16).astype(np.float64)
    _lib.check(_lib.load().gms_debug_set_stamps(wl.m._h, None))
    s[s == 0] = np.nan
    t0 = np.nanmin(s[0, :, 0])
    s[np.abs(s - t0) > 1e5] = np.nan          # (a slot left over from an earlier launch of another shape: more than a millisecond away)
    # a stage a workgroup did not reach in THIS step keeps an earlier step's stamp: older than the workgroup's own entry
    for kid in range(4):
        stale = s[kid] < s[kid][:, 0:1]
        s[kid][stale] = np.nan
    print(f"{a.config}: stage stamps of the last of {a.steps} steps, microseconds after the first scoring workgroup entered")
    for kid, (name, slots) in STAGES.items():
        for slot, what in slots.items():
            v = (s[kid, :, slot] - t0) * 0.01
            v = v[~np.isnan(v)]
            if v.size:
                print(f"  {name:16s} {what:34s} n={v.size:4d}  first {v.min():7.2f}  median {np.median(v):7.2f}  last {v.max():7.2f}")
    # likelihood workgroups that blurred a tile: time from entering to leaving
    L = s[3]
    blur = ~np.isnan(L[:, 4])
    if blur.any():
        d = (L[blur, 2] - L[blur, 0]) * 0.01
        e = (L[blur, 3] - L[blur, 0]) * 0.01
        h = (L[blur, 4] - L[blur, 3]) * 0.01
        v = (L[blur, 2] - L[blur, 4]) * 0.01
        print(f"  likelihood workgroups with a non-uniform first tile: {int(blur.sum())}; enter->staged median {np.median(e):.2f} max {e.max():.2f}; "
              f"staged->H done median {np.median(h):.2f} max {h.max():.2f}; H done->left median {np.median(v):.2f} max {v.max():.2f}; total median {np.median(d):.2f} max {d.max():.2f}")
    sc = s[0]
    dur = (sc[:256, 2] - sc[:256, 1]) * 0.01
    print("  k_score_c lookup phase per workgroup (us), by blockIdx & 7 (XCD) rows and blockIdx >> 3 columns:")
    for x in range(8):
        print("   ", " ".join(f"{dur[x + 8 * j]:5.1f}" for j in range(32)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
