#!/usr/bin/env python3
"""Where could the reference's FastMath.sin / cos (commons-math3 3.6.1, absent from the tree) make Transform.fromRobotToWorld's
float-rounded trig (J/math/Transform.java:15-16 via J/math/MathUtil.java:30-40) differ from the oracle's glibc restatement?

The Java narrows a double to float.  Two libms whose doubles both lie within 1 ulp of the true value narrow to the same float
unless a float ROUNDING BOUNDARY (the midpoint of two adjacent floats) lies within 1 ulp of the true value.  This script sweeps every
float in [-2 pi, 2 pi] (2.17e9 angles, cos and sin), takes the candidates whose glibc double lies within 3 ulps of a boundary
(oracle/gms_oracle.c orc_trig_near_float_boundary, OpenMP), and settles each with 120-digit arithmetic (mpmath): the true value's
distance to the boundary in ulps and the correctly rounded float.  Result: tests/golden/trig_fragile.json --
  "within_1_ulp": the angles at which an implementation with error < 1 ulp COULD narrow differently (everywhere else it cannot);
  for every candidate: what the oracle returns and what the correctly rounded float is.
  usage: python tests/golden/make_trig_fragile.py [--threads 8]   (about a minute on 8 cores)"""
import argparse, json, os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc
import mpmath

TWO_PI_BITS = int(np.float32(2 * np.pi).view(np.uint32))


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def settle(theta, which):
    mpmath.mp.dps = 120
    t = mpmath.mpf(float(theta))                              # the widened float, exactly
    true = mpmath.cos(t) if which == "cos" else mpmath.sin(t)
    d = float(true)                                           # correctly rounded double
    f = np.float32(d)
    lo = np.nextafter(f, np.float32(-np.inf)); hi = np.nextafter(f, np.float32(np.inf))
    mids = [(mpmath.mpf(float(f)) + mpmath.mpf(float(lo))) / 2, (mpmath.mpf(float(f)) + mpmath.mpf(float(hi))) / 2]
    ulp = mpmath.mpf(float(np.nextafter(abs(d), np.inf) - abs(d)))
    dist = min(abs(true - m) for m in mids) / ulp
    # the correctly rounded float of the TRUE value (ties cannot occur: cos/sin of a non-zero rational-in-binary is irrational)
    cands = sorted([lo, f, hi], key=lambda c: abs(true - mpmath.mpf(float(c))))
    return float(dist), float(cands[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "trig_fragile.json"))
    a = ap.parse_args()
    orc.build()
    cands = []
    for neg in (False, True):
        cands += orc.trig_near_float_boundary(0, TWO_PI_BITS, neg, 3.0, a.threads)
    rows = []
    for theta, which, dist_glibc in cands:
        dist_true, rn = settle(theta, which)
        c, s = orc.pose_trig(theta)
        got = float(np.float32(c if which == "cos" else s))
        rows.append({"theta": theta, "theta_bits": f32_bits(theta), "function": which, "ulps_from_boundary_glibc": dist_glibc,
                     "ulps_from_boundary_true": dist_true, "correctly_rounded_float": rn, "oracle_float": got,
                     "oracle_is_correctly_rounded": got == rn})
    out = {"domain": "every float in [-2 pi, 2 pi], cos and sin of the widened float", "angles_swept": 2 * (TWO_PI_BITS + 1),
           "candidate_window_ulps": 3.0, "candidates": rows,
           "within_1_ulp": [r for r in rows if r["ulps_from_boundary_true"] < 1.0],
           "statement": "outside within_1_ulp, every cos/sin whose double is within 1 ulp of the true value narrows to the float the oracle uses"}
    json.dump(out, open(a.out, "w"), indent=1)
    print(f"{len(rows)} candidates within 3 ulps of a float rounding boundary, {len(out['within_1_ulp'])} within 1 ulp; "
          f"oracle correctly rounded at {sum(r['oracle_is_correctly_rounded'] for r in rows)} of {len(rows)}")


if __name__ == "__main__":
    main()
