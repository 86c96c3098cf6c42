"""Assertions shared by the GPU parity tests."""
import numpy as np


AGREEMENT = []      # one record per checked resampling step: conftest writes them to gpurun_out/resample_agreement.json at session end


def assert_resample_indices(idx, want, n_ambiguous, label=None):
    """Source indices of a resampling step (SLAM.java:133-153) against the sequential oracle's.

    The device forms the cumulative weights as a blocked scan, the reference as one running sum; a slot whose threshold
    U lies within rounding distance of a cumulative boundary is counted by the device (`n_ambiguous`) and may land on
    the neighbouring particle.  So: every index within one slot of the oracle's, and never more differing slots than the
    device flagged -- with no flagged slot that is plain equality.
    """
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    want = np.asarray(want, dtype=np.int64).reshape(-1)
    assert idx.shape == want.shape
    diff = idx != want
    assert (np.abs(idx - want) <= 1).all(), f"{int((np.abs(idx - want) > 1).sum())} slots further than a neighbour from the oracle's"
    assert int(diff.sum()) <= int(n_ambiguous), f"{int(diff.sum())} slots differ, {int(n_ambiguous)} flagged ambiguous"
    assert (np.diff(idx) >= 0).all()                      # systematic resampling is order-preserving (SLAM.java:140-149)
    if label is None:
        import inspect
        f = inspect.stack()[1]
        label = f"{f.filename.rsplit('/', 1)[-1]}::{f.function}"
    # BASELINE.md asks for "indices equal for fixed r": how far from that a blocked scan is, as numbers
    AGREEMENT.append({"where": label, "slots": int(idx.size), "slots_differing": int(diff.sum()), "n_ambiguous": int(n_ambiguous)})
    return int(diff.sum())


def near_boundary_slots(wn, r01, rel_tol=1e-9):
    """How many resampling slots have their threshold U (SLAM.java:136,141) within rel_tol * total of a cumulative-weight
    boundary of `wn`.  When the oracle scans weights that agree with the device's only to ~1e-11 (its own normalisation
    instead of the device's), these are the slots where the two may legitimately pick neighbours."""
    wn = np.asarray(wn, dtype=np.float64)
    n = wn.size
    cum = np.cumsum(wn)                                   # sequential fp64 adds, the reference's order
    U = r01 * 1.0 / n + np.arange(n, dtype=np.float64) * 1.0 / n
    j = np.clip(np.searchsorted(cum, U, side="left"), 0, n - 1)
    tol = rel_tol * cum[-1]
    near = np.abs(U - cum[j]) <= tol
    near |= (j > 0) & (np.abs(U - cum[np.maximum(j - 1, 0)]) <= tol)
    return int(near.sum())
