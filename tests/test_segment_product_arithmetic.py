"""The device combines a particle's per-segment products (GridMap.java:259-294's one product, cut into segments of the scan) as
mantissa and exponent: the mantissas are multiplied in segment order without renormalising, the exponents are added
(csrc/gms_pf_kernels.hip, combine_segments).  Two claims behind that, checked here in IEEE double arithmetic on the host:
  1. it is the renormalise-after-every-factor chain, bit for bit (scaling by a power of two does not change how a product rounds);
  2. wherever the plain sequential product stays a normal double, it is that product, bit for bit."""
import math

import numpy as np


def _chain_renormalised(vals):
    m, e = math.frexp(vals[0])
    for v in vals[1:]:
        m2, e2 = math.frexp(v)
        m, de = math.frexp(m * m2)
        e += e2 + de
    return m, e


def _chain_plain_mantissas(vals):
    M, e = 1.0, 0
    for v in vals:
        m2, e2 = math.frexp(v)
        M *= m2
        e += e2
    m, de = math.frexp(M)
    return m, e + de


def test_mantissa_chain_is_the_renormalised_chain():
    rng = np.random.default_rng(11)
    for trial in range(4000):
        nseg = int(rng.integers(1, 33))
        # segment products: up to 128 factors in [0.01, 1] each, i.e. anything from 1e-256 to 1
        vals = [float(10.0 ** (-rng.uniform(0.0, 256.0)) * rng.uniform(0.5, 1.0)) for _ in range(nseg)]
        assert _chain_renormalised(vals) == _chain_plain_mantissas(vals)


def test_it_is_the_plain_product_while_that_stays_normal():
    rng = np.random.default_rng(12)
    for trial in range(4000):
        nseg = int(rng.integers(1, 33))
        vals = [float(10.0 ** (-rng.uniform(0.0, 9.0)) * rng.uniform(0.5, 1.0)) for _ in range(nseg)]     # product >= 1e-288 * 2^-32
        plain = 1.0
        for v in vals:
            plain *= v
        if plain < 2.3e-308:
            continue
        m, e = _chain_plain_mantissas(vals)
        assert math.ldexp(m, e) == plain
