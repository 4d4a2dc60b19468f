"""Shared by the N = 5000 golden tests (CPU: oracle vs reference; GPU: HIP vs reference).  Test infrastructure."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N = 5000
FIRTH_BITS = (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 6)     # bad-chisq, high-bse, separable, inversion, firth-fail: the Firth routes


def unpack(bits, n):
    return np.unpackbits(bits, axis=1, bitorder="little")[:, :n].astype(np.float64)


def lmm_design(seed=1003, n=N):
    """bench.synth_lmm_inputs' kinship and phenotype, numpy only (the golden stores the seed, not K): K = G G^T of lineage-structured
    binary markers (exact integers in float64) and a binary phenotype with a heritable component."""
    rng = np.random.default_rng(seed)
    nlin, nmark = 40, 2000
    lin = rng.integers(0, nlin, n)
    base = rng.uniform(0.05, 0.95, (nlin, nmark))
    Gm = (rng.random((n, nmark)) < base[lin]).astype(np.float64)
    K = Gm.dot(Gm.T)
    g = Gm[:, :60].dot(rng.standard_normal(60)); g = (g - g.mean()) / g.std()
    y = ((0.7 * g + 0.7 * rng.standard_normal(n)) > 0).astype(np.float64)
    return K, y


def noise_row_mask(rows, y):
    """A variant identical to the phenotype (or its complement) fits perfectly: the reference's residual variance is +-1e-17 rounding
    noise, bse = sqrt(noise) (NaN in the golden), p = noise.  Not a parity target (case 2)."""
    return (rows == y).all(axis=1) | (rows == 1 - y).all(axis=1)
