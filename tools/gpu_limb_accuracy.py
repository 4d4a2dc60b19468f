"""How far are the int8-limb results from the fp64 restatement on a hard spectrum?  Strongly structured kinship (a few deep
lineages, large top eigenvalues), high h2, variants that ARE lineage markers (their projection lives in the top eigenvectors,
where 1/Sd is smallest and x^T K^-1 x is a small difference of large numbers).  Prints the worst relative deviation of beta, bse
and p from the oracle for L = 4, 5, 6 limbs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.lmm import initialise_lmm_arrays
from oracle import oracle as orc

rng = np.random.default_rng(0)
N = int(os.environ.get("N", 1200))
nl = 6
lin = rng.integers(0, nl, N)
# markers: lineage-defining (deep branches) + shallow noise
G = np.concatenate([(lin[None, :] == rng.integers(0, nl, 400)[:, None]).astype(float),
                    (rng.random((300, N)) < 0.3).astype(float)])
K = G.T @ G
y = ((lin < 3).astype(float) * 0.8 + rng.standard_normal(N) * 0.6 > 0.4).astype(float)
for h2_force in (None, 0.9, 0.99):
    U, S, h2, nll, C = initialise_lmm_arrays(K, y)
    if h2_force is not None:
        h2 = h2_force
    V = 400
    Kv = np.concatenate([(lin[None, :] == rng.integers(0, nl, V // 2)[:, None]).astype(np.uint8),
                         (rng.random((V // 2, N)) < rng.uniform(0.05, 0.95, V // 2)[:, None]).astype(np.uint8)])
    flip = rng.random(Kv.shape) < 0.01                      # lineage markers with a few exceptions
    Kv = np.where(flip, 1 - Kv, Kv).astype(np.uint8)
    af = Kv.mean(axis=1); Kv = Kv[(af > 0.01) & (af < 0.99)]
    wb, ws, wf, wp = orc.LmmOracle(U, S, y, C).block(h2, Kv.astype(float))
    print("h2 = %.4f  S range %.3g .. %.3g  Sd_max/Sd_min = %.3g" % (h2, S.min(), S.max(), (h2 * S.max() + 1 - h2) / (h2 * S.min() + 1 - h2)))
    for L in (0, 4, 5, 6):                                    # 0 = the automatic choice
        e = Engine(N); e.lmm_setup(U, S, y, C, h2, n_limbs=L)
        r = e.lmm_batch(pack_variants(Kv)); e.close()
        ok = np.isfinite(ws) & (ws > 1e-7)
        rel = lambda a, b: np.nanmax(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-300))
        print("   L=%d: max rel dev  beta %.2e  bse %.2e  p %.2e   (lineage-marker variants: beta %.2e)" % (
            L, rel(r["beta"], wb), rel(r["bse"], ws), rel(r["pvalue"], wp),
            np.nanmax(np.abs(r["beta"][:len(Kv) // 2] - wb[:len(Kv) // 2]) / np.maximum(np.abs(wb[:len(Kv) // 2]), 1e-300))))
