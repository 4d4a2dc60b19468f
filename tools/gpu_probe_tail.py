"""Development probe: latency of the list-driven / small-batch paths at N = 5000 (a handful of variants per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null

N, q = int(os.environ.get("N", 5000)), 10
rng = np.random.default_rng(5)
W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
W[:, q - 1] = (rng.random(N) < 0.3).astype(float)                      # a binary covariate
eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
e0 = np.zeros((0, 0))
nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
V = 4096
K = (rng.random((V, N)) < rng.uniform(0.05, 0.95, (V, 1))).astype(np.uint8)


def timed(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3


e = Engine(N); e.glm_setup(y, W, False, nl, nf)
base = pack_variants(K)
print("logistic, %d ordinary variants: %.2f ms" % (V, timed(lambda: e.glm_batch(base))))
K2 = K.copy(); K2[7] = W[:, q - 1].astype(np.uint8)                      # a k-mer identical to the binary covariate -> pinv path
b2 = pack_variants(K2)
print("  + 1 variant equal to a covariate (pinv Firth): %.2f ms" % timed(lambda: e.glm_batch(b2)), e.glm_info())
K3 = K.copy(); K3[9] = y.astype(np.uint8); K3[11] = (y * (rng.random(N) < 0.9)).astype(np.uint8)   # separated / quasi-separated
b3 = pack_variants(K3)
print("  + 2 (quasi-)separated variants (fp64 restart + Firth): %.2f ms" % timed(lambda: e.glm_batch(b3)), e.glm_info())
e.close()
lin = rng.standard_normal((N, 9)); lin /= np.abs(lin).max(axis=0)
e = Engine(N); e.lineage_setup(lin, None)
Kbig = np.tile(K, (32, 1))
for n in (1, 64, 4096, 131072):
    bb = pack_variants(Kbig[:n])
    print("lineage_batch, %d variants: %.2f ms" % (n, timed(lambda: e.lineage_batch(bb))))
e.close()
yc = eta + rng.standard_normal(N)
nlc = fit_null(yc, W, e0, True).llf
e = Engine(N); e.glm_setup(yc, W, True, nlc, None)
print("OLS, %d ordinary variants: %.2f ms" % (V, timed(lambda: e.glm_batch(base))))
print("  + 1 variant equal to a covariate (rank-deficient, pinv OLS): %.2f ms" % timed(lambda: e.glm_batch(b2)), e.glm_info())
e.close()
