"""Development aid: forced-Firth fits against the oracle on the sweep design of tools/gpu_glm_sweep.py (strong effects, awkward covariates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null
N = int(os.environ.get("N", 1000)); q = int(os.environ.get("Q", 10)); V = int(os.environ.get("V", 20000)); seed = int(os.environ.get("SEED", 1))
rng = np.random.default_rng(seed)
W = rng.standard_normal((N, q)); W[:, 0] = rng.random(N) < 0.3; W[:, 1] = 2000 + 10 * W[:, 1]
eta = -0.5 + 0.9 * W[:, 0] + 0.5 * W[:, min(2, q - 1)]
y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
af = np.concatenate([rng.uniform(0.02, 0.98, V // 2), rng.beta(0.3, 0.3, V - V // 2)])
K = (rng.random((V, N)) < af[:, None])
eff = rng.random(V) < 0.15
K[eff] = rng.random((int(eff.sum()), N)) < (0.05 + 0.8 * y)[None, :] * rng.uniform(0.1, 1.0, int(eff.sum()))[:, None]
K = K.astype(np.uint8); K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)]
e0 = np.zeros((0, 0))
nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
t0 = time.time(); want = orc.firth_batch(y, K.astype(float), W); t1 = time.time()
e = Engine(N); e.glm_setup(y, W, False, nl, nf, force_firth=True); r = e.glm_batch(pack_variants(K)); e.close()
failed = ((r["flags"] >> 6) & 1) == 1
ok = (want["status"] == 0) & ~failed
out = {"N": N, "variants": int(K.shape[0]), "oracle_s": round(t1 - t0, 1), "oracle_failed": int((want["status"] != 0).sum()), "ours_failed": int(failed.sum()),
       "status_mismatch": int((failed != (want["status"] != 0)).sum())}
for f in ("kbeta", "bse", "intercept"):
    d = np.abs(r[f][ok] - want[f][ok]) / np.maximum(np.abs(want[f][ok]), 1e-6)
    out[f] = (float(np.max(d)), float(np.quantile(d, 0.999)), int((d > 1e-6).sum()))
    da = np.abs(r[f][ok] - want[f][ok])
    out[f + "_abs"] = (float(np.max(da)), float(np.quantile(da, 0.999)), int((da > 1e-6).sum()), int((da > 1e-7).sum()))
print(out)
