"""Development aid: a large randomised comparison of the logistic path against the oracle (N, q, V from the environment): maximum relative
deviation per field over the Newton rows, note mismatches, and how many rows went through Firth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null
N = int(os.environ.get("N", 1000)); q = int(os.environ.get("Q", 10)); V = int(os.environ.get("V", 100000)); seed = int(os.environ.get("SEED", 1)); CONT = int(os.environ.get("CONT", 0))
rng = np.random.default_rng(seed)
W = rng.standard_normal((N, q)); W[:, 0] = rng.random(N) < 0.3; W[:, 1] = 2000 + 10 * W[:, 1]       # a binary and an un-centred column
eta = -0.5 + 0.9 * W[:, 0] + 0.5 * W[:, min(2, q - 1)]
y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
if CONT:
    y = eta + rng.standard_normal(N)
af = np.concatenate([rng.uniform(0.02, 0.98, V // 2), rng.beta(0.3, 0.3, V - V // 2)])
K = (rng.random((V, N)) < af[:, None])
eff = rng.random(V) < 0.15                                        # some real effects, some near-separating
K[eff] = rng.random((int(eff.sum()), N)) < (0.05 + 0.8 * (y > np.median(y)))[None, :] * rng.uniform(0.1, 1.0, int(eff.sum()))[:, None]
K = K.astype(np.uint8)
K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)]
e0 = np.zeros((0, 0))
nl = fit_null(y, W, e0, bool(CONT)).llf; nf = np.nan if CONT else fit_null(y, W, e0, False, firth=True)
t0 = time.time(); want = orc.fixed_effects_batch(y, K.astype(float), W, bool(CONT), 1.0, 1.0, nl, nf); t1 = time.time()
e = Engine(N); e.glm_setup(y, W, bool(CONT), nl, nf); r = e.glm_batch(pack_variants(K)); e.close()
firth = (want["notes"] & 0x7C) != 0
out = {"N": N, "q": q, "variants": int(K.shape[0]), "oracle_s": round(t1 - t0, 1), "firth_rows": int(firth.sum()),
       "note_mismatches": int(((r["flags"] & 0x1FF) != want["notes"]).sum())}
for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
    a, b = r[f][~firth], want[f][~firth]
    ok = np.isfinite(b)
    out[f] = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-300))) if ok.any() else 0.0
    out[f + "_nan_mismatch"] = int((np.isfinite(a) != np.isfinite(b)).sum())
print(out)
bad = np.flatnonzero((r["flags"] & 0x1FF) != want["notes"])
for i in bad[:12]:
    print("  note mismatch row %d: got %#x want %#x  af %.4f  kbeta %.6g/%.6g bse %.6g/%.6g p %.4g/%.4g" % (i, r["flags"][i] & 0x1FF, want["notes"][i], K[i].mean(),
          r["kbeta"][i], want["kbeta"][i], r["bse"][i], want["bse"][i], r["pvalue"][i], want["pvalue"][i]))
worst = np.argsort(-np.abs(np.where(firth | ~np.isfinite(want["bse"]), 0, (r["bse"] - want["bse"]) / want["bse"])))[:5]
for i in worst:
    print("  worst bse row %d: af %.4f kbeta %.6g bse %.9g/%.9g rel %.2e" % (i, K[i].mean(), want["kbeta"][i], r["bse"][i], want["bse"][i], (r["bse"][i] - want["bse"][i]) / want["bse"][i]))
worst = np.argsort(-np.abs(np.where(firth | ~np.isfinite(want["pvalue"]) | (want["pvalue"] == 0), 0, (r["pvalue"] - want["pvalue"]) / np.maximum(want["pvalue"], 1e-300))))[:5]
for i in worst:
    print("  worst p row %d: af %.4f kbeta %.9g/%.9g p %.9g/%.9g rel %.2e" % (i, K[i].mean(), r["kbeta"][i], want["kbeta"][i], r["pvalue"][i], want["pvalue"][i],
          (r["pvalue"][i] - want["pvalue"][i]) / want["pvalue"][i]))
