"""One-off scale check: N beyond 46 341 samples (N*N > 2^31) through the LMM and logistic paths against the oracle on a few variants.
Index arithmetic that is fine at N = 5000 overflows 32 bits here.  Needs ~60 GB of host memory and a few minutes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.lmm import mask_like_fit_lmm
from pyseer_amd.model import fit_null

N = int(os.environ.get("N", 46400)); V = 40
rng = np.random.default_rng(7)
t0 = time.time()
k = N - 1
U = rng.standard_normal((N, k), dtype=np.float32).astype(np.float64); U /= np.sqrt(N)
S = np.sort(rng.gamma(0.5, 2.0, k))[::-1].copy()
covar = np.ones((N, 1)); y = (rng.random(N) < 0.4).astype(float); h2 = 0.37
af = np.concatenate([rng.uniform(0.05, 0.95, V - 4), [0.002, 0.999, 0.5, 0.011]])
Kv = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
print("inputs %.0f s" % (time.time() - t0), flush=True)
t0 = time.time()
e = Engine(N); e.set_af_filter(0.01, 0.99)
e.lmm_setup(U, S, y, covar, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0)
r = mask_like_fit_lmm(e.lmm_batch(pack_variants(Kv)))
print("engine set-up + batch %.0f s" % (time.time() - t0), e.lmm_info(), flush=True)
e.close()
t0 = time.time()
afm = (~((Kv.mean(axis=1) >= 0.01) & (Kv.mean(axis=1) <= 0.99))).astype(np.uint8)
want = orc.LmmOracle(U, S, y, covar).fit_lmm(h2, Kv.astype(float), afm, False, 1.0, 1.0)
print("oracle %.0f s" % (time.time() - t0), flush=True)
ok = True
for f, g in (("prep", "prep"), ("pvalue", "pvalue"), ("beta", "kbeta"), ("bse", "bse"), ("frac_h2", "frac_h2")):
    a, b = r[f], want[g]
    with np.errstate(invalid="ignore"):
        m = np.isclose(a, b, rtol=1e-6, atol=1e-12) | (np.isnan(a) & np.isnan(b))
    print("LMM %-8s max rel dev %.2e  %s" % (f, np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)), "ok" if m.all() else "MISMATCH at %s" % np.flatnonzero(~m)[:5]))
    ok &= bool(m.all())
ok &= bool(((r["flags"] & 0x1FF) == want["notes"]).all())
del U
W = rng.standard_normal((N, 2)); W /= np.abs(W).max(axis=0)
yb = (rng.random(N) < 1 / (1 + np.exp(0.3 - W[:, 0]))).astype(float)
e0 = np.zeros((0, 0)); nl = fit_null(yb, W, e0, False).llf; nf = fit_null(yb, W, e0, False, firth=True)
wantg = orc.fixed_effects_batch(yb, Kv[:-4].astype(float), W, False, 1.0, 1.0, nl, nf)
e = Engine(N); e.glm_setup(yb, W, False, nl, nf); rg = e.glm_batch(pack_variants(Kv[:-4])); e.close()
for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
    m = np.isclose(rg[f], wantg[f], rtol=1e-6, atol=1e-12) | (np.isnan(rg[f]) & np.isnan(wantg[f]))
    print("GLM %-9s %s" % (f, "ok" if m.all() else "MISMATCH at %s" % np.flatnonzero(~m)[:5])); ok &= bool(m.all())
print("RESULT", "PASS" if ok else "FAIL")
