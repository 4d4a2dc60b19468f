"""A million random + strong-effect logistic rows per sample count against the oracle, with ASSERTED maxima per field ("next
round" 1b): fixed_effects_regression (model.py:202-394) through sh_glm_batch at N = 1000 and N = 5000, 10 covariates (one binary, one
age-like: mean 50, sd 10), half of the rows at uniform allele frequencies, half from a U-shaped distribution, 15 % with real / near-separating effects
(|kbeta| 2 .. 6).  One timed bench step is 2^20 (N = 1000) / 2^18 (N = 5000) rows; the fixtures the reference made hold 64 .. 256 rows each.

What is asserted (measured values in profiles/r06/glm_sweep_2e20.txt):
  * notes bit-exact, except rows the ORACLE calls firth-fail (its rounding-noise failures, DESIGN.md section 6: the HIP path fits them) --
    every one of those must carry SH_FLAG_FIRTH_SENSITIVE (bit 18), and there are at most 1e-4 of them;
  * Newton-fitted rows: kbeta <= 5e-8, intercept <= 1e-7, bse <= 2e-7 relative (north_star: 1e-6; kbeta against max(|kbeta|, 1e-4) -- a
    five-hundredth of the smallest bse --, the intercept against max(|intercept|, 1e-2): in a million rows one cancels to 1e-4).  bse is the thin one: the final
    information matrix carries single-precision DIFFERENCES from the exact null-model matrix (csrc/glm_kernels.hip: eta_const, weight_even,
    GLM_DIRECT_ROW); before round 6 this sweep measured 2.3e-7 .. 3.0e-7 on a tenth of the rows;
  * p-value <= 1e-6 relative, where a likelihood-ratio statistic below the noise of its two log-likelihoods (32 ulp of |llf|: 2e-11 at
    N = 5000) is compared as a statistic, not as its tail (p = 1 - 2e-6 against p = 1 is lr = 1e-11 against lr <= 0);
  * Firth-routed rows (bad-chisq / high-bse / separation): 1e-6 relative + the 3e-7 halving-tie slack on the rows the tie detector names
    (tests/_firth_tol.py).  (The covariate is age-like, not offset by thousands: with a column 2000 +- 10 the reference's own Firth answers move by
    4e-6 under a 2e-13 |F| bias of its halving test on half of the routed rows -- its output there is noise; the Newton rows do not care.)"""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CEIL = {"kbeta": 5e-8, "intercept": 1e-7, "bse": 2e-7}
FLOOR = {"kbeta": 1e-4, "intercept": 1e-2, "bse": 0.0}     # |d| / max(|want|, floor): one row in a million has an intercept that cancels to ~1e-4


def _rows(rng, v, N, y):
    af = np.concatenate([rng.uniform(0.02, 0.98, v // 2), rng.beta(0.3, 0.3, v - v // 2)]).astype(np.float32)
    K = rng.random((v, N), dtype=np.float32) < af[:, None]
    eff = np.flatnonzero(rng.random(v) < 0.15)
    hi = (0.05 + 0.8 * (y > np.median(y))).astype(np.float32)
    K[eff] = rng.random((eff.size, N), dtype=np.float32) < hi[None, :] * rng.uniform(0.1, 1.0, eff.size).astype(np.float32)[:, None]
    m = K.mean(axis=1)
    return K[(m >= 0.01) & (m <= 0.99)].astype(np.uint8)


@pytest.mark.parametrize("N,q,seed", [(1000, 10, 12), (5000, 10, 11)])
def test_a_million_logistic_rows_against_the_oracle_with_asserted_maxima(N, q, seed):
    from scipy.special import erfcinv
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _firth_tol import firth_rows_close
    V = int(os.environ.get("SEERHIP_SWEEP_ROWS", 1 << 20))
    CH = 32768
    orc.set_threads(max(1, min(os.cpu_count() or 4, len(os.sched_getaffinity(0)), 64)))
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((N, q)); W[:, 0] = rng.random(N) < 0.3; W[:, 1] = 50 + 10 * W[:, 1]
    eta = -0.5 + 0.9 * W[:, 0] + 0.5 * W[:, 2]
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    lr_noise = 32 * 2.2e-16 * abs(nl)
    e = Engine(N); e.glm_setup(y, W, False, nl, nf)
    mx = {k: 0.0 for k in ("kbeta", "intercept", "bse", "pvalue", "betas", "firth_kbeta", "firth_bse")}
    rows = newton = firth_rows = oracle_fail = 0
    strong = tie_rows = 0
    t_or = t_gpu = 0.0
    while rows < V:
        K = _rows(rng, CH, N, y)
        t0 = time.time(); want = orc.fixed_effects_batch(y, K.astype(float), W, False, 1.0, 1.0, nl, nf); t_or += time.time() - t0
        t0 = time.time(); r = e.glm_batch(pack_variants(K)); t_gpu += time.time() - t0
        notes = want["notes"]; fl = r["flags"]
        ofail = (notes & 0x40) != 0                                     # the oracle's firth-fail (rounding-noise failures: fitted here)
        diff = (fl & 0x1FF) != notes
        assert not (diff & ~ofail).any(), ("notes differ", np.flatnonzero(diff & ~ofail)[:5], fl[diff & ~ofail][:5] & 0x1FF, notes[diff & ~ofail][:5])
        assert (((fl >> 18) & 1)[ofail & diff] == 1).all(), "a row the oracle fails and this library fits does not carry SH_FLAG_FIRTH_SENSITIVE"
        oracle_fail += int((ofail & diff).sum())
        fr = ((notes & 0x7C) != 0) & ~ofail
        nw = ~fr & ~ofail & np.isfinite(want["kbeta"])
        rows += K.shape[0]; newton += int(nw.sum()); firth_rows += int(fr.sum()); strong += int((np.abs(want["kbeta"][nw]) >= 2).sum())
        for f in ("kbeta", "intercept", "bse"):
            a, b = r[f][nw], want[f][nw]
            assert np.isfinite(a).all(), f
            mx[f] = max(mx[f], float(np.max(np.abs(a - b) / np.maximum(np.abs(b), FLOOR[f]))))
        mx["betas"] = max(mx["betas"], float(np.max(np.abs(r["betas"][nw] - want["betas"][nw]) / np.maximum(np.abs(want["betas"][nw]), 1e-3))))
        # p: relative, or -- for statistics inside the noise of the two log-likelihoods -- through the statistic
        pg, pw = r["pvalue"][nw], want["pvalue"][nw]
        lrw = 2 * erfcinv(np.clip(pw, 1e-300, 1.0)) ** 2
        allow = 1e-6 * pw + lr_noise / np.sqrt(2 * np.pi * np.maximum(lrw, lr_noise))
        bad = np.abs(pg - pw) > allow
        assert not bad.any(), ("pvalue", pg[bad][:5], pw[bad][:5])
        big = lrw > 1e-6
        if big.any():
            mx["pvalue"] = max(mx["pvalue"], float(np.max(np.abs(pg[big] - pw[big]) / np.maximum(pw[big], 1e-300))))
        if fr.any():
            # (tests/_firth_tol.py: 1e-6 relative, the 3e-7 slack only where the tie detector fires on the row)
            Kf = K[fr].astype(float)
            vs = orc.firth_noise_variants(lambda: orc.fixed_effects_batch(y, Kf, W, False, 1.0, 1.0, nl, nf))
            allr = np.ones(int(fr.sum()), bool)
            for f in ("kbeta", "bse", "intercept"):
                good, nt = firth_rows_close(r[f][fr], vs, f, allr); tie_rows += nt
                assert good.all(), ("firth " + f, r[f][fr][~good][:5], vs[0][f][~good][:5])
            mx["firth_kbeta"] = max(mx["firth_kbeta"], float(np.max(np.abs(r["kbeta"][fr] - want["kbeta"][fr]))))
            mx["firth_bse"] = max(mx["firth_bse"], float(np.max(np.abs(r["bse"][fr] - want["bse"][fr]) / want["bse"][fr])))
    e.close()
    print("N=%d q=%d: %d rows (%d Newton-fitted, %d of them |kbeta| >= 2; %d Firth-routed, %d values of those on a halving tie; %d oracle firth-fails fitted here), oracle %.0f s, engine calls %.1f s; "
          "max relative deviation %s" % (N, q, rows, newton, strong, firth_rows, tie_rows, oracle_fail, t_or, t_gpu, {k: float("%.3g" % v) for k, v in mx.items()}))
    assert rows >= V and strong >= 0.1 * newton
    for f, c in CEIL.items():
        assert mx[f] <= c, (f, mx[f], c)
    assert mx["pvalue"] <= 1e-6 and mx["betas"] <= 1e-6
    assert oracle_fail <= 1e-4 * rows
