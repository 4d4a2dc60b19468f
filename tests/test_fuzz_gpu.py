"""Seeded random sweep of small configurations through both engines against the oracle: sample counts around word and tile
edges, few and many covariates, binary and continuous phenotypes, rare and near-fixed variants, random thresholds."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-6, atol=0.0):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
        (np.abs(a - b) <= atol + rtol * np.abs(b))


def _variants(rng, N, V):
    af = np.concatenate([rng.uniform(0.03, 0.97, V - 8), [1.5 / N, 2.5 / N, 1 - 1.5 / N, 0.5, 0.5, 0.02, 0.98, 0.5]])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[-1] = 0; K[-1, :max(1, N // 50)] = 1                      # a handful of carriers at the front
    return K


def _fixed_case(seed):
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.choice([40, 63, 64, 65, 100, 128, 129, 250, 300]))
    q = int(rng.choice([0, 1, 2, 3, 5, 8, 11, 14, 16]))
    cont = bool(rng.integers(0, 2))
    V = 72
    W = rng.standard_normal((N, q))
    if q:
        W /= np.abs(W).max(axis=0)
    eta = -0.2 + (1.0 * W[:, 0] if q else 0.0)
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    pret, lrtt = (1.0, 1.0) if seed % 3 == 0 else (float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.3, 0.9)))
    K = _variants(rng, N, V)
    af = K.mean(axis=1)
    K = K[(af >= 0.01) & (af <= 0.99)]
    return N, q, cont, W, y, pret, lrtt, K


def _firth_case(seed):
    rng = np.random.default_rng(3000 + seed)
    N = int(rng.choice([60, 64, 100, 129, 200, 320]))
    q = int(rng.choice([0, 1, 3, 6, 10, 14, 15]))
    V = 40
    W = rng.standard_normal((N, q))
    if q:
        W /= np.abs(W).max(axis=0)
    eta = -0.2 + (1.0 * W[:, 0] if q else 0.0)
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    K = _variants(rng, N, V)
    af = K.mean(axis=1)
    K = K[(af >= 0.01) & (af <= 0.99)]
    return N, q, W, y, K


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEEDS", 12))))
def test_lmm_random_configurations(seed):
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.lmm import mask_like_fit_lmm
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([31, 63, 64, 65, 127, 128, 129, 200, 257, 383]))
    D = int(rng.integers(1, 6))
    cont = bool(rng.integers(0, 2))
    V = 72
    k = N - D
    U = rng.standard_normal((N, k)) / np.sqrt(N)
    S = np.sort(rng.gamma(0.5, 2.0, k))[::-1].copy()
    covar = np.ones((N, 1)) if D == 1 else np.c_[rng.standard_normal((N, D - 1)), np.ones((N, 1))]
    y = rng.standard_normal(N) if cont else (rng.random(N) < 0.4).astype(float)
    h2 = float(rng.uniform(0.0, 0.95))
    fp, lp = (1.0, 1.0) if seed % 3 == 0 else (float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.2, 0.9)))
    Kv = _variants(rng, N, V)
    af = Kv.mean(axis=1)
    afm = (~((af >= 0.01) & (af <= 0.99))).astype(np.uint8)      # 1 = af-filtered (load_var_block leaves pattern None)
    want = orc.LmmOracle(U, S, y, covar).fit_lmm(h2, Kv.astype(float), afm, cont, fp, lp)
    e = Engine(N)
    e.set_af_filter(0.01, 0.99)
    e.lmm_setup(U, S, y, covar, h2, continuous=cont, filter_pvalue=fp, lrt_pvalue=lp)
    r = mask_like_fit_lmm(e.lmm_batch(pack_variants(Kv)))
    e.close()
    noise = np.zeros(V, bool) if cont else ((Kv == y).all(axis=1) | (Kv == 1 - y).all(axis=1))
    for f, g in (("prep", "prep"), ("pvalue", "pvalue"), ("beta", "kbeta"), ("bse", "bse"), ("frac_h2", "frac_h2")):
        ok = _close(r[f], want[g], atol=1e-12 if f in ("beta", "frac_h2") else 1e-300) | noise
        assert ok.all(), (f, N, D, cont, np.argwhere(~ok)[:4].tolist(), r[f][~ok][:4], want[g][~ok][:4])
    assert (((r["flags"] & 0x1FF) == want["notes"]) | noise).all()
    assert ((((r["flags"] >> 16) & 1) == want["prefilter"])).all()
    assert (((((r["flags"] >> 17) & 1) == want["filter"])) | noise).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEEDS", 12))))
def test_fixed_effects_random_configurations(seed):
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    N, q, cont, W, y, pret, lrtt, K = _fixed_case(seed)
    e0 = np.zeros((0, 0))
    null = fit_null(y, W, e0, cont)
    if null is None:
        pytest.skip("null model not estimable for this draw")
    nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    if nf is None:
        pytest.skip("Firth null model did not converge for this draw")
    want = orc.fixed_effects_batch(y, K.astype(float), W if q else None, cont, pret, lrtt, null.llf, nf)
    e = Engine(N)
    e.glm_setup(y, W, cont, null.llf, nf, pret, lrtt)
    r = e.glm_batch(pack_variants(K))
    e.close()
    firth = (want["notes"] & 0x7C) != 0
    # A Firth fit whose step halving the reference decides on a last-bit tie of F (case 1) has two
    # legitimate answers; the oracle is re-run with the comparison biased by +-2e-13*|F| and either answer is accepted.
    alts = [want]
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        def close_to(w):
            return _close(r[f], w[f], rtol=np.where(firth, 2e-6, 1e-6), atol=np.where(firth, 1e-6 if f != "pvalue" else 1e-300, 1e-12))
        ok = close_to(want)
        if not ok.all() and firth[~ok].all():
            if len(alts) == 1:
                for tie in (2e-13, -2e-13):
                    orc.set_firth_tie(tie)
                    try:
                        alts.append(orc.fixed_effects_batch(y, K.astype(float), W if q else None, cont, pret, lrtt, null.llf, nf))
                    finally:
                        orc.set_firth_tie(0.0)
            ok = ok | close_to(alts[1]) | close_to(alts[2])
        assert ok.all(), (f, N, q, cont, np.argwhere(~ok)[:4].tolist(), r[f][~ok][:4], want[f][~ok][:4])
    tested = np.isfinite(want["kbeta"])
    if q:
        okb = np.zeros(r["betas"][tested].shape, bool)
        for w in alts:
            okb |= _close(r["betas"][tested], w["betas"][tested], rtol=2e-6, atol=1e-6)
        assert okb.all()
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()
    assert (((r["flags"] >> 16) & 1) == want["prefilter"]).all() and (((r["flags"] >> 17) & 1) == want["filter"]).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEEDS", 12))))
def test_forced_firth_random_configurations(seed):
    """Every variant through fit_firth (BASELINE config C4 shape, small): state-machine kernels, and the run-time-width kernel for
    q > 14, against the oracle's fit_firth."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    N, q, W, y, K = _firth_case(seed)
    e0 = np.zeros((0, 0))
    null = fit_null(y, W, e0, False)
    nf = fit_null(y, W, e0, False, firth=True)
    if null is None or nf is None:
        pytest.skip("null model not estimable for this draw")
    want = orc.firth_batch(y, K.astype(float), W if q else None)
    e = Engine(N)
    e.glm_setup(y, W, False, null.llf, nf, force_firth=True)
    r = e.glm_batch(pack_variants(K))
    e.close()
    ok = (want["status"] == 0) & np.isfinite(want["fitll"])
    failed = ((r["flags"] >> 6) & 1) == 1
    assert (failed[want["status"] != 0]).all() and not failed[ok].any()
    def pvals(w):
        return np.array([orc.chi2_sf1(x) if x > 0 else 1.0 for x in -2 * (nf - w["fitll"][ok])])

    def matches(w):
        m = np.ones(int(ok.sum()), bool)
        for f in ("intercept", "kbeta", "bse"):
            m &= _close(r[f][ok], w[f][ok], rtol=2e-6, atol=1e-6)
        return m & _close(r["pvalue"][ok], pvals(w), rtol=5e-6, atol=1e-300)
    good = matches(want)
    if not good.all():                                            # last-bit ties of the halving test: see the test above
        for tie in (2e-13, -2e-13):
            orc.set_firth_tie(tie)
            try:
                good |= matches(orc.firth_batch(y, K.astype(float), W if q else None))
            finally:
                orc.set_firth_tie(0.0)
    assert good.all(), (N, q, np.argwhere(~good)[:4].tolist())
