"""SEERHIP_ROUTE firth_strict=1: the Firth step-halving test is the reference's literal `F(new) > F(old)` (pyseer/model.py:465-474) -- no
"accept steps below 1e-10", no "an increase within 4 ulp of F is noise" (case 1; GlmParams.firth_noise /
firth_accept).  In this mode the kernels are compared with the UNMODIFIED oracle (orc_fit_firth with both test knobs at 0), spurious
firth-fails included.  The only fallback allowed is exact-tie detection: a row that differs must equal the oracle re-run with the first
comparison of each iteration biased by +-2e-13 |F| (orc_set_firth_tie) -- i.e. the reference itself decided that halving on the last
bits of F -- and with the accept-below knob still 0.  The number of rows that needed the fallback is printed and bounded."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pytestmark = pytest.mark.gpu

from test_fuzz_gpu import _firth_case, _close  # noqa: E402

TIES_ONLY = [(2e-13, 0.0), (-2e-13, 0.0)]


def _oracle_variants(fn):
    from oracle import oracle as orc
    out = []
    for tie, eps in [(0.0, 0.0)] + TIES_ONLY:
        orc.set_firth_tie(tie); orc.set_firth_accept_below(eps)
        try:
            out.append(fn())
        finally:
            orc.set_firth_tie(0.0); orc.set_firth_accept_below(0.0)
    return out


def _run_strict(monkeypatch, N, q, W, y, K, nl, nf):
    from pyseer_amd.engine import Engine, pack_variants
    monkeypatch.setenv("SEERHIP_ROUTE", "firth_strict=1")
    e = Engine(N)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)             # the switch is read by sh_glm_setup
    r = e.glm_batch(pack_variants(K))
    e.close()
    monkeypatch.delenv("SEERHIP_ROUTE")
    return r


@pytest.mark.parametrize("seed", range(16))
def test_strict_firth_matches_the_unmodified_oracle(seed, monkeypatch):
    from oracle import oracle as orc
    from pyseer_amd.model import fit_null
    N, q, W, y, K = _firth_case(seed)
    e0 = np.zeros((0, 0))
    null = fit_null(y, W, e0, False); nf = fit_null(y, W, e0, False, firth=True)
    if null is None or nf is None:
        pytest.skip("null model not estimable for this draw")
    r = _run_strict(monkeypatch, N, q, W, y, K, null.llf, nf)
    wants = _oracle_variants(lambda: orc.firth_batch(y, K.astype(float), W if q else None))
    failed = ((r["flags"] >> 6) & 1) == 1

    def rows_matching(w):
        ok = (w["status"] == 0) & np.isfinite(w["fitll"])
        m = failed == (w["status"] != 0)                           # firth-fail exactly where the reference fails, spurious ones included
        for f in ("intercept", "kbeta", "bse"):
            m &= ~ok | _close(r[f], w[f], rtol=2e-6, atol=1e-6)
        with np.errstate(invalid="ignore"):
            lr = -2 * (nf - w["fitll"])
        want_p = np.array([orc.chi2_sf1(x) if x > 0 else 1.0 for x in lr])
        return m & (~ok | _close(r["pvalue"], want_p, rtol=5e-6, atol=1e-300))
    strict = rows_matching(wants[0])
    tied = strict.copy()
    for w in wants[1:]:
        tied |= rows_matching(w)
    nfall = int((tied & ~strict).sum())
    print("strict Firth seed %d: N=%d q=%d rows=%d, matched the unmodified oracle %d, needed the exact-tie re-run %d, reference firth-fails %d"
          % (seed, N, q, K.shape[0], int(strict.sum()), nfall, int((wants[0]["status"] != 0).sum())))
    assert tied.all(), (N, q, np.argwhere(~tied)[:4].tolist())
    assert nfall <= max(1, K.shape[0] // 20), nfall


def _run_env(monkeypatch, env, N, W, y, K, nl, nf):
    from pyseer_amd.engine import Engine, pack_variants
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = Engine(N); e.glm_setup(y, W, False, nl, nf, force_firth=True)
    r = e.glm_batch(pack_variants(K)); e.close()
    for k in env:
        monkeypatch.delenv(k)
    return r


def test_the_three_firth_modes_differ_only_where_documented(monkeypatch):
    """One forced-Firth batch in the three modes: default (the two noise rules: an increase of F within 4 ulp is not an increase, steps
    below 1e-10 are accepted), SEERHIP_ROUTE firth_literal=1 (the reference's literal `F(new) > F(old)` on the rounds' own evaluation of F; the
    1000-step walk at the fixed point of the halving map is cut short with the same verdict) and SEERHIP_ROUTE firth_strict=1 (literal rule, one
    log per sample, the reference's start vector).  Statistics agree to the noise floor of the halving test (3e-7 absolute) wherever two
    modes both converge; the only flag that may differ is firth-fail (with the filter bits that follow it); the default mode never fails
    where a literal mode converges -- a literal-mode failure on such a row is the reference's spurious step_limit exhaustion, which depends
    on the last bit of F and therefore on the evaluation order (profiles/r03/firth_modes_vs_reference_restatement.json)."""
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(44)
    N, q, V = 700, 6, 2048
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = (rng.random(N) < 1 / (1 + np.exp(0.2 - W[:, 0]))).astype(float)
    K = (rng.random((V, N)) < rng.uniform(0.02, 0.98, V)[:, None]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    d = _run_env(monkeypatch, {"SEERHIP_ROUTE": "firth_literal=1"}, N, W, y, K, nl, nf)
    t = _run_env(monkeypatch, {}, N, W, y, K, nl, nf)
    s = _run_env(monkeypatch, {"SEERHIP_ROUTE": "firth_strict=1"}, N, W, y, K, nl, nf)
    ff = {k: (r["flags"] >> 6) & 1 for k, r in (("literal", d), ("default", t), ("strict", s))}
    assert (ff["default"] <= ff["literal"]).all() and (ff["default"] <= ff["strict"]).all()
    for a, b, na, nb in ((d, t, "literal", "default"), (d, s, "literal", "strict"), (t, s, "default", "strict")):
        both = (ff[na] == 0) & (ff[nb] == 0)
        for f in ("kbeta", "bse", "intercept"):
            assert _close(a[f][both], b[f][both], rtol=1e-6, atol=3e-7).all(), (na, nb, f)
        assert ((a["flags"] ^ b["flags"]) & ~np.uint32((1 << 6) | (1 << 8) | (1 << 17) | (1 << 18)))[...].max() == 0   # (bit 18: a firth-fail of this library marks the row order-sensitive)
    print("%d forced-Firth fits (N=%d): firth-fail default %d, literal %d, strict %d; max |dkbeta| literal vs strict %.2e"
          % (V, N, int(ff["default"].sum()), int(ff["literal"].sum()), int(ff["strict"].sum()),
             float(np.nanmax(np.abs(d["kbeta"] - s["kbeta"])[(ff["literal"] == 0) & (ff["strict"] == 0)]))))
