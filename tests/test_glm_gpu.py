"""GPU parity of the fixed-effects path (logistic Newton, Firth, OLS kernels through the C ABI) against the reference's
golden vectors and the CPU oracle.  Tolerances: 1e-6 relative (north_star); Firth-fitted values additionally get the
3e-7 absolute slack explained in tests/test_oracle_golden.py (the reference's step-halving test is decided by rounding
noise once steps are < ~1e-6)."""
import glob
import os

import numpy as np
import pytest
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FA = 3e-7


def close(a, b, rtol=1e-6, atol=0.0, what=""):
    a = np.atleast_1d(np.asarray(a, dtype=float)); b = np.atleast_1d(np.asarray(b, dtype=float))
    assert a.shape == b.shape, (a.shape, b.shape)
    with np.errstate(invalid="ignore"):
        ok = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
             (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), "%s mismatch at %s: got %s want %s" % (what, np.argwhere(~ok)[:5].tolist(), a[~ok][:5], b[~ok][:5])


def _run(d, force_firth=False):
    from pyseer_amd.engine import Engine, pack_variants
    y, m, K = d["y"], d["m"], d["K"]
    e = Engine(int(d["N"]))
    e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, m, bool(d["continuous"]), float(d["null_llf"]), float(d["null_firth"]), float(d["pret"]), float(d["lrtt"]),
                force_firth=force_firth)
    r = e.glm_batch(pack_variants(K))
    e.close()
    return r


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "glm_*.npz")) + glob.glob(os.path.join(G, "ols_*.npz"))))
def test_fixed_effects_golden(path):
    d = np.load(path)
    r = _run(d)
    main = d["main"]
    # exactly singular designs (variant column duplicates a covariate): the reference's Firth log det is LAPACK rounding
    # noise -> its p-value is not a parity target (see tests/test_oracle_golden.py)
    dup = np.zeros(main.shape[0], bool)
    if "bincov" in path:
        dup[[8, 9]] = True
    # A `firth-fail` of the reference that it does not repeat with the samples in some other order (perm_notes, make_exit_golden.py) is its
    # rounding-noise failure: F at the candidate exceeds F at the iterate by the last bits of an N-term sum, for
    # every halving.  Such a row may come back as the reference has it, OR as the reference has it in the other orders -- the whole row
    # then: notes, and every statistic to 1e-6 (checked at the end); it is left out of the comparisons with the reference's NaNs.
    fl = r["flags"]
    alt_row = np.zeros(main.shape[0], bool)
    if "perm_notes" in d.files:
        alt_row = ((d["notes"] & 0x40) != 0) & ((d["perm_notes"] & 0x40) == 0).any(axis=1) & ((fl & 0x1FF) != d["notes"])
    dup = dup | alt_row
    close(r["prep"], main[:, 0], what="prep")
    close(r["pvalue"][~dup], main[~dup, 1], atol=1e-300, what="pvalue")
    # rows the reference fitted by Newton / OLS: 1e-6 relative, no slack.  Rows it sent through fit_firth (notes bits 2-6): 1e-6 relative, and
    # the halving-noise slack FA ONLY where the tie detector fires on the row (tests/_firth_tol.py)
    firth = (d["notes"] & 0x7C) != 0
    tested = np.isfinite(main[:, 2])
    plain = tested & ~firth
    for j, f in ((2, "kbeta"), (3, "bse"), (4, "intercept")):
        close(r[f][~firth], main[~firth, j], what=f)
    if int(d["q"]):
        close(r["betas"][plain], d["betas"][plain], atol=1e-12, what="betas")
        assert np.isnan(r["betas"][~tested & ~alt_row]).all()
    fr = firth & tested
    needed = 0
    if fr.any():
        from oracle import oracle as orc
        from _firth_tol import golden_firth_rows_close
        Kf = d["K"][fr].astype(float)
        vs = orc.firth_noise_variants(lambda: orc.fixed_effects_batch(d["y"], Kf, d["m"] if int(d["q"]) else None, False, float(d["pret"]), float(d["lrtt"]),
                                                                      float(d["null_llf"]), float(d["null_firth"])))
        allr = np.ones(int(fr.sum()), bool)
        for j, f in ((2, "kbeta"), (3, "bse"), (4, "intercept")):
            good, nt = golden_firth_rows_close(r[f][fr], main[fr, j], vs, f, allr); needed += nt
            assert good.all(), (f, np.where(fr)[0][~good][:5], r[f][fr][~good][:5], main[fr, j][~good][:5])
        for c in range(int(d["q"])):
            vc = [dict(b=v["betas"][:, c]) for v in vs]
            good, nt = golden_firth_rows_close(r["betas"][fr, c], d["betas"][fr, c], vc, "b", allr); needed += nt
            assert good.all(), ("betas", c, np.where(fr)[0][~good][:5])
    print("%s: %d Firth-routed rows, %d statistic values needed the tie detector" % (os.path.basename(path), int(fr.sum()), needed))
    # (measured in round 5: 0 on every fixture but glm_N300_q10_filt (1) and glm_exit_highbse_N100_q1 (2: the row whose stop rule sits at 1.00003e-4))
    assert needed <= 2, needed
    same = ~alt_row
    for v in np.where(alt_row)[0]:
        alt = [j for j in range(d["perm_notes"].shape[1]) if (fl[v] & 0x1FF) == d["perm_notes"][v, j]]
        assert alt, (v, fl[v] & 0x1FF, d["perm_notes"][v])
        pm = d["perm_main"][v, alt[0]]
        close(np.array([r["prep"][v], r["pvalue"][v], r["kbeta"][v], r["bse"][v], r["intercept"][v]]), pm, atol=1e-300,
              what="row %d as the reference has it in another sample order" % v)
        if int(d["q"]):
            close(r["betas"][v], d["perm_betas"][v, alt[0]], atol=1e-12, what="betas of row %d" % v)
        print("%s: row %d is a rounding-noise firth-fail of the reference; fitted here as the reference fits it in another sample order" % (os.path.basename(path), v))
    assert ((fl & 0x1FF)[same] == d["notes"][same]).all(), np.argwhere(((fl & 0x1FF) != d["notes"]) & same).ravel()
    assert (((fl >> 16) & 1) == d["prefilter"]).all() and (((fl >> 17) & 1)[same] == d["filter"][same]).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "glm_*.npz"))))
def test_forced_firth_golden(path):
    """BASELINE config C4: every variant through fit_firth (model.py:414)."""
    d = np.load(path)
    if float(d["pret"]) < 1:
        pytest.skip("prefilter active in this fixture")
    r = _run(d, force_firth=True)
    ok = d["firth_ok"] == 1
    if "bincov" in path:
        ok[[8, 9]] = False
    fm = d["firth_main"]
    from oracle import oracle as orc
    from _firth_tol import golden_firth_rows_close
    Kf = d["K"][ok].astype(float)
    vs = orc.firth_noise_variants(lambda: orc.firth_batch(d["y"], Kf, d["m"] if int(d["q"]) else None))
    allr = np.ones(int(ok.sum()), bool); needed = 0
    for j, f in ((0, "intercept"), (1, "kbeta"), (2, "bse")):
        good, nt = golden_firth_rows_close(r[f][ok], fm[ok, j], vs, f, allr); needed += nt
        assert good.all(), (f, np.where(ok)[0][~good][:5], r[f][ok][~good][:5], fm[ok, j][~good][:5])
    for c in range(int(d["q"])):
        vc = [dict(b=v["betas"][:, c]) for v in vs]
        good, nt = golden_firth_rows_close(r["betas"][ok, c], d["firth_betas"][ok, c], vc, "b", allr); needed += nt
        assert good.all(), ("betas", c, np.where(ok)[0][~good][:5])
    print("%s: %d forced-Firth rows, %d statistic values needed the tie detector" % (os.path.basename(path), int(ok.sum()), needed))
    assert needed <= 2, needed                                         # (measured: 0 everywhere but glm_exit_highbse_N100_q1: 2)
    # p-value from fitll: lrstat = -2 (null_firth - fitll)
    lr = -2 * (float(d["null_firth"]) - fm[ok, 3])
    want_p = np.array([orc.chi2_sf1(x) if x > 0 else 1.0 for x in lr])
    close(r["pvalue"][ok], want_p, rtol=2e-6, atol=1e-300)


@pytest.mark.parametrize("N,q", [(5000, 10), (1500, 6), (900, 0)])
def test_firth_last_pass_saved_changes_nothing_at_the_bar(N, q, monkeypatch):
    """Forced Firth (C4's mode): by default k_firth_step2 finishes a fit whose stop rule is already met from the beta before its last candidate
    (F to second order, I11 from the factor on record) instead of taking the likelihood pass there; SEERHIP_ROUTE=firth_last=0 takes the pass.  Same
    flags, kbeta / intercept / betas identical (the candidate is formed by the same instructions), p-value to 1e-9, bse to 1e-7."""
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(31 + N)
    V = 512
    W = rng.standard_normal((N, q)); W /= np.maximum(np.abs(W).max(axis=0), 1e-300) if q else 1.0
    eta = -0.3 + (1.1 * W[:, 0] - 0.6 * W[:, 1] if q >= 2 else 0.0)
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.03, 0.97, V - V // 6), rng.uniform(0.004, 0.03, V // 6)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[: V // 8] = (rng.random((V // 8, N)) < (0.2 + 0.5 * y)[None, :]).astype(np.uint8)          # real effects
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    out = []
    for last in ("1", "0"):
        monkeypatch.setenv("SEERHIP_ROUTE", "firth_last=" + last)
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.glm_setup(y, W, False, nl, nf, force_firth=True); out.append(e.glm_batch(bits)); e.close()
    monkeypatch.delenv("SEERHIP_ROUTE")
    a, b = out
    assert np.array_equal(a["flags"], b["flags"])
    ok = np.isfinite(b["kbeta"])
    assert ok.sum() > V // 2
    for f in ("kbeta", "intercept"):
        assert np.array_equal(a[f][ok], b[f][ok]), f
    if q:
        assert np.array_equal(a["betas"][ok], b["betas"][ok])
    close(a["pvalue"][ok], b["pvalue"][ok], rtol=1e-9, atol=1e-300, what="pvalue")
    close(a["bse"][ok], b["bse"][ok], rtol=1e-7, what="bse")


@pytest.mark.parametrize("N,q,V,cont", [(1000, 10, 640, False), (517, 5, 300, False), (800, 7, 256, True), (5000, 10, 192, False)])
def test_fixed_effects_vs_oracle_random(N, q, V, cont):
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(100 + N + q)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 8), rng.uniform(0.002, 0.02, V // 8)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    keep = (K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)
    K = K[keep]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, cont).llf
    nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    wants = orc.firth_noise_variants(lambda: orc.fixed_effects_batch(y, K.astype(float), W, cont, 1.0, 1.0, nl, nf))
    want = wants[0]
    e = Engine(N)
    e.glm_setup(y, W, cont, nl, nf)
    r = e.glm_batch(pack_variants(K))
    e.close()
    firth = (want["notes"] & 0x7C) != 0          # any of bad-chisq/high-bse/sep/inv/firth-fail
    from _firth_tol import firth_rows_close
    needed = 0
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f][~firth], want[f][~firth], what=f)
        # Firth rows: 1e-6 relative; the halving test's noise floor only on rows the tie detector identifies (tests/_firth_tol.py)
        if f == "prep":
            close(r[f][firth], want[f][firth], what=f + "(firth)")
            continue
        good, nt = firth_rows_close(r[f], wants, f, firth)
        needed += nt
        assert good.all(), (f, np.argwhere(~good)[:4].tolist(), r[f][firth][~good][:4], want[f][firth][~good][:4])
    print("N=%d q=%d: %d Firth-routed rows, %d statistic values needed the tie detector" % (N, q, int(firth.sum()), needed))
    assert needed <= 1, needed                                         # (measured: 0 in all four configurations)
    close(r["betas"][~firth], want["betas"][~firth], atol=1e-12, what="betas")
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "lineage_*.npz"))))
def test_lineage_effect_golden(path):
    """a6 fit_lineage_effect (model.py:151-199) through k_glm_lineage."""
    from pyseer_amd.engine import Engine, pack_variants
    d = np.load(path)
    e = Engine(int(d["N"]))
    e.lineage_setup(d["lin"], d["cov"] if int(d["j"]) else None)
    got = e.lineage_batch(pack_variants(d["K"]))
    e.close()
    from test_oracle_golden import _same_or_tied
    _same_or_tied([None if x < 0 else int(x) for x in got], [None if x < 0 else int(x) for x in d["max_lineage"]], d["lin"],
                  d["cov"] if int(d["j"]) else None, d["K"])


@pytest.mark.parametrize("N,q,cont", [(130, 0, False), (333, 14, False), (200, 1, False), (333, 14, True), (130, 0, True)])
def test_design_width_extremes_vs_oracle(N, q, cont):
    """Narrowest (no covariates) and widest (q = 14) register-resident designs, odd sample counts."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(7 + N + q)
    W = rng.standard_normal((N, q)); W = W / np.abs(W).max(axis=0) if q else W
    eta = -0.2 + (0.8 * W[:, 0] if q else 0.0)
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    K = (rng.random((96, N)) < rng.uniform(0.1, 0.9, 96)[:, None]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W if q else e0, e0, cont).llf
    nf = np.nan if cont else fit_null(y, W if q else e0, e0, False, firth=True)
    wants = orc.firth_noise_variants(lambda: orc.fixed_effects_batch(y, K.astype(float), W if q else None, cont, 1.0, 1.0, nl, nf))
    want = wants[0]
    e = Engine(N); e.glm_setup(y, W, cont, nl, nf)
    r = e.glm_batch(pack_variants(K)); e.close()
    firth = (want["notes"] & 0x7C) != 0
    from _firth_tol import firth_rows_close
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f][~firth], want[f][~firth], atol=1e-12 if f in ("kbeta", "intercept") else 0.0, what=f)   # exact zeros come out as +-1e-16
        # Firth rows: 1e-6 relative; the halving-noise slack only on rows where the tie detector fires (tests/_firth_tol.py)
        if f in ("prep", "pvalue"):
            close(r[f][firth], want[f][firth], rtol=1e-6, atol=1e-300, what=f + "(firth)")
        else:
            good, _ = firth_rows_close(r[f], wants, f, firth)
            assert good.all(), (f, np.where(firth)[0][~good][:4], r[f][firth][~good][:4], want[f][firth][~good][:4])
    if q:
        close(r["betas"][~firth], want["betas"][~firth], atol=1e-12, what="betas")
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()


def test_empty_and_single_variant_batches():
    from pyseer_amd.engine import Engine, pack_variants
    d = np.load(os.path.join(G, "glm_N100_q3.npz"))
    e = Engine(100); e.glm_setup(d["y"], d["m"], False, float(d["null_llf"]), float(d["null_firth"]))
    r0 = e.glm_batch(np.zeros((0, 16), dtype=np.uint8))
    assert r0["kbeta"].shape == (0,) and r0["betas"].shape == (0, 3)
    r1 = e.glm_batch(pack_variants(d["K"][20:21]))
    assert (int(d["notes"][20]) & 0x7C) == 0                  # a Newton row: no slack
    close(r1["kbeta"], d["main"][20:21, 2]); close(r1["pvalue"], d["main"][20:21, 1])
    e.close()


def test_ols_rank_deficient_design_follows_pinv():
    """Continuous phenotype, k-mer identical to (or the complement of) a binary covariate: statsmodels' OLS.fit() is pinv-based
    (model.py:299-312), so the reference still reports the minimum-norm coefficients, their pinv standard errors and a t-test
    with N - rank degrees of freedom."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(77)
    N, q = 300, 4
    W = rng.standard_normal((N, q)); W[:, 2] = (rng.random(N) < 0.4).astype(float)
    y = 0.4 * W[:, 0] + 0.8 * W[:, 2] + rng.standard_normal(N)
    K = (rng.random((6, N)) < 0.3).astype(np.uint8)
    K[1] = W[:, 2].astype(np.uint8)                  # duplicates the covariate
    K[3] = 1 - W[:, 2].astype(np.uint8)              # intercept minus the covariate
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, True).llf
    want = orc.fixed_effects_batch(y, K.astype(float), W, True, 1.0, 1.0, nl, np.nan)
    e = Engine(N)
    e.glm_setup(y, W, True, nl, np.nan)
    r = e.glm_batch(pack_variants(K))
    e.close()
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f], want[f], atol=1e-9, what=f)
    close(r["betas"], want["betas"], atol=1e-9, what="betas")
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()


def test_drop_in_signature_with_lineage():
    """pyseer_amd.model.fixed_effects_regression keeps the reference's one-variant signature, lineage effects included
    (model.py:202, :379-382); checked against the oracle's fit_lineage_effect and fixed-effects rows."""
    from oracle import oracle as orc
    from pyseer_amd.model import fixed_effects_regression, fit_null
    d = np.load(os.path.join(G, "lineage_N200_l3.npz"))
    lin, K = d["lin"].astype(float), d["K"].astype(float)
    n = lin.shape[0]
    rng = np.random.default_rng(3)
    y = (rng.random(n) < 0.4).astype(float)
    m = lin[:, :2].copy()
    e0 = np.zeros((n, 0))
    null = fit_null(y, m, e0, False)
    nf = fit_null(y, m, e0, False, firth=True)
    want = orc.fixed_effects_batch(y, K[:12], m, False, 1.0, 1.0, null.llf, nf)
    for v in range(12):
        s = fixed_effects_regression("v%d" % v, y, K[v], m, e0, K[v].mean(), b"", True, lin, 1.0, 1.0, null, nf, [], [], False)
        assert np.isclose(s.kbeta, want["kbeta"][v], rtol=1e-6, atol=1e-9, equal_nan=True)
        if not s.prefilter and 'firth-fail' not in s.notes:
            ref = orc.lineage_effect(lin, None, K[v])
            assert (s.max_lineage is None) == (ref is None)        # the index itself: Wald ties, see test_lineage_effect_golden


@pytest.mark.parametrize("tag", ["bin", "bin_pre", "bin_lrt", "bin_cov", "bin_bad", "cont", "cont_pre", "cont_lrt",
                                 "cont_cov", "bin_nodist", "cont_nodist"])
def test_reference_unit_scenarios_through_the_drop_in(tag):
    """The scenarios of the reference's tests/model_test.py (TestFixedEffectsRegression*, its own N = 100 unit-test data), called the
    way that file calls them -- fixed_effects_regression(variant, p, k, m, c, af, pattern, lineage_effects, lin, pret, lrtt,
    null_res, null_firth, kstrains, nkstrains, continuous) -- and compared like its eq_seer (abs tol 1e-7) with the values the
    reference produced (tests/golden/make_golden.py run_model_unit)."""
    from pyseer_amd.model import fixed_effects_regression, NullFit
    from pyseer_amd.classes import NOTE_ORDER
    d = np.load(os.path.join(G, "model_unit.npz"))
    cont = tag.startswith("cont")
    p = d["p_continuous"] if cont else d["p_binary"]
    k = np.array([1.0] * 5 + [0.0] * 95) if tag == "bin_bad" else d["k"]
    m = np.zeros((100, 0)) if "nodist" in tag else d["m"]
    c = d["cov"] if "cov" in tag else np.zeros((100, 0))
    pret = 0.05 if tag.endswith("_pre") else 1.0
    lrtt = 0.05 if tag.endswith("_lrt") else 1.0
    nl, nf = d["s_%s_null" % tag]
    s = fixed_effects_regression("variant", p, k, m, c, 0.2, b"pattern", False, None, pret, lrtt, NullFit(float(nl), None),
                                 None if cont else float(nf), ["k1"], ["nk1"], cont)
    want = d["s_%s_main" % tag]
    got = np.array([s.prep, s.pvalue, s.kbeta, s.bse, s.intercept])
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want), initial=0.0) < 1e-7
    wb = d["s_%s_betas" % tag]
    if wb.shape[0] and np.isfinite(wb).all():
        assert np.abs(np.asarray(s.betas) - wb).max() < 1e-7
    nm, pf, fl, _ = d["s_%s_flags" % tag]
    assert set(n for i, n in enumerate(NOTE_ORDER) if (int(nm) >> i) & 1) == s.notes
    assert (s.prefilter, s.filter) == (bool(pf), bool(fl))
    assert (s.kmer, s.pattern, s.af, s.kstrains, s.nkstrains, s.max_lineage) == ("variant", b"pattern", 0.2, ["k1"], ["nk1"], None)


def test_pre_filtering_and_fit_lineage_effect_by_name():
    """pyseer.model.pre_filtering / fit_lineage_effect under their own names: the literals of tests/model_test.py:89-115 and the
    lineage goldens."""
    from pyseer_amd.model import pre_filtering, fit_lineage_effect
    d = np.load(os.path.join(G, "model_unit.npz"))
    prep, bad = pre_filtering(d["p_binary"], d["k"], False)
    assert abs(prep - 0.5365065578449575) < 1e-12 and not bad
    prep, bad = pre_filtering(d["p_continuous"], d["k"], True)
    assert abs(prep - 0.29623810011571716) < 1e-12 and not bad
    p2 = np.concatenate((np.ones(50), np.zeros(50))); k2 = np.concatenate((np.ones(45), np.zeros(55)))
    prep, bad = pre_filtering(p2, k2, False)
    assert abs(prep / 1.4919966396986922e-19 - 1) < 1e-9 and bad
    g = np.load(os.path.join(G, "lineage_N300_l10_j2.npz"))
    lin, cov, K, want = g["lin"].astype(float), g["cov"].astype(float), g["K"], g["max_lineage"]
    from test_oracle_golden import _same_or_tied
    idx = list(range(0, K.shape[0], 5))
    got = [fit_lineage_effect(lin, cov, K[v]) for v in idx]
    _same_or_tied(got, [None if want[v] < 0 else int(want[v]) for v in idx], lin, cov, K[idx])


@pytest.mark.parametrize("N,q,cont", [(400, 15, False), (600, 22, False), (500, 32, False), (400, 18, True)])
def test_wide_designs_vs_oracle(N, q, cont):
    """More than 14 covariate columns (the reference has no limit: pyseer/model.py:274-297): run-time-width kernels on scratch
    arrays (glm_wide.hip), same statistics, notes and filters as the oracle, Firth-routed variants included."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(31 + N + q)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.2 * W[:, 0] - 0.8 * W[:, 1]
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    V = 96
    af = np.concatenate([rng.uniform(0.05, 0.95, V - 12), rng.uniform(0.004, 0.02, 12)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    keep = (K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)
    K = K[keep]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, cont).llf
    nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    wants = orc.firth_noise_variants(lambda: orc.fixed_effects_batch(y, K.astype(float), W, cont, 1.0, 1.0, nl, nf))
    want = wants[0]
    e = Engine(N)
    e.glm_setup(y, W, cont, nl, nf)
    r = e.glm_batch(pack_variants(K))
    e.close()
    firth = (want["notes"] & 0x7C) != 0
    assert cont or firth.any()
    from _firth_tol import firth_rows_close
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f][~firth], want[f][~firth], atol=1e-12, what=f)
        # Firth rows: 1e-6 relative; the halving-noise slack only on rows where the tie detector fires (tests/_firth_tol.py)
        if f in ("prep", "pvalue"):
            close(r[f][firth], want[f][firth], rtol=1e-6, atol=1e-300, what=f + "(firth)")
        else:
            good, _ = firth_rows_close(r[f], wants, f, firth)
            assert good.all(), (f, np.where(firth)[0][~good][:4], r[f][firth][~good][:4], want[f][firth][~good][:4])
    close(r["betas"][~firth], want["betas"][~firth], atol=1e-12, what="betas")
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()


@pytest.mark.parametrize("N,q,cont", [(400, 17, False), (400, 17, True), (300, 6, False), (300, 6, True), (600, 24, False)])
def test_unscaled_covariates_vs_oracle(N, q, cont):
    """Covariates on very different scales and far from zero mean (age in years next to a 0/1 flag next to a 1e-2-sized principal
    component): the information matrix is badly scaled, so the pivoted LU of the run-time-width kernels really interchanges rows
    (a solve that applied the interchanges in the wrong order passed every test with unit-scale columns)."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(977 + N + q + cont)
    Z = rng.standard_normal((N, q))
    scale = 10.0 ** rng.uniform(-2.0, 1.7, q); shift = rng.uniform(-2, 2, q) * scale
    W = Z * scale + shift
    W[:, q - 1] = (rng.random(N) < 0.3).astype(float)                                  # a 0/1 covariate
    W[:, q - 2] = 2000.0 + np.round(8.0 * Z[:, q - 2])                                 # year of isolation: far from zero, small spread
    eta = -0.3 + 1.0 * Z[:, 0] - 0.7 * Z[:, 1]
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    V = 80
    af = np.concatenate([rng.uniform(0.05, 0.95, V - 8), rng.uniform(0.006, 0.02, 8)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, cont).llf
    nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    want = orc.fixed_effects_batch(y, K.astype(float), W, cont, 1.0, 1.0, nl, nf)
    e = Engine(N)
    e.glm_setup(y, W, cont, nl, nf)
    r = e.glm_batch(pack_variants(K))
    e.close()
    firth = (want["notes"] & 0x7C) != 0
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f][~firth], want[f][~firth], rtol=1e-6, atol=1e-12, what=f)
        # Firth rows: the un-centred year column puts cond(X^T W X) at ~1e11, so every solve with it carries ~1e-5 of rounding in
        # either implementation (numpy's pinv in the reference included); those rows agree to that level, not to 1e-6
        close(r[f][firth], want[f][firth], rtol=5e-5, atol=1e-6 if f != "pvalue" else 1e-300, what=f + "(firth)")
    close(r["betas"][~firth], want["betas"][~firth], rtol=1e-6, atol=1e-12, what="betas")
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()


@pytest.mark.parametrize("nl,j", [(20, 0), (30, 3), (16, 0)])
def test_wide_lineage_vs_oracle(nl, j):
    """fit_lineage_effect with more than 15 lineage/covariate columns (e.g. --lineage-clusters with many clusters)."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from test_oracle_golden import _same_or_tied
    rng = np.random.default_rng(nl + j)
    N, V = 900, 40
    cl = rng.integers(0, nl + 1, N)
    lin = np.zeros((N, nl)); lin[np.arange(N)[cl > 0], cl[cl > 0] - 1] = 1.0          # cluster indicators, one reference class
    cov = rng.standard_normal((N, j)) if j else None
    base = rng.uniform(0.3, 0.7, (V, nl + 1))
    K = (rng.random((V, N)) < base[:, cl]).astype(np.uint8)
    e = Engine(N)
    e.lineage_setup(lin, cov)
    got = e.lineage_batch(pack_variants(K))
    e.close()
    want = [orc.lineage_effect(lin, cov, K[v].astype(float)) for v in range(V)]
    # A cluster whose members all carry (or all lack) the k-mer is quasi-separated: its coefficient diverges, and whether the
    # weights underflow to an exactly singular Hessian (LinAlgError -> None) within the 35 iterations is decided by the rounding
    # of the linear solves (numpy inverts, these kernels solve).  Not a parity target; everything else must agree.
    sep = np.array([any(K[v][cl == c].sum() in (0, (cl == c).sum()) for c in range(nl + 1)) for v in range(V)])
    assert (~sep).sum() >= V // 2
    g = [None if x < 0 else int(x) for x in got]
    _same_or_tied([g[v] for v in range(V) if not sep[v]], [want[v] for v in range(V) if not sep[v]], lin, cov, K[~sep])


def test_firth_long_iterations_finish_in_the_workgroup_kernel():
    """(Quasi-)separated variants need far more than 16 Firth steps; after 16 the rounds hand a variant to k_firth_blk (one
    workgroup per variant), which must continue the same iteration: same decisions, same results as the oracle."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null, _cdf, _info, firth_likelihood
    rng = np.random.default_rng(90210)
    N, q = 700, 3
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = (rng.random(N) < 1 / (1 + np.exp(0.2 - W[:, 0]))).astype(float)
    K = np.stack([y, 1 - y, y * (rng.random(N) < 0.9), 1 - y * (rng.random(N) < 0.8), np.maximum(y, rng.random(N) < 0.05),
                  (rng.random(N) < 0.4).astype(float)]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)

    def firth_steps(k):                                        # accepted steps of fit_firth (numpy restatement of model.py:414-504)
        X = np.c_[np.ones(N), k, W]; b = np.zeros(q + 2); b[0] = np.log(y.mean() / (1 - y.mean())); prev = np.inf
        for it in range(1000):
            pi = _cdf(X.dot(b)); w = pi * (1 - pi); Vm = np.linalg.pinv(_info(X, b))
            h = w * np.einsum("ij,jk,ik->i", X, Vm, X)
            nb = b + Vm.dot(X.T.dot(y - pi + h * (0.5 - pi)))
            while firth_likelihood(nb, X, y) > firth_likelihood(b, X, y):
                nb = b + 0.5 * (nb - b)
            sn = np.linalg.norm(nb - b); b = nb
            if it > 0 and prev < 1e-4:
                return it + 1
            prev = sn
        return 1000
    steps = [firth_steps(k.astype(float)) for k in K]
    assert max(steps) > 40 and min(steps) < 16, steps          # both sides of the hand-off are exercised
    want = orc.firth_batch(y, K.astype(float), W)
    e = Engine(N)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    r = e.glm_batch(pack_variants(K))
    e.close()
    assert (want["status"] == 0).all() and not ((r["flags"] >> 6) & 1).any()
    for f in ("intercept", "kbeta", "bse"):
        close(r[f], want[f], rtol=2e-6, atol=1e-6, what=f)
    close(r["betas"], want["betas"], rtol=2e-6, atol=1e-6, what="betas")


@pytest.mark.parametrize("cont,dedup", [(False, False), (False, True), (True, False)])
def test_af_compaction_changes_nothing(cont, dedup, monkeypatch):
    """Rows outside the AF window all give the same output; when >= 3 % of a batch is such rows only the kept ones (plus one
    representative) go through the kernels.  Bit-identical outputs with the compaction forced, off, and adaptive; oracle on the rest."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(4242)
    N, q, V = 400, 3, 2600
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.2 + W[:, 0]
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    K = (rng.random((V, N)) < rng.uniform(0.05, 0.95, (V, 1))).astype(np.uint8)
    rare = rng.random(V) < 0.45
    K[rare] = (rng.random((int(rare.sum()), N)) < 0.003).astype(np.uint8)
    K[5] = K[6]                                                     # a duplicated pattern for the de-duplication inside
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, cont).llf; nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    res = []
    for mode in ("2", "0", "1"):
        monkeypatch.setenv("SEERHIP_ROUTE", "afcompact=" + mode)
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.set_dedup(dedup)
        e.glm_setup(y, W, cont, nl, nf)
        res.append(e.glm_batch(bits)); res.append(e.glm_batch(bits)); e.close()
    for other in res[1:]:
        for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas"):
            assert np.array_equal(res[0][f], other[f], equal_nan=True), f
        assert np.array_equal(res[0]["flags"], other["flags"])
    af = K.mean(axis=1); kept = (af >= 0.01) & (af <= 0.99)
    assert (~kept).sum() > 0.3 * V
    assert ((res[0]["flags"][~kept] & 0x101FF) == 0x10001).all() and np.isnan(res[0]["pvalue"][~kept]).all() and np.isnan(res[0]["prep"][~kept]).all()
    want = orc.fixed_effects_batch(y, K[kept].astype(float), W, cont, 1.0, 1.0, nl, nf)
    firth = (want["notes"] & 0x7C) != 0
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(res[0][f][kept][~firth], want[f][~firth], rtol=1e-6, atol=1e-12, what=f)
    assert ((res[0]["flags"][kept] & 0x1FF) == want["notes"]).all()


def test_routed_firth_at_large_n_goes_through_the_workgroup_kernel():
    """For N >= 2048 the Firth-routed variants of an ordinary run skip the rounds (firth_handoff = 0): k_firth_blk fits them from the
    first step.  Rare k-mers carried by cases only (bad-chisq, quasi-separated) and by controls only, against the oracle."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(2048)
    N, q, V = 2500, 4, 160
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = (rng.random(N) < 1 / (1 + np.exp(0.4 - W[:, 0]))).astype(float)
    cases, controls = np.flatnonzero(y == 1), np.flatnonzero(y == 0)
    K = np.zeros((V, N), dtype=np.uint8)
    for v in range(V):
        pool = cases if v % 2 == 0 else controls
        K[v, rng.choice(pool, int(rng.integers(26, 60)), replace=False)] = 1
        if v % 5 == 0:
            K[v, rng.choice(controls if v % 2 == 0 else cases, int(rng.integers(1, 4)), replace=False)] = 1   # not quite separated
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    want = orc.fixed_effects_batch(y, K.astype(float), W, False, 1.0, 1.0, nl, nf)
    routed = (want["notes"] & 0x3C) != 0
    assert routed.sum() > 0.7 * V
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf)
    r = e.glm_batch(pack_variants(K))
    info = e.glm_info(); e.close()
    assert info["firth_routed"] == int(routed.sum())
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(r[f], want[f], rtol=2e-6, atol=1e-6 if f != "pvalue" else 1e-300, what=f)
    close(r["betas"], want["betas"], rtol=2e-6, atol=1e-6, what="betas")


@pytest.mark.parametrize("dedup", [False, True])
def test_prefilter_compaction_changes_nothing(dedup, monkeypatch):
    """--filter-pvalue: rows whose 2x2 chi-square p-value exceeds the threshold are never fitted (model.py:266); their output is written
    by the classification kernel and only the rows to fit go through the regression kernels.  Bit-identical outputs with the compaction
    forced, off and adaptive; oracle on everything."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(777)
    N, q, V, pret = 500, 3, 3000, 0.05
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = (rng.random(N) < 1 / (1 + np.exp(0.2 - W[:, 0]))).astype(float)
    K = (rng.random((V, N)) < rng.uniform(0.03, 0.97, (V, 1))).astype(np.uint8)
    causal = rng.random(V) < 0.05
    K[causal] = ((rng.random((int(causal.sum()), N)) < 0.25) | ((y == 1) & (rng.random((int(causal.sum()), N)) < 0.3))).astype(np.uint8)
    K[::50] = (rng.random((len(K[::50]), N)) < 0.004).astype(np.uint8)             # some AF-filtered rows as well
    K[7] = K[8]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    res = []
    for mode in ("2", "0", "1"):
        monkeypatch.setenv("SEERHIP_ROUTE", "afcompact=" + mode)
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.set_dedup(dedup)
        e.glm_setup(y, W, False, nl, nf, pret, 1.0)
        res.append(e.glm_batch(bits)); res.append(e.glm_batch(bits)); e.close()
    for other in res[1:]:
        for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas"):
            assert np.array_equal(res[0][f], other[f], equal_nan=True), f
        assert np.array_equal(res[0]["flags"], other["flags"])
    af = K.mean(axis=1); in_af = (af >= 0.01) & (af <= 0.99)
    want = orc.fixed_effects_batch(y, K[in_af].astype(float), W, False, pret, 1.0, nl, nf)
    got = {f: res[0][f][in_af] for f in ("prep", "pvalue", "kbeta", "bse", "intercept")}
    assert want["prefilter"].mean() > 0.5 and (want["prefilter"] == 0).sum() > 50          # both kinds of rows are present
    assert ((res[0]["flags"][in_af] & 0x1FF) == want["notes"]).all()
    assert (((res[0]["flags"][in_af] >> 16) & 1) == want["prefilter"]).all()
    firth = (want["notes"] & 0x7C) != 0
    for f in got:
        close(got[f][~firth], want[f][~firth], rtol=1e-6, atol=1e-12, what=f)
        close(got[f][firth], want[f][firth], rtol=2e-6, atol=1e-6 if f != "pvalue" else 1e-300, what=f + "(firth)")


@pytest.mark.parametrize("N,q", [(1000, 10), (333, 3)])
def test_logistic_fit_paths_agree(N, q, monkeypatch):
    """The logistic fit has one product path (rounds of lean kernels: carrier-sum first step, packed fp32 Newton rounds, chord rounds,
    finishing kernels, workgroup restarts) and switches that route variants through its alternatives.  Every route must land on the
    same fixed point: the default against the oracle, the alternatives against the default to 2e-8 (the exact fp64 Newton step each
    takes last is certified <= 5e-7, i.e. quadratically below that); bse to 5e-7 (the correction-form information matrix carries the
    single-precision rounding of w - w0, ~1e-7 at these sample counts)."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(77 + N)
    V = 900
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.4 + 1.2 * W[:, 0] - 0.7 * W[:, 1]
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.03, 0.97, V - V // 6), rng.uniform(0.004, 0.03, V // 6)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[: V // 10] = (rng.random((V // 10, N)) < (0.15 + 0.6 * y)[None, :] * rng.uniform(0.3, 1.0, V // 10)[:, None]).astype(np.uint8)   # real effects
    K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    want = orc.fixed_effects_batch(y, K.astype(float), W, False, 1.0, 1.0, nl, nf)
    bits = pack_variants(K)

    def run(**route):                                            # the one test hook: csrc/route.h
        if route:
            monkeypatch.setenv("SEERHIP_ROUTE", ",".join("%s=%s" % kv for kv in route.items()))
        e = Engine(N); e.glm_setup(y, W, False, nl, nf); r = e.glm_batch(bits); e.close()
        if route:
            monkeypatch.delenv("SEERHIP_ROUTE")
        return r
    base = run()
    firth = (want["notes"] & 0x7C) != 0
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept"):
        close(base[f][~firth], want[f][~firth], what=f)
    assert ((base["flags"] & 0x1FF) == want["notes"]).all()
    routes = [dict(chord="0"),                                       # the three-kernel form (what a run without covariates takes)
              dict(chord_enter="5e-2"),                              # early hand-over: several chord rounds per variant
              dict(chord_n32="2"),                                   # stragglers of the Newton rounds restarted in fp64 (workgroup kernel)
              dict(bitdot="0"),                                      # first step by a pass, k_glm_final instead of the finishing kernels
              dict(bitdot="2"),                                      # carrier sums by round 3's nibble-table kernel (product: int8 matrix cores)
              dict(fin_rounds="0"), dict(pk="0"), dict(warm="0"), dict(newton="1"),
              dict(ll_first="0"),                                    # score pass first, likelihood pass last (one more fp64 pass per variant)
              dict(ll_first="0", chord_enter="5e-2"),
              dict(firth_last="0"),                                  # the routed variants' Firth fits take their last likelihood pass
              dict(first_bordered="0"),                              # the first Newton step through the general 12 x 12 kernel (product: bordered solve)
              dict(first_bordered="0", chord_enter="5e-2")]
    for env in routes:
        r = run(**env)
        assert (r["flags"] == base["flags"]).all(), env
        for f in ("pvalue", "kbeta", "bse", "intercept"):
            close(r[f][~firth], base[f][~firth], rtol=5e-7 if f == "bse" else 2e-8, atol=1e-12, what="%s under %s" % (f, env))
        close(r["betas"][~firth], base["betas"][~firth], rtol=2e-8, atol=1e-10, what="betas under %s" % env)


@pytest.mark.parametrize("N,q,V,seed", [(1000, 10, 30000, 1), (5000, 10, 6000, 2)])
def test_strong_effects_and_awkward_covariates_sweep(N, q, V, seed):
    """A sweep the round kernels were tuned against after it found them out (tools/gpu_glm_sweep.py): a binary covariate, an un-centred one
    (2000 +- 10), a U-shaped allele-frequency spectrum and 15 % of the variants built FROM the phenotype (|beta| up to 5, some
    near-separating).  The correction-form information matrix sums w - w0 in single precision; for such variants those differences are
    large, which is why the intercept / variant rows are summed per 64-sample word and in fp64 across words (1.7e-6 on bse before)."""
    from oracle import oracle as orc
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((N, q)); W[:, 0] = rng.random(N) < 0.3; W[:, 1] = 2000 + 10 * W[:, 1]
    eta = -0.5 + 0.9 * W[:, 0] + 0.5 * W[:, 2]
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V // 2), rng.beta(0.3, 0.3, V - V // 2)])
    K = (rng.random((V, N)) < af[:, None])
    eff = rng.random(V) < 0.15
    K[eff] = rng.random((int(eff.sum()), N)) < (0.05 + 0.8 * y)[None, :] * rng.uniform(0.1, 1.0, int(eff.sum()))[:, None]
    K = K.astype(np.uint8)
    K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)]
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    want = orc.fixed_effects_batch(y, K.astype(float), W, False, 1.0, 1.0, nl, nf)
    e = Engine(N); e.glm_setup(y, W, False, nl, nf); r = e.glm_batch(pack_variants(K)); e.close()
    firth = (want["notes"] & 0x7C) != 0                     # Firth rows have their own tests (tie semantics)
    assert ((r["flags"] & 0x1FF)[~firth] == want["notes"][~firth]).all()
    assert (np.abs(want["kbeta"][~firth]) > 3).sum() > 20    # the strong effects are really there
    close(r["prep"][~firth], want["prep"][~firth], rtol=1e-12, what="prep")
    close(r["kbeta"][~firth], want["kbeta"][~firth], rtol=1e-8, atol=1e-12, what="kbeta")
    close(r["intercept"][~firth], want["intercept"][~firth], rtol=1e-7, atol=1e-12, what="intercept")
    close(r["bse"][~firth], want["bse"][~firth], rtol=6e-7, what="bse")
    close(r["pvalue"][~firth], want["pvalue"][~firth], rtol=1e-6, atol=1e-300, what="pvalue")


@pytest.mark.parametrize("N,q,V", [(5000, 10, 8192), (4100, 1, 1024), (4099, 3, 777), (6007, 7, 1500), (4096, 10, 640), (8200, 5, 512)])
def test_one_pass_firth_equals_the_two_pass_rounds(N, q, V, monkeypatch):
    """Forced Firth at N >= 4096 (BASELINE config C4's mode): the one-pass iteration (firth_fast.hip: third-moment tensor and information matrix
    on the matrix cores, fits finished in the kernel) against the exact two-pass rounds (SEERHIP_ROUTE firth_fast=0).  Same flags; statistics to
    1e-6 relative (kbeta: or 2e-8 absolute -- the penalty's share of the fixed point carries the matrix-core sums' 1e-6); sample counts that
    are not multiples of 16 or 64, every supported design width, majority-carrier rows (taken by their complement), rare rows, rows no
    iteration can fit (all carriers share the phenotype: those leave the fast passes for the exact kernels)."""
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(97 + N + q)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.2 * W[:, 0] - (0.7 * W[:, 1] if q > 1 else 0.0)
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 4), rng.uniform(0.0101, 0.03, V // 8), rng.uniform(0.97, 0.9899, V // 4 - V // 8)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[: V // 10] = (rng.random((V // 10, N)) < (0.15 + 0.5 * y)[None, :]).astype(np.uint8)              # real effects
    ones = np.where(y == 1)[0]
    for r in range(V // 10, V // 10 + 8):                                                              # carriers only among the phenotype-1 samples
        K[r] = 0; K[r, rng.choice(ones, 60 + 10 * (r % 5), replace=False)] = 1
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    out = {}
    # two: the exact rounds;  one: the one-pass kernels, their single-precision first pass at two wavefronts per SIMD (firth_fast_w.hip: 16-row
    # tiles, 16 variants per wavefront, sums handed to a per-fit kernel);  one_w0: that pass at one wavefront per SIMD (firth_fast.hip, round 4)
    # one_f64: every pass of the one-pass kernel in fp64, the first included (route key firth_first32)
    # one_r5: round 5's sequence -- ONE single-precision pass, then two fp64 passes (since round 6 the default is two single-precision passes at
    # two wavefronts per SIMD and ONE fp64 pass, which finishes the fit: its last step is ~1e-8)
    for mode, route in (("two", "firth_fast=0"), ("one", None), ("one_w0", "firth_w=0"), ("one_f64", "firth_first32=0"), ("one_r5", "firth_first32=1")):
        if route is None:
            monkeypatch.delenv("SEERHIP_ROUTE", raising=False)
        else:
            monkeypatch.setenv("SEERHIP_ROUTE", route)
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.glm_setup(y, W, False, nl, nf, force_firth=True); out[mode] = e.glm_batch(bits); e.close()
    monkeypatch.delenv("SEERHIP_ROUTE", raising=False)
    a = out["two"]
    ok = np.isfinite(a["kbeta"])
    assert ok.sum() > V // 2
    for mode in ("one", "one_w0", "one_f64", "one_r5"):
        b = out[mode]
        # (bit 18, SH_FLAG_FIRTH_SENSITIVE, is informational: "the stop rule was met within 1e-8 of its limit" may fall either way between two
        # arithmetic orders -- at most a couple of rows; every other bit is held exactly)
        assert int(((a["flags"] ^ b["flags"]) >> 18 & 1).sum()) <= 2, mode
        df_ = np.where((a["flags"] & ~np.uint32(1 << 18)) != (b["flags"] & ~np.uint32(1 << 18)))[0]
        assert df_.size == 0, (mode, df_[:10], [hex(int(x)) for x in a["flags"][df_[:10]]], [hex(int(x)) for x in b["flags"][df_[:10]]], a["kbeta"][df_[:10]], b["kbeta"][df_[:10]])
        assert np.array_equal(ok, np.isfinite(b["kbeta"])), mode
        dev = {f: float(np.max(np.abs(b[f][ok] - a[f][ok]) / np.maximum(np.abs(a[f][ok]), 1e-300))) for f in ("pvalue", "bse")}
        # kbeta in units of the test's own tolerance (1e-6 relative or 2e-8 absolute): |d| / (1e-6 |kbeta| + 2e-8)
        dev["kbeta_tol_units"] = float(np.max(np.abs(b["kbeta"][ok] - a["kbeta"][ok]) / (1e-6 * np.abs(a["kbeta"][ok]) + 2e-8)))
        print("one-pass Firth (%s) against the exact rounds, N=%d q=%d, %d fits: max rel pvalue %.2e bse %.2e, kbeta %.2f of its tolerance"
              % (mode, N, q, int(ok.sum()), dev["pvalue"], dev["bse"], dev["kbeta_tol_units"]))
        close(b["pvalue"][ok], a["pvalue"][ok], rtol=1e-6, atol=1e-300, what="pvalue")
        close(b["bse"][ok], a["bse"][ok], rtol=1e-6, what="bse")
        close(b["kbeta"][ok], a["kbeta"][ok], rtol=1e-6, atol=2e-8, what="kbeta")
        close(b["intercept"][ok], a["intercept"][ok], rtol=1e-6, atol=2e-8, what="intercept")
        close(b["betas"][ok], a["betas"][ok], rtol=1e-6, atol=2e-8, what="betas")
        # measured on the C4 workload (profiles/r04/firth_fast_check_q10.json, 259 560 fits): kbeta 4.9e-9 absolute, bse 2.3e-8, p 1.0e-8;
        # the ceilings here leave a factor ~10 on this test's harsher rows (real effects, rare and majority carriers)
        # round 6 (two single-precision passes, then ONE fp64 pass): a rare variant (55 .. 70 carriers: V_11 ~ 0.08) keeps up to 1e-8 of the second
        # single-precision pass' arithmetic (each carrier's mu - y rounded to float), shrunk once by the iteration's rate instead of twice:
        # 0.33 of the tolerance on this test's rows, 0.41 on the worst of 259 560 C4 fits (tools/gpu_firth_n32_diag.py; round 5's sequence: 0.31)
        assert dev["kbeta_tol_units"] <= (0.5 if mode == "one" else 0.25) and dev["bse"] <= 3e-7 and dev["pvalue"] <= 1.5e-7, (mode, dev)


def test_contexts_on_one_device_run_concurrently_and_agree_bit_for_bit():
    """Several engine contexts of ONE device driven from their own host threads at once (the job path of `--gpus 0,0,0`; bench.py's
    three_contexts line; the counterpart of the reference's --cpu N, pyseer/__main__.py:541-568): every context returns the bytes the
    single context returns for the same rows -- ordinary fits, Firth-routed rows, AF-filtered rows."""
    import threading
    import torch
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    N, q, V = 700, 5, 40000
    rng = np.random.default_rng(31)
    W = rng.standard_normal((N, q))
    y = (rng.random(N) < 1 / (1 + np.exp(-(-0.3 + 0.9 * W[:, 0])))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 5), rng.uniform(0.0, 0.02, V // 5)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[:300] = (rng.random((300, N)) < (0.1 + 0.8 * y)[None, :]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    rows = [torch.from_numpy(pack_variants(K[s::3])).cuda() for s in range(3)]          # three different batches
    dev = torch.device("cuda:0")

    def make():
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
            e.glm_setup(y, W, False, nl, nf, 1.0, 1.0)
        return st, e
    st0, e0_ = make()
    want = []
    with torch.cuda.stream(st0):
        for r in rows:
            o, f = e0_.glm_batch_dev(r); st0.synchronize()
            want.append((o.cpu().numpy().copy(), f.cpu().numpy().copy()))
    ctxs = [(st0, e0_)] + [make() for _ in range(2)]
    torch.cuda.synchronize()
    got = [[None] * 6 for _ in range(3)]

    def work(c):
        st, e = ctxs[c]
        with torch.cuda.stream(st):
            for it in range(6):                                       # every context sees every batch, in another order than its neighbours
                k = (it + c) % 3
                o, f = e.glm_batch_dev(rows[k]); st.synchronize()
                got[c][it] = (k, o.cpu().numpy().copy(), f.cpu().numpy().copy())
    th = [threading.Thread(target=work, args=(c,)) for c in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    for c in range(3):
        for k, o, f in got[c]:
            assert np.array_equal(o.view(np.uint8), want[k][0].view(np.uint8)) and np.array_equal(f, want[k][1]), (c, k)
    assert (np.bitwise_and(want[0][1], 0x7C) != 0).sum() > 10         # Firth-routed rows were among them
    for _, e in ctxs:
        e.close()


def test_the_lanes_of_a_context_return_the_synchronous_rows(monkeypatch):
    """sh_glm_batch_dev_async (csrc/lanes_api.inc): batches handed to the worker threads of ONE context -- each with its own stream and
    workspaces, set up from the arguments and the route string of the context's own sh_glm_setup -- return the bytes of the synchronous call
    for the same rows, with 1, 2 and 4 lanes (sh_set_lanes; SEERHIP_ROUTE lanes=n sets the default), more batches in flight than lanes, and
    after the model of the context has been replaced.  The counterpart of the reference's pool of --cpu N workers over blocks of
    variants (pyseer/__main__.py:541-568)."""
    import torch
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    N, q, V = 700, 5, 30000
    rng = np.random.default_rng(32)
    W = rng.standard_normal((N, q))
    y = (rng.random(N) < 1 / (1 + np.exp(-(-0.3 + 0.9 * W[:, 0])))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 5), rng.uniform(0.0, 0.02, V // 5)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[:300] = (rng.random((300, N)) < (0.1 + 0.8 * y)[None, :]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    rows = [torch.from_numpy(pack_variants(K[s::5])).cuda() for s in range(5)]          # five different batches
    monkeypatch.setenv("SEERHIP_ROUTE", "lanes=2")
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    monkeypatch.delenv("SEERHIP_ROUTE")
    assert e.get_lanes() == 2

    def both(y_, W_, force):
        nl = fit_null(y_, W_, e0, False).llf; nf = fit_null(y_, W_, e0, False, firth=True)
        e.glm_setup(y_, W_, False, nl, nf, 1.0, 1.0, force_firth=force)
        want = []
        for r in rows:
            o, f = e.glm_batch_dev(r); torch.cuda.synchronize()
            want.append((o.cpu().numpy().copy(), f.cpu().numpy().copy()))
        for lanes in (None, 1, 4):
            if lanes is not None:
                e.set_lanes(lanes)
            outs = [torch.zeros((5 + W_.shape[1], r.shape[0]), dtype=torch.float64, device="cuda") for r in rows * 2]
            fls = [torch.zeros((r.shape[0],), dtype=torch.int32, device="cuda") for r in rows * 2]
            for i in range(10):                                        # ten batches in flight at most: more than 2 x lanes
                e.glm_batch_dev_async(rows[i % 5], outs[i], fls[i])
            e.wait(); torch.cuda.synchronize()
            for i in range(10):
                assert np.array_equal(outs[i].cpu().numpy().view(np.uint8), want[i % 5][0].view(np.uint8)), (lanes, i)
                assert np.array_equal(fls[i].cpu().numpy(), want[i % 5][1]), (lanes, i)
        return want
    w1 = both(y, W, False)
    assert (np.bitwise_and(w1[0][1], 0x7C) != 0).sum() > 10           # Firth-routed rows were among them
    # another model on the same context (the lanes are torn down and set up again from it), every variant through Firth
    y2 = (rng.random(N) < 0.35).astype(float)
    w2 = both(y2, W[:, :3].copy(), True)
    assert not np.array_equal(w1[0][0][2], w2[0][0][2])
    # a shape error is reported by the call itself
    import pytest as _pt
    from pyseer_amd._abi import SeerHipError
    with _pt.raises(SeerHipError):
        e.glm_batch_dev_async(rows[0][:, :40].contiguous(), torch.zeros((8, rows[0].shape[0]), dtype=torch.float64, device="cuda"),
                              torch.zeros((rows[0].shape[0],), dtype=torch.int32, device="cuda"))
    e.close()


def test_firth_sensitive_bit_marks_the_reference_firth_fails_and_little_else():
    """SH_FLAG_FIRTH_SENSITIVE (include/seerhip.h, flags bit 18): set on a Firth-fitted row whose reference answer is fragile -- a variant that
    all but separates the phenotype (a cell of its 2 x 2 table <= 1), >= 12 accepted steps, a stop rule met within 1e-8 of its limit, or a
    firth-fail of this library.  Every row on which the reference itself returned `firth-fail` (glm_exit_firthfail_*.npz; it fits each of
    them with the samples in another order, model.py:465-484) carries it, in the routed run and in the forced-Firth run; so does the row
    whose stop rule sits at 1.00003e-4 (glm_exit_highbse_N100_q1.npz row 5); on a forced-Firth run over random k-mers fewer than 1e-3 of the
    rows do, and no row that was not Firth-fitted ever does."""
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    from pyseer_amd.classes import FLAG_FIRTH_SENSITIVE
    nref = 0
    for path in sorted(glob.glob(os.path.join(G, "glm_exit_firthfail_*.npz"))):
        d = np.load(path)
        ff = (d["notes"] & 0x40) != 0
        for force in (False, True):
            fl = _run(d, force_firth=force)["flags"] if force else _run(d)["flags"]
            sens = (fl & FLAG_FIRTH_SENSITIVE) != 0
            print("%s force_firth=%d: reference firth-fail rows %d, of them marked %d; controls marked %d of %d"
                  % (os.path.basename(path), force, int(ff.sum()), int((sens & ff).sum()), int((sens & ~ff).sum()), int((~ff).sum())))
            assert sens[ff].all(), (path, force, np.where(ff & ~sens)[0])
        nref += int(ff.sum())
    assert nref == 13                                           # 9 rows at N <= 1000 and 4 at N = 5000
    d = np.load(os.path.join(G, "glm_exit_highbse_N100_q1.npz"))
    assert _run(d)["flags"][5] & FLAG_FIRTH_SENSITIVE          # 52 steps, the 52nd of norm 1.00003e-4 (exception 2)
    for path in sorted(glob.glob(os.path.join(G, "glm_N*.npz"))):   # ordinary runs: only rows the reference sent through fit_firth carry the bit
        d = np.load(path); fl = _run(d)["flags"]
        assert not ((fl & FLAG_FIRTH_SENSITIVE) != 0)[(d["notes"] & 0x7C) == 0].any(), path
    # random k-mers, every variant through Firth (BASELINE config C4's kind of row)
    N, q, V = 5000, 10, 32768
    rng = np.random.default_rng(515)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = (rng.random(N) < 1 / (1 + np.exp(-(-0.2 + 0.8 * W[:, 0])))).astype(float)
    af = rng.uniform(0.01, 0.99, V)
    bits = pack_variants((rng.random((V, N)) < af[:, None]).astype(np.uint8))
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    e = Engine(N); e.set_af_filter(0.01, 0.99); e.glm_setup(y, W, False, nl, nf, force_firth=True)
    fl = e.glm_batch(bits)["flags"]; e.close()
    tested = (fl & 1) == 0
    frac = float(((fl & FLAG_FIRTH_SENSITIVE) != 0)[tested].mean())
    print("forced Firth on %d random k-mers: %.2e of the rows marked order-sensitive" % (int(tested.sum()), frac))
    assert frac < 1e-3
