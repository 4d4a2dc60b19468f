"""Pin the CPU oracle at the BASELINE size against the reference ITSELF: tests/golden/n5000_{glm,firth,lmm}.npz were produced by
importing pyseer (statsmodels 0.12.2) in the build container on the bench's own N = 5000 designs (tests/golden/make_n5000_golden.py).
Until round 4 the largest reference-generated fixture was N = 1000 (GLM) / N = 300 (LMM)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc
from _n5000 import G, unpack, lmm_design, FIRTH_BITS, noise_row_mask

N = 5000


def close(a, b, rtol, atol=0.0, what=""):
    a = np.atleast_1d(np.asarray(a, dtype=float)); b = np.atleast_1d(np.asarray(b, dtype=float))
    assert a.shape == b.shape, (a.shape, b.shape)
    with np.errstate(invalid="ignore"):
        ok = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), "%s mismatch at %s: got %s want %s" % (what, np.where(~ok)[0][:5], a[~ok][:5], b[~ok][:5])


def test_oracle_fixed_effects_n5000():
    """64 rows of fixed_effects_regression (model.py:202) at N = 5000, q = 10: ordinary Newton rows, 8 bad-chisq rows the reference sent
    through fit_firth, one af-filtered row."""
    d = np.load(os.path.join(G, "n5000_glm.npz"))
    rows = unpack(d["bits"], N); y = d["y"].astype(float); W = d["W"]
    af = rows.mean(axis=1); keep = (af >= 0.01) & (af <= 0.99)
    r = orc.fixed_effects_batch(y, rows[keep], W, False, 1.0, 1.0, float(d["null_llf"]), float(d["null_firth"]))
    main = d["main"][keep]; notes = d["notes"][keep]
    assert (r["notes"] == notes).all() and (r["prefilter"] == d["prefilter"][keep]).all() and (r["filter"] == d["filter"][keep]).all()
    assert (d["notes"][~keep] == 1).all()
    assert ((notes & FIRTH_BITS) != 0).sum() >= 8
    for j, f in enumerate(["prep", "pvalue", "kbeta", "bse", "intercept"]):
        close(r[f], main[:, j], rtol=1e-6 if f == "pvalue" else 1e-7, atol=1e-300, what=f)      # no Firth slack needed: see test below
    close(r["betas"], d["betas"][keep], rtol=1e-7, atol=1e-12, what="betas")


def test_oracle_fit_firth_n5000_disputed_rows():
    """fit_firth (model.py:414) on the 82 rows of the C4 workload where round 3's three step-halving modes and the C restatement
    disagreed, and 24 controls.  The reference fits every one of them.  With the log-likelihood summed in numpy's order (oracle/
    seer_oracle.c np_pairwise_sum) the restatement does too and agrees to 1e-6 RELATIVE with no absolute slack."""
    d = np.load(os.path.join(G, "n5000_firth.npz"))
    rows = unpack(d["bits"], N); y = d["y"].astype(float); W = d["W"]
    assert (d["firth_ok"] == 1).all() and int(d["disputed"].sum()) == 82
    w = orc.firth_batch(y, rows, W)
    assert (w["status"] == 0).all(), np.where(w["status"] != 0)[0]
    fm = d["firth_main"]
    close(w["intercept"], fm[:, 0], rtol=1e-6, what="intercept"); close(w["kbeta"], fm[:, 1], rtol=1e-6, what="kbeta")
    close(w["bse"], fm[:, 2], rtol=1e-7, what="bse"); close(w["fitll"], fm[:, 3], rtol=1e-12, what="fitll")
    close(w["betas"], d["firth_betas"], rtol=1e-6, atol=1e-9, what="betas")


@pytest.fixture(scope="module")
def lmm5k():
    d = np.load(os.path.join(G, "n5000_lmm.npz"))
    K, y = lmm_design(int(d["seed"]))
    assert np.trace(K) == float(d["K_trace"]) and K.sum() == float(d["K_sum"]) and np.array_equal(y, d["y"].astype(float))
    from pyseer_amd.lmm import initialise_lmm_arrays
    U, S, h2, nll, C = initialise_lmm_arrays(K, y, None, use_gpu=False)       # host set-up of the product (numpy eigh), ~15 s
    return d, U, S, h2, nll, C, y


def test_lmm_setup_n5000(lmm5k):
    """initialise_lmm (lmm.py:26) -> findH2: h2 is the end point of a Brent search on a flat objective, reproducible to its tolerance
    (1.5e-8), not to 1e-12; nLL at the optimum to 1e-10; the largest eigenvalues of the projected kernel to 1e-9."""
    d, U, S, h2, nll, C, y = lmm5k
    assert abs(h2 - float(d["h2"])) < 1e-6
    close(nll, float(d["nLL"]), rtol=1e-10)
    close(np.sort(S)[-8:], d["S_top"], rtol=1e-9)


def test_oracle_lmm_block_n5000(lmm5k):
    """fit_lmm_block (lmm.py:228) on 64 rows at the reference's h2 and at h2 = 0; fit_lmm (lmm.py:125) with two threshold sets."""
    d, U, S, h2, nll, C, y = lmm5k
    rows = unpack(d["bits"], N)
    L = orc.LmmOracle(U, S, y, C)
    noise = noise_row_mask(rows, y)
    for hh, key in ((float(d["h2"]), "blk"), (0.0, "blk_h2_0")):
        b, s, f, p = L.block(hh, rows)
        want = d[key]
        close(b[~noise], want[~noise, 0], rtol=1e-8, atol=1e-14, what="beta"); close(s[~noise], want[~noise, 1], rtol=1e-8, what="bse")
        close(f[~noise], want[~noise, 2], rtol=1e-8, atol=1e-12, what="frac_h2"); close(p[~noise], want[~noise, 3], rtol=1e-6, atol=1e-300, what="p")
    af = rows.mean(axis=1); afmask = ~((af >= 0.01) & (af <= 0.99))
    for o, (fp, lp) in (("o1", (1.0, 1.0)), ("o2", (0.05, 0.01))):
        r = L.fit_lmm(float(d["h2"]), rows, afmask, False, fp, lp)
        order = d[o + "_order"]
        got = np.stack([r["prep"], r["pvalue"], r["kbeta"], r["bse"], r["frac_h2"]], axis=1)[order]
        nz = noise[order]
        close(got[~nz], d[o + "_rows"][~nz], rtol=1e-6, atol=1e-12, what=o)
        assert (r["notes"][order][~nz] == d[o + "_notes"][~nz]).all()
        assert (r["prefilter"][order] == d[o + "_prefilter"]).all() and (r["filter"][order][~nz] == d[o + "_filter"][~nz]).all()


# ---- round 5: rows nobody picked, and the routes that were reference-pinned at N <= 300 only (tests/golden/make_n5000_golden.py) --------------
def test_oracle_random_bench_rows_n5000():
    """The FIRST 256 rows of the bench's fixed-effects batch (synth_bits seed 4242, dumped on the GPU box by tools/gpu_dump_c4_rows.py) through the
    reference's fit_firth (model.py:414, C4's mode) and fixed_effects_regression (model.py:202, C2N5000's)."""
    d = np.load(os.path.join(G, "n5000_random.npz"))
    rows = unpack(d["bits"], N); y = d["y"].astype(float); W = d["W"]
    ok = d["firth_ok"] == 1
    assert ok.sum() >= 250 and (d["firth_ok"][~ok] == 2).all()          # the rest are outside the AF window, nothing failed
    w = orc.firth_batch(y, rows[ok], W)
    fm = d["firth_main"][ok]
    # a restatement `firth-fail` on a row the reference fits is the rounding-noise tie of DESIGN.md section 6; counted, bounded, not compared
    fit = w["status"] == 0
    assert (~fit).sum() <= 2, np.where(~fit)[0]
    close(w["intercept"][fit], fm[fit, 0], rtol=1e-6, what="intercept"); close(w["kbeta"][fit], fm[fit, 1], rtol=1e-6, what="kbeta")
    close(w["bse"][fit], fm[fit, 2], rtol=1e-7, what="bse"); close(w["fitll"][fit], fm[fit, 3], rtol=1e-12, what="fitll")
    keep = d["notes"] != 1
    r = orc.fixed_effects_batch(y, rows[keep], W, False, 1.0, 1.0, float(d["null_llf"]), float(d["null_firth"]))
    assert (r["notes"] == d["notes"][keep]).all()
    for j, f in enumerate(["prep", "pvalue", "kbeta", "bse", "intercept"]):
        close(r[f], d["main"][keep, j], rtol=1e-6 if f == "pvalue" else 1e-7, atol=1e-300, what=f)
    close(r["betas"], d["betas"][keep], rtol=1e-7, atol=1e-12, what="betas")


@pytest.mark.parametrize("thr", ["", "_thr"])
def test_oracle_ols_n5000(thr):
    """model.py:299-312 (continuous phenotype) on 64 rows at N = 5000, q = 10, without and with (--filter-pvalue 0.05, --lrt-pvalue 0.01)."""
    d = np.load(os.path.join(G, "n5000_ols.npz"))
    rows = unpack(d["bits"], N); y = d["y"]; W = d["W"]
    af = rows.mean(axis=1); keep = (af >= 0.01) & (af <= 0.99)
    pret, lrtt = (float(d["pret_thr"]), float(d["lrtt_thr"])) if thr else (1.0, 1.0)
    r = orc.fixed_effects_batch(y, rows[keep], W, True, pret, lrtt, np.nan, np.nan)
    assert (r["notes"] == d["notes" + thr][keep]).all() and (r["prefilter"] == d["prefilter" + thr][keep]).all() and (r["filter"] == d["filter" + thr][keep]).all()
    for j, f in enumerate(["prep", "pvalue", "kbeta", "bse", "intercept"]):
        close(r[f], d["main" + thr][keep, j], rtol=1e-6 if f in ("pvalue", "prep") else 1e-8, atol=1e-300, what=f)
    tested = np.isfinite(d["main" + thr][keep, 2])
    close(r["betas"][tested], d["betas" + thr][keep][tested], rtol=1e-8, atol=1e-13, what="betas")


@pytest.mark.parametrize("tag", ["mds", "mds_nocov", "clusters"])
def test_oracle_lineage_n5000(tag):
    """model.py:151-199 fit_lineage_effect at N = 5000: 10 MDS-like columns (+ 2 covariates), 11 cluster indicators."""
    from test_oracle_golden import _same_or_tied
    d = np.load(os.path.join(G, "n5000_lineage.npz"))
    rows = unpack(d["bits"], N)
    lin = d["clusters"] if tag == "clusters" else d["W"]
    cov = d["cov"] if tag == "mds" else None
    got = [orc.lineage_effect(lin, cov, rows[v]) for v in range(rows.shape[0])]
    want = [None if x < 0 else int(x) for x in d["max_lineage_" + tag]]
    _same_or_tied(got, want, lin, cov, rows.astype(np.uint8))
    assert sum(x is None for x in want) >= (1 if tag == "clusters" else 0)


def test_oracle_newton_at_the_35_iteration_cap_n5000():
    """model.py:316-330 where statsmodels' Newton stops at maxiter = 35 without converging (SM:base/optimizer.py:407-427; a covariate that
    quasi-separates the phenotype): the reference takes what it has -- 8 rows -- and the rows whose Hessian becomes singular on the way go
    through fit_firth (matrix-inversion-error), one of them to firth-fail."""
    d = np.load(os.path.join(G, "n5000_cap35.npz"))
    rows = unpack(d["bits"], N); y = d["y"].astype(float); W = d["W"]
    cap = (d["newton_iterations"] == 35) & (d["newton_converged"] == 0)
    assert cap.sum() >= 8
    r = orc.fixed_effects_batch(y, rows, W, False, 1.0, 1.0, float(d["null_llf"]), float(d["null_firth"]))
    assert (r["notes"][cap] == d["notes"][cap]).all() and (d["notes"][cap] == 0).all()
    for j, f in enumerate(["prep", "pvalue", "kbeta", "bse", "intercept"]):
        close(r[f][cap], d["main"][cap, j], rtol=1e-6, atol=1e-300, what=f)
    # column 0 is the quasi-separating covariate: its slope is what diverges (8.7e3 after 35 steps through nearly singular Hessians) and is
    # reproducible to 1e-5 only; everything the reference prints by default (the variant's own statistics) holds at 1e-6
    close(r["betas"][cap][:, 1:], d["betas"][cap][:, 1:], rtol=1e-6, atol=1e-12, what="betas")
    close(r["betas"][cap][:, 0], d["betas"][cap][:, 0], rtol=1e-5, what="the diverging slope")
    other = ~cap
    assert ((r["notes"][other] & 0x20) == (d["notes"][other] & 0x20)).all()         # the rows whose Hessian went singular: matrix-inversion-error on both sides


@pytest.fixture(scope="module")
def lmm5k_d3():
    d = np.load(os.path.join(G, "n5000_lmm_d3.npz"))
    K, y = lmm_design(int(d["seed"]))
    assert np.trace(K) == float(d["K_trace"]) and K.sum() == float(d["K_sum"]) and np.array_equal(y, d["y"].astype(float))
    from pyseer_amd.lmm import initialise_lmm_arrays
    U3, S3, h23, nll3, C3 = initialise_lmm_arrays(K, y, d["cov"], use_gpu=False)
    U1, S1, h21, nll1, C1 = initialise_lmm_arrays(K, y, None, use_gpu=False)
    return d, (U3, S3, h23, nll3, C3), (U1, S1, h21), y


def test_oracle_lmm_three_covariates_and_the_mismatched_cache_n5000(lmm5k_d3):
    """lmm.py:26-122 with two covariates + intercept (D = 3): findH2 on the projected kernel; fit_lmm_block (lmm.py:228) on 64 rows; and the
    reference's `--load-lmm` of a D = 1 cache under D = 3 covariates (lmm.py:57-76: U, S, h2 of the cache, X of the run; run_test.sh:47)."""
    d, (U3, S3, h23, nll3, C3), (U1, S1, h21), y = lmm5k_d3
    assert abs(h23 - float(d["h2_D3"])) < 1e-6 and abs(h21 - float(d["h2_D1"])) < 1e-6
    close(nll3, float(d["nLL_D3"]), rtol=1e-10); close(np.sort(S3)[-8:], d["S_top_D3"], rtol=1e-9); close(np.sort(S1)[-8:], d["S_top_D1"], rtol=1e-9)
    rows = unpack(d["bits"], N)
    for (U, S, hh, key) in ((U3, S3, float(d["h2_D3"]), "blk_D3"), (U1, S1, float(d["h2_D1"]), "blk_mismatch")):
        L = orc.LmmOracle(U, S, y, C3)
        b, s, f, p = L.block(hh, rows)
        want = d[key]
        ok = np.isfinite(want[:, 1]) & (want[:, 1] > 0) & np.isfinite(want[:, 0])
        close(b[ok], want[ok, 0], rtol=1e-7, atol=1e-13, what=key + " beta"); close(s[ok], want[ok, 1], rtol=1e-7, what=key + " bse")
        close(f[ok], want[ok, 2], rtol=1e-7, atol=1e-11, what=key + " frac_h2"); close(p[ok], want[ok, 3], rtol=1e-6, atol=1e-300, what=key + " p")
