#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors by IMPORTING the reference (pyseer @ /root/reference).

Runs ONLY in the build container (the reference and its statsmodels/scipy stack do
not exist on the GPU box).  Usage:

    cd /root/repo/tests/golden && \
    PYTHONPATH=_harness:_harness/stubs:/root/reference PYTHONDONTWRITEBYTECODE=1 \
    OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 \
    /opt/conda/bin/python3.9 -W ignore make_golden.py

Outputs (committed): *.npz / *.json in this directory.  Only DATA is written:
inputs we synthesise (or the reference's own test data files) and the numbers the
reference returns for them.  No reference source is copied.

Reference entry points exercised (file:line in /root/reference):
  pyseer/model.py:31   pre_filtering
  pyseer/model.py:73   fit_null
  pyseer/model.py:202  fixed_effects_regression
  pyseer/model.py:397  firth_likelihood
  pyseer/model.py:414  fit_firth
  pyseer/lmm.py:26     initialise_lmm
  pyseer/lmm.py:125    fit_lmm
  pyseer/lmm.py:228    fit_lmm_block
  pyseer/fastlmm/lmm_cov.py:427 findH2 ; :597 nLLeval
  pyseer/utils.py:39   format_output
  pyseer/input.py:710  hash_pattern
"""
import shim  # noqa: F401  must be first
import os
import sys
import json
import warnings
import numpy as np
import pandas as pd

warnings.simplefilter("ignore")

import statsmodels.formula.api as smf
from pyseer.model import (pre_filtering, fit_null, fit_firth, firth_likelihood,
                          fixed_effects_regression, fit_lineage_effect)
from pyseer.lmm import initialise_lmm, fit_lmm, fit_lmm_block
from pyseer.fastlmm.lmm_cov import LMM as lmm_cov
from pyseer.classes import Seer, LMM
from pyseer.utils import format_output
from pyseer.input import hash_pattern

REF_TESTS = "/root/reference/tests"
OUT = os.path.dirname(os.path.abspath(__file__))

NOTE_ORDER = ["af-filter", "pre-filtering-failed", "bad-chisq", "high-bse",
              "perfectly-separable-data", "matrix-inversion-error", "firth-fail",
              "missing-data-error", "lrt-filtering-failed"]


def notes_mask(notes):
    m = 0
    for n in notes:
        m |= 1 << NOTE_ORDER.index(n)
    return m


def nan_if_none(x):
    return np.nan if x is None else float(x)


def seer_to_row(s, q):
    """Seer -> flat numeric row: prep,pvalue,kbeta,bse,intercept,betas[q],notes,prefilter,filter,max_lineage"""
    betas = np.full(q, np.nan)
    b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
    if b.ndim == 1 and b.shape[0] == q and q > 0:
        betas[:] = b
    ml = -1 if s.max_lineage is None else int(s.max_lineage)
    return ([nan_if_none(s.prep), nan_if_none(s.pvalue), nan_if_none(s.kbeta), nan_if_none(s.bse),
             nan_if_none(s.intercept)], betas, notes_mask(s.notes), int(bool(s.prefilter)),
            int(bool(s.filter)), ml)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------------------------
# Synthetic data (inputs are stored in the fixtures, so the GPU box never regenerates them)
# ----------------------------------------------------------------------------------------------
def synth_design(seed, N, q, nlin=8, continuous=False, binary_cov=False):
    rng = np.random.default_rng(seed)
    lin = rng.integers(0, nlin, N)
    shifts = rng.standard_normal((nlin, max(q, 1)))
    m = rng.standard_normal((N, q)) + (shifts[lin][:, :q] if q else 0.0)
    if binary_cov and q > 0:
        m[:, q - 1] = (lin == 0).astype(float)
    if q:
        m = m / np.abs(m).max(axis=0)
    eta = -0.3 * np.ones(N)
    if q >= 1:
        eta += 1.5 * m[:, 0]
    if q >= 2:
        eta -= 1.0 * m[:, 1]
    if continuous:
        y = eta + 0.8 * rng.standard_normal(N)
    else:
        y = (rng.random(N) < sigmoid(eta)).astype(float)
    return rng, lin, m, y


def synth_variants(rng, lin, y, m, V, nlin=8, specials=True, continuous=False):
    N = y.shape[0]
    K = np.zeros((V, N))
    afs = np.concatenate([np.linspace(0.005, 0.995, V - V // 4), rng.uniform(0.02, 0.98, V // 4)])
    for v in range(V):
        jitter = rng.uniform(-0.15, 0.15, nlin)
        pr = np.clip(afs[v] + jitter[lin] * min(afs[v], 1 - afs[v]) * 4, 0.0, 1.0)
        if v % 7 == 3:  # causal
            if continuous:
                pr = np.clip(pr + 0.25 * np.tanh(y - np.median(y)), 0, 1)
            else:
                pr = np.clip(pr + 0.25 * (y - 0.5), 0, 1)
        K[v] = (rng.random(N) < pr).astype(float)
    if specials and V >= 16:
        yb = (y > np.median(y)).astype(float) if continuous else y
        K[1] = yb                                   # perfectly separating
        K[2] = yb.copy(); fl = rng.choice(N, 2, replace=False); K[2, fl] = 1 - K[2, fl]   # nearly separating
        K[4] = 0.0; K[4, rng.choice(N, 3, replace=False)] = 1.0            # very rare
        K[5] = 1.0; K[5, rng.choice(N, 3, replace=False)] = 0.0            # very common
        K[6] = 1 - yb                               # perfectly anti-separating
        if m.shape[1] > 0:
            K[8] = (m[:, -1] > np.median(m[:, -1])).astype(float)          # collinear-ish with a covariate
            if set(np.unique(m[:, -1])) <= {0.0, 1.0}:
                K[9] = m[:, -1]                     # exact duplicate of a binary covariate column
    return K


def run_fixed_case(name, seed, N, q, V, continuous=False, binary_cov=False, pret=1.0, lrtt=1.0, sep=False):
    rng, lin, m, y = synth_design(seed, N, q, continuous=continuous, binary_cov=binary_cov)
    if sep:
        # phenotype that is linearly separable only once the variant column joins the covariates:
        # 2x2 table is healthy (no bad-chisq) but Newton diverges -> perfectly-separable-data / high-bse
        k_sep = (rng.random(N) < 0.5).astype(float)
        y = ((2.0 * m[:, 0] + 1.0 * (k_sep - 0.5)) > 0).astype(float)
        k_quasi = k_sep.copy()
        idx = np.where(np.abs(2.0 * m[:, 0] + 1.0 * (k_sep - 0.5)) < 0.05)[0][:3]
        k_quasi[idx] = 1 - k_quasi[idx]
    K = synth_variants(rng, lin, y, m, V, continuous=continuous)
    if sep:
        K[11] = k_sep; K[12] = k_quasi; K[13] = 1 - k_sep
    mm = m if q > 0 else np.zeros((0, 0))
    cov_df = pd.DataFrame([])
    c = cov_df.values
    null_res = fit_null(y, mm, cov_df, continuous, firth=False)
    if continuous:
        null_llf = float(null_res.llf)
        null_firth = np.nan
        null_arg, firth_arg = null_res, None
    else:
        null_llf = float(null_res.llf)
        null_firth = float(fit_null(y, mm, cov_df, continuous, firth=True))
        null_arg, firth_arg = null_llf, null_firth

    main = np.zeros((V, 5)); betas = np.full((V, q), np.nan)
    notes = np.zeros(V, dtype=np.int64); pf = np.zeros(V, dtype=np.int64); fl = np.zeros(V, dtype=np.int64)
    af = K.mean(axis=1)
    for v in range(V):
        # AF filter exactly as iter_variants does (input.py:608): p=None marks an af-filtered variant
        p_arg = y if (0.01 <= af[v] <= 0.99) else None
        s = fixed_effects_regression("v%d" % v, p_arg, K[v], mm, c, af[v], b"x", False, None,
                                     pret, lrtt, null_arg, firth_arg, [], [], continuous)
        r, b, nm, p_, f_, _ = seer_to_row(s, q)
        main[v] = r; betas[v] = b; notes[v] = nm; pf[v] = p_; fl[v] = f_

    out = dict(N=N, q=q, V=V, continuous=int(continuous), pret=pret, lrtt=lrtt, y=y, m=m, K=K.astype(np.uint8),
               null_llf=null_llf, null_firth=null_firth, main=main, betas=betas, notes=notes,
               prefilter=pf, filter=fl)

    # forced Firth on every variant (binary only): fit_firth() directly, model.py:414
    if not continuous:
        fm = np.full((V, 4), np.nan); fb = np.full((V, q), np.nan); fok = np.zeros(V, dtype=np.int64)
        for v in range(V):
            X = np.concatenate((np.ones((N, 1)), K[v].reshape(-1, 1), m), axis=1) if q else \
                np.concatenate((np.ones((N, 1)), K[v].reshape(-1, 1)), axis=1)
            mod = smf.Logit(y, X)
            sv = np.zeros(X.shape[1]); sv[0] = np.log(np.mean(y) / (1 - np.mean(y)))
            try:
                res = fit_firth(mod, sv, X, y) if (0.01 <= af[v] <= 0.99) else None
            except Exception as e:  # LinAlgError etc.
                res = None
                fok[v] = -1
            if res is not None:
                ic, kb, be, bse, fitll = res
                fm[v] = [ic, kb, bse, fitll]
                if q:
                    fb[v] = np.asarray(be, dtype=float)
                fok[v] = 1
        out.update(firth_main=fm, firth_betas=fb, firth_ok=fok)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "notes histogram:", {NOTE_ORDER[i]: int(((notes >> i) & 1).sum()) for i in range(9)})


# ----------------------------------------------------------------------------------------------
# LMM
# ----------------------------------------------------------------------------------------------
def synth_kinship(rng, N, nlin=10, nmark=600):
    lin = rng.integers(0, nlin, N)
    base = rng.uniform(0.05, 0.95, (nlin, nmark))
    G = (rng.random((N, nmark)) < base[lin]).astype(float)
    K = G.dot(G.T)
    K *= float(N) / np.trace(K)           # lmm.py:107-112
    return lin, G, K


def lmm_block_all(lmm, h2, X):
    r = fit_lmm_block(lmm, h2, X.copy())
    return np.stack([r["beta"], r["bse"], r["frac_h2"], r["p_values"]], axis=1)


def run_lmm_case(name, seed, N, D, V, h2_list, continuous=False):
    rng = np.random.default_rng(seed)
    lin, G, K = synth_kinship(rng, N)
    g = G[:, :40].dot(rng.standard_normal(40)); g = (g - g.mean()) / g.std()
    liab = g * 0.8 + rng.standard_normal(N) * 0.7
    y = liab if continuous else (liab > 0).astype(float)
    if D > 1:
        cov = rng.standard_normal((N, D - 1))
        cov[:, 0] = (lin == 1).astype(float)            # a binary covariate
        covar = np.c_[cov, np.ones((N, 1))]            # intercept LAST, lmm.py:95-99
    else:
        covar = np.ones((N, 1))
    lmm = lmm_cov(X=covar, Y=y.reshape(-1, 1), K=K.copy(), G=None, inplace=True)
    res = lmm.findH2()
    h2 = float(res["h2"]); nll = float(res["nLL"][0])
    S, U = lmm.getSU()
    m0 = np.zeros((N, 0))
    Kv = synth_variants(rng, lin, (y > np.median(y)).astype(float) if continuous else y, m0, V, nlin=10,
                        specials=True)
    Kv[3] = 1.0                                        # constant column  -> zeroed (lmm_cov.py:180-181)
    if D > 1:
        Kv[7] = covar[:, 0]                            # explained by a covariate -> zeroed
        Kv[10] = 1.0 - covar[:, 0]
    X = Kv.T.copy()
    blocks = {}
    for h in [h2] + list(h2_list):
        blocks["h2_%0.6f" % h] = lmm_block_all(lmm, h, X)
    # orchestration fit_lmm (lmm.py:125) with two threshold sets
    def orchestrate(fp, lp):
        variants = []
        for v in range(V):
            afv = float(Kv[v].mean())
            # load_var_block (input.py:693): pattern None marks an af-filtered variant
            var = LMM("v%d" % v, b"pat" if (0.01 <= afv <= 0.99) else None, afv,
                      np.nan, np.nan, np.nan, np.nan, np.nan, np.nan, [], [], set(), True, True)
            variants.append((var, y, Kv[v]))
        res = fit_lmm(lmm, h2, variants, X.copy(), False, [], covar, continuous, fp, lp)
        rows = np.array([[nan_if_none(r.prep), nan_if_none(r.pvalue), nan_if_none(r.kbeta),
                          nan_if_none(r.bse), nan_if_none(r.frac_h2)] for r in res])
        return rows, np.array([notes_mask(r.notes) for r in res]), \
            np.array([int(r.prefilter) for r in res]), np.array([int(r.filter) for r in res]), \
            [r.kmer for r in res]
    o1 = orchestrate(1.0, 1.0)
    o2 = orchestrate(0.05, 0.01)
    out = dict(N=N, D=D, V=V, continuous=int(continuous), y=y, covar=covar, U=U, S=S, h2=h2, nLL=nll, K=K,
               Kv=Kv.astype(np.uint8),
               h2_keys=np.array(sorted(blocks.keys())),
               o1_rows=o1[0], o1_notes=o1[1], o1_prefilter=o1[2], o1_filter=o1[3], o1_order=np.array(o1[4]),
               o2_rows=o2[0], o2_notes=o2[1], o2_prefilter=o2[2], o2_filter=o2[3], o2_order=np.array(o2[4]))
    for k_, v_ in blocks.items():
        out["blk_" + k_] = v_

    # cache-mismatch: U,S from an intercept-only fit reused with D covariates (run_test.sh:47 behaviour,
    # lmm.py:53-74).  Only when D > 1.
    if D > 1:
        lmm1 = lmm_cov(X=np.ones((N, 1)), Y=y.reshape(-1, 1), K=K.copy(), G=None, inplace=True)
        r1 = lmm1.findH2(); S1, U1 = lmm1.getSU()
        lmm_mis = lmm_cov(X=covar, Y=y.reshape(-1, 1), G=None, K=None)
        lmm_mis.U = U1; lmm_mis.S = S1
        out.update(mis_U=U1, mis_S=S1, mis_h2=float(r1["h2"]),
                   mis_blk=lmm_block_all(lmm_mis, float(r1["h2"]), X))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "h2 =", h2, "nLL =", nll, "o1 notes:", {NOTE_ORDER[i]: int(((o1[1] >> i) & 1).sum()) for i in range(9)})


def run_lmm_unit():
    """The reference's own LMM pins (tests/lmm_test.py) on its own data files, at full precision."""
    p = pd.read_csv(os.path.join(REF_TESTS, "subset.pheno"), index_col=0, sep="\t")["binary"]
    Sfile = os.path.join(REF_TESTS, "similarity_subset.tsv.gz")
    k100 = np.loadtxt(os.path.join(REF_TESTS, "unit_tests_data", "k.txt"))
    out = {}
    for tag, cov in (("nocov", pd.DataFrame([])),
                     ("cov", pd.read_csv(os.path.join(REF_TESTS, "covariates.txt"), index_col=0, sep="\t"))):
        x, lmm, h2 = initialise_lmm(p, cov, Sfile, lmm_cache_in=None, lmm_cache_out=None)
        r = lmm.findH2()
        S, U = lmm.getSU()
        k = k100[:x.shape[0]]
        blk = lmm_block_all(lmm, h2, k.reshape(-1, 1))
        bad_k = np.array([1.0] * 5 + [0.0] * (x.shape[0] - 5))
        blk_bad = lmm_block_all(lmm, h2, bad_k.reshape(-1, 1))
        out.update({tag + "_y": x.values.astype(float), tag + "_covar": lmm.X, tag + "_U": U, tag + "_S": S,
                    tag + "_h2": float(h2), tag + "_nLL": float(r["nLL"][0]), tag + "_k": k,
                    tag + "_blk": blk, tag + "_badk": bad_k, tag + "_blk_bad": blk_bad})
        # K after the reference's own subsetting and trace normalisation (lmm.py:77-112)
        Kdf = pd.read_csv(Sfile, index_col=0, sep="\t"); Kdf.index = Kdf.index.astype(str)
        Kdf = Kdf.loc[x.index, x.index]
        out[tag + "_K"] = Kdf.values * (float(len(x)) / np.diag(Kdf.values).sum())
        pb, bc = pre_filtering(x.values, k, False)
        pc, _ = pre_filtering(x.values, k, True)
        out[tag + "_prep_binary"] = pb; out[tag + "_prep_cont"] = pc
    np.savez_compressed(os.path.join(OUT, "lmm_unit.npz"), **out)
    print("lmm_unit: nLL", out["nocov_nLL"], out["cov_nLL"], "blk", out["nocov_blk"])


# ----------------------------------------------------------------------------------------------
# model.py unit-test replay on the reference's own data files
# ----------------------------------------------------------------------------------------------
def run_model_unit():
    d = os.path.join(REF_TESTS, "unit_tests_data")
    pb = np.loadtxt(os.path.join(d, "p_binary.txt")); pc = np.loadtxt(os.path.join(d, "p_continuous.txt"))
    k = np.loadtxt(os.path.join(d, "k.txt")); m = np.loadtxt(os.path.join(d, "m.txt"))
    cov = np.loadtxt(os.path.join(d, "cov.txt")); lin = np.loadtxt(os.path.join(d, "lin.txt"))
    fv = np.loadtxt(os.path.join(d, "firth_vars.txt"))
    out = dict(p_binary=pb, p_continuous=pc, k=k, m=m, cov=cov, lin=lin, firth_vars=fv)
    out["prep_binary"] = pre_filtering(pb, k, False)[0]
    out["prep_cont"] = pre_filtering(pc, k, True)[0]
    p2 = np.concatenate((np.ones(50), np.zeros(50))); k2 = np.concatenate((np.ones(45), np.zeros(55)))
    out["prep_bad"] = pre_filtering(p2, k2, False)[0]; out["prep_bad_cont"] = pre_filtering(p2, k2, True)[0]
    covdf = pd.DataFrame([])
    out["null_llf_binary"] = float(fit_null(pb, m, covdf, False).llf)
    out["null_firth_binary"] = float(fit_null(pb, m, covdf, False, firth=True))
    out["null_llf_cont"] = float(fit_null(pc, m, covdf, True).llf)
    out["null_llf_binary_cov"] = float(fit_null(pb, m, pd.DataFrame(cov), False).llf)
    out["null_firth_binary_cov"] = float(fit_null(pb, m, pd.DataFrame(cov), False, firth=True))
    # firth_likelihood / fit_firth as in model_test.py:199-234
    X = np.concatenate((np.ones(100).reshape(-1, 1), k.reshape(-1, 1)), axis=1)
    mod = smf.Logit(pb, X)
    out["firth_like_fv"] = float(firth_likelihood(fv[:2] if fv.shape[0] > 2 else fv, mod)) if fv.ndim == 1 else np.nan
    sv = np.zeros(2); sv[0] = np.log(np.mean(pb) / (1 - np.mean(pb)))
    out["fit_firth_k"] = np.array([nan_if_none(z) if not isinstance(z, (list, tuple)) else np.nan
                                   for z in fit_firth(mod, sv, X, pb)], dtype=float)
    # fixed_effects_regression scenarios
    scen = []
    def add(tag, p, kk, mm, cc, pret, lrtt, continuous, lineage=False):
        cdf = pd.DataFrame(cc) if cc.shape[0] else pd.DataFrame([])
        nr = fit_null(p, mm, cdf, continuous)
        if continuous:
            na, fa = nr, None
        else:
            na, fa = float(nr.llf), float(fit_null(p, mm, cdf, False, firth=True))
        s = fixed_effects_regression("variant", p, kk, mm, cc, 0.2, "pattern", lineage, lin if lineage else None,
                                     pret, lrtt, na, fa, [], [], continuous)
        q = mm.shape[1] + (cc.shape[1] if cc.ndim == 2 and cc.shape[0] else 0)
        r, b, nm, pf, fl, ml = seer_to_row(s, q)
        scen.append(tag)
        out["s_%s_main" % tag] = np.array(r); out["s_%s_betas" % tag] = b
        out["s_%s_flags" % tag] = np.array([nm, pf, fl, ml])
        out["s_%s_null" % tag] = np.array([float(nr.llf), np.nan if fa is None else fa])
    e = np.empty((0,))
    add("bin", pb, k, m, e, 1, 1, False)
    add("bin_pre", pb, k, m, e, 0.05, 1, False)
    add("bin_lrt", pb, k, m, e, 1, 0.05, False)
    add("bin_cov", pb, k, m, cov, 1, 1, False)
    badk = np.array([1] * 5 + [0] * 95, dtype=float)
    add("bin_bad", pb, badk, m, e, 1, 1, False)
    add("bin_lin", pb, k, m, e, 1, 1, False, lineage=True)
    add("cont", pc, k, m, e, 1, 1, True)
    add("cont_pre", pc, k, m, e, 0.05, 1, True)
    add("cont_lrt", pc, k, m, e, 1, 0.05, True)
    add("cont_cov", pc, k, m, cov, 1, 1, True)
    add("bin_nodist", pb, k, np.zeros((0, 0)), e, 1, 1, False)
    add("cont_nodist", pc, k, np.zeros((0, 0)), e, 1, 1, True)
    out["scenarios"] = np.array(scen)
    out["lineage_idx"] = np.array([nan_if_none(fit_lineage_effect(lin, e, k)),
                                   nan_if_none(fit_lineage_effect(lin, cov, k))])
    np.savez_compressed(os.path.join(OUT, "model_unit.npz"), **out)
    print("model_unit:", {s: out["s_%s_main" % s] for s in scen[:3]})


# ----------------------------------------------------------------------------------------------
# boundary: format_output / hash_pattern
# ----------------------------------------------------------------------------------------------
def run_boundary():
    items = []
    def seer(**kw):
        base = dict(kmer="ACGT", pattern=b"x", af=0.2, prep=0.5365065578449575, pvalue=0.0012345678, kbeta=1.25e-3,
                    bse=0.31415, intercept=-0.3333, betas=np.array([0.1, -0.2, 3e-11]), max_lineage=None,
                    kstrains=["s1", "s2"], nkstrains=["s3"], notes=set(), prefilter=False, filter=False)
        base.update(kw); return Seer(**base)
    def lmm(**kw):
        base = dict(kmer="TTGA", pattern=b"x", af=0.77, prep=1e-30, pvalue=2.5e-301, kbeta=-12.5, bse=1e3,
                    frac_h2=0.0123, max_lineage=None, kstrains=["a"], nkstrains=["b", "c"], notes=set(),
                    prefilter=False, filter=False)
        base.update(kw); return LMM(**base)
    cases = [
        ("seer", seer(), None, False), ("seer", seer(notes={"bad-chisq"}), None, False),
        ("seer", seer(pvalue=np.nan, kbeta=np.nan, bse=np.nan, intercept=np.nan, betas=np.array([]),
                      notes={"pre-filtering-failed"}), None, False),
        ("seer", seer(max_lineage=1), ["A", "B", "C"], False), ("seer", seer(), ["A", "B"], True),
        ("seer", seer(betas=np.array([np.nan, 1.0])), None, True),
        ("seer", seer(af=1.0, prep=9.995e-5, pvalue=9.994999e-5, kbeta=0.0, bse=np.inf), None, False),
        ("lmm", lmm(), None, False), ("lmm", lmm(frac_h2=np.nan, notes={"lrt-filtering-failed"}), None, False),
        ("lmm", lmm(max_lineage=0), ["L0", "L1"], True), ("lmm", lmm(pvalue=1.0, kbeta=0.0, bse=np.inf, frac_h2=0.0), None, False),
    ]
    for model, it, ld, ps in cases:
        d = it._asdict()
        d["pattern"] = d["pattern"].decode()
        d["notes"] = sorted(d["notes"])
        if "betas" in d:
            d["betas"] = [None if not np.isfinite(x) else float(x) for x in d["betas"]]
        for kf in ("prep", "pvalue", "kbeta", "bse", "intercept", "frac_h2", "af"):
            if kf in d:
                x = d[kf]
                d[kf] = ("nan" if np.isnan(x) else ("inf" if x > 0 else "-inf")) if not np.isfinite(x) else float(x)
        items.append(dict(model=model, item=d, lineage_dict=ld, print_samples=ps,
                          expected=format_output(it, ld, model, ps)))
    hashes = []
    rng = np.random.default_rng(7)
    for n in (1, 8, 50, 100, 1000):
        k = (rng.random(n) < 0.3).astype(np.int64)
        hashes.append(dict(dtype="int64", k=k.tolist(), digest=hash_pattern(k).decode()))
        kf = k.astype(np.float64)
        hashes.append(dict(dtype="float64", k=kf.tolist(), digest=hash_pattern(kf).decode()))
    with open(os.path.join(OUT, "boundary.json"), "w") as fh:
        json.dump(dict(format_output=items, hash_pattern=hashes, note_order=NOTE_ORDER), fh, indent=1)
    print("boundary:", len(items), "format cases,", len(hashes), "hashes")


def run_lineage_case(name, seed, N, nl, j, V, clusters=False):
    """fit_lineage_effect (model.py:151-199) over a block of variants."""
    rng, lin_id, m, y = synth_design(seed, N, max(nl, 2), nlin=nl + 1 if clusters else 8)
    if clusters:
        lin = np.stack([(lin_id == c).astype(float) for c in range(nl + 1)], axis=1)[:, 1:]    # one cluster dropped (__main__.py:409-412)
    else:
        lin = m[:, :nl]
    cov = rng.standard_normal((N, j)) if j else np.empty((0,))
    K = synth_variants(rng, lin_id, y, m, V, nlin=nl + 1 if clusters else 8)
    if clusters:
        K[9] = lin[:, 0]                       # variant == a cluster indicator -> separation
    out = np.array([-1 if (lambda r: r is None)(fit_lineage_effect(lin, cov, K[v])) else int(fit_lineage_effect(lin, cov, K[v]))
                    for v in range(V)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), N=N, nl=nl, j=j, V=V, lin=lin, cov=cov if j else np.zeros((N, 0)),
                        K=K.astype(np.uint8), max_lineage=out)
    print(name, "None:", int((out < 0).sum()), "hist:", np.bincount(out[out >= 0], minlength=nl))


if __name__ == "__main__":
    what = sys.argv[1:] or ["unit", "glm", "lmm", "lineage", "boundary"]
    if "unit" in what:
        run_model_unit()
        run_lmm_unit()
    if "glm" in what:
        run_fixed_case("glm_N100_q0", 11, 100, 0, 48)
        run_fixed_case("glm_N100_q3", 12, 100, 3, 48)
        run_fixed_case("glm_N300_q10", 13, 300, 10, 64)
        run_fixed_case("glm_N300_q4_bincov", 14, 300, 4, 48, binary_cov=True)
        run_fixed_case("glm_N1000_q10", 15, 1000, 10, 64)
        run_fixed_case("glm_N300_q10_filt", 16, 300, 10, 64, pret=0.2, lrtt=0.05)
        run_fixed_case("glm_N300_q2_sep", 17, 300, 2, 32, sep=True)
        run_fixed_case("ols_N100_q3", 21, 100, 3, 48, continuous=True)
        run_fixed_case("ols_N300_q10", 22, 300, 10, 64, continuous=True)
        run_fixed_case("ols_N300_q0", 23, 300, 0, 48, continuous=True)
    if "ols_rankdef" in what or "glm" in what:
        run_fixed_case("ols_N300_q4_bincov", 24, 300, 4, 48, continuous=True, binary_cov=True)   # K[9] duplicates a covariate: pinv
    if "lmm" in what:
        run_lmm_case("lmm_N50_D1", 31, 50, 1, 32, [0.0, 0.25, 0.9])
        run_lmm_case("lmm_N300_D1", 32, 300, 1, 64, [0.0, 0.25, 0.5, 0.9])
        run_lmm_case("lmm_N300_D3", 33, 300, 3, 64, [0.0, 0.5, 0.9])
        run_lmm_case("lmm_N200_D2_cont", 34, 200, 2, 48, [0.3], continuous=True)
    if "lineage" in what:
        run_lineage_case("lineage_N200_l3", 41, 200, 3, 0, 48)
        run_lineage_case("lineage_N300_l10_j2", 42, 300, 10, 2, 48)
        run_lineage_case("lineage_N300_clusters5", 43, 300, 5, 0, 48, clusters=True)
    if "boundary" in what:
        run_boundary()
