#!/opt/conda/bin/python3.9
"""Reference-generated goldens at the BASELINE size (N = 5000): the imported reference itself on the bench's own designs.

    part lmm     pyseer/lmm.py:26 initialise_lmm (similarity TSV written to /tmp from a stored seed -> findH2) and :228 fit_lmm_block
                 on 64 rows (AF 0.01-0.05, 0.95-0.99, mid AF, lineage markers that load on the top eigenvectors, causal, constant),
                 plus :125 fit_lmm on the same rows with thresholds (0.05, 0.01)                         -> n5000_lmm.npz
    part glm     pyseer/model.py:202 fixed_effects_regression on 64 rows of bench.synth_glm_inputs(5000, 10)   -> n5000_glm.npz
    part firth   pyseer/model.py:414 fit_firth on the rows tools/gpu_dump_disputed_firth.py dumped from the C4 workload
                 (every row on which the library's three step-halving modes or the C restatement disagree, and 24 controls)
                 -> n5000_firth.npz; `--slice i/n` does rows i::n (one process per slice, merged by `--merge`)
  round 5 (VERDICT r04 "next" 4):
    part firth_random  pyseer/model.py:414 fit_firth and :202 fixed_effects_regression on the FIRST 256 rows of the bench's C4 / C2N5000
                 batch (tools/gpu_dump_c4_rows.py: synth_bits(2^18, 5000, rb, 4242), nobody chose them)       -> n5000_random.npz
                 (`firth_random i/n` per slice, `merge_random n`)
    part ols     pyseer/model.py:299-312 (continuous phenotype, OLS) on 64 rows                          -> n5000_ols.npz
    part lineage pyseer/model.py:151-199 fit_lineage_effect on 32 rows: 10 MDS-like columns + 2 covariates; 12 cluster indicators  -> n5000_lineage.npz
    part lmm_d3  pyseer/lmm.py:26,95-99,228: D = 3 covariates (fresh decomposition of the projected kernel) and the cache-mismatch case
                 (U, S, h2 of the D = 1 cache with the D = 3 covariates, run_test.sh:47)                  -> n5000_lmm_d3.npz
    part cap35   pyseer/model.py:316-330 on a design with a quasi-separating covariate: statsmodels' Newton stops at its 35-iteration cap
                 (SM:base/optimizer.py:407-427) without failing                                          -> n5000_cap35.npz

Runs ONLY in the build container:
    cd /root/repo/tests/golden && PYTHONPATH=_harness:_harness/stubs:/root/reference PYTHONDONTWRITEBYTECODE=1 \
    OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 /opt/conda/bin/python3.9 -W ignore make_n5000_golden.py lmm|glm|firth ...

Committed: seeds, packed presence bits, phenotype, the 10 covariate columns, and the numbers the reference returned.  NOT committed: U
(the tests decompose K again from the seed; the statistics do not depend on eigenvector signs or on the solver)."""
import shim  # noqa: F401  must be first
import os
import sys
import time
import warnings
import numpy as np
import pandas as pd

warnings.simplefilter("ignore")
import statsmodels.formula.api as smf
from pyseer.model import fit_null, fit_firth, fixed_effects_regression, fit_lineage_effect
from pyseer.lmm import initialise_lmm, fit_lmm, fit_lmm_block
from pyseer.classes import LMM

OUT = os.path.dirname(os.path.abspath(__file__))
NOTE_ORDER = ["af-filter", "pre-filtering-failed", "bad-chisq", "high-bse", "perfectly-separable-data", "matrix-inversion-error",
              "firth-fail", "missing-data-error", "lrt-filtering-failed"]
N = 5000
E0 = pd.DataFrame([])


def notes_mask(notes):
    m = 0
    for n in notes:
        m |= 1 << NOTE_ORDER.index(n)
    return m


def nn(x):
    return np.nan if x is None else float(x)


def pack(K):
    return np.packbits(np.asarray(K, dtype=np.uint8), axis=1, bitorder="little")


def unpack(bits, n):
    return np.unpackbits(bits, axis=1, bitorder="little")[:, :n].astype(np.float64)


# ------------------------------------------------------------------------------------------------------------------
# the bench's designs, restated with numpy only (bench.py synth_lmm_inputs / synth_glm_inputs use the same draws in the same order)
# ------------------------------------------------------------------------------------------------------------------
def lmm_design(seed=1003):
    rng = np.random.default_rng(seed)
    nlin, nmark = 40, 2000
    lin = rng.integers(0, nlin, N)
    base = rng.uniform(0.05, 0.95, (nlin, nmark))
    G = (rng.random((N, nmark)) < base[lin]).astype(np.float64)
    K = G.dot(G.T)
    g = G[:, :60].dot(rng.standard_normal(60)); g = (g - g.mean()) / g.std()
    y = ((0.7 * g + 0.7 * rng.standard_normal(N)) > 0).astype(np.float64)
    return lin, G, K, y


def glm_design(seed=1002, q=10):
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    return y, W


def part_lmm():
    t0 = time.time()
    lin, G, K, y = lmm_design()
    names = ["s%04d" % i for i in range(N)]
    tsv = "/tmp/n5000_similarity.tsv"
    pd.DataFrame(K, index=names, columns=names).to_csv(tsv, sep="\t")
    p = pd.Series(y, index=names)
    print("K written (%.0f s); initialise_lmm ..." % (time.time() - t0), flush=True)
    x, lmm, h2 = initialise_lmm(p, E0, tsv, lmm_cache_in=None, lmm_cache_out=None)
    res = lmm.findH2()
    h2 = float(h2); nll = float(res["nLL"][0])
    print("h2 = %.12g nLL = %.12g (%.0f s)" % (h2, nll, time.time() - t0), flush=True)
    rng = np.random.default_rng(55003)
    rows = []
    for af in np.r_[np.linspace(0.0102, 0.05, 10), np.linspace(0.95, 0.9898, 10), np.linspace(0.06, 0.94, 20)]:
        rows.append((rng.random(N) < af).astype(float))
    # lineage markers (they load on the top eigenvectors of K) and noisy copies of them
    for j in range(8):
        rows.append(G[:, 100 + 37 * j].copy())
    for j in range(4):
        r = G[:, 500 + 11 * j].copy(); f = rng.choice(N, 250, replace=False); r[f] = 1 - r[f]; rows.append(r)
    # lineage indicators (blocks of relatives) and causal rows
    for l_ in (0, 7, 19):
        rows.append((lin == l_).astype(float))
    for s in (0.15, 0.3, 0.5):
        rows.append((rng.random(N) < np.clip(0.3 + s * (y - 0.5), 0, 1)).astype(float))
    rows.append(np.ones(N)); rows.append(np.zeros(N)); rows.append(y.copy())            # constant columns (zeroed, lmm_cov.py:180), the phenotype itself
    r = np.zeros(N); r[rng.choice(N, 30, replace=False)] = 1; rows.append(r)            # AF 0.006: af-filtered upstream
    while len(rows) < 64:
        rows.append((rng.random(N) < rng.uniform(0.02, 0.98)).astype(float))
    Kv = np.array(rows)
    X = Kv.T.copy()
    r = fit_lmm_block(lmm, h2, X.copy())
    blk = np.stack([r["beta"], r["bse"], r["frac_h2"], r["p_values"]], axis=1)
    r0 = fit_lmm_block(lmm, 0.0, X.copy()); blk0 = np.stack([r0["beta"], r0["bse"], r0["frac_h2"], r0["p_values"]], axis=1)

    def orchestrate(fp, lp):
        variants = []
        for v in range(Kv.shape[0]):
            afv = float(Kv[v].mean())
            var = LMM("v%d" % v, b"pat" if (0.01 <= afv <= 0.99) else None, afv, np.nan, np.nan, np.nan, np.nan, np.nan, np.nan, [], [],
                      set(), True, True)
            variants.append((var, y, Kv[v]))
        out = fit_lmm(lmm, h2, variants, X.copy(), False, [], lmm.X, False, fp, lp)
        return (np.array([[nn(o.prep), nn(o.pvalue), nn(o.kbeta), nn(o.bse), nn(o.frac_h2)] for o in out]),
                np.array([notes_mask(o.notes) for o in out]), np.array([int(o.prefilter) for o in out]), np.array([int(o.filter) for o in out]),
                np.array([int(o.kmer[1:]) for o in out]))                    # fit_lmm returns filtered variants first, then the tested ones
    o1 = orchestrate(1.0, 1.0); o2 = orchestrate(0.05, 0.01)
    Kn = K * (float(N) / np.trace(K))
    np.savez_compressed(os.path.join(OUT, "n5000_lmm.npz"), N=N, seed=1003, y=y.astype(np.uint8), bits=pack(Kv), h2=h2, nLL=nll, blk=blk, blk_h2_0=blk0,
                        o1_rows=o1[0], o1_notes=o1[1], o1_prefilter=o1[2], o1_filter=o1[3], o1_order=o1[4],
                        o2_rows=o2[0], o2_notes=o2[1], o2_prefilter=o2[2], o2_filter=o2[3], o2_order=o2[4],
                        K_trace=np.trace(K), K_sum=K.sum(), K_probe=Kn[[0, 1, 17, 4999], [0, 2, 4000, 4998]],
                        S_top=np.sort(np.asarray(lmm.S))[-8:])
    print("n5000_lmm: %d rows, %.0f s; notes o2 %s" % (Kv.shape[0], time.time() - t0,
          {NOTE_ORDER[i]: int(((o2[1] >> i) & 1).sum()) for i in range(9)}))


def part_glm():
    t0 = time.time()
    y, W = glm_design()
    nl = float(fit_null(y, W, E0, False, firth=False).llf)
    nf = float(fit_null(y, W, E0, False, firth=True))
    print("null llf %.12g firth %.12g (%.0f s)" % (nl, nf, time.time() - t0), flush=True)
    rng = np.random.default_rng(55002)
    rows = []
    for af in np.r_[np.linspace(0.02, 0.98, 30), rng.uniform(0.02, 0.98, 8)]:
        rows.append((rng.random(N) < af).astype(float))
    for s in (0.1, 0.2, 0.35, 0.5, -0.2, -0.4):                                          # causal, both signs
        rows.append((rng.random(N) < np.clip(0.25 + s * (y - 0.5), 0, 1)).astype(float))
    for j in (0, 1, 5):                                                                  # correlated with a covariate
        rows.append((W[:, j] + 0.3 * rng.standard_normal(N) > 0.2).astype(float))
    for c in (51, 60, 4945):                                                             # rare / common, healthy table at this N
        r = np.zeros(N); r[rng.choice(N, c, replace=False)] = 1; rows.append(r)
    ones = np.where(y == 1)[0]; zeros = np.where(y == 0)[0]
    for c1, c0 in ((55, 0), (60, 1), (200, 0), (0, 70), (1, 64), (52, 4), (1200, 1)):    # a cell <= 1 (or two <= 5): bad-chisq -> fit_firth
        r = np.zeros(N); r[rng.choice(ones, c1, replace=False)] = 1; r[rng.choice(zeros, c0, replace=False)] = 1; rows.append(r)
    r = np.ones(N); r[rng.choice(zeros, 55, replace=False)] = 0; rows.append(r)          # common, every non-carrier has phenotype 0
    # carriers only among the phenotype-1 samples but 3: strong effect; and all carriers at the extremes of W[:,0]
    r = np.zeros(N); r[rng.choice(ones, 80, replace=False)] = 1; r[rng.choice(zeros, 3, replace=False)] = 1; rows.append(r)
    r = np.zeros(N); r[rng.choice(ones, 300, replace=False)] = 1; r[rng.choice(zeros, 6, replace=False)] = 1; rows.append(r)
    r = np.zeros(N); r[rng.choice(N, 40, replace=False)] = 1; rows.append(r)             # AF 0.008: af-filtered
    while len(rows) < 64:
        rows.append((rng.random(N) < rng.uniform(0.02, 0.98)).astype(float))
    Kv = np.array(rows); V = Kv.shape[0]; q = W.shape[1]
    main = np.zeros((V, 5)); betas = np.full((V, q), np.nan); notes = np.zeros(V, dtype=np.int64)
    pf = np.zeros(V, dtype=np.int64); fl = np.zeros(V, dtype=np.int64)
    for v in range(V):
        af = float(Kv[v].mean())
        s = fixed_effects_regression("v%d" % v, y if 0.01 <= af <= 0.99 else None, Kv[v], W, E0.values, af, b"x", False, None, 1.0, 1.0,
                                     nl, nf, [], [], False)
        main[v] = [nn(s.prep), nn(s.pvalue), nn(s.kbeta), nn(s.bse), nn(s.intercept)]
        b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
        if b.shape == (q,):
            betas[v] = b
        notes[v] = notes_mask(s.notes); pf[v] = int(bool(s.prefilter)); fl[v] = int(bool(s.filter))
        print(v, "%.1f s" % (time.time() - t0), sorted(s.notes), flush=True)
    np.savez_compressed(os.path.join(OUT, "n5000_glm.npz"), N=N, q=q, seed=1002, y=y.astype(np.uint8), W=W, bits=pack(Kv), null_llf=nl, null_firth=nf,
                        main=main, betas=betas, notes=notes, prefilter=pf, filter=fl)
    print("n5000_glm:", {NOTE_ORDER[i]: int(((notes >> i) & 1).sum()) for i in range(9)})


def part_firth(sl):
    i0, n = (int(x) for x in sl.split("/"))
    d = np.load(os.path.join("/root/repo/gpurun_out/r04/disputed_firth_rows.npz"))
    y, W = glm_design()
    assert np.array_equal(y, d["y"]) and np.array_equal(W, d["W"]), "the dump was made on another design"
    rows = unpack(d["bits"], N)
    idx = np.arange(rows.shape[0])[i0::n]
    q = W.shape[1]
    out = np.full((rows.shape[0], 4 + q + 1), np.nan)            # intercept, kbeta, bse, fitll, betas[q], status (1 fit, 0 None, -1 exception)
    t0 = time.time()
    for v in idx:
        X = np.concatenate((np.ones((N, 1)), rows[v].reshape(-1, 1), W), axis=1)
        sv = np.zeros(X.shape[1]); sv[0] = np.log(np.mean(y) / (1 - np.mean(y)))
        try:
            res = fit_firth(smf.Logit(y, X), sv, X, y)
            st = 1 if res is not None else 0
        except Exception as e:
            res, st = None, -1
        if res is not None:
            ic, kb, be, bse, fitll = res
            out[v, :4] = [ic, kb, bse, fitll]; out[v, 4:4 + q] = np.asarray(be, dtype=float)
        out[v, -1] = st
        print(v, st, "%.1f s" % (time.time() - t0), flush=True)
    np.save("/tmp/n5000_firth_part_%d_of_%d.npy" % (i0, n), out)


def merge_firth(n):
    d = np.load(os.path.join("/root/repo/gpurun_out/r04/disputed_firth_rows.npz"))
    parts = [np.load("/tmp/n5000_firth_part_%d_of_%d.npy" % (i, n)) for i in range(n)]
    out = parts[0].copy()
    for i in range(n):
        out[i::n] = parts[i][i::n]
    assert np.isfinite(out[:, -1]).all()
    y, W = glm_design()
    nf = float(fit_null(y, W, E0, False, firth=True)); nl = float(fit_null(y, W, E0, False, firth=False).llf)
    q = W.shape[1]
    keep = {k: d[k] for k in d.files if k.split("_")[0] in ("restatement", "noise", "literal", "strict")}
    np.savez_compressed(os.path.join(OUT, "n5000_firth.npz"), N=N, q=q, seed=1002, seed_bits=int(d["seed_bits"]), index=d["index"], bits=d["bits"],
                        disputed=d["disputed"], y=y.astype(np.uint8), W=W, null_llf=nl, null_firth=nf,
                        firth_main=out[:, :4], firth_betas=out[:, 4:4 + q], firth_ok=out[:, -1].astype(np.int64), **{"r3_" + k: v for k, v in keep.items()})
    print("n5000_firth: %d rows, reference fit %d, None %d, exception %d" % (out.shape[0], (out[:, -1] == 1).sum(), (out[:, -1] == 0).sum(),
                                                                          (out[:, -1] == -1).sum()))


RANDOM_DUMP = "/root/repo/gpurun_out/r05/c4_random_rows.npz"


def part_firth_random(sl):
    """The reference on rows nobody picked: fit_firth (C4's mode) and fixed_effects_regression (C2N5000's) on the first rows of the bench batch."""
    i0, n = (int(x) for x in sl.split("/"))
    d = np.load(RANDOM_DUMP)
    y, W = glm_design()
    assert np.array_equal(y, d["y"].astype(float)) and np.array_equal(W, d["W"]), "the dump was made on another design"
    rows = unpack(d["bits"], N); q = W.shape[1]
    nl = float(fit_null(y, W, E0, False, firth=False).llf); nf = float(fit_null(y, W, E0, False, firth=True))
    assert abs(nl - float(d["null_llf"])) < 1e-9 * abs(nl) and abs(nf - float(d["null_firth"])) < 1e-9 * abs(nf)
    out = np.full((rows.shape[0], 4 + q + 1), np.nan)            # intercept, kbeta, bse, fitll, betas[q], status
    glm = np.full((rows.shape[0], 5 + q + 3), np.nan)            # prep, pvalue, kbeta, bse, intercept, betas[q], notes, prefilter, filter
    t0 = time.time()
    for v in np.arange(rows.shape[0])[i0::n]:
        af = float(rows[v].mean())
        if 0.01 <= af <= 0.99:
            X = np.concatenate((np.ones((N, 1)), rows[v].reshape(-1, 1), W), axis=1)
            sv = np.zeros(X.shape[1]); sv[0] = np.log(np.mean(y) / (1 - np.mean(y)))
            try:
                res = fit_firth(smf.Logit(y, X), sv, X, y); st = 1 if res is not None else 0
            except Exception:
                res, st = None, -1
            if res is not None:
                ic, kb, be, bse, fitll = res
                out[v, :4] = [ic, kb, bse, fitll]; out[v, 4:4 + q] = np.asarray(be, dtype=float)
            out[v, -1] = st
        else:
            out[v, -1] = 2                                        # af-filtered upstream: fit_firth never sees it
        s = fixed_effects_regression("v%d" % v, y if 0.01 <= af <= 0.99 else None, rows[v], W, E0.values, af, b"x", False, None, 1.0, 1.0,
                                     nl, nf, [], [], False)
        glm[v, :5] = [nn(s.prep), nn(s.pvalue), nn(s.kbeta), nn(s.bse), nn(s.intercept)]
        b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
        if b.shape == (q,):
            glm[v, 5:5 + q] = b
        glm[v, 5 + q:] = [notes_mask(s.notes), int(bool(s.prefilter)), int(bool(s.filter))]
        print(v, int(out[v, -1]), sorted(s.notes), "%.1f s" % (time.time() - t0), flush=True)
    np.save("/tmp/n5000_random_part_%d_of_%d.npy" % (i0, n), np.concatenate([out, glm], axis=1))


def merge_random(n):
    d = np.load(RANDOM_DUMP)
    parts = [np.load("/tmp/n5000_random_part_%d_of_%d.npy" % (i, n)) for i in range(n)]
    al = parts[0].copy()
    for i in range(n):
        al[i::n] = parts[i][i::n]
    y, W = glm_design(); q = W.shape[1]
    out, glm = al[:, :4 + q + 1], al[:, 4 + q + 1:]
    assert np.isfinite(out[:, -1]).all()
    nf = float(fit_null(y, W, E0, False, firth=True)); nl = float(fit_null(y, W, E0, False, firth=False).llf)
    np.savez_compressed(os.path.join(OUT, "n5000_random.npz"), N=N, q=q, seed=1002, seed_bits=int(d["seed_bits"]), bits=d["bits"], y=y.astype(np.uint8), W=W,
                        null_llf=nl, null_firth=nf, firth_main=out[:, :4], firth_betas=out[:, 4:4 + q], firth_ok=out[:, -1].astype(np.int64),
                        main=glm[:, :5], betas=glm[:, 5:5 + q], notes=glm[:, 5 + q].astype(np.int64), prefilter=glm[:, 6 + q].astype(np.int64),
                        filter=glm[:, 7 + q].astype(np.int64))
    print("n5000_random: %d rows; fit_firth fit %d, None %d, exception %d, af-filtered %d; notes %s" % (
        out.shape[0], (out[:, -1] == 1).sum(), (out[:, -1] == 0).sum(), (out[:, -1] == -1).sum(), (out[:, -1] == 2).sum(),
        {NOTE_ORDER[i]: int(((glm[:, 5 + q].astype(np.int64) >> i) & 1).sum()) for i in range(9)}))


def variant_rows(rng, y01, W, n_rows):
    """A spread of rows for the OLS / lineage parts: AF grid, random AF, rows tied to the phenotype and to covariates, rare / common rows."""
    rows = []
    for af in np.r_[np.linspace(0.02, 0.98, n_rows // 2), rng.uniform(0.02, 0.98, n_rows // 8)]:
        rows.append((rng.random(N) < af).astype(float))
    for s_ in (0.1, 0.25, 0.4, -0.2, -0.35):
        rows.append((rng.random(N) < np.clip(0.3 + s_ * (y01 - 0.5), 0, 1)).astype(float))
    for j in (0, 1, 7):
        rows.append((W[:, j] + 0.3 * rng.standard_normal(N) > 0.2).astype(float))
    for c in (51, 75, 4940):
        r = np.zeros(N); r[rng.choice(N, c, replace=False)] = 1; rows.append(r)
    r = np.zeros(N); r[rng.choice(N, 30, replace=False)] = 1; rows.append(r)             # AF 0.006: af-filtered
    while len(rows) < n_rows:
        rows.append((rng.random(N) < rng.uniform(0.02, 0.98)).astype(float))
    return np.array(rows[:n_rows])


def part_ols():
    t0 = time.time()
    _, W = glm_design()
    rng = np.random.default_rng(55012)
    yc = 0.4 - 1.2 * W[:, 0] + 0.8 * W[:, 2] + 0.9 * rng.standard_normal(N)              # continuous phenotype
    Kv = variant_rows(rng, (yc > np.median(yc)).astype(float), W, 64)
    Kv[40] = (rng.random(N) < np.clip(0.3 + 0.15 * np.tanh(yc - np.median(yc)), 0, 1)).astype(float)   # causal for the continuous trait
    Kv[41] = (rng.random(N) < np.clip(0.5 - 0.3 * np.tanh(yc - np.median(yc)), 0, 1)).astype(float)
    nr = fit_null(yc, W, E0, True)
    V = Kv.shape[0]; q = W.shape[1]
    main = np.zeros((V, 5)); betas = np.full((V, q), np.nan); notes = np.zeros(V, dtype=np.int64)
    pf = np.zeros(V, dtype=np.int64); fl = np.zeros(V, dtype=np.int64)
    for thr, tag in (((1.0, 1.0), ""), ((0.05, 0.01), "_thr")):
        for v in range(V):
            af = float(Kv[v].mean())
            s = fixed_effects_regression("v%d" % v, yc if 0.01 <= af <= 0.99 else None, Kv[v], W, E0.values, af, b"x", False, None, thr[0], thr[1],
                                         nr, None, [], [], True)
            main[v] = [nn(s.prep), nn(s.pvalue), nn(s.kbeta), nn(s.bse), nn(s.intercept)]
            b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
            betas[v] = b if b.shape == (q,) else np.nan
            notes[v] = notes_mask(s.notes); pf[v] = int(bool(s.prefilter)); fl[v] = int(bool(s.filter))
        if tag == "":
            keep = dict(main=main.copy(), betas=betas.copy(), notes=notes.copy(), prefilter=pf.copy(), filter=fl.copy())
        else:
            keep.update({"main_thr": main.copy(), "betas_thr": betas.copy(), "notes_thr": notes.copy(), "prefilter_thr": pf.copy(), "filter_thr": fl.copy()})
    np.savez_compressed(os.path.join(OUT, "n5000_ols.npz"), N=N, q=q, seed=1002, y=yc, W=W, bits=pack(Kv), pret_thr=0.05, lrtt_thr=0.01, **keep)
    print("n5000_ols: %d rows, %.0f s; notes %s; with thresholds %s" % (V, time.time() - t0, {NOTE_ORDER[i]: int(((keep["notes"] >> i) & 1).sum()) for i in range(9)},
          {NOTE_ORDER[i]: int(((keep["notes_thr"] >> i) & 1).sum()) for i in range(9)}))


def part_lineage():
    t0 = time.time()
    y, W = glm_design()
    rng = np.random.default_rng(55022)
    lab = rng.integers(0, 12, N)
    Kv = variant_rows(rng, y, W, 32)
    for j, l_ in enumerate((0, 3, 7)):                                                    # rows that follow a cluster / an MDS axis
        Kv[20 + j] = ((lab == l_) ^ (rng.random(N) < 0.03)).astype(float)
    Kv[23] = (W[:, 4] > 0.1).astype(float); Kv[24] = (W[:, 9] + 0.05 * rng.standard_normal(N) > 0).astype(float)
    Kv[25] = (lab == 5).astype(float)                                                     # a cluster indicator itself: separation -> None
    cov = np.c_[(rng.random(N) < 0.4).astype(float), rng.standard_normal(N)]
    cl = np.zeros((N, 12)); cl[np.arange(N), lab] = 1.0; cl = cl[:, 1:]                   # indicators, one dropped as __main__.py:417-432 does
    out = {}
    for tag, lin, c in (("mds", W, cov), ("mds_nocov", W, E0.values), ("clusters", cl, E0.values)):
        ml = []
        for v in range(Kv.shape[0]):
            r = fit_lineage_effect(lin, c, Kv[v])
            ml.append(-1 if r is None else int(r))
        out["max_lineage_" + tag] = np.array(ml, dtype=np.int64)
        print(tag, ml, "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "n5000_lineage.npz"), N=N, bits=pack(Kv), W=W, cov=cov, clusters=cl, **out)


def part_lmm_d3():
    t0 = time.time()
    lin, G, K, y = lmm_design()
    names = ["s%04d" % i for i in range(N)]
    tsv = "/tmp/n5000_similarity.tsv"
    if not os.path.exists(tsv):
        pd.DataFrame(K, index=names, columns=names).to_csv(tsv, sep="\t")
    p = pd.Series(y, index=names)
    rng = np.random.default_rng(55033)
    cv = np.c_[(rng.random(N) < 0.35).astype(float) + 0.2 * (lin % 3 == 0), rng.standard_normal(N) + 0.5 * y]
    cov = pd.DataFrame(cv, index=names, columns=["c1", "c2"])
    cache1 = "/tmp/n5000_lmm_cache_D1.npz"
    if os.path.exists(cache1):
        os.remove(cache1)
    _, lmm1, h2_1 = initialise_lmm(p, E0, tsv, lmm_cache_in=None, lmm_cache_out=cache1)   # D = 1: writes the cache (arr_0 U, arr_1 S, arr_2 [h2])
    print("D=1 h2 = %.12g (%.0f s)" % (float(h2_1), time.time() - t0), flush=True)
    _, lmm3, h2_3 = initialise_lmm(p, cov, tsv, lmm_cache_in=None, lmm_cache_out=None)    # D = 3, fresh: eigh of the projected kernel
    nll3 = float(lmm3.findH2()["nLL"][0])
    print("D=3 h2 = %.12g nLL = %.12g (%.0f s)" % (float(h2_3), nll3, time.time() - t0), flush=True)
    _, lmmx, h2_x = initialise_lmm(p, cov, tsv, lmm_cache_in=cache1, lmm_cache_out=None)  # the cache of D = 1 under D = 3 covariates (lmm.py:57-76)
    assert float(h2_x) == float(h2_1)
    rows = []
    for af in np.r_[np.linspace(0.0102, 0.05, 6), np.linspace(0.95, 0.9898, 6), np.linspace(0.06, 0.94, 20)]:
        rows.append((rng.random(N) < af).astype(float))
    for j in range(8):
        rows.append(G[:, 150 + 41 * j].copy())
    for l_ in (2, 11):
        rows.append((lin == l_).astype(float))
    for s_ in (0.15, 0.3, 0.5):
        rows.append((rng.random(N) < np.clip(0.3 + s_ * (y - 0.5), 0, 1)).astype(float))
    rows.append((cv[:, 0] > 0.5).astype(float))                                          # a function of a covariate (not in its span: c1 has 4 levels)
    rows.append(np.ones(N)); rows.append(np.zeros(N))
    while len(rows) < 64:
        rows.append((rng.random(N) < rng.uniform(0.02, 0.98)).astype(float))
    Kv = np.array(rows); X = Kv.T.copy()

    def blk(l_, h):
        r = fit_lmm_block(l_, float(h), X.copy())
        return np.stack([r["beta"], r["bse"], r["frac_h2"], r["p_values"]], axis=1)
    b3 = blk(lmm3, h2_3); bx = blk(lmmx, h2_x)
    np.savez_compressed(os.path.join(OUT, "n5000_lmm_d3.npz"), N=N, seed=1003, y=y.astype(np.uint8), cov=cv, bits=pack(Kv),
                        h2_D1=float(h2_1), h2_D3=float(h2_3), nLL_D3=nll3, blk_D3=b3, blk_mismatch=bx, K_trace=np.trace(K), K_sum=K.sum(),
                        S_top_D3=np.sort(np.asarray(lmm3.S))[-8:], S_top_D1=np.sort(np.asarray(lmmx.S))[-8:])
    print("n5000_lmm_d3: %d rows, %.0f s" % (Kv.shape[0], time.time() - t0))


def part_cap35():
    """A covariate that quasi-separates the phenotype (every sample with z > 0 has y = 1, every one with z < 0 has y = 0, a tied group at
    z = 0 carries both): Newton's slope on it grows by about one per iteration and never meets the 1e-8 step tolerance; statsmodels stops at
    maxiter = 35 with a ConvergenceWarning and returns what it has (SM:base/optimizer.py:407-427), and the reference uses that."""
    t0 = time.time()
    rng = np.random.default_rng(55035)
    q = 3
    W = rng.standard_normal((N, q))
    tie = rng.random(N) < 0.04
    W[tie, 0] = 0.0
    W /= np.abs(W).max(axis=0)
    y = (W[:, 0] > 0).astype(float)
    y[tie] = (rng.random(int(tie.sum())) < 0.5).astype(float)
    nl_res = fit_null(y, W, E0, False, firth=False)
    nf = fit_null(y, W, E0, False, firth=True)
    nl = float(nl_res.llf)
    print("null: iterations %s converged %s llf %.12g firth %s" % (nl_res.mle_retvals.get("iterations"), nl_res.mle_retvals.get("converged"), nl,
          nf), flush=True)
    rows = []
    for af in np.linspace(0.05, 0.95, 10):
        rows.append((rng.random(N) < af).astype(float))
    for s_ in (0.2, -0.3):
        rows.append((rng.random(N) < np.clip(0.3 + s_ * (y - 0.5), 0, 1)).astype(float))
    r = np.zeros(N); r[rng.choice(np.where(tie)[0], 60, replace=False)] = 1; rows.append(r)       # carried inside the tied group only
    r = (rng.random(N) < 0.3).astype(float); rows.append(r * tie + (rng.random(N) < 0.02) * (~tie))
    Kv = np.array(rows); V = Kv.shape[0]
    main = np.zeros((V, 5)); betas = np.full((V, q), np.nan); notes = np.zeros(V, dtype=np.int64); its = np.zeros(V, dtype=np.int64); conv = np.zeros(V, dtype=np.int64)
    pf = np.zeros(V, dtype=np.int64); fl = np.zeros(V, dtype=np.int64)
    for v in range(V):
        af = float(Kv[v].mean())
        X = np.concatenate((np.ones((N, 1)), Kv[v].reshape(-1, 1), W), axis=1)
        sv = np.zeros(X.shape[1]); sv[0] = np.log(np.mean(y) / (1 - np.mean(y)))
        try:
            rr = smf.Logit(y, X).fit(start_params=sv, method="newton", disp=False)
            its[v] = int(rr.mle_retvals["iterations"]); conv[v] = int(bool(rr.mle_retvals["converged"]))
        except Exception as e:
            its[v] = -1
        s = fixed_effects_regression("v%d" % v, y, Kv[v], W, E0.values, af, b"x", False, None, 1.0, 1.0, nl, nf, [], [], False)
        main[v] = [nn(s.prep), nn(s.pvalue), nn(s.kbeta), nn(s.bse), nn(s.intercept)]
        b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
        betas[v] = b if b.shape == (q,) else np.nan
        notes[v] = notes_mask(s.notes); pf[v] = int(bool(s.prefilter)); fl[v] = int(bool(s.filter))
        print(v, "iterations", its[v], "converged", conv[v], sorted(s.notes), main[v], "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "n5000_cap35.npz"), N=N, q=q, y=y.astype(np.uint8), W=W, bits=pack(Kv), null_llf=nl,
                        null_firth=(np.nan if nf is None else float(nf)), main=main, betas=betas, notes=notes, prefilter=pf, filter=fl,
                        newton_iterations=its, newton_converged=conv, null_iterations=int(nl_res.mle_retvals["iterations"]))
    print("n5000_cap35: %d rows, %d at the 35-iteration cap without convergence" % (V, int(((its == 35) & (conv == 0)).sum())))


if __name__ == "__main__":
    part = sys.argv[1]
    if part == "firth_random":
        part_firth_random(sys.argv[2]); sys.exit(0)
    if part == "merge_random":
        merge_random(int(sys.argv[2])); sys.exit(0)
    if part in ("ols", "lineage", "lmm_d3", "cap35"):
        {"ols": part_ols, "lineage": part_lineage, "lmm_d3": part_lmm_d3, "cap35": part_cap35}[part](); sys.exit(0)
    if part == "lmm":
        part_lmm()
    elif part == "glm":
        part_glm()
    elif part == "firth":
        part_firth(sys.argv[2])
    elif part == "merge":
        merge_firth(int(sys.argv[2]))
