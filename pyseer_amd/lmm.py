"""LMM front-end mirroring pyseer/lmm.py: fit_lmm_block / fit_lmm over the HIP engine."""
import numpy as np

from .classes import FLAG_FILTER, FLAG_PREFILTER, notes_from_flags

NOTE_LRT = 1 << 8


def mask_like_fit_lmm(r):
    """Apply the NaN masking the reference's fit_lmm leaves in its LMM tuples (pyseer/lmm.py:160-217):
    af/pre-filtered variants keep only prep; LRT-filtered ones keep prep and pvalue."""
    fl = r["flags"]
    pre = (fl & FLAG_PREFILTER) != 0
    af = (fl & 1) != 0
    lrt = (fl & NOTE_LRT) != 0
    out = dict(r)
    for f in ("pvalue", "beta", "bse", "frac_h2"):
        a = np.array(r[f], dtype=float, copy=True)
        a[pre] = np.nan
        if f != "pvalue":
            a[lrt] = np.nan
        out[f] = a
    p = np.array(r["prep"], dtype=float, copy=True)
    p[af] = np.nan
    out["prep"] = p
    return out


# ---------------------------------------------------------------------------------------------------------------
# One-off LMM setup on the host (pyseer/lmm.py:26-122 initialise_lmm; pyseer/fastlmm/lmm_cov.py:88-104 setSU_fromK,
# :427-478 findH2; pyseer/fastlmm/mingrid.py:13-103 minimize1D).  O(N^3) once per run, never per variant.
# ---------------------------------------------------------------------------------------------------------------
def spectral_decomposition(K, covar, use_gpu=False, device=0):
    """setSU_fromK: K += I; K_ = P K P; eigh; U = U[:, D:], S = S[D:] - 1   (lmm_cov.py:88-104).

    covar: (N, D) covariates incl. the intercept (last column).  use_gpu=True runs the symmetric eigensolver through
    torch (rocSOLVER); the result feeds only per-run constants."""
    K = np.array(K, dtype=np.float64, copy=True)
    N = K.shape[0]
    D = covar.shape[1]
    K.flat[::N + 1] += 1.0
    Xd = np.linalg.pinv(covar)                              # Linreg.Xdagger, lmm_cov.py:869
    K_ = K - covar.dot(Xd.dot(K))
    K_ = K_.T
    K_ = K_ - covar.dot(Xd.dot(K_))
    if use_gpu:
        import torch
        S, U = torch.linalg.eigh(torch.from_numpy(K_).to("cuda:%d" % device))
        S = S.cpu().numpy(); U = U.cpu().numpy()
    else:
        S, U = np.linalg.eigh(K_)
    return np.ascontiguousarray(U[:, D:N]), S[D:N] - 1.0


def lmm_nll(h2, S, UY, n_dof):
    """nLLeval(h2) without SNPs (lmm_cov.py:597-684, 686-838): the objective findH2 minimises."""
    if h2 < 0.0 or h2 >= 1.0:
        return 3e20
    Sd = h2 * S + (1.0 - h2)
    yKy = float(np.sum(UY * UY / Sd))
    logdet = float(np.sum(np.log(Sd)))
    sigma2 = yKy / n_dof
    return 0.5 * (logdet + n_dof * (np.log(2.0 * np.pi * sigma2) + 1.0))


def _rotate_y(U, y, covar):
    """getUY -> rotate(Y): regress out the covariates, zero a constant residual, rotate (lmm_cov.py:165-218)."""
    a = y - covar.dot(np.linalg.pinv(covar).dot(y))
    if a.std() <= 1e-10:
        a = np.zeros_like(a)
    return U.T.dot(a)


def find_h2(U, S, y, covar, n_grid=10, min_h2=0.0, max_h2=0.99999):
    """findH2 + minimize1D (lmm_cov.py:427-478, mingrid.py:13-73): grid, fminbound at the edges, Brent on interior
    triplets.  Returns (h2, nLL) of the best evaluated point, as the reference's resmin[0] does."""
    from scipy import optimize as opt
    UY = _rotate_y(U, np.asarray(y, dtype=np.float64).reshape(-1), covar)
    n_dof = y.shape[0] - covar.shape[1]
    best = [None, np.inf]

    def f(x):
        v = lmm_nll(float(x), S, UY, n_dof)
        if best[0] is None or v < best[1]:
            best[0], best[1] = float(x), v
        return v

    step = (max_h2 - min_h2) / n_grid
    grid = np.arange(min_h2, max_h2 + step, step)           # mingrid.py:91-92
    res = np.array([f(x) for x in grid])
    if res[0] < res[1]:
        opt.fminbound(f, grid[0], grid[1], full_output=True)
    if res[-1] < res[-2]:
        opt.fminbound(f, grid[-2], grid[-1], full_output=True)
    for i in range(res.shape[0] - 2):
        if res[i + 1] < res[i + 2] and res[i + 1] < res[i]:
            opt.brent(f, brack=(grid[i], grid[i + 1], grid[i + 2]), full_output=True)
    return best[0], best[1]


def initialise_lmm_arrays(K, y, covar=None, use_gpu=False):
    """Array-level initialise_lmm (lmm.py:77-122): normalise K by N/trace, append the intercept LAST, decompose, fit h2.
    Returns (U, S, h2, nLL, covar_with_intercept)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    N = y.shape[0]
    K = np.array(K, dtype=np.float64, copy=True)
    if covar is None or np.size(covar) == 0:
        C = np.ones((N, 1))
    else:
        C = np.c_[np.asarray(covar, dtype=np.float64).reshape(N, -1), np.ones((N, 1))]
    factor = float(N) / np.diag(K).sum()
    if abs(factor - 1.0) > 1e-15:
        K *= factor
    U, S = spectral_decomposition(K, C, use_gpu=use_gpu)
    h2, nll = find_h2(U, S, y, C)
    return U, S, h2, nll, C


class LmmState(object):
    """What the per-variant path needs from the reference's fastlmm LMM object: U, S, Y (phenotype column), X (covariates
    with the intercept LAST).  Attribute names follow pyseer/fastlmm/lmm_cov.py so cache files and call sites carry over."""
    __slots__ = ("U", "S", "Y", "X")

    def __init__(self, U, S, Y, X):
        self.U, self.S, self.Y, self.X = U, S, Y, X


def initialise_lmm(p, cov, K_in, lmm_cache_in=None, lmm_cache_out=None, lineage_samples=None, use_gpu=False, device=0):
    """pyseer/lmm.py:26-122: returns (p restricted to the samples of the similarity matrix, LmmState, h2).
    Reads / writes the reference's cache layout (np.savez: arr_0 = U, arr_1 = S, arr_2 = [h2])."""
    import os
    import sys
    import math
    import pandas as pd

    def covariates_for(pp):
        if len(pp.index.intersection(cov.index)) == pp.shape[0]:
            return np.c_[cov.loc[pp.index].values, np.ones((pp.shape[0], 1))]
        if (cov.shape[0] == 0 and cov.shape[1] == 0) or len(cov.shape) == 0:
            return np.ones((pp.shape[0], 1))
        sys.stderr.write("Phenotype and covariate file should have matching samples for LMM\n")
        sys.exit(1)

    if lmm_cache_in is not None and os.path.exists(lmm_cache_in):
        covar = covariates_for(p)
        with np.load(lmm_cache_in) as data:
            U, S, h2 = data['arr_0'], data['arr_1'], float(data['arr_2'][0])
        if U.shape[0] != len(p):
            sys.stderr.write("Phenotype different length from cache file\n")
            sys.exit(1)
        return p, LmmState(np.ascontiguousarray(U), S, p.values.astype(float), covar), h2

    K = pd.read_csv(K_in, index_col=0, sep='\t')
    K.index = K.index.astype(str)
    sys.stderr.write("Similarity matrix has dimension " + str(K.shape) + "\n")
    if lineage_samples is not None and set(K.index) != set(lineage_samples):
        sys.stderr.write("Lineage file and similarity matrix contain different sets of samples\n")
        sys.exit(1)
    inter = p.index.intersection(K.index)
    sys.stderr.write("Analysing " + str(len(inter)) + " samples found in both phenotype and similarity matrix\n")
    p = p.loc[inter]
    K = K.loc[p.index, p.index]
    covar = covariates_for(p)
    factor = float(len(p)) / np.diag(K.values).sum()
    if factor == math.inf:
        sys.stderr.write("Invalid similarity matrix. Did you use --calc-C?\n")
        sys.exit(1)
    Kv = K.values.astype(float)
    if abs(factor - 1.0) > 1e-15:
        Kv = Kv * factor
    U, S = spectral_decomposition(Kv, covar, use_gpu=use_gpu, device=device)
    h2, _ = find_h2(U, S, p.values.astype(float), covar)
    if lmm_cache_out is not None and not os.path.exists(lmm_cache_out):
        np.savez(lmm_cache_out, U, S, np.array([h2]))
    return p, LmmState(U, S, p.values.astype(float), covar), h2


# ---------------------------------------------------------------------------------------------------------------
# Drop-ins with the reference's signatures (pyseer/lmm.py:125-260).  The batched driver in pyseer_amd/__main__.py does not go
# through them (it keeps results as arrays); they exist so that code written against pyseer.lmm keeps working unchanged.
# ---------------------------------------------------------------------------------------------------------------
_engines = {}


def _engine_for(lmm, h2, continuous, filter_pvalue, lrt_pvalue):
    from .engine import Engine
    Y = np.asarray(lmm.Y, dtype=float).reshape(-1)
    key = (id(lmm), float(h2), bool(continuous), float(filter_pvalue), float(lrt_pvalue))
    eng = _engines.get(key)
    if eng is None:
        for old in list(_engines.values()):
            old.close()
        _engines.clear()
        eng = Engine(Y.shape[0])
        eng.lmm_setup(np.asarray(lmm.U, dtype=float), np.asarray(lmm.S, dtype=float), Y, np.asarray(lmm.X, dtype=float), h2,
                      continuous, filter_pvalue, lrt_pvalue)          # raises KeyError('beta') for h2 outside [0, 1)
        _engines[key] = eng
    return eng


def fit_lmm_block(lmm, h2, variant_block):
    """pyseer/lmm.py:228-260: variant_block (n, B) of 0/1 -> {'p_values', 'beta', 'bse', 'frac_h2'}, arrays of length B.
    KeyError('beta') for h2 outside [0, 1), AssertionError on a sample-count mismatch (tests/lmm_test.py:416-420)."""
    from .packing import pack_variants
    variant_block = np.asarray(variant_block, dtype=float)
    n = np.asarray(lmm.Y).reshape(-1).shape[0]
    eng = _engine_for(lmm, h2, False, np.inf, np.inf)
    assert variant_block.ndim == 2 and variant_block.shape[0] == n, "shape missmatch between snps and Y"
    if np.isnan(variant_block).any():
        raise ValueError("missing calls cannot be packed; the reference propagates NaN here")
    r = eng.lmm_batch(pack_variants(variant_block.T))
    return {'p_values': r["pvalue"], 'beta': r["beta"], 'bse': r["bse"], 'frac_h2': r["frac_h2"]}


def fit_lmm(lmm, h2, variants, variant_mat, lineage_effects, lineage_clusters, covariates, continuous, filter_pvalue,
            lrt_pvalue):
    """pyseer/lmm.py:125-226: LMM tuples for one block, af/pre-filtered variants first, then the tested ones, each in input
    order; filtered columns of variant_mat are zeroed in place as the reference does."""
    from .packing import pack_variants
    from .model import host_pre_filtering
    eng = _engine_for(lmm, h2, continuous, filter_pvalue, lrt_pvalue)
    first, pending = [], []                                  # (input index, tuple): the reference's first and second loop
    rows, where = [], []
    for var_idx, (var, p, k) in enumerate(variants):
        if var.pattern is None or k is None:
            first.append((var_idx, var._replace(notes={'af-filter'}, prefilter=True, filter=False)))
            variant_mat[:, var_idx] = 0.0
            continue
        kf = np.asarray(k, dtype=float)
        if np.isnan(kf).any():                               # missing calls: pre_filtering on the host, NaN through the block fit
            prep, bad = host_pre_filtering(np.asarray(p, dtype=float), kf, continuous)
            notes = {'bad-chisq'} if bad else set()
            if prep >= filter_pvalue or not np.isfinite(prep):
                notes.add('pre-filtering-failed')
                first.append((var_idx, var._replace(notes=notes, prep=prep, prefilter=True, filter=False)))
                variant_mat[:, var_idx] = 0.0
            else:
                notes.add('lrt-filtering-failed')
                pending.append((var_idx, var._replace(prep=prep, notes=notes, prefilter=False, pvalue=np.nan, filter=True)))
            continue
        rows.append(kf); where.append(var_idx)
    if rows:
        r = mask_like_fit_lmm(eng.lmm_batch(pack_variants(np.asarray(rows))))
        stale_k = variants[-1][2]                            # lmm.py:209-213 reads the loop variable of the first loop
        ml = None
        if lineage_effects:
            eng.lineage_setup(np.asarray(lineage_clusters, dtype=float),
                              covariates if (covariates is not None and np.size(covariates)) else None)
            if stale_k is not None and not np.isnan(np.asarray(stale_k, dtype=float)).any():
                v = int(eng.lineage_batch(pack_variants(np.asarray(stale_k, dtype=float).reshape(1, -1)))[0])
                ml = None if v < 0 else v
        for j, var_idx in enumerate(where):
            var = variants[var_idx][0]
            fl = int(r["flags"][j])
            notes = notes_from_flags(fl)
            if fl & FLAG_PREFILTER:
                first.append((var_idx, var._replace(notes=notes, prep=float(r["prep"][j]), prefilter=True, filter=False)))
                variant_mat[:, var_idx] = 0.0
            elif fl & FLAG_FILTER:
                pending.append((var_idx, var._replace(prep=float(r["prep"][j]), notes=notes, prefilter=False,
                                                      pvalue=float(r["pvalue"][j]), filter=True)))
            else:
                pending.append((var_idx, var._replace(prep=float(r["prep"][j]), pvalue=float(r["pvalue"][j]),
                                                      kbeta=float(r["beta"][j]), bse=float(r["bse"][j]),
                                                      frac_h2=float(r["frac_h2"][j]), notes=notes, prefilter=False, filter=False,
                                                      max_lineage=ml)))
    return [v for _, v in sorted(first, key=lambda t: t[0])] + [v for _, v in sorted(pending, key=lambda t: t[0])]
