"""Fixed-effects front-end mirroring pyseer/model.py over the HIP engine.

Host side = the once-per-run null fits (fit_null, model.py:73-148) and the Seer materialisation; every per-variant
fit runs in libseerhip (k_glm_logit / k_glm_firth / k_glm_ols).
"""
import sys

import numpy as np

from .classes import Seer, FLAG_FILTER, FLAG_PREFILTER, notes_from_flags
from .packing import pack_variants


# ---------------------------------------------------------------------------------------------------------------
# Null models (host, once per run)
# ---------------------------------------------------------------------------------------------------------------
def _cdf(x):
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-x))


def _loglike(X, y, b):
    with np.errstate(divide="ignore"):
        return float(np.sum(np.log(_cdf((2.0 * y - 1.0) * X.dot(b)))))


def _info(X, b):
    L = _cdf(X.dot(b))
    return (X.T * (L * (1.0 - L))).dot(X)


def logit_newton(X, y, start, maxiter=35, tol=1e-8, ridge=1e-10):
    """statsmodels Logit.fit(method='newton') semantics (SM:base/optimizer.py:407-446): returns (beta, llf) or raises
    np.linalg.LinAlgError / PerfectSeparation."""
    n = X.shape[0]
    new = np.asarray(start, dtype=float).copy()
    old = np.full_like(new, np.inf)
    it = 0
    while it < maxiter and np.any(np.abs(new - old) > tol):
        L = _cdf(X.dot(new))
        H = -(X.T * (L * (1.0 - L))).dot(X) / n
        H[np.diag_indices(H.shape[0])] += ridge
        old = new
        new = old - np.linalg.inv(H).dot(X.T.dot(y - L) / n)
        if np.allclose(_cdf(X.dot(new)) - y, 0):
            raise PerfectSeparation()
        it += 1
    np.linalg.inv(_info(X, new) / n)         # the results object inverts the final Hessian (SM:base/model.py:534)
    return new, _loglike(X, y, new)


class PerfectSeparation(Exception):
    pass


def firth_likelihood(beta, X, y):
    """model.py:397-411"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return -(_loglike(X, y, beta) + 0.5 * np.log(np.linalg.det(_info(X, beta))))


def fit_firth_host(X, y, start, step_limit=1000, convergence_limit=1e-4):
    """model.py:414-504 on the host (used only for the null model); returns (beta, bse, fitll) or None."""
    bs = [np.asarray(start, dtype=float)]
    i = 0
    for i in range(step_limit):
        b = bs[i]
        pi = _cdf(X.dot(b))
        w = pi * (1.0 - pi)
        V = np.linalg.pinv(_info(X, b))
        h = w * np.einsum("ij,jk,ik->i", X, V, X)
        U = X.T.dot(y - pi + h * (0.5 - pi))
        nb = b + V.dot(U)
        j = 0
        fb = firth_likelihood(b, X, y)
        while firth_likelihood(nb, X, y) > fb:
            nb = b + 0.5 * (nb - b)
            j += 1
            if j > step_limit:
                return None
        bs.append(nb)
        if i > 0 and np.linalg.norm(bs[i] - bs[i - 1]) < convergence_limit:
            break
    if np.linalg.norm(bs[i] - bs[i - 1]) >= convergence_limit:
        return None
    bl = bs[-1]
    bse = float(np.sqrt(_info(X, bl)[1, 1])) if bl.shape[0] > 1 else None
    return bl, bse, -firth_likelihood(bl, X, y)


class NullFit(object):
    """What the driver needs from statsmodels' results object: .llf (pyseer/__main__.py:449-450)."""

    def __init__(self, llf, params, bse=None):
        self.llf = llf
        self.params = params
        self.bse = bse


def null_design(p, m, cov):
    p = np.asarray(p, dtype=float).reshape(-1)
    v = np.ones((p.shape[0], 1))
    m = np.asarray(m, dtype=float)
    if m.ndim == 2 and m.shape[0] == p.shape[0] and m.shape[1] > 0:
        v = np.concatenate((v, m), axis=1)
    c = np.asarray(getattr(cov, "values", cov), dtype=float)
    if c.ndim == 2 and c.shape[0] == p.shape[0] and c.shape[1] > 0:
        v = np.concatenate((v, c), axis=1)
    return p, v


def fit_null(p, m, cov, continuous, firth=False):
    """model.py:73-148.  Returns a NullFit (or the Firth log-likelihood when firth=True), None on failure."""
    p, v = null_design(p, m, cov)
    if continuous:
        beta, res, rank, sv = np.linalg.lstsq(v, p, rcond=None)
        ssr = float(np.sum((p - v.dot(beta)) ** 2)); n = p.shape[0]
        llf = -0.5 * n * (np.log(2 * np.pi) + np.log(ssr / n) + 1.0)
        bse = np.sqrt(np.diag(np.linalg.pinv(v.T.dot(v))) * ssr / (n - rank))
        return NullFit(llf, beta, bse)
    start = np.zeros(v.shape[1])
    start[0] = np.log(np.mean(p) / (1 - np.mean(p)))
    if firth:
        r = fit_firth_host(v, p, start)
        if r is None:
            sys.stderr.write('Firth regression did not converge for null model\n')
            return None
        return r[2]
    try:
        try:
            beta, llf = logit_newton(v, p, start)
        except np.linalg.LinAlgError:
            # model.py:134-137: "Null fit with default optimiser may fail, Powell optimizer might work".  statsmodels'
            # _fit_powell = scipy fmin_powell on -loglike/nobs with xtol = ftol = 1e-4 and fit()'s maxiter = 35
            # (SM:base/optimizer.py:631-665); the separation callback stays armed; the covariance is then taken from an
            # eigendecomposition and silently dropped when the Hessian is not positive definite (SM:base/model.py:543-556).
            from scipy import optimize
            n = v.shape[0]

            def callback(b):
                if np.allclose(_cdf(v.dot(b)) - p, 0):
                    raise PerfectSeparation()

            beta = optimize.fmin_powell(lambda b: -_loglike(v, p, b) / n, start, xtol=1e-4, ftol=1e-4, maxiter=35,
                                        full_output=1, disp=0, callback=callback)[0]
            beta = np.atleast_1d(np.asarray(beta, dtype=float))
            H = _info(v, beta)
            bse = np.full(beta.shape[0], np.nan)
            if np.all(np.isfinite(H)):
                w, _ = np.linalg.eigh(H)
                if np.min(w) > 0:
                    bse = np.sqrt(np.diag(np.linalg.inv(H)))
            return NullFit(_loglike(v, p, beta), beta, bse)
    except np.linalg.LinAlgError:
        sys.stderr.write('Matrix inversion error for null model\n')
        return None
    except PerfectSeparation:
        sys.stderr.write('Perfectly separable data error for null model\n')
        return None
    return NullFit(llf, beta, np.sqrt(np.diag(np.linalg.inv(_info(v, beta)))))


def host_pre_filtering(p, k, continuous):
    """pre_filtering (model.py:31-70) for the rare variants that carry missing calls and therefore never reach the GPU."""
    from scipy import stats
    if continuous:
        return stats.ttest_ind(p[k == 1], p[k == 0], equal_var=False)[1], False
    table = np.array([[np.sum((p == 1) & (k == 1)), np.sum((p == 1) & (k == 0))],
                      [np.sum((p == 0) & (k == 1)), np.sum((p == 0) & (k == 0))]])
    bad = bool((table <= 1).sum() > 0 or (table <= 5).sum() > 1)
    try:
        prep = stats.chi2_contingency(table, correction=False)[1]
    except ValueError:
        prep = np.nan
    return prep, bad


# ---------------------------------------------------------------------------------------------------------------
# Per-variant results
# ---------------------------------------------------------------------------------------------------------------
def covariate_block(n, m, c):
    """[m | c] exactly as the design is assembled at model.py:274-297 (either block may be absent)."""
    cols = []
    m = np.asarray(m, dtype=float)
    if m.ndim == 2 and m.shape[0] == n and m.shape[1] > 0:
        cols.append(m)
    c = np.asarray(getattr(c, "values", c), dtype=float)
    if c.ndim == 2 and c.shape[0] == n and c.shape[1] > 0:
        cols.append(c)
    if not cols:
        return np.zeros((n, 0))
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def seer_from_row(r, i, variant, pattern, af, kstrains, nkstrains, max_lineage=None):
    fl = int(r["flags"][i])
    notes = notes_from_flags(fl)
    tested = np.isfinite(r["kbeta"][i]) or np.isfinite(r["pvalue"][i])
    betas = np.array(r["betas"][i], dtype=float) if (tested and r["betas"].shape[1]) else np.array([])
    if tested and not r["betas"].shape[1]:
        betas = np.array([])
    return Seer(variant, pattern, af, float(r["prep"][i]), float(r["pvalue"][i]), float(r["kbeta"][i]),
                float(r["bse"][i]), float(r["intercept"][i]), betas, max_lineage, kstrains, nkstrains, notes,
                bool(fl & FLAG_PREFILTER), bool(fl & FLAG_FILTER))


class FixedEffects(object):
    """Batch driver: holds one Engine configured for (p, m, c, null fits, thresholds)."""

    def __init__(self, p, m, c, continuous, pret, lrtt, null_res, null_firth, device=0, force_firth=False):
        from .engine import Engine
        p = np.asarray(p, dtype=float).reshape(-1)
        self.n = p.shape[0]
        self.W = covariate_block(self.n, m, c)
        self.continuous = bool(continuous)
        self.engine = Engine(self.n, device=device)
        null_llf = float(getattr(null_res, "llf", null_res if null_res is not None else np.nan)) if not continuous else np.nan
        self.engine.glm_setup(p, self.W, continuous, null_llf, null_firth, pret, lrtt, force_firth=force_firth)

    def batch(self, bits):
        return self.engine.glm_batch(bits)

    def batch_from_dense(self, K):
        return self.batch(pack_variants(K))


_cache = {}


def fixed_effects_regression(variant, p, k, m, c, af, pattern, lineage_effects, lin, pret, lrtt, null_res, null_firth,
                             kstrains, nkstrains, continuous):
    """Drop-in for pyseer.model.fixed_effects_regression (model.py:202): same arguments, returns a Seer.
    One variant per call is the reference's interface; the engine is cached across calls with the same setup.
    (The batched entry point FixedEffects.batch is what the driver in pyseer_amd/__main__.py uses.)"""
    if p is None:                                                         # model.py:255-260
        return Seer(variant, pattern, af, np.nan, np.nan, np.nan, np.nan, np.nan, np.array([]), None,
                    kstrains, nkstrains, {'af-filter'}, True, False)
    k = np.asarray(k)
    if np.isnan(k.astype(float)).any():
        # missing calls never reach the GPU.  The reference prefilters first (model.py:262-271: NaN falls in no cell of the 2x2
        # table / in neither Welch group) and only then meets statsmodels' MissingDataError (model.py:371-377)
        pv = np.asarray(getattr(p, "values", p), dtype=float).reshape(-1)
        prep, bad = host_pre_filtering(pv, k.astype(float).reshape(-1), continuous)
        notes = {'bad-chisq'} if bad else set()
        if prep > pret or not np.isfinite(prep):
            notes.add('pre-filtering-failed')
            return Seer(variant, pattern, af, prep, np.nan, np.nan, np.nan, np.nan, np.array([]), None, kstrains, nkstrains,
                        notes, True, False)
        notes.add('missing-data-error')
        return Seer(variant, pattern, af, prep, np.nan, np.nan, np.nan, np.nan, np.array([]), None, kstrains, nkstrains,
                    notes, False, True)
    key = (id(p), id(m), id(c), bool(continuous), float(pret), float(lrtt), id(null_res), repr(null_firth))
    fe = _cache.get(key)
    if fe is None:
        _cache.clear()
        fe = FixedEffects(p, m, c, continuous, pret, lrtt, null_res, null_firth)
        _cache[key] = fe
    bits = pack_variants(k.reshape(1, -1))
    r = fe.batch(bits)
    s = seer_from_row(r, 0, variant, pattern, af, kstrains, nkstrains)
    if lineage_effects and not s.prefilter and 'firth-fail' not in s.notes:           # model.py:379-382 -> fit_lineage_effect
        lkey = (id(lin), id(c))
        if getattr(fe, "_lineage_key", None) != lkey:
            cv = np.asarray(getattr(c, "values", c), dtype=float)
            fe.engine.lineage_setup(np.asarray(lin, dtype=float), cv if (cv.ndim == 2 and cv.shape[0] == fe.n and cv.shape[1] > 0) else None)
            fe._lineage_key = lkey
        ml = int(fe.engine.lineage_batch(bits)[0])
        s = s._replace(max_lineage=None if ml < 0 else ml)
    return s


# ---------------------------------------------------------------------------------------------------------------
# The remaining names of pyseer/model.py's per-variant surface, with the reference's signatures
# ---------------------------------------------------------------------------------------------------------------
_prefilter_engine = {}


def pre_filtering(p, k, continuous):
    """pyseer/model.py:31-70 -> (prep, bad_chisq).  Same kernel code as every batch (the 2x2 table / Welch sums from the packed bits,
    csrc/common.h): the variant goes through the fixed-effects entry point with a filter threshold below any p-value, so the engine
    classifies it as pre-filtered and returns prep and the bad-chisq note without fitting anything.  A variant with missing calls never
    reaches the engine (host_pre_filtering, as in the driver)."""
    from .engine import Engine
    p = np.ascontiguousarray(np.asarray(getattr(p, "values", p), dtype=float).reshape(-1))
    k = np.asarray(k, dtype=float).reshape(-1)
    if np.isnan(k).any() or np.isnan(p).any():
        return host_pre_filtering(p, k, continuous)
    key = (p.shape[0], bool(continuous), p.tobytes())
    e = _prefilter_engine.get(key)
    if e is None:
        _prefilter_engine.clear()
        e = Engine(p.shape[0])
        e.glm_setup(p, None, bool(continuous), 0.0, None, pret=-1.0, lrtt=1.0)
        _prefilter_engine[key] = e
    r = e.glm_batch(pack_variants(k.reshape(1, -1)))
    return float(r["prep"][0]), bool(r["flags"][0] & 4)


_lineage_engine = {}


def fit_lineage_effect(lin, c, k):
    """pyseer/model.py:151-199: index of the lineage most associated with the variant (largest |beta|/bse in the logistic fit of
    k on [1, lin, c]), or None when that fit fails.  Runs k_glm_lineage for the one variant."""
    from .engine import Engine
    lin = np.asarray(lin, dtype=float)
    n = lin.shape[0]
    cv = np.asarray(getattr(c, "values", c), dtype=float) if c is not None else np.zeros((n, 0))
    key = (id(lin), id(c), n)
    eng = _lineage_engine.get(key)
    if eng is None:
        for old in list(_lineage_engine.values()):
            old.close()
        _lineage_engine.clear()
        eng = Engine(n)
        eng.lineage_setup(lin, cv if (cv.ndim == 2 and cv.shape[0] == n and cv.shape[1] > 0) else None)
        _lineage_engine[key] = eng
    ml = int(eng.lineage_batch(pack_variants(np.asarray(k, dtype=float).reshape(1, -1)))[0])
    return None if ml < 0 else ml
