#!/opt/conda/bin/python3.9
"""Reference-generated goldens for the Firth-fallback exits the reference's own tests leave unpinned
(/root/reference/tests/model_test.py:382-386):

    high-bse                  pyseer/model.py:332-335   Newton converges, bse[1] > 3           -> fit_firth
    perfectly-separable-data  pyseer/model.py:345-347   statsmodels PerfectSeparationError      -> fit_firth
    firth-fail                pyseer/model.py:355-362   fit_firth returns None (:464-474, :482-484)

The script SEARCHES seeded designs, calls the imported `fixed_effects_regression` on every candidate row and keeps the rows
for which the REFERENCE ITSELF emits the wanted note (plus a few ordinary rows of the same design as controls).  Runs only in
the build container:

    cd /root/repo/tests/golden && \
    PYTHONPATH=_harness:_harness/stubs:/root/reference PYTHONDONTWRITEBYTECODE=1 \
    OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 \
    /opt/conda/bin/python3.9 -W ignore make_exit_golden.py

Output: glm_exit_*.npz with the fields of make_golden.run_fixed_case (so every test that globs glm_*.npz picks them up) plus
    family          which candidate generator made the design
    perm_main/perm_betas/perm_notes   for rows the reference reports as `firth-fail`: what the reference answers for the SAME
                    row with the samples in another order (three fixed permutations of (y, k, m) together).  A failure that
                    disappears under a permutation is the rounding-noise failure of DESIGN.md section 6 (every halved candidate
                    compares higher than a beta whose F is a low outlier of its own summation noise); the statistics of the
                    permuted fit are then the row's legitimate answer and are what another summation order is held to.
Only data is written (designs we synthesise, numbers the reference returns)."""
import shim  # noqa: F401  must be first
import os
import sys
import time
import warnings
import numpy as np
import pandas as pd

warnings.simplefilter("ignore")
import statsmodels.formula.api as smf
from pyseer.model import fit_null, fit_firth, fixed_effects_regression

OUT = os.path.dirname(os.path.abspath(__file__))
NOTE_ORDER = ["af-filter", "pre-filtering-failed", "bad-chisq", "high-bse", "perfectly-separable-data", "matrix-inversion-error",
              "firth-fail", "missing-data-error", "lrt-filtering-failed"]
BIT = {n: 1 << i for i, n in enumerate(NOTE_ORDER)}
E0 = pd.DataFrame([])


def notes_mask(notes):
    m = 0
    for n in notes:
        m |= BIT[n]
    return m


def nn(x):
    return np.nan if x is None else float(x)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def ref_row(y, k, m, null_llf, null_firth, q):
    """One call of the reference (model.py:202); returns (main[5], betas[q], notes, prefilter, filter)."""
    af = float(k.mean())
    p_arg = y if (0.01 <= af <= 0.99) else None                    # iter_variants' AF filter (input.py:608)
    s = fixed_effects_regression("v", p_arg, k, m if q else np.zeros((0, 0)), E0.values, af, b"x", False, None, 1.0, 1.0,
                                 null_llf, null_firth, [], [], False)
    betas = np.full(q, np.nan)
    b = np.asarray(s.betas, dtype=float) if s.betas is not None else np.array([])
    if b.ndim == 1 and b.shape[0] == q and q > 0:
        betas[:] = b
    return np.array([nn(s.prep), nn(s.pvalue), nn(s.kbeta), nn(s.bse), nn(s.intercept)]), betas, notes_mask(s.notes), \
        int(bool(s.prefilter)), int(bool(s.filter))


def ref_forced_firth(y, k, m, q):
    """fit_firth on the row's design directly (model.py:414), as make_golden.run_fixed_case does for BASELINE config C4."""
    N = y.shape[0]
    X = np.concatenate((np.ones((N, 1)), k.reshape(-1, 1), m), axis=1) if q else np.concatenate((np.ones((N, 1)), k.reshape(-1, 1)), axis=1)
    sv = np.zeros(X.shape[1]); sv[0] = np.log(np.mean(y) / (1 - np.mean(y)))
    af = float(k.mean())
    if not (0.01 <= af <= 0.99):
        return None, 0
    try:
        res = fit_firth(smf.Logit(y, X), sv, X, y)
    except Exception:
        return None, -1
    return res, (1 if res is not None else 0)


def nulls(y, m, q):
    mm = m if q else np.zeros((y.shape[0], 0))
    r = fit_null(y, mm, E0, False, firth=False)
    nf = fit_null(y, mm, E0, False, firth=True)
    return (None if r is None else float(r.llf)), (None if nf is None else float(nf))


# ------------------------------------------------------------------------------------------------------------------
# candidate designs.  Each returns (y, m) for a seed and a generator of candidate rows k.
# ------------------------------------------------------------------------------------------------------------------
def design_strong_covariate(rng, N, q, slope):
    """high-bse: one covariate predicts the phenotype almost deterministically; a k-mer whose carriers all sit at the extremes of it has a
    healthy 2x2 table but carries no information (weights mu(1-mu) ~ 1e-5) -> Newton converges with bse[1] >> 3."""
    m = rng.standard_normal((N, q)); m /= np.abs(m).max(axis=0)
    y = (rng.random(N) < sigmoid(slope * m[:, 0])).astype(float)
    order = np.argsort(m[:, 0])
    lo = [i for i in order if y[i] == 0][: max(12, N // 12)]        # most negative covariate, phenotype 0
    hi = [i for i in order[::-1] if y[i] == 1][: max(12, N // 12)]  # most positive, phenotype 1

    def rows():
        while True:
            k = np.zeros(N)
            k[rng.choice(hi, rng.integers(2, 9), replace=False)] = 1
            k[rng.choice(lo, rng.integers(2, 9), replace=False)] = 1
            r = rng.integers(0, 14)
            if r:                                                    # carriers anywhere: pull bse down to and below the threshold of 3
                k[rng.choice(N, r, replace=False)] = 1
            yield k
    return y, m, rows()


def design_strong_effect(rng, N, q, slope):
    """high-bse, the route of a strong-effect k-mer: nearly every carrier has the phenotype; the two or three that do not are samples the
    covariates already put at mu ~ 0, so the coefficient of k can run away without a PerfectSeparationError."""
    m = rng.standard_normal((N, q)); m /= np.abs(m).max(axis=0)
    y = (rng.random(N) < sigmoid(slope * m[:, 0] + 0.2)).astype(float)
    order = np.argsort(m[:, 0])
    lo = [i for i in order if y[i] == 0][: max(8, N // 20)]
    ones = np.where(y == 1)[0]

    def rows():
        while True:
            k = np.zeros(N)
            k[rng.choice(ones, rng.integers(6, max(8, N // 6)), replace=False)] = 1
            k[rng.choice(lo, rng.integers(2, 4), replace=False)] = 1
            yield k
    return y, m, rows()


def design_lattice_separable(rng, N, q, zone_p):
    """perfectly-separable-data: the first covariate takes the six values +-0.2, +-0.6, +-1; outside the +-0.2 band the phenotype is its
    sign, inside the band it is a coin (so the null model is NOT separable: 1s at -0.2 and 0s at +0.2 interleave).  A k-mer that
    equals the phenotype inside the band and is anything outside separates the data together with the covariate, with a healthy 2x2
    table and MANY samples on the minimal margin -- X'WX/n stays far above statsmodels' 1e-10 ridge when every fitted value is
    within 1e-8 of its label, so the callback raises before the Hessian degenerates (a generic continuous design ends in
    `matrix-inversion-error` instead: make_golden.py glm_N300_q2_sep)."""
    lev = np.array([-1.0, -0.6, -0.2, 0.2, 0.6, 1.0])
    m = rng.standard_normal((N, q)); m /= np.abs(m).max(axis=0)
    m[:, 0] = lev[rng.integers(0, 6, N)]
    band = np.abs(m[:, 0]) < 0.3
    y = (m[:, 0] > 0).astype(float)
    y[band] = (rng.random(band.sum()) < zone_p).astype(float)

    def rows():
        while True:
            k = (rng.random(N) < rng.uniform(0.15, 0.85)).astype(float)
            k[band] = y[band]
            if rng.random() < 0.3:                                   # one or two flips inside the band: no longer separable, large bse
                f = rng.choice(np.where(band)[0], rng.integers(1, 3), replace=False); k[f] = 1 - k[f]
            yield k
    return y, m, rows()


def design_rare(rng, N, q, slope):
    """firth-fail: ordinary designs, rare k-mers (a cell <= 1 -> bad-chisq -> fit_firth).  The failures found are the reference's
    rounding-noise ones; `perm_*` records what it answers for the same row in another sample order."""
    m = rng.standard_normal((N, q)); m = m / np.abs(m).max(axis=0) if q else m
    eta = -0.3 + (slope * m[:, 0] if q else 0.0) - (m[:, 1] if q > 1 else 0.0)
    y = (rng.random(N) < sigmoid(eta)).astype(float)

    def rows():
        while True:
            c = rng.integers(max(2, int(0.011 * N) + 1), max(4, int(0.05 * N)))
            k = np.zeros(N); k[rng.choice(N, c, replace=False)] = 1
            yield k
    return y, m, rows()


def design_rare_screened(rng, N, q, slope):
    """firth-fail, looked for where it can be found: rows that are bad-chisq BY CONSTRUCTION (all carriers but at most one in one phenotype
    class: a 2x2 cell <= 1, so every one of them goes through fit_firth), pre-screened by the C restatement (oracle/seer_oracle.c, which sums
    the log-likelihood in numpy's order) -- only the rows IT calls firth-fail, and a control per batch, are handed to the reference, which
    has the last word on every row kept."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import oracle as orc
    m = rng.standard_normal((N, q)); m = m / np.abs(m).max(axis=0) if q else m
    eta = -0.3 + (slope * m[:, 0] if q else 0.0) - (m[:, 1] if q > 1 else 0.0)
    y = (rng.random(N) < sigmoid(eta)).astype(float)
    nl, nf = nulls(y, m, q)
    cls = [np.where(y == 0)[0], np.where(y == 1)[0]]
    lo, hi = max(2, int(0.011 * N) + 1), max(4, int(0.05 * N))

    def rows():
        screened = 0
        while True:
            B = 20000
            K = np.zeros((B, N))
            for b in range(B):
                c = int(rng.integers(lo, hi)); a = int(rng.integers(0, 2)); other = int(rng.random() < 0.5)
                K[b, rng.choice(cls[a], min(c - other, cls[a].size), replace=False)] = 1
                if other:
                    K[b, rng.choice(cls[1 - a], 1)] = 1
            res = orc.fixed_effects_batch(y, K, m if q else None, False, 1.0, 1.0, nl, nf)
            flagged = np.where(res["notes"] & BIT["firth-fail"])[0]
            screened += B
            print("    screened %d rows, the restatement flags %d of this batch" % (screened, flagged.size), flush=True)
            for b in flagged:
                yield K[b]
            yield K[0]
    return y, m, rows()


FAMILIES = {"rarescreened": design_rare_screened, "strongcov": design_strong_covariate, "strongeff": design_strong_effect, "lattice": design_lattice_separable,
            "rare": design_rare}


def search(name, family, seed, N, q, par, want, n_want, max_tries, n_ctrl=6, budget_s=600.0):
    rng = np.random.default_rng(seed)
    y, m, gen = FAMILIES[family](rng, N, q, par)
    nl, nf = nulls(y, m, q)
    if nl is None or nf is None:
        print(name, "null model did not fit (llf %s, firth %s): design skipped" % (nl, nf)); return None
    kept, seen, hist = [], set(), {}
    t0 = time.time(); tries = 0; n_hit = 0; n_c = 0
    while tries < max_tries and (n_hit < n_want or n_c < n_ctrl) and time.time() - t0 < budget_s:
        k = next(gen); tries += 1
        key = hash(k.tobytes())
        if key in seen:
            continue
        seen.add(key)
        main, betas, nm, pf, fl = ref_row(y, k, m, nl, nf, q)
        for i, n_ in enumerate(NOTE_ORDER):
            if nm >> i & 1:
                hist[n_] = hist.get(n_, 0) + 1
        hit = bool(nm & BIT[want])
        if hit or tries % 20000 == 0:
            print("    [%s] candidate %d after %.0f s: notes %d%s" % (name, tries, time.time() - t0, nm, "  <-- " + want if hit else ""), flush=True)
        if (hit and n_hit < n_want) or (not hit and n_c < n_ctrl and tries % 2 == 0):
            kept.append((k, main, betas, nm, pf, fl)); n_hit += hit; n_c += (not hit)
    print("%-34s %6d candidates in %5.1f s: %d x %s, %d controls; notes seen %s" % (name, tries, time.time() - t0, n_hit, want, n_c, hist))
    if n_hit == 0:
        return None
    V = len(kept)
    K = np.array([r[0] for r in kept]); main = np.array([r[1] for r in kept]); betas = np.array([r[2] for r in kept]).reshape(V, q)
    notes = np.array([r[3] for r in kept], dtype=np.int64)
    out = dict(N=N, q=q, V=V, continuous=0, pret=1.0, lrtt=1.0, y=y, m=m, K=K.astype(np.uint8), null_llf=nl, null_firth=nf, main=main,
               betas=betas, notes=notes, prefilter=np.array([r[4] for r in kept], dtype=np.int64),
               filter=np.array([r[5] for r in kept], dtype=np.int64), family=family, seed=seed)
    # forced Firth on every kept row (BASELINE config C4's mode), as make_golden.run_fixed_case
    fm = np.full((V, 4), np.nan); fb = np.full((V, q), np.nan); fok = np.zeros(V, dtype=np.int64)
    for v in range(V):
        res, ok = ref_forced_firth(y, K[v], m, q)
        fok[v] = ok
        if res is not None:
            ic, kb, be, bse, fitll = res
            fm[v] = [ic, kb, bse, fitll]
            if q:
                fb[v] = np.asarray(be, dtype=float)
    out.update(firth_main=fm, firth_betas=fb, firth_ok=fok)
    # the reference's answer for the firth-fail rows with the samples in three other orders
    ff = np.where(notes & BIT["firth-fail"])[0]
    if ff.size:
        prng = np.random.default_rng(seed + 77)
        perms = np.array([prng.permutation(N) for _ in range(3)])
        pm = np.full((V, 3, 5), np.nan); pb = np.full((V, 3, q), np.nan); pn = np.full((V, 3), -1, dtype=np.int64)
        for v in ff:
            for j, pi in enumerate(perms):
                a, b, nm, _, _ = ref_row(y[pi], K[v][pi], m[pi] if q else m, nl, nf, q)
                pm[v, j] = a; pb[v, j] = b; pn[v, j] = nm
        out.update(perms=perms, perm_main=pm, perm_betas=pb, perm_notes=pn)
        print("    firth-fail rows:", ff.tolist(), "notes under the three permutations:", pn[ff].tolist())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return out


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    jobs = [
        # name, family, seed, N, q, parameter, wanted note, rows wanted, candidate cap
        ("glm_exit_highbse_N100_q1", "strongcov", 4101, 100, 1, 14.0, "high-bse", 8, 4000),
        ("glm_exit_highbse_N300_q3", "strongcov", 4102, 300, 3, 12.0, "high-bse", 10, 4000),
        ("glm_exit_highbse_N300_q10", "strongeff", 4103, 300, 10, 10.0, "high-bse", 8, 4000),
        ("glm_exit_sep_N100_q1", "lattice", 4201, 100, 1, 0.5, "perfectly-separable-data", 8, 3000),
        ("glm_exit_sep_N300_q3", "lattice", 4202, 300, 3, 0.5, "perfectly-separable-data", 10, 3000),
        # blind searches for firth-fail (0 hits in 60 000 / 60 000 / 40 000 / 30 000 candidates; see --long):
        # ("glm_exit_firthfail_N100_q0", "rare", 4301, 100, 0, ...), (N100_q3, 4302), (N300_q2, 4303), (N300_q10, 4304)
    ]
    if only == "--long":
        # the reference's own firth-fail is a < 1e-5 event (0 in 7e5 rare-variant fits at N <= 300 by the C restatement with numpy's
        # summation order, 1 in 6e4 at N = 1000): `--long SEED N q MINUTES` is the hours-long search, one process per seed
        seed, N_, q_, minutes = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
        search("glm_exit_firthfail_N%d_q%d_s%d" % (N_, q_, seed), "rare", seed, N_, q_, 1.5, "firth-fail", 2, 10 ** 9, n_ctrl=4, budget_s=60 * minutes)
        sys.exit(0)
    if only == "--screened":
        # the same search with the C restatement in front (design_rare_screened): --screened SEED N q MINUTES
        seed, N_, q_, minutes = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
        search("glm_exit_firthfail_N%d_q%d_s%d" % (N_, q_, seed), "rarescreened", seed, N_, q_, 1.5, "firth-fail", 4, 10 ** 9, n_ctrl=4, budget_s=60 * minutes)
        sys.exit(0)
    for j in jobs:
        if only and only not in j[0]:
            continue
        search(*j)
