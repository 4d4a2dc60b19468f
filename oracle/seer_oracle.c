/*
 * seer_oracle.c -- CPU restatement of pyseer's per-variant association hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP engine in
 * pyseer_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product path (everything under pyseer_amd/) never imports, links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py)
 * against vectors produced by importing the reference itself (tests/golden/make_golden.py,
 * reference @ /root/reference, statsmodels 0.12.2, scipy 1.7.1), including the literals in
 * the reference's own tests/model_test.py and tests/lmm_test.py.
 *
 * Each function cites the reference lines it restates.  "SM:" = statsmodels 0.12.2 (an
 * un-vendored dependency of the reference, pin statsmodels>=0.10.0, requirements.txt:12);
 * "SP:" = scipy 1.7.1 (pin scipy>=1.1.0, requirements.txt:10).
 *
 * All arithmetic is IEEE fp64, plain loops, no BLAS.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* note bits: same order as pyseer_amd/_notes.py and include/seerhip.h */
#define NOTE_AF_FILTER      (1u << 0)
#define NOTE_PRE_FILTER     (1u << 1)
#define NOTE_BAD_CHISQ      (1u << 2)
#define NOTE_HIGH_BSE       (1u << 3)
#define NOTE_PERFECT_SEP    (1u << 4)
#define NOTE_MATRIX_INV     (1u << 5)
#define NOTE_FIRTH_FAIL     (1u << 6)
#define NOTE_MISSING_DATA   (1u << 7)
#define NOTE_LRT_FILTER     (1u << 8)

/* ------------------------------------------------------------------------------------------
 * Special functions (SP: scipy.stats tails used at model.py:53,68,339,369 and lmm.py:251)
 * ---------------------------------------------------------------------------------------- */

/* continued fraction for the regularised incomplete beta (modified Lentz). */
static double betacf(double a, double b, double x)
{
    const double TINY = 1e-300, EPS = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < TINY) d = TINY;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < EPS) break;
    }
    return h;
}

/* I_x(a,b) given ln(x) and ln(1-x) separately (callers know them accurately). */
static double betainc_core(double a, double b, double x, double lnx, double ln1mx)
{
    if (isnan(x) || isnan(a) || isnan(b)) return NAN;
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double lnB = lgamma(a) + lgamma(b) - lgamma(a + b);
    double front = exp(a * lnx + b * ln1mx - lnB);
    if (x < (a + 1.0) / (a + b + 2.0))
        return front * betacf(a, b, x) / a;
    return 1.0 - front * betacf(b, a, 1.0 - x) / b;
}

/* scipy.stats.chi2.sf(x, 1) = chdtrc(1, x) = igamc(1/2, x/2) = erfc(sqrt(x/2)).  model.py:68,339,369 */
ORC_API double orc_chi2_sf1(double x)
{
    if (isnan(x)) return NAN;
    if (x <= 0.0) return 1.0;
    return erfc(sqrt(0.5 * x));
}

/* scipy.stats.f.sf(x, 1, d) = fdtrc(1, d, x) = incbet(d/2, 1/2, d/(d+x)).  lmm.py:251 */
ORC_API double orc_f_sf_1(double x, double d)
{
    if (isnan(x)) return NAN;
    if (x <= 0.0) return 1.0;
    if (isinf(x)) return 0.0;
    double w = d / (d + x);
    double lnw = -log1p(x / d);
    double ln1mw = log(x / (d + x));
    return betainc_core(0.5 * d, 0.5, w, lnw, ln1mw);
}

/* 2*scipy.stats.t.sf(|t|, df) = I_{df/(df+t^2)}(df/2, 1/2).  model.py:53 (ttest_ind), SM OLS pvalues */
ORC_API double orc_t_sf2(double t, double df)
{
    if (isnan(t) || isnan(df)) return NAN;
    double t2 = t * t;
    if (t2 == 0.0) return 1.0;
    if (isinf(t2)) return 0.0;
    double w = df / (df + t2);
    return betainc_core(0.5 * df, 0.5, w, -log1p(t2 / df), log(t2 / (df + t2)));
}

/* ------------------------------------------------------------------------------------------
 * Small dense linear algebra
 * ---------------------------------------------------------------------------------------- */

/* LU with partial pivoting (LAPACK getrf semantics: exact-zero pivot => singular).
 * A is n*n row-major, overwritten.  Returns 0, or 1 if an exact zero pivot was met.  sign gets the
 * permutation sign. */
static int lu_decomp(double *A, int n, int *piv, int *sign)
{
    *sign = 1;
    int singular = 0;
    for (int c = 0; c < n; c++) {
        int p = c; double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) { double v = fabs(A[r * n + c]); if (v > best) { best = v; p = r; } }
        piv[c] = p;
        if (p != c) { for (int j = 0; j < n; j++) { double t = A[c * n + j]; A[c * n + j] = A[p * n + j]; A[p * n + j] = t; } *sign = -*sign; }
        double d = A[c * n + c];
        if (d == 0.0 || isnan(d)) { if (d == 0.0) singular = 1; continue; }
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] / d;
            A[r * n + c] = f;
            if (f != 0.0) for (int j = c + 1; j < n; j++) A[r * n + j] -= f * A[c * n + j];
        }
    }
    return singular;
}

/* numpy.linalg.inv: returns 1 (LinAlgError "Singular matrix") on exact singularity. Ainv row-major. */
static int mat_inv(const double *A, int n, double *Ainv)
{
    double *L = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    int sign;
    memcpy(L, A, sizeof(double) * n * n);
    int sing = lu_decomp(L, n, piv, &sign);
    if (sing) { free(L); free(piv); return 1; }
    for (int col = 0; col < n; col++) {
        double b[64];
        for (int i = 0; i < n; i++) b[i] = (i == col) ? 1.0 : 0.0;
        for (int c = 0; c < n; c++) { int p = piv[c]; if (p != c) { double t = b[c]; b[c] = b[p]; b[p] = t; } }
        for (int i = 0; i < n; i++) { double s = b[i]; for (int j = 0; j < i; j++) s -= L[i * n + j] * b[j]; b[i] = s; }
        for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int j = i + 1; j < n; j++) s -= L[i * n + j] * b[j]; b[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; i++) Ainv[i * n + col] = b[i];
    }
    free(L); free(piv);
    return 0;
}

/* numpy.linalg.det via LU. */
static double mat_det(const double *A, int n)
{
    double *L = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    int sign;
    memcpy(L, A, sizeof(double) * n * n);
    lu_decomp(L, n, piv, &sign);
    double d = sign;
    for (int i = 0; i < n; i++) d *= L[i * n + i];
    free(L); free(piv);
    return d;
}

/* cyclic Jacobi eigen-decomposition of a symmetric n*n matrix (row-major).  w: eigenvalues, V: columns. */
static void jacobi_eigh(const double *Ain, int n, double *w, double *V)
{
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; i++) { dg += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-34 * (dg + off) || off == 0.0) break;
        for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) {
            double apq = A[p * n + q];
            if (apq == 0.0) continue;
            double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < n; k++) { double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
            for (int k = 0; k < n; k++) { double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
            for (int k = 0; k < n; k++) { double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    free(A);
}

/* numpy.linalg.pinv of a symmetric matrix (rcond = 1e-15 on |eigenvalue| = singular value).  model.py:450 */
static void sym_pinv(const double *A, int n, double *P)
{
    double *w = (double *)malloc(sizeof(double) * n), *V = (double *)malloc(sizeof(double) * n * n);
    jacobi_eigh(A, n, w, V);
    double smax = 0.0;
    for (int i = 0; i < n; i++) if (fabs(w[i]) > smax) smax = fabs(w[i]);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) if (fabs(w[k]) > 1e-15 * smax) s += V[i * n + k] * V[j * n + k] / w[k];
        P[i * n + j] = s;
    }
    free(w); free(V);
}

/* one-sided Jacobi SVD of X (n*p row-major, n>=p): X = U diag(s) V^T.  U overwrites a copy (n*p). */
static void jacobi_svd(const double *X, int n, int p, double *U, double *s, double *V)
{
    memcpy(U, X, sizeof(double) * n * p);
    for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) V[i * p + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        int rotated = 0;
        for (int a = 0; a < p; a++) for (int b = a + 1; b < p; b++) {
            double alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < n; i++) { double ua = U[i * p + a], ub = U[i * p + b]; alpha += ua * ua; beta += ub * ub; gamma += ua * ub; }
            if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
            rotated = 1;
            double zeta = (beta - alpha) / (2.0 * gamma);
            double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
            for (int i = 0; i < n; i++) { double ua = U[i * p + a], ub = U[i * p + b]; U[i * p + a] = c * ua - sn * ub; U[i * p + b] = sn * ua + c * ub; }
            for (int i = 0; i < p; i++) { double va = V[i * p + a], vb = V[i * p + b]; V[i * p + a] = c * va - sn * vb; V[i * p + b] = sn * va + c * vb; }
        }
        if (!rotated) break;
    }
    for (int a = 0; a < p; a++) {
        double nn = 0; for (int i = 0; i < n; i++) nn += U[i * p + a] * U[i * p + a];
        s[a] = sqrt(nn);
        if (s[a] > 0) for (int i = 0; i < n; i++) U[i * p + a] /= s[a];
    }
}

/* numpy.linalg.pinv(X) for n*p X (rcond 1e-15) -> Xd (p*n row-major); also returns singular values. */
static void mat_pinv(const double *X, int n, int p, double *Xd, double *sv)
{
    double *U = (double *)malloc(sizeof(double) * n * p), *V = (double *)malloc(sizeof(double) * p * p);
    jacobi_svd(X, n, p, U, sv, V);
    double smax = 0; for (int a = 0; a < p; a++) if (sv[a] > smax) smax = sv[a];
    for (int j = 0; j < p; j++) for (int i = 0; i < n; i++) {
        double acc = 0;
        for (int a = 0; a < p; a++) if (sv[a] > 1e-15 * smax) acc += V[j * p + a] * U[i * p + a] / sv[a];
        Xd[j * n + i] = acc;
    }
    free(U); free(V);
}

/* ------------------------------------------------------------------------------------------
 * a1: pre_filtering  (pyseer/model.py:31-70)
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_pre_filtering(const double *p, const double *k, int n, int continuous,
                               double *prep, int *bad_chisq)
{
    *bad_chisq = 0;
    if (continuous) {
        /* SP: ttest_ind(p[k==1], p[k==0], equal_var=False)  model.py:53-55 */
        double s1 = 0, s0 = 0; int n1 = 0, n0 = 0;
        for (int i = 0; i < n; i++) { if (k[i] == 1.0) { s1 += p[i]; n1++; } else if (k[i] == 0.0) { s0 += p[i]; n0++; } }
        double m1 = s1 / n1, m0 = s0 / n0, q1 = 0, q0 = 0;
        for (int i = 0; i < n; i++) { if (k[i] == 1.0) { double d = p[i] - m1; q1 += d * d; } else if (k[i] == 0.0) { double d = p[i] - m0; q0 += d * d; } }
        double v1 = q1 / (n1 - 1), v0 = q0 / (n0 - 1);
        double vn1 = v1 / n1, vn0 = v0 / n0;
        double df = (vn1 + vn0) * (vn1 + vn0) / (vn1 * vn1 / (n1 - 1) + vn0 * vn0 / (n0 - 1));
        if (isnan(df)) df = 1.0;
        double t = (m1 - m0) / sqrt(vn1 + vn0);
        *prep = orc_t_sf2(t, df);
        return;
    }
    /* contingency table, model.py:57-61: exact == tests, NaN falls in no cell */
    double tb[2][2] = {{0, 0}, {0, 0}};
    for (int i = 0; i < n; i++) {
        if (p[i] == 1.0 && k[i] == 1.0) tb[0][0] += 1;
        else if (p[i] == 1.0 && k[i] == 0.0) tb[0][1] += 1;
        else if (p[i] == 0.0 && k[i] == 1.0) tb[1][0] += 1;
        else if (p[i] == 0.0 && k[i] == 0.0) tb[1][1] += 1;
    }
    int le1 = 0, le5 = 0;                                   /* model.py:65 */
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { if (tb[a][b] <= 1) le1++; if (tb[a][b] <= 5) le5++; }
    if (le1 > 0 || le5 > 1) *bad_chisq = 1;
    /* SP: chi2_contingency(table, correction=False): expected = outer(margins)/total; sum (o-e)^2/e */
    double r0 = tb[0][0] + tb[0][1], r1 = tb[1][0] + tb[1][1], c0 = tb[0][0] + tb[1][0], c1 = tb[0][1] + tb[1][1];
    double tot = r0 + r1;
    double e[2][2] = {{r0 * c0 / tot, r0 * c1 / tot}, {r1 * c0 / tot, r1 * c1 / tot}};
    double chi2 = 0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { double d = tb[a][b] - e[a][b]; chi2 += d * d / e[a][b]; }
    /* scipy raises ValueError when an expected count is 0 (reference would crash); we return NaN */
    if (e[0][0] == 0 || e[0][1] == 0 || e[1][0] == 0 || e[1][1] == 0) { *prep = NAN; return; }
    *prep = orc_chi2_sf1(chi2);
}

/* ------------------------------------------------------------------------------------------
 * a3: statsmodels Logit.fit(method='newton')  (SM:base/optimizer.py:407-446, SM:base/model.py:497-534,
 *     SM:discrete/discrete_model.py:205-211,1810,1861,1895,1950) as driven from model.py:316-330
 * status: 0 ok, 1 PerfectSeparationError, 2 LinAlgError
 * ---------------------------------------------------------------------------------------- */
static double logit_cdf(double x) { return 1.0 / (1.0 + exp(-x)); }

__attribute__((target_clones("avx2","default")))
static void logit_hess_score(const double *X, const double *y, int n, int pc, const double *beta,
                             double *XtWX, double *score)
{
    for (int a = 0; a < pc * pc; a++) XtWX[a] = 0;
    for (int a = 0; a < pc; a++) score[a] = 0;
    for (int i = 0; i < n; i++) {
        const double *x = X + (size_t)i * pc;
        double eta = 0; for (int a = 0; a < pc; a++) eta += x[a] * beta[a];
        double L = logit_cdf(eta), w = L * (1 - L), r = y[i] - L;
        for (int a = 0; a < pc; a++) {
            score[a] += r * x[a];
            double wa = w * x[a];
            for (int b = 0; b < pc; b++) XtWX[a * pc + b] += wa * x[b];
        }
    }
}

/* numpy's float64 add.reduce over one contiguous run (numpy/core/src/umath/loops_utils.h.src, pairwise_sum): plain loop below 8
 * elements, eight interleaved partial sums up to 128, recursive halves (first half rounded down to a multiple of 8) above.  The
 * reference's log-likelihood is np.sum of N logs (SM:discrete/discrete_model.py Logit.loglike); fit_firth's step-halving test
 * (model.py:467) compares two such sums that differ by less than their rounding noise near convergence, so the ORDER of the sum
 * decides how often the comparison is a coin flip: with a running sum this restatement reported a spurious `firth-fail` on 26 of
 * 24 332 forced-Firth rows at N = 5000, the reference itself on none of them (tests/golden/n5000_firth.npz). */
static double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) { double r = 0.; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

static double logit_loglike(const double *X, const double *y, int n, int pc, const double *beta)
{
    double stack_t[1024], *t = n <= 1024 ? stack_t : (double *)malloc(sizeof(double) * n);
    stack_t[0] = 0;
    for (int i = 0; i < n; i++) {
        const double *x = X + (size_t)i * pc;
        double eta = 0; for (int a = 0; a < pc; a++) eta += x[a] * beta[a];
        double q = 2 * y[i] - 1;
        t[i] = log(logit_cdf(q * eta));         /* SM Logit.loglike: sum(log(cdf(q*Xb))) */
    }
    double ll = np_pairwise_sum(t, n);
    if (t != stack_t) free(t);
    return ll;
}

ORC_API int orc_logit_newton(const double *X, const double *y, int n, int pc, const double *start,
                             int check_separation, double *beta, double *bse, double *llf, int *iters)
{
    double H[64 * 64], Hinv[64 * 64], sc[64], oldb[64];
    int it = 0, first = 1;
    for (int a = 0; a < pc; a++) beta[a] = start[a];
    for (;;) {
        /* while iterations < maxiter and any(|new - old| > tol): (old = inf on the first pass) */
        int moving = first;
        if (!first) for (int a = 0; a < pc; a++) if (fabs(beta[a] - oldb[a]) > 1e-8) moving = 1;
        if (!(it < 35 && moving)) break;
        first = 0;
        logit_hess_score(X, y, n, pc, beta, H, sc);
        for (int a = 0; a < pc * pc; a++) H[a] = -H[a] / n;      /* hess = hessian/nobs */
        for (int a = 0; a < pc; a++) H[a * pc + a] += 1e-10;     /* ridge_factor */
        for (int a = 0; a < pc; a++) { oldb[a] = beta[a]; sc[a] /= n; }
        if (mat_inv(H, pc, Hinv)) { *iters = it; return 2; }
        for (int a = 0; a < pc; a++) { double s = 0; for (int b = 0; b < pc; b++) s += Hinv[a * pc + b] * sc[b]; beta[a] = oldb[a] - s; }
        /* callback: _check_perfect_pred(newparams) */
        if (check_separation) {
            int all_close = 1;
            for (int i = 0; i < n && all_close; i++) {
                const double *x = X + (size_t)i * pc;
                double eta = 0; for (int a = 0; a < pc; a++) eta += x[a] * beta[a];
                double dv = logit_cdf(eta) - y[i];
                if (!(fabs(dv) <= 1e-8)) all_close = 0;           /* np.allclose(d, 0): |d| <= atol */
            }
            if (all_close) { *iters = it + 1; return 1; }
        }
        it++;
    }
    *iters = it;
    *llf = logit_loglike(X, y, n, pc, beta);
    /* Hinv = inv(-Hessian/nobs)/nobs, no ridge (SM:base/model.py:533-534) */
    logit_hess_score(X, y, n, pc, beta, H, sc);
    for (int a = 0; a < pc * pc; a++) H[a] = H[a] / n;
    if (mat_inv(H, pc, Hinv)) return 2;
    for (int a = 0; a < pc; a++) bse[a] = sqrt(Hinv[a * pc + a] / n);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a5: firth_likelihood + fit_firth  (pyseer/model.py:397-504)
 * The reference builds the N*N hat matrix (model.py:449-458); only its diagonal
 * h_i = w_i x_i^T V x_i is used (model.py:462), which is what is computed here.
 * returns 0 ok, 1 = None (no convergence / step-halving exhausted)
 * ---------------------------------------------------------------------------------------- */
static double firth_like(const double *X, const double *y, int n, int pc, const double *beta)
{
    double H[64 * 64], sc[64];
    logit_hess_score(X, y, n, pc, beta, H, sc);
    /* -(loglike + 0.5*log(det(-hessian)))   model.py:410-411 */
    return -(logit_loglike(X, y, n, pc, beta) + 0.5 * log(mat_det(H, pc)));
}

ORC_API double orc_firth_likelihood(const double *X, const double *y, int n, int pc, const double *beta)
{
    return firth_like(X, y, n, pc, beta);
}

/* Test-only knob (default 0 = the reference's comparison exactly): the step-halving test becomes
 * F(new) > F(old) + tie*|F(old)| for the FIRST comparison of each outer iteration.  With tie = +-2e-13 the parity tests can tell a rounding-decided halving (the reference's own
 * result is then a coin flip, DESIGN.md section 6 case 1) from a real disagreement. */
static double g_firth_tie = 0.0, g_firth_accept = 0.0;
ORC_API void orc_set_firth_tie(double tie) { g_firth_tie = tie; }
/* Second test-only knob (default 0 = off): a candidate whose largest coordinate change is below `eps` is accepted without the
 * F comparison.  The HIP kernels use eps = 1e-10 (DESIGN.md section 6 case 1): the reference halves such a step until the
 * comparison lands by rounding, or gets stuck one ulp away from beta (cur + 0.5 ulp rounds back up) and reports a spurious
 * firth-fail after step_limit halvings. */
ORC_API void orc_set_firth_accept_below(double eps) { g_firth_accept = eps; }
/* Third test-only knob (default 1 = the reference): convergence_limit is multiplied by `s`.  A Firth iteration that converges
 * linearly with a ratio near -1 (quasi-separated designs: steps alternate in sign and shrink by a few % each) moves beta by about
 * the limit itself per iteration when it stops, so WHICH iteration first sees a previous step below 1e-4 (model.py:477) changes
 * the answer by ~1e-4; with s = 1 -+ 1e-4 the parity tests recognise a stop decided within 1e-8 of the limit
 * (tests/golden/glm_exit_highbse_N100_q1.npz row 5: step 52 has norm 1.00003e-4, step 53 9.58e-5). */
static double g_firth_conv_scale = 1.0;
ORC_API void orc_set_firth_conv_scale(double s) { g_firth_conv_scale = s; }
static int firth_tiny_step(const double *nb, const double *cur, int pc)
{
    double m = 0; for (int a = 0; a < pc; a++) { double d = fabs(nb[a] - cur[a]); if (d > m) m = d; }
    return m < g_firth_accept;
}

/* trace (test-only, may be NULL): [0] halvings taken, [1] the smallest rise F(new) - F(old) > 1e-12 |F| that caused one (INFINITY: none), [2] the
 * largest rise, [3] the step's norm at [1], [4] iterations: lets the fixture search (tests/golden/make_subtau_golden.py) find fits on which the
 * reference halves a step for a rise that is real (above the rounding noise of F) yet small. */
static int fit_firth_impl(const double *X, const double *y, int n, int pc, const double *start,
                          int step_limit, double convergence_limit,
                          double *beta_out, double *bse1, double *fitll, double *trace);
ORC_API int orc_fit_firth(const double *X, const double *y, int n, int pc, const double *start,
                          int step_limit, double convergence_limit,
                          double *beta_out, double *bse1, double *fitll)
{
    return fit_firth_impl(X, y, n, pc, start, step_limit, convergence_limit, beta_out, bse1, fitll, NULL);
}
ORC_API int orc_fit_firth_traced(const double *X, const double *y, int n, int pc, const double *start,
                                 int step_limit, double convergence_limit,
                                 double *beta_out, double *bse1, double *fitll, double *trace)
{
    trace[0] = 0; trace[1] = INFINITY; trace[2] = 0; trace[3] = 0; trace[4] = 0;
    return fit_firth_impl(X, y, n, pc, start, step_limit, convergence_limit, beta_out, bse1, fitll, trace);
}
static int fit_firth_impl(const double *X, const double *y, int n, int pc, const double *start,
                          int step_limit, double convergence_limit,
                          double *beta_out, double *bse1, double *fitll, double *trace)
{
    double I[64 * 64], V[64 * 64], sc[64], U[64];
    double *cur = (double *)malloc(sizeof(double) * pc), *prev = (double *)malloc(sizeof(double) * pc),
           *nb = (double *)malloc(sizeof(double) * pc);
    double *pi = (double *)malloc(sizeof(double) * n);
    for (int a = 0; a < pc; a++) { cur[a] = start[a]; prev[a] = start[a]; }
    int i, ok = 0;
    double last_step_norm = INFINITY;   /* ||beta_i - beta_{i-1}|| */
    convergence_limit *= g_firth_conv_scale;
    for (i = 0; i < step_limit; i++) {
        logit_hess_score(X, y, n, pc, cur, I, sc);
        sym_pinv(I, pc, V);                                     /* model.py:450 */
        for (int a = 0; a < pc; a++) U[a] = 0;
        for (int r = 0; r < n; r++) {
            const double *x = X + (size_t)r * pc;
            double eta = 0; for (int a = 0; a < pc; a++) eta += x[a] * cur[a];
            double p_ = logit_cdf(eta), w = p_ * (1 - p_);
            double q = 0; for (int a = 0; a < pc; a++) { double s = 0; for (int b = 0; b < pc; b++) s += V[a * pc + b] * x[b]; q += x[a] * s; }
            double h = w * q;                                   /* diag(H), model.py:455-458,462 */
            double res = y[r] - p_ + h * (0.5 - p_);
            for (int a = 0; a < pc; a++) U[a] += x[a] * res;
        }
        for (int a = 0; a < pc; a++) { double s = 0; for (int b = 0; b < pc; b++) s += V[a * pc + b] * U[b]; nb[a] = cur[a] + s; }
        /* step halving, model.py:465-474 (NaN comparison is False -> accept) */
        int j = 0;
        double fcur = firth_like(X, y, n, pc, cur);
        double fnew = 0;
        while (!firth_tiny_step(nb, cur, pc) && (fnew = firth_like(X, y, n, pc, nb)) > fcur + (j == 0 ? g_firth_tie * fabs(fcur) : 0.0)) {
            if (trace) {
                double rise = fnew - fcur, sn = 0;
                for (int a = 0; a < pc; a++) sn += (nb[a] - cur[a]) * (nb[a] - cur[a]);
                trace[0] += 1;
                if (rise > 1e-12 * fabs(fcur) && rise < trace[1]) { trace[1] = rise; trace[3] = sqrt(sn); }
                if (rise > trace[2]) trace[2] = rise;
            }
            for (int a = 0; a < pc; a++) nb[a] = cur[a] + 0.5 * (nb[a] - cur[a]);
            j++;
            if (j > step_limit) { free(cur); free(prev); free(nb); free(pi); return 1; }
        }
        /* append; then test ||beta_i - beta_{i-1}|| (the PREVIOUS step), model.py:476-479 */
        double nrm = 0;
        if (i > 0) { for (int a = 0; a < pc; a++) { double d = cur[a] - prev[a]; nrm += d * d; } nrm = sqrt(nrm); last_step_norm = nrm; }
        for (int a = 0; a < pc; a++) { prev[a] = cur[a]; cur[a] = nb[a]; }
        if (trace) trace[4] = i + 1;
        if (i > 0 && nrm < convergence_limit) { ok = 1; break; }
    }
    /* after the loop model.py:482-484 re-tests ||beta_iterations[i]-beta_iterations[i-1]|| with the final i;
     * on break that is the same norm; on exhaustion (i = step_limit-1) it is the last computed one. */
    if (!ok) { if (last_step_norm < convergence_limit) ok = 1; }
    if (ok) {
        for (int a = 0; a < pc; a++) beta_out[a] = cur[a];      /* beta_iterations[-1] */
        *fitll = -firth_like(X, y, n, pc, cur);
        logit_hess_score(X, y, n, pc, cur, I, sc);
        *bse1 = pc > 1 ? sqrt(I[1 * pc + 1]) : NAN;             /* sqrt(-hessian[1,1]), model.py:491 */
    }
    free(cur); free(prev); free(nb); free(pi);
    return ok ? 0 : 1;
}

/* ------------------------------------------------------------------------------------------
 * a4: statsmodels OLS(p, v).fit()   (SM:regression/linear_model.py fit(method='pinv')) via model.py:300-312
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_ols(const double *X, const double *y, int n, int pc, double *beta, double *bse, double *pvalues)
{
    double *Xd = (double *)malloc(sizeof(double) * n * pc), sv[64], ncov[64 * 64];
    mat_pinv(X, n, pc, Xd, sv);
    for (int a = 0; a < pc; a++) { double s = 0; for (int i = 0; i < n; i++) s += Xd[(size_t)a * n + i] * y[i]; beta[a] = s; }
    for (int a = 0; a < pc; a++) for (int b = 0; b < pc; b++) { double s = 0; for (int i = 0; i < n; i++) s += Xd[(size_t)a * n + i] * Xd[(size_t)b * n + i]; ncov[a * pc + b] = s; }
    /* rank = matrix_rank(diag(singular_values)): tol = smax * p * eps */
    double smax = 0; for (int a = 0; a < pc; a++) if (sv[a] > smax) smax = sv[a];
    int rank = 0; for (int a = 0; a < pc; a++) if (sv[a] > smax * pc * DBL_EPSILON) rank++;
    double ssr = 0;
    for (int i = 0; i < n; i++) { double f = 0; for (int a = 0; a < pc; a++) f += X[(size_t)i * pc + a] * beta[a]; double r = y[i] - f; ssr += r * r; }
    double dfr = n - rank, scale = ssr / dfr;
    for (int a = 0; a < pc; a++) { bse[a] = sqrt(scale * ncov[a * pc + a]); pvalues[a] = orc_t_sf2(beta[a] / bse[a], dfr); }
    free(Xd);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a2: fixed_effects_regression  (pyseer/model.py:202-394), without lineage (a6) and without the
 * af-filter branch (model.py:255-260, decided by the caller).  Z = [m | c] (n*q row-major).
 * out5 = prep, pvalue, kbeta, bse, intercept ; betas[q]
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_fixed_effects(const double *y, const double *k, const double *Z, int n, int q,
                               int continuous, double pret, double lrtt, double null_llf, double null_firth,
                               double *out5, double *betas, unsigned *notes_out, int *prefilter, int *filter)
{
    unsigned notes = 0;
    double prep; int bad;
    for (int a = 0; a < 5; a++) out5[a] = NAN;
    for (int a = 0; a < q; a++) betas[a] = NAN;
    orc_pre_filtering(y, k, n, continuous, &prep, &bad);
    out5[0] = prep;
    if (bad) notes |= NOTE_BAD_CHISQ;
    if (prep > pret || !isfinite(prep)) {                       /* model.py:266 (strict >) */
        notes |= NOTE_PRE_FILTER; *notes_out = notes; *prefilter = 1; *filter = 0; return;
    }
    int pc = q + 2;
    double *X = (double *)malloc(sizeof(double) * (size_t)n * pc);
    for (int i = 0; i < n; i++) { X[(size_t)i * pc] = 1.0; X[(size_t)i * pc + 1] = k[i]; for (int a = 0; a < q; a++) X[(size_t)i * pc + 2 + a] = Z[(size_t)i * q + a]; }
    double beta[64], bse[64], pv[64], lrt_p = NAN, kbeta = NAN, kbse = NAN, icpt = NAN;
    if (continuous) {
        orc_ols(X, y, n, pc, beta, bse, pv);
        icpt = beta[0]; kbeta = beta[1]; kbse = bse[1]; lrt_p = pv[1];
        for (int a = 0; a < q; a++) betas[a] = beta[2 + a];
    } else {
        double start[64]; for (int a = 0; a < pc; a++) start[a] = 0;
        double mean = 0; for (int i = 0; i < n; i++) mean += y[i]; mean /= n;
        start[0] = log(mean / (1 - mean));                      /* model.py:323-324 */
        if (!bad) {
            double llf; int iters;
            int st = orc_logit_newton(X, y, n, pc, start, 1, beta, bse, &llf, &iters);
            if (st == 1) { bad = 1; notes |= NOTE_PERFECT_SEP; }
            else if (st == 2) { bad = 1; notes |= NOTE_MATRIX_INV; }
            else if (bse[1] > 3) { bad = 1; notes |= NOTE_HIGH_BSE; }
            else {
                double lrstat = -2 * (null_llf - llf);
                lrt_p = 1; if (lrstat > 0) lrt_p = orc_chi2_sf1(lrstat);
                icpt = beta[0]; kbeta = beta[1]; kbse = bse[1];
                for (int a = 0; a < q; a++) betas[a] = beta[2 + a];
            }
        }
        if (bad) {
            double b1, fitll;
            int st = orc_fit_firth(X, y, n, pc, start, 1000, 1e-4, beta, &b1, &fitll);
            if (st) {
                notes |= NOTE_FIRTH_FAIL; free(X);
                *notes_out = notes; *prefilter = 0; *filter = 1; return;   /* model.py:357-362 */
            }
            icpt = beta[0]; kbeta = beta[1]; kbse = b1;
            for (int a = 0; a < q; a++) betas[a] = beta[2 + a];
            double lrstat = -2 * (null_firth - fitll);
            lrt_p = 1; if (lrstat > 0) lrt_p = orc_chi2_sf1(lrstat);
        }
    }
    free(X);
    out5[1] = lrt_p; out5[2] = kbeta; out5[3] = kbse; out5[4] = icpt;
    *prefilter = 0;
    if (lrt_p > lrtt || !isfinite(lrt_p) || !isfinite(kbeta)) { notes |= NOTE_LRT_FILTER; *filter = 1; }   /* model.py:384 */
    else *filter = 0;
    *notes_out = notes;
}

/* ------------------------------------------------------------------------------------------
 * LMM: a8 fit_lmm_block (lmm.py:228-260), a9 rotate (lmm_cov.py:165-194, Linreg.regress :874-880),
 *      a10 nLLeval/nLLcore (lmm_cov.py:597-838), computeAKA/AKB (:885-916)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int n, k, D;
    double *U;       /* n*k row-major */
    double *S;       /* k */
    double *C;       /* n*D covariates incl. intercept (LAST), lmm.py:95-99 */
    double *Cd;      /* D*n pinv(C)  (Linreg.Xdagger, lmm_cov.py:869) */
    double *y;       /* n */
    double *UY;      /* k   rotate(y) */
} orc_lmm;

static void lmm_regress(const orc_lmm *L, const double *a, double *out)
{
    double b[64];
    for (int d = 0; d < L->D; d++) { double s = 0; for (int i = 0; i < L->n; i++) s += L->Cd[(size_t)d * L->n + i] * a[i]; b[d] = s; }
    for (int i = 0; i < L->n; i++) { double s = 0; for (int d = 0; d < L->D; d++) s += L->C[(size_t)i * L->D + d] * b[d]; out[i] = a[i] - s; }
}

/* population std (numpy .std(0)) of a vector */
static double pop_std(const double *a, int n)
{
    double m = 0; for (int i = 0; i < n; i++) m += a[i]; m /= n;
    double q = 0; for (int i = 0; i < n; i++) { double d = a[i] - m; q += d * d; }
    return sqrt(q / n);
}

/* rotate up to 8 columns at once (shares each row of U across the columns) */
__attribute__((target_clones("avx2","default")))
static void lmm_rotate_cols(const orc_lmm *L, double *const *acols, int nb, double *const *ua)
{
    int n = L->n, k = L->k;
    for (int b = 0; b < nb; b++) {
        /* treat pathological case where a variable is explained by the covariates, lmm_cov.py:179-181 */
        if (pop_std(acols[b], n) <= 1e-10) for (int i = 0; i < n; i++) acols[b][i] = 0.0;
        for (int r = 0; r < k; r++) ua[b][r] = 0.0;
    }
    for (int i = 0; i < n; i++) {
        const double *u = L->U + (size_t)i * k;
        for (int b = 0; b < nb; b++) {
            double a = acols[b][i];
            if (a == 0.0) continue;
            double *o = ua[b];
            for (int r = 0; r < k; r++) o[r] += a * u[r];
        }
    }
}

ORC_API orc_lmm *orc_lmm_create(const double *U, const double *S, int n, int k,
                                const double *y, const double *C, int D)
{
    orc_lmm *L = (orc_lmm *)calloc(1, sizeof(orc_lmm));
    L->n = n; L->k = k; L->D = D;
    L->U = (double *)malloc(sizeof(double) * (size_t)n * k); memcpy(L->U, U, sizeof(double) * (size_t)n * k);
    L->S = (double *)malloc(sizeof(double) * k); memcpy(L->S, S, sizeof(double) * k);
    L->C = (double *)malloc(sizeof(double) * (size_t)n * D); memcpy(L->C, C, sizeof(double) * (size_t)n * D);
    L->y = (double *)malloc(sizeof(double) * n); memcpy(L->y, y, sizeof(double) * n);
    L->Cd = (double *)malloc(sizeof(double) * (size_t)n * D);
    double sv[64];
    mat_pinv(C, n, D, L->Cd, sv);
    L->UY = (double *)malloc(sizeof(double) * k);
    double *a = (double *)malloc(sizeof(double) * n);
    lmm_regress(L, y, a);
    double *ac[1] = {a}; double *uc[1] = {L->UY};
    lmm_rotate_cols(L, ac, 1, uc);
    free(a);
    return L;
}

ORC_API void orc_lmm_destroy(orc_lmm *L)
{
    if (!L) return;
    free(L->U); free(L->S); free(L->C); free(L->Cd); free(L->y); free(L->UY); free(L);
}

/* nLLeval(h2) without SNPs: the objective of findH2 (lmm_cov.py:427-478, 597-684, 686-838). */
ORC_API double orc_lmm_nll(const orc_lmm *L, double h2)
{
    if (h2 < 0.0 || h2 >= 1.0) return 3e20;
    int k = L->k; double N = L->n - L->D;
    double yKy = 0, logdet = 0;
    for (int r = 0; r < k; r++) { double sd = h2 * L->S[r] + (1.0 - h2); yKy += L->UY[r] * L->UY[r] / sd; logdet += log(sd); }
    double sigma2 = yKy / N;
    return 0.5 * (logdet + N * (log(2.0 * M_PI * sigma2) + 1));
}

/* fit_lmm_block: Xv = B variants, variant-major (B*n).  returns 0, or -1 when h2 is outside [0,1)
 * (reference: nLLeval returns no 'beta' -> KeyError, lmm_cov.py:667-670, lmm_test.py:416-417). */
ORC_API int orc_lmm_block(const orc_lmm *L, double h2, const double *Xv, int B,
                          double *beta, double *bse, double *frac_h2, double *pval)
{
    if (h2 < 0.0 || h2 >= 1.0) return -1;
    int n = L->n, k = L->k; double N = n - L->D;
    double *Sd = (double *)malloc(sizeof(double) * k);
    double yKy = 0;
    for (int r = 0; r < k; r++) { Sd[r] = h2 * L->S[r] + (1.0 - h2); yKy += L->UY[r] * L->UY[r] / Sd[r]; }
#pragma omp parallel
    {
        double *abuf = (double *)malloc(sizeof(double) * (size_t)n * 8), *ubuf = (double *)malloc(sizeof(double) * (size_t)k * 8);
#pragma omp for schedule(dynamic, 1)
        for (int b0 = 0; b0 < B; b0 += 8) {
            int nb = B - b0 < 8 ? B - b0 : 8;
            double *ac[8], *uc[8];
            for (int b = 0; b < nb; b++) { ac[b] = abuf + (size_t)b * n; uc[b] = ubuf + (size_t)b * k; lmm_regress(L, Xv + (size_t)(b0 + b) * n, ac[b]); }
            lmm_rotate_cols(L, ac, nb, uc);
            for (int b = 0; b < nb; b++) {
                double sKs = 0, sKy = 0;
                for (int r = 0; r < k; r++) { double u = uc[b][r]; sKs += u * u / Sd[r]; sKy += u * L->UY[r] / Sd[r]; }
                double bt = sKy / sKs;                          /* lmm_cov.py:802 */
                if (isnan(bt) && sKy == 0.0) bt = 0.0;           /* :803-805 */
                double veb = sKy * bt, r2 = yKy - veb;
                double var = r2 / (N - 1.0) / sKs;              /* :813 */
                double frac = veb / yKy;                        /* :814 */
                double chi2 = bt * bt / var;                    /* lmm.py:248 */
                beta[b0 + b] = bt; bse[b0 + b] = sqrt(var); frac_h2[b0 + b] = sqrt(frac);
                pval[b0 + b] = orc_f_sf_1(chi2, n - (L->D + 1)); /* lmm.py:251-253 */
            }
        }
        free(abuf); free(ubuf);
    }
    free(Sd);
    return 0;
}

/* a7: fit_lmm orchestration (lmm.py:125-226) minus lineage: per variant prefilter (>=), zeroing, drop of
 * all-zero columns, block fit, LRT filter (>=).  afmask[v]!=0 marks an af-filtered variant (pattern None).
 * rows: V*5 = prep,pvalue,kbeta,bse,frac_h2 */
ORC_API int orc_fit_lmm(const orc_lmm *L, double h2, const double *Xv, const unsigned char *afmask, int V,
                        int continuous, double filter_pvalue, double lrt_pvalue,
                        double *rows, unsigned *notes, int *prefilter, int *filter)
{
    int n = L->n;
    int *idx = (int *)malloc(sizeof(int) * V); int nt = 0;
    double *Xt = (double *)malloc(sizeof(double) * (size_t)V * n);
    for (int v = 0; v < V; v++) {
        for (int a = 0; a < 5; a++) rows[v * 5 + a] = NAN;
        notes[v] = 0; prefilter[v] = 1; filter[v] = 1;
        if (afmask && afmask[v]) { notes[v] = NOTE_AF_FILTER; prefilter[v] = 1; filter[v] = 0; continue; }
        double prep; int bad;
        orc_pre_filtering(L->y, Xv + (size_t)v * n, n, continuous, &prep, &bad);
        if (bad) notes[v] |= NOTE_BAD_CHISQ;
        if (prep >= filter_pvalue || !isfinite(prep)) {          /* lmm.py:174 (>=) */
            notes[v] |= NOTE_PRE_FILTER; rows[v * 5] = prep; prefilter[v] = 1; filter[v] = 0; continue;
        }
        rows[v * 5] = prep; prefilter[v] = 0;
        /* lmm.py:188 drops all-zero columns AFTER the loop; an unfiltered all-zero column would desync the
         * reference (assert, lmm.py:195).  Mirror: it is dropped from the block but stays in the list. */
        int allzero = 1; for (int i = 0; i < n; i++) if (Xv[(size_t)v * n + i] != 0.0) { allzero = 0; break; }
        if (allzero) { free(idx); free(Xt); return -2; }
        memcpy(Xt + (size_t)nt * n, Xv + (size_t)v * n, sizeof(double) * n);
        idx[nt++] = v;
    }
    if (nt) {
        double *b = (double *)malloc(sizeof(double) * nt * 4);
        int rc = orc_lmm_block(L, h2, Xt, nt, b, b + nt, b + 2 * nt, b + 3 * nt);
        if (rc) { free(b); free(idx); free(Xt); return rc; }
        for (int t = 0; t < nt; t++) {
            int v = idx[t]; double pv = b[3 * nt + t];
            rows[v * 5 + 1] = pv;
            if (pv >= lrt_pvalue || !isfinite(pv)) { notes[v] |= NOTE_LRT_FILTER; filter[v] = 1; }   /* lmm.py:201 */
            else { rows[v * 5 + 2] = b[t]; rows[v * 5 + 3] = b[nt + t]; rows[v * 5 + 4] = b[2 * nt + t]; filter[v] = 0; }
        }
        free(b);
    }
    free(idx); free(Xt);
    return 0;
}

/* batch driver for the fixed-effects path (used by the CPU-baseline leg and the parity tests):
 * Kv = V variants, variant-major (V*n doubles). */
ORC_API void orc_fixed_effects_batch(const double *y, const double *Kv, const double *Z, int n, int q, int V,
                                     int continuous, double pret, double lrtt, double null_llf, double null_firth,
                                     double *out5, double *betas, unsigned *notes, int *prefilter, int *filter)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int v = 0; v < V; v++)
        orc_fixed_effects(y, Kv + (size_t)v * n, Z, n, q, continuous, pret, lrtt, null_llf, null_firth,
                          out5 + (size_t)v * 5, betas + (size_t)v * (q > 0 ? q : 1), notes + v, prefilter + v, filter + v);
}

/* forced-Firth batch (config C4): fit_firth on every variant, start vector as model.py:323-324 */
ORC_API void orc_firth_batch(const double *y, const double *Kv, const double *Z, int n, int q, int V,
                             double *out4 /* intercept,kbeta,bse,fitll */, double *betas, int *status)
{
    double mean = 0; for (int i = 0; i < n; i++) mean += y[i]; mean /= n;
#pragma omp parallel for schedule(dynamic, 1)
    for (int v = 0; v < V; v++) {
        int pc = q + 2;
        double *X = (double *)malloc(sizeof(double) * (size_t)n * pc);
        for (int i = 0; i < n; i++) { X[(size_t)i * pc] = 1.0; X[(size_t)i * pc + 1] = Kv[(size_t)v * n + i]; for (int a = 0; a < q; a++) X[(size_t)i * pc + 2 + a] = Z[(size_t)i * q + a]; }
        double start[64], beta[64], b1, fitll;
        for (int a = 0; a < pc; a++) start[a] = 0;
        start[0] = log(mean / (1 - mean));
        int st = orc_fit_firth(X, y, n, pc, start, 1000, 1e-4, beta, &b1, &fitll);
        status[v] = st;
        for (int a = 0; a < 4; a++) out4[(size_t)v * 4 + a] = NAN;
        for (int a = 0; a < q; a++) betas[(size_t)v * (q > 0 ? q : 1) + a] = NAN;
        if (!st) { out4[(size_t)v * 4] = beta[0]; out4[(size_t)v * 4 + 1] = beta[1]; out4[(size_t)v * 4 + 2] = b1; out4[(size_t)v * 4 + 3] = fitll;
                   for (int a = 0; a < q; a++) betas[(size_t)v * (q > 0 ? q : 1) + a] = beta[2 + a]; }
        free(X);
    }
}

/* ------------------------------------------------------------------------------------------
 * a6: fit_lineage_effect (pyseer/model.py:151-199): Logit(k ~ [1, lin, c]).fit(newton) with the default zero start;
 * returns argmax over the lineage columns of |params|/bse, or -1 (None) on PerfectSeparationError / LinAlgError.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_lineage_effect(const double *lin, int l, const double *cov, int j, const double *k, int n)
{
    int pc = 1 + l + j;
    double *X = (double *)malloc(sizeof(double) * (size_t)n * pc);
    for (int i = 0; i < n; i++) {
        X[(size_t)i * pc] = 1.0;
        for (int a = 0; a < l; a++) X[(size_t)i * pc + 1 + a] = lin[(size_t)i * l + a];
        for (int a = 0; a < j; a++) X[(size_t)i * pc + 1 + l + a] = cov[(size_t)i * j + a];
    }
    double start[64], beta[64], bse[64], llf; int iters;
    for (int a = 0; a < pc; a++) start[a] = 0;
    int st = orc_logit_newton(X, k, n, pc, start, 1, beta, bse, &llf, &iters);
    free(X);
    if (st) return -1;
    int best = 0; double bw = fabs(beta[1]) / bse[1];
    for (int a = 1; a < l; a++) { double w = fabs(beta[1 + a]) / bse[1 + a]; if (w > bw) { bw = w; best = a; } }
    return best;
}

#ifdef _OPENMP
#include <omp.h>
#endif
/* threads of the batch entry points (a container's CPU quota can be far below its visible CPU count) */
ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_abi_version(void) { return 1; }
