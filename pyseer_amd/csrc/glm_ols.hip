// glm_ols.hip -- the continuous phenotype (statsmodels OLS.fit, model.py:299-312; split off glm_kernels.hip in round 5): k_glm_ols_tab (the
// product path: the variant's sums by nibble table), k_glm_ols (per-sample kernel), k_glm_ols_pinv (rank-deficient designs, numpy pinv).
#include "glm_passes.h"
#include "glm_blk.h"

// =====================================================================================================================
// OLS (continuous phenotype; model.py:299-312) -- closed form through the normal equations, one variant per lane.
// XtX_shared: packed lower (Q+1)x(Q+1) of [1, W] ; Xty_shared: Q+1
// =====================================================================================================================
template <int Q>
__global__ __launch_bounds__(64) void k_glm_ols(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                const double *__restrict__ y, const double *__restrict__ W,
                                                const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                const double *__restrict__ yc, const double *__restrict__ ZtZ,
                                                const double *__restrict__ Zty, GlmParams P,
                                                double *__restrict__ out, uint32_t *__restrict__ flags,
                                                int *__restrict__ pinv_list, int *__restrict__ pinv_count)
{
    constexpr int PC = Q + 2;
    const int64_t v = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const int N = P.N, NB64 = P.NB64;
    uint32_t fl = 0;
    bool bad = false; int m = 0;
    double prep = glm_prefilter(T, Vpad, v, NB64, N, y1, y0, yc, P, &bad, &m);
    double pval = NAN, kbeta = NAN, kbse = NAN, icpt = NAN;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = NAN;
    bool go = true;
    if (P.af_on) {
        const double af = (double)m / (double)N;
        if (!(P.min_af <= af && af <= P.max_af)) { fl = SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER; go = false; prep = NAN; }
    }
    if (go && (prep > P.pret || !isfinite(prep))) { fl |= SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER; go = false; }
    if (go) {
        // X^T X and X^T y: shared blocks from the host, variant column accumulated here
        double sxz[Q > 0 ? Q : 1], sxy = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) sxz[j] = 0.0;
        for (int sb = 0; sb < NB64; ++sb) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
            for (int b = 0; b < nb; ++b) {
                const int i = sb * 64 + b;
                const double xd = (double)(unsigned)((w64 >> b) & 1ull);
                sxy = fma(xd, y[i], sxy);
#pragma unroll
                for (int j = 0; j < Q; ++j) sxz[j] = fma(xd, W[(int64_t)i * Q + j], sxz[j]);
            }
        }
        double A[PC * (PC + 1) / 2], rhs[PC], e1[PC];
        A[sidx(0, 0)] = ZtZ[0]; A[sidx(1, 0)] = (double)m; A[sidx(1, 1)] = (double)m;
        rhs[0] = Zty[0]; rhs[1] = sxy;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            A[sidx(2 + j, 0)] = ZtZ[sidx(1 + j, 0)]; A[sidx(2 + j, 1)] = sxz[j]; rhs[2 + j] = Zty[1 + j];
#pragma unroll
            for (int k = 0; k <= j; ++k) A[sidx(2 + j, 2 + k)] = ZtZ[sidx(1 + j, 1 + k)];
        }
        double det;
        const bool ok = ldl_factor<PC>(A, 4.0e-16, &det);
        ldl_solve<PC>(A, rhs);
#pragma unroll
        for (int a = 0; a < PC; ++a) { beta[a] = rhs[a]; e1[a] = (a == 1) ? 1.0 : 0.0; }
        ldl_solve<PC>(A, e1);
        double ssr = 0.0;
        for (int sb = 0; sb < NB64; ++sb) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
            for (int b = 0; b < nb; ++b) {
                const int i = sb * 64 + b;
                double f = fma(beta[1], (double)(unsigned)((w64 >> b) & 1ull), beta[0]);
#pragma unroll
                for (int j = 0; j < Q; ++j) f = fma(beta[2 + j], W[(int64_t)i * Q + j], f);
                const double r = y[i] - f;
                ssr = fma(r, r, ssr);
            }
        }
        const double dfr = (double)(N - PC);
        const double scale = ssr / dfr;
        kbse = sqrt(scale * e1[1]);
        kbeta = beta[1]; icpt = beta[0];
        pval = sh_t_sf2(kbeta / kbse, dfr);                                   // res.pvalues[1]
        if (!ok) {                          // rank-deficient design: statsmodels' OLS is pinv-based -> k_glm_ols_pinv finishes it
            const int s2 = atomicAdd(pinv_count, 1); pinv_list[s2] = (int)v;
            out[v] = prep; flags[v] = fl;
            return;
        }
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(kbeta)) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
    }
    out[v] = prep; out[V + v] = pval; out[2 * V + v] = kbeta; out[3 * V + v] = kbse; out[4 * V + v] = icpt;
#pragma unroll
    for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = go ? beta[2 + j] : NAN;
    flags[v] = fl;
}


// OLS with the per-variant sums by nibble-table lookup (the same device as k_lmm_linear_tabn): per 4-sample nibble of the variant's
// word, tab[sb][nib][value] = (sum yc, sum yc^2, sum z_0 .. sum z_{Q-1}) over the set bits, built once per run.  One walk over the
// words gives the Welch prefilter sums, X^T y and the variant column of X^T X; the residual sum of squares is y^T y - beta^T X^T y
// (the normal equations hold at beta), recomputed sample by sample only where that difference cancels (R^2 > 1 - 1e-6).
template <int Q>
__global__ __launch_bounds__(256) void k_glm_ols_build_tab(const double *__restrict__ yc, const double *__restrict__ W, int N, int NB64,
                                                           double *__restrict__ tab)
{
    constexpr int NE = Q + 2;
    const int e = blockIdx.x * 256 + threadIdx.x;              // entry = (sb * 16 + nib) * 16 + value
    if (e >= NB64 * 256) return;
    const int value = e & 15, base = (e >> 4) * 4;
    for (int c = 0; c < NE; ++c) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) {
            const int i = base + k;
            if (!((value >> k) & 1) || i >= N) continue;
            a += (c == 0) ? yc[i] : ((c == 1) ? yc[i] * yc[i] : W[(int64_t)i * Q + (c - 2)]);
        }
        tab[(int64_t)e * NE + c] = a;
    }
}

template <int Q>
__global__ __launch_bounds__(256) void k_glm_ols_tab(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                     const double *__restrict__ y, const double *__restrict__ W,
                                                     const double *__restrict__ tab, const double *__restrict__ ZtZ,
                                                     const double *__restrict__ Zty, GlmParams P, double ymean, double yty,
                                                     double *__restrict__ out, uint32_t *__restrict__ flags,
                                                     int *__restrict__ pinv_list, int *__restrict__ pinv_count)
{
    constexpr int PC = Q + 2, NE = Q + 2;
    __shared__ __attribute__((aligned(16))) double lt[2][256 * NE];
    const int tid = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;               // Vpad is a multiple of 256: every thread walks (rows >= V are zero)
    const int N = P.N, NB64 = P.NB64;
    double acc[NE];
#pragma unroll
    for (int c = 0; c < NE; ++c) acc[c] = 0.0;
    int m = 0;
#pragma unroll
    for (int c = 0; c < NE; ++c) lt[0][tid * NE + c] = tab[(int64_t)tid * NE + c];
    for (int sb = 0; sb < NB64; sb++) {
        double nxt[NE];
        const bool more = sb + 1 < NB64;
#pragma unroll
        for (int c = 0; c < NE; ++c) nxt[c] = more ? tab[((int64_t)(sb + 1) * 256 + tid) * NE + c] : 0.0;
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        __syncthreads();
        m += __popcll(w);
        const double *cur = lt[sb & 1];
#pragma unroll 4
        for (int nib = 0; nib < 16; ++nib) {
            const double *e = cur + (nib * 16 + (int)((w >> (4 * nib)) & 15ull)) * NE;
#pragma unroll
            for (int c = 0; c < NE; ++c) acc[c] += e[c];
        }
#pragma unroll
        for (int c = 0; c < NE; ++c) lt[(sb + 1) & 1][tid * NE + c] = nxt[c];
    }
    if (v >= V) return;
    uint32_t fl = 0;
    double prep = sh_prefilter_welch((double)m, acc[0], acc[1], (double)(N - m), P.yc_sum - acc[0], P.yc_sq - acc[1]);
    double pval = NAN, kbeta = NAN, kbse = NAN, icpt = NAN;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = NAN;
    bool go = true;
    if (P.af_on) {
        const double af = (double)m / (double)N;
        if (!(P.min_af <= af && af <= P.max_af)) { fl = SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER; go = false; prep = NAN; }
    }
    if (go && (prep > P.pret || !isfinite(prep))) { fl |= SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER; go = false; }
    if (go) {
        double A[PC * (PC + 1) / 2], rhs[PC], xty[PC], e1[PC];
        const double sxy = acc[0] + ymean * (double)m;                        // sum x y = sum x yc + ymean * sum x
        A[sidx(0, 0)] = ZtZ[0]; A[sidx(1, 0)] = (double)m; A[sidx(1, 1)] = (double)m;
        rhs[0] = Zty[0]; rhs[1] = sxy;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            A[sidx(2 + j, 0)] = ZtZ[sidx(1 + j, 0)]; A[sidx(2 + j, 1)] = acc[2 + j]; rhs[2 + j] = Zty[1 + j];
#pragma unroll
            for (int k = 0; k <= j; ++k) A[sidx(2 + j, 2 + k)] = ZtZ[sidx(1 + j, 1 + k)];
        }
#pragma unroll
        for (int a = 0; a < PC; ++a) xty[a] = rhs[a];
        double det;
        const bool ok = ldl_factor<PC>(A, 4.0e-16, &det);
        if (!ok) {                          // rank-deficient design: statsmodels' OLS is pinv-based -> k_glm_ols_pinv finishes it
            const int s2 = atomicAdd(pinv_count, 1); pinv_list[s2] = (int)v;
            out[v] = prep; flags[v] = fl;
            return;
        }
        ldl_solve<PC>(A, rhs);
#pragma unroll
        for (int a = 0; a < PC; ++a) { beta[a] = rhs[a]; e1[a] = (a == 1) ? 1.0 : 0.0; }
        ldl_solve<PC>(A, e1);
        double bxy = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) bxy = fma(beta[a], xty[a], bxy);
        double ssr = yty - bxy;
        if (ssr <= 1e-6 * yty) {                                               // near-perfect fit: walk the samples for the exact residual
            ssr = 0.0;
            for (int sb = 0; sb < NB64; ++sb) {
                const uint64_t w64 = T[(int64_t)sb * Vpad + v];
                const int nb = min(64, N - sb * 64);
                for (int b = 0; b < nb; ++b) {
                    const int i = sb * 64 + b;
                    double f = fma(beta[1], (double)(unsigned)((w64 >> b) & 1ull), beta[0]);
#pragma unroll
                    for (int j = 0; j < Q; ++j) f = fma(beta[2 + j], W[(int64_t)i * Q + j], f);
                    const double r = y[i] - f;
                    ssr = fma(r, r, ssr);
                }
            }
        }
        const double dfr = (double)(N - PC);
        const double scale = ssr / dfr;
        kbse = sqrt(scale * e1[1]);
        kbeta = beta[1]; icpt = beta[0];
        pval = sh_t_sf2(kbeta / kbse, dfr);                                   // res.pvalues[1]
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(kbeta)) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
    }
    out[v] = prep; out[V + v] = pval; out[2 * V + v] = kbeta; out[3 * V + v] = kbse; out[4 * V + v] = icpt;
#pragma unroll
    for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = go ? beta[2 + j] : NAN;
    flags[v] = fl;
}

// OLS slow path for rank-deficient designs (a k-mer that duplicates, or complements, a binary covariate): statsmodels OLS.fit()
// is pinv-based (SM:regression/linear_model.py, method='pinv'): beta = pinv(X) y, normalized_cov = pinv(X) pinv(X)^T = (X^T X)^+,
// df_resid = N - rank(X).  Here (X^T X)^+ comes from a Jacobi eigendecomposition of X^T X; its eigenvalues resolve the singular
// values of X only down to sqrt(eps) * s_max, so directions with lambda <= 1e-10 * lambda_max (s <= 1e-5 * s_max) are the null
// space -- exact collinearity, which is what 0/1 columns against real covariates produce; numpy's own cut-off is s <= 1e-15 s_max.
// One workgroup of 256 threads per listed variant (see k_glm_firth_pinv): thread t takes samples t, t+256, ...
template <int Q>
__global__ __launch_bounds__(256) void k_glm_ols_pinv(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                      const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                      const int *__restrict__ pinv_list, const int *__restrict__ pinv_count,
                                                      double *__restrict__ out, uint32_t *__restrict__ flags)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    __shared__ double s_beta[PC], s_red[4 * (NH + PC)];
    const int cnt = *pinv_count, tid = threadIdx.x, N = P.N;
    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int64_t v = pinv_list[slot];
        double acc[NH + PC];                                        // packed lower X^T X, then X^T y
#pragma unroll
        for (int a = 0; a < NH + PC; ++a) acc[a] = 0.0;
        for (int i = tid; i < N; i += 256) {
            const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
            double x[PC];
            x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
            for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
            const double yi = y[i];
#pragma unroll
            for (int a = 0; a < PC; ++a) {
                acc[NH + a] = fma(x[a], yi, acc[NH + a]);
#pragma unroll
                for (int c = 0; c <= a; ++c) acc[sidx(a, c)] = fma(x[a], x[c], acc[sidx(a, c)]);
            }
        }
        blk_sum<NH + PC>(acc, s_red, tid);
        double Pm[PC * PC];
        int rank = PC;
        if (tid == 0) {
            double A[PC * PC];
            for (int a = 0; a < PC; ++a)
                for (int c = 0; c <= a; ++c) { A[a * PC + c] = acc[sidx(a, c)]; A[c * PC + a] = acc[sidx(a, c)]; }
            slow_pinv<PC>(A, Pm, 1e-10, &rank);
            for (int a = 0; a < PC; ++a) {
                double t = 0.0;
                for (int c = 0; c < PC; ++c) t = fma(Pm[a * PC + c], acc[NH + c], t);
                s_beta[a] = t;
            }
        }
        __syncthreads();
        double ssr[1] = {0.0}, beta[PC];
#pragma unroll
        for (int a = 0; a < PC; ++a) beta[a] = s_beta[a];
        for (int i = tid; i < N; i += 256) {
            const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
            double f = fma(beta[1], (double)(unsigned)((w64 >> (i & 63)) & 1ull), beta[0]);
#pragma unroll
            for (int j = 0; j < Q; ++j) f = fma(beta[2 + j], W[(int64_t)i * Q + j], f);
            const double r = y[i] - f;
            ssr[0] = fma(r, r, ssr[0]);
        }
        blk_sum<1>(ssr, s_red, tid);
        if (tid == 0) {
            const double dfr = (double)(N - rank);
            const double kbse = sqrt(ssr[0] / dfr * Pm[PC + 1]);
            const double pval = sh_t_sf2(beta[1] / kbse, dfr);
            uint32_t fl = flags[v];
            if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
            out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = kbse; out[4 * V + v] = beta[0];
            for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = beta[2 + j];
            flags[v] = fl;
        }
        __syncthreads();                                             // s_beta / s_red are reused by the next variant
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
// ---- OLS by table lookup: which = 0 builds the tables (once per setup), 1 runs a batch ----
template <int Q>
static hipError_t launch_ols_tab(hipStream_t st, int which, const uint64_t *T, int64_t Vpad, int64_t V, const double *y, const double *yc,
                                 const double *W, double *tab, const double *ZtZ, const double *Zty, GlmParams P, double ymean, double yty,
                                 double *out, uint32_t *flags, int *plist, int *pcount)
{
    if (which == 0) hipLaunchKernelGGL(k_glm_ols_build_tab<Q>, dim3((unsigned)P.NB64), dim3(256), 0, st, yc, W, P.N, P.NB64, tab);
    else hipLaunchKernelGGL(k_glm_ols_tab<Q>, dim3((unsigned)(Vpad / 256)), dim3(256), 0, st, T, Vpad, V, y, W, tab, ZtZ, Zty, P, ymean, yty,
                            out, flags, plist, pcount);
    return hipGetLastError();
}

extern "C" hipError_t shk_glm_ols_tab(hipStream_t st, int Q, int which, const uint64_t *T, int64_t Vpad, int64_t V, const double *y,
                                      const double *yc, const double *W, double *tab, const double *ZtZ, const double *Zty, GlmParams P,
                                      double ymean, double yty, double *out, uint32_t *flags, int *plist, int *pcount)
{
#define OLS_CASE(q) case q: return launch_ols_tab<q>(st, which, T, Vpad, V, y, yc, W, tab, ZtZ, Zty, P, ymean, yty, out, flags, plist, pcount);
    switch (Q) {
        OLS_CASE(0) OLS_CASE(1) OLS_CASE(2) OLS_CASE(3) OLS_CASE(4) OLS_CASE(5) OLS_CASE(6) OLS_CASE(7)
        OLS_CASE(8) OLS_CASE(9) OLS_CASE(10) OLS_CASE(11) OLS_CASE(12) OLS_CASE(13) OLS_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef OLS_CASE
}


template <int Q>
static hipError_t launch_ols(hipStream_t st, int which, const uint64_t *T, int64_t Vpad, int64_t V, const double *y, const double *W, const uint64_t *y1,
                             const uint64_t *y0, const double *yc, const double *ZtZ, const double *Zty, GlmParams P, double *out, uint32_t *flags, int *plist, int *pcount)
{
    if (which == 6) hipLaunchKernelGGL(k_glm_ols_pinv<Q>, dim3(512), dim3(256), 0, st, T, Vpad, V, y, W, P, plist, pcount, out, flags);
    else hipLaunchKernelGGL(k_glm_ols<Q>, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, st, T, Vpad, V, y, W, y1, y0, yc, ZtZ, Zty, P, out, flags, plist, pcount);
    return hipGetLastError();
}
// shk_glm_launch's which = 2 (k_glm_ols over the batch) and 6 (k_glm_ols_pinv over the listed variants)
extern "C" hipError_t shk_glm_ols_launch(hipStream_t st, int Q, int which, const uint64_t *T, int64_t Vpad, int64_t V, const double *y, const double *W,
                                         const uint64_t *y1, const uint64_t *y0, const double *yc, const double *ZtZ, const double *Zty, GlmParams P,
                                         double *out, uint32_t *flags, int *plist, int *pcount)
{
#define OLSL_CASE(q) case q: return launch_ols<q>(st, which, T, Vpad, V, y, W, y1, y0, yc, ZtZ, Zty, P, out, flags, plist, pcount);
    switch (Q) {
        OLSL_CASE(0) OLSL_CASE(1) OLSL_CASE(2) OLSL_CASE(3) OLSL_CASE(4) OLSL_CASE(5) OLSL_CASE(6) OLSL_CASE(7)
        OLSL_CASE(8) OLSL_CASE(9) OLSL_CASE(10) OLSL_CASE(11) OLSL_CASE(12) OLSL_CASE(13) OLSL_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef OLSL_CASE
}
