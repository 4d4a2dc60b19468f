// glm_lineage.hip -- fit_lineage_effect (model.py:151-199; split off glm_kernels.hip in round 5)
#include "glm_passes.h"

// =====================================================================================================================
// a6 fit_lineage_effect (model.py:151-199): logistic regression of the VARIANT on [1, lineages, covariates] (statsmodels
// Newton, default zero start), returns argmax_j |beta_j| / bse_j over the lineage columns, or -1 (None) on
// PerfectSeparationError / LinAlgError.  Here the whole design row is wave-uniform and the response is the per-lane bit.
// X: N x PC row-major with the intercept in column 0.
// =====================================================================================================================
// LinList (the job stream, job_api.inc): fit only the rows a compacted list names -- the PRINTED rows of a block, list[p] = the row's index in
// the block, *cnt of them -- and of those only the ones the reference reaches fit_lineage_effect with (mode 1, fixed effects: not
// pre-filtered, no firth-fail, model.py:355-382; mode 2, LMM per variant: not pre-filtered, not lrt-filtered, lmm.py:200-213); out[p] in list
// order, -1 for the others.  list == nullptr: rows 0 .. V-1, out[v] (sh_lineage_batch).
template <int PC>
__global__ __launch_bounds__(256) void k_glm_lineage(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N, int NB64,
                                                    const double *__restrict__ X, int nlin, int *__restrict__ out, LinList L)
{
    const XWave xw = xwave();                                        // up to four wavefronts share the 64 variants of a block
    int64_t v = (int64_t)blockIdx.x * 64 + xw.lane;
    bool live = v < V;
    const int64_t slot = v;                                          // where the answer goes
    bool wanted = true;
    if (L.list) {
        const long long n = *L.cnt;
        if ((int64_t)blockIdx.x * 64 >= n) return;                   // (the whole block: the launch is sized for every row of the block)
        live = slot < n;
        v = live ? (int64_t)L.list[slot] : 0;
        wanted = live && lin_wanted(L.flags[v], L.mode);
    }
    const int64_t vr = live ? v : 0;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = 0.0;
    int it = 0, status = 0, best = -1;
    bool fin = false, active = live && wanted;
    const double nobs = (double)N;
    while (__any(active)) {                                          // `active` is kept identical in all the waves of a block
        double H[PC * (PC + 1) / 2], g[PC], maxdev = 0.0, unused = 0.0;
#pragma unroll
        for (int a = 0; a < PC * (PC + 1) / 2; ++a) H[a] = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) g[a] = 0.0;
        if (active) {
            for (int sb = xw.w; sb < NB64; sb += xw.S) {
                const uint64_t w64 = T[(int64_t)sb * Vpad + vr];
                const int nb = min(64, N - sb * 64);
                for (int b = 0; b < nb; ++b) {
                    const int i = sb * 64 + b;
                    double x[PC];
#pragma unroll
                    for (int a = 0; a < PC; ++a) x[a] = X[(int64_t)i * PC + a];
                    double eta = 0.0;
#pragma unroll
                    for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
                    const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
                    const double r = (double)(unsigned)((w64 >> b) & 1ull) - mu;
                    maxdev = fmax(maxdev, fabs(r));
#pragma unroll
                    for (int a = 0; a < PC; ++a) {
                        g[a] = fma(r, x[a], g[a]);
                        const double wa = wgt * x[a];
#pragma unroll
                        for (int c = 0; c <= a; ++c) H[sidx(a, c)] = fma(wa, x[c], H[sidx(a, c)]);
                    }
                }
            }
        }
        xw_sum(xw, H); xw_sum(xw, g); xw_sum_max(xw, unused, maxdev);
        if (active && xw.w == 0) {
#pragma unroll
            for (int a = 0; a < PC * (PC + 1) / 2; ++a) H[a] = H[a] / nobs;
            double det;
            if (it > 0 && maxdev <= 1e-8) { status = 1; active = false; }
            else if (fin) {
                // numpy.linalg.inv only fails on an EXACT zero pivot; a numerically rank-deficient Hessian (quasi-separation after
                // 35 iterations) yields huge/NaN standard errors and the argmax simply moves on (np.argmax: first NaN wins).
                if (!ldl_factor<PC>(H, 0.0, &det)) status = 2;
                else {
                    double bestw = -1.0; int first_nan = -1;
#pragma unroll
                    for (int a = 1; a < PC; ++a) {
                        double e[PC];
#pragma unroll
                        for (int c = 0; c < PC; ++c) e[c] = (c == a) ? 1.0 : 0.0;
                        ldl_solve<PC>(H, e);
                        const double wald = fabs(beta[a]) / sqrt(e[a] / nobs);
                        if (a <= nlin) {
                            if (isnan(wald)) { if (first_nan < 0) first_nan = a - 1; }
                            else if (wald > bestw) { bestw = wald; best = a - 1; }
                        }
                    }
                    if (first_nan >= 0) best = first_nan;
                }
                active = false;
            } else {
#pragma unroll
                for (int a = 0; a < PC; ++a) { H[sidx(a, a)] -= 1e-10; g[a] = g[a] / nobs; }
                if (!ldl_factor<PC>(H, 0.0, &det)) { status = 2; active = false; }
                else {
                    ldl_solve<PC>(H, g);
                    bool moving = false;
#pragma unroll
                    for (int a = 0; a < PC; ++a) { beta[a] += g[a]; moving = moving || (fabs(g[a]) > 1e-8); }
                    ++it;
                    if (!moving || it >= 35) fin = true;
                }
            }
        }
        xw_bcast(xw, beta, active);
    }
    if (live && xw.w == 0) out[slot] = (status == 0 && wanted) ? best : -1;
}

extern "C" hipError_t shk_glm_lineage(hipStream_t st, int PC, const uint64_t *T, int64_t Vpad, int64_t V, int N, int NB64,
                                      const double *X, int nlin, int *out, LinList L)
{
    const int S = std::min(4, glm_split_waves(NB64));
    const dim3 grid((unsigned)((V + 63) / 64)), blk(64 * S);
#define LIN_CASE(p) case p: hipLaunchKernelGGL(k_glm_lineage<p>, grid, blk, glm_split_lds(S), st, T, Vpad, V, N, NB64, X, nlin, out, L); break;
    switch (PC) {
        LIN_CASE(2) LIN_CASE(3) LIN_CASE(4) LIN_CASE(5) LIN_CASE(6) LIN_CASE(7) LIN_CASE(8) LIN_CASE(9) LIN_CASE(10)
        LIN_CASE(11) LIN_CASE(12) LIN_CASE(13) LIN_CASE(14) LIN_CASE(15) LIN_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef LIN_CASE
    return hipGetLastError();
}

