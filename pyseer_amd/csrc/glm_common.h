// glm_common.h -- pieces shared by the register-resident fixed-effects kernels (glm_kernels.hip, q <= 14) and the wide-design
// kernels (glm_wide.hip, q <= 32)
#pragma once
#include "common.h"

__host__ __device__ constexpr int sidx(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, i >= j

#include "glm_params.h"

// Firth step halving: the two noise rules of the default mode are GlmParams.firth_noise / firth_accept (glm_params.h)

__device__ __forceinline__ double logit_cdf(double x) { return 1.0 / (1.0 + exp(-x)); }    // SM Logit.cdf
// SM Logit.loglike of a sample with y = 0: log(cdf(-eta)).  cdf(-eta) = exp(-eta) cdf(eta), so it is log(mu) - eta with the logarithm the
// y = 1 branch needs anyway -- except where mu = cdf(eta) has underflowed (eta < -700: a fit that diverges along a quasi-separating covariate,
// tests/golden/n5000_cap35.npz row 10), where the identity reads -inf + |eta|; there 1 + exp(eta) == 1 and the reference's value is log(1) = 0.
__device__ __forceinline__ double ll_y0(double lm, double eta) { return eta < -700.0 ? 0.0 : lm - eta; }


// ---- a1 prefilter from the packed bits ------------------------------------------------------------------------------
__device__ __forceinline__ double glm_prefilter(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int NB64, int N,
                                                const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                const double *__restrict__ yc, const GlmParams &P, bool *bad, int *mcount, bool *cell1 = nullptr)
{
    int t11 = 0, t01 = 0, m = 0;
    double s1 = 0, q1 = 0;
    for (int sb = 0; sb < NB64; ++sb) {
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        m += __popcll(w); t11 += __popcll(w & y1[sb]); t01 += __popcll(w & y0[sb]);
        if (P.continuous) {
            const int nb = min(64, N - sb * 64);
            for (int b = 0; b < nb; ++b) {
                const double xd = (double)(unsigned)((w >> b) & 1ull), t = yc[sb * 64 + b];
                s1 = fma(xd, t, s1); q1 = fma(xd, t * t, q1);
            }
        }
    }
    *mcount = m;
    *bad = false;
    // (cell1: a cell of the 2 x 2 table holds at most one sample -- the variant all but separates the phenotype; SH_FLAG_FIRTH_SENSITIVE)
    if (cell1) *cell1 = !P.continuous && min(min(t11, P.n1 - t11), min(t01, P.n0 - t01)) <= 1;
    if (P.continuous) return sh_prefilter_welch((double)m, s1, q1, (double)(N - m), P.yc_sum - s1, P.yc_sq - q1);
    return sh_prefilter_binary(t11, P.n1 - t11, t01, P.n0 - t01, bad);
}

// SH_FLAG_FIRTH_SENSITIVE (include/seerhip.h): accepted steps of the fit, and the step norm its stop rule tested (model.py:477-479)
#define FIRTH_SLOW_ITERS 12
__device__ __forceinline__ uint32_t firth_sensitive(int accepted_steps, double tested_step_norm)
{
    return (accepted_steps >= FIRTH_SLOW_ITERS || fabs(tested_step_norm - 1e-4) <= 1e-8) ? SH_FLAG_FIRTH_SENSITIVE : 0u;
}
