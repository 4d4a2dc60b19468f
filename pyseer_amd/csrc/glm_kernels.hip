// glm_kernels.hip -- fixed-effects per-variant regression on MI355X (gfx950).
//
// Reference path restated here:
//   pyseer/model.py:202-394  fixed_effects_regression   (orchestration, notes, filters)
//   statsmodels Logit.fit(method='newton')               (SM:base/optimizer.py:407-446, SM:base/model.py:497-534)
//   pyseer/model.py:397-504  firth_likelihood / fit_firth
//   statsmodels OLS.fit()                                (model.py:299-312)
//
// Mapping: ONE VARIANT PER LANE.  The design matrix is [1, k, W] where only the k column differs between variants,
// so a wavefront walks the samples in lock-step: the sample's phenotype and covariate row are wave-uniform (they
// arrive through the scalar cache as SGPR operands), the variant bit, eta, mu, the score vector and the p x p
// information matrix are per-lane registers.  No cross-lane reduction, no LDS, no atomics on the hot path; the p x p
// solves (LDL^T) run per lane, fully unrolled in registers.  Variants that need Firth (bad-chisq, high-bse, separation,
// singular) are appended to a list and handled by the Firth kernel with the same mapping.
#include <algorithm>
#include "common.h"

#define GLM_MAXQ 14
#ifndef GLM_FAST_WAVES
#define GLM_FAST_WAVES 2
#endif

#include "glm_common.h"
#include "glm_device.h"

// ---- one pass over the samples at beta: X^T W X (packed), optional score, log-likelihood, max |mu - y| --------------
// column order of the design: 0 = intercept, 1 = variant, 2.. = W columns (model.py:286-297)
template <int Q, bool SCORE, bool LOGLIK>
__device__ __forceinline__ void info_pass(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64,
                                          const double *__restrict__ y, const double *__restrict__ W,
                                          const double (&beta)[Q + 2], double (&H)[(Q + 2) * (Q + 3) / 2],
                                          double (&g)[Q + 2], double &ll, double &maxdev, bool want_ll = true,
                                          int sb0 = 0, int sbs = 1)
{
    constexpr int P = Q + 2;
#pragma unroll
    for (int a = 0; a < P * (P + 1) / 2; ++a) H[a] = 0.0;
    if (SCORE) {
#pragma unroll
        for (int a = 0; a < P; ++a) g[a] = 0.0;
    }
    ll = 0.0; maxdev = 0.0;
    for (int sb = sb0; sb < NB64; sb += sbs) {                                 // (sb0, sbs) = (wave, waves) of a sample-split block
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
        for (int b = 0; b < nb; ++b) {
            const int i = sb * 64 + b;
            const double xd = (double)(unsigned)((w64 >> b) & 1ull);
            const double yi = y[i];
            double z[Q > 0 ? Q : 1];
#pragma unroll
            for (int j = 0; j < Q; ++j) z[j] = W[(int64_t)i * Q + j];
            double eta = fma(beta[1], xd, beta[0]);
#pragma unroll
            for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], z[j], eta);
            const double en = exp(-eta);
            const double mu = 1.0 / (1.0 + en);                               // SM Logit.cdf
            const double wgt = mu * (1.0 - mu);
            const double r = yi - mu;
            maxdev = fmax(maxdev, fabs(r));
            if (LOGLIK && want_ll) {
                // SM Logit.loglike: log(cdf(q*eta)), q = 2y-1.  y = 1: log(mu).  y = 0: cdf(-eta) = exp(-eta) * cdf(eta), so
                // log(cdf(-eta)) = log(mu) - eta: one logarithm, no second reciprocal, and no cancellation when mu -> 1.
                const double lm = log(mu);
                ll += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
            }
            if (SCORE) {
                g[0] += r; g[1] = fma(r, xd, g[1]);
#pragma unroll
                for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, z[j], g[2 + j]);
            }
            const double wx = wgt * xd;
            H[sidx(0, 0)] += wgt;
            H[sidx(1, 0)] += wx;                       // H11 == H10 (xd^2 == xd), filled in after the loop
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const double wz = wgt * z[j];
                H[sidx(2 + j, 0)] += wz;
                H[sidx(2 + j, 1)] = fma(wx, z[j], H[sidx(2 + j, 1)]);
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = fma(wz, z[k], H[sidx(2 + j, 2 + k)]);
            }
        }
    }
    H[sidx(1, 1)] = H[sidx(1, 0)];
}

// ---- info_pass for y in {0, 1}, leaner and with the sample's record (covariates, y) fetched one sample ahead ----------------------
// Same sums as info_pass<Q, false, true> (information matrix, log-likelihood, max |y - mu|), all fp64.  What changes is how a sample's
// mu and log-likelihood term are formed (see k_glm_ll): t = exp(-|eta|) by exp_neg, 1 / (1 + t) by v_rcp_f64 + two Newton steps, and
// ll_i = -max(a_i, 0) - log(1 + t) with the logs folded into a running product (one log per wavefront at the end).  R: per sample Q
// covariates then y (GlmParams.rec_o, the covariates as given).
template <int Q>
__device__ __forceinline__ void info_pass_bin(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64,
                                              const double *__restrict__ R, const double (&beta)[Q + 2],
                                              double (&H)[(Q + 2) * (Q + 3) / 2], double &ll, double &maxdev, int sb0, int sbs)
{
    constexpr int P = Q + 2, RS = Q + 1;
#pragma unroll
    for (int a = 0; a < P * (P + 1) / 2; ++a) H[a] = 0.0;
    maxdev = 0.0;
    double apos = 0.0, prod = 1.0;
    int pexp = 0;
    auto one = [&](const double (&rc)[RS], bool xb) {
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double yi = rc[Q];
        const double t = exp_neg(fabs(eta)), u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        const double mu = (eta >= 0.0) ? inv : t * inv;
        const double wgt = t * inv * inv;                                         // mu (1 - mu) = t / (1 + t)^2, without the cancellation
        maxdev = fmax(maxdev, fabs(yi - mu));
        apos += fmax(fma(-2.0 * yi, eta, eta), 0.0);
        prod *= u;
        const double wx = xb ? wgt : 0.0;
        H[sidx(0, 0)] += wgt;
        H[sidx(1, 0)] += wx;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const double wz = wgt * rc[j];
            H[sidx(2 + j, 0)] += wz;
            H[sidx(2 + j, 1)] = fma(wx, rc[j], H[sidx(2 + j, 1)]);
#pragma unroll
            for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = fma(wz, rc[k], H[sidx(2 + j, 2 + k)]);
        }
    };
    double ra[RS], rb[RS];
    {
        const int i0 = min(sb0 * 64, N - 1);
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i0 * RS + k];
    }
    for (int sb = sb0; sb < NB64; sb += sbs) {                                 // (sb0, sbs) = (wave, waves) of a sample-split block
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
        if (nb == 64) {
            const int inext = min((sb + sbs) * 64, N - 1);                    // first sample of this wavefront's next word
            for (int b = 0; b < 64; b += 2) {
                const int i = sb * 64 + b, i2 = (b == 62) ? inext : i + 2;
                const int za = pipe_zero(ra[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                one(ra, (w64 >> b) & 1ull);
                const int zb = pipe_zero(rb[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                one(rb, (w64 >> (b + 1)) & 1ull);
            }
        } else {
            for (int b = 0; b < nb; ++b) {
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(sb * 64 + b) * RS + k];
                one(ra, (w64 >> b) & 1ull);
            }
        }
        int e2; prod = frexp(prod, &e2); pexp += e2;
    }
    H[sidx(1, 1)] = H[sidx(1, 0)];
    ll = -(apos + fma((double)pexp, 0.6931471805599453, log(prod)));
}

// =====================================================================================================================
// Logistic Newton (binary phenotype) -- one variant per lane
//
// Three phases per wavefront, all following statsmodels' Newton (start vector, ridge, |d beta|_inf <= 1e-8, 35 iterations,
// separation callback after every update):
//  A. fast iterations: score and mu in fp64 (the fixed point of the iteration is the exact MLE), the Hessian accumulated in
//     fp32 (77 accumulators in 77 VGPRs instead of 154; fp32 FMAs at twice the fp64 rate).  An inexact Hessian only changes the
//     path, not the limit: with X^T W X good to ~1e-5 the error contracts by ~1e-5 per step once inside the quadratic basin, and
//     the stopping rule is the reference's.  A lane leaves phase A as "converged" only if it converges within 12 iterations with
//     healthy pivots and no separation signal;
//  B. anything else (separation, divergence, ill-conditioning, slow convergence) is RESTARTED from the start vector with the
//     all-fp64 iteration, which reproduces the reference's trajectory and therefore its notes (matrix-inversion-error vs
//     perfectly-separable-data vs high-bse are decided by that trajectory);
//  C. one fp64 pass at the final beta for llf, the separation check and bse[1] = sqrt((X^T W X)^-1_11) (no ridge).
// =====================================================================================================================
template <int Q>
__device__ __forceinline__ void fast_pass(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64,
                                          const double *__restrict__ y, const double *__restrict__ W,
                                          const float *__restrict__ Wf, const double (&beta)[Q + 2],
                                          float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], double &maxdev)
{
    constexpr int P = Q + 2;
#pragma unroll
    for (int a = 0; a < P * (P + 1) / 2; ++a) H[a] = 0.0f;
#pragma unroll
    for (int a = 0; a < P; ++a) g[a] = 0.0;
    maxdev = 0.0;
    for (int sb = 0; sb < NB64; ++sb) {
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
#pragma unroll 2
        for (int b = 0; b < nb; ++b) {
            const int i = sb * 64 + b;
            const bool xb = (w64 >> b) & 1ull;
            const double xd = xb ? 1.0 : 0.0;
            double eta = fma(beta[1], xd, beta[0]);
#pragma unroll
            for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], W[(int64_t)i * Q + j], eta);
            const double mu = 1.0 / (1.0 + exp(-eta));
            const double r = y[i] - mu;
            maxdev = fmax(maxdev, fabs(r));
            g[0] += r; g[1] += xb ? r : 0.0;
#pragma unroll
            for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, W[(int64_t)i * Q + j], g[2 + j]);
            const float wf = (float)(mu * (1.0 - mu));
            const float wx = xb ? wf : 0.0f;
            H[sidx(0, 0)] += wf;
            H[sidx(1, 0)] += wx;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const float zj = Wf[(int64_t)i * Q + j];
                const float wz = wf * zj;
                H[sidx(2 + j, 0)] += wz;
                H[sidx(2 + j, 1)] = fmaf(wx, zj, H[sidx(2 + j, 1)]);
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = fmaf(wz, Wf[(int64_t)i * Q + k], H[sidx(2 + j, 2 + k)]);
            }
        }
    }
    H[sidx(1, 1)] = H[sidx(1, 0)];
}

// ---- the same pass with the covariate block of the fp32 Hessian on the matrix pipe --------------------------------------
// H = sum_i w_i x_i x_i^T with x_i = [1, k_i, z_i].  The Q(Q+1)/2 entries H(2+j,2+k) = sum_i w_i (z_ij z_ik) are a skinny GEMM
//   [64 variants x samples](w, per lane)  x  [samples x Q(Q+1)/2](z_j z_k, wave-uniform: a per-run table)
// and go to v_mfma_f32_32x32x2_f32, two samples per issue: A = 32 variants x 2 samples.  A lane's w for samples (i, i+1) sit in
// two registers; one v_permlane32_swap turns them into the A operands of the two 32-variant halves.  The 2 + 2Q entries that
// involve the intercept or the variant column stay on the VALU (fp32), as do eta, mu and the fp64 score.  At the end of the pass
// the 32x32 C tiles go through LDS once so that every lane holds its own variant's packed H.
typedef float v16f __attribute__((ext_vector_type(16)));
template <int Q> struct FastCols {
    static constexpr int NPROD = Q * (Q + 1) / 2;
    static constexpr int NCB = (NPROD + 31) / 32 > 0 ? (NPROD + 31) / 32 : 1;      // 32-column blocks of the products table
    static constexpr int STRIDE = NCB * 32;                                          // floats per sample in the table
    static constexpr int LDS_FLOATS = 32 * (STRIDE + 1);
};

template <int Q, bool F32>
__device__ __forceinline__ void fast_pass_mfma(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64,
                                               const double *__restrict__ y, const double *__restrict__ W,
                                               const float *__restrict__ Wf, const float *__restrict__ ZZ,
                                               const double (&beta)[Q + 2], float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2],
                                               double &maxdev, float *tr)
{
    constexpr int P = Q + 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    v16f acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][h][r] = 0.0f;
    float h00 = 0.0f, h10 = 0.0f, hz0[Q > 0 ? Q : 1], hz1[Q > 0 ? Q : 1];
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = 0.0f; hz1[j] = 0.0f; }
#pragma unroll
    for (int a = 0; a < P; ++a) g[a] = 0.0;
    maxdev = 0.0;
    // F32: the whole sample in single precision (one v_exp_f32 instead of the fp64 exp sequence, half-cost FMAs).  Used for the first
    // Newton steps only; they need to land within ~1e-4 of the optimum, and the fp64 steps that follow set the fixed point.
    float bf[P], gf[P];
#pragma unroll
    for (int a = 0; a < P; ++a) { bf[a] = (float)beta[a]; gf[a] = 0.0f; }
    auto sample = [&](int i, bool xb) -> float {
        if constexpr (F32) {
            float eta = bf[0] + (xb ? bf[1] : 0.0f);
#pragma unroll
            for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j], Wf[(int64_t)i * Q + j], eta);
            const float mu = 1.0f / (1.0f + __expf(-eta));
            const float r = (float)y[i] - mu;
            gf[0] += r; gf[1] += xb ? r : 0.0f;
            const float wf = mu * (1.0f - mu);
            const float wx = xb ? wf : 0.0f;
            h00 += wf; h10 += wx;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const float zj = Wf[(int64_t)i * Q + j];
                gf[2 + j] = fmaf(r, zj, gf[2 + j]); hz0[j] = fmaf(wf, zj, hz0[j]); hz1[j] = fmaf(wx, zj, hz1[j]);
            }
            return wf;
        }
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], W[(int64_t)i * Q + j], eta);
        // eta and the score sums in fp64, the logistic function itself in fp32 (v_exp_f32 + v_rcp_f32 instead of ~45 fp64 operations): each mu
        // carries an independent ~6e-8 relative rounding, the score its sum over N (~5e-6 against a Hessian ~N/5): beta to ~1e-8, which the
        // final pass' exact fp64 Newton step absorbs.  -DGLM_MU64 builds the fp64 evaluation instead (A/B).
#ifdef GLM_MU64
        const double mu = 1.0 / (1.0 + exp(-eta));
#else
        const double mu = (double)(1.0f / (1.0f + __expf(-(float)eta)));
#endif
        const double r = y[i] - mu;
        maxdev = fmax(maxdev, fabs(r));
        g[0] += r; g[1] += xb ? r : 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, W[(int64_t)i * Q + j], g[2 + j]);
        const float wf = (float)(mu * (1.0 - mu));
        const float wx = xb ? wf : 0.0f;
        h00 += wf; h10 += wx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float zj = Wf[(int64_t)i * Q + j]; hz0[j] = fmaf(wf, zj, hz0[j]); hz1[j] = fmaf(wx, zj, hz1[j]); }
        return wf;
    };
    // (Fetching the next pair's wave-uniform rows a pair ahead was tried: 128 SGPR spills, 25 % slower.)
    if (Q > 0) {
        const int nfull = N >> 1;                                             // pairs (2p, 2p+1); a pair never straddles a 64-sample word
        for (int pr = 0; pr < nfull; ++pr) {
            const int i = 2 * pr, b = i & 63;
            const float *zrow = ZZ + (int64_t)(i + lh) * STRIDE + l31;
            float bz[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bz[cb] = zrow[cb * 32];
            const uint64_t w = T[(int64_t)(i >> 6) * Vpad + v];
            const float wf0 = sample(i, (w >> b) & 1ull);
            const float wf1 = sample(i + 1, (w >> (b + 1)) & 1ull);
            // A operands: lanes 0-31 = sample i, lanes 32-63 = sample i+1, for the variants of each 32-lane half
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wf0), __float_as_uint(wf1), false, false);
            const float a0 = __uint_as_float(sw[0]), a1 = __uint_as_float(sw[1]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bz[cb], acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bz[cb], acc[cb][1], 0, 0, 0);
            }
        }
        if (N & 1) {                                                          // the odd sample: k = 1 rows of A are zero
            const int i = N - 1;
            const float *zrow = ZZ + (int64_t)i * STRIDE + l31;
            const float wf0 = sample(i, (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wf0), 0u, false, false);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), zrow[cb * 32], acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), zrow[cb * 32], acc[cb][1], 0, 0, 0);
            }
        }
    } else {
        for (int i = 0; i < N; ++i) sample(i, (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
    }
    if constexpr (F32) {
#pragma unroll
        for (int a = 0; a < P; ++a) g[a] = (double)gf[a];
        maxdev = 1.0;                                                         // no separation verdict from a single-precision pass
    }
    H[sidx(0, 0)] = h00; H[sidx(1, 0)] = h10; H[sidx(1, 1)] = h10;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0[j]; H[sidx(2 + j, 1)] = hz1[j]; }
    // C layout of the 32x32 tile: row (variant) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = lane & 31
    if (Q > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + cb * 32 + l31] = acc[cb][h][r];
            __syncthreads();
            if (lh == h) {
                const float *row = tr + l31 * (STRIDE + 1);
#pragma unroll
                for (int j = 0; j < Q; ++j)
#pragma unroll
                    for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = row[j * (j + 1) / 2 + k];
            }
        }
    }
}

// ---- the final pass with the covariate block of the information matrix as a CORRECTION to the null model's -------------------
// k_glm_final needs X^T W X in double precision at the final beta: 78 fp64 accumulators per lane at q = 10, which do not fit beside the
// score and spill (468-708 bytes of scratch per lane, 11-14 ms per 262 144 variants at N = 5000).  But the [1, z] x [1, z] part of it
// differs from the NULL model's -- a per-run constant, A0 = sum_i w0_i zz_i, summed once on the host in fp64 -- only through
// w_i - w0_i, which is small (the variant's effect) and whose sum tolerates single precision:  sum_i (w_i - w0_i) z_ij z_ik  carries a
// relative rounding of ~1e-7 of a term that is itself a few per cent of A0.  So the z x z block is accumulated exactly as in the fast
// passes (v_mfma_f32_32x32x2_f32 against the per-run products table) with A operand (float)(w_i - w0_i), and only the intercept and
// variant rows (2 + 2q entries) stay fp64 accumulators next to the score.  Works in the standardised coordinates of the fast phase.
template <int Q>
__device__ __forceinline__ void final_pass_mfma(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N,
                                                const double *__restrict__ y, const double *__restrict__ Ws, const float *__restrict__ ZZ,
                                                const double *__restrict__ w0, const double *__restrict__ a0, const double (&beta)[Q + 2],
                                                double (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], double &ll, double &maxdev, float *tr)
{
    constexpr int P = Q + 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    v16f acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][h][r] = 0.0f;
    double h00 = 0.0, h10 = 0.0, hz0[Q > 0 ? Q : 1], hz1[Q > 0 ? Q : 1];
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = 0.0; hz1[j] = 0.0; }
#pragma unroll
    for (int a = 0; a < P; ++a) g[a] = 0.0;
    ll = 0.0; maxdev = 0.0;
    auto sample = [&](int i, bool xb) -> float {
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], Ws[(int64_t)i * Q + j], eta);
        const double yi = y[i];
        const double mu = 1.0 / (1.0 + exp(-eta));                            // SM Logit.cdf
        const double r = yi - mu;
        maxdev = fmax(maxdev, fabs(r));
        const double lm = log(mu);                                            // SM Logit.loglike, as info_pass
        ll += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
        g[0] += r; g[1] += xb ? r : 0.0;
        const double wgt = mu * (1.0 - mu), wx = xb ? wgt : 0.0;
        h00 += wgt; h10 += wx;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const double zj = Ws[(int64_t)i * Q + j];
            g[2 + j] = fma(r, zj, g[2 + j]); hz0[j] = fma(wgt, zj, hz0[j]); hz1[j] = fma(wx, zj, hz1[j]);
        }
        return (float)(wgt - w0[i]);
    };
    const int nfull = N >> 1;
    for (int pr = 0; pr < nfull; ++pr) {
        const int i = 2 * pr, b = i & 63;
        const float *zrow = ZZ + (int64_t)(i + lh) * STRIDE + l31;
        float bz[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bz[cb] = zrow[cb * 32];
        const uint64_t w = T[(int64_t)(i >> 6) * Vpad + v];
        const float d0 = sample(i, (w >> b) & 1ull);
        const float d1 = sample(i + 1, (w >> (b + 1)) & 1ull);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false);
        const float a0f = __uint_as_float(sw[0]), a1f = __uint_as_float(sw[1]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0f, bz[cb], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1f, bz[cb], acc[cb][1], 0, 0, 0);
        }
    }
    if (N & 1) {
        const int i = N - 1;
        const float *zrow = ZZ + (int64_t)i * STRIDE + l31;
        const float d0 = sample(i, (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), 0u, false, false);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), zrow[cb * 32], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), zrow[cb * 32], acc[cb][1], 0, 0, 0);
        }
    }
    H[sidx(0, 0)] = h00; H[sidx(1, 0)] = h10; H[sidx(1, 1)] = h10;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0[j]; H[sidx(2 + j, 1)] = hz1[j]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + cb * 32 + l31] = acc[cb][h][r];
        __syncthreads();
        if (lh == h) {
            const float *row = tr + l31 * (STRIDE + 1);
#pragma unroll
            for (int j = 0; j < Q; ++j)
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = a0[j * (j + 1) / 2 + k] + (double)row[j * (j + 1) / 2 + k];
        }
    }
}

// beta workspace: SoA, bw[a * Vpad + v]; state[v]: 0 = nothing more to fit here, 1 = beta ready for the final pass
struct GlmWork { double *bw; int *state; int *slow_list; int *slow_count; int *tile_list; int *tile_count; };

// ---- kernel 1: a1 prefilter + routing + phase A (fast Newton) ---------------------------------------------------------------
template <int Q, bool CHORD>
__global__ __launch_bounds__(CHORD ? 256 : 64, CHORD ? 1 : GLM_FAST_WAVES) void k_glm_fast(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                 const double *__restrict__ y, const double *__restrict__ W,
                                                 const float *__restrict__ Wf,
                                                 const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                 const double *__restrict__ yc, GlmParams P, GlmWork wk,
                                                 double *__restrict__ out, uint32_t *__restrict__ flags,
                                                 int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    constexpr int PC = Q + 2;
    const int64_t v = (int64_t)blockIdx.x * (CHORD ? 256 : 64) + threadIdx.x;   // (the classifier of the rounds: four wavefronts per workgroup, list_push_block)
    __shared__ int push_lds[CHORD ? 17 : 1];
    const bool live = v < V;
    const int64_t vr = live ? v : 0;           // dead lanes shadow variant 0 and never write
    const int N = P.N, NB64 = P.NB64;
    const double nobs = (double)N;
    uint32_t fl = 0;
    bool want_fit = live, to_firth = false, bad = false, cell1 = false;
    int m = 0;
    double prep = glm_prefilter(T, Vpad, vr, NB64, N, y1, y0, yc, P, &bad, &m, &cell1);
    if (P.af_on) {
        const double af = (double)m / (double)N;
        if (!(P.min_af <= af && af <= P.max_af)) { fl = SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER; want_fit = false; prep = NAN; }
    }
    if (want_fit) {
        if (bad) fl |= SH_NOTE_BAD_CHISQ;
        if (prep > P.pret || !isfinite(prep)) { fl |= SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER; want_fit = false; }   // model.py:266 (>)
        else if (cell1) fl |= SH_FLAG_FIRTH_SENSITIVE;               // a (quasi-)separating variant on its way to fit_firth: include/seerhip.h
    }
    if (want_fit && (bad || P.force_firth)) { to_firth = true; want_fit = false; }

    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = 0.0;
    beta[0] = P.ymean_logit;
    if (P.warm_on) {                                                 // null-model MLE (glm_params.h): [b0, 0, bz] in this phase's coordinates
        beta[0] = P.warm[0];
#pragma unroll
        for (int j = 0; j < Q; ++j) beta[2 + j] = P.warm[1 + j];
    }
    bool need_slow = want_fit && (P.newton_mode == 1);
    bool active = want_fit && !need_slow;
    int it = 0, pass = 0;
    // CHORD: this kernel only classifies; the fits run as rounds of lean kernels over lists (k_glm_pass32 ... k_glm_chord below)
    const bool chord_go = CHORD && active;
    if (CHORD) {
        if (chord_go) {
#pragma unroll
            for (int a = 0; a < PC; ++a) P.ch_bs[(int64_t)a * Vpad + v] = beta[a];
        }
        list_push_block(chord_go, P.ch_list[0], P.ch_cnt, (int)v, push_lds);
        active = false;
    }
    // Newton's iteration is affine invariant, so this phase runs on covariates standardised per column (Wf and the products table
    // are built from them too): a column like "year of isolation" (2000 +- 10) would otherwise defeat the fp32 Hessian and send every
    // variant to the fp64 restart.  Same start vector (the slopes start at 0), same fixed point; beta is mapped back at the end.
    const double *__restrict__ Wx = P.ws ? P.ws : W;
    __shared__ float tr[CHORD ? 1 : FastCols<Q>::LDS_FLOATS];
    while (!CHORD && __any(active)) {
        float Hf[PC * (PC + 1) / 2];
        double g[PC], maxdev;
        // the matrix-pipe pass is wave-wide (permlane swap, MFMA): lanes that already stopped ride along with their frozen beta.
        // The first P.f32_steps passes of a wavefront run entirely in single precision (all its lanes start together, so a lane's
        // step count is the wavefront's pass count); a lane cannot be declared converged by such a pass.
        const bool f32 = pass < P.f32_steps;
        if (P.zz) {
            if (f32) fast_pass_mfma<Q, true>(T, Vpad, vr, N, NB64, y, Wx, Wf, P.zz, beta, Hf, g, maxdev, tr);
            else fast_pass_mfma<Q, false>(T, Vpad, vr, N, NB64, y, Wx, Wf, P.zz, beta, Hf, g, maxdev, tr);
        }
        ++pass;
        if (active) {
            if (!P.zz) fast_pass<Q>(T, Vpad, vr, N, NB64, y, Wx, Wf, beta, Hf, g, maxdev);
            if (it > 0 && maxdev <= 1e-8) { need_slow = true; active = false; }
            else {
                double A[PC * (PC + 1) / 2];
#pragma unroll
                for (int a = 0; a < PC * (PC + 1) / 2; ++a) A[a] = (double)Hf[a] / nobs;
#pragma unroll
                for (int a = 0; a < PC; ++a) { A[sidx(a, a)] -= 1e-10; g[a] = g[a] / nobs; }
                double det;
                if (!ldl_factor<PC>(A, 1e-4, &det)) { need_slow = true; active = false; }          // fp32 cannot resolve this design
                else {
                    ldl_solve<PC>(A, g);
                    bool moving = false, finite = true;
#pragma unroll
                    for (int a = 0; a < PC; ++a) { beta[a] += g[a]; moving = moving || (fabs(g[a]) > P.fast_tol); finite = finite && isfinite(beta[a]); }
                    ++it;
                    if (!finite) { need_slow = true; active = false; }
                    else if (f32 && P.zz) {                                                         // single-precision pass: its own stopping rule
                        double stp = 0.0;
#pragma unroll
                        for (int a = 0; a < PC; ++a) stp = fmax(stp, fabs(g[a]));
                        if (stp <= P.f32_tol) active = false;
                        else if (it >= 12) { need_slow = true; active = false; }
                    }
                    else if (!moving) active = false;                                               // converged (fp64 score)
                    else if (it >= 12) { need_slow = true; active = false; }
                }
            }
        }
    }
    if (P.dbg) {
        if (threadIdx.x == 0) { atomicAdd(&P.dbg[0], pass); atomicAdd(&P.dbg[1], 1); }
        if (want_fit && live) { atomicAdd(&P.dbg[2], it); atomicAdd(&P.dbg[3], 1); }
    }
    if (!live) return;
    out[v] = prep; out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
#pragma unroll
    for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = NAN;
    flags[v] = fl;
    wk.state[v] = (want_fit && !need_slow && !chord_go) ? 1 : 0;
    if (want_fit && !need_slow && !chord_go) {
        if (P.ws) {                                                  // z' = (z - mean) / scale  =>  b = b' / scale, b0 = b0' - sum b' mean / scale
#pragma unroll
            for (int j = 0; j < Q; ++j) { beta[2 + j] = beta[2 + j] / P.wstd[Q + j]; beta[0] = fma(-beta[2 + j], P.wstd[j], beta[0]); }
        }
#pragma unroll
        for (int a = 0; a < PC; ++a) wk.bw[(int64_t)a * Vpad + v] = beta[a];
    }
    if (need_slow) { const int slot = atomicAdd(wk.slow_count, 1); wk.slow_list[slot] = (int)v; }
    if (to_firth) { const int slot = atomicAdd(firth_count, 1); firth_list[slot] = (int)v; }
}

// ---- a single-precision pass with the two samples of an MFMA issue as the halves of packed-fp32 operations -------------------------
// v_pk_fma_f32 and friends do two fp32 operations per lane per issue; an unpacked fp32 instruction costs the same four cycles per
// wavefront as an fp64 one.  Samples (2p, 2p + 1) feed one MFMA issue anyway, so eta, mu, w and every VALU accumulator are carried as
// float2 = (even sample, odd sample) and their sums are folded at the end of the pass.  Needs the covariates in pair layout
// (GlmParams.wfp: [pair][Q] float2) and y, w0 as float arrays (yf, w0f).  DELTA: the A operand is w - w0 and no score is formed (k_glm_dpass).
typedef float v2f __attribute__((ext_vector_type(2)));
// the logistic function on the transcendental unit alone: v_exp_f32 and v_rcp_f32 (1 ulp each) instead of expf's range fix-ups and an IEEE
// division (~12 instructions).  exp2 overflows to inf -> mu = 0, underflows to 0 -> mu = 1: the right limits.
__device__ __forceinline__ float sigmoid_fast(float eta) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(eta * -1.4426950408889634f)); }
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
template <int Q, bool DELTA>
__device__ __forceinline__ void pass32_pk_f32(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, const GlmParams &P, const float *__restrict__ Wf,
                                          const double (&beta)[Q + 2], float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], float *tr,
                                          int part = 0, int nparts = 1, double *hdl = nullptr)
{
    // hdl (DELTA): [2 + 2Q][64] doubles of LDS, zeroed by the caller.  The intercept and variant rows of X^T (W - W0) X are summed in single
    // precision WITHIN a 64-sample word and in double precision across words: for a variant with a large effect (|beta| ~ 4) the differences
    // w - w0 of its carriers are ~0.2 each, their running sum reaches hundreds, and 2500 fp32 additions at that magnitude put 1e-6 on the
    // variant's own diagonal entry, i.e. on bse (measured on near-separating variants at N = 5000: 1.7e-6 against the fp64 restatement).
    // part / nparts: this wavefront's share of the samples (k_glm_pass32_split): whole 64-sample words [wd0, wd1); the last part also takes
    // the partial word and the odd sample
    constexpr int PC = Q + 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE;
    const int N = P.N;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    const float *__restrict__ ZZ = P.zz;
    // per pair of samples one wave-uniform record: z (Q float2), y pair, w0 pair -- fetched one pair AHEAD (scalar loads otherwise sit
    // exposed in front of every pair: ~900 cycles per pair and wavefront, measured), as is the lane's B operand of the MFMA
    constexpr int RS = Q + 2;
    const v2f *__restrict__ Rp = (const v2f *)P.wfp;
    v16f acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][h][r] = 0.0f;
    v2f bf[PC], gf[PC], h00 = {0.0f, 0.0f}, h10 = {0.0f, 0.0f}, hz0[Q > 0 ? Q : 1], hz1[Q > 0 ? Q : 1];
#pragma unroll
    for (int a = 0; a < PC; ++a) { const float b = (float)beta[a]; bf[a] = v2f{b, b}; gf[a] = v2f{0.0f, 0.0f}; }
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = v2f{0.0f, 0.0f}; hz1[j] = v2f{0.0f, 0.0f}; }
    const int nfull = N >> 1;
    auto pair = [&](const v2f (&rec)[RS], const float (&bz)[NCB], uint32_t two) {
        const v2f xb = {(float)(two & 1u), (float)(two >> 1)};
        v2f eta = pkfma(xb, bf[1], bf[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = pkfma(bf[2 + j], rec[j], eta);
        v2f mu;
        mu.x = sigmoid_fast(eta.x); mu.y = sigmoid_fast(eta.y);
        const v2f wf = pkfma(-mu, mu, mu);
        v2f d = wf;
        if (DELTA) d = wf - rec[Q + 1];
        else {
            const v2f r = rec[Q] - mu;
            gf[0] += r; gf[1] = pkfma(xb, r, gf[1]);
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j] = pkfma(r, rec[j], gf[2 + j]);
        }
        const v2f dx = xb * d;
        h00 += d; h10 += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { hz0[j] = pkfma(d, rec[j], hz0[j]); hz1[j] = pkfma(dx, rec[j], hz1[j]); }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d.x), __float_as_uint(d.y), false, false);
        const float a0f = __uint_as_float(sw[0]), a1f = __uint_as_float(sw[1]);
#ifndef P32_ABL_NO_MFMA                                                  /* timing ablation (results meaningless): the pass without its MFMAs */
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0f, bz[cb], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1f, bz[cb], acc[cb][1], 0, 0, 0);
        }
#else
        acc[0][0][0] += a0f * bz[0]; acc[0][1][0] += a1f * bz[NCB - 1];
#endif
    };
    auto fetch_rec = [&](int pr, v2f (&rec)[RS]) {
#pragma unroll
        for (int k = 0; k < RS; ++k) rec[k] = Rp[(int64_t)pr * RS + k];
    };
    auto fetch_bz = [&](int pr, float (&bz)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bz[cb] = ZZ[(int64_t)(2 * pr + lh) * STRIDE + l31 + cb * 32];
    };
    // Records: two buffers in turn, pair p + 1 fetched while pair p is computed (scalar loads, ordered by pipe_zero).  The lane's B operands
    // come from L2 (the products table is 1.3 MB at N = 5000): four buffers, fetched two pairs ahead.  A 64-sample word holds 32 pairs.
    v2f ra[RS], rb[RS];
    float za[NCB], zb[NCB], zc[NCB], zd[NCB];
    const int nwords = nfull >> 5;                                            // whole words: pipelined
    const int plast = nfull > 0 ? nfull - 1 : 0;
    const int wd0 = (int)((int64_t)nwords * part / nparts), wd1 = (int)((int64_t)nwords * (part + 1) / nparts);
    const bool tail = part == nparts - 1;
    fetch_rec(min(wd0 * 32, plast), ra); fetch_bz(min(wd0 * 32, plast), za); fetch_bz(min(wd0 * 32 + 1, plast), zb);
    uint64_t w = T[(int64_t)min(wd0, (N - 1) >> 6) * Vpad + v];
    for (int wd = wd0; wd < wd1; ++wd) {
        const uint64_t wn = T[(int64_t)min(wd + 1, (N - 1) >> 6) * Vpad + v];
        for (int k = 0; k < 32; k += 2) {
            const int pr = wd * 32 + k;
            fetch_bz(min(pr + 2, plast), zc); fetch_bz(min(pr + 3, plast), zd);
            fetch_rec(pr + 1 + pipe_zero(ra[0].x), rb);
            pair(ra, za, (uint32_t)(w >> (2 * k)) & 3u);
            fetch_rec(min(pr + 2, plast) + pipe_zero(rb[0].x), ra);
            pair(rb, zb, (uint32_t)(w >> (2 * k + 2)) & 3u);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { za[cb] = zc[cb]; zb[cb] = zd[cb]; }
        }
        w = wn;
        if (DELTA && hdl) {
            hdl[lane] += (double)(h00.x + h00.y); hdl[64 + lane] += (double)(h10.x + h10.y);
            h00 = v2f{0.0f, 0.0f}; h10 = v2f{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                hdl[(2 + j) * 64 + lane] += (double)(hz0[j].x + hz0[j].y); hdl[(2 + Q + j) * 64 + lane] += (double)(hz1[j].x + hz1[j].y);
                hz0[j] = v2f{0.0f, 0.0f}; hz1[j] = v2f{0.0f, 0.0f};
            }
        }
    }
    for (int pr = nwords * 32; tail && pr < nfull; ++pr) {                    // the last partial word, plainly
        if (pr == nwords * 32) w = T[(int64_t)(pr >> 5) * Vpad + v];
        fetch_rec(pr, ra); fetch_bz(pr, za);
        pair(ra, za, (uint32_t)(w >> (2 * (pr & 31))) & 3u);
    }
    float h00s = h00.x + h00.y, h10s = h10.x + h10.y, gs[PC], hz0s[Q > 0 ? Q : 1], hz1s[Q > 0 ? Q : 1];
#pragma unroll
    for (int a = 0; a < PC; ++a) gs[a] = gf[a].x + gf[a].y;
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0s[j] = hz0[j].x + hz0[j].y; hz1s[j] = hz1[j].x + hz1[j].y; }
    if ((N & 1) && tail) {                                                    // the odd sample: k = 1 rows of A are zero
        const int i = N - 1;
        const float *zrow = ZZ + (int64_t)i * STRIDE + l31;
        const bool xb = (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull;
        float eta = bf[0].x + (xb ? bf[1].x : 0.0f);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j].x, Wf[(int64_t)i * Q + j], eta);
        const float mu = sigmoid_fast(eta);
        const float wf = mu * (1.0f - mu);
        float d = wf;
        if (DELTA) d = wf - P.w0f[i];
        else {
            const float r = P.yf[i] - mu;
            gs[0] += r; gs[1] += xb ? r : 0.0f;
#pragma unroll
            for (int j = 0; j < Q; ++j) gs[2 + j] = fmaf(r, Wf[(int64_t)i * Q + j], gs[2 + j]);
        }
        const float dx = xb ? d : 0.0f;
        h00s += d; h10s += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float zj = Wf[(int64_t)i * Q + j]; hz0s[j] = fmaf(d, zj, hz0s[j]); hz1s[j] = fmaf(dx, zj, hz1s[j]); }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d), 0u, false, false);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), zrow[cb * 32], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), zrow[cb * 32], acc[cb][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) g[a] = (double)gs[a];
    if (DELTA && hdl) {
        h00s = (float)(hdl[lane] + (double)h00s); h10s = (float)(hdl[64 + lane] + (double)h10s);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            hz0s[j] = (float)(hdl[(2 + j) * 64 + lane] + (double)hz0s[j]); hz1s[j] = (float)(hdl[(2 + Q + j) * 64 + lane] + (double)hz1s[j]);
        }
    }
    H[sidx(0, 0)] = h00s; H[sidx(1, 0)] = h10s; H[sidx(1, 1)] = h10s;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0s[j]; H[sidx(2 + j, 1)] = hz1s[j]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + cb * 32 + l31] = acc[cb][h][r];
        __syncthreads();
        if (lh == h) {
            const float *row = tr + l31 * (STRIDE + 1);
#pragma unroll
            for (int j = 0; j < Q; ++j)
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = row[j * (j + 1) / 2 + k];
        }
    }
}


// ---- the same pass with the covariate block on the HALF-precision matrix pipe (round 3) ---------------------------------------------------
// Measured in round 3 (timing build without the MFMAs): the four v_mfma_f32_32x32x2_f32 per pair of samples cost 1.2 of the 3.15 ms of
// k_glm_pass32 -- on gfx950 an f32 MFMA runs on the vector ALUs' rate and its time ADDS to the VALU's.  The f16 MFMA
// (v_mfma_f32_32x32x16_f16) runs on the matrix cores proper, 16x the MAC rate: 16 samples per issue instead of 2.
//   A operand = the weights of 32 variants x 16 samples as halves; B = the products z_j z_k of those 16 samples (GlmParams.zz16, per 16-sample
//   group and 32-column block one uint4 per lane, already in the B layout: lane (n = lane & 31, kg = lane >> 5) holds column n of samples
//   8 kg .. 8 kg + 7); fp32 accumulation, the C layout of the f32 form, so the transposition at the end of the pass is unchanged.
//   A lane computes its own variant's weight for all 16 samples (8 registers of half2, one per pair); four v_permlane32_swap turn them into
//   the A operands of the two 32-variant halves (lanes 32..63 supply k = 8..15 of variants 0..31 and vice versa).
//   Newton steering (DELTA = false): the weights and products rounded to half (2^-11 relative, random over 5000 terms: ~1e-5 on an entry
//   of the Hessian, what its fp32 accumulation already carries).  The final information matrix (DELTA = true) needs single precision of
//   the DIFFERENCES w - w0: both operands are split hi + lo (hi = half(x), lo = half(x - hi)) and three products are accumulated
//   (hi hi, hi lo, lo hi: 2^-21 relative, the dropped lo lo term is 2^-22); the differences are scaled by 2^10 first so that their lo parts stay
//   normal halves, and the sums are scaled back (exact).  Three f16 MFMAs cost 3/8 of one f32 MFMA per sample.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define P32_ASCALE 1024.0f
template <int Q, bool DELTA>
__device__ __forceinline__ void pass32_pk_f16(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, const GlmParams &P, const float *__restrict__ Wf,
                                          const double (&beta)[Q + 2], float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], float *tr,
                                          int part = 0, int nparts = 1, double *hdl = nullptr)
{
    // hdl, part / nparts: as pass32_pk_f32 above
    constexpr int PC = Q + 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE, NPART = DELTA ? 2 : 1;
    const int N = P.N;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    constexpr int RS = Q + 2;
    const v2f *__restrict__ Rp = (const v2f *)P.wfp;
    const v4u *__restrict__ Z16 = (const v4u *)P.zz16;                        // [group][NCB][2 (hi, lo)][64 lanes] x 16 bytes
    v16f acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][h][r] = 0.0f;
    v2f bf[PC], gf[PC], h00 = {0.0f, 0.0f}, h10 = {0.0f, 0.0f}, hz0[Q > 0 ? Q : 1], hz1[Q > 0 ? Q : 1];
#pragma unroll
    for (int a = 0; a < PC; ++a) { const float b = (float)beta[a]; bf[a] = v2f{b, b}; gf[a] = v2f{0.0f, 0.0f}; }
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = v2f{0.0f, 0.0f}; hz1[j] = v2f{0.0f, 0.0f}; }
    const int nfull = N >> 1;
    uint32_t Ah[8], Al[DELTA ? 8 : 1];                                        // this group's weights, one half2 per pair (hi, and lo for DELTA)
    auto stash = [&](int slot, v2f d) {
        if (DELTA) d = d * P32_ASCALE;
        const v2h hh = __builtin_convertvector(d, v2h);
        Ah[slot] = __builtin_bit_cast(uint32_t, hh);
        if (DELTA) {
            const v2h ll = __builtin_convertvector(d - __builtin_convertvector(hh, v2f), v2h);
            Al[slot] = __builtin_bit_cast(uint32_t, ll);
        }
    };
    auto pair = [&](const v2f (&rec)[RS], int slot, uint32_t two) {
        const v2f xb = {(float)(two & 1u), (float)(two >> 1)};
        v2f eta = pkfma(xb, bf[1], bf[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = pkfma(bf[2 + j], rec[j], eta);
        v2f mu;
        mu.x = sigmoid_fast(eta.x); mu.y = sigmoid_fast(eta.y);
        const v2f wf = pkfma(-mu, mu, mu);
        v2f d = wf;
        if (DELTA) d = wf - rec[Q + 1];
        else {
            const v2f r = rec[Q] - mu;
            gf[0] += r; gf[1] = pkfma(xb, r, gf[1]);
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j] = pkfma(r, rec[j], gf[2 + j]);
        }
        const v2f dx = xb * d;
        h00 += d; h10 += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { hz0[j] = pkfma(d, rec[j], hz0[j]); hz1[j] = pkfma(dx, rec[j], hz1[j]); }
        stash(slot, d);
    };
    auto fetch_rec = [&](int pr, v2f (&rec)[RS]) {
#pragma unroll
        for (int k = 0; k < RS; ++k) rec[k] = Rp[(int64_t)pr * RS + k];
    };
    auto fetch_bz = [&](int grp, v4u (&bz)[NCB][NPART]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q2 = 0; q2 < NPART; ++q2) bz[cb][q2] = Z16[(((int64_t)grp * NCB + cb) * 2 + q2) * 64 + lane];
    };
    // the 16 samples stashed in Ah / Al against the group's product columns
    auto flush = [&](const v4u (&bz)[NCB][NPART]) {
        v4u a0, a1, l0, l1;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            const auto sw = __builtin_amdgcn_permlane32_swap(Ah[q2], Ah[4 + q2], false, false);
            a0[q2] = sw[0]; a1[q2] = sw[1];
            if (DELTA) {
                const auto sl = __builtin_amdgcn_permlane32_swap(Al[q2], Al[4 + q2], false, false);
                l0[q2] = sl[0]; l1[q2] = sl[1];
            }
        }
        const v8h A0 = __builtin_bit_cast(v8h, a0), A1 = __builtin_bit_cast(v8h, a1);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const v8h Bh = __builtin_bit_cast(v8h, bz[cb][0]);
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bh, acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bh, acc[cb][1], 0, 0, 0);
            if (DELTA) {
                const v8h Bl = __builtin_bit_cast(v8h, bz[cb][NPART - 1]);
                const v8h L0 = __builtin_bit_cast(v8h, l0), L1 = __builtin_bit_cast(v8h, l1);
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bl, acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bl, acc[cb][1], 0, 0, 0);
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(L0, Bh, acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(L1, Bh, acc[cb][1], 0, 0, 0);
            }
        }
    };
    // Records: two buffers in turn, pair p + 1 fetched while pair p is computed (scalar loads, ordered by pipe_zero).  The lane's B operands of
    // a 16-sample group come from L2 (the table is 1.3 MB at N = 5000) one group ahead.  A 64-sample word holds 32 pairs = 4 groups.
    v2f ra[RS], rb[RS];
    v4u zc[NCB][NPART], zn[NCB][NPART];
    const int nwords = nfull >> 5;                                            // whole words: pipelined
    const int plast = nfull > 0 ? nfull - 1 : 0;
    const int glast = (N + 15) / 16 - 1;
    const int wd0 = (int)((int64_t)nwords * part / nparts), wd1 = (int)((int64_t)nwords * (part + 1) / nparts);
    const bool tail = part == nparts - 1;
    fetch_rec(min(wd0 * 32, plast), ra); fetch_bz(min(wd0 * 4, glast), zc);
    uint64_t w = T[(int64_t)min(wd0, (N - 1) >> 6) * Vpad + v];
    for (int wd = wd0; wd < wd1; ++wd) {
        const uint64_t wn = T[(int64_t)min(wd + 1, (N - 1) >> 6) * Vpad + v];
#pragma unroll 1
        for (int g4 = 0; g4 < 4; ++g4) {
            const int grp = wd * 4 + g4;
            fetch_bz(min(grp + 1, glast), zn);
            const uint32_t wbits = (uint32_t)(w >> (16 * g4)) & 0xFFFFu;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const int pr = grp * 8 + k;
                fetch_rec(pr + 1 + pipe_zero(ra[0].x), rb);
                pair(ra, k, (wbits >> (2 * k)) & 3u);
                fetch_rec(min(pr + 2, plast) + pipe_zero(rb[0].x), ra);
                pair(rb, k + 1, (wbits >> (2 * k + 2)) & 3u);
            }
            flush(zc);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q2 = 0; q2 < NPART; ++q2) zc[cb][q2] = zn[cb][q2];
        }
        w = wn;
        if (DELTA && hdl) {
            hdl[lane] += (double)(h00.x + h00.y); hdl[64 + lane] += (double)(h10.x + h10.y);
            h00 = v2f{0.0f, 0.0f}; h10 = v2f{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                hdl[(2 + j) * 64 + lane] += (double)(hz0[j].x + hz0[j].y); hdl[(2 + Q + j) * 64 + lane] += (double)(hz1[j].x + hz1[j].y);
                hz0[j] = v2f{0.0f, 0.0f}; hz1[j] = v2f{0.0f, 0.0f};
            }
        }
    }
    float h00s, h10s, gs[PC], hz0s[Q > 0 ? Q : 1], hz1s[Q > 0 ? Q : 1];
    // The last partial word and the odd last sample (the last part only), plainly: groups of 8 pairs, absent pairs stashed as zeros (the table is
    // zero behind sample N as well).  The odd sample is the first half of pair `nfull`; its record is not in the pair table (Wf / yf / w0f).
    float odd_d = 0.0f;
    bool odd_x = false;
    if ((N & 1) && tail) {
        const int i = N - 1;
        odd_x = (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull;
        float eta = bf[0].x + (odd_x ? bf[1].x : 0.0f);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j].x, Wf[(int64_t)i * Q + j], eta);
        const float mu = sigmoid_fast(eta);
        const float wf = mu * (1.0f - mu);
        odd_d = DELTA ? wf - P.w0f[i] : wf;
        if (!DELTA) {
            const float r = P.yf[i] - mu;
            gf[0].x += r; gf[1].x += odd_x ? r : 0.0f;
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j].x = fmaf(r, Wf[(int64_t)i * Q + j], gf[2 + j].x);
        }
        const float dx = odd_x ? odd_d : 0.0f;
        h00.x += odd_d; h10.x += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float zj = Wf[(int64_t)i * Q + j]; hz0[j].x = fmaf(odd_d, zj, hz0[j].x); hz1[j].x = fmaf(dx, zj, hz1[j].x); }
    }
    if (tail) {
        const int pend = nfull + (N & 1);                                     // pair slots in use, the odd sample's included
        for (int p0 = nwords * 32; p0 < pend; p0 += 8) {
            if (p0 == nwords * 32) w = T[(int64_t)min(p0 >> 5, (N - 1) >> 6) * Vpad + v];
            fetch_bz(min(p0 >> 3, glast), zc);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pr = p0 + k;
                if (pr < nfull) { fetch_rec(pr, ra); pair(ra, k, (uint32_t)(w >> (2 * (pr & 31))) & 3u); }
                else if (pr == nfull && (N & 1)) stash(k, v2f{odd_d, 0.0f});
                else stash(k, v2f{0.0f, 0.0f});
            }
            flush(zc);
        }
    }
    h00s = h00.x + h00.y; h10s = h10.x + h10.y;
#pragma unroll
    for (int a = 0; a < PC; ++a) gs[a] = gf[a].x + gf[a].y;
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0s[j] = hz0[j].x + hz0[j].y; hz1s[j] = hz1[j].x + hz1[j].y; }
#pragma unroll
    for (int a = 0; a < PC; ++a) g[a] = (double)gs[a];
    if (DELTA && hdl) {
        h00s = (float)(hdl[lane] + (double)h00s); h10s = (float)(hdl[64 + lane] + (double)h10s);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            hz0s[j] = (float)(hdl[(2 + j) * 64 + lane] + (double)hz0s[j]); hz1s[j] = (float)(hdl[(2 + Q + j) * 64 + lane] + (double)hz1s[j]);
        }
    }
    H[sidx(0, 0)] = h00s; H[sidx(1, 0)] = h10s; H[sidx(1, 1)] = h10s;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0s[j]; H[sidx(2 + j, 1)] = hz1s[j]; }
    const float unscale = DELTA ? 1.0f / P32_ASCALE : 1.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + cb * 32 + l31] = acc[cb][h][r] * unscale;
        __syncthreads();
        if (lh == h) {
            const float *row = tr + l31 * (STRIDE + 1);
#pragma unroll
            for (int j = 0; j < Q; ++j)
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = row[j * (j + 1) / 2 + k];
        }
    }
}

#ifndef P32_F16
#define P32_F16 1                    /* 0: the covariate block of the packed passes on the f32 MFMA (round 2), for A/B */
#endif
template <int Q, bool DELTA>
__device__ __forceinline__ void pass32_pk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, const GlmParams &P, const float *__restrict__ Wf,
                                          const double (&beta)[Q + 2], float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], float *tr,
                                          int part = 0, int nparts = 1, double *hdl = nullptr)
{
#if P32_F16
    pass32_pk_f16<Q, DELTA>(T, Vpad, v, P, Wf, beta, H, g, tr, part, nparts, hdl);
#else
    pass32_pk_f32<Q, DELTA>(T, Vpad, v, P, Wf, beta, H, g, tr, part, nparts, hdl);
#endif
}

// ---- the fast phase as ROUNDS of lean kernels over lists of variants (GlmParams.chord_on) -------------------------------------
// k_glm_fast<Q, false> above keeps a variant on its lane through every pass and the solves between them: 256 VGPRs + 570-890 bytes of
// scratch at two wavefronts per SIMD, every wavefront iterating as long as its slowest lane, and the fp64-score passes at 3.4x the cost of
// a single-precision one (8.6 vs 2.5 ms per 262 144 variants at N = 5000, q = 10).  Here each pass is its own kernel with only the pass'
// registers, the 12x12 LDL^T solves are small kernels of their own, and between rounds the variants still iterating are re-listed, so
// every wavefront of every round is full:
//   k_glm_fast<Q, true>  prefilter + routing; lists the variants to fit, beta = the warm start                    (ch_list[0])
//   k_glm_pass32 / k_glm_solve32   one single-precision Newton step per round (Hessian on the matrix pipe) until the step is <= chord_enter
//   k_glm_score / k_glm_chord      fp64 refinement WITHOUT refreshing the Hessian: with the factor of the last single-precision pass (taken
//       at distance s from the optimum) the iteration  beta += H1^-1 g(beta)  contracts by ~|H(beta) - H1| / |H| ~ 8 s per round, and the fp64
//       score (eta in fp64, the logistic function in fp32 as in the fast passes) is all a round evaluates -- 63 VGPRs, 8 wavefronts per SIMD
//   then k_glm_slow / k_glm_final as before: the final pass takes its exact fp64 Newton step and repeats itself should a lane have been
//       handed over too early.
// Anything that does not behave (pivot test, non-finite, no contraction, round limits) goes to k_glm_slow's list: the reference's iteration.
__device__ __forceinline__ bool round_lane(const int *__restrict__ list, const int *__restrict__ cnt, int64_t idx, int64_t &v)
{
    const int n = *cnt;
    const bool on = idx < n;
    v = list[on ? idx : 0];
    return on;
}

// From the second pass round on the lists are short (a per cent of the batch), most of the chip idles, and a pass is as long as its 5000
// dependent samples: those rounds run k_glm_pass32_split, where SplitCfg<Q>::S wavefronts share a variant's samples and add their partial
// sums in a fixed order.  Which kernel a round uses depends on the round only, never on the list length: a variant's result must not
// depend on what else is in its batch.
template <int Q> struct SplitCfg {                                  // wavefronts per variant: as many as their LDS (transposition tile + partial sums) allows
    static constexpr int PC = Q + 2, NV = PC * (PC + 1) / 2 + PC;
    static constexpr int PER_WAVE = 4 * (FastCols<Q>::LDS_FLOATS + NV * 64);
    static constexpr int S = 4 * PER_WAVE <= 150 * 1024 ? 4 : (2 * PER_WAVE <= 150 * 1024 ? 2 : 1);
};
template <int Q>
__global__ __launch_bounds__(64 * SplitCfg<Q>::S) void k_glm_pass32_split(const uint64_t *__restrict__ T, int64_t Vpad, const float *__restrict__ Wf,
                                                                           GlmParams P, const int *__restrict__ list, const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2, S = SplitCfg<Q>::S, NV = NH + PC;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    extern __shared__ float sm[];                                    // [S] transposition tiles, then partial sums [S][NV][64]
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // wave-uniform, and known to be
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + lane, v);
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = P.ch_bs[(int64_t)a * Vpad + v];
    float Hf[NH];
    // pass32_pk's __syncthreads() pairs are hit by all S wavefronts the same number of times
    pass32_pk<Q, false>(T, Vpad, v, P, Wf, beta, Hf, g, sm + wv * FastCols<Q>::LDS_FLOATS, wv, S);
    float *part = sm + S * FastCols<Q>::LDS_FLOATS;
#pragma unroll
    for (int a = 0; a < NH; ++a) part[(wv * NV + a) * 64 + lane] = Hf[a];
#pragma unroll
    for (int a = 0; a < PC; ++a) part[(wv * NV + NH + a) * 64 + lane] = (float)g[a];
    __syncthreads();
    if (wv != 0 || !on) return;
#pragma unroll
    for (int a = 0; a < NH; ++a) { float t = part[a * 64 + lane]; for (int s2 = 1; s2 < S; ++s2) t += part[(s2 * NV + a) * 64 + lane]; P.ch_hf[(int64_t)a * Vpad + v] = t; }
#pragma unroll
    for (int a = 0; a < PC; ++a) { float t = part[(NH + a) * 64 + lane]; for (int s2 = 1; s2 < S; ++s2) t += part[(s2 * NV + NH + a) * 64 + lane]; P.ch_g[(int64_t)a * Vpad + v] = (double)t; }
}

template <int Q, bool PK>
__global__ __launch_bounds__(64, PK ? 2 : 3) void k_glm_pass32(const uint64_t *__restrict__ T, int64_t Vpad, const double *__restrict__ y,
                                                      const float *__restrict__ Wf, GlmParams P, const int *__restrict__ list,
                                                      const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + threadIdx.x, v);
    double beta[PC], g[PC], maxdev;
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = P.ch_bs[(int64_t)a * Vpad + v];
    float Hf[NH];
    __shared__ float tr[FastCols<Q>::LDS_FLOATS];
    if (PK) pass32_pk<Q, false>(T, Vpad, v, P, Wf, beta, Hf, g, tr);
    else fast_pass_mfma<Q, true>(T, Vpad, v, P.N, P.NB64, y, P.ws, Wf, P.zz, beta, Hf, g, maxdev, tr);
    if (!on) return;
#pragma unroll
    for (int a = 0; a < NH; ++a) P.ch_hf[(int64_t)a * Vpad + v] = Hf[a];
#pragma unroll
    for (int a = 0; a < PC; ++a) P.ch_g[(int64_t)a * Vpad + v] = g[a];
}

template <int Q, bool FIRST>
__global__ __launch_bounds__(256) void k_glm_solve32(int64_t Vpad, GlmParams P, GlmWork wk, const int *__restrict__ list, const int *__restrict__ cnt,
                                                     int *__restrict__ next, int *__restrict__ next_cnt, int *__restrict__ chord, int *__restrict__ chord_cnt,
                                                     int last_round)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    __shared__ int push_lds[17];
    if ((int64_t)blockIdx.x * 256 >= *cnt) return;                    // (workgroups of four wavefronts: one list atomic per 256 variants, list_push_block)
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 256 + threadIdx.x, v);
    const double nobs = (double)P.N;
    bool go_next = false, go_chord = false, go_slow = false;
    if (on) {
        double A[NH], g[PC];
        if (FIRST) {
            // the step from the warm start needs no pass over the samples: there eta_i is the NULL model's, so w_i = w0_i and r_i = r0_i do
            // not depend on the variant, and the variant's row of X^T W X and its score are sums of per-run vectors over the carriers
            // (k_glm_bitdot); the [1, z] block is the null model's (null_h, a0).  All fp64.
            A[sidx(0, 0)] = P.null_h[0]; A[sidx(1, 0)] = A[sidx(1, 1)] = P.ch_bd[v];
            g[0] = P.null_g[0]; g[1] = P.ch_bd[(int64_t)(Q + 1) * Vpad + v];
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                A[sidx(2 + j, 0)] = P.null_h[1 + j]; A[sidx(2 + j, 1)] = P.ch_bd[(int64_t)(1 + j) * Vpad + v]; g[2 + j] = P.null_g[1 + j];
#pragma unroll
                for (int k = 0; k <= j; ++k) A[sidx(2 + j, 2 + k)] = P.a0[j * (j + 1) / 2 + k];
            }
#pragma unroll
            for (int a = 0; a < NH; ++a) A[a] = A[a] / nobs;
#pragma unroll
            for (int a = 0; a < PC; ++a) { A[sidx(a, a)] -= 1e-10; g[a] = g[a] / nobs; }
        } else {
#pragma unroll
            for (int a = 0; a < NH; ++a) A[a] = (double)P.ch_hf[(int64_t)a * Vpad + v] / nobs;
#pragma unroll
            for (int a = 0; a < PC; ++a) { A[sidx(a, a)] -= 1e-10; g[a] = P.ch_g[(int64_t)a * Vpad + v] / nobs; }
        }
        double det;
        if (!ldl_factor<PC>(A, 1e-4, &det)) go_slow = true;                                          // fp32 cannot resolve this design
        else {
            ldl_solve<PC>(A, g);
            double stp = 0.0; bool finite = true;
#pragma unroll
            for (int a = 0; a < PC; ++a) {
                const double b = P.ch_bs[(int64_t)a * Vpad + v] + g[a];
                stp = fmax(stp, fabs(g[a])); finite = finite && isfinite(b);
                P.ch_bs[(int64_t)a * Vpad + v] = b;
            }
            if (!finite) go_slow = true;
            else if (stp <= P.chord_enter) {
                go_chord = true;
#pragma unroll
                for (int a = 0; a < NH; ++a) P.ch_fac[(int64_t)a * Vpad + v] = A[a];
                P.ch_rho[v] = (float)fmin(0.5, fmax(8.0 * stp, 1e-4));
            }
            else if (last_round) go_slow = true;
            else go_next = true;
        }
    }
    list_push_block(go_next, next, next_cnt, (int)v, push_lds);
    list_push_block(go_chord, chord, chord_cnt, (int)v, push_lds);
    list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
}

// The first Newton step (from the null model, carrier sums instead of a pass: see k_glm_solve32<Q, true>) as a bordered solve.  With the
// variant's column ordered LAST the leading (Q+1) x (Q+1) block of the matrix is the null model's and the same for every variant: its factor
// and the constant halves of the solve come from the host (GlmParams.b1), and a variant costs two triangular solves with that factor and
// one division -- ~160 fp64 FMAs instead of a 12 x 12 factorisation (~450 and 12 divisions, 156 registers of matrix per lane).  Same
// system, same scaling and ridge; the pivot test applies to the variant's own pivot (its residual against intercept AND covariates).  A
// first step that is already <= chord_enter (a few % of the variants) takes one more Newton round instead of entering the chord rounds
// here: those need the factor in the general kernel's order.
template <int Q>
__global__ __launch_bounds__(256) void k_glm_first_step(int64_t Vpad, GlmParams P, GlmWork wk, const int *__restrict__ list, const int *__restrict__ cnt,
                                                        int *__restrict__ next, int *__restrict__ next_cnt)
{
    constexpr int M = Q + 1, NL = M * (M + 1) / 2;
    __shared__ int push_lds[17];
    if ((int64_t)blockIdx.x * 256 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 256 + threadIdx.x, v);
    const double nobs = (double)P.N;
    const double *__restrict__ L = P.b1, *__restrict__ iD = P.b1 + NL, *__restrict__ yc = iD + M, *__restrict__ xc0 = yc + M;
    bool go_next = false, go_slow = false;
    if (on) {
        double w[M], u[M];
        const double bd0 = P.ch_bd[v];
        w[0] = bd0 / nobs;                                             // column of the variant: with the intercept, with covariate j
#pragma unroll
        for (int j = 0; j < Q; ++j) w[1 + j] = P.ch_bd[(int64_t)(1 + j) * Vpad + v] / nobs;
        const double alpha = bd0 / nobs - 1e-10, gk = P.ch_bd[(int64_t)(Q + 1) * Vpad + v] / nobs;
#pragma unroll
        for (int i = 0; i < M; ++i) {                                  // w = L^-1 a
#pragma unroll
            for (int k = 0; k < i; ++k) w[i] = fma(-L[sidx(i, k)], w[k], w[i]);
        }
        double d = alpha, yk = gk;
#pragma unroll
        for (int i = 0; i < M; ++i) { u[i] = w[i] * iD[i]; d = fma(-w[i], u[i], d); yk = fma(-u[i], yc[i], yk); }   // l = w / D;  d = alpha - w.l;  y_k = g_k - l.y_c
        if (d == 0.0 || fabs(d) <= 1e-4 * fabs(alpha) || !isfinite(d)) go_slow = true;
        else {
            const double xk = yk / d;
#pragma unroll
            for (int i = M - 1; i >= 0; --i) {                         // u = L^-T l
#pragma unroll
                for (int k = i + 1; k < M; ++k) u[i] = fma(-L[sidx(k, i)], u[k], u[i]);
            }
            double stp = fabs(xk); bool finite = true;
            {
                const double b = P.ch_bs[(int64_t)1 * Vpad + v] + xk;
                finite = finite && isfinite(b);
                P.ch_bs[(int64_t)1 * Vpad + v] = b;
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {                              // x_c = x_c0 - x_k u;  row 0 = intercept, rows 2.. = covariates
                const double x = fma(-xk, u[i], xc0[i]);
                const int a = i == 0 ? 0 : i + 1;
                const double b = P.ch_bs[(int64_t)a * Vpad + v] + x;
                stp = fmax(stp, fabs(x)); finite = finite && isfinite(b);
                P.ch_bs[(int64_t)a * Vpad + v] = b;
            }
            if (!finite) go_slow = true; else go_next = true;
            (void)stp;
        }
    }
    list_push_block(go_next, next, next_cnt, (int)v, push_lds);        // (nearly every lane: one atomic per 256 variants)
    list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);           // (rare: a wavefront without one issues nothing)
}

#ifndef GLM_SCORE_BLOCKS
#define GLM_SCORE_BLOCKS 1
#endif
template <int Q>
__global__ __launch_bounds__(256, GLM_SCORE_BLOCKS) void k_glm_score(const uint64_t *__restrict__ T, int64_t Vpad, const double *__restrict__ y,
                                                   const double *__restrict__ W, GlmParams P, const int *__restrict__ list,
                                                   const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, RS = Q + 1;                            // record of a sample: its standardised covariates, then y (GlmParams.rec)
    if ((int64_t)blockIdx.x * 256 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 256 + threadIdx.x, v);
    if (!__any(on)) return;
    const int N = P.N, NB64 = P.NB64;
    const double *__restrict__ R = P.rec;
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) { beta[a] = P.ch_bs[(int64_t)a * Vpad + v]; g[a] = 0.0; }
    double maxdev = 0.0;
    // Two record buffers in turn, each fetched while the other sample is computed: the scalar loads are otherwise exposed in front of every
    // sample (measured: 2.3x the VALU time).  Whole 64-sample words run in this pipelined form, the last partial word plainly.
    auto one = [&](const double (&rc)[RS], bool xb) {
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double mu = (double)sigmoid_fast((float)eta);
        const double r = rc[Q] - mu;
        maxdev = fmax(maxdev, fabs(r));
        g[0] += r; g[1] += xb ? r : 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, rc[j], g[2 + j]);
    };
    double ra[RS], rb[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) ra[k] = R[k];
    for (int wd = 0; wd < NB64; ++wd) {
        const uint64_t w = T[(int64_t)wd * Vpad + v];
        const int lim = min(64, N - wd * 64);
        if (lim == 64) {
            for (int b = 0; b < 64; b += 2) {
                const int i = wd * 64 + b, i2 = min(i + 2, N - 1);
                const int za = pipe_zero(ra[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                one(ra, (w >> b) & 1ull);
                const int zb = pipe_zero(rb[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                one(rb, (w >> (b + 1)) & 1ull);
            }
        } else {
            for (int b = 0; b < lim; ++b) {
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(wd * 64 + b) * RS + k];
                one(ra, (w >> b) & 1ull);
            }
        }
    }
    if (!on) return;
#pragma unroll
    for (int a = 0; a < PC; ++a) P.ch_g[(int64_t)a * Vpad + v] = g[a];
    P.ch_md[v] = maxdev;
}

// k_glm_score for the chord rounds after the first: their lists are short (nothing on the benchmark rows), and a pass by one wavefront is as
// long as its 5000 dependent samples (1.8 ms).  SCORE_SPLIT wavefronts share a variant's samples -- contiguous runs of 64-sample words, the
// last one also the partial word -- and wavefront 0 adds the partial scores in a fixed order.  As with k_glm_pass32_split the kernel is
// chosen by the round, never by the list length.
#define SCORE_SPLIT 8
template <int Q>
__global__ __launch_bounds__(64 * SCORE_SPLIT) void k_glm_score_split(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P,
                                                                      const int *__restrict__ list, const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, RS = Q + 1, S = SCORE_SPLIT;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    __shared__ double part[S - 1][PC + 1][64];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + lane, v);
    const int N = P.N, NB64 = P.NB64;
    const double *__restrict__ R = P.rec;
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) { beta[a] = P.ch_bs[(int64_t)a * Vpad + v]; g[a] = 0.0; }
    double maxdev = 0.0;
    auto one = [&](const double (&rc)[RS], bool xb) {
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double mu = (double)sigmoid_fast((float)eta);
        const double r = rc[Q] - mu;
        maxdev = fmax(maxdev, fabs(r));
        g[0] += r; g[1] += xb ? r : 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, rc[j], g[2 + j]);
    };
    const int wd0 = (int)((int64_t)NB64 * wv / S), wd1 = (int)((int64_t)NB64 * (wv + 1) / S);
    double ra[RS], rb[RS];
    {
        const int i0 = min(wd0 * 64, N - 1);
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i0 * RS + k];
    }
    for (int wd = wd0; wd < wd1; ++wd) {
        const uint64_t w = T[(int64_t)wd * Vpad + v];
        const int lim = min(64, N - wd * 64);
        if (lim == 64) {
            for (int b = 0; b < 64; b += 2) {
                const int i = wd * 64 + b, i2 = min(i + 2, N - 1);
                const int za = pipe_zero(ra[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                one(ra, (w >> b) & 1ull);
                const int zb = pipe_zero(rb[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                one(rb, (w >> (b + 1)) & 1ull);
            }
        } else {
            for (int b = 0; b < lim; ++b) {
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(wd * 64 + b) * RS + k];
                one(ra, (w >> b) & 1ull);
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int a = 0; a < PC; ++a) part[wv - 1][a][lane] = g[a];
        part[wv - 1][PC][lane] = maxdev;
    }
    __syncthreads();
    if (wv != 0 || !on) return;
    for (int s2 = 0; s2 < S - 1; ++s2) {
#pragma unroll
        for (int a = 0; a < PC; ++a) g[a] += part[s2][a][lane];
        maxdev = fmax(maxdev, part[s2][PC][lane]);
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) P.ch_g[(int64_t)a * Vpad + v] = g[a];
    P.ch_md[v] = maxdev;
}

template <int Q>
__global__ __launch_bounds__(256) void k_glm_chord(int64_t Vpad, GlmParams P, GlmWork wk, const int *__restrict__ list, const int *__restrict__ cnt,
                                                   int *__restrict__ next, int *__restrict__ next_cnt, int last_round, int first_ll)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    __shared__ int push_lds[17];
    if ((int64_t)blockIdx.x * 256 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 256 + threadIdx.x, v);
    const double nobs = (double)P.N;
    bool go_next = false, go_slow = false, go_fin = false, go_direct = false;
    if (on) {
        double A[NH], g[PC], beta[PC], bin[PC];
#pragma unroll
        for (int a = 0; a < NH; ++a) A[a] = P.ch_fac[(int64_t)a * Vpad + v];
#pragma unroll
        for (int a = 0; a < PC; ++a) { g[a] = P.ch_g[(int64_t)a * Vpad + v] / nobs; beta[a] = bin[a] = P.ch_bs[(int64_t)a * Vpad + v]; }
        ldl_solve<PC>(A, g);
        double stp = 0.0; bool finite = true;
#pragma unroll
        for (int a = 0; a < PC; ++a) { beta[a] += g[a]; stp = fmax(stp, fabs(g[a])); finite = finite && isfinite(beta[a]); }
        const bool sep = P.ch_md[v] <= 1e-8;
        // (a variant finished from the likelihood pass takes its beta from the exact Newton step there, not from this chord step: the chord result
        // only says where the information matrix is evaluated, and bse tolerates 6e-7 of distance from the optimum as well as 2.5e-7)
        const double rs = (double)P.ch_rho[v] * stp;
        if (finite && !sep && (rs <= P.chord_tol || (first_ll && stp <= 1e-4 && rs <= 2.4 * P.chord_tol)) && P.fin_rounds) {   // done: on to the finishing kernels
            // first_ll: this round's pass was k_glm_ll, so ch_g / ch_ll / ch_md are the exact score, log-likelihood and callback value at the beta
            // that came in.  With a small step that is all the finishing needs (k_glm_finish<Q, true>): the variant skips the final likelihood pass.
            go_direct = first_ll && stp <= 1e-4;
            go_fin = !go_direct;
#pragma unroll
            for (int a = 0; a < PC; ++a) {
                if (go_direct) P.ch_b0[(int64_t)a * Vpad + v] = bin[a];              // the beta the pass was taken at
                P.ch_bs[(int64_t)a * Vpad + v] = beta[a];
            }
        }
        else if (finite && !sep && (double)P.ch_rho[v] * stp <= P.chord_tol) {                     // done: beta to the final pass
            wk.state[v] = 1;
            if (P.ws) {
#pragma unroll
                for (int j = 0; j < Q; ++j) { beta[2 + j] = beta[2 + j] / P.wstd[Q + j]; beta[0] = fma(-beta[2 + j], P.wstd[j], beta[0]); }
            }
#pragma unroll
            for (int a = 0; a < PC; ++a) wk.bw[(int64_t)a * Vpad + v] = beta[a];
        }
        else if (!finite || sep || last_round || stp > 0.05) go_slow = true;                       // not contracting: the fp64 restart decides
        else {
            go_next = true;
#pragma unroll
            for (int a = 0; a < PC; ++a) P.ch_bs[(int64_t)a * Vpad + v] = beta[a];
        }
    }
    list_push_block(go_next, next, next_cnt, (int)v, push_lds);
    list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
    list_push_block(go_fin, P.ch_list[4], P.ch_cnt + 30, (int)v, push_lds);
    list_push_block(go_direct, P.ch_list[5], P.ch_cnt + 29, (int)v, push_lds);
}

// ---- sums of per-run vectors over a variant's carriers, by nibble table (the device of k_glm_ols_tab) --------------------------
// vals[i][c], c < NE = Q + 2: (w0_i, w0_i z_i0 .. w0_i z_i,Q-1, r0_i) at the null model.  tab[(sb * 16 + nib) * 16 + value][c] = the sum
// over the set bits of `value` of the four samples of nibble `nib` of word `sb`.
__global__ __launch_bounds__(256) void k_bitdot_build_tab(const double *__restrict__ vals, int NE, int N, int NB64, double *__restrict__ tab)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= NB64 * 256) return;
    const int value = e & 15, base = (e >> 4) * 4;
    for (int c = 0; c < NE; ++c) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) {
            const int i = base + k;
            if (((value >> k) & 1) && i < N) a += vals[(int64_t)i * NE + c];
        }
        tab[(int64_t)e * NE + c] = a;
    }
}

extern "C" hipError_t shk_bitdot_i8(hipStream_t st, const uint64_t *T, int64_t Vpad, int NB64, int NE, const void *atab, const double *scale, double *out);
template <int Q>
__global__ __launch_bounds__(256) void k_glm_bitdot(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P)
{
    constexpr int NE = Q + 2;
    __shared__ __attribute__((aligned(16))) double lt[2][256 * NE];
    const int tid = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;               // Vpad is a multiple of 256: every thread walks (rows >= V are zero)
    const int NB64 = P.NB64;
    const double *__restrict__ tab = P.bd_tab;
    double acc[NE];
#pragma unroll
    for (int c = 0; c < NE; ++c) { acc[c] = 0.0; lt[0][tid * NE + c] = tab[(int64_t)tid * NE + c]; }
    for (int sb = 0; sb < NB64; sb++) {
        double nxt[NE];
        const bool more = sb + 1 < NB64;
#pragma unroll
        for (int c = 0; c < NE; ++c) nxt[c] = more ? tab[((int64_t)(sb + 1) * 256 + tid) * NE + c] : 0.0;
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        __syncthreads();
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
#pragma unroll
    for (int c = 0; c < NE; ++c) P.ch_bd[(int64_t)c * Vpad + v] = acc[c];
}

// the decisions of model.py:332-344, 384 on a finished fit, and its output row (shared by k_glm_final and k_glm_finish)
template <int Q>
__device__ __forceinline__ void glm_emit(int status, double bse1, double llf, double (&beta)[Q + 2], bool standardised, int64_t v, int64_t V,
                                         const GlmParams &P, double *__restrict__ out, uint32_t *__restrict__ flags,
                                         int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    uint32_t fl = flags[v];
    bool to_firth = false;
    if (status == 1) { fl |= SH_NOTE_PERFECT_SEP; to_firth = true; }
    else if (status == 2) { fl |= SH_NOTE_MATRIX_INV; to_firth = true; }
    else if (bse1 > 3.0) { fl |= SH_NOTE_HIGH_BSE; to_firth = true; }                         // model.py:332-334
    else {
        if (standardised) {                                          // back to the covariates as given
#pragma unroll
            for (int j = 0; j < Q; ++j) { beta[2 + j] = beta[2 + j] / P.wstd[Q + j]; beta[0] = fma(-beta[2 + j], P.wstd[j], beta[0]); }
        }
        const double lrstat = -2.0 * (P.null_llf - llf);
        double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);                        // model.py:336-339
        out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = bse1; out[4 * V + v] = beta[0];
#pragma unroll
        for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = beta[2 + j];
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;   // model.py:384
    }
    flags[v] = fl;
    if (to_firth) { const int slot = atomicAdd(firth_count, 1); firth_list[slot] = (int)v; }
}

// ---- the fp64 restart, one WORKGROUP per listed variant (round 2's lane-per-variant form is gone) ----------------
// The restart list is short in every ordinary batch (nothing on the benchmark rows, a handful of separated or ill-conditioned k-mers in real
// data), and with a lane per variant each of its ~8 iterations is a walk over N / S samples by one wavefront, then k_glm_final's walk over all
// N: 1.5-4 ms for a single listed variant, whatever the batch.  Here 256 threads share a variant's samples (thread t: t, t + 256, ...),
// sums by xor-shuffle then waves 0..3 (blk_sum), the p x p algebra on thread 0: the same iteration and decisions as k_glm_slow followed by
// k_glm_final<Q, false> -- statsmodels' Newton from the reference's start vector (ridge 1e-10, |step|_inf <= 1e-8, 35 iterations, the
// perfect-prediction callback), then llf, bse[1] and the notes at the final beta.  ~25 us per iteration.  Every listed variant goes through
// this kernel whatever the list's length, so its result does not depend on what else is in the batch.
template <int NA> __device__ __forceinline__ void blk_sum(double (&a)[NA], double *red, int tid);      // defined with the Firth workgroup kernels below
template <int Q>
__global__ __launch_bounds__(256) void k_glm_slow_blk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                     const double *__restrict__ y, const double *__restrict__ W, GlmParams P, GlmWork wk,
                                                     double *__restrict__ out, uint32_t *__restrict__ flags,
                                                     int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2, NA = NH + PC + 1;      // packed information, score, log-likelihood
    __shared__ double s_beta[PC], s_red[4 * NA], s_mx[4];
    __shared__ int s_ctl;                                            // 0 = another Newton pass, 1 = evaluation pass at the final beta, 2 = variant done
    const int cnt = *wk.slow_count, tid = threadIdx.x, N = P.N;
    const double nobs = (double)N;
    for (int idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const int64_t v = wk.slow_list[idx];
        int it = 0, status = 0, reps = 0;                            // thread-0 state
        bool capped = false;                                         //   the Newton loop ended at its 35th step, still moving
        double llf = NAN, bse1 = NAN;
        __syncthreads();
        if (tid == 0) { for (int a = 0; a < PC; ++a) s_beta[a] = 0.0; s_beta[0] = P.ymean_logit; s_ctl = 0; }
        __syncthreads();
        while (s_ctl != 2) {
            const bool want_ll = s_ctl == 1;
            double acc[NA], beta[PC], mx = 0.0;
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = 0.0;
#pragma unroll
            for (int a = 0; a < PC; ++a) beta[a] = s_beta[a];
            for (int i = tid; i < N; i += 256) {
                const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
                double x[PC];
                x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
                for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
                double eta = 0.0;
#pragma unroll
                for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
                const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
                const double yi = y[i], r = yi - mu;
                mx = fmax(mx, fabs(r));
                if (want_ll) {
                    const double lm = log(mu);                                                   // as info_pass
                    acc[NH + PC] += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
                }
#pragma unroll
                for (int a = 0; a < PC; ++a) {
                    const double wa = wgt * x[a];
                    acc[NH + a] = fma(r, x[a], acc[NH + a]);
#pragma unroll
                    for (int c = 0; c <= a; ++c) acc[sidx(a, c)] = fma(wa, x[c], acc[sidx(a, c)]);
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m));
            blk_sum<NA>(acc, s_red, tid);                            // (its barriers also order the s_mx accesses below)
            if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
            __syncthreads();
            if (tid == 0) {
                const double maxdev = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
                double H[NH], g[PC], det;
#pragma unroll
                for (int a = 0; a < NH; ++a) H[a] = acc[a] / nobs;
#pragma unroll
                for (int a = 0; a < PC; ++a) g[a] = acc[NH + a] / nobs;
                if (!want_ll) {                                      // a Newton step (k_glm_slow)
                    if (it > 0 && maxdev <= 1e-8) { status = 1; s_ctl = 2; }                    // _check_perfect_pred
                    else {
#pragma unroll
                        for (int a = 0; a < PC; ++a) H[sidx(a, a)] -= 1e-10;                    // optimizer.py:415-423
                        if (!ldl_factor<PC>(H, 0.0, &det)) { status = 2; s_ctl = 2; }
                        else {
                            ldl_solve<PC>(H, g);
                            bool moving = false;
#pragma unroll
                            for (int a = 0; a < PC; ++a) { s_beta[a] = beta[a] + g[a]; moving = moving || (fabs(g[a]) > 1e-8); }
                            ++it;
                            if (!moving || it >= 35) { s_ctl = 1; capped = moving; }
                        }
                    }
                } else {                                             // evaluation at the final beta (k_glm_final<Q, false>)
                    if (maxdev <= 1e-8) { status = 1; s_ctl = 2; }                              // callback after the last update
                    else {
                        llf = acc[NH + PC];
                        if (!ldl_factor<PC>(H, 4.0e-16, &det)) { status = 2; s_ctl = 2; }
                        else {
                            double e[PC];
#pragma unroll
                            for (int a = 0; a < PC; ++a) e[a] = (a == 1) ? 1.0 : 0.0;
                            ldl_solve<PC>(H, e);
                            bse1 = sqrt(e[1] / nobs);                // Hinv = inv(-Hessian/nobs)/nobs, no ridge (SM:base/model.py:533-534)
                            ldl_solve<PC>(H, g);
                            double smax = 0.0; bool finite = true;
#pragma unroll
                            for (int a = 0; a < PC; ++a) { smax = fmax(smax, fabs(g[a])); finite = finite && isfinite(g[a]); }
                            s_ctl = 2;
                            // (a fit that stopped at statsmodels' 35-iteration cap without converging -- SM:base/optimizer.py:407-427: a covariate
                            // that quasi-separates the phenotype -- is reported where it stopped, as the reference does: no further step)
                            if (finite && !capped) {
#pragma unroll
                                for (int a = 0; a < PC; ++a) s_beta[a] = beta[a] + g[a];
                                if (smax > 5e-7 && ++reps < 6) s_ctl = 1;                      // llf and bse belong to a beta this far from the fixed point: again
                            }
                        }
                    }
                }
                if (s_ctl == 2) {
                    double b[PC];
#pragma unroll
                    for (int a = 0; a < PC; ++a) b[a] = s_beta[a];
                    glm_emit<Q>(status, bse1, llf, b, false, v, V, P, out, flags, firth_list, firth_count);
                }
            }
            __syncthreads();
        }
    }
}

// ---- kernel 3: phase C, fp64 evaluation at the final beta + the decisions of model.py:332-344, 384 --------------------------
template <int Q, bool DELTA>
__global__ __launch_bounds__(64, 2) void k_glm_final(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                  const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                  GlmWork wk, double *__restrict__ out, uint32_t *__restrict__ flags,
                                                  int *__restrict__ firth_list, int *__restrict__ firth_count, int want_state)
{
    constexpr int PC = Q + 2;
    const int64_t v = (int64_t)blockIdx.x * 64 + threadIdx.x;
    // DELTA: lanes in state 1; a lane whose information matrix looks (nearly) singular there is left in state 2 for the plain fp64 pass
    // (launched right behind with want_state = 2): an exactly singular design must be SEEN as singular (note matrix-inversion-error), and
    // the fp32 part of the DELTA sums blurs a zero pivot to ~1e-7.
    const bool fin = (v < V) && (wk.state[v < V ? v : 0] == want_state);
    if (!__any(fin)) return;
    const int64_t vr = fin ? v : 0;
    const int N = P.N, NB64 = P.NB64;
    const double nobs = (double)N;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = fin ? wk.bw[(int64_t)a * Vpad + vr] : 0.0;
    __shared__ float tr[DELTA ? FastCols<Q>::LDS_FLOATS : 1];
    if (DELTA) {                                                     // to the standardised coordinates: b' = b * scale, b0' = b0 + sum b mean
#pragma unroll
        for (int j = 0; j < Q; ++j) { beta[0] = fma(beta[2 + j], P.wstd[j], beta[0]); beta[2 + j] = beta[2 + j] * P.wstd[Q + j]; }
    }
    int status = 0;
    double llf = NAN, bse1 = NAN;
    // The fast phase hands over a beta whose last step was <= P.fast_tol.  This pass evaluates llf, the separation callback and the
    // information matrix there in fp64 AND the score, and takes the exact Newton step (no ridge, fp64 Hessian): quadratic from ~1e-7, i.e.
    // the fixed point itself.  Should that step exceed 5e-7 (llf and bse[1] are evaluated BEFORE the step: at 5e-7 they are still good to
    // ~5e-7 relative), the pass is repeated at the stepped beta, so what is reported always satisfies the reference's stopping rule.
    bool redo = fin;
    for (int rep = 0; rep < 6 && __any(redo); ++rep) {
        double H[PC * (PC + 1) / 2], g[PC], ll, maxdev;
        if (DELTA) final_pass_mfma<Q>(T, Vpad, vr, N, y, P.ws, P.zz, P.w0, P.a0, beta, H, g, ll, maxdev, tr);   // wave-wide (MFMA, permlane)
        if (redo) {
            if (!DELTA) info_pass<Q, true, true>(T, Vpad, vr, N, NB64, y, W, beta, H, g, ll, maxdev, true);
            redo = false;
            status = 0;
            if (maxdev <= 1e-8) status = 1;                                                      // callback after the last update
            else {
                llf = ll;
                // Hinv = inv(-Hessian/nobs)/nobs, no ridge (SM:base/model.py:533-534); only bse[1] is used (model.py:332)
#pragma unroll
                for (int a = 0; a < PC * (PC + 1) / 2; ++a) H[a] = H[a] / nobs;
                double det;
                if (!ldl_factor<PC>(H, DELTA ? 1.0e-5 : 4.0e-16, &det)) status = DELTA ? 9 : 2;
                else {
                    double e[PC];
#pragma unroll
                    for (int a = 0; a < PC; ++a) { e[a] = (a == 1) ? 1.0 : 0.0; g[a] = g[a] / nobs; }
                    ldl_solve<PC>(H, e);
                    bse1 = sqrt(e[1] / nobs);
                    ldl_solve<PC>(H, g);
                    double smax = 0.0; bool finite = true;
#pragma unroll
                    for (int a = 0; a < PC; ++a) { smax = fmax(smax, fabs(g[a])); finite = finite && isfinite(g[a]); }
                    if (finite) {
#pragma unroll
                        for (int a = 0; a < PC; ++a) beta[a] += g[a];
                        redo = smax > 5e-7;
                        if (redo && P.dbg) atomicAdd(&P.dbg[4], 1);
                    }
                }
            }
        }
    }
    if (!fin) return;
    if (DELTA && status == 9) { wk.state[v] = 2; return; }           // the fp64 pass decides (bw still holds the beta that came in)
    glm_emit<Q>(status, bse1, llf, beta, DELTA, v, V, P, out, flags, firth_list, firth_count);
}

// ---- the finishing rounds: what k_glm_final<Q, true> does, as three lean kernels over the list of converged variants ------------
//   k_glm_ll      fp64: eta, mu, the log-likelihood, the separation callback's max |y - mu| and the score        (~70 VGPRs)
//   k_glm_dpass   fp32: X^T (W - W0) X, every entry a difference from the null model's -- the z x z block on the matrix pipe against the
//                 products table, the intercept and variant rows on the VALU.  The parts they are differences FROM are exact: the null model's
//                 block (null_h, a0) and, for the variant's row, the carrier sums of k_glm_bitdot.
//   k_glm_finish  fp64: assemble and factor the information matrix, bse, the exact Newton step (certificate: <= 5e-7, else the variant is
//                 restarted by k_glm_slow), decisions and the output row.
#ifndef GLM_LL_BLOCKS
#define GLM_LL_BLOCKS 1
#endif
template <int Q>
__global__ __launch_bounds__(256, GLM_LL_BLOCKS) void k_glm_ll(const uint64_t *__restrict__ T, int64_t Vpad, const double *__restrict__ y, GlmParams P,
                                                const int *__restrict__ list, const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, RS = Q + 1;
    if ((int64_t)blockIdx.x * 256 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 256 + threadIdx.x, v);
    if (!__any(on)) return;
    const int N = P.N, NB64 = P.NB64;
    const double *__restrict__ R = P.rec;
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) { beta[a] = P.ch_bs[(int64_t)a * Vpad + v]; g[a] = 0.0; }
    // y is 0/1 here (sh_glm_setup turns the finishing kernels off otherwise): ll_i = -softplus(a_i), a_i = (1 - 2 y_i) eta_i (= log mu_i for
    // y = 1, log(1 - mu_i) for y = 0: SM Logit.loglike), and with t = exp(-|eta|):  softplus(a) = max(a, 0) + log(1 + t),  mu = 1 / (1 + t)
    // or t / (1 + t).  The logs are not taken one by one: the factors 1 + t in (1, 2] are MULTIPLIED (one rounding each, like a sum's),
    // renormalised once per 64-sample word, and one log at the end turns the product into the sum.  One exp (argument reduced, degree-13
    // polynomial, v_ldexp) and one reciprocal per sample, no log.  The sample's record is fetched one sample ahead.
    double maxdev = 0.0, apos = 0.0, prod = 1.0;
    int pexp = 0;
    auto one = [&](const double (&rc)[RS], bool xb) {
        double eta = beta[0] + (xb ? beta[1] : 0.0);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double yi = rc[Q];
        const double t = exp_neg(fabs(eta)), u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);                                      // two Newton steps: 1 / u to the last bit or two
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        const double mu = (eta >= 0.0) ? inv : t * inv;
        const double r = yi - mu;
        apos += fmax(fma(-2.0 * yi, eta, eta), 0.0);                              // a = (1 - 2 y) eta
        prod *= u;
        maxdev = fmax(maxdev, fabs(r));
        g[0] += r; g[1] += xb ? r : 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) g[2 + j] = fma(r, rc[j], g[2 + j]);
    };
    double ra[RS], rb[RS];                                           // two record buffers in turn, as in k_glm_score
#pragma unroll
    for (int k = 0; k < RS; ++k) ra[k] = R[k];
    for (int wd = 0; wd < NB64; ++wd) {
        const uint64_t w = T[(int64_t)wd * Vpad + v];
        const int lim = min(64, N - wd * 64);
        if (lim == 64) {
            for (int b = 0; b < 64; b += 2) {
                const int i = wd * 64 + b, i2 = min(i + 2, N - 1);
                const int za = pipe_zero(ra[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                one(ra, (w >> b) & 1ull);
                const int zb = pipe_zero(rb[0]);
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                one(rb, (w >> (b + 1)) & 1ull);
            }
        } else {
            for (int b = 0; b < lim; ++b) {
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(wd * 64 + b) * RS + k];
                one(ra, (w >> b) & 1ull);
            }
        }
        int e2; prod = frexp(prod, &e2); pexp += e2;
    }
    if (!on) return;
#pragma unroll
    for (int a = 0; a < PC; ++a) P.ch_g[(int64_t)a * Vpad + v] = g[a];
    P.ch_md[v] = maxdev; P.ch_ll[v] = -(apos + fma((double)pexp, 0.6931471805599453, log(prod)));
}

template <int Q>
__global__ __launch_bounds__(64, 3) void k_glm_dpass(const uint64_t *__restrict__ T, int64_t Vpad, const float *__restrict__ Wf, GlmParams P,
                                                     const int *__restrict__ list, const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + threadIdx.x, v);
    const int N = P.N;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    const float *__restrict__ ZZ = P.zz;
    const double *__restrict__ w0 = P.w0;
    __shared__ float tr[FastCols<Q>::LDS_FLOATS];
    float bf[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) bf[a] = (float)P.ch_bs[(int64_t)a * Vpad + v];
    v16f acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][h][r] = 0.0f;
    float h00 = 0.0f, h10 = 0.0f, hz0[Q > 0 ? Q : 1], hz1[Q > 0 ? Q : 1];
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = 0.0f; hz1[j] = 0.0f; }
    auto sample = [&](int i, bool xb) -> float {
        float eta = bf[0] + (xb ? bf[1] : 0.0f);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j], Wf[(int64_t)i * Q + j], eta);
        const float mu = 1.0f / (1.0f + __expf(-eta));
        const float d = mu * (1.0f - mu) - (float)w0[i];
        const float dx = xb ? d : 0.0f;
        h00 += d; h10 += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float zj = Wf[(int64_t)i * Q + j]; hz0[j] = fmaf(d, zj, hz0[j]); hz1[j] = fmaf(dx, zj, hz1[j]); }
        return d;
    };
    const int nfull = N >> 1;
    for (int pr = 0; pr < nfull; ++pr) {
        const int i = 2 * pr, b = i & 63;
        const float *zrow = ZZ + (int64_t)(i + lh) * STRIDE + l31;
        float bz[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bz[cb] = zrow[cb * 32];
        const uint64_t w = T[(int64_t)(i >> 6) * Vpad + v];
        const float d0 = sample(i, (w >> b) & 1ull);
        const float d1 = sample(i + 1, (w >> (b + 1)) & 1ull);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false);
        const float a0f = __uint_as_float(sw[0]), a1f = __uint_as_float(sw[1]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0f, bz[cb], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1f, bz[cb], acc[cb][1], 0, 0, 0);
        }
    }
    if (N & 1) {
        const int i = N - 1;
        const float *zrow = ZZ + (int64_t)i * STRIDE + l31;
        const float d0 = sample(i, (T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), 0u, false, false);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), zrow[cb * 32], acc[cb][0], 0, 0, 0);
            acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), zrow[cb * 32], acc[cb][1], 0, 0, 0);
        }
    }
    float H[NH];
    H[sidx(0, 0)] = h00; H[sidx(1, 0)] = h10; H[sidx(1, 1)] = h10;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0[j]; H[sidx(2 + j, 1)] = hz1[j]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + cb * 32 + l31] = acc[cb][h][r];
        __syncthreads();
        if (lh == h) {
            const float *row = tr + l31 * (STRIDE + 1);
#pragma unroll
            for (int j = 0; j < Q; ++j)
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = row[j * (j + 1) / 2 + k];
        }
    }
    if (!on) return;
#pragma unroll
    for (int a = 0; a < NH; ++a) P.ch_hf[(int64_t)a * Vpad + v] = H[a];
}

template <int Q>
__global__ __launch_bounds__(64, 2) void k_glm_dpass_pk(const uint64_t *__restrict__ T, int64_t Vpad, const float *__restrict__ Wf, GlmParams P,
                                                        const int *__restrict__ list, const int *__restrict__ cnt)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + threadIdx.x, v);
    __shared__ float tr[FastCols<Q>::LDS_FLOATS];
    __shared__ double hdl[(2 + 2 * Q) * 64];
#pragma unroll
    for (int k = 0; k < 2 + 2 * Q; ++k) hdl[k * 64 + threadIdx.x] = 0.0;     // a lane only ever touches its own column
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = P.ch_bs[(int64_t)a * Vpad + v];
    float H[NH];
    pass32_pk<Q, true>(T, Vpad, v, P, Wf, beta, H, g, tr, 0, 1, hdl);
    if (!on) return;
#pragma unroll
    for (int a = 0; a < NH; ++a) P.ch_hf[(int64_t)a * Vpad + v] = H[a];
}

// DIRECT (GlmParams.ll_first): the score, log-likelihood and callback value on record were taken at ch_b0, one chord step in front of ch_bs (where
// the information matrix was just evaluated): the result is the exact Newton step from ch_b0 with that matrix, beta = ch_b0 + H^-1 g (error
// O(step^2), step <= 1e-4), its log-likelihood ll(ch_b0) + g . step / 2 (error O(N step^3)); the certificate is the distance of that beta from
// the chord result (what the chord factor got wrong), <= 5e-7 as for the step of the plain form.
template <int Q, bool DIRECT = false>
__global__ __launch_bounds__(64) void k_glm_finish(int64_t Vpad, int64_t V, GlmParams P, GlmWork wk, const int *__restrict__ list,
                                                   const int *__restrict__ cnt, double *__restrict__ out, uint32_t *__restrict__ flags,
                                                   int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    if ((int64_t)blockIdx.x * 64 >= *cnt) return;
    int64_t v;
    const bool on = round_lane(list, cnt, (int64_t)blockIdx.x * 64 + threadIdx.x, v);
    const double nobs = (double)P.N;
    bool go_slow = false;
    if (on) {
        double beta[PC];
#pragma unroll
        for (int a = 0; a < PC; ++a) beta[a] = P.ch_bs[(int64_t)a * Vpad + v];
        int status = 0;
        double bse1 = NAN;
        double llf_adj = 0.0;
        bool emit = true;
        if (P.ch_md[v] <= 1e-8) status = 1;                                                       // callback after the last update
        else {
            double H[NH], g[PC];
            H[sidx(0, 0)] = P.null_h[0]; H[sidx(1, 0)] = H[sidx(1, 1)] = P.ch_bd[v];
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                H[sidx(2 + j, 0)] = P.null_h[1 + j]; H[sidx(2 + j, 1)] = P.ch_bd[(int64_t)(1 + j) * Vpad + v];
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = P.a0[j * (j + 1) / 2 + k];
            }
#pragma unroll
            for (int a = 0; a < NH; ++a) H[a] = (H[a] + (double)P.ch_hf[(int64_t)a * Vpad + v]) / nobs;
            double det;
            if (!ldl_factor<PC>(H, 1.0e-5, &det)) {                  // (nearly) singular: the all-fp64 restart decides (an exactly singular design must be SEEN as such)
                emit = false; go_slow = true;
            } else {
                double e[PC];
#pragma unroll
                for (int a = 0; a < PC; ++a) { e[a] = (a == 1) ? 1.0 : 0.0; g[a] = P.ch_g[(int64_t)a * Vpad + v] / nobs; }
                ldl_solve<PC>(H, e);
                bse1 = sqrt(e[1] / nobs);
                double gl = 0.0;                                    // g . step over the samples (DIRECT: the log-likelihood's second-order term)
                ldl_solve<PC>(H, g);
                double smax = 0.0, stepmax = 0.0; bool finite = true;
                if (DIRECT) {
                    gl = 0.0;
#pragma unroll
                    for (int a = 0; a < PC; ++a) {
                        const double b0 = P.ch_b0[(int64_t)a * Vpad + v], bn = b0 + g[a];
                        gl = fma(P.ch_g[(int64_t)a * Vpad + v], g[a], gl);
                        smax = fmax(smax, fabs(bn - beta[a])); stepmax = fmax(stepmax, fabs(g[a])); finite = finite && isfinite(g[a]);
                        g[a] = bn - beta[a];                        // what is added to beta (= the chord result) below
                    }
                    if (stepmax > 2e-4) smax = 1.0;                 // (the chord kernel let <= 1e-4 through; H^-1 g and the chord step differ by << that)
                } else {
#pragma unroll
                    for (int a = 0; a < PC; ++a) { smax = fmax(smax, fabs(g[a])); finite = finite && isfinite(g[a]); }
                }
                if (!finite || smax > 5e-7) { emit = false; go_slow = true; if (P.dbg) atomicAdd(&P.dbg[4], 1); }
                else {
#pragma unroll
                    for (int a = 0; a < PC; ++a) beta[a] += g[a];
                    if (DIRECT) llf_adj = 0.5 * gl;
                }
            }
        }
        if (emit) glm_emit<Q>(status, bse1, P.ch_ll[v] + llf_adj, beta, true, v, V, P, out, flags, firth_list, firth_count);
    }
    list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
}


// =====================================================================================================================
// Firth as a device-resident state machine (round 1's single-kernel form is gone).
//
// fit_firth (model.py:414-504) alternates two sample passes: the penalised score at beta (hat diagonal through the factored
// information) and the penalised likelihood at a candidate (a fresh information matrix, its determinant, the step-halving
// test).  Each pass is its own kernel so that each is compiled to <= 256 VGPRs (two waves per SIMD instead of one 400-register
// wave), and between passes the live variants are re-listed, so a wavefront never idles on lanes that have converged or are
// in the other phase.  Per-variant state lives in HBM, SoA over the slot index: beta, cand, the LDL^T factor of I(beta),
// F(beta), I11, the previous step norm, counters.
// =====================================================================================================================
template <int Q>
__global__ __launch_bounds__(64) void k_firth_init(const int *__restrict__ firth_list, const int *__restrict__ firth_count, GlmParams P,
                                                   FirthWork fw, int *__restrict__ eval_list, int *__restrict__ eval_count)
{
    constexpr int PC = Q + 2;
    const int cnt = *firth_count;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s == 0) *eval_count = cnt;
    if (s >= cnt) return;
    // Start vector: the reference's (model.py:323-324), or -- Firth on every variant (force_firth) -- the null-model fit (GlmParams.fwarm).
    // fit_firth stops one iteration AFTER a step below 1e-4 (model.py:477-479) and contracts by ~p/N per iteration, so at large N (the only
    // case that sets firth_warm, see sh_glm_setup) the start shows in the result below 1e-7.  Where it converges slowly the path matters
    // more: a variant still iterating FIRTH_WARM_LIMIT accepted steps after a warm start is restarted from the reference's vector.
    // A warm slot is marked by the complemented variant index in fw.var.
    const bool warm = P.firth_warm != 0;
#pragma unroll
    for (int a = 0; a < PC; ++a) {
        const double b0 = warm ? ((a == 0) ? P.fwarm[0] : (a == 1) ? 0.0 : P.fwarm[a - 1]) : ((a == 0) ? P.ymean_logit : 0.0);
        fw.st[(int64_t)(fw_beta<PC>() + a) * fw.cap + s] = b0;
        fw.st[(int64_t)(fw_cand<PC>() + a) * fw.cap + s] = b0;
    }
    fw.st[(int64_t)fw_snp<PC>() * fw.cap + s] = INFINITY;
    fw.iter[s] = -1; fw.halv[s] = 0; fw.var[s] = warm ? ~firth_list[s] : firth_list[s];
    eval_list[s] = s;
}

// penalised likelihood at cand; accept / halve / converge / fail (the state == 1 arm of k_glm_firth)
#ifndef FIRTH_EVAL_THREADS
#define FIRTH_EVAL_THREADS 512     /* A/B: 256 = four wavefronts per block, one per SIMD, 512 registers each (36 instead of 820 bytes of scratch): 61.5 vs 60.1 ms per C4 batch, not kept */
#endif
template <int Q, bool LEAN>
__global__ __launch_bounds__(LEAN ? FIRTH_EVAL_THREADS : 512) void k_firth_eval(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                      const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                      FirthWork fw, const int *__restrict__ eval_list, const int *__restrict__ eval_count,
                                                      int *__restrict__ next_eval, int *__restrict__ next_eval_count,
                                                      int *__restrict__ step_list, int *__restrict__ step_count,
                                                      double *__restrict__ out, uint32_t *__restrict__ flags,
                                                      int *__restrict__ pinv_list, int *__restrict__ pinv_count)
{
    constexpr int PC = Q + 2;
    const double SING_TOL = 1e-12;
    const int cnt = *eval_count;
    if ((int64_t)blockIdx.x * 64 >= cnt) return;
    const XWave xw = xwave();
    const int li = blockIdx.x * 64 + xw.lane;
    const bool live = li < cnt;
    const int s = eval_list[live ? li : 0];
    const int vraw = fw.var[s];
    const bool warm = vraw < 0;
    const int64_t v = warm ? ~vraw : vraw;
    const int64_t cap = fw.cap;
    double cand[PC], A[PC * (PC + 1) / 2], dummy[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s];
    double ll, maxdev, det;
    if constexpr (LEAN) info_pass_bin<Q>(T, Vpad, v, P.N, P.NB64, P.rec_o, cand, A, ll, maxdev, xw.w, xw.S);
    else info_pass<Q, false, true>(T, Vpad, v, P.N, P.NB64, y, W, cand, A, dummy, ll, maxdev, true, xw.w, xw.S);
    xw_sum(xw, A); xw_sum_max(xw, ll, maxdev);
    if (!live || xw.w != 0) return;
    const double i11c = A[sidx(1, 1)];
    const bool singular = !ldl_factor<PC>(A, SING_TOL, &det);
    if (singular) {                     // handled by k_glm_firth_pinv (numpy.linalg.pinv semantics, model.py:450)
        const int s2 = atomicAdd(pinv_count, 1); pinv_list[s2] = (int)v;
        return;
    }
    const double Fcand = -(ll + 0.5 * log(det));                     // firth_likelihood, model.py:410-411
    int iter = fw.iter[s];
    bool accept = true, failed = false, conv = false;
    double sn = 0.0, sn_tested = INFINITY;
    if (iter < 0) {                                                  // F(beta_0): nothing to compare with
        iter = 0;
    } else {
        const double Fcur = fw.st[(int64_t)fw_fcur<PC>() * cap + s];
        double stepmax = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) {
            const double d = cand[a] - fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
            stepmax = fmax(stepmax, fabs(d)); sn = fma(d, d, sn);
        }
        // see k_glm_firth: steps below 1e-10 are accepted outright (F(new) > F(old) is rounding noise there)
        if (Fcand > Fcur + P.firth_noise * fabs(Fcur) && !(stepmax < P.firth_accept)) {   // step halving, model.py:467-474
            accept = false;
            const int h = fw.halv[s] + 1;
            fw.halv[s] = h;
            bool moved = false;                                      // the halving map at its fixed point: see k_firth_eval2 (firth_rounds.hip)
            double nc[PC];
#pragma unroll
            for (int a = 0; a < PC; ++a) {
                const double b = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
                nc[a] = b + 0.5 * (cand[a] - b);
                moved = moved || (nc[a] != cand[a]);
            }
            if (h > 1000 || !moved) failed = true;
            else {
#pragma unroll
                for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = nc[a];
                next_eval[atomicAdd(next_eval_count, 1)] = s;
            }
        } else {
            sn = sqrt(sn);
            const double snp = fw.st[(int64_t)fw_snp<PC>() * cap + s];
            conv = (iter > 0) && (snp < 1e-4);                       // tests the PREVIOUS step, model.py:477-479
            sn_tested = snp;
            fw.st[(int64_t)fw_snp<PC>() * cap + s] = sn;
            ++iter;
            if (!conv && iter >= 1000) failed = true;                // step_limit exhausted, model.py:482-484
        }
    }
    if (accept && !failed && !conv && warm && iter >= FIRTH_WARM_LIMIT) {   // not a quadratic convergence: the reference's own path decides
#pragma unroll
        for (int a = 0; a < PC; ++a) {
            const double b0 = (a == 0) ? P.ymean_logit : 0.0;
            fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s] = b0;
            fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = b0;
        }
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = INFINITY;
        fw.iter[s] = -1; fw.halv[s] = 0; fw.var[s] = (int)v;
        next_eval[atomicAdd(next_eval_count, 1)] = s;
        return;
    }
    if (accept && !failed && !conv) {                                // beta <- cand; keep the factor for the score pass
#pragma unroll
        for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s] = cand[a];
#pragma unroll
        for (int a = 0; a < PC * (PC + 1) / 2; ++a) fw.st[(int64_t)(fw_fac<PC>() + a) * cap + s] = A[a];
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = Fcand;
        fw.iter[s] = iter; fw.halv[s] = 0;
        if (iter >= P.firth_handoff) fw.blk_list[atomicAdd(fw.blk_count, 1)] = s;
        else step_list[atomicAdd(step_count, 1)] = s;
        return;
    }
    if (!failed && !conv) return;                                    // halved: queued above
    uint32_t fl = flags[v] | firth_sensitive(iter, sn_tested);
    if (failed) {
        fl |= SH_NOTE_FIRTH_FAIL | SH_FLAG_FILTER | SH_FLAG_FIRTH_SENSITIVE;                           // model.py:357-362
        out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
#pragma unroll
        for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = NAN;
    } else {
        const double fitll = -Fcand;
        const double lrstat = -2.0 * (P.null_firth - fitll);
        double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);      // model.py:366-369
        out[V + v] = pval; out[2 * V + v] = cand[1]; out[3 * V + v] = sqrt(i11c); out[4 * V + v] = cand[0];   // bse = sqrt(I11), model.py:491
#pragma unroll
        for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = cand[2 + j];
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(cand[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
    }
    flags[v] = fl;
}

// penalised score at beta through the stored factor, Newton step -> cand (the state == 0 arm of k_glm_firth)
template <int Q, bool LEAN>
__global__ __launch_bounds__(512) void k_firth_step(const uint64_t *__restrict__ T, int64_t Vpad, const double *__restrict__ y,
                                                      const double *__restrict__ W, GlmParams P, FirthWork fw,
                                                      const int *__restrict__ step_list, const int *__restrict__ step_count,
                                                      int *__restrict__ next_eval, int *__restrict__ next_eval_count)
{
    constexpr int PC = Q + 2;
    const int cnt = *step_count;
    if ((int64_t)blockIdx.x * 64 >= cnt) return;
    const XWave xw = xwave();
    const int li = blockIdx.x * 64 + xw.lane;
    const bool live = li < cnt;
    const int s = step_list[live ? li : 0];
    const int64_t v = fw.var[s] < 0 ? ~fw.var[s] : fw.var[s];
    const int64_t cap = fw.cap;
    const int N = P.N, NB64 = P.NB64;
    double beta[PC], A[PC * (PC + 1) / 2], U[PC], dinv[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
#pragma unroll
    for (int a = 0; a < PC * (PC + 1) / 2; ++a) A[a] = fw.st[(int64_t)(fw_fac<PC>() + a) * cap + s];
#pragma unroll
    for (int a = 0; a < PC; ++a) { U[a] = 0.0; dinv[a] = 1.0 / A[sidx(a, a)]; }
    constexpr int RS = Q + 1;
    // one sample: x = (1, bit, covariates).  LEAN (y in {0, 1}, GlmParams.rec_o): mu by exp_neg + v_rcp_f64 as in info_pass_bin, and the
    // sample's record (covariates, y) fetched one sample ahead; else the plain form
    auto one = [&](const double (&cv)[RS], bool xbit) {
        double x[PC];
        x[0] = 1.0; x[1] = xbit ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) x[2 + j] = cv[j];
        double eta = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
        double mu, wgt;
        if constexpr (LEAN) {
            const double t = exp_neg(fabs(eta)), u = 1.0 + t;
            double inv = __builtin_amdgcn_rcp(u);
            inv = fma(fma(-u, inv, 1.0), inv, inv);
            inv = fma(fma(-u, inv, 1.0), inv, inv);
            mu = (eta >= 0.0) ? inv : t * inv; wgt = t * inv * inv;
        } else { mu = logit_cdf(eta); wgt = mu * (1.0 - mu); }
        double zt[PC]; double qf = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) {
            double t = x[a];
#pragma unroll
            for (int k = 0; k < a; ++k) t = fma(-A[sidx(a, k)], zt[k], t);
            zt[a] = t;
            qf = fma(t * t, dinv[a], qf);
        }
        const double h = wgt * qf;                                   // diagonal of the hat matrix, model.py:455-462
        const double res = cv[Q] - mu + h * (0.5 - mu);
#pragma unroll
        for (int a = 0; a < PC; ++a) U[a] = fma(x[a], res, U[a]);
    };
    double ra[RS], rb[RS];
    if constexpr (LEAN) {
        const double *__restrict__ R = P.rec_o;
        {
            const int i0 = min(xw.w * 64, N - 1);
#pragma unroll
            for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i0 * RS + k];
        }
        for (int sb = xw.w; sb < NB64; sb += xw.S) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
            if (nb == 64) {
                const int inext = min((sb + xw.S) * 64, N - 1);
                for (int b = 0; b < 64; b += 2) {
                    const int i = sb * 64 + b, i2 = (b == 62) ? inext : i + 2;
                    const int za = pipe_zero(ra[0]);
#pragma unroll
                    for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                    one(ra, (w64 >> b) & 1ull);
                    const int zb = pipe_zero(rb[0]);
#pragma unroll
                    for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                    one(rb, (w64 >> (b + 1)) & 1ull);
                }
            } else {
                for (int b = 0; b < nb; ++b) {
#pragma unroll
                    for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(sb * 64 + b) * RS + k];
                    one(ra, (w64 >> b) & 1ull);
                }
            }
        }
    } else {
        for (int sb = xw.w; sb < NB64; sb += xw.S) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
            for (int b = 0; b < nb; ++b) {
                const int i = sb * 64 + b;
#pragma unroll
                for (int j = 0; j < Q; ++j) ra[j] = W[(int64_t)i * Q + j];
                ra[Q] = y[i];
                one(ra, (w64 >> b) & 1ull);
            }
        }
    }
    xw_sum(xw, U);
    if (!live || xw.w != 0) return;
    ldl_solve<PC>(A, U);                                             // var_covar_mat . U, model.py:463
#pragma unroll
    for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = beta[a] + U[a];
    next_eval[atomicAdd(next_eval_count, 1)] = s;
}

// ---- workgroup reductions for the one-variant-per-workgroup kernels: lanes by xor-shuffle, then waves 0..3, result in thread 0
template <int NA>
__device__ __forceinline__ void blk_sum(double (&a)[NA], double *red /* [4][NA] */, int tid)
{
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        double t = a[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
        a[k] = t;
    }
    __syncthreads();                                              // red may still be read from the previous use
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) red[(tid >> 6) * NA + k] = a[k];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) a[k] = ((red[k] + red[NA + k]) + red[2 * NA + k]) + red[3 * NA + k];
    }
}

// packed lower I(b) and the log-likelihood at b (b in shared memory), samples t, t+256, ... per thread; result in thread 0
template <int PC>
__device__ __forceinline__ void blk_info_packed(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N,
                                                const double *__restrict__ y, const double *__restrict__ W, const double *b_sh,
                                                double (&acc)[PC * (PC + 1) / 2 + 1], double *red, int tid)
{
    constexpr int Q = PC - 2, NH = PC * (PC + 1) / 2;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < NH + 1; ++a) acc[a] = 0.0;
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = b_sh[a];
    for (int i = tid; i < N; i += 256) {
        const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
        double x[PC];
        x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
        for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
        double eta = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
        const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
        const double yi = y[i], lm = log(mu);
        acc[NH] += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));   // as info_pass
#pragma unroll
        for (int a = 0; a < PC; ++a) {
            const double wa = wgt * x[a];
#pragma unroll
            for (int c = 0; c <= a; ++c) acc[sidx(a, c)] = fma(wa, x[c], acc[sidx(a, c)]);
        }
    }
    blk_sum<NH + 1>(acc, red, tid);
}

// Continues fit_firth for the slots on the hand-off list, one workgroup per variant, from the state the rounds left (beta, F(beta),
// previous step norm, step count): the same iteration and the same decisions as k_firth_eval / k_firth_step, with the samples
// spread over 256 threads, so an iteration costs ~30 us instead of two latency-bound launches.
template <int Q>
__global__ __launch_bounds__(256) void k_firth_blk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                   const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                   FirthWork fw, double *__restrict__ out, uint32_t *__restrict__ flags,
                                                   int *__restrict__ pinv_list, int *__restrict__ pinv_count)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    const double SING_TOL = 1e-12;
    __shared__ double s_beta[PC], s_cand[PC], s_fac[NH], s_dinv[PC], s_red[4 * (NH + 1)];
    __shared__ int s_ctl;                                          // 0 = evaluate the candidate again, 1 = step accepted, 2 = variant done
    const int cnt = *fw.blk_count, tid = threadIdx.x, N = P.N;
    const int64_t cap = fw.cap;
    for (int idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const int s = fw.blk_list[idx];
        const int64_t v = fw.var[s] < 0 ? ~fw.var[s] : fw.var[s];
        double acc[NH + 1];
        // thread-0 state
        double Fcur = 0.0, snp = 0.0;
        int iter = 0, halv = 0;
        // a slot handed over in the middle of a step halving (k_firth_eval2 after FIRTH_HALV_HANDOFF rejections): its pending candidate and
        // halving count are in the state; the first pass below skips the score and goes straight to the comparison, against F(beta) as
        // THIS kernel evaluates it
        bool pending = fw.halv[s] > 0;
        __syncthreads();
        if (tid == 0) {
            for (int a = 0; a < PC; ++a) s_beta[a] = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
            if (pending) { for (int a = 0; a < PC; ++a) s_cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s]; halv = fw.halv[s]; }
            Fcur = fw.st[(int64_t)fw_fcur<PC>() * cap + s]; snp = fw.st[(int64_t)fw_snp<PC>() * cap + s]; iter = fw.iter[s];
        }
        __syncthreads();
        blk_info_packed<PC>(T, Vpad, v, N, y, W, s_beta, acc, s_red, tid);             // the factor of I(beta) for the score pass
        if (tid == 0) {
            double A[NH], det;
            for (int a = 0; a < NH; ++a) A[a] = acc[a];
            if (!ldl_factor<PC>(A, SING_TOL, &det)) { pinv_list[atomicAdd(pinv_count, 1)] = (int)v; s_ctl = 2; }
            else {
                for (int a = 0; a < NH; ++a) s_fac[a] = A[a]; for (int a = 0; a < PC; ++a) s_dinv[a] = 1.0 / A[sidx(a, a)]; s_ctl = 1;
                // F(beta) as THIS kernel evaluates it: the step-halving test compares F values to 4 ulp, and the rounds' F (info_pass_bin: logs
                // folded into a product; another summation order) differs from this kernel's by more than that
                Fcur = -(acc[NH] + 0.5 * log(det));
            }
        }
        __syncthreads();
        while (s_ctl != 2) {
            // ---- penalised score at beta through the factor, Newton step -> cand (k_firth_step)
            double U[PC], beta[PC];
#pragma unroll
            for (int a = 0; a < PC; ++a) { U[a] = 0.0; beta[a] = s_beta[a]; }
            for (int i = tid; i < N && !pending; i += 256) {
                const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
                double x[PC];
                x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
                for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
                double eta = 0.0;
#pragma unroll
                for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
                const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
                double zt[PC], qf = 0.0;
#pragma unroll
                for (int a = 0; a < PC; ++a) {
                    double t = x[a];
#pragma unroll
                    for (int k = 0; k < a; ++k) t = fma(-s_fac[sidx(a, k)], zt[k], t);
                    zt[a] = t;
                    qf = fma(t * t, s_dinv[a], qf);
                }
                const double res = y[i] - mu + wgt * qf * (0.5 - mu);                    // model.py:455-462
#pragma unroll
                for (int a = 0; a < PC; ++a) U[a] = fma(x[a], res, U[a]);
            }
            if (!pending) {
                blk_sum<PC>(U, s_red, tid);
                if (tid == 0) {
                    double A[NH];
                    for (int a = 0; a < NH; ++a) A[a] = s_fac[a];
                    ldl_solve<PC>(A, U);                                               // var_covar_mat . U, model.py:463
                    for (int a = 0; a < PC; ++a) s_cand[a] = s_beta[a] + U[a];
                    halv = 0;
                }
            }
            pending = false;
            // ---- penalised likelihood at cand; accept / halve / converge / fail (k_firth_eval)
            for (;;) {
                __syncthreads();
                blk_info_packed<PC>(T, Vpad, v, N, y, W, s_cand, acc, s_red, tid);
                if (tid == 0) {
                    double A[NH], det;
                    for (int a = 0; a < NH; ++a) A[a] = acc[a];
                    const double i11c = A[sidx(1, 1)];
                    if (!ldl_factor<PC>(A, SING_TOL, &det)) { pinv_list[atomicAdd(pinv_count, 1)] = (int)v; s_ctl = 2; }
                    else {
                        const double Fcand = -(acc[NH] + 0.5 * log(det));
                        double stepmax = 0.0, sn = 0.0;
                        for (int a = 0; a < PC; ++a) { const double d = s_cand[a] - s_beta[a]; stepmax = fmax(stepmax, fabs(d)); sn = fma(d, d, sn); }
                        bool failed = false, conv = false;
                        if (Fcand > Fcur + P.firth_noise * fabs(Fcur) && !(stepmax < P.firth_accept)) {   // step halving, model.py:467-474
                            bool moved = false;                                          // fixed point of the halving map = the reference's 1000 identical comparisons
                            for (int a = 0; a < PC; ++a) { const double nc = s_beta[a] + 0.5 * (s_cand[a] - s_beta[a]); moved = moved || (nc != s_cand[a]); s_cand[a] = nc; }
                            if (++halv > 1000 || !moved) failed = true;
                            else s_ctl = 0;
                        } else {
                            sn = sqrt(sn);
                            conv = (iter > 0) && (snp < 1e-4);                           // the PREVIOUS step, model.py:477-479
                            const double sn_tested = snp;
                            snp = sn; ++iter;
                            if (failed || conv || iter >= 1000) flags[v] |= firth_sensitive(iter, sn_tested);
                            if (!conv && iter >= 1000) failed = true;                    // step_limit exhausted, model.py:482-484
                            if (!conv && !failed) {
                                for (int a = 0; a < PC; ++a) s_beta[a] = s_cand[a];
                                for (int a = 0; a < NH; ++a) s_fac[a] = A[a];
                                for (int a = 0; a < PC; ++a) s_dinv[a] = 1.0 / A[sidx(a, a)];
                                Fcur = Fcand; s_ctl = 1;
                            }
                        }
                        if (failed || conv) {
                            uint32_t fl = flags[v];
                            if (failed) {
                                fl |= SH_NOTE_FIRTH_FAIL | SH_FLAG_FILTER | SH_FLAG_FIRTH_SENSITIVE;               // model.py:357-362
                                out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
                                for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = NAN;
                            } else {
                                const double lrstat = -2.0 * (P.null_firth - (-Fcand));
                                double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);      // model.py:366-369
                                out[V + v] = pval; out[2 * V + v] = s_cand[1]; out[3 * V + v] = sqrt(i11c); out[4 * V + v] = s_cand[0];
                                for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = s_cand[2 + j];
                                if (pval > P.lrtt || !isfinite(pval) || !isfinite(s_cand[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
                            }
                            flags[v] = fl;
                            s_ctl = 2;
                        }
                    }
                }
                __syncthreads();
                if (s_ctl != 0) break;
            }
            __syncthreads();
        }
    }
}

// =====================================================================================================================
// Firth slow path: literal restatement of fit_firth with numpy.linalg.pinv semantics (model.py:450) for variants whose
// information matrix is (near-)singular, e.g. a k-mer that duplicates a binary covariate.
// =====================================================================================================================
// numpy.linalg.det: LU with partial pivoting on a copy
template <int PC>
__device__ __noinline__ double slow_det(const double *Ain)
{
    double A[PC * PC];
    for (int a = 0; a < PC * PC; ++a) A[a] = Ain[a];
    double det = 1.0;
#pragma unroll 1
    for (int c = 0; c < PC; ++c) {
        int p = c; double best = fabs(A[c * PC + c]);
#pragma unroll 1
        for (int r = c + 1; r < PC; ++r) { const double t = fabs(A[r * PC + c]); if (t > best) { best = t; p = r; } }
        if (p != c) {
#pragma unroll 1
            for (int j = 0; j < PC; ++j) { const double t = A[c * PC + j]; A[c * PC + j] = A[p * PC + j]; A[p * PC + j] = t; }
            det = -det;
        }
        const double d = A[c * PC + c];
        det *= d;
        if (d == 0.0) return 0.0;
#pragma unroll 1
        for (int r = c + 1; r < PC; ++r) {
            const double f = A[r * PC + c] / d;
#pragma unroll 1
            for (int j = c + 1; j < PC; ++j) A[r * PC + j] = fma(-f, A[c * PC + j], A[r * PC + j]);
        }
    }
    return det;
}

// numpy.linalg.pinv of a symmetric matrix: cyclic Jacobi eigen-decomposition, eigenvalues <= 1e-15 * max dropped
template <int PC>
__device__ __noinline__ void slow_pinv(const double *Ain, double *Pm, double rcond = 1e-15, int *rank = nullptr)
{
    double A[PC * PC], Vv[PC * PC];
    for (int a = 0; a < PC * PC; ++a) { A[a] = Ain[a]; Vv[a] = 0.0; }
    for (int a = 0; a < PC; ++a) Vv[a * PC + a] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < PC; ++i) { dg = fma(A[i * PC + i], A[i * PC + i], dg); for (int j = i + 1; j < PC; ++j) off = fma(A[i * PC + j], A[i * PC + j], off); }
        if (off <= 1e-34 * (dg + off) || off == 0.0) break;
#pragma unroll 1
        for (int p = 0; p < PC; ++p)
#pragma unroll 1
            for (int q = p + 1; q < PC; ++q) {
                const double apq = A[p * PC + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * PC + q] - A[p * PC + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll 1
                for (int k = 0; k < PC; ++k) { const double akp = A[k * PC + p], akq = A[k * PC + q]; A[k * PC + p] = c * akp - s * akq; A[k * PC + q] = s * akp + c * akq; }
#pragma unroll 1
                for (int k = 0; k < PC; ++k) { const double apk = A[p * PC + k], aqk = A[q * PC + k]; A[p * PC + k] = c * apk - s * aqk; A[q * PC + k] = s * apk + c * aqk; }
#pragma unroll 1
                for (int k = 0; k < PC; ++k) { const double vkp = Vv[k * PC + p], vkq = Vv[k * PC + q]; Vv[k * PC + p] = c * vkp - s * vkq; Vv[k * PC + q] = s * vkp + c * vkq; }
            }
    }
    double smax = 0.0;
    for (int i = 0; i < PC; ++i) smax = fmax(smax, fabs(A[i * PC + i]));
    if (rank) { int r = 0; for (int k = 0; k < PC; ++k) r += (fabs(A[k * PC + k]) > rcond * smax) ? 1 : 0; *rank = r; }
#pragma unroll 1
    for (int i = 0; i < PC; ++i)
#pragma unroll 1
        for (int j = 0; j < PC; ++j) {
            double s = 0.0;
#pragma unroll 1
            for (int k = 0; k < PC; ++k) { const double w = A[k * PC + k]; if (fabs(w) > rcond * smax) s = fma(Vv[i * PC + k] / w, Vv[j * PC + k], s); }
            Pm[i * PC + j] = s;
        }
}

// One workgroup of 256 threads per listed variant, thread t takes samples t, t+256, ...: the variants that come here are few
// (a k-mer that duplicates a binary covariate), and a single lane walking all N samples through un-unrolled loops for every
// pass of every iteration cost about a second per batch at N = 5000.  The p x p algebra (numpy's pinv and det) stays on thread
// 0; sums are combined in a fixed order (lanes by xor-shuffle, then waves 0..3).
// I(b) (full PC x PC into I_out, thread 0) and the log-likelihood at b (shared memory vector)
template <int PC>
__device__ __forceinline__ void blk_info(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N,
                                         const double *__restrict__ y, const double *__restrict__ W, const double *b_sh,
                                         double *I_out, double *ll_out, double *red, int tid)
{
    constexpr int Q = PC - 2, NH = PC * (PC + 1) / 2;
    double acc[NH + 1], beta[PC];
#pragma unroll
    for (int a = 0; a < NH + 1; ++a) acc[a] = 0.0;
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = b_sh[a];
    for (int i = tid; i < N; i += 256) {
        const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
        double x[PC];
        x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
        for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
        double eta = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
        const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
        acc[NH] += log(logit_cdf((2.0 * y[i] - 1.0) * eta));
#pragma unroll
        for (int a = 0; a < PC; ++a) {
            const double wa = wgt * x[a];
#pragma unroll
            for (int c = 0; c <= a; ++c) acc[sidx(a, c)] = fma(wa, x[c], acc[sidx(a, c)]);
        }
    }
    blk_sum<NH + 1>(acc, red, tid);
    if (tid == 0) {
#pragma unroll
        for (int a = 0; a < PC; ++a)
#pragma unroll
            for (int c = 0; c <= a; ++c) { I_out[a * PC + c] = acc[sidx(a, c)]; I_out[c * PC + a] = acc[sidx(a, c)]; }
        *ll_out = acc[NH];
    }
}

template <int Q>
__global__ __launch_bounds__(256) void k_glm_firth_pinv(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V,
                                                        const double *__restrict__ y, const double *__restrict__ W,
                                                        GlmParams P, const int *__restrict__ pinv_list,
                                                        const int *__restrict__ pinv_count,
                                                        double *__restrict__ out, uint32_t *__restrict__ flags)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    __shared__ double s_beta[PC], s_cand[PC], s_Vm[PC * PC], s_red[4 * (NH + 1)];
    __shared__ int s_ctl;                                          // 0 = halve again, 1 = step accepted, 2 = done (converged or failed)
    const int cnt = *pinv_count, tid = threadIdx.x, N = P.N;
    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int64_t v = pinv_list[slot];
        // thread-0 state
        double I[PC * PC], ll = 0.0, Fcur = 0.0, Fcand = 0.0, i11 = 0.0, sn_prev = INFINITY;
        bool failed = false, conv = false;
        int halvings = 0;
        __syncthreads();
        if (tid == 0) { for (int a = 0; a < PC; ++a) s_beta[a] = 0.0; s_beta[0] = P.ymean_logit; }
        __syncthreads();
        blk_info<PC>(T, Vpad, v, N, y, W, s_beta, I, &ll, s_red, tid);
        if (tid == 0) { Fcur = -(ll + 0.5 * log(slow_det<PC>(I))); i11 = I[PC + 1]; }
        for (int iter = 0; iter < 1000; ++iter) {
            if (tid == 0) {
                double Vm[PC * PC];
                slow_pinv<PC>(I, Vm);                                                  // model.py:450
                for (int a = 0; a < PC * PC; ++a) s_Vm[a] = Vm[a];
            }
            __syncthreads();
            double U[PC], beta[PC];
#pragma unroll
            for (int a = 0; a < PC; ++a) { U[a] = 0.0; beta[a] = s_beta[a]; }
            for (int i = tid; i < N; i += 256) {
                const uint64_t w64 = T[(int64_t)(i >> 6) * Vpad + v];
                double x[PC];
                x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> (i & 63)) & 1ull);
#pragma unroll
                for (int j = 0; j < Q; ++j) x[2 + j] = W[(int64_t)i * Q + j];
                double eta = 0.0, qf = 0.0;
#pragma unroll
                for (int a = 0; a < PC; ++a) eta = fma(beta[a], x[a], eta);
                const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
#pragma unroll
                for (int a = 0; a < PC; ++a) {
                    double t = 0.0;
#pragma unroll
                    for (int c = 0; c < PC; ++c) t = fma(s_Vm[a * PC + c], x[c], t);
                    qf = fma(x[a], t, qf);
                }
                const double res = y[i] - mu + wgt * qf * (0.5 - mu);                    // model.py:455-462
#pragma unroll
                for (int a = 0; a < PC; ++a) U[a] = fma(x[a], res, U[a]);
            }
            blk_sum<PC>(U, s_red, tid);
            if (tid == 0) {
                for (int a = 0; a < PC; ++a) {
                    double t = 0.0;
                    for (int c = 0; c < PC; ++c) t = fma(s_Vm[a * PC + c], U[c], t);
                    s_cand[a] = s_beta[a] + t;
                }
                halvings = 0;
            }
            for (;;) {                                                                 // step halving, model.py:465-474
                __syncthreads();
                blk_info<PC>(T, Vpad, v, N, y, W, s_cand, I, &ll, s_red, tid);
                if (tid == 0) {
                    Fcand = -(ll + 0.5 * log(slow_det<PC>(I)));
                    if (!(Fcand > Fcur + P.firth_noise * fabs(Fcur))) s_ctl = 1;
                    else {
                        bool moved = false;                                              // fixed point of the halving map: see k_firth_eval2
                        for (int a = 0; a < PC; ++a) { const double nc = s_beta[a] + 0.5 * (s_cand[a] - s_beta[a]); moved = moved || (nc != s_cand[a]); s_cand[a] = nc; }
                        if (++halvings > 1000 || !moved) { failed = true; s_ctl = 2; } else s_ctl = 0;
                    }
                }
                __syncthreads();
                if (s_ctl != 0) break;
            }
            if (tid == 0 && !failed) {
                double sn = 0.0;
                for (int a = 0; a < PC; ++a) { const double d = s_cand[a] - s_beta[a]; sn = fma(d, d, sn); s_beta[a] = s_cand[a]; }
                sn = sqrt(sn); Fcur = Fcand; i11 = I[PC + 1];
                if (iter > 0 && sn_prev < 1e-4) conv = true;                          // the PREVIOUS step, model.py:477-479
                if (conv || iter + 1 >= FIRTH_SLOW_ITERS) flags[v] |= firth_sensitive(iter + 1, sn_prev);
                sn_prev = sn;
                s_ctl = conv ? 2 : 1;
            }
            __syncthreads();
            if (s_ctl == 2) break;
        }
        if (tid == 0) {
            if (!conv) failed = true;
            uint32_t fl = flags[v];
            if (failed) {
                fl |= SH_NOTE_FIRTH_FAIL | SH_FLAG_FILTER | SH_FLAG_FIRTH_SENSITIVE;
                out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
                for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = NAN;
            } else {
                const double lrstat = -2.0 * (P.null_firth - (-Fcur));
                double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);
                out[V + v] = pval; out[2 * V + v] = s_beta[1]; out[3 * V + v] = sqrt(i11); out[4 * V + v] = s_beta[0];
                for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = s_beta[2 + j];
                if (pval > P.lrtt || !isfinite(pval) || !isfinite(s_beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
            }
            flags[v] = fl;
        }
    }
}

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

// =====================================================================================================================
// a6 fit_lineage_effect (model.py:151-199): logistic regression of the VARIANT on [1, lineages, covariates] (statsmodels
// Newton, default zero start), returns argmax_j |beta_j| / bse_j over the lineage columns, or -1 (None) on
// PerfectSeparationError / LinAlgError.  Here the whole design row is wave-uniform and the response is the per-lane bit.
// X: N x PC row-major with the intercept in column 0.
// =====================================================================================================================
template <int PC>
__global__ __launch_bounds__(256) void k_glm_lineage(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N, int NB64,
                                                    const double *__restrict__ X, int nlin, int *__restrict__ out)
{
    const XWave xw = xwave();                                        // up to four wavefronts share the 64 variants of a block
    const int64_t v = (int64_t)blockIdx.x * 64 + xw.lane;
    const bool live = v < V;
    const int64_t vr = live ? v : 0;
    double beta[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = 0.0;
    int it = 0, status = 0, best = -1;
    bool fin = false, active = live;
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
    if (live && xw.w == 0) out[v] = (status == 0) ? best : -1;
}

extern "C" hipError_t shk_glm_lineage(hipStream_t st, int PC, const uint64_t *T, int64_t Vpad, int64_t V, int N, int NB64,
                                      const double *X, int nlin, int *out)
{
    const int S = std::min(4, glm_split_waves(NB64));
    const dim3 grid((unsigned)((V + 63) / 64)), blk(64 * S);
#define LIN_CASE(p) case p: hipLaunchKernelGGL(k_glm_lineage<p>, grid, blk, glm_split_lds(S), st, T, Vpad, V, N, NB64, X, nlin, out); break;
    switch (PC) {
        LIN_CASE(2) LIN_CASE(3) LIN_CASE(4) LIN_CASE(5) LIN_CASE(6) LIN_CASE(7) LIN_CASE(8) LIN_CASE(9) LIN_CASE(10)
        LIN_CASE(11) LIN_CASE(12) LIN_CASE(13) LIN_CASE(14) LIN_CASE(15) LIN_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef LIN_CASE
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------

template <int Q>
static hipError_t launch_glm(hipStream_t st, int which, const uint64_t *T, int64_t Vpad, int64_t V, const double *y,
                             const double *W, const float *Wf, const uint64_t *y1, const uint64_t *y0, const double *yc, const double *ZtZ,
                             const double *Zty, GlmParams P, double *out, uint32_t *flags, int *flist, int *fcount,
                             int *plist, int *pcount, GlmWork wk)
{
    const dim3 grid((unsigned)((V + 63) / 64)), blk(64);
    if (which == 0) {
        if (P.chord_on && Q > 0 && P.zz) {
            // counters: ch_cnt[0 .. n32] Newton lists, ch_cnt[n32 + 1 ..] chord lists (zeroed by the caller); lists ping-pong
            const int n32 = P.chord_n32, nc = P.chord_rounds;
            int *cc = P.ch_cnt + n32 + 1;
            const dim3 g256((unsigned)(Vpad / 256)), b256(256);
            hipLaunchKernelGGL((k_glm_fast<Q, true>), g256, b256, 0, st, T, Vpad, V, y, W, Wf, y1, y0, yc, P, wk, out, flags, flist, fcount);
            int r0 = 0;
            if (P.bd_tab) {                                          // carrier sums: the first Newton step needs no pass, the finishing rounds use them too
                // int8 matrix cores (bitdot_i8.hip: up to 16 columns = GLM_MAXQ + 2); the nibble-table kernel if it ever refuses
                if (!P.bd_i8 || shk_bitdot_i8(st, T, Vpad, P.NB64, Q + 2, P.bd_i8, P.bd_scale, P.ch_bd) != hipSuccess)
                    hipLaunchKernelGGL(k_glm_bitdot<Q>, g256, b256, 0, st, T, Vpad, P);
                if (P.b1 && n32 > 1) hipLaunchKernelGGL(k_glm_first_step<Q>, g256, b256, 0, st, Vpad, P, wk, P.ch_list[0], P.ch_cnt, P.ch_list[1], P.ch_cnt + 1);
                else hipLaunchKernelGGL((k_glm_solve32<Q, true>), g256, b256, 0, st, Vpad, P, wk, P.ch_list[0], P.ch_cnt, P.ch_list[1], P.ch_cnt + 1,
                                        P.ch_list[2], cc, n32 == 1 ? 1 : 0);
                r0 = 1;
            }
            for (int r = r0; r < n32; ++r) {
                if (P.wfp && r > r0) {
                    const size_t lds = (size_t)SplitCfg<Q>::S * SplitCfg<Q>::PER_WAVE;
                    static bool attr_set = false;                    // per instantiation (one Q per process in practice)
                    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_glm_pass32_split<Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
                    hipLaunchKernelGGL(k_glm_pass32_split<Q>, grid, dim3(64 * SplitCfg<Q>::S), lds, st, T, Vpad, Wf, P, P.ch_list[r & 1], P.ch_cnt + r);
                }
                else if (P.wfp) hipLaunchKernelGGL((k_glm_pass32<Q, true>), grid, blk, 0, st, T, Vpad, y, Wf, P, P.ch_list[r & 1], P.ch_cnt + r);
                else hipLaunchKernelGGL((k_glm_pass32<Q, false>), grid, blk, 0, st, T, Vpad, y, Wf, P, P.ch_list[r & 1], P.ch_cnt + r);
                hipLaunchKernelGGL((k_glm_solve32<Q, false>), g256, b256, 0, st, Vpad, P, wk, P.ch_list[r & 1], P.ch_cnt + r, P.ch_list[(r + 1) & 1], P.ch_cnt + r + 1,
                                   P.ch_list[2], cc, r == n32 - 1 ? 1 : 0);
            }
            for (int r = 0; r < nc; ++r) {
                const int first_ll = (r == 0 && P.ll_first && P.fin_rounds) ? 1 : 0;
                if (first_ll) hipLaunchKernelGGL(k_glm_ll<Q>, g256, b256, 0, st, T, Vpad, y, P, P.ch_list[2 + (r & 1)], cc + r);
                else if (r == 0) hipLaunchKernelGGL(k_glm_score<Q>, g256, b256, 0, st, T, Vpad, y, W, P, P.ch_list[2 + (r & 1)], cc + r);
                else hipLaunchKernelGGL(k_glm_score_split<Q>, grid, dim3(64 * SCORE_SPLIT), 0, st, T, Vpad, P, P.ch_list[2 + (r & 1)], cc + r);
                hipLaunchKernelGGL(k_glm_chord<Q>, g256, b256, 0, st, Vpad, P, wk, P.ch_list[2 + (r & 1)], cc + r, P.ch_list[2 + ((r + 1) & 1)], cc + r + 1,
                                   r == nc - 1 ? 1 : 0, first_ll);
            }
            if (P.fin_rounds && P.ll_first) {                        // the variants the first chord round finished: no further likelihood pass
                if (P.wfp) hipLaunchKernelGGL(k_glm_dpass_pk<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[5], P.ch_cnt + 29);
                else hipLaunchKernelGGL(k_glm_dpass<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[5], P.ch_cnt + 29);
                hipLaunchKernelGGL((k_glm_finish<Q, true>), grid, blk, 0, st, Vpad, V, P, wk, P.ch_list[5], P.ch_cnt + 29, out, flags, flist, fcount);
            }
            if (P.fin_rounds) {
                hipLaunchKernelGGL(k_glm_ll<Q>, g256, b256, 0, st, T, Vpad, y, P, P.ch_list[4], P.ch_cnt + 30);
                if (P.wfp) hipLaunchKernelGGL(k_glm_dpass_pk<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[4], P.ch_cnt + 30);
                else hipLaunchKernelGGL(k_glm_dpass<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[4], P.ch_cnt + 30);
                hipLaunchKernelGGL((k_glm_finish<Q, false>), grid, blk, 0, st, Vpad, V, P, wk, P.ch_list[4], P.ch_cnt + 30, out, flags, flist, fcount);
            }
        } else hipLaunchKernelGGL((k_glm_fast<Q, false>), grid, blk, 0, st, T, Vpad, V, y, W, Wf, y1, y0, yc, P, wk, out, flags, flist, fcount);
    }
    else if (which == 4) {
        hipLaunchKernelGGL(k_glm_slow_blk<Q>, dim3((unsigned)std::min<int64_t>(V, 2048)), dim3(256), 0, st, T, Vpad, V, y, W, P, wk, out, flags, flist, fcount);
    }
    else if (which == 5) {
        if (Q > 0 && P.a0 && P.w0 && P.zz && P.ws) {
            hipLaunchKernelGGL((k_glm_final<Q, true>), grid, blk, 0, st, T, Vpad, V, y, W, P, wk, out, flags, flist, fcount, 1);
            hipLaunchKernelGGL((k_glm_final<Q, false>), grid, blk, 0, st, T, Vpad, V, y, W, P, wk, out, flags, flist, fcount, 2);
        } else hipLaunchKernelGGL((k_glm_final<Q, false>), grid, blk, 0, st, T, Vpad, V, y, W, P, wk, out, flags, flist, fcount, 1);
    }
    else if (which == 1) return hipErrorInvalidValue;                // (the single-kernel Firth form of round 1 is gone: firth_rounds.hip)
    else if (which == 3) hipLaunchKernelGGL(k_glm_firth_pinv<Q>, dim3(512), dim3(256), 0, st, T, Vpad, V, y, W, P, plist, pcount, out, flags);
    else if (which == 6) hipLaunchKernelGGL(k_glm_ols_pinv<Q>, dim3(512), dim3(256), 0, st, T, Vpad, V, y, W, P, plist, pcount, out, flags);
    else if (which > 6) return hipErrorInvalidValue;
    else hipLaunchKernelGGL(k_glm_ols<Q>, grid, blk, 0, st, T, Vpad, V, y, W, y1, y0, yc, ZtZ, Zty, P, out, flags, plist, pcount);
    return hipGetLastError();
}

extern "C" hipError_t shk_glm_launch(hipStream_t st, int Q, int which, const uint64_t *T, int64_t Vpad, int64_t V,
                                     const double *y, const double *W, const float *Wf, const uint64_t *y1, const uint64_t *y0,
                                     const double *yc, const double *ZtZ, const double *Zty, GlmParams P, double *out,
                                     uint32_t *flags, int *flist, int *fcount, int *plist, int *pcount, double *bw, int *state,
                                     int *slow_list, int *slow_count, int *tile_list, int *tile_count)
{
    GlmWork wk{bw, state, slow_list, slow_count, tile_list, tile_count};
#define GLM_CASE(q) case q: return launch_glm<q>(st, which, T, Vpad, V, y, W, Wf, y1, y0, yc, ZtZ, Zty, P, out, flags, flist, fcount, plist, pcount, wk);
    switch (Q) {
        GLM_CASE(0) GLM_CASE(1) GLM_CASE(2) GLM_CASE(3) GLM_CASE(4) GLM_CASE(5) GLM_CASE(6) GLM_CASE(7)
        GLM_CASE(8) GLM_CASE(9) GLM_CASE(10) GLM_CASE(11) GLM_CASE(12) GLM_CASE(13) GLM_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef GLM_CASE
}

extern "C" hipError_t shk_bitdot_build_tab(hipStream_t st, const double *vals, int NE, int N, int NB64, double *tab)
{
    hipLaunchKernelGGL(k_bitdot_build_tab, dim3((unsigned)NB64), dim3(256), 0, st, vals, NE, N, NB64, tab);
    return hipGetLastError();
}

// ---- Firth state machine launchers: which = 0 init, 1 eval, 2 step, 3 hand-off list; n = upper bound of the list length ----
template <int Q>
static hipError_t launch_firth(hipStream_t st, int which, int64_t n, const uint64_t *T, int64_t Vpad, int64_t V, const double *y,
                               const double *W, GlmParams P, FirthWork fw, const int *in_list, const int *in_count, int *next_eval,
                               int *next_eval_count, int *step_list, int *step_count, double *out, uint32_t *flags, int *plist, int *pcount)
{
    if (n <= 0) return hipSuccess;
    const int S = glm_split_waves(P.NB64);
    const dim3 grid((unsigned)((n + 63) / 64)), blk(64), blks(64 * S);
    if (which == 0) hipLaunchKernelGGL(k_firth_init<Q>, grid, blk, 0, st, in_list, in_count, P, fw, next_eval, next_eval_count);
    else if (which == 1 && P.rec_o && (P.firth_lean & 1)) hipLaunchKernelGGL((k_firth_eval<Q, true>), grid, dim3(64 * std::min(S, FIRTH_EVAL_THREADS / 64)), glm_split_lds(std::min(S, FIRTH_EVAL_THREADS / 64)), st, T, Vpad, V, y, W, P, fw, in_list, in_count, next_eval,
                                            next_eval_count, step_list, step_count, out, flags, plist, pcount);
    else if (which == 1) hipLaunchKernelGGL((k_firth_eval<Q, false>), grid, blks, glm_split_lds(S), st, T, Vpad, V, y, W, P, fw, in_list, in_count, next_eval,
                                            next_eval_count, step_list, step_count, out, flags, plist, pcount);
    else if (which == 2 && P.rec_o && (P.firth_lean & 2)) hipLaunchKernelGGL((k_firth_step<Q, true>), grid, blks, glm_split_lds(S), st, T, Vpad, y, W, P, fw, in_list, in_count, next_eval, next_eval_count);
    else if (which == 2) hipLaunchKernelGGL((k_firth_step<Q, false>), grid, blks, glm_split_lds(S), st, T, Vpad, y, W, P, fw, in_list, in_count, next_eval, next_eval_count);
    else hipLaunchKernelGGL(k_firth_blk<Q>, dim3((unsigned)std::min<int64_t>(n, 2048)), dim3(256), 0, st, T, Vpad, V, y, W, P, fw, out, flags, plist, pcount);
    return hipGetLastError();
}

extern "C" int shk_firth_state_doubles(int Q) { const int PC = Q + 2; return 2 * PC + PC * (PC + 1) / 2 + 3; }

extern "C" hipError_t shk_firth_launch(hipStream_t st, int Q, int which, int64_t n, const uint64_t *T, int64_t Vpad, int64_t V,
                                       const double *y, const double *W, GlmParams P, double *fst, int *fiter, int *fhalv, int *fvar,
                                       int64_t fcap, const int *in_list, const int *in_count, int *next_eval, int *next_eval_count,
                                       int *step_list, int *step_count, double *out, uint32_t *flags, int *plist, int *pcount,
                                       int *blk_list, int *blk_count)
{
    FirthWork fw{fst, fiter, fhalv, fvar, fcap, blk_list, blk_count, nullptr, nullptr};
#define FIRTH_CASE(q) case q: return launch_firth<q>(st, which, n, T, Vpad, V, y, W, P, fw, in_list, in_count, next_eval, next_eval_count, step_list, step_count, out, flags, plist, pcount);
    switch (Q) {
        FIRTH_CASE(0) FIRTH_CASE(1) FIRTH_CASE(2) FIRTH_CASE(3) FIRTH_CASE(4) FIRTH_CASE(5) FIRTH_CASE(6) FIRTH_CASE(7)
        FIRTH_CASE(8) FIRTH_CASE(9) FIRTH_CASE(10) FIRTH_CASE(11) FIRTH_CASE(12) FIRTH_CASE(13) FIRTH_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef FIRTH_CASE
}

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
