// glm_passes.h -- the sample passes of the fixed-effects kernels as device functions (one variant per lane; the sample's phenotype and covariate
// row are wave-uniform): info_pass / info_pass_bin (X^T W X, score, log-likelihood at beta), fast_pass and its matrix-pipe forms (the
// single-precision Newton phase), the final pass as a correction to the null model.  Shared by glm_kernels.hip (logistic: the product path),
// glm_firth_v1.hip (round 2's Firth kernels) and glm_ols.hip.  Until round 5 these were the first 450 lines of glm_kernels.hip.
#pragma once
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

