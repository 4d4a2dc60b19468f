// glm_wide.hip -- fixed-effects regression for WIDE designs (15 <= q <= 32 covariate columns), one variant per lane.
//
// glm_kernels.hip keeps the p x p matrices of a variant in registers, fully unrolled, which stops at q = 14 (p = 16: 136 fp64
// accumulators).  The reference has no such limit (pyseer/model.py:274-297 concatenates any number of MDS components and
// covariate columns; a categorical covariate alone contributes one column per level).  These kernels restate the same
// algorithms with a run-time p <= 34 on per-lane scratch arrays and rolled loops: statsmodels' Newton (start vector, ridge,
// 35 iterations, separation callback, final un-ridged inverse), fit_firth with numpy's pinv, OLS through pinv.  Same notes,
// same filters, same outputs; throughput is that of scratch memory, not of registers (they exist so that a wide design runs
// at all: the reference manages ~100 variants/s/core on them).
#include <algorithm>
#include "glm_common.h"

#define WIDE_PM 34

// ---- dense helpers, row-major pc x pc inside arrays of capacity WIDE_PM^2 ------------------------------------------------
// LU with partial pivoting in place; returns the determinant (0 when a pivot is exactly zero: numpy's LinAlgError / det 0)
__device__ __noinline__ double w_lu(double *A, int *piv, int pc)
{
    double det = 1.0;
#pragma unroll 1
    for (int c = 0; c < pc; ++c) {
        int p = c; double best = fabs(A[c * pc + c]);
#pragma unroll 1
        for (int r = c + 1; r < pc; ++r) { const double t = fabs(A[r * pc + c]); if (t > best) { best = t; p = r; } }
        piv[c] = p;
        if (p != c) {
#pragma unroll 1
            for (int j = 0; j < pc; ++j) { const double t = A[c * pc + j]; A[c * pc + j] = A[p * pc + j]; A[p * pc + j] = t; }
            det = -det;
        }
        const double d = A[c * pc + c];
        det *= d;
        if (d == 0.0) return 0.0;
#pragma unroll 1
        for (int r = c + 1; r < pc; ++r) {
            const double f = A[r * pc + c] / d;
            A[r * pc + c] = f;
#pragma unroll 1
            for (int j = c + 1; j < pc; ++j) A[r * pc + j] = fma(-f, A[c * pc + j], A[r * pc + j]);
        }
    }
    return det;
}

__device__ __noinline__ void w_lu_solve(const double *LU, const int *piv, int pc, double *b)
{
    // w_lu swaps whole rows (multipliers of earlier columns included), so the row interchanges are applied to b first
#pragma unroll 1
    for (int c = 0; c < pc; ++c) { const int p = piv[c]; if (p != c) { const double t = b[c]; b[c] = b[p]; b[p] = t; } }
#pragma unroll 1
    for (int c = 0; c < pc; ++c) {
#pragma unroll 1
        for (int r = c + 1; r < pc; ++r) b[r] = fma(-LU[r * pc + c], b[c], b[r]);
    }
#pragma unroll 1
    for (int c = pc - 1; c >= 0; --c) {
#pragma unroll 1
        for (int j = c + 1; j < pc; ++j) b[c] = fma(-LU[c * pc + j], b[j], b[c]);
        b[c] /= LU[c * pc + c];
    }
}

// numpy.linalg.pinv of a symmetric matrix: cyclic Jacobi eigen-decomposition, eigenvalues <= rcond * max dropped
__device__ __noinline__ void w_pinv(const double *Ain, double *Pm, int pc, double rcond, int *rank)
{
    double A[WIDE_PM * WIDE_PM], Vv[WIDE_PM * WIDE_PM];
#pragma unroll 1
    for (int a = 0; a < pc * pc; ++a) { A[a] = Ain[a]; Vv[a] = 0.0; }
#pragma unroll 1
    for (int a = 0; a < pc; ++a) Vv[a * pc + a] = 1.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll 1
        for (int i = 0; i < pc; ++i) { dg = fma(A[i * pc + i], A[i * pc + i], dg);
#pragma unroll 1
            for (int j = i + 1; j < pc; ++j) off = fma(A[i * pc + j], A[i * pc + j], off); }
        if (off <= 1e-34 * (dg + off) || off == 0.0) break;
#pragma unroll 1
        for (int p = 0; p < pc; ++p)
#pragma unroll 1
            for (int q = p + 1; q < pc; ++q) {
                const double apq = A[p * pc + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * pc + q] - A[p * pc + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll 1
                for (int k = 0; k < pc; ++k) { const double akp = A[k * pc + p], akq = A[k * pc + q]; A[k * pc + p] = c * akp - s * akq; A[k * pc + q] = s * akp + c * akq; }
#pragma unroll 1
                for (int k = 0; k < pc; ++k) { const double apk = A[p * pc + k], aqk = A[q * pc + k]; A[p * pc + k] = c * apk - s * aqk; A[q * pc + k] = s * apk + c * aqk; }
#pragma unroll 1
                for (int k = 0; k < pc; ++k) { const double vkp = Vv[k * pc + p], vkq = Vv[k * pc + q]; Vv[k * pc + p] = c * vkp - s * vkq; Vv[k * pc + q] = s * vkp + c * vkq; }
            }
    }
    double smax = 0.0;
#pragma unroll 1
    for (int i = 0; i < pc; ++i) smax = fmax(smax, fabs(A[i * pc + i]));
    if (rank) { int r = 0;
#pragma unroll 1
        for (int k = 0; k < pc; ++k) r += (fabs(A[k * pc + k]) > rcond * smax) ? 1 : 0;
        *rank = r; }
#pragma unroll 1
    for (int i = 0; i < pc; ++i)
#pragma unroll 1
        for (int j = 0; j < pc; ++j) {
            double s = 0.0;
#pragma unroll 1
            for (int k = 0; k < pc; ++k) { const double w = A[k * pc + k]; if (fabs(w) > rcond * smax) s = fma(Vv[i * pc + k] / w, Vv[j * pc + k], s); }
            Pm[i * pc + j] = s;
        }
}

// one pass over the samples at beta: I = X^T W X (full, row-major), optional score g, log-likelihood, max |y - mu|
__device__ __noinline__ void w_info(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64, int q,
                                    const double *__restrict__ y, const double *__restrict__ W, const double *beta, double *I,
                                    double *g, double *ll_out, double *maxdev_out)
{
    const int pc = q + 2;
#pragma unroll 1
    for (int a = 0; a < pc * pc; ++a) I[a] = 0.0;
    if (g) {
#pragma unroll 1
        for (int a = 0; a < pc; ++a) g[a] = 0.0; }
    double ll = 0.0, maxdev = 0.0;
#pragma unroll 1
    for (int sb = 0; sb < NB64; ++sb) {
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            const int i = sb * 64 + b;
            double x[WIDE_PM];
            x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> b) & 1ull);
#pragma unroll 1
            for (int j = 0; j < q; ++j) x[2 + j] = W[(int64_t)i * q + j];
            double eta = 0.0;
#pragma unroll 1
            for (int a = 0; a < pc; ++a) eta = fma(beta[a], x[a], eta);
            const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu), yi = y[i], r = yi - mu;
            maxdev = fmax(maxdev, fabs(r));
            const double lm = log(mu);
            ll += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
#pragma unroll 1
            for (int a = 0; a < pc; ++a) {
                if (g) g[a] = fma(r, x[a], g[a]);
                const double wa = wgt * x[a];
#pragma unroll 1
                for (int c = 0; c <= a; ++c) I[a * pc + c] = fma(wa, x[c], I[a * pc + c]);
            }
        }
    }
#pragma unroll 1
    for (int a = 0; a < pc; ++a)
#pragma unroll 1
        for (int c = a + 1; c < pc; ++c) I[a * pc + c] = I[c * pc + a];
    *ll_out = ll; *maxdev_out = maxdev;
}

// ---- kernel 1: prefilter + routing (+ logistic Newton and its decisions for a binary phenotype) -------------------------
__global__ __launch_bounds__(64) void k_glm_wide(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int q,
                                                 const double *__restrict__ y, const double *__restrict__ W,
                                                 const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                 const double *__restrict__ yc, GlmParams P, double *__restrict__ out,
                                                 uint32_t *__restrict__ flags, int *__restrict__ firth_list,
                                                 int *__restrict__ firth_count, int *__restrict__ ols_list, int *__restrict__ ols_count,
                                                 int *__restrict__ newton_list, int *__restrict__ newton_count)
{
    const int64_t v = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const int pc = q + 2, N = P.N, NB64 = P.NB64;
    const double nobs = (double)N;
    uint32_t fl = 0;
    bool want_fit = true, to_firth = false, bad = false, cell1 = false;
    int m = 0;
    double prep = glm_prefilter(T, Vpad, v, NB64, N, y1, y0, yc, P, &bad, &m, &cell1);
    if (P.af_on) {
        const double af = (double)m / (double)N;
        if (!(P.min_af <= af && af <= P.max_af)) { fl = SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER; want_fit = false; prep = NAN; }
    }
    if (want_fit) {
        if (bad) fl |= SH_NOTE_BAD_CHISQ;
        if (prep > P.pret || !isfinite(prep)) { fl |= SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER; want_fit = false; }   // model.py:266 (>)
        else if (cell1) fl |= SH_FLAG_FIRTH_SENSITIVE;
    }
    out[v] = prep; out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
#pragma unroll 1
    for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = NAN;
    if (P.continuous) {
        flags[v] = fl;
        if (want_fit) { const int s = atomicAdd(ols_count, 1); ols_list[s] = (int)v; }
        return;
    }
    if (want_fit && (bad || P.force_firth)) { to_firth = true; want_fit = false; }
    if (want_fit && newton_list) {                                     // fitted by k_glm_wide_newton_blk, one workgroup per variant
        flags[v] = fl;
        newton_list[atomicAdd(newton_count, 1)] = (int)v;
        return;
    }
    if (want_fit) {
        double beta[WIDE_PM], g[WIDE_PM], I[WIDE_PM * WIDE_PM];
        int piv[WIDE_PM];
#pragma unroll 1
        for (int a = 0; a < pc; ++a) beta[a] = 0.0;
        beta[0] = P.ymean_logit;                                                       // model.py:323-324
        int status = 0, it = 0;
        double ll, maxdev;
#pragma unroll 1
        for (;;) {
            w_info(T, Vpad, v, N, NB64, q, y, W, beta, I, g, &ll, &maxdev);
            if (it > 0 && maxdev <= 1e-8) { status = 1; break; }                       // _check_perfect_pred
            // newparams = oldparams - inv(H/n + 1e-10 I) . score/n   with H = -X^T W X   (optimizer.py:415-423)
#pragma unroll 1
            for (int a = 0; a < pc * pc; ++a) I[a] = I[a] / nobs;
#pragma unroll 1
            for (int a = 0; a < pc; ++a) { I[a * pc + a] -= 1e-10; g[a] = g[a] / nobs; }
            if (w_lu(I, piv, pc) == 0.0) { status = 2; break; }
            w_lu_solve(I, piv, pc, g);
            bool moving = false;
#pragma unroll 1
            for (int a = 0; a < pc; ++a) { beta[a] += g[a]; moving = moving || (fabs(g[a]) > 1e-8); }
            ++it;
            if (!moving || it >= 35) break;
        }
        double llf = NAN, bse1 = NAN;
        if (status == 0) {                                                             // results at the final beta
            w_info(T, Vpad, v, N, NB64, q, y, W, beta, I, nullptr, &ll, &maxdev);
            if (maxdev <= 1e-8) status = 1;                                            // callback after the last update
            else {
                llf = ll;
                // Hinv = inv(-Hessian/nobs)/nobs, no ridge (SM:base/model.py:533-534); only bse[1] is used (model.py:332)
                double amax = 0.0;
#pragma unroll 1
                for (int a = 0; a < pc * pc; ++a) { I[a] = I[a] / nobs; amax = fmax(amax, fabs(I[a])); }
                const double det = w_lu(I, piv, pc);
                bool tiny = det == 0.0;
#pragma unroll 1
                for (int a = 0; a < pc && !tiny; ++a) tiny = fabs(I[a * pc + a]) <= 4.0e-16 * amax;
                if (tiny) status = 2;
                else {
#pragma unroll 1
                    for (int a = 0; a < pc; ++a) g[a] = (a == 1) ? 1.0 : 0.0;
                    w_lu_solve(I, piv, pc, g);
                    bse1 = sqrt(g[1] / nobs);
                }
            }
        }
        if (status == 1) { fl |= SH_NOTE_PERFECT_SEP; to_firth = true; }               // model.py:345-352
        else if (status == 2) { fl |= SH_NOTE_MATRIX_INV; to_firth = true; }
        else if (bse1 > 3.0) { fl |= SH_NOTE_HIGH_BSE; to_firth = true; }              // model.py:332-334
        else {
            const double lrstat = -2.0 * (P.null_llf - llf);
            double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);           // model.py:336-339
            out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = bse1; out[4 * V + v] = beta[0];
#pragma unroll 1
            for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = beta[2 + j];
            if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;   // model.py:384
        }
    }
    flags[v] = fl;
    if (to_firth) { const int s = atomicAdd(firth_count, 1); firth_list[s] = (int)v; }
}

// ---- kernel 1b: the Newton fit of kernel 1, one workgroup per variant --------------------------------------------------------------
// A lane cannot hold a 34 x 34 information matrix in registers, and one lane updating it in scratch memory for every sample ran at
// 1.5-7 k variants/s.  Here 256 threads share one variant: samples are taken 128 at a time -- thread t evaluates sample t of the chunk
// (eta, mu, weight, residual) and stores its design row and weight in LDS -- and the matrix is accumulated in 3 x 3 register tiles:
// thread (group g, tile (ta, tc)) adds the samples g, g+G, ... of the chunk to its tile, G = 256 / #tiles groups working on disjoint
// samples.  Partial tiles are summed in a fixed order, thread 0 does the p x p algebra with the same decisions as kernel 1.
#define WB_CH 128
#define WB_XS 37
// pivoted LU of the pc x pc matrix in LDS by the whole workgroup: the same operations on every element as w_lu (one lane), four barriers
// per pivot instead of ~pc^3/3 dependent LDS round trips on thread 0 (0.5 ms per factorisation at pc = 22).  Returns the determinant
// (0 when a pivot is exactly zero) in every thread.  ctl: 2 shared ints + 1 shared double of scratch.
__device__ __forceinline__ double wb_lu(double *A, int *piv, int pc, int tid, int *s_i, double *s_d)
{
    if (tid == 0) s_d[0] = 1.0;
    for (int c = 0; c < pc; ++c) {
        __syncthreads();
        if (tid == 0) {
            int p = c; double best = fabs(A[c * pc + c]);
            for (int r = c + 1; r < pc; ++r) { const double t = fabs(A[r * pc + c]); if (t > best) { best = t; p = r; } }
            piv[c] = p; s_i[0] = p;
        }
        __syncthreads();
        const int p = s_i[0];
        if (p != c && tid < pc) { const double t = A[c * pc + tid]; A[c * pc + tid] = A[p * pc + tid]; A[p * pc + tid] = t; }
        __syncthreads();
        const double d = A[c * pc + c];
        if (tid == 0) s_d[0] = (p != c ? -s_d[0] : s_d[0]) * d;
        if (d == 0.0) { __syncthreads(); return 0.0; }
        if (tid > c && tid < pc) A[tid * pc + c] = A[tid * pc + c] / d;
        __syncthreads();
        const int w = pc - c - 1;
        for (int e = tid; e < w * w; e += 256) {
            const int r = c + 1 + e / w, j = c + 1 + e % w;
            A[r * pc + j] = fma(-A[r * pc + c], A[c * pc + j], A[r * pc + j]);
        }
    }
    __syncthreads();
    return s_d[0];
}

// b <- A^-1 b through that factorisation (b in LDS), column-oriented substitutions by the workgroup
__device__ __forceinline__ void wb_solve(const double *LU, const int *piv, int pc, double *b, int tid)
{
    __syncthreads();
    if (tid == 0) for (int c = 0; c < pc; ++c) { const int p = piv[c]; if (p != c) { const double t = b[c]; b[c] = b[p]; b[p] = t; } }
    for (int c = 0; c < pc; ++c) {
        __syncthreads();
        if (tid > c && tid < pc) b[tid] = fma(-LU[tid * pc + c], b[c], b[tid]);
    }
    for (int c = pc - 1; c >= 0; --c) {
        __syncthreads();
        if (tid == 0) b[c] = b[c] / LU[c * pc + c];
        __syncthreads();
        if (tid < c) b[tid] = fma(-LU[tid * pc + c], b[c], b[tid]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_glm_wide_newton_blk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int q,
                                                             const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                             const int *__restrict__ nlist, const int *__restrict__ ncount,
                                                             double *__restrict__ out, uint32_t *__restrict__ flags,
                                                             int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    __shared__ double xs[WB_CH * WB_XS], wch[WB_CH], rch[WB_CH], s_beta[WIDE_PM + 2], s_g[WIDE_PM + 2], s_H[WIDE_PM * WIDE_PM];
    __shared__ double s_red[256 * 9], s_sc[8];
    __shared__ int s_ctl, s_piv[WIDE_PM], s_i[2];                  // s_ctl: 0 = iterate, 1 = evaluate at the final beta, 2 = done
    const int pc = q + 2, N = P.N, tid = threadIdx.x;
    const int NT = (pc + 2) / 3, NTT = NT * (NT + 1) / 2, G = 256 / NTT;
    const int grp = tid / NTT, tile = tid - grp * NTT;
    const bool tiler = grp < G;
    int ta = 0; { int t = tile; while (t >= ta + 1) { t -= ta + 1; ++ta; } }
    int tc = tile - ta * (ta + 1) / 2;
    const double nobs = (double)N;
    const int cnt = *ncount;
    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int64_t v = nlist[slot];
        int it = 0, status = 0;                                    // kept identical in every thread
        __syncthreads();
        if (tid < WIDE_PM + 2) s_beta[tid] = (tid == 0) ? P.ymean_logit : 0.0;         // model.py:323-324
        if (tid == 0) s_ctl = 0;
        __syncthreads();
        for (;;) {
            // ---- one pass over the samples at s_beta: X^T W X, score, log-likelihood, max |y - mu|
            double acc[9], gacc = 0.0, sc[2] = {0.0, 0.0};                              // sc: log-likelihood, max deviation
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            for (int c0 = 0; c0 < N; c0 += WB_CH) {
                if (tid < WB_CH) {
                    const int i = c0 + tid;
                    double *row = xs + tid * WB_XS;
                    if (i < N) {
                        const double xb = (double)(unsigned)((T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
                        row[0] = 1.0; row[1] = xb;
                        double eta = fma(s_beta[1], xb, s_beta[0]);
#pragma unroll 1
                        for (int j = 0; j < q; ++j) { const double z = W[(int64_t)i * q + j]; row[2 + j] = z; eta = fma(s_beta[2 + j], z, eta); }
#pragma unroll 1
                        for (int a = pc; a < 3 * NT; ++a) row[a] = 0.0;
                        const double mu = logit_cdf(eta), yi = y[i], r = yi - mu;
                        wch[tid] = mu * (1.0 - mu); rch[tid] = r;
                        sc[1] = fmax(sc[1], fabs(r));
                        const double lm = log(mu);
                        sc[0] += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
                    } else {
#pragma unroll 1
                        for (int a = 0; a < 3 * NT; ++a) row[a] = 0.0;
                        wch[tid] = 0.0; rch[tid] = 0.0;
                    }
                }
                __syncthreads();
                if (tiler) {
                    for (int ii = grp; ii < WB_CH; ii += G) {
                        const double *row = xs + ii * WB_XS;
                        const double w = wch[ii];
                        const double a0 = w * row[3 * ta], a1 = w * row[3 * ta + 1], a2 = w * row[3 * ta + 2];
                        const double c0v = row[3 * tc], c1v = row[3 * tc + 1], c2v = row[3 * tc + 2];
                        acc[0] = fma(a0, c0v, acc[0]); acc[1] = fma(a0, c1v, acc[1]); acc[2] = fma(a0, c2v, acc[2]);
                        acc[3] = fma(a1, c0v, acc[3]); acc[4] = fma(a1, c1v, acc[4]); acc[5] = fma(a1, c2v, acc[5]);
                        acc[6] = fma(a2, c0v, acc[6]); acc[7] = fma(a2, c1v, acc[7]); acc[8] = fma(a2, c2v, acc[8]);
                    }
                }
                if (tid < pc) {
                    for (int ii = 0; ii < WB_CH; ++ii) gacc = fma(rch[ii], xs[ii * WB_XS + tid], gacc);
                }
                __syncthreads();
            }
            // ---- combine: tiles over the groups (fixed order), the two scalars over the 128 evaluating threads
#pragma unroll
            for (int k = 0; k < 9; ++k) s_red[tid * 9 + k] = acc[k];
            if (tid < pc) s_g[tid] = gacc;
            __syncthreads();
            for (int e = tid; e < NTT * 9; e += 256) {
                const int tl = e / 9, k = e - tl * 9;
                double t = 0.0;
                for (int gq = 0; gq < G; ++gq) t += s_red[(gq * NTT + tl) * 9 + k];
                int ra = 0; { int u = tl; while (u >= ra + 1) { u -= ra + 1; ++ra; } }
                const int rc = tl - ra * (ra + 1) / 2;
                const int a = 3 * ra + k / 3, c = 3 * rc + k % 3;
                if (a < pc && c <= a) { s_H[a * pc + c] = t; s_H[c * pc + a] = t; }     // diagonal tiles: the lower entry defines both
            }
            __syncthreads();
            if (tid < WB_CH) { s_red[tid] = sc[0]; s_red[WB_CH + tid] = sc[1]; }
            __syncthreads();
            if (tid == 0) {
                double ll = 0.0, maxdev = 0.0;
                for (int k = 0; k < WB_CH; ++k) { ll += s_red[k]; maxdev = fmax(maxdev, s_red[WB_CH + k]); }
                s_sc[2] = ll; s_sc[3] = maxdev;
            }
            __syncthreads();
            const double ll = s_sc[2], maxdev = s_sc[3];
            const int mode = s_ctl;                                                    // block-uniform from here on
            __syncthreads();
            if (mode == 0) {                                                           // a Newton step (the loop of kernel 1)
                if (it > 0 && maxdev <= 1e-8) { status = 1; if (tid == 0) s_ctl = 2; } // _check_perfect_pred
                else {
                    for (int e = tid; e < pc * pc; e += 256) s_H[e] = s_H[e] / nobs;
                    __syncthreads();
                    if (tid < pc) { s_H[tid * pc + tid] -= 1e-10; s_g[tid] = s_g[tid] / nobs; }
                    if (wb_lu(s_H, s_piv, pc, tid, s_i, s_sc + 4) == 0.0) { status = 2; if (tid == 0) s_ctl = 2; }
                    else {
                        wb_solve(s_H, s_piv, pc, s_g, tid);
                        bool moving = false;
                        for (int a = 0; a < pc; ++a) moving = moving || (fabs(s_g[a]) > 1e-8);
                        __syncthreads();
                        if (tid < pc) s_beta[tid] += s_g[tid];
                        ++it;
                        if (tid == 0 && (!moving || it >= 35)) s_ctl = 1;
                    }
                }
            } else {                                                                   // results at the final beta
                double bse1 = NAN, llf = NAN;
                if (maxdev <= 1e-8) status = 1;                                        // callback after the last update
                else {
                    llf = ll;
                    for (int e = tid; e < pc * pc; e += 256) s_H[e] = s_H[e] / nobs;
                    __syncthreads();
                    double amax = 0.0;
                    for (int a = 0; a < pc * pc; ++a) amax = fmax(amax, fabs(s_H[a]));
                    const double det = wb_lu(s_H, s_piv, pc, tid, s_i, s_sc + 4);
                    bool tiny = det == 0.0;
                    for (int a = 0; a < pc && !tiny; ++a) tiny = fabs(s_H[a * pc + a]) <= 4.0e-16 * amax;
                    if (tiny) status = 2;
                    else {
                        if (tid < pc) s_g[tid] = (tid == 1) ? 1.0 : 0.0;
                        wb_solve(s_H, s_piv, pc, s_g, tid);
                        bse1 = sqrt(s_g[1] / nobs);
                    }
                }
                __syncthreads();
                if (tid == 0) { s_sc[0] = bse1; s_sc[1] = llf; s_ctl = 2; }
            }
            __syncthreads();
            if (s_ctl == 2) break;
        }
        if (tid == 0) {
            uint32_t fl = flags[v];
            bool to_firth = false;
            const double bse1 = s_sc[0], llf = s_sc[1];
            if (status == 1) { fl |= SH_NOTE_PERFECT_SEP; to_firth = true; }           // model.py:345-352
            else if (status == 2) { fl |= SH_NOTE_MATRIX_INV; to_firth = true; }
            else if (bse1 > 3.0) { fl |= SH_NOTE_HIGH_BSE; to_firth = true; }          // model.py:332-334
            else {
                const double lrstat = -2.0 * (P.null_llf - llf);
                double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);       // model.py:336-339
                out[V + v] = pval; out[2 * V + v] = s_beta[1]; out[3 * V + v] = bse1; out[4 * V + v] = s_beta[0];
                for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = s_beta[2 + j];
                if (pval > P.lrtt || !isfinite(pval) || !isfinite(s_beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;   // model.py:384
            }
            flags[v] = fl;
            if (to_firth) { const int s2 = atomicAdd(firth_count, 1); firth_list[s2] = (int)v; }
        }
    }
}

// ---- kernel 2b: fit_firth by one workgroup per variant (same staging and tiles as kernel 1b) ----------------------------------------------
// The information matrix is inverted through its LU factors (thread a solves for column a); a variant whose matrix is singular to
// 1e-12 needs numpy's pinv semantics and is handed to kernel 2 (one lane, Jacobi pinv) through `fallback`.
__global__ __launch_bounds__(256) void k_glm_wide_firth_blk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int q,
                                                            const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                            const int *__restrict__ list, const int *__restrict__ count,
                                                            double *__restrict__ out, uint32_t *__restrict__ flags,
                                                            int *__restrict__ fallback, int *__restrict__ fallback_count)
{
    __shared__ double xs[WB_CH * WB_XS], wch[WB_CH], rch[WB_CH], s_beta[WIDE_PM + 2], s_cand[WIDE_PM + 2], s_U[WIDE_PM + 2];
    __shared__ double s_H[WIDE_PM * WIDE_PM], s_L[WIDE_PM * WIDE_PM], s_V[WIDE_PM * WIDE_PM], s_red[256 * 9], s_sc[8];
    __shared__ int s_piv[WIDE_PM], s_i[2];
    const int pc = q + 2, N = P.N, tid = threadIdx.x;
    const int NT = (pc + 2) / 3, NTT = NT * (NT + 1) / 2, G = 256 / NTT;
    const int grp = tid / NTT, tile = tid - grp * NTT;
    const bool tiler = grp < G;
    int ta = 0; { int t = tile; while (t >= ta + 1) { t -= ta + 1; ++ta; } }
    const int tc = tile - ta * (ta + 1) / 2;
    const int cnt = *count;

    // I(b) -> s_H (full), log-likelihood -> returned (block-uniform); b in LDS
    auto info_at = [&](const double *b, int64_t v) -> double {
        double acc[9], ll = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0;
        for (int c0 = 0; c0 < N; c0 += WB_CH) {
            __syncthreads();
            if (tid < WB_CH) {
                const int i = c0 + tid;
                double *row = xs + tid * WB_XS;
                if (i < N) {
                    const double xb = (double)(unsigned)((T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
                    row[0] = 1.0; row[1] = xb;
                    double eta = fma(b[1], xb, b[0]);
#pragma unroll 1
                    for (int j = 0; j < q; ++j) { const double z = W[(int64_t)i * q + j]; row[2 + j] = z; eta = fma(b[2 + j], z, eta); }
#pragma unroll 1
                    for (int a = pc; a < 3 * NT; ++a) row[a] = 0.0;
                    const double mu = logit_cdf(eta), yi = y[i];
                    wch[tid] = mu * (1.0 - mu);
                    const double lm = log(mu);
                    ll += (yi == 1.0) ? lm : ((yi == 0.0) ? ll_y0(lm, eta) : log(logit_cdf((2.0 * yi - 1.0) * eta)));
                } else {
#pragma unroll 1
                    for (int a = 0; a < 3 * NT; ++a) row[a] = 0.0;
                    wch[tid] = 0.0;
                }
            }
            __syncthreads();
            if (tiler) {
                for (int ii = grp; ii < WB_CH; ii += G) {
                    const double *row = xs + ii * WB_XS;
                    const double w = wch[ii];
                    const double a0 = w * row[3 * ta], a1 = w * row[3 * ta + 1], a2 = w * row[3 * ta + 2];
                    const double c0v = row[3 * tc], c1v = row[3 * tc + 1], c2v = row[3 * tc + 2];
                    acc[0] = fma(a0, c0v, acc[0]); acc[1] = fma(a0, c1v, acc[1]); acc[2] = fma(a0, c2v, acc[2]);
                    acc[3] = fma(a1, c0v, acc[3]); acc[4] = fma(a1, c1v, acc[4]); acc[5] = fma(a1, c2v, acc[5]);
                    acc[6] = fma(a2, c0v, acc[6]); acc[7] = fma(a2, c1v, acc[7]); acc[8] = fma(a2, c2v, acc[8]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) s_red[tid * 9 + k] = acc[k];
        __syncthreads();
        for (int e = tid; e < NTT * 9; e += 256) {
            const int tl = e / 9, k = e - tl * 9;
            double t = 0.0;
            for (int gq = 0; gq < G; ++gq) t += s_red[(gq * NTT + tl) * 9 + k];
            int ra = 0; { int u = tl; while (u >= ra + 1) { u -= ra + 1; ++ra; } }
            const int rc = tl - ra * (ra + 1) / 2;
            const int a = 3 * ra + k / 3, c = 3 * rc + k % 3;
            if (a < pc && c <= a) { s_H[a * pc + c] = t; s_H[c * pc + a] = t; }
        }
        __syncthreads();
        if (tid < WB_CH) s_red[tid] = ll;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int k = 0; k < WB_CH; ++k) t += s_red[k]; s_sc[2] = t; }
        __syncthreads();
        return s_sc[2];
    };
    // log det of s_H through a factorisation of a copy (s_L, s_piv stay valid for the inverse); -inf / nan as numpy for det <= 0
    auto logdet = [&]() -> double {
        for (int e = tid; e < pc * pc; e += 256) s_L[e] = s_H[e];
        __syncthreads();
        return log(wb_lu(s_L, s_piv, pc, tid, s_i, s_sc + 4));
    };

    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int64_t v = list[slot];
        __syncthreads();
        if (tid < WIDE_PM + 2) s_beta[tid] = (tid == 0) ? P.ymean_logit : 0.0;
        __syncthreads();
        double ll = info_at(s_beta, v);
        double Fcur = -(ll + 0.5 * logdet());                                          // firth_likelihood, model.py:410-411
        double i11 = s_H[pc + 1], sn_prev = INFINITY;
        bool failed = false, conv = false, handed = false;
        for (int iter = 0; iter < 1000 && !failed && !conv && !handed; ++iter) {
            // ---- V = I(beta)^-1 from the factors left in s_L by the last logdet (they are those of I(beta))
            double amax = 0.0, pmin = INFINITY;
            for (int a = 0; a < pc; ++a) { amax = fmax(amax, fabs(s_H[a * pc + a])); pmin = fmin(pmin, fabs(s_L[a * pc + a])); }
            if (!(pmin > 1e-12 * amax)) { handed = true; break; }                      // (near-)singular: numpy's pinv semantics, kernel 2
            __syncthreads();
            if (tid < pc) {                                                            // column tid of the inverse
                double col[WIDE_PM];
                for (int a = 0; a < pc; ++a) col[a] = (a == tid) ? 1.0 : 0.0;
                for (int c = 0; c < pc; ++c) { const int p = s_piv[c]; if (p != c) { const double t = col[c]; col[c] = col[p]; col[p] = t; } }
                for (int c = 0; c < pc; ++c) for (int r = c + 1; r < pc; ++r) col[r] = fma(-s_L[r * pc + c], col[c], col[r]);
                for (int c = pc - 1; c >= 0; --c) {
                    for (int j = c + 1; j < pc; ++j) col[c] = fma(-s_L[c * pc + j], col[j], col[c]);
                    col[c] = col[c] / s_L[c * pc + c];
                }
                for (int a = 0; a < pc; ++a) s_V[a * pc + tid] = col[a];
            }
            __syncthreads();
            // ---- penalised score at beta: U* = X^T (y - mu + h (1/2 - mu)), h = w x^T V x   (model.py:455-463)
            double uacc = 0.0;
            for (int c0 = 0; c0 < N; c0 += WB_CH) {
                __syncthreads();
                if (tid < WB_CH) {
                    const int i = c0 + tid;
                    double *row = xs + tid * WB_XS;
                    double res = 0.0;
                    if (i < N) {
                        const double xb = (double)(unsigned)((T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
                        row[0] = 1.0; row[1] = xb;
                        double eta = fma(s_beta[1], xb, s_beta[0]);
#pragma unroll 1
                        for (int j = 0; j < q; ++j) { const double z = W[(int64_t)i * q + j]; row[2 + j] = z; eta = fma(s_beta[2 + j], z, eta); }
                        const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
                        double qf = 0.0;
#pragma unroll 1
                        for (int a = 0; a < pc; ++a) {
                            double t = 0.0;
#pragma unroll 1
                            for (int c = 0; c < pc; ++c) t = fma(s_V[a * pc + c], row[c], t);
                            qf = fma(row[a], t, qf);
                        }
                        res = y[i] - mu + wgt * qf * (0.5 - mu);
                    } else {
#pragma unroll 1
                        for (int a = 0; a < pc; ++a) row[a] = 0.0;
                    }
                    rch[tid] = res;
                }
                __syncthreads();
                if (tid < pc) for (int ii = 0; ii < WB_CH; ++ii) uacc = fma(rch[ii], xs[ii * WB_XS + tid], uacc);
            }
            __syncthreads();
            if (tid < pc) s_U[tid] = uacc;
            __syncthreads();
            if (tid < pc) {
                double t = 0.0;
                for (int c = 0; c < pc; ++c) t = fma(s_V[tid * pc + c], s_U[c], t);
                s_cand[tid] = s_beta[tid] + t;
            }
            __syncthreads();
            // ---- step halving (model.py:465-474)
            int halvings = 0; double Fcand = 0.0;
            for (;;) {
                ll = info_at(s_cand, v);
                Fcand = -(ll + 0.5 * logdet());
                double stepmax = 0.0;
                for (int a = 0; a < pc; ++a) stepmax = fmax(stepmax, fabs(s_cand[a] - s_beta[a]));
                if (!(Fcand > Fcur + P.firth_noise * fabs(Fcur)) || stepmax < P.firth_accept) break;   // noise steps accepted outright, see k_glm_firth
                __syncthreads();
                // fixed point of the halving map (every later comparison repeats this one up to step_limit, model.py:471-473): same verdict now
                bool moved = false;
                for (int a = 0; a < pc; ++a) moved = moved || ((s_beta[a] + 0.5 * (s_cand[a] - s_beta[a])) != s_cand[a]);
                __syncthreads();
                if (tid < pc) s_cand[tid] = s_beta[tid] + 0.5 * (s_cand[tid] - s_beta[tid]);
                __syncthreads();
                if (++halvings > 1000 || !moved) { failed = true; break; }
            }
            if (failed) break;
            double sn = 0.0;
            for (int a = 0; a < pc; ++a) { const double d = s_cand[a] - s_beta[a]; sn = fma(d, d, sn); }
            __syncthreads();
            if (tid < pc) s_beta[tid] = s_cand[tid];
            __syncthreads();
            sn = sqrt(sn); Fcur = Fcand; i11 = s_H[pc + 1];
            if (iter > 0 && sn_prev < 1e-4) conv = true;                               // the PREVIOUS step, model.py:477-479
            if (tid == 0 && (conv || iter + 1 >= FIRTH_SLOW_ITERS)) flags[v] |= firth_sensitive(iter + 1, sn_prev);
            sn_prev = sn;
        }
        if (tid == 0) {
            if (handed) fallback[atomicAdd(fallback_count, 1)] = (int)v;
            else {
                if (!conv) failed = true;
                uint32_t fl = flags[v];
                if (failed) {
                    fl |= SH_NOTE_FIRTH_FAIL | SH_FLAG_FILTER | SH_FLAG_FIRTH_SENSITIVE;                         // model.py:357-362
                    out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
                    for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = NAN;
                } else {
                    const double lrstat = -2.0 * (P.null_firth - (-Fcur));
                    double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);   // model.py:366-369
                    out[V + v] = pval; out[2 * V + v] = s_beta[1]; out[3 * V + v] = sqrt(i11); out[4 * V + v] = s_beta[0];
                    for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = s_beta[2 + j];
                    if (pval > P.lrtt || !isfinite(pval) || !isfinite(s_beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
                }
                flags[v] = fl;
            }
        }
    }
}

// ---- kernel 2: fit_firth (model.py:414-504) with numpy's pinv, for the listed variants ------------------------------------
__global__ __launch_bounds__(64) void k_glm_wide_firth(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int q,
                                                       const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                       const int *__restrict__ list, const int *__restrict__ count,
                                                       double *__restrict__ out, uint32_t *__restrict__ flags)
{
    const int cnt = *count;
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= cnt) return;
    const int64_t v = list[slot];
    const int pc = q + 2, N = P.N, NB64 = P.NB64;
    double beta[WIDE_PM], cand[WIDE_PM], U[WIDE_PM], I[WIDE_PM * WIDE_PM], Vm[WIDE_PM * WIDE_PM], L[WIDE_PM * WIDE_PM];
    int piv[WIDE_PM];
#pragma unroll 1
    for (int a = 0; a < pc; ++a) beta[a] = 0.0;
    beta[0] = P.ymean_logit;
    double ll, md;
    w_info(T, Vpad, v, N, NB64, q, y, W, beta, I, nullptr, &ll, &md);
    auto logdet = [&]() { for (int a = 0; a < pc * pc; ++a) L[a] = I[a]; return log(w_lu(L, piv, pc)); };
    double Fcur = -(ll + 0.5 * logdet());                                              // firth_likelihood, model.py:410-411
    double i11 = I[pc + 1], sn_prev = INFINITY;
    bool failed = false, conv = false;
#pragma unroll 1
    for (int iter = 0; iter < 1000 && !failed && !conv; ++iter) {
        w_pinv(I, Vm, pc, 1e-15, nullptr);                                             // model.py:450
#pragma unroll 1
        for (int a = 0; a < pc; ++a) U[a] = 0.0;
#pragma unroll 1
        for (int sb = 0; sb < NB64; ++sb) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
#pragma unroll 1
            for (int b = 0; b < nb; ++b) {
                const int i = sb * 64 + b;
                double x[WIDE_PM];
                x[0] = 1.0; x[1] = (double)(unsigned)((w64 >> b) & 1ull);
#pragma unroll 1
                for (int j = 0; j < q; ++j) x[2 + j] = W[(int64_t)i * q + j];
                double eta = 0.0, qf = 0.0;
#pragma unroll 1
                for (int a = 0; a < pc; ++a) eta = fma(beta[a], x[a], eta);
                const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
#pragma unroll 1
                for (int a = 0; a < pc; ++a) { double s = 0.0;
#pragma unroll 1
                    for (int c = 0; c < pc; ++c) s = fma(Vm[a * pc + c], x[c], s);
                    qf = fma(x[a], s, qf); }
                const double res = y[i] - mu + wgt * qf * (0.5 - mu);                  // hat diagonal, model.py:455-462
#pragma unroll 1
                for (int a = 0; a < pc; ++a) U[a] = fma(x[a], res, U[a]);
            }
        }
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { double s = 0.0;
#pragma unroll 1
            for (int c = 0; c < pc; ++c) s = fma(Vm[a * pc + c], U[c], s);
            cand[a] = beta[a] + s; }
        int halvings = 0; double Fcand;
#pragma unroll 1
        for (;;) {
            w_info(T, Vpad, v, N, NB64, q, y, W, cand, I, nullptr, &ll, &md);
            Fcand = -(ll + 0.5 * logdet());
            double stepmax = 0.0;
#pragma unroll 1
            for (int a = 0; a < pc; ++a) stepmax = fmax(stepmax, fabs(cand[a] - beta[a]));
            // steps below 1e-10 are accepted outright: F(new) > F(old) is rounding noise there (see k_glm_firth)
            if (!(Fcand > Fcur + P.firth_noise * fabs(Fcur)) || stepmax < P.firth_accept) break;       // step halving, model.py:467-474
            bool moved = false;                                       // fixed point of the halving map: the reference's remaining comparisons repeat this one
#pragma unroll 1
            for (int a = 0; a < pc; ++a) { const double nc = beta[a] + 0.5 * (cand[a] - beta[a]); moved = moved || (nc != cand[a]); cand[a] = nc; }
            if (++halvings > 1000 || !moved) { failed = true; break; }
        }
        if (failed) break;
        double sn = 0.0;
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { const double d = cand[a] - beta[a]; sn = fma(d, d, sn); beta[a] = cand[a]; }
        sn = sqrt(sn); Fcur = Fcand; i11 = I[pc + 1];
        if (iter > 0 && sn_prev < 1e-4) conv = true;                                   // tests the PREVIOUS step, model.py:477-479
        if (conv || iter + 1 >= FIRTH_SLOW_ITERS) flags[v] |= firth_sensitive(iter + 1, sn_prev);
        sn_prev = sn;
    }
    if (!conv) failed = true;
    uint32_t fl = flags[v];
    if (failed) {
        fl |= SH_NOTE_FIRTH_FAIL | SH_FLAG_FILTER | SH_FLAG_FIRTH_SENSITIVE;                                     // model.py:357-362
        out[V + v] = NAN; out[2 * V + v] = NAN; out[3 * V + v] = NAN; out[4 * V + v] = NAN;
#pragma unroll 1
        for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = NAN;
    } else {
        const double lrstat = -2.0 * (P.null_firth - (-Fcur));
        double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);               // model.py:366-369
        out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = sqrt(i11); out[4 * V + v] = beta[0];   // bse = sqrt(I11), model.py:491
#pragma unroll 1
        for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = beta[2 + j];
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
    }
    flags[v] = fl;
}

// ---- kernel 3: OLS through pinv (statsmodels OLS.fit(), model.py:299-312) for the listed variants ---------------------------
__global__ __launch_bounds__(64) void k_glm_wide_ols(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int q,
                                                     const double *__restrict__ y, const double *__restrict__ W, GlmParams P,
                                                     const int *__restrict__ list, const int *__restrict__ count,
                                                     double *__restrict__ out, uint32_t *__restrict__ flags)
{
    const int cnt = *count;
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= cnt) return;
    const int64_t v = list[slot];
    const int pc = q + 2, N = P.N, NB64 = P.NB64;
    double A[WIDE_PM * WIDE_PM], Pm[WIDE_PM * WIDE_PM], rhs[WIDE_PM], beta[WIDE_PM];
    // X^T X and X^T y: only the variant's row / column depends on the variant (sum x, sum x z_j, sum x y); the [1, W] block is the
    // per-run Z^T Z, Z^T y from the host (as in k_glm_ols) -- N (q + 1) masked FMAs per variant instead of N (q + 2)^2
    double sxz[WIDE_PM - 2], sxy = 0.0;
    int m = 0;
#pragma unroll
    for (int j = 0; j < WIDE_PM - 2; ++j) sxz[j] = 0.0;
#pragma unroll 1
    for (int sb = 0; sb < NB64; ++sb) {
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
        m += __popcll(w64);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            const int i = sb * 64 + b;
            const double xd = (double)(unsigned)((w64 >> b) & 1ull);
            sxy = fma(xd, y[i], sxy);
#pragma unroll
            for (int j = 0; j < WIDE_PM - 2; ++j) if (j < q) sxz[j] = fma(xd, W[(int64_t)i * q + j], sxz[j]);
        }
    }
    A[0] = P.ztz[0]; A[1] = (double)m; A[pc] = (double)m; A[pc + 1] = (double)m;
    rhs[0] = P.zty[0]; rhs[1] = sxy;
#pragma unroll
    for (int j = 0; j < WIDE_PM - 2; ++j) {
        if (j < q) {
            const double z0 = P.ztz[sidx(1 + j, 0)];
            A[(2 + j) * pc] = z0; A[2 + j] = z0; A[(2 + j) * pc + 1] = sxz[j]; A[pc + 2 + j] = sxz[j]; rhs[2 + j] = P.zty[1 + j];
        }
    }
#pragma unroll 1
    for (int j = 0; j < q; ++j)
#pragma unroll 1
        for (int k = 0; k <= j; ++k) { const double t = P.ztz[sidx(1 + j, 1 + k)]; A[(2 + j) * pc + 2 + k] = t; A[(2 + k) * pc + 2 + j] = t; }
    // Full-rank designs (the rule): solve the normal equations on the equilibrated matrix D A D, D = diag(A)^-1/2, by pivoted LU.
    // Equilibration matters: with an un-centred column such as a year (2000 +- 10) next to a 1e-2-sized one the eigenvalues of A span
    // more than the 1e-10 window of the pinv path below, which would then drop a genuine direction.  Rank-deficient designs (a
    // pivot of the equilibrated matrix below 1e-11: exact collinearity) keep numpy's pinv semantics.
    int rank = pc;
    double p11 = 0.0;
    bool solved = false;
    {
        double *E = Pm;                                    // scratch: the equilibrated copy
        double d[WIDE_PM];
        int piv[WIDE_PM];
        bool okd = true;
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { okd = okd && (A[a * pc + a] > 0.0); d[a] = 1.0 / sqrt(A[a * pc + a]); }
        if (okd) {
#pragma unroll 1
            for (int a = 0; a < pc; ++a)
#pragma unroll 1
                for (int c = 0; c < pc; ++c) E[a * pc + c] = A[a * pc + c] * d[a] * d[c];
            bool full = w_lu(E, piv, pc) != 0.0;
#pragma unroll 1
            for (int a = 0; a < pc && full; ++a) full = fabs(E[a * pc + a]) > 1e-11;
            if (full) {
#pragma unroll 1
                for (int a = 0; a < pc; ++a) beta[a] = rhs[a] * d[a];
                w_lu_solve(E, piv, pc, beta);
#pragma unroll 1
                for (int a = 0; a < pc; ++a) beta[a] *= d[a];
                double e1[WIDE_PM];
#pragma unroll 1
                for (int a = 0; a < pc; ++a) e1[a] = (a == 1) ? 1.0 : 0.0;
                w_lu_solve(E, piv, pc, e1);
                p11 = e1[1] * d[1] * d[1];                 // ((X^T X)^-1)_11
                solved = true;
            }
        }
    }
    if (!solved) {
        w_pinv(A, Pm, pc, 1e-10, &rank);                   // see k_glm_ols_pinv for the cut-off
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { double s = 0.0;
#pragma unroll 1
            for (int c = 0; c < pc; ++c) s = fma(Pm[a * pc + c], rhs[c], s);
            beta[a] = s; }
        p11 = Pm[pc + 1];
    }
    double ssr = 0.0;
#pragma unroll 1
    for (int sb = 0; sb < NB64; ++sb) {
        const uint64_t w64 = T[(int64_t)sb * Vpad + v];
        const int nb = min(64, N - sb * 64);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            const int i = sb * 64 + b;
            double f = fma(beta[1], (double)(unsigned)((w64 >> b) & 1ull), beta[0]);
#pragma unroll 1
            for (int j = 0; j < q; ++j) f = fma(beta[2 + j], W[(int64_t)i * q + j], f);
            const double r = y[i] - f;
            ssr = fma(r, r, ssr);
        }
    }
    const double dfr = (double)(N - rank);
    const double kbse = sqrt(ssr / dfr * p11);
    const double pval = sh_t_sf2(beta[1] / kbse, dfr);
    uint32_t fl = flags[v];
    if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
    out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = kbse; out[4 * V + v] = beta[0];
#pragma unroll 1
    for (int j = 0; j < q; ++j) out[(5 + j) * V + v] = beta[2 + j];
    flags[v] = fl;
}

// ---- kernel 4: fit_lineage_effect (model.py:151-199) for 17 <= 1 + lineages + covariates <= WIDE_LIN_PM ------------------------
// (k_glm_lineage in glm_kernels.hip, with a run-time width.)  X: N x pc row-major, intercept in column 0, lineages next.
#define WIDE_LIN_PM 50
__global__ __launch_bounds__(64) void k_glm_wide_lineage(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N, int NB64,
                                                         const double *__restrict__ X, int pc, int nlin, int *__restrict__ out)
{
    const int64_t v = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    double beta[WIDE_LIN_PM], g[WIDE_LIN_PM], H[WIDE_LIN_PM * WIDE_LIN_PM];
    int piv[WIDE_LIN_PM];
#pragma unroll 1
    for (int a = 0; a < pc; ++a) beta[a] = 0.0;
    int it = 0, status = 0, best = -1;
    bool fin = false;
    const double nobs = (double)N;
#pragma unroll 1
    for (;;) {
        double maxdev = 0.0;
#pragma unroll 1
        for (int a = 0; a < pc * pc; ++a) H[a] = 0.0;
#pragma unroll 1
        for (int a = 0; a < pc; ++a) g[a] = 0.0;
#pragma unroll 1
        for (int sb = 0; sb < NB64; ++sb) {
            const uint64_t w64 = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
#pragma unroll 1
            for (int b = 0; b < nb; ++b) {
                const double *x = X + (int64_t)(sb * 64 + b) * pc;
                double eta = 0.0;
#pragma unroll 1
                for (int a = 0; a < pc; ++a) eta = fma(beta[a], x[a], eta);
                const double mu = logit_cdf(eta), wgt = mu * (1.0 - mu);
                const double r = (double)(unsigned)((w64 >> b) & 1ull) - mu;
                maxdev = fmax(maxdev, fabs(r));
#pragma unroll 1
                for (int a = 0; a < pc; ++a) {
                    g[a] = fma(r, x[a], g[a]);
                    const double wa = wgt * x[a];
#pragma unroll 1
                    for (int c = 0; c <= a; ++c) H[a * pc + c] = fma(wa, x[c], H[a * pc + c]);
                }
            }
        }
#pragma unroll 1
        for (int a = 0; a < pc; ++a)
#pragma unroll 1
            for (int c = 0; c <= a; ++c) { H[a * pc + c] /= nobs; H[c * pc + a] = H[a * pc + c]; }
        if (it > 0 && maxdev <= 1e-8) { status = 1; break; }                           // PerfectSeparationError -> None
        if (fin) {
            // numpy.linalg.inv only fails on an EXACT zero pivot; huge/NaN standard errors go through np.argmax (first NaN wins)
            if (w_lu(H, piv, pc) == 0.0) { status = 2; break; }
            double bestw = -1.0; int first_nan = -1;
#pragma unroll 1
            for (int a = 1; a <= nlin; ++a) {
#pragma unroll 1
                for (int c = 0; c < pc; ++c) g[c] = (c == a) ? 1.0 : 0.0;
                w_lu_solve(H, piv, pc, g);
                const double wald = fabs(beta[a]) / sqrt(g[a] / nobs);
                if (isnan(wald)) { if (first_nan < 0) first_nan = a - 1; }
                else if (wald > bestw) { bestw = wald; best = a - 1; }
            }
            if (first_nan >= 0) best = first_nan;
            break;
        }
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { H[a * pc + a] -= 1e-10; g[a] = g[a] / nobs; }
        if (w_lu(H, piv, pc) == 0.0) { status = 2; break; }                            // LinAlgError -> None
        w_lu_solve(H, piv, pc, g);
        bool moving = false;
#pragma unroll 1
        for (int a = 0; a < pc; ++a) { beta[a] += g[a]; moving = moving || (fabs(g[a]) > 1e-8); }
        ++it;
        if (!moving || it >= 35) fin = true;
    }
    out[v] = (status == 0) ? best : -1;
}

// ---- kernel 4b: fit_lineage_effect by one workgroup per variant (the staging and tiles of kernel 1b; the response is the variant's bit,
// the design row comes from X).  One lane per variant in scratch memory ran at 1.8 k variants/s for 30 lineage clusters at N = 5000.
#define WL_XS 53
__global__ __launch_bounds__(256) void k_glm_wide_lineage_blk(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N,
                                                              const double *__restrict__ X, int pc, int nlin, int *__restrict__ out, LinList L)
{
    __shared__ double xs[WB_CH * WL_XS], wch[WB_CH], rch[WB_CH], s_beta[WIDE_LIN_PM + 2], s_g[WIDE_LIN_PM + 2];
    __shared__ double s_H[WIDE_LIN_PM * WIDE_LIN_PM], s_red[256 * 9], s_sc[8], s_wald[WIDE_LIN_PM];
    __shared__ int s_ctl, s_piv[WIDE_LIN_PM], s_i[2];
    const int tid = threadIdx.x;
    const int NT = (pc + 2) / 3, NTT = NT * (NT + 1) / 2, G = 256 / NTT > 0 ? 256 / NTT : 1;
    const double nobs = (double)N;
    const int64_t nrows = L.list ? (int64_t)*L.cnt : V;             // (LinList: the compacted rows of the job stream, glm_lineage.hip)
    for (int64_t slot = blockIdx.x; slot < nrows; slot += gridDim.x) {
        const int64_t v = L.list ? (int64_t)L.list[slot] : slot;
        if (L.list && !lin_wanted(L.flags[v], L.mode)) { if (threadIdx.x == 0) out[slot] = -1; continue; }
        int it = 0, status = 0, best = -1;                          // kept identical in every thread
        __syncthreads();
        if (tid < WIDE_LIN_PM + 2) s_beta[tid] = 0.0;               // statsmodels' default start
        if (tid == 0) s_ctl = 0;
        __syncthreads();
        for (;;) {
            double gacc = 0.0, mdev = 0.0;
            for (int e = tid; e < pc * pc; e += 256) s_H[e] = 0.0;
            // tiles: with up to 153 of them a thread may own none or (pc <= 27) several sample groups' worth; accumulate per (group, tile)
            double acc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            const int grp = tid / NTT, tile = tid - grp * NTT;
            const bool tiler = grp < G && tid < G * NTT;
            int ta = 0; { int t = tile; while (t >= ta + 1) { t -= ta + 1; ++ta; } }
            const int tc = tile - ta * (ta + 1) / 2;
            for (int c0 = 0; c0 < N; c0 += WB_CH) {
                __syncthreads();
                if (tid < WB_CH) {
                    const int i = c0 + tid;
                    double *row = xs + tid * WL_XS;
                    if (i < N) {
                        const double yi = (double)(unsigned)((T[(int64_t)(i >> 6) * Vpad + v] >> (i & 63)) & 1ull);
                        double eta = 0.0;
#pragma unroll 1
                        for (int a = 0; a < pc; ++a) { const double z = X[(int64_t)i * pc + a]; row[a] = z; eta = fma(s_beta[a], z, eta); }
#pragma unroll 1
                        for (int a = pc; a < 3 * NT; ++a) row[a] = 0.0;
                        const double mu = logit_cdf(eta), r = yi - mu;
                        wch[tid] = mu * (1.0 - mu); rch[tid] = r;
                        mdev = fmax(mdev, fabs(r));
                    } else {
#pragma unroll 1
                        for (int a = 0; a < 3 * NT; ++a) row[a] = 0.0;
                        wch[tid] = 0.0; rch[tid] = 0.0;
                    }
                }
                __syncthreads();
                if (tiler) {
                    for (int ii = grp; ii < WB_CH; ii += G) {
                        const double *row = xs + ii * WL_XS;
                        const double w = wch[ii];
                        const double a0 = w * row[3 * ta], a1 = w * row[3 * ta + 1], a2 = w * row[3 * ta + 2];
                        const double c0v = row[3 * tc], c1v = row[3 * tc + 1], c2v = row[3 * tc + 2];
                        acc[0] = fma(a0, c0v, acc[0]); acc[1] = fma(a0, c1v, acc[1]); acc[2] = fma(a0, c2v, acc[2]);
                        acc[3] = fma(a1, c0v, acc[3]); acc[4] = fma(a1, c1v, acc[4]); acc[5] = fma(a1, c2v, acc[5]);
                        acc[6] = fma(a2, c0v, acc[6]); acc[7] = fma(a2, c1v, acc[7]); acc[8] = fma(a2, c2v, acc[8]);
                    }
                }
                if (tid < pc) for (int ii = 0; ii < WB_CH; ++ii) gacc = fma(rch[ii], xs[ii * WL_XS + tid], gacc);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 9; ++k) s_red[tid * 9 + k] = tiler ? acc[k] : 0.0;
            if (tid < pc) s_g[tid] = gacc;
            __syncthreads();
            for (int e = tid; e < NTT * 9; e += 256) {
                const int tl = e / 9, k = e - tl * 9;
                double t = 0.0;
                for (int gq = 0; gq < G; ++gq) t += s_red[(gq * NTT + tl) * 9 + k];
                int ra = 0; { int u = tl; while (u >= ra + 1) { u -= ra + 1; ++ra; } }
                const int rc = tl - ra * (ra + 1) / 2;
                const int a = 3 * ra + k / 3, c = 3 * rc + k % 3;
                if (a < pc && c <= a) { s_H[a * pc + c] = t; s_H[c * pc + a] = t; }
            }
            __syncthreads();
            if (tid < WB_CH) s_red[tid] = mdev;
            __syncthreads();
            if (tid == 0) { double t = 0.0; for (int k = 0; k < WB_CH; ++k) t = fmax(t, s_red[k]); s_sc[3] = t; }
            __syncthreads();
            const double maxdev = s_sc[3];
            const int mode = s_ctl;
            __syncthreads();
            for (int e = tid; e < pc * pc; e += 256) s_H[e] = s_H[e] / nobs;
            __syncthreads();
            if (it > 0 && maxdev <= 1e-8) { status = 1; break; }                       // PerfectSeparationError -> None
            if (mode == 1) {
                // numpy.linalg.inv only fails on an EXACT zero pivot; huge/NaN standard errors go through np.argmax (first NaN wins)
                if (wb_lu(s_H, s_piv, pc, tid, s_i, s_sc + 4) == 0.0) { status = 2; break; }
                if (tid >= 1 && tid <= nlin) {                                         // diagonal element tid of the inverse
                    double col[WIDE_LIN_PM];
                    for (int a = 0; a < pc; ++a) col[a] = (a == tid) ? 1.0 : 0.0;
                    for (int c = 0; c < pc; ++c) { const int p = s_piv[c]; if (p != c) { const double t = col[c]; col[c] = col[p]; col[p] = t; } }
                    for (int c = 0; c < pc; ++c) for (int r = c + 1; r < pc; ++r) col[r] = fma(-s_H[r * pc + c], col[c], col[r]);
                    for (int c = pc - 1; c >= tid; --c) {
                        for (int j = c + 1; j < pc; ++j) col[c] = fma(-s_H[c * pc + j], col[j], col[c]);
                        col[c] = col[c] / s_H[c * pc + c];
                    }
                    s_wald[tid] = fabs(s_beta[tid]) / sqrt(col[tid] / nobs);
                }
                __syncthreads();
                double bestw = -1.0; int first_nan = -1;
                for (int a = 1; a <= nlin; ++a) {
                    const double wald = s_wald[a];
                    if (isnan(wald)) { if (first_nan < 0) first_nan = a - 1; }
                    else if (wald > bestw) { bestw = wald; best = a - 1; }
                }
                if (first_nan >= 0) best = first_nan;
                break;
            }
            if (tid < pc) { s_H[tid * pc + tid] -= 1e-10; s_g[tid] = s_g[tid] / nobs; }
            if (wb_lu(s_H, s_piv, pc, tid, s_i, s_sc + 4) == 0.0) { status = 2; break; }     // LinAlgError -> None
            wb_solve(s_H, s_piv, pc, s_g, tid);
            bool moving = false;
            for (int a = 0; a < pc; ++a) moving = moving || (fabs(s_g[a]) > 1e-8);
            __syncthreads();
            if (tid < pc) s_beta[tid] += s_g[tid];
            ++it;
            if (tid == 0 && (!moving || it >= 35)) s_ctl = 1;
            __syncthreads();
        }
        if (tid == 0) out[slot] = (status == 0) ? best : -1;
    }
}

extern "C" hipError_t shk_glm_wide_lineage(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, int N, int NB64, const double *X,
                                           int pc, int nlin, int *out, LinList L)
{
    if (pc > WIDE_LIN_PM) return hipErrorInvalidValue;
    const int blk = 1;
    if (blk || L.list) hipLaunchKernelGGL(k_glm_wide_lineage_blk, dim3((unsigned)std::min<int64_t>(V, 2048)), dim3(256), 0, st, T, Vpad, V, N, X, pc, nlin, out, L);
    else hipLaunchKernelGGL(k_glm_wide_lineage, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, st, T, Vpad, V, N, NB64, X, pc, nlin, out);
    return hipGetLastError();
}

extern "C" hipError_t shk_glm_wide(hipStream_t st, int which, int q, const uint64_t *T, int64_t Vpad, int64_t V, const double *y,
                                   const double *W, const uint64_t *y1, const uint64_t *y0, const double *yc, GlmParams P,
                                   double *out, uint32_t *flags, int *flist, int *fcount, int *olist, int *ocount, int *nlist, int *ncount)
{
    const dim3 grid((unsigned)((V + 63) / 64)), blk(64);
    if (which == 0) hipLaunchKernelGGL(k_glm_wide, grid, blk, 0, st, T, Vpad, V, q, y, W, y1, y0, yc, P, out, flags, flist, fcount, olist, ocount,
                                       nlist, ncount);
    else if (which == 4) hipLaunchKernelGGL(k_glm_wide_firth_blk, dim3((unsigned)std::min<int64_t>(V, 2048)), dim3(256), 0, st, T, Vpad, V, q, y, W, P,
                                            flist, fcount, out, flags, olist, ocount);
    else if (which == 5) hipLaunchKernelGGL(k_glm_wide_firth, grid, blk, 0, st, T, Vpad, V, q, y, W, P, olist, ocount, out, flags);
    else if (which == 3) hipLaunchKernelGGL(k_glm_wide_newton_blk, dim3((unsigned)std::min<int64_t>(V, 2048)), dim3(256), 0, st, T, Vpad, V, q, y, W, P,
                                            nlist, ncount, out, flags, flist, fcount);
    else if (which == 1) hipLaunchKernelGGL(k_glm_wide_firth, grid, blk, 0, st, T, Vpad, V, q, y, W, P, flist, fcount, out, flags);
    else hipLaunchKernelGGL(k_glm_wide_ols, grid, blk, 0, st, T, Vpad, V, q, y, W, P, olist, ocount, out, flags);
    return hipGetLastError();
}
