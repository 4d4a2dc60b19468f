// glm_blk.h -- helpers of the one-WORKGROUP-per-variant kernels (k_glm_slow_blk in glm_kernels.hip; k_firth_blk, k_glm_firth_pinv in
// glm_firth_v1.hip; k_glm_ols_pinv in glm_ols.hip): workgroup reductions in a fixed order, the information matrix and log-likelihood with the
// samples spread over 256 threads, numpy.linalg.det / pinv restated (model.py:450).
#pragma once
#include "glm_passes.h"

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

