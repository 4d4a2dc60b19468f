// glm_tile.hip -- logistic Newton (statsmodels Logit.fit(method='newton'), SM:base/optimizer.py:407-446, as called from
// pyseer/model.py:316-344) with every sum over the samples on the matrix pipe.
//
// The per-lane kernels of glm_kernels.hip keep one variant per lane and all p x p sums in that lane's registers: 78 fp64 (or fp32)
// accumulators, 200-250 VGPRs, two wavefronts per SIMD, dependent fp64 chains -- they issue on ~40 % of their cycles and the fp64
// passes spill.  Here a wavefront owns 16 variants and walks the samples 16 at a time; the three sums of a Newton pass are small GEMMs
// whose shared operand is a per-run table of the design [1, z] (the variant column is the only per-variant part of the design):
//
//   eta  [16 samples x 16 variants] = Zx [16 x K] . B [K x 16]                       v_mfma_f64_16x16x4_f64, K = 1 + q padded to 4
//          + x . beta_1   (per element, from the packed bits), then mu, w = mu(1-mu), r = y - mu on the VALU: 4 elements per lane
//   score[16 variants x 16 cols]    = R^T [16 x 16 samples] . Zx [16 samples x 16]    v_mfma_f64: the C fragment of eta IS the A fragment
//   X^T W X: pairs of [1, z] columns = W^T . ZZ [16 samples x NP],  variant row/col   = (W o X)^T . Zx
//          fp32 (v_mfma_f32_16x16x4_f32, table of products in fp32) while iterating; fp64 in the last pass
//
// The C/D layout of the f64 MFMA (row = (lane >> 4) + 4 reg, col = lane & 15) makes register `reg` of the eta fragment exactly the A
// operand (variant = lane & 15, k = lane >> 4) of the MFMA that contracts samples 4 reg .. 4 reg + 3: no shuffles, no LDS round trip.
// Per lane: ~110 VGPRs instead of 250, nothing spills; what is left on the VALU is the logistic function itself.
//
// A block = W wavefronts (8, or 4 for wide designs) that share the table tiles through LDS (double-buffered, one barrier per 16 samples).
// Iteration: start at the null-model MLE (GlmParams.warm), fp64 score with fp32 Hessian until the step is <= GlmParams.fast_tol, then ONE
// pass with the fp64 Hessian that evaluates llf, the separation callback and bse[1] at that beta and takes the exact Newton step.  Variants
// that do not converge cleanly (separation signal, failed pivot, 12 steps) go to k_glm_slow, which restarts them from the reference's start
// vector on the reference's trajectory (and then to k_glm_final), exactly as from k_glm_fast.
#include <algorithm>
#include "common.h"
#include "glm_common.h"
#include "glm_tile.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct GlmWork { double *bw; int *state; int *slow_list; int *slow_count; int *tile_list; int *tile_count; };

// LDL^T helpers of glm_kernels.hip (same arithmetic; kept local to this translation unit)
template <int P>
__device__ __forceinline__ bool t_ldl_factor(double (&A)[P * (P + 1) / 2], double rel_tol)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double v[P];
        const double ajj = A[sidx(j, j)];
        double d = ajj;
#pragma unroll
        for (int k = 0; k < j; ++k) { v[k] = A[sidx(j, k)] * A[sidx(k, k)]; d = fma(-A[sidx(j, k)], v[k], d); }
        if (d == 0.0 || fabs(d) <= rel_tol * fabs(ajj)) ok = false;
        A[sidx(j, j)] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = A[sidx(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-A[sidx(i, k)], v[k], s);
            A[sidx(i, j)] = s * inv;
        }
    }
    return ok;
}

template <int P>
__device__ __forceinline__ void t_ldl_solve(const double (&A)[P * (P + 1) / 2], double (&b)[P])
{
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] = fma(-A[sidx(i, k)], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) b[i] = b[i] / A[sidx(i, i)];
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
#pragma unroll
        for (int k = i + 1; k < P; ++k) b[i] = fma(-A[sidx(k, i)], b[k], b[i]);
    }
}

// One pass over the samples for the 16 variants of this wavefront.  FINAL: Hessian in fp64 + log-likelihood; else Hessian in fp32.
// Leaves, per variant row of `gat` (stride GST doubles): [0,16) score over [1, z]; [16,32) sum w x [1, z]; [32, 32 + NC) sum w zz pairs;
// then g_1 = sum r x, llf, max |y - mu|.
#ifndef TILE_ABL
#define TILE_ABL 0
#endif
template <int Q, bool FINAL, int ABL = TILE_ABL>
__device__ __forceinline__ void tile_pass(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, const uint64_t *__restrict__ y1,
                                          const double *__restrict__ tab, double *tbuf, double *gat, const double (&bq)[TileCfg<Q>::KK],
                                          double b1)
{
    using C = TileCfg<Q>;
    constexpr int TBUF = C::TBUF, NC = C::NC, GST = C::GST;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, lc = lane & 15, lg = lane >> 4;
    const int ntile = (N + 15) >> 4;
    // what a block stages per tile: [Zt | ZtT] always, then the products table in the precision of this pass
    constexpr int N16_HEAD = 512 / 2, N16_ZZ = (FINAL ? 16 * NC : 8 * NC) / 2, N16 = N16_HEAD + N16_ZZ;
    constexpr int ZZ_OFF = FINAL ? C::OFF_ZZ64 : C::OFF_ZZ32;
    constexpr int MAXR = (N16 + 255) / 256;                         // rounds of 16-byte chunks per thread at the smallest block (256 threads)
    uint4 stg[MAXR];
    auto fetch = [&](int t) {
        const double *src = tab + (int64_t)t * C::TILE_D;
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int c = tid + j * nthr;
            if (c < N16) {
                const int d = c < N16_HEAD ? 2 * c : ZZ_OFF + 2 * (c - N16_HEAD);
                stg[j] = *reinterpret_cast<const uint4 *>(src + d);
            }
        }
    };
    auto stash = [&](double *buf) {
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const int c = tid + j * nthr;
            if (c < N16) *reinterpret_cast<uint4 *>(buf + 2 * c) = stg[j];
        }
    };
    v4d G = {0, 0, 0, 0}, Hx64 = {0, 0, 0, 0};
    v4f Hx32 = {0, 0, 0, 0};
    v4d Hp64[FINAL ? C::NCB : 1];
    v4f Hp32[FINAL ? 1 : C::NCB];
#pragma unroll
    for (int cb = 0; cb < (FINAL ? C::NCB : 1); ++cb) Hp64[cb] = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int cb = 0; cb < (FINAL ? 1 : C::NCB); ++cb) Hp32[cb] = (v4f){0, 0, 0, 0};
    double g1 = 0.0, ll = 0.0, maxdev = 0.0;
    uint64_t w64 = 0, y64 = 0;

    fetch(0);
    __syncthreads();                                                 // nobody reads the table buffers of the previous pass any more
    stash(tbuf);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const double *buf = tbuf + (t & 1) * TBUF;
        if (t + 1 < ntile) fetch(t + 1);
        if ((t & 3) == 0) { w64 = T[(int64_t)(t >> 2) * Vpad + v]; y64 = y1[t >> 2]; }
        const int boff = (t & 3) * 16 + lg;
        // ---- every LDS operand of this tile into registers first: the MFMAs below then issue back to back instead of each waiting
        // for its own ds_read (the compiler sinks a load next to its use), and the other wavefront of the SIMD fills the wait
        double ztt[C::KK], zt[4];
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) ztt[kk] = buf[C::OFF_ZTT + (4 * kk + lg) * 16 + lc];
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[r] = buf[C::OFF_ZT + (4 * r + lg) * 16 + lc];
        double zz64[FINAL ? C::NCB : 1][4];
        float zz32[FINAL ? 1 : C::NCB][4];
        if (!(ABL & 2)) {
            if (FINAL) {
                const double *zz = buf + 512;
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zz64[cb][r] = zz[(4 * r + lg) * NC + cb * 16 + lc];
            } else {
                const float *zz = reinterpret_cast<const float *>(buf + 512);
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zz32[cb][r] = zz[(4 * r + lg) * NC + cb * 16 + lc];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- eta = Zx . B  (+ x beta_1 per element)
        v4d Ce = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) Ce = __builtin_amdgcn_mfma_f64_16x16x4f64(ztt[kk], bq[kk], Ce, 0, 0, 0);
        double rr[4], ww[4], wx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bit = boff + 4 * r;
            const bool xb = (w64 >> bit) & 1ull, yb = (y64 >> bit) & 1ull;
            const bool valid = t * 16 + lg + 4 * r < N;
            const double eta = Ce[r] + (xb ? b1 : 0.0);
            double mu;
            if (ABL & 1) mu = fmin(fmax(0.5 + 0.1 * eta, 0.01), 0.99);          // timing ablation: no exp, no division
            else { const double en = exp(-eta); mu = 1.0 / (1.0 + en); }      // SM Logit.cdf
            double wgt = mu * (1.0 - mu), res = (yb ? 1.0 : 0.0) - mu;
            if (!valid) { wgt = 0.0; res = 0.0; }                             // padding rows of the last tile (their table rows are zero too)
            maxdev = fmax(maxdev, fabs(res));
            if (FINAL) { const double lm = log(mu); ll += valid ? (yb ? lm : lm - eta) : 0.0; }   // SM Logit.loglike, as info_pass
            g1 += xb ? res : 0.0;
            rr[r] = res; ww[r] = wgt; wx[r] = xb ? wgt : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- score, the variant row of X^T W X (same B operand: rows of [1, z]) and the [1, z] x [1, z] block (products table);
        // consecutive MFMAs go to different accumulators
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[r], zt[r], G, 0, 0, 0);
            if (FINAL) {
                Hx64 = __builtin_amdgcn_mfma_f64_16x16x4f64(wx[r], zt[r], Hx64, 0, 0, 0);
                if (!(ABL & 2)) {
#pragma unroll
                    for (int cb = 0; cb < C::NCB; ++cb) Hp64[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ww[r], zz64[cb][r], Hp64[cb], 0, 0, 0);
                }
            } else {
                const float wf = (float)ww[r];
                Hx32 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)wx[r], (float)zt[r], Hx32, 0, 0, 0);
                if (!(ABL & 2)) {
#pragma unroll
                    for (int cb = 0; cb < C::NCB; ++cb) Hp32[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, zz32[cb][r], Hp32[cb], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntile) stash(tbuf + ((t + 1) & 1) * TBUF);
        if (!(ABL & 4)) __syncthreads();                                     // timing ablation 4: no barrier (results garbage)
    }
    // the four lanes (lane >> 4 = 0..3) of a variant hold partial sums over their samples
    g1 += __shfl_xor(g1, 16, 64); g1 += __shfl_xor(g1, 32, 64);
    ll += __shfl_xor(ll, 16, 64); ll += __shfl_xor(ll, 32, 64);
    maxdev = fmax(maxdev, __shfl_xor(maxdev, 16, 64)); maxdev = fmax(maxdev, __shfl_xor(maxdev, 32, 64));
    // C/D layouts: f64 row = (lane >> 4) + 4 reg ; f32 row = 4 (lane >> 4) + reg ; col = lane & 15.  Rows are variants here.
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gat[(lg + 4 * r) * GST + lc] = G[r];
        if (FINAL) gat[(lg + 4 * r) * GST + 16 + lc] = Hx64[r];
        else gat[(4 * lg + r) * GST + 16 + lc] = (double)Hx32[r];
#pragma unroll
        for (int cb = 0; cb < C::NCB; ++cb) {
            if (FINAL) gat[(lg + 4 * r) * GST + 32 + cb * 16 + lc] = Hp64[cb][r];
            else gat[(4 * lg + r) * GST + 32 + cb * 16 + lc] = (double)Hp32[cb][r];
        }
    }
    if (lg == 0) { gat[lc * GST + 32 + NC] = g1; gat[lc * GST + 32 + NC + 1] = ll; gat[lc * GST + 32 + NC + 2] = maxdev; }
    __syncthreads();
}

template <int Q>
__global__ __launch_bounds__(512) void k_glm_tile(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, const uint64_t *__restrict__ y1,
                                                  const double *__restrict__ tab, GlmParams P, GlmWork wk, double *__restrict__ out,
                                                  uint32_t *__restrict__ flags, int *__restrict__ firth_list, int *__restrict__ firth_count)
{
    using C = TileCfg<Q>;
    constexpr int PC = Q + 2, GST = C::GST, NC = C::NC;
    extern __shared__ double tl_lds[];
    const int nlist = *wk.tile_count;
    const int nw = blockDim.x >> 6;
    const int vb0 = blockIdx.x * (16 * nw);
    if (vb0 >= nlist) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
    double *tbuf = tl_lds;
    double *gat = tl_lds + 2 * C::TBUF + (size_t)wave * 16 * GST;
    // lanes 0..15 of a wavefront double as the per-variant "solver" lanes (variant = lane); every lane carries the operand fragments
    const int slot = vb0 + wave * 16 + lc;
    const bool live = slot < nlist;
    const int64_t v = wk.tile_list[live ? slot : 0];
    const bool solver = lane < 16;
    const int N = P.N;
    const double nobs = (double)N;

    double beta[PC];                                                 // solver lanes: [intercept, variant, z...] in the standardised coordinates
#pragma unroll
    for (int a = 0; a < PC; ++a) beta[a] = 0.0;
    beta[0] = P.warm_on ? P.warm[0] : P.ymean_logit;
#pragma unroll
    for (int j = 0; j < Q; ++j) beta[2 + j] = P.warm_on ? P.warm[1 + j] : 0.0;
    bool active = live, need_slow = false;
    int it = 0;
    double bq[C::KK], b1 = 0.0;
    auto spread = [&]() {                                            // solver lanes -> operand fragments of all lanes, through the gather area
        if (solver) {
            gat[lane * GST + 0] = beta[0];
#pragma unroll
            for (int j = 0; j < Q; ++j) gat[lane * GST + 1 + j] = beta[2 + j];
#pragma unroll
            for (int k = Q + 1; k < 4 * C::KK; ++k) gat[lane * GST + k] = 0.0;
            gat[lane * GST + 16] = beta[1];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) bq[kk] = gat[lc * GST + 4 * kk + lg];
        b1 = gat[lc * GST + 16];
        __syncthreads();
    };
    auto load_sums = [&](double (&H)[PC * (PC + 1) / 2], double (&g)[PC], double &ll, double &maxdev) {
        const double *gp = gat + lane * GST;
        g[0] = gp[0]; g[1] = gp[32 + NC];
#pragma unroll
        for (int j = 0; j < Q; ++j) g[2 + j] = gp[1 + j];
        H[sidx(1, 0)] = gp[16]; H[sidx(1, 1)] = gp[16];                // x^2 = x
#pragma unroll
        for (int j = 0; j < Q; ++j) H[sidx(2 + j, 1)] = gp[17 + j];
#pragma unroll
        for (int a = 0; a <= Q; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) H[sidx(a ? a + 1 : 0, b ? b + 1 : 0)] = gp[32 + a * (a + 1) / 2 + b];
        ll = gp[32 + NC + 1]; maxdev = gp[32 + NC + 2];
    };

    // ---- iterate: fp64 score, fp32 Hessian
#if TILE_ABL
    int abl_pass = 0;
#endif
    for (;;) {
        spread();
        tile_pass<Q, false>(T, Vpad, v, N, y1, tab, tbuf, gat, bq, b1);
        if (solver && active) {
            double H[PC * (PC + 1) / 2], g[PC], ll, maxdev;
            load_sums(H, g, ll, maxdev);
            if (it > 0 && maxdev <= 1e-8) { need_slow = true; active = false; }       // separation signal: the reference's trajectory decides
            else {
#pragma unroll
                for (int a = 0; a < PC * (PC + 1) / 2; ++a) H[a] = H[a] / nobs;
#pragma unroll
                for (int a = 0; a < PC; ++a) { H[sidx(a, a)] -= 1e-10; g[a] = g[a] / nobs; }     // optimizer.py:415-423 (sign folded)
                if (!t_ldl_factor<PC>(H, 1e-4)) { need_slow = true; active = false; }
                else {
                    t_ldl_solve<PC>(H, g);
                    double step = 0.0; bool finite = true;
#pragma unroll
                    for (int a = 0; a < PC; ++a) { beta[a] += g[a]; step = fmax(step, fabs(g[a])); finite = finite && isfinite(beta[a]); }
                    ++it;
                    if (!finite) { need_slow = true; active = false; }
                    else if (step <= P.fast_tol) active = false;
                    else if (it >= 12) { need_slow = true; active = false; }
                }
            }
        }
#if TILE_ABL
        if (++abl_pass == 3) break;                                  // timing builds: exactly three passes whatever the numbers say
        active = live;
#else
        if (!__syncthreads_or(solver && active)) break;
#endif
    }
#if TILE_ABL
    need_slow = false;
#endif
    // ---- the pass at the final beta: llf, separation callback, fp64 information matrix, exact Newton step (see k_glm_final)
    spread();
    tile_pass<Q, true>(T, Vpad, v, N, y1, tab, tbuf, gat, bq, b1);
    if (!(solver && live)) return;
    if (P.dbg) atomicAdd(&P.dbg[2], it), atomicAdd(&P.dbg[3], 1);
    if (!need_slow) {
        double H[PC * (PC + 1) / 2], g[PC], llf, maxdev, bse1 = NAN;
        load_sums(H, g, llf, maxdev);
        int status = 0;
        if (maxdev <= 1e-8) status = 1;
        else {
#pragma unroll
            for (int a = 0; a < PC * (PC + 1) / 2; ++a) H[a] = H[a] / nobs;
            if (!t_ldl_factor<PC>(H, 4.0e-16)) status = 2;
            else {
                double e[PC];
#pragma unroll
                for (int a = 0; a < PC; ++a) { e[a] = (a == 1) ? 1.0 : 0.0; g[a] = g[a] / nobs; }
                t_ldl_solve<PC>(H, e);
                bse1 = sqrt(e[1] / nobs);                                                  // SM:base/model.py:533-534, model.py:332
                t_ldl_solve<PC>(H, g);
                double step = 0.0; bool finite = true;
#pragma unroll
                for (int a = 0; a < PC; ++a) { step = fmax(step, fabs(g[a])); finite = finite && isfinite(g[a]); }
                if ((!finite || step > 1e-7) && !TILE_ABL) need_slow = true;                 // not where the fast phase promised: restart
                else {
#pragma unroll
                    for (int a = 0; a < PC; ++a) beta[a] += g[a];
                }
            }
        }
        if (!need_slow) {
            uint32_t fl = flags[v];
            bool to_firth = false;
            if (status == 1) { fl |= SH_NOTE_PERFECT_SEP; to_firth = true; }
            else if (status == 2) { fl |= SH_NOTE_MATRIX_INV; to_firth = true; }
            else if (bse1 > 3.0) { fl |= SH_NOTE_HIGH_BSE; to_firth = true; }                 // model.py:332-334
            else {
                if (P.ws) {                                          // z' = (z - mean) / scale  =>  b = b' / scale, b0 = b0' - sum b' mean / scale
#pragma unroll
                    for (int j = 0; j < Q; ++j) { beta[2 + j] = beta[2 + j] / P.wstd[Q + j]; beta[0] = fma(-beta[2 + j], P.wstd[j], beta[0]); }
                }
                const double lrstat = -2.0 * (P.null_llf - llf);
                double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);              // model.py:336-339
                out[V + v] = pval; out[2 * V + v] = beta[1]; out[3 * V + v] = bse1; out[4 * V + v] = beta[0];
#pragma unroll
                for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = beta[2 + j];
                if (pval > P.lrtt || !isfinite(pval) || !isfinite(beta[1])) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;   // model.py:384
            }
            flags[v] = fl;
            if (to_firth) { const int s2 = atomicAdd(firth_count, 1); firth_list[s2] = (int)v; }
        }
    }
    if (need_slow) { const int s2 = atomicAdd(wk.slow_count, 1); wk.slow_list[s2] = (int)v; }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int Q>
static hipError_t launch_tile(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, const uint64_t *y1, const double *tab, GlmParams P,
                              GlmWork wk, double *out, uint32_t *flags, int *flist, int *fcount)
{
    using C = TileCfg<Q>;
    const int nw = C::NC <= 80 ? 8 : 4;
    const size_t lds = (size_t)(2 * C::TBUF + nw * 16 * C::GST) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_glm_tile<Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    const int per = 16 * nw;
    hipLaunchKernelGGL(k_glm_tile<Q>, dim3((unsigned)((V + per - 1) / per)), dim3(64 * nw), lds, st, T, Vpad, V, y1, tab, P, wk, out, flags, flist, fcount);
    return hipGetLastError();
}

extern "C" hipError_t shk_glm_tile(hipStream_t st, int Q, const uint64_t *T, int64_t Vpad, int64_t V, const uint64_t *y1, const double *tab,
                                   GlmParams P, double *bw, int *state, int *slow_list, int *slow_count, int *tile_list, int *tile_count,
                                   double *out, uint32_t *flags, int *flist, int *fcount)
{
    GlmWork wk{bw, state, slow_list, slow_count, tile_list, tile_count};
#define TILE_CASE(q) case q: return launch_tile<q>(st, T, Vpad, V, y1, tab, P, wk, out, flags, flist, fcount);
    switch (Q) {
        TILE_CASE(0) TILE_CASE(1) TILE_CASE(2) TILE_CASE(3) TILE_CASE(4) TILE_CASE(5) TILE_CASE(6) TILE_CASE(7)
        TILE_CASE(8) TILE_CASE(9) TILE_CASE(10) TILE_CASE(11) TILE_CASE(12) TILE_CASE(13) TILE_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef TILE_CASE
}

extern "C" int shk_glm_tile_doubles(int Q)                           // doubles per 16-sample tile of the table (glm_tile.h)
{
    switch (Q) {
#define TD_CASE(q) case q: return TileCfg<q>::TILE_D;
        TD_CASE(0) TD_CASE(1) TD_CASE(2) TD_CASE(3) TD_CASE(4) TD_CASE(5) TD_CASE(6) TD_CASE(7)
        TD_CASE(8) TD_CASE(9) TD_CASE(10) TD_CASE(11) TD_CASE(12) TD_CASE(13) TD_CASE(14)
#undef TD_CASE
    default: return 0;
    }
}
