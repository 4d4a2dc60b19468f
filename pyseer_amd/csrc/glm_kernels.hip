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
//
// This file: the LOGISTIC product path (k_glm_fast: prefilter + routing; the rounds of the single-precision Newton phase and the chord rounds;
// k_glm_slow_blk: the fp64 restart; k_glm_final / k_glm_ll + k_glm_dpass(_pk) + k_glm_finish: the fp64 evaluation and the decisions) and its
// launcher.  Round 5 split the other families off: glm_passes.h (the sample passes as device functions), glm_blk.h (workgroup-per-variant
// helpers), glm_firth_v1.hip (round 2's Firth state machine, k_firth_blk, k_glm_firth_pinv), glm_ols.hip (continuous phenotype),
// glm_lineage.hip (fit_lineage_effect).  The Firth product path of a forced-Firth run is firth_fast.hip / firth_fast_w.hip / firth_rounds.hip.
#include "glm_passes.h"
#include "glm_blk.h"

extern "C" hipError_t shk_glm_firth_pinv_launch(hipStream_t, int, const uint64_t *, int64_t, int64_t, const double *, const double *, GlmParams, int *, int *,
                                                double *, uint32_t *);
extern "C" hipError_t shk_glm_ols_launch(hipStream_t, int, int, const uint64_t *, int64_t, int64_t, const double *, const double *, const uint64_t *,
                                         const uint64_t *, const double *, const double *, const double *, GlmParams, double *, uint32_t *, int *, int *);

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
// The final information matrix is the exact null-model matrix + sums of the DIFFERENCES w - w0 taken in single precision (the DELTA passes).
// Two things put 1e-7 .. 3e-7 on bse where the variant's effect is strong (|beta_k| >= 2: tools/gpu_bse_probe.py, profiles/r06/bse_probe_before.txt;
// 1e-8 elsewhere) and are taken out here (round 6):
//  (a) beta rounded to float: the error of the intercept and of the variant's coefficient (2^-25 |beta|: 1.2e-7 at |beta| = 4) is the SAME for
//      every carrier -- a shift of eta that does not average out over the samples as a sample's own rounding does.  The two constants of eta,
//      c(k) = beta_0 + k beta_k, are split hi + lo against the value the loop's own fma produces (c0h + k dh), and the lo parts (lo0 + k dl)
//      start the covariates' chain.  (The covariates' coefficients stay rounded: their part is 4e-9 on bse.)
//  (b) w = mu - mu^2 cancels for mu -> 1 (the carriers of a strong positive effect: 1 - mu = 0.02 from a float mu is 3e-6 relative).  w is even
//      in eta: t = exp(-|eta|) <= 1 cannot overflow, and w = a - a^2 with a = t / (1 + t) = min(mu, 1 - mu) is relatively exact on both sides.
//  (c) the variant's own row (sum over the CARRIERS of d, d z_j) is accumulated in single precision within a 64-sample word and in double
//      precision across words.  Where the carriers' eta has moved far from the null model's (beta_0 + beta_k - b0(null) = +3.5: w0 = 0.22,
//      w = 0.05) the word's partial sums of d = w - w0 are 3..4 times those of w, and H_11 = base - |sum d| then cancels: 1e-7 on bse.
//      Where |beta_0 + beta_k - b0(null)| >= GLM_DIRECT_ROW the row sums w ITSELF (no base from the carrier sums: finish_lane); below it
//      the differences are the smaller terms and stay -- a variant carried by nearly everyone has a large beta_k but carriers whose eta
//      hardly moved, and its Schur complement cancels against the intercept: there only the differences are accurate enough (measured:
//      direct sums chosen by |beta_k| alone put 4e-6 on bse at allele frequency 0.985).  (Emulated in numpy before it was built:
//      differences max 7e-8 / direct 1.5e-8 on 150 strong rows.)
#define GLM_DIRECT_ROW 1.5
struct EtaConst { float c0h, dh, lo0, dl; };
__device__ __forceinline__ EtaConst eta_const(double b0, double b1)
{
    EtaConst c;
    const double c1 = b0 + b1;
    c.c0h = (float)b0;
    c.dh = (float)(c1 - (double)c.c0h);
    const float c1h = __builtin_fmaf(1.0f, c.dh, c.c0h);          // what pkfma(k = 1, dh, c0h) gives a carrier
    c.lo0 = (float)(b0 - (double)c.c0h);
    c.dl = (float)(c1 - (double)c1h) - c.lo0;
    return c;
}
__device__ __forceinline__ float weight_even(float eta)
{
    const float t = __builtin_amdgcn_exp2f(__builtin_fabsf(eta) * -1.4426950408889634f);
    const float m = __builtin_amdgcn_rcpf(1.0f + t);
    const float a = t * m;                                            // = min(mu, 1 - mu) <= 1/2: a - a^2 does not cancel, and is as insensitive to the
    return __builtin_fmaf(-a, a, a);                                  // error of the reciprocal near mu = 1/2 as mu - mu^2 is (most samples)
}
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
    const EtaConst ec = eta_const(beta[0], beta[1]);                        // (DELTA only: the constants of eta with their lo parts)
    const float vsub = fabs(beta[0] + beta[1] - P.warm[0]) >= GLM_DIRECT_ROW ? 0.0f : -1.0f;      // (DELTA only: the variant's row sums w - w0, or w itself)
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0[j] = v2f{0.0f, 0.0f}; hz1[j] = v2f{0.0f, 0.0f}; }
    const int nfull = N >> 1;
    auto pair = [&](const v2f (&rec)[RS], const float (&bz)[NCB], uint32_t two) {
        const v2f xb = {(float)(two & 1u), (float)(two >> 1)};
        v2f eta = DELTA ? pkfma(xb, v2f{ec.dl, ec.dl}, v2f{ec.lo0, ec.lo0}) : pkfma(xb, bf[1], bf[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = pkfma(bf[2 + j], rec[j], eta);
        if (DELTA) eta = eta + pkfma(xb, v2f{ec.dh, ec.dh}, v2f{ec.c0h, ec.c0h});
        v2f mu, wf;
        if (DELTA) { wf.x = weight_even(eta.x); wf.y = weight_even(eta.y); }
        else { mu.x = sigmoid_fast(eta.x); mu.y = sigmoid_fast(eta.y); wf = pkfma(-mu, mu, mu); }
        v2f d = wf;
        if (DELTA) d = wf - rec[Q + 1];
        else {
            const v2f r = rec[Q] - mu;
            gf[0] += r; gf[1] = pkfma(xb, r, gf[1]);
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j] = pkfma(r, rec[j], gf[2 + j]);
        }
        const v2f dx = xb * (DELTA ? pkfma(v2f{vsub, vsub}, rec[Q + 1], wf) : d);
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
        float eta = DELTA ? ec.lo0 + (xb ? ec.dl : 0.0f) : bf[0].x + (xb ? bf[1].x : 0.0f);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j].x, Wf[(int64_t)i * Q + j], eta);
        if (DELTA) eta += fmaf(xb ? 1.0f : 0.0f, ec.dh, ec.c0h);
        const float mu = sigmoid_fast(eta);
        const float wf = DELTA ? weight_even(eta) : mu * (1.0f - mu);
        float d = wf;
        if (DELTA) d = wf - P.w0f[i];
        else {
            const float r = P.yf[i] - mu;
            gs[0] += r; gs[1] += xb ? r : 0.0f;
#pragma unroll
            for (int j = 0; j < Q; ++j) gs[2 + j] = fmaf(r, Wf[(int64_t)i * Q + j], gs[2 + j]);
        }
        const float dx = xb ? (DELTA ? fmaf(vsub, P.w0f[i], wf) : d) : 0.0f;
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
#ifndef P16_ABL
#define P16_ABL 0
#endif
#ifndef P16_ABL_D
#define P16_ABL_D 0
#endif
template <int Q, bool DELTA>
__device__ __forceinline__ void pass32_pk_f16(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, const GlmParams &P, const float *__restrict__ Wf,
                                          const double (&beta)[Q + 2], float (&H)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], float *tr,
                                          int part = 0, int nparts = 1, double *hdl = nullptr)
{
    // hdl, part / nparts: as pass32_pk_f32 above
    constexpr int PC = Q + 2, NCB = FastCols<Q>::NCB, STRIDE = FastCols<Q>::STRIDE, NPART = DELTA ? 2 : 1;
    // Round 5: the table's spare columns (STRIDE - NPROD: 9 at Q = 10) hold z_j itself, so the intercept row's entries sum (w) z_j come out of the
    // same MFMAs as the block's (column NPROD + j), at the block's precision, instead of a packed fma per pair and entry on the vector ALU
    // (72 of a group's 528 vector instructions at Q = 10).  Entries j >= NZ0 stay on the vector ALU.
    constexpr int NPROD = FastCols<Q>::NPROD, NZ0 = (STRIDE - NPROD) < Q ? (STRIDE - NPROD) : Q;
    // LIN (the Newton-steering pass only; round 5): the variant's row of the Hessian (sum w k, sum w k z_j) is a GEMM against (1, z) as
    // well, and leaves the vector ALU: the group's w k as a second A operand against GlmParams.zl16 ((1, z) in columns 0..Q, B layout of the
    // products table, halves without a lo part -- the block's own precision), one more 32 x 32 accumulator tile per 32 variants.  88 of the
    // group's 455 vector instructions for 2 MFMAs, 8 conversions and 4 lane swaps.  The score stays on the vector ALU: through halves
    // (2^-11 per term) it sent a quarter of the variants into a second chord round and moved final estimates by 1e-5 relative (measured);
    // hi + lo operands for it would give back half of the gain.  The final information matrix (DELTA) keeps the variant's row on the
    // vector ALU too: its entries are sums of same-signed differences that need the per-word fp64 accumulation described above.
    constexpr bool LIN = !DELTA;
    static_assert(Q + 1 <= 32, "the linear block holds (1, z) in one 32-column block");
    const v4u *__restrict__ ZL = (const v4u *)P.zl16;                         // [group][64 lanes] x 16 bytes
    v16f accl[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) accl[h][r] = 0.0f;
    uint32_t Xh[LIN ? 8 : 1];                                               // this group's w k, one half2 per pair
    const int N = P.N;
    const int lane = threadIdx.x & 63, lh = lane >> 5, l31 = lane & 31;
    constexpr int RS = Q + 2;
    // timing ablations (results meaningless; profiles/r05/pass32_ablations.txt): 1 no MFMAs, 2 no table loads, 4 no transcendentals, 8 no score,
    // 16 every record load reads record 0 (always a scalar-cache hit)
    constexpr int ABL = DELTA ? P16_ABL_D : P16_ABL;
    // (constant address space: the touches below are volatile asm, which would otherwise take the "never clobbered" property from the records
    // and turn their scalar loads into per-lane vector loads -- firth_rounds.hip)
    typedef const __attribute__((address_space(4))) v2f *crec_t;
    const crec_t Rp = (crec_t)P.wfp;
    // Round 5: the pass waits for its record loads (1.85 ms against 1.35 with every load a scalar-cache hit, profiles/r05/pass32_ablations.txt):
    // a miss outlasts a pair's arithmetic, and scalar loads return out of order, so only s_waitcnt lgkmcnt(0) exists for them and a load
    // issued further ahead is waited for at the next pair anyway.  What CAN be taken off the pairs is the miss itself: once per 16-sample
    // group the cache lines of the NEXT group's eight records (64 (Q + 2) bytes = Q + 2 lines) are touched by loads into a register nobody
    // reads, all at once -- one exposed miss per group instead of one per pair, the records' own loads then hit.  (At most 15 scalar loads can
    // be outstanding: 12 touches + a record's two.)  The touched register must not be written by anything else while a touch can be in flight:
    // it is a variable that lives from the first touch to the end of the loops, and `make check-touch` (tools/check_touch_regs.py) verifies on the
    // device assembly of every instantiation that nothing else writes it inside that window.
    uint32_t touch_sink = 0;
    auto touch_group = [&](int grp, int lo, int hi) {
        // (the last group may be partial: the touched window is pulled back inside the table -- nfull >= 32 wherever the word loop runs)
        const char *base = (const char *)P.wfp + min((int64_t)grp * (64 * RS), (int64_t)(N >> 1) * (RS * 8) - 64 * RS);
#pragma unroll
        for (int l = 0; l < RS; ++l)
            if (l >= lo && l < hi) asm volatile("s_load_dword %0, %1, %2" : "+s"(touch_sink) : "s"(base), "n"(64 * l));
    };
    // (Q <= 10 only: with more covariates the two record buffers take 2 x 2 (Q + 2) of the ~100 scalar registers, the compiler spills and MOVES
    // the touched register -- `make check-touch` found that for Q = 14 -- and a touch in flight would land in whatever took its place)
    constexpr bool TOUCH = RS <= 12;
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
    const EtaConst ec = eta_const(beta[0], beta[1]);                        // (DELTA only: the constants of eta with their lo parts)
    const float vsub = fabs(beta[0] + beta[1] - P.warm[0]) >= GLM_DIRECT_ROW ? 0.0f : -1.0f;      // (DELTA only: the variant's row sums w - w0, or w itself)
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
        v2f eta = DELTA ? pkfma(xb, v2f{ec.dl, ec.dl}, v2f{ec.lo0, ec.lo0}) : pkfma(xb, bf[1], bf[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = pkfma(bf[2 + j], rec[j], eta);
        if (DELTA) eta = eta + pkfma(xb, v2f{ec.dh, ec.dh}, v2f{ec.c0h, ec.c0h});
        v2f mu, wf;
        if (ABL & 4) { mu = pkfma(eta, v2f{0.25f, 0.25f}, v2f{0.5f, 0.5f}); wf = pkfma(-mu, mu, mu); }
        else if (DELTA) { wf.x = weight_even(eta.x); wf.y = weight_even(eta.y); }
        else { mu.x = sigmoid_fast(eta.x); mu.y = sigmoid_fast(eta.y); wf = pkfma(-mu, mu, mu); }
        v2f d = wf;
        if (DELTA) d = wf - rec[Q + 1];
        else if (ABL & 8) { gf[0] += rec[Q] - mu; }
        else {
            const v2f r = rec[Q] - mu;
            gf[0] += r; gf[1] = pkfma(xb, r, gf[1]);
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j] = pkfma(r, rec[j], gf[2 + j]);
        }
        const v2f dx = xb * (DELTA ? pkfma(v2f{vsub, vsub}, rec[Q + 1], wf) : d);
        h00 += d;
        if constexpr (LIN) Xh[slot] = __builtin_bit_cast(uint32_t, __builtin_convertvector(dx, v2h));
        else h10 += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { if (j >= NZ0) hz0[j] = pkfma(d, rec[j], hz0[j]); if constexpr (!LIN) hz1[j] = pkfma(dx, rec[j], hz1[j]); }
        stash(slot, d);
    };
    auto fetch_rec = [&](int pr, v2f (&rec)[RS]) {
#pragma unroll
        for (int k = 0; k < RS; ++k) rec[k] = Rp[(int64_t)((ABL & 16) ? 0 : pr) * RS + k];
    };
    auto fetch_bz = [&](int grp, v4u (&bz)[NCB][NPART]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q2 = 0; q2 < NPART; ++q2) bz[cb][q2] = (ABL & 2) ? v4u{(uint32_t)grp, 0u, 0u, 0u} : Z16[(((int64_t)grp * NCB + cb) * 2 + q2) * 64 + lane];
    };
    auto fetch_bl = [&](int grp, v4u &bl) { if constexpr (LIN) bl = ZL[(int64_t)grp * 64 + lane]; };
    // the group's w k against the linear block (operands swapped into place as the weights' are)
    auto flush_lin = [&](const v4u &bl) {
        if constexpr (LIN) {
            v4u x0, x1;
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const auto sw = __builtin_amdgcn_permlane32_swap(Xh[q2], Xh[4 + q2], false, false);
                x0[q2] = sw[0]; x1[q2] = sw[1];
            }
            const v8h B = __builtin_bit_cast(v8h, bl);
            accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, x0), B, accl[0], 0, 0, 0);
            accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, x1), B, accl[1], 0, 0, 0);
        }
    };
    // the 16 samples stashed in Ah / Al against the group's product columns
    auto flush = [&](const v4u (&bz)[NCB][NPART]) {
        if (ABL & 1) { acc[0][0][0] += __builtin_bit_cast(float, Ah[0] ^ Ah[7] ^ bz[0][0][0] ^ bz[NCB - 1][NPART - 1][0]) + (DELTA ? __builtin_bit_cast(float, Al[0]) : 0.0f); return; }
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
    v4u zc[NCB][NPART], zn[NCB][NPART], lc, ln2;
    const int nwords = nfull >> 5;                                            // whole words: pipelined
    const int plast = nfull > 0 ? nfull - 1 : 0;
    const int glast = (N + 15) / 16 - 1;
    const int wd0 = (int)((int64_t)nwords * part / nparts), wd1 = (int)((int64_t)nwords * (part + 1) / nparts);
    const bool tail = part == nparts - 1;
    fetch_rec(min(wd0 * 32, plast), ra); fetch_bz(min(wd0 * 4, glast), zc); fetch_bl(min(wd0 * 4, glast), lc);
    uint64_t w = T[(int64_t)min(wd0, (N - 1) >> 6) * Vpad + v];
    for (int wd = wd0; wd < wd1; ++wd) {
        const uint64_t wn = T[(int64_t)min(wd + 1, (N - 1) >> 6) * Vpad + v];
#pragma unroll 1
        for (int g4 = 0; g4 < 4; ++g4) {
            const int grp = wd * 4 + g4;
            fetch_bz(min(grp + 1, glast), zn); fetch_bl(min(grp + 1, glast), ln2);
            const uint32_t wbits = (uint32_t)(w >> (16 * g4)) & 0xFFFFu;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const int pr = grp * 8 + k;
                fetch_rec(pr + 1 + pipe_zero(ra[0].x), rb);
                // (the touches go out BEHIND the wait for this pair's record and the request for the next one: issued in front of that wait
                // they were waited for at once -- a whole miss exposed per group instead of a miss less a pair's arithmetic)
                if (TOUCH && k == 0 && !(ABL & 32)) touch_group(min(grp + 1, (plast >> 3)), 0, RS);
                pair(ra, k, (wbits >> (2 * k)) & 3u);
                fetch_rec(min(pr + 2, plast) + pipe_zero(rb[0].x), ra);
                pair(rb, k + 1, (wbits >> (2 * k + 2)) & 3u);
            }
            flush(zc); flush_lin(lc);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q2 = 0; q2 < NPART; ++q2) zc[cb][q2] = zn[cb][q2];
            lc = ln2;
        }
        w = wn;
        if (DELTA && hdl) {
            hdl[lane] += (double)(h00.x + h00.y); hdl[64 + lane] += (double)(h10.x + h10.y);
            h00 = v2f{0.0f, 0.0f}; h10 = v2f{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                if (j >= NZ0) { hdl[(2 + j) * 64 + lane] += (double)(hz0[j].x + hz0[j].y); hz0[j] = v2f{0.0f, 0.0f}; }
                hdl[(2 + Q + j) * 64 + lane] += (double)(hz1[j].x + hz1[j].y); hz1[j] = v2f{0.0f, 0.0f};
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
        float eta = DELTA ? ec.lo0 + (odd_x ? ec.dl : 0.0f) : bf[0].x + (odd_x ? bf[1].x : 0.0f);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j].x, Wf[(int64_t)i * Q + j], eta);
        if (DELTA) eta += fmaf(odd_x ? 1.0f : 0.0f, ec.dh, ec.c0h);
        const float mu = sigmoid_fast(eta);
        const float wf = DELTA ? weight_even(eta) : mu * (1.0f - mu);
        odd_d = DELTA ? wf - P.w0f[i] : wf;
        if (!DELTA) {
            const float r = P.yf[i] - mu;
            gf[0].x += r; gf[1].x += odd_x ? r : 0.0f;
#pragma unroll
            for (int j = 0; j < Q; ++j) gf[2 + j].x = fmaf(r, Wf[(int64_t)i * Q + j], gf[2 + j].x);
        }
        const float dx = odd_x ? (DELTA ? fmaf(vsub, P.w0f[i], wf) : odd_d) : 0.0f;
        h00.x += odd_d; if constexpr (!LIN) h10.x += dx;
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float zj = Wf[(int64_t)i * Q + j]; if (j >= NZ0) hz0[j].x = fmaf(odd_d, zj, hz0[j].x); if constexpr (!LIN) hz1[j].x = fmaf(dx, zj, hz1[j].x); }
    }
    if (tail) {
        const int pend = nfull + (N & 1);                                     // pair slots in use, the odd sample's included
        for (int p0 = nwords * 32; p0 < pend; p0 += 8) {
            if (p0 == nwords * 32) w = T[(int64_t)min(p0 >> 5, (N - 1) >> 6) * Vpad + v];
            fetch_bz(min(p0 >> 3, glast), zc); fetch_bl(min(p0 >> 3, glast), lc);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pr = p0 + k;
                if (pr < nfull) { fetch_rec(pr, ra); pair(ra, k, (uint32_t)(w >> (2 * (pr & 31))) & 3u); }
                else {
                    const bool odd = pr == nfull && (N & 1);
                    stash(k, v2f{odd ? odd_d : 0.0f, 0.0f});
                    if constexpr (LIN) Xh[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{odd && odd_x ? odd_d : 0.0f, 0.0f}, v2h));
                }
            }
            flush(zc); flush_lin(lc);
        }
    }
    h00s = h00.x + h00.y; h10s = h10.x + h10.y;
#pragma unroll
    for (int a = 0; a < PC; ++a) gs[a] = gf[a].x + gf[a].y;
#pragma unroll
    for (int j = 0; j < Q; ++j) { hz0s[j] = hz0[j].x + hz0[j].y; hz1s[j] = hz1[j].x + hz1[j].y; }
    if (DELTA && hdl) {
        h00s = (float)(hdl[lane] + (double)h00s); h10s = (float)(hdl[64 + lane] + (double)h10s);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            if (j >= NZ0) hz0s[j] = (float)(hdl[(2 + j) * 64 + lane] + (double)hz0s[j]);
            hz1s[j] = (float)(hdl[(2 + Q + j) * 64 + lane] + (double)hz1s[j]);
        }
    }
    H[sidx(0, 0)] = h00s; H[sidx(1, 0)] = h10s; H[sidx(1, 1)] = h10s;
#pragma unroll
    for (int j = 0; j < Q; ++j) { H[sidx(2 + j, 0)] = hz0s[j]; H[sidx(2 + j, 1)] = hz1s[j]; }
    asm volatile("" ::"s"(touch_sink));                                   // (the touches' register stays reserved until here)
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
#pragma unroll
            for (int j = 0; j < NZ0; ++j) H[sidx(2 + j, 0)] = row[NPROD + j];
        }
    }
    if constexpr (LIN) {                                                     // the linear tile: columns 0..Q = the variant's row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * (STRIDE + 1) + l31] = accl[h][r];
            __syncthreads();
            if (lh == h) {
                const float *row = tr + l31 * (STRIDE + 1);
                H[sidx(1, 0)] = row[0]; H[sidx(1, 1)] = row[0];
#pragma unroll
                for (int j = 0; j < Q; ++j) H[sidx(2 + j, 1)] = row[1 + j];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) g[a] = (double)gs[a];
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

// One single-precision Newton step's solve for one variant (k_glm_solve32<Q, false>, and k_glm_pass32<Q, true> which runs it in its own epilogue:
// the Hessian and the score then never leave the lane -- 408 bytes per variant less to write and read back, one launch less per round): factor
// H / n - ridge, solve, beta += step, and where the variant goes next.
template <int Q>
__device__ __forceinline__ void solve32_lane(int64_t v, int64_t Vpad, const GlmParams &P, double (&A)[(Q + 2) * (Q + 3) / 2], double (&g)[Q + 2], int last_round,
                                             bool &go_next, bool &go_chord, bool &go_slow)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
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
            for (int a = 0; a < NH; ++a) ((float *)P.ch_fac)[(int64_t)a * Vpad + v] = (float)A[a];     // (single precision: the factor of a single-precision Hessian; half the chord rounds' traffic)
            P.ch_rho[v] = (float)fmin(0.5, fmax(8.0 * stp, 1e-4));
        }
        else if (last_round) go_slow = true;
        else go_next = true;
    }
}

template <int Q, bool PK, bool FUSE = false>
__global__ __launch_bounds__(64, PK ? 2 : 3) void k_glm_pass32(const uint64_t *__restrict__ T, int64_t Vpad, const double *__restrict__ y,
                                                      const float *__restrict__ Wf, GlmParams P, const int *__restrict__ list,
                                                      const int *__restrict__ cnt, GlmWork wk = GlmWork{}, int *__restrict__ next = nullptr,
                                                      int *__restrict__ next_cnt = nullptr, int *__restrict__ chord = nullptr,
                                                      int *__restrict__ chord_cnt = nullptr, int last_round = 0)
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
    if constexpr (FUSE) {                                            // the round's solve, here (solve32_lane): H and g stay in the lane
        bool go_next = false, go_chord = false, go_slow = false;
        if (on) {
            const double nobs = (double)P.N;
            double A[NH];
#pragma unroll
            for (int a = 0; a < NH; ++a) A[a] = (double)Hf[a] / nobs;
#pragma unroll
            for (int a = 0; a < PC; ++a) { A[sidx(a, a)] -= 1e-10; g[a] = g[a] / nobs; }
            solve32_lane<Q>(v, Vpad, P, A, g, last_round, go_next, go_chord, go_slow);
        }
        list_push(go_next, next, next_cnt, (int)v);
        list_push(go_chord, chord, chord_cnt, (int)v);
        list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
        return;
    }
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
        solve32_lane<Q>(v, Vpad, P, A, g, last_round, go_next, go_chord, go_slow);
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
        for (int a = 0; a < NH; ++a) A[a] = (double)((const float *)P.ch_fac)[(int64_t)a * Vpad + v];
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
    // (touching the next samples' cache lines ahead, as pass32_pk_f16 does for its records, does not pay here: 2.72 -> 2.83 ms; four wavefronts per
    // SIMD already cover the scalar loads' misses)
    double beta[PC], g[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) { beta[a] = P.ch_bs[(int64_t)a * Vpad + v]; g[a] = 0.0; }
    // y is 0/1 here (sh_glm_setup turns the finishing kernels off otherwise): ll_i = -softplus(a_i), a_i = (1 - 2 y_i) eta_i (= log mu_i for
    // y = 1, log(1 - mu_i) for y = 0: SM Logit.loglike), and with t = exp(-|eta|):  softplus(a) = max(a, 0) + log(1 + t).  The logs are not
    // taken one by one: the factors 1 + t in (1, 2] are MULTIPLIED (one rounding each, like a sum's), renormalised once per 64-sample word,
    // and one log at the end turns the product into the sum.  One exp (argument reduced, degree-11 polynomial: 6e-15 relative, v_ldexp) and
    // one reciprocal per sample, no log.  The sample's record is fetched one sample ahead.
    // Round 5, 68.5 -> 54.5 vector instructions per sample (the kernel runs at the fp64 issue rate, 4.8 cycles per instruction at four
    // wavefronts per SIMD): the variant's bit enters as a double (v_bfe, v_cvt: eta and g_1 by one fma each instead of compare + two selects
    // + add); max(s eta, 0) = (|eta| + s eta) / 2 summed as its two halves (an add with the |.| modifier and an fma); mu - 1/2 =
    // sign(eta) (1 - t) / (2 (1 + t)) instead of a compare, a product and two selects; s = 1 - 2 y and y - 1/2 are made from y's bits on the
    // scalar unit (the record is wave-uniform); the polynomial's leading coefficient sits in a register (two scalar operands in one fma
    // cost a move).
    double maxdev = 0.0, abs_eta = 0.0, s_eta = 0.0, prod = 1.0;
    int pexp = 0;
    double ctop = 2.505210838544172e-08;                             // 1/11!
    asm("" : "+v"(ctop));                                       // (not volatile: a volatile asm counts as a possible store, and the records' loads stop being scalar)
    auto one = [&](const double (&rc)[RS], uint32_t bit) {
        const double xd = (double)bit;
        double eta = fma(beta[1], xd, beta[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        // (wave-uniform: y = 1.0 is 0x3ff00000 in the high word -- bit 29 is its highest -- and y = 0 is 0)
        const uint32_t shi = 0x3ff00000u | (((uint32_t)__double2hiint(rc[Q]) << 2) & 0x80000000u);
        const double sg = __hiloint2double((int)shi, 0), ymh = __hiloint2double((int)(shi ^ 0x80100000u), 0);   // 1 - 2 y;  y - 1/2
        double t;
        {
            const double u0 = -fmin(fabs(eta), 800.0);
            // (k = round(u0 log2 e) by the magic constant 1.5 x 2^52: the sum's low word IS k as an integer -- one fma and one subtraction
            // instead of multiply, v_rndne, v_cvt_i32)
            const double kfm = fma(u0, 1.4426950408889634074, 6755399441055744.0);
            const double kf = kfm - 6755399441055744.0;
            double r = fma(kf, -6.93147180369123816490e-01, u0);
            r = fma(kf, -1.90821492927058770002e-10, r);
            double p = fma(ctop, r, 2.755731922398589e-07);          // 1/10!
            p = fma(p, r, 2.7557319223985893e-06);                   // 1/9!
            p = fma(p, r, 2.48015873015873e-05);                     // 1/8!
            p = fma(p, r, 1.984126984126984e-04);                    // 1/7!
            p = fma(p, r, 1.3888888888888889e-03);                   // 1/6!
            p = fma(p, r, 8.333333333333333e-03);                    // 1/5!
            p = fma(p, r, 4.1666666666666664e-02);                   // 1/4!
            p = fma(p, r, 1.6666666666666666e-01);                   // 1/3!
            p = fma(p, r, 0.5);
            p = fma(p, r, 1.0);
            p = fma(p, r, 1.0);
            t = ldexp(p, __double2loint(kfm));
        }
        const double u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);                                      // one Newton step: 2.2e-15 relative (tools/ubench/exp_rcp_acc.hip), against 6e-15 in t
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        const double hm = copysign(fma(-0.5, t, 0.5) * inv, eta);                  // mu - 1/2
        const double r = ymh - hm;                                                 // y - mu
        abs_eta += fabs(eta);
        s_eta = fma(sg, eta, s_eta);
        prod *= u;
        maxdev = fmax(maxdev, fabs(r));
        g[0] += r; g[1] = fma(xd, r, g[1]);
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
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) {                         // (32-bit halves: the bit is one v_bfe_u32 with the position in a scalar register)
                const uint32_t wh = hf ? (uint32_t)(w >> 32) : (uint32_t)w;
#pragma unroll 1
                for (int b = 0; b < 32; b += 2) {
                    const int i = wd * 64 + hf * 32 + b, i2 = min(i + 2, N - 1);
                    const int za = pipe_zero(ra[0]);
#pragma unroll
                    for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
                    __builtin_amdgcn_sched_barrier(0);
                    one(ra, __builtin_amdgcn_ubfe(wh, (uint32_t)b, 1u));
                    // (the fences keep the two samples' instructions apart: interleaved, the odd sample's record is wanted five instructions
                    // behind its load instead of a whole sample behind it -- measured 3.05 ms against 2.6)
                    __builtin_amdgcn_sched_barrier(0);
                    const int zb = pipe_zero(rb[0]);
#pragma unroll
                    for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
                    __builtin_amdgcn_sched_barrier(0);
                    one(rb, __builtin_amdgcn_ubfe(wh, (uint32_t)(b + 1), 1u));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int b = 0; b < lim; ++b) {
#pragma unroll
                for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(wd * 64 + b) * RS + k];
                one(ra, (uint32_t)(w >> b) & 1u);
            }
        }
        int e2; prod = frexp(prod, &e2); pexp += e2;
    }
    if (!on) return;
#pragma unroll
    for (int a = 0; a < PC; ++a) P.ch_g[(int64_t)a * Vpad + v] = g[a];
    P.ch_md[v] = maxdev; P.ch_ll[v] = -(0.5 * (abs_eta + s_eta) + fma((double)pexp, 0.6931471805599453, log(prod)));
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
    const EtaConst ec = eta_const(P.ch_bs[v], P.ch_bs[Vpad + v]);
    const float vsub = fabs(P.ch_bs[v] + P.ch_bs[Vpad + v] - P.warm[0]) >= GLM_DIRECT_ROW ? 0.0f : -1.0f;
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
        float eta = ec.lo0 + (xb ? ec.dl : 0.0f);                            // (eta_const / weight_even: as the packed pass)
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bf[2 + j], Wf[(int64_t)i * Q + j], eta);
        eta += fmaf(xb ? 1.0f : 0.0f, ec.dh, ec.c0h);
        const float wf = weight_even(eta), d = wf - (float)w0[i];
        const float dx = xb ? fmaf(vsub, (float)w0[i], wf) : 0.0f;
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

// What k_glm_finish does for one variant: assemble the information matrix (the null model's block and the carrier sums, exact, + the
// single-precision differences hf(a) of k_glm_dpass), factor it, bse, the exact Newton step, the decisions and the output row.  Also run by
// k_glm_dpass_pk<Q, FIN> in its own epilogue, where the differences never leave the lane (312 bytes per variant less to write and read back).
template <int Q, bool DIRECT, class HF>
__device__ __forceinline__ void finish_lane(int64_t v, int64_t Vpad, int64_t V, const GlmParams &P, HF hf, double *__restrict__ out, uint32_t *__restrict__ flags,
                                            int *__restrict__ firth_list, int *__restrict__ firth_count, bool &go_slow)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    const double nobs = (double)P.N;
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
            // (the variant's row: the carrier sums at the null model + the pass' differences -- or, for a strong effect, the pass' sums of w
            // itself: GLM_DIRECT_ROW, decided from the same beta the pass read)
            const bool direct = fabs(beta[0] + beta[1] - P.warm[0]) >= GLM_DIRECT_ROW;
            H[sidx(0, 0)] = P.null_h[0]; H[sidx(1, 0)] = H[sidx(1, 1)] = direct ? 0.0 : P.ch_bd[v];
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                H[sidx(2 + j, 0)] = P.null_h[1 + j]; H[sidx(2 + j, 1)] = direct ? 0.0 : P.ch_bd[(int64_t)(1 + j) * Vpad + v];
#pragma unroll
                for (int k = 0; k <= j; ++k) H[sidx(2 + j, 2 + k)] = P.a0[j * (j + 1) / 2 + k];
            }
#pragma unroll
            for (int a = 0; a < NH; ++a) H[a] = (H[a] + (double)hf(a)) / nobs;
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

template <int Q, int FIN = 0>                                       // FIN: 1 = k_glm_finish<Q, true> in the epilogue, 2 = k_glm_finish<Q, false>
__global__ __launch_bounds__(64, 2) void k_glm_dpass_pk(const uint64_t *__restrict__ T, int64_t Vpad, const float *__restrict__ Wf, GlmParams P,
                                                        const int *__restrict__ list, const int *__restrict__ cnt, int64_t V = 0, GlmWork wk = GlmWork{},
                                                        double *__restrict__ out = nullptr, uint32_t *__restrict__ flags = nullptr,
                                                        int *__restrict__ firth_list = nullptr, int *__restrict__ firth_count = nullptr)
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
    if constexpr (FIN != 0) {
        bool go_slow = false;
        if (on) finish_lane<Q, FIN == 1>(v, Vpad, V, P, [&](int a) { return H[a]; }, out, flags, firth_list, firth_count, go_slow);
        list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
        return;
    }
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
    bool go_slow = false;
    if (on) finish_lane<Q, DIRECT>(v, Vpad, V, P, [&](int a) { return P.ch_hf[(int64_t)a * Vpad + v]; }, out, flags, firth_list, firth_count, go_slow);
    list_push(go_slow, wk.slow_list, wk.slow_count, (int)v);
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
                else if (P.wfp) {
                    // (the round's solve in the pass' own epilogue: k_glm_pass32<Q, true, true>)
                    hipLaunchKernelGGL((k_glm_pass32<Q, true, true>), grid, blk, 0, st, T, Vpad, y, Wf, P, P.ch_list[r & 1], P.ch_cnt + r, wk,
                                       P.ch_list[(r + 1) & 1], P.ch_cnt + r + 1, P.ch_list[2], cc, r == n32 - 1 ? 1 : 0);
                    continue;
                }
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
                // (with the pair table the finishing runs in the pass' own epilogue: k_glm_dpass_pk<Q, 1>)
                if (P.wfp) hipLaunchKernelGGL((k_glm_dpass_pk<Q, 1>), grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[5], P.ch_cnt + 29, V, wk, out, flags, flist, fcount);
                else {
                    hipLaunchKernelGGL(k_glm_dpass<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[5], P.ch_cnt + 29);
                    hipLaunchKernelGGL((k_glm_finish<Q, true>), grid, blk, 0, st, Vpad, V, P, wk, P.ch_list[5], P.ch_cnt + 29, out, flags, flist, fcount);
                }
            }
            if (P.fin_rounds) {
                hipLaunchKernelGGL(k_glm_ll<Q>, g256, b256, 0, st, T, Vpad, y, P, P.ch_list[4], P.ch_cnt + 30);
                if (P.wfp) hipLaunchKernelGGL((k_glm_dpass_pk<Q, 2>), grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[4], P.ch_cnt + 30, V, wk, out, flags, flist, fcount);
                else {
                    hipLaunchKernelGGL(k_glm_dpass<Q>, grid, blk, 0, st, T, Vpad, Wf, P, P.ch_list[4], P.ch_cnt + 30);
                    hipLaunchKernelGGL((k_glm_finish<Q, false>), grid, blk, 0, st, Vpad, V, P, wk, P.ch_list[4], P.ch_cnt + 30, out, flags, flist, fcount);
                }
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
    else if (which == 3) return shk_glm_firth_pinv_launch(st, Q, T, Vpad, V, y, W, P, plist, pcount, out, flags);          // glm_firth_v1.hip
    else if (which == 6 || which == 2) return shk_glm_ols_launch(st, Q, which, T, Vpad, V, y, W, y1, y0, yc, ZtZ, Zty, P, out, flags, plist, pcount);   // glm_ols.hip
    else return hipErrorInvalidValue;
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
