// firth_rounds.hip -- the two sample passes of fit_firth (pyseer/model.py:414-504) as round kernels for gfx950, y in {0, 1}.
//
// Reference path restated here:
//   pyseer/model.py:397-411  firth_likelihood      F(b) = -(loglike(b) + 0.5 log det X^T W X)
//   pyseer/model.py:440-479  fit_firth             hat diagonal through (X^T W X)^-1, penalised score, Newton step, step halving
//
// Mapping (as glm_kernels.hip): ONE VARIANT PER LANE, S wavefronts of a block share the same 64 variants and split the samples by
// 64-sample word; a sample's covariates arrive wave-uniform through the scalar cache.  Round 3 rewrote both passes around two findings
// of the round-2 profile (profiles/r02/rocprofv3_pmc_summary_C4.csv): the passes are VALU-bound (71 % / 87 % VALU busy at two
// wavefronts per SIMD), so what counts is the number of VALU instructions per sample, and the 78 fp64 accumulators of X^T W X leave
// no room for anything else in 256 VGPRs (342 spilled registers, 54 GB of scratch traffic per C4 batch).
//   k_firth_eval2   information matrix + log-likelihood at the candidate.  Per sample: the bit as v_bfe_u32 + v_cvt_f64_u32 (no 64-bit
//                   shift / compare / select chain), no mu and no max |y - mu| (fit_firth never looks at them), the record holds
//                   s = 1 - 2y so that the likelihood term is max(s eta, 0); the variant's word is fetched one word ahead.
//   k_firth_step2   penalised score.  The hat diagonal h_i = w_i x_i^T (X^T W X)^-1 x_i is invariant under a change of basis of the
//                   design, and the score only needs it to ~1e-6 relative (it multiplies (1/2 - mu_i) and sums to p = q + 2 over all
//                   samples, against a score whose other part sums N terms of order 1): it is evaluated in SINGLE precision, two samples
//                   per v_pk_fma_f32, in the standardised covariate basis (where single precision is safe whatever the scale of the
//                   user's columns).  eta, mu, the residual and the score sums stay fp64.
#include "glm_device.h"

typedef float v2f __attribute__((ext_vector_type(2)));
// pipe_zero (glm_device.h) that stays where it is written: the plain form has no side effects and may be hoisted to the load it reads from
__device__ __forceinline__ int pipe_zero_v(double x) { int z; asm volatile("s_and_b32 %0, %1, 0" : "=s"(z) : "s"(__double2loint(x)) : "scc"); return z; }
__device__ __forceinline__ v2f pkfma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

#ifndef FIRTH_STEP2_AHEAD
#define FIRTH_STEP2_AHEAD 1          /* the pairs' covariates fetched one pair ahead (A/B: 6.35 -> 5.86 ms per 259 k variants); 0 = at the top of the pair */
#endif
#ifndef FIRTH_STEP2_ABL
#define FIRTH_STEP2_ABL 0          /* timing ablations of k_firth_step2 (results meaningless): 1 = no hat diagonal, 2 = no LDS reads, 3 = no exp */
#endif
#ifndef FIRTH_HAT32_PIVOT
#define FIRTH_HAT32_PIVOT 1e-2     /* smallest pivot of the standardised factor, relative to its own diagonal entry, for the single-precision hat diagonal */
#endif

// ---- start of the rounds (force_firth at large N, GlmParams.firth_warm): the null-model fit plus ONE Newton step of the ordinary likelihood -------
// fit_firth's fixed point does not depend on where the iteration starts, and at N >= 4096 neither does where it stops, to 1e-7 (sh_glm_setup).
// Round 2 started at the null-model fit [b0, 0, bz]; there eta is the null model's, so the score and the variant's row of X^T W X are sums of
// per-run vectors over the variant's carriers (k_glm_bitdot, GlmParams.ch_bd) and the first Newton step of the ORDINARY likelihood costs a
// 12 x 12 solve per variant instead of a pass over the samples.  Starting the Firth iteration one step closer saves a whole round (a
// likelihood pass and a score pass: 12 of 55 ms per C4 batch).  The step is taken in the standardised coordinates of the logistic kernels
// (warm, null_h, null_g, a0) and mapped back to the columns as given; anything not finite keeps the round-2 start.  A warm slot is marked by
// the complemented variant index in fw.var; one that is still iterating after FIRTH_WARM_LIMIT accepted steps restarts from the reference's
// vector (k_firth_eval2).
template <int Q>
__global__ __launch_bounds__(64) void k_firth_init2(const int *__restrict__ firth_list, const int *__restrict__ firth_count, GlmParams P, int64_t Vpad,
                                                    FirthWork fw, int *__restrict__ eval_list, int *__restrict__ eval_count)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
    const int cnt = *firth_count;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s == 0) *eval_count = cnt;
    if (s >= cnt) return;
    const int v = firth_list[s];
    const bool warm = P.firth_warm != 0;
    double b0[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) b0[a] = warm ? ((a == 0) ? P.fwarm[0] : (a == 1) ? 0.0 : P.fwarm[a - 1]) : ((a == 0) ? P.ymean_logit : 0.0);
    if (warm && P.firth_warm >= 2) {
        const double nobs = (double)P.N;
        double A[NH], g[PC], det;
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
        for (int a = 0; a < PC; ++a) g[a] = g[a] / nobs;
        if (ldl_factor<PC>(A, 1e-6, &det)) {
            ldl_solve<PC>(A, g);
            double bs[PC], bo[PC];
            bs[0] = P.warm[0] + g[0]; bs[1] = g[1];
#pragma unroll
            for (int j = 0; j < Q; ++j) bs[2 + j] = P.warm[1 + j] + g[2 + j];
            bo[0] = bs[0]; bo[1] = bs[1];
            bool fin = isfinite(bs[0]) && isfinite(bs[1]) && fabs(bs[1]) < 30.0;
#pragma unroll
            for (int j = 0; j < Q; ++j) { bo[2 + j] = bs[2 + j] / P.wstd[Q + j]; bo[0] = fma(-bo[2 + j], P.wstd[j], bo[0]); fin = fin && isfinite(bo[2 + j]); }
            if (fin && isfinite(bo[0])) {
#pragma unroll
                for (int a = 0; a < PC; ++a) b0[a] = bo[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) {
        fw.st[(int64_t)(fw_beta<PC>() + a) * fw.cap + s] = b0[a];
        fw.st[(int64_t)(fw_cand<PC>() + a) * fw.cap + s] = b0[a];
    }
    fw.st[(int64_t)fw_snp<PC>() * fw.cap + s] = INFINITY;
    fw.iter[s] = -1; fw.halv[s] = 0; fw.var[s] = warm ? ~v : v;
    eval_list[s] = s;
}

// ---- information matrix (packed lower) and log-likelihood at beta; R: per sample Q covariates (as given) then s = 1 - 2 y ----------
template <int Q>
__device__ __forceinline__ void firth_info_pass(const uint64_t *__restrict__ T, int64_t Vpad, int64_t v, int N, int NB64,
                                                const double *__restrict__ R, const double (&beta)[Q + 2],
                                                double (&H)[(Q + 2) * (Q + 3) / 2], double &ll, int sb0, int sbs)
{
    constexpr int P = Q + 2, RS = Q + 1;
#pragma unroll
    for (int a = 0; a < P * (P + 1) / 2; ++a) H[a] = 0.0;
    double apos = 0.0, prod = 1.0;
    int pexp = 0;
    auto one = [&](const double (&rc)[RS], uint32_t bit) {
        const double xd = (double)bit;
        double eta = fma(beta[1], xd, beta[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double t = exp_neg(fabs(eta)), u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        const double wgt = (t * inv) * inv;                                      // mu (1 - mu) = t / (1 + t)^2
        apos += fmax(rc[Q] * eta, 0.0);                                          // -log-likelihood term = max(s eta, 0) + log(1 + t)
        prod *= u;
        const double wx = wgt * xd;
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
    // One loop body for every word: pairs of samples, the record of the next sample fetched while this one is computed (scalar loads,
    // ordered by pipe_zero).  A partial last word runs fewer pairs; the odd last sample of an odd N is taken after the word loop (a third
    // copy of the sample's code INSIDE the word loop made the register allocator shuffle ~90 registers through scratch per word).
    double ra[RS], rb[RS];
    {
        const int i0 = min(sb0 * 64, N - 1);
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i0 * RS + k];
    }
    uint64_t wnext = T[(int64_t)min(sb0, NB64 - 1) * Vpad + v];
    uint64_t w64 = 0;
    for (int sb = sb0; sb < NB64; sb += sbs) {                                 // (sb0, sbs) = (wave, waves) of a sample-split block
        w64 = wnext;
        wnext = T[(int64_t)min(sb + sbs, NB64 - 1) * Vpad + v];                // the next word of this wavefront, one word ahead
        const int npair = min(64, N - sb * 64) >> 1;
        const int inext = min((sb + sbs) * 64, N - 1);                        // first sample of this wavefront's next word
        uint32_t wh = (uint32_t)w64;
#pragma unroll 1
        for (int pr = 0; pr < npair; ++pr) {
            if (pr == 16) wh = (uint32_t)(w64 >> 32);
            const int i = sb * 64 + 2 * pr, i2 = (pr == npair - 1) ? inext : i + 2;
            const int sh = (2 * pr) & 31;
            const int za = pipe_zero(ra[0]);
#pragma unroll
            for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
            one(ra, (wh >> sh) & 1u);
            const int zb = pipe_zero(rb[0]);
#pragma unroll
            for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
            one(rb, (wh >> (sh + 1)) & 1u);
        }
        int e2; prod = frexp(prod, &e2); pexp += e2;
    }
    if ((N & 1) && ((NB64 - 1 - sb0) % sbs) == 0 && sb0 < NB64) {              // the odd last sample: in the last word, which this wavefront walked
        const int i = N - 1;
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i * RS + k];
        one(ra, (uint32_t)(w64 >> (i & 63)) & 1u);
    }
    H[sidx(1, 1)] = H[sidx(1, 0)];
    ll = -(apos + fma((double)pexp, 0.6931471805599453, log(prod)));
}

// penalised likelihood at cand; accept / halve / converge / fail (model.py:465-484)
template <int Q>
__global__ __launch_bounds__(512) void k_firth_eval2(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, GlmParams P,
                                                     FirthWork fw, const int *__restrict__ eval_list, const int *__restrict__ eval_count,
                                                     int *__restrict__ next_eval, int *__restrict__ next_eval_count,
                                                     int *__restrict__ step_list, int *__restrict__ step_count,
                                                     double *__restrict__ out, uint32_t *__restrict__ flags,
                                                     int *__restrict__ pinv_list, int *__restrict__ pinv_count)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2;
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
    double cand[PC], A[NH];
#pragma unroll
    for (int a = 0; a < PC; ++a) cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s];
    double ll, det;
    firth_info_pass<Q>(T, Vpad, v, P.N, P.NB64, P.rec_f, cand, A, ll, xw.w, xw.S);
    double llx[1] = {ll};
    xw_sum(xw, A); xw_sum(xw, llx); ll = llx[0];
    if (!live || xw.w != 0) return;
    const double i11c = A[sidx(1, 1)];
    // diagonal of the information matrix in the standardised basis (up to the factor 1 / s_j^2 it shares with the pivot D_s[2+j]), for the pivot
    // guard of the single-precision hat diagonal below
    double hdiag[Q > 0 ? Q : 1];
    {
        const double h00 = A[sidx(0, 0)];
#pragma unroll
        for (int j = 0; j < Q; ++j) { const double m = P.wstd[j]; hdiag[j] = fma(m, fma(m, h00, -2.0 * A[sidx(2 + j, 0)]), A[sidx(2 + j, 2 + j)]); }
    }
    const bool singular = !ldl_factor<PC>(A, SING_TOL, &det);
    if (singular) {                     // handled by k_glm_firth_pinv (numpy.linalg.pinv semantics, model.py:450)
        list_push(true, pinv_list, pinv_count, (int)v);
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
        // default: the two noise rules (GlmParams.firth_noise / firth_accept); SEERHIP_ROUTE firth_literal=1 sets both to 0 = the reference's
        // literal F(new) > F(old)
        if (Fcand > Fcur + P.firth_noise * fabs(Fcur) && !(stepmax < P.firth_accept)) {   // step halving, model.py:467-474
            accept = false;
            const int h = fw.halv[s] + 1;
            fw.halv[s] = h;
            // new_beta = beta + 0.5 (new_beta - beta).  Once that map returns its argument bit for bit (the candidate sits one rounding
            // away from beta: beta + 0.5 ulp rounds back up), every later comparison is this one again: the reference walks on to
            // j > step_limit and returns None (model.py:471-473).  Same verdict, without the walk.
            bool moved = false;
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
                // With the literal rule the LAST step of most fits is compared inside the rounding noise of F (|dF| ~ N step^2 with a step of
                // ~1e-7), so about half of them are halved once, a quarter twice, ...: the first few halvings are ordinary rounds with long
                // lists.  After firth_halv_handoff of them the lists are short (a round costs a pass' latency whatever its length) and the
                // variant is finished by one workgroup, which re-evaluates F(beta) its own way before it compares anything.
                if (h >= P.firth_halv_handoff) list_push(true, fw.blk_list, fw.blk_count, s);
                else list_push(true, next_eval, next_eval_count, s);
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
        list_push(true, next_eval, next_eval_count, s);
        return;
    }
    if (accept && !failed && !conv) {                                // beta <- cand; keep the factor for the score pass
#pragma unroll
        for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s] = cand[a];
#pragma unroll
        for (int a = 0; a < NH; ++a) fw.st[(int64_t)(fw_fac<PC>() + a) * cap + s] = A[a];
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = Fcand;
        fw.iter[s] = iter; fw.halv[s] = 0;
        if (iter >= P.firth_handoff) list_push(true, fw.blk_list, fw.blk_count, s);
        else {
            // k_firth_step2 evaluates the hat diagonal in single precision in the standardised basis: safe while that basis' factor is well
            // conditioned.  Every pivot against its own diagonal entry (1 - R^2 of that column on the ones before it):
            bool well = A[sidx(1, 1)] >= FIRTH_HAT32_PIVOT * i11c;
#pragma unroll
            for (int j = 0; j < Q; ++j) well = well && (A[sidx(2 + j, 2 + j)] >= FIRTH_HAT32_PIVOT * hdiag[j]);
            if (fw.s64_list && !well) list_push(true, fw.s64_list, fw.s64_count, s);
            else list_push(true, step_list, step_count, s);
        }
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


// ---- penalised score at beta through the stored factor, Newton step -> cand (model.py:440-463) ------------------------------------------
// Per-lane constants: beta, the LDL^T factor of I(beta) moved to the standardised basis and rounded to single precision.  With
// x = B x_s (x_s0 = 1, x_s1 = the bit, x_s[2+j] = (z_j - m_j) / s_j) the factor of I_s = B^-1 I B^-T follows from the stored one without
// factoring again: L_s[2+j][0] = (L[2+j][0] - m_j) / s_j, L_s[2+j][1] = L[2+j][1] / s_j, L_s[2+j][2+k] = L[2+j][2+k] s_k / s_j,
// D_s[2+j] = D[2+j] / s_j^2; rows 0 and 1 are unchanged.  The hat diagonal is w_i sum_a zt_a^2 / D_s[a] with zt = L_s^-1 x_s: forward
// substitution on float2 = (even sample, odd sample), two L entries per 64-bit register (op_sel picks the half), 87 packed
// instructions per pair of samples against 90 fp64 instructions per sample.
// A variant whose standardised factor has a pivot below FIRTH_HAT32_PIVOT of D_s[0] (collinear covariates, a k-mer that is nearly a
// function of them) is not taken here: k_firth_eval2 lists it for the all-fp64 kernel (k_firth_step<Q, true>).
template <int Q>
__global__ __launch_bounds__(512) void k_firth_step2(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P, FirthWork fw,
                                                     const int *__restrict__ step_list, const int *__restrict__ step_count,
                                                     int *__restrict__ next_eval, int *__restrict__ next_eval_count,
                                                     int64_t V, double *__restrict__ out, uint32_t *__restrict__ flags)
{
    constexpr int PC = Q + 2, NH = PC * (PC + 1) / 2, RS = Q + 1, NL = PC * (PC - 1) / 2;
    const int cnt = *step_count;
    if ((int64_t)blockIdx.x * 64 >= cnt) return;
    const XWave xw = xwave();
    const int li = blockIdx.x * 64 + xw.lane;
    const bool live = li < cnt;
    const int s = step_list[live ? li : 0];
    const int64_t v = fw.var[s] < 0 ? ~fw.var[s] : fw.var[s];
    const int64_t cap = fw.cap;
    const int N = P.N, NB64 = P.NB64;
    double beta[PC], nU[PC];
    // L_s below the diagonal, row by row, as floats packed two per register pair: entry e = a (a - 1) / 2 + k  (a > k)
    v2f Lp[(NL + 1) / 2], dinv[(PC + 1) / 2];
    {
        // The transform reads 78 fp64 state values per lane and is register-hungry (the compiler issues the loads first and spills around
        // them).  The S wavefronts of a block need the same 64 variants' factors: wavefront 0 alone computes them and hands the packed floats
        // to the others through LDS (the reduction area of xw_sum, idle until the end of the pass), so the spill code runs in one wavefront
        // of eight (scratch traffic of a C4 batch 6 -> <1 GB).
        extern __shared__ double xw_lds[];
        float *lsh = (float *)xw_lds;                                            // [NL + PC + 2][64]
        constexpr int NF = NL + 1 + PC + 1;
        float Lf[NL + 1], df[PC + 1];
        if (xw.S == 1 || xw.w == 0) {
            Lf[NL] = 0.0f; df[PC] = 0.0f;
#pragma unroll
            for (int a = 0; a < PC; ++a) {
                const double sa = (a >= 2) ? P.wstd[Q + a - 2] : 1.0;
                const double isa = 1.0 / sa;
#pragma unroll
                for (int k = 0; k < a; ++k) {
                    double l = fw.st[(int64_t)(fw_fac<PC>() + sidx(a, k)) * cap + s];
                    if (a >= 2 && k == 0) l -= P.wstd[a - 2];
                    const double sk = (k >= 2) ? P.wstd[Q + k - 2] : 1.0;
                    Lf[a * (a - 1) / 2 + k] = (float)(l * sk * isa);
                }
                const double d = fw.st[(int64_t)(fw_fac<PC>() + sidx(a, a)) * cap + s];
                df[a] = (float)((sa * sa) / d);
            }
            if (xw.S > 1) {
#pragma unroll
                for (int e = 0; e < NL + 1; ++e) lsh[e * 64 + xw.lane] = Lf[e];
#pragma unroll
                for (int e = 0; e < PC + 1; ++e) lsh[(NL + 1 + e) * 64 + xw.lane] = df[e];
            }
        }
        if (xw.S > 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < NL + 1; ++e) Lf[e] = lsh[e * 64 + xw.lane];
#pragma unroll
            for (int e = 0; e < PC + 1; ++e) df[e] = lsh[(NL + 1 + e) * 64 + xw.lane];
            __syncthreads();                                                     // the area is the reduction's again
        }
        (void)NF;                                                               // the launcher sizes the area: NF x 64 floats
#pragma unroll
        for (int e = 0; e < (NL + 1) / 2; ++e) Lp[e] = v2f{Lf[2 * e], Lf[2 * e + 1]};
#pragma unroll
        for (int e = 0; e < (PC + 1) / 2; ++e) dinv[e] = v2f{df[2 * e], df[2 * e + 1]};
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) { beta[a] = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s]; nU[a] = 0.0; }
    auto Lbc = [&](int a, int k) -> v2f {                                       // L_s[a][k] in both halves
        const int e = a * (a - 1) / 2 + k;
        return (e & 1) ? __builtin_shufflevector(Lp[e >> 1], Lp[e >> 1], 1, 1) : __builtin_shufflevector(Lp[e >> 1], Lp[e >> 1], 0, 0);
    };
    auto Dbc = [&](int a) -> v2f {
        return (a & 1) ? __builtin_shufflevector(dinv[a >> 1], dinv[a >> 1], 1, 1) : __builtin_shufflevector(dinv[a >> 1], dinv[a >> 1], 0, 0);
    };
    // fp64 half of a sample: eta, t = exp(-|eta|), inv = 1 / (1 + t); returns w = mu (1 - mu) and hm = mu - 1/2
    auto front = [&](const double (&rc)[RS], double xd, double &wgt, double &hm) {
        double eta = fma(beta[1], xd, beta[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(beta[2 + j], rc[j], eta);
        const double t = exp_neg(fabs(eta)), u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        wgt = (t * inv) * inv;
        hm = copysign(fma(-0.5, t, 0.5) * inv, eta);                           // mu - 1/2 = sign(eta) (1 - t) / (2 (1 + t))
    };
    // the score's sample term with the sign flipped: -(y - mu + h (1/2 - mu)) = (mu - 1/2)(1 + h) + s / 2,  s = 1 - 2 y
    auto back = [&](const double (&rc)[RS], double xd, double hm, double h) {
        const double r = fma(0.5, rc[Q], fma(hm, h, hm));
        nU[0] += r; nU[1] = fma(xd, r, nU[1]);
#pragma unroll
        for (int j = 0; j < Q; ++j) nU[2 + j] = fma(rc[j], r, nU[2 + j]);
    };
    // x_s^T I_s^-1 x_s for the two samples of a pair; xb = their bits as floats, zs = their standardised covariates
    auto hat = [&](v2f xb, const v2f *zs) -> v2f {
        v2f zt[PC];
        zt[1] = xb - Lbc(1, 0);
        v2f qf = pkfma2(zt[1] * zt[1], Dbc(1), Dbc(0));                          // zt_0 = 1
#pragma unroll
        for (int a = 2; a < PC; ++a) {
            v2f t = zs[a - 2] - Lbc(a, 0);
#pragma unroll
            for (int k = 1; k < a; ++k) t = pkfma2(-Lbc(a, k), zt[k], t);
            zt[a] = t;
            qf = pkfma2(t * t, Dbc(a), qf);
        }
        return qf;
    };
    // The pairs' standardised covariates (GlmParams.rec_pf: Q float2 per pair) are fetched by VECTOR loads of a wave-uniform address, one
    // pair's worth at the top of the pair and used after both samples' fp64 arithmetic.  Not scalar loads: the SGPR file is full (two fp64
    // records in flight + the exp coefficients).  Not LDS either (first version of this kernel: records staged per word by LDS-DMA, read back
    // as broadcast ds_reads): LDS reads and scalar loads share one counter (lgkmcnt) and scalar loads return out of order, so a wait for
    // either is a wait for both, and every pair paid the full scalar-load latency -- 6.0 instead of 4.4 cycles per instruction.  Vector
    // loads count on vmcnt and return in order; a wave-uniform address is one cache line per instruction.
    const v2f *__restrict__ RP = (const v2f *)P.rec_pf;
    // constant address space: the kernel writes memory the compiler cannot see through (LDS-DMA, the counted waits), which would otherwise
    // take the "never clobbered" property from the records and turn their scalar loads into per-lane vector loads
    typedef const __attribute__((address_space(4))) double *cdptr;
    const cdptr R = (cdptr)P.rec_f;
    double ra[RS], rb[RS];
    {
        const int i0 = min(xw.w * 64, N - 1);
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i0 * RS + k];
    }
    uint64_t wnext = T[(int64_t)min(xw.w, NB64 - 1) * Vpad + v];
    uint64_t w64 = 0;
    for (int sb = xw.w; sb < NB64; sb += xw.S) {
        w64 = wnext;
        wnext = T[(int64_t)min(sb + xw.S, NB64 - 1) * Vpad + v];
        const int npair = min(64, N - sb * 64) >> 1;
        const int inext = min((sb + xw.S) * 64, N - 1);
        const v2f *zrec = RP + (int64_t)sb * 32 * Q;
        uint32_t wh = (uint32_t)w64;
#if FIRTH_STEP2_AHEAD
        // variant: the pair's standardised covariates fetched ONE PAIR AHEAD (20 more registers), so that the hat diagonal's 89 packed
        // instructions are free to fill the bubbles of the two samples' dependent fp64 chains instead of following them
        v2f zs[Q], zn[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) zs[j] = zrec[j];
#pragma unroll 1
        for (int pr = 0; pr < npair; ++pr) {
            if (pr == 16) wh = (uint32_t)(w64 >> 32);
            const int i = sb * 64 + 2 * pr, i2 = (pr == npair - 1) ? inext : i + 2;
            const int sh = (2 * pr) & 31;
            const uint32_t b0 = (wh >> sh) & 1u, b1 = (wh >> (sh + 1)) & 1u;
            const double xd0 = (double)b0, xd1 = (double)b1;
            double w0, w1, hm0, hm1;
            const int za = pipe_zero_v(ra[0]);
#pragma unroll
            for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
            {
                const v2f *zr = zrec + (int64_t)min(pr + 1, 31) * Q;
#pragma unroll
                for (int j = 0; j < Q; ++j) zn[j] = zr[j];
            }
            front(ra, xd0, w0, hm0);
            const int zb = pipe_zero_after(rb[0], w0);
            double xd1f = xd1;
            asm volatile("" : "+v"(xd1f) : "v"(w0));
            double rc0[RS];
#pragma unroll
            for (int k = 0; k < RS; ++k) rc0[k] = ra[k];
#pragma unroll
            for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
            front(rb, xd1f, w1, hm1);
            const v2f qf = hat(v2f{(float)b0, (float)b1}, zs);
            const v2f h = v2f{(float)w0, (float)w1} * qf;
            back(rc0, xd0, hm0, (double)h.x);
            back(rb, xd1, hm1, (double)h.y);
#pragma unroll
            for (int j = 0; j < Q; ++j) { zs[j] = zn[j]; asm volatile("" : "+v"(zs[j]) : "v"(w1)); }   // the copy (and the wait for zn) behind the pair's arithmetic
        }
#else
#pragma unroll 1
        for (int pr = 0; pr < npair; ++pr) {
            if (pr == 16) wh = (uint32_t)(w64 >> 32);
            const int i = sb * 64 + 2 * pr, i2 = (pr == npair - 1) ? inext : i + 2;
            const int sh = (2 * pr) & 31;
            const uint32_t b0 = (wh >> sh) & 1u, b1 = (wh >> (sh + 1)) & 1u;
            const double xd0 = (double)b0, xd1 = (double)b1;
            double w0, w1, hm0, hm1;
            const int za = pipe_zero_v(ra[0]);
#pragma unroll
            for (int k = 0; k < RS; ++k) rb[k] = R[(int64_t)(i + 1 + za) * RS + k];
            v2f zs[Q];
            {
                const v2f *zr = zrec + (int64_t)pr * Q;
#if FIRTH_STEP2_ABL == 2                                                   /* timing ablation: no LDS reads */
#pragma unroll
                for (int j = 0; j < Q; ++j) zs[j] = v2f{(float)pr, (float)j};
#else
#pragma unroll
                for (int j = 0; j < Q; ++j) zs[j] = zr[j];
#endif
            }
            front(ra, xd0, w0, hm0);
            // the wait for rb (and with it for zs) stays BEHIND the first sample's arithmetic: the fence takes that sample's weight as an
            // (unused) operand; a sched_barrier alone orders the fence but not the pure arithmetic around it
            const int zb = pipe_zero_after(rb[0], w0);
            double xd1f = xd1;
            asm volatile("" : "+v"(xd1f) : "v"(w0));                            // the second sample's chain starts from here (no instruction)
            double rc0[RS];
#pragma unroll
            for (int k = 0; k < RS; ++k) rc0[k] = ra[k];
#pragma unroll
            for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)(i2 + zb) * RS + k];
            front(rb, xd1f, w1, hm1);
#pragma unroll
            for (int j = 0; j < Q; ++j) asm volatile("" : "+v"(zs[j]) : "v"(w1));   // ... and the hat diagonal behind the second's (no instruction)
#if FIRTH_STEP2_ABL == 1                                                   /* timing ablation: no hat diagonal */
            const v2f qf = zs[0];
#else
            const v2f qf = hat(v2f{(float)b0, (float)b1}, zs);
#endif
            const v2f h = v2f{(float)w0, (float)w1} * qf;
            back(rc0, xd0, hm0, (double)h.x);
            back(rb, xd1, hm1, (double)h.y);
        }
#endif
    }
    if ((N & 1) && xw.w < NB64 && ((NB64 - 1 - xw.w) % xw.S) == 0) {            // the odd last sample, by the wavefront that walked the last word
        const int i = N - 1;
#pragma unroll
        for (int k = 0; k < RS; ++k) ra[k] = R[(int64_t)i * RS + k];
        const uint32_t b0 = (uint32_t)(w64 >> (i & 63)) & 1u;
        const double xd0 = (double)b0;
        double w0, hm0;
        front(ra, xd0, w0, hm0);
        v2f zs[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float z = (float)P.ws[(int64_t)i * Q + j]; zs[j] = v2f{z, z}; }
        const v2f qf = hat(v2f{(float)b0, (float)b0}, zs);
        back(ra, xd0, hm0, (double)((float)w0 * qf.x));
    }
    xw_sum(xw, nU);
    if (!live || xw.w != 0) return;
    double A[NH];
#pragma unroll
    for (int a = 0; a < NH; ++a) A[a] = fw.st[(int64_t)(fw_fac<PC>() + a) * cap + s];
    double u0[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) u0[a] = nU[a];
    ldl_solve<PC>(A, nU);                                                      // var_covar_mat . U, model.py:463 (nU = -U)
    // The fit's LAST likelihood pass, saved.  The reference stops once the PREVIOUS step is below 1e-4 (model.py:477-479), i.e. one iteration
    // after it could have: the candidate formed here is then the result, a step of ~1e-8 away (quadratic convergence), and the pass over the
    // samples that would follow only re-evaluates F and I11 there.  Both are known to more digits than the pass delivers: F(beta + d) =
    // F(beta) - U.d / 2 (U = the penalised score = -grad F, d = V U; third-order terms ~ N |d|^3 < 1e-17), I11(beta + d) = I11(beta) (1 + O(|d|)),
    // taken from the factor on record (I11 = D1 + L10^2 D0).  With the noise rules the step would be accepted (F does not increase but by rounding);
    // the literal rule (firth_noise = 0) keeps the pass.
    if (P.firth_last_taylor && P.firth_noise > 0.0 && fw.iter[s] > 0 && fw.st[(int64_t)fw_snp<PC>() * cap + s] < 1e-4) {
        double stepmax = 0.0, ud = 0.0; bool fin = true;
#pragma unroll
        for (int a = 0; a < PC; ++a) { stepmax = fmax(stepmax, fabs(nU[a])); fin = fin && isfinite(nU[a]); ud = fma(u0[a], nU[a], ud); }
        if (fin && stepmax <= 1e-7) {
            const double Fcand = fw.st[(int64_t)fw_fcur<PC>() * cap + s] - 0.5 * ud;
            const double i11 = fma(A[sidx(1, 0)] * A[sidx(1, 0)], A[sidx(0, 0)], A[sidx(1, 1)]);
            const double fitll = -Fcand;
            const double lrstat = -2.0 * (P.null_firth - fitll);
            double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);      // model.py:366-369
            const double b1 = beta[1] - nU[1];
            uint32_t fl = flags[v] | firth_sensitive(fw.iter[s] + 1, fw.st[(int64_t)fw_snp<PC>() * cap + s]);
            out[V + v] = pval; out[2 * V + v] = b1; out[3 * V + v] = sqrt(i11); out[4 * V + v] = beta[0] - nU[0];   // bse = sqrt(I11), model.py:491
#pragma unroll
            for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = beta[2 + j] - nU[2 + j];
            if (pval > P.lrtt || !isfinite(pval) || !isfinite(b1)) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
            flags[v] = fl;
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = beta[a] - nU[a];
    list_push(true, next_eval, next_eval_count, s);
}


// ---- launchers: which = 0 start vectors (k_firth_init2), 1 likelihood pass (k_firth_eval2), 2 score pass (k_firth_step2); n = upper bound of the list length -----------------
template <int Q>
static hipError_t launch_firth2(hipStream_t st, int which, int64_t n, const uint64_t *T, int64_t Vpad, int64_t V, GlmParams P, FirthWork fw,
                                const int *in_list, const int *in_count, int *next_eval, int *next_eval_count, int *step_list, int *step_count,
                                double *out, uint32_t *flags, int *plist, int *pcount)
{
    if (n <= 0) return hipSuccess;
    const int S = glm_split_waves(P.NB64);
    const dim3 grid((unsigned)((n + 63) / 64)), blks(64 * S);
    if (which == 0)
        hipLaunchKernelGGL(k_firth_init2<Q>, grid, dim3(64), 0, st, in_list, in_count, P, Vpad, fw, next_eval, next_eval_count);
    else if (which == 1)
        hipLaunchKernelGGL(k_firth_eval2<Q>, grid, blks, glm_split_lds(S), st, T, Vpad, V, P, fw, in_list, in_count, next_eval, next_eval_count,
                           step_list, step_count, out, flags, plist, pcount);
    else {
        // dynamic LDS: the reduction area of xw_sum, which also carries the factor from wavefront 0 to the others ((NL + PC + 2) x 64 floats)
        const size_t lds = S > 1 ? std::max(glm_split_lds(S), (size_t)(((Q + 2) * (Q + 1) / 2 + Q + 4) * 64 * sizeof(float))) : 0;
        hipLaunchKernelGGL(k_firth_step2<Q>, grid, blks, lds, st, T, Vpad, P, fw, in_list, in_count, next_eval, next_eval_count, V, out, flags);
    }
    return hipGetLastError();
}

extern "C" hipError_t shk_firth2_launch(hipStream_t st, int Q, int which, int64_t n, const uint64_t *T, int64_t Vpad, int64_t V, GlmParams P,
                                        double *fst, int *fiter, int *fhalv, int *fvar, int64_t fcap, const int *in_list, const int *in_count,
                                        int *next_eval, int *next_eval_count, int *step_list, int *step_count, double *out, uint32_t *flags,
                                        int *plist, int *pcount, int *blk_list, int *blk_count, int *s64_list, int *s64_count)
{
    FirthWork fw{fst, fiter, fhalv, fvar, fcap, blk_list, blk_count, s64_list, s64_count};
#define FIRTH2_CASE(q) case q: return launch_firth2<q>(st, which, n, T, Vpad, V, P, fw, in_list, in_count, next_eval, next_eval_count, step_list, step_count, out, flags, plist, pcount);
    switch (Q) {
#ifdef FIRTH2_ONLY_Q
        FIRTH2_CASE(FIRTH2_ONLY_Q)
#else
        FIRTH2_CASE(1) FIRTH2_CASE(2) FIRTH2_CASE(3) FIRTH2_CASE(4) FIRTH2_CASE(5) FIRTH2_CASE(6) FIRTH2_CASE(7)
        FIRTH2_CASE(8) FIRTH2_CASE(9) FIRTH2_CASE(10) FIRTH2_CASE(11) FIRTH2_CASE(12) FIRTH2_CASE(13) FIRTH2_CASE(14)
#endif
    default: return hipErrorInvalidValue;
    }
#undef FIRTH2_CASE
}
