// glm_firth_v1.hip -- round 2's Firth kernels (split off glm_kernels.hip in round 5).  What still runs them: a Firth fit without covariates, a
// non-0/1 phenotype, SEERHIP_ROUTE firth_strict=1 (k_firth_init / k_firth_eval / k_firth_step: the device-resident state machine); the few
// Firth-routed rows of an ordinary run at N >= 768 and every fit the rounds hand off (k_firth_blk: one workgroup per variant); a variant whose
// information matrix is singular (k_glm_firth_pinv: numpy.linalg.pinv semantics, model.py:450).  With covariates the rounds are
// firth_rounds.hip's, and a forced-Firth run at N >= 4096 goes through firth_fast.hip / firth_fast_w.hip first.
#include "glm_passes.h"
#include "glm_blk.h"

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


// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
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


template <int Q>
static hipError_t launch_firth_pinv(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, const double *y, const double *W, GlmParams P, int *plist, int *pcount,
                                    double *out, uint32_t *flags)
{
    hipLaunchKernelGGL(k_glm_firth_pinv<Q>, dim3(512), dim3(256), 0, st, T, Vpad, V, y, W, P, plist, pcount, out, flags);
    return hipGetLastError();
}
// shk_glm_launch's which = 3 (glm_kernels.hip): the listed variants through k_glm_firth_pinv
extern "C" hipError_t shk_glm_firth_pinv_launch(hipStream_t st, int Q, const uint64_t *T, int64_t Vpad, int64_t V, const double *y, const double *W, GlmParams P,
                                                int *plist, int *pcount, double *out, uint32_t *flags)
{
#define PINV_CASE(q) case q: return launch_firth_pinv<q>(st, T, Vpad, V, y, W, P, plist, pcount, out, flags);
    switch (Q) {
        PINV_CASE(0) PINV_CASE(1) PINV_CASE(2) PINV_CASE(3) PINV_CASE(4) PINV_CASE(5) PINV_CASE(6) PINV_CASE(7)
        PINV_CASE(8) PINV_CASE(9) PINV_CASE(10) PINV_CASE(11) PINV_CASE(12) PINV_CASE(13) PINV_CASE(14)
    default: return hipErrorInvalidValue;
    }
#undef PINV_CASE
}
