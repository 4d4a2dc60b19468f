// glm_params.h -- the per-run constants of the fixed-effects kernels: one definition for the kernels (glm_common.h) and the host side
// (glm_api.inc), which pass it by value across the launcher boundary
#pragma once

struct GlmParams {
    int N, NB64, continuous, force_firth;
    int n1, n0;                   // #(y==1), #(y==0)
    double ymean_logit;           // log(mean(y)/(1-mean(y)))   model.py:323-324
    double yc_sum, yc_sq;         // centred-phenotype sums (Welch prefilter)
    double null_llf, null_firth, pret, lrtt;
    double min_af, max_af; int af_on;
    int newton_mode;              // 0 = fp32-Hessian fast path with fp64 fallback (default), 1 = all-fp64 (reference trajectory)
    const void *zz16;             // the same products as halves in the B layout of v_mfma_f32_32x32x16_f16, hi and lo parts (pass32_pk): [16-sample group][32-column block][hi, lo][64 lanes] x 16 bytes
    const void *zl16;             // pass32_pk_f16 (Newton steering): (1, z_1 .. z_Q) as halves in the B layout of the products table: [16-sample group][64 lanes] x 16 bytes, lane (n = lane & 31, kg = lane >> 5) = column n of samples 16 g + 8 kg .. + 7
    const float *zz;              // per-sample products table for fast_pass_mfma (FastCols<Q>::STRIDE floats per sample), or null
    const double *ws;             // N x q covariates standardised per column (the fast Newton path iterates in these coordinates), or null
    const double *wstd;           // [2q] column means, then column scales, of that standardisation
    const double *ztz, *zty;      // Z^T Z (packed lower, (q+1) x (q+1)) and Z^T y for Z = [1, W]: the variant-independent part of the OLS normal equations
    int firth_halv_handoff;       // rejected halvings of one step after which a variant leaves the rounds for k_firth_blk (default 6)
    int firth_handoff;            // accepted Firth steps after which a variant leaves the rounds for k_firth_blk (16, or 0 for the routed variants of an ordinary run at N >= 768)
    int f32_steps;                // first Newton steps of the fast path taken entirely in single precision (default 3, 2 with the warm start)
    // Firth step halving (model.py:465-474).  Default: an increase of F within firth_noise * |F| (4 ulp) is evaluation noise, and a step
    // whose largest component is below firth_accept (1e-10) is accepted outright.  SEERHIP_ROUTE firth_literal=1 / SEERHIP_ROUTE firth_strict=1 set both
    // to 0: the reference's literal `F(new) > F(old)`, spurious firth-fails on last-bit ties included (case 1).
    double firth_noise, firth_accept;
    int firth_last_taylor;        // 1 (default with the noise rules): k_firth_step2 finishes a fit whose stop rule is already met and whose step is <= 1e-7
                                  // without the last likelihood pass (F to second order, I11 from the factor on record); SEERHIP_ROUTE firth_last=0 turns it off
    // Warm start of the fast Newton phase: the maximum-likelihood fit WITHOUT the variant column, [b0, bz...] in the coordinates that phase
    // iterates in (standardised covariates), computed once per run (sh_glm_setup).  The likelihood is concave, so the iteration reaches the
    // same fixed point as from the reference's start vector (model.py:323-324) in about half the steps; anything that does not converge
    // cleanly is restarted by k_glm_slow from the reference's start vector on the reference's trajectory, as before.  warm_on = 0: off.
    int warm_on;
    double warm[16];
    const double *w0, *a0;        // final pass as a correction to the null model (k_glm_final<Q, true>): w0[N] = mu0 (1 - mu0) at the null MLE,
                                  // a0 = sum_i w0_i z_ij z_ik packed (j >= k), both in the standardised coordinates; null = plain fp64 pass
    int firth_warm;               // 1: Firth rounds start at fwarm = the null-model fit [b0, bz...] in the ORIGINAL covariate coordinates (force_firth only)
    double fwarm[16];
    int *dbg;                     // development counters (unused: null): [0] wave passes, [1] waves, [2] lane steps, [3] fitted lanes, [4] final-pass repeats
    double f32_tol;               // a single-precision pass whose step is <= f32_tol ends the fast phase (the fp32 score's noise floor is ~1e-7)
    double fast_tol;              // largest step (any coordinate) at which the fast phase hands over to the final pass' exact Newton step
    // The fast phase as rounds of lean kernels over lists (glm_kernels.hip, "the fast phase as ROUNDS"): single-precision Newton rounds until a
    // lane's step is <= chord_enter, then chord rounds (fp64 score, the last single-precision Hessian factor kept) until rho * step <= chord_tol.
    // Workspaces are SoA over the batch's padded variant count: ch_bs / ch_g [PC][Vpad], ch_fac [PC(PC+1)/2][Vpad] (floats in the first half of a double array: the LDL^T factor of the single-precision Hessian), ch_hf the same
    // shape in float, ch_md / ch_rho [Vpad]; ch_list: two ping-pong lists for the Newton rounds, two for the chord rounds; ch_cnt their counters.
    // ch_list[4] + ch_cnt[30]: the converged variants, for the finishing kernels (fin_rounds; k_glm_ll / k_glm_dpass / k_glm_finish).
    // bd_tab: nibble tables of (w0, w0 z, r0) at the null model (k_glm_bitdot -> ch_bd [Q+2][Vpad]); null_h = (sum w0, sum w0 z_j),
    // null_g = the null model's score (sum r0, sum r0 z_j; ~1e-12), both in the standardised coordinates.
    int chord_on, chord_n32, chord_rounds, fin_rounds;
    // ll_first (round 3): the FIRST chord round's pass is the exact fp64 likelihood pass (k_glm_ll: ll, separation callback, score) instead of
    // the score pass, and a variant that leaves the chord rounds there is finished from that pass: ch_b0 = its beta before the chord step,
    // ch_list[5] + ch_cnt[29] = those variants (k_glm_dpass at the chord result, k_glm_finish<DIRECT>).  One fp64 sample pass per variant less.
    int ll_first;
    double *ch_b0;
    double chord_enter, chord_tol;
    double *ch_bs, *ch_fac, *ch_g, *ch_md, *ch_bd, *ch_ll;
    float *ch_hf, *ch_rho;
    int *ch_list[6], *ch_cnt;
    const double *bd_tab;
    const void *bd_i8; const double *bd_scale;   // bitdot_i8.hip: the same sums on the int8 matrix cores (digit table, column scales); null = k_glm_bitdot
    const float *wfp, *yf, *w0f;  // packed-fp32 passes (pass32_pk): wfp = per pair of samples a record of Q + 2 float2 (standardised covariates, y, w0;
                                  // each (even sample, odd sample)); yf, w0f = y and w0 as float arrays (the odd last sample); null = unpacked passes
    const double *rec;            // per sample a record of Q + 1 doubles (standardised covariates, then y): k_glm_score / k_glm_ll
    int firth_lean;               // bit 0: k_firth_eval<Q, true>, bit 1: k_firth_step<Q, true> (both when rec_o is set)
    const double *rec_f;          // Firth round kernels (firth_rounds.hip): per sample Q covariates as given, then s = 1 - 2 y
    const float *rec_pf;          // per PAIR of samples Q float2 = (even, odd) standardised covariates: k_firth_step2 (staged through LDS)
    const double *rec_o;          // the same with the covariates as given, y in {0, 1} only: the Firth rounds (info_pass_bin); null = info_pass
    double null_h[16], null_g[16];
    // the first Newton step as a BORDERED solve (k_glm_first_step): with the variant's column ordered last, the leading (Q+1) x (Q+1) block of
    // X^T W0 X / n -- intercept and covariates at the null model -- is the same for every variant; its unit-lower factor, the reciprocal
    // pivots and the constant halves of the solve are computed once on the host.  Layout: L [m(m+1)/2, sidx], 1/D [m], y_c = L^-1 g_c [m],
    // x_c0 = L^-T (y_c / D) [m], m = Q + 1.  nullptr: the general kernel (k_glm_solve32<Q, true>).
    const double *b1;
    // One-pass Firth iteration (firth_fast.hip; force_firth at N >= 4096 with Q <= 10): ff_tab = the monomials of degree <= 2 and <= 3 in
    // (1, standardised covariates) as halves, hi and lo parts, in the A layout of v_mfma_f32_32x32x16_f16: [16-sample group][tile][hi, lo][64
    // lanes] x 16 bytes (lane (m, h): row m of the tile, samples 16 g + 8 h .. + 7); ff_rec = per sample (padded to whole groups) the
    // standardised covariates, s = 1 - 2 y, live = 1 (0 behind sample N), w0 = the null model's weight, as doubles (FFC<Q>::RS per sample).  null = the two-pass rounds.
    const void *ff_tab;
    const double *ff_rec;
    const float *ff_rec32;        // the same records in single precision (same slots, half filled): the first pass
    const double *ff_inull;       // sum_i w0_i m2(i) over the degree-2 table's rows, with the table's own (hi + lo) values: the null model's covariate block of I
};
#define FIRTH_F_NOISE 8.9e-16      /* default of GlmParams.firth_noise: four ulp of F */
#define FIRTH_ACCEPT_BELOW 1e-10   /* default of GlmParams.firth_accept */
#define FIRTH_WARM_LIMIT 7         /* accepted steps after which a warm-started Firth fit is restarted from the reference's start vector */

// fit_lineage_effect over a compacted list of rows (glm_lineage.hip, glm_wide.hip; the job stream of csrc/job_api.inc)
struct LinList { const int *list; const long long *cnt; const uint32_t *flags; int mode; };
__host__ __device__ inline bool lin_wanted(uint32_t f, int mode)
{
    if (f & SH_FLAG_PREFILTER) return false;
    return mode == 2 ? (f & SH_FLAG_FILTER) == 0 : (f & SH_NOTE_FIRTH_FAIL) == 0;
}

