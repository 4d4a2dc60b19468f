// firth_fast.hip -- fit_firth's iteration (pyseer/model.py:414-504) ONE sample pass per iteration, for the case that sends every variant through
// it at the BASELINE size (force_firth, N >= 4096, Q <= 10 covariates: BASELINE config C4).
//
// The reference's iteration at beta needs, per variant, sums over the samples that depend on V = I(beta)^-1 (the hat diagonal h_i = w_i x_i^T V
// x_i inside the penalised score U* = X^T(y - mu + h (1/2 - mu))), and V is only known once the sums of I(beta) are complete: two passes per
// iteration (k_firth_eval2: I, log-likelihood;  k_firth_step2: h, U*), each paying the eta -> exp -> 1/(1 + t) chain of every sample.
//
// One pass is enough when the V-dependent sum is taken apart:  sum_i x_ia c_i (x_i^T V x_i) = sum_bc V_bc T_abc  with the third-moment tensor
// T_abc = sum_i c_i x_ia x_ib x_ic,  c_i = w_i (1/2 - mu_i), which does NOT depend on V.  With x = (1, k, z) (k the variant's 0/1 column,
// z the run's Q covariates) the entries of T without k are sum_i c_i m(z_i) over the 286 monomials m of degree <= 3 in (1, z), the entries
// with k are sum_i c_i k_i m2(z_i) over the 66 monomials of degree <= 2 (k^2 = k); likewise I(beta)'s covariate block is sum_i w_i m2(z_i).
// The monomials are per-RUN tables; the sums are GEMMs [monomials x samples] . [samples x variants] and run on the matrix cores
// (v_mfma_f32_32x32x16_f16, operands split hi + lo: three products, 2^-21 relative; fp32 accumulation) BESIDE the vector ALU, which is left
// with eta, the exponential, the log-likelihood, the exact fp64 score X^T(y - mu) and the k-row of I: ~120 instructions per sample against
// 134 + 120 for the two passes.  T and I from this route carry ~1e-6 relative error.  T enters the fixed point only through the penalty
// (O(p / N) of the score): measured 1e-10 on beta; I only steers the iteration (V) and the step-halving guard.
//
// What stays exact: the iteration is the reference's own (same start vector as the rounds: k_firth_init2; same accepted steps; same stop
// rule: one step after a step below 1e-4).  Every fit is FINISHED by the exact kernels: the candidate that meets the stop rule goes to
// k_firth_eval2, which evaluates F and I11 there in fp64 and writes the outputs.  A fit that is not an ordinary one -- F rises by more than
// FF_TAU (a rise of F above its evaluation noise), a pivot of I fails, FF_MAXIT passes -- leaves for the exact rounds at its last accepted beta.
//
// Layout: a wavefront owns 32 variants; lane (n = lane & 31, h = lane >> 5) takes samples 16 g + 8 h + j (j = 0..7) of every 16-sample group g of
// variant n: exactly the B operand of the MFMA (column n, k = 8 h + j), so a lane's eight weights go to the matrix core as they are.  The A
// operand (32 monomials x 16 samples) is read from the run's table (GlmParams.ff_tab) one 16-byte fragment per lane.  Accumulators: T2 tiles
// (w . m2), T3 tiles (c . m3), T2 tiles (c k . m2), 16 registers each: 240 AGPRs at Q = 10; one wavefront per SIMD.
#include "firth_fast_common.h"
#ifndef FF_F32_TU
extern "C" int shk_firth_fast_supported(int Q) { return Q >= 1 && Q <= 10; }
extern "C" int shk_firth_fast_row2(int Q, int a, int b) { return ff_row2(a, b, Q + 1); }
extern "C" int shk_firth_fast_tiles(int Q, int *t2, int *t3, int *rs)
{
    const int z1 = Q + 1, n2 = ff_tri(z1), n3 = ff_tet(z1);
    *t2 = (n2 + 31) / 32; *t3 = (n3 + 31) / 32; *rs = (Q + 3 + 1) & ~1;
    return 0;
}
#endif

// the 32 rows of one accumulator tile, every lane seeing all of them: register r of a lane holds row (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ void ff_rows(const ff_v16f &acc, int h, float (&row)[32])
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x = acc[r];
        const float o = __shfl_xor(x, 32);
        const int r0 = (r & 3) + 8 * (r >> 2);
        row[r0] = h ? o : x;
        row[r0 + 4] = h ? x : o;
    }
}
template <int Q, int... Ts>
__device__ __forceinline__ void ff_all3(const ff_v16f (&acc)[FFC<Q>::NACC], int h, float unscale, const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC],
                                        std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
        ff_rows(acc[FFC<Q>::T2 + TL], h, row);
#pragma unroll
        for (int r = 0; r < 32; ++r) row[r] *= unscale;
        ff_tile3<Q, TL>(row, V, g, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}
template <int Q, int... Ts>
__device__ __forceinline__ void ff_all2k(const ff_v16f (&acc)[FFC<Q>::NACC], int h, float unscale, const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC],
                                         std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
        ff_rows(acc[FFC<Q>::T2 + FFC<Q>::T3 + TL], h, row);
#pragma unroll
        for (int r = 0; r < 32; ++r) row[r] *= unscale;
        ff_tile2k<Q, TL>(row, V, g, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}
template <int Q, int... Ts>
__device__ __forceinline__ void ff_all2i(const ff_v16f (&acc)[FFC<Q>::NACC], int h, float unscale, const double *__restrict__ inull, double (&I)[FFC<Q>::NH],
                                         double &rho, std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
        ff_rows(acc[TL], h, row);
#pragma unroll
        for (int r = 0; r < 32; ++r) row[r] *= unscale;
        ff_tile2i<Q, TL>(row, inull, I, rho, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}

// e^-x for x >= 0 to 1e-11 relative (degree 9; the fixed point needs eta -> mu to ~1e-9, the one-pass F sums 5000 log(1 + t): 2e-8): as exp_neg
// (glm_device.h) with four terms less
__device__ __forceinline__ double ff_exp_neg(double x)
{
    const double u = -fmin(x, 800.0);
    const double kf = rint(u * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, u);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 2.7557319223985893e-06;                       // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                     // 1/8!
    p = fma(p, r, 1.984126984126984e-04);                    // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);                   // 1/6!
    p = fma(p, r, 8.333333333333333e-03);                    // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);                   // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                   // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

__device__ __forceinline__ double ff_xor32(double x) { return __shfl_xor(x, 32); }

// One pass at `cand` for the slots of fast_list: F (one-pass), I, U*, the next candidate; routing as described in the header.
//   fw state on entry (as k_firth_eval2): beta = last accepted point, cand = the point to evaluate, fcur = one-pass F at beta, snp = norm of the
//   step that led to beta, iter = accepted steps so far (-1: cand is the start vector, nothing to compare with).
//   exact_list: slots for k_firth_eval2 (the finishing evaluation, or a fit that leaves the fast passes)
template <int Q, bool F32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_firth_fast(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P, FirthWork fw, const int *__restrict__ fast_list,
                  const int *__restrict__ fast_count, int *__restrict__ next_fast, int *__restrict__ next_fast_count,
                  int *__restrict__ exact_list, int *__restrict__ exact_count, int64_t V, double *__restrict__ out, uint32_t *__restrict__ flags)
{
    typedef FFC<Q> C;
    constexpr int PC = C::PC, NH = C::NH, RS = C::RS, T2 = C::T2, T3 = C::T3, NTA = C::NTA, NACC = C::NACC;
    const int cnt = *fast_count;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int base = ((int)blockIdx.x * 4 + wave) * 32;
    // (no early exit: the four wavefronts of a block share the LDS copies of the run's tables and each issues a quarter of them; a wavefront
    // past the end of the list walks the samples of the list's last slot and writes nothing)
    const int li = base + n;
    const bool live = li < cnt;
    const int s = fast_list[min(li, cnt - 1)];
    const int vraw = fw.var[s];
    const int64_t v = vraw < 0 ? ~vraw : vraw;
    const int64_t cap = fw.cap;
    const int N = P.N;
    const int NG = (N + 15) >> 4;
    // beta (columns as given) -> the standardised basis x_s = (1, k, (z - m) / s):  b_s0 = b0 + sum m_j b_j,  b_s(2+j) = s_j b_(2+j)
    typedef std::conditional_t<F32, float, double> RT;     // the arithmetic type of the sample loop
    double cand[PC], bs[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s];
    bs[0] = cand[0]; bs[1] = cand[1];
#pragma unroll
    for (int j = 0; j < Q; ++j) { bs[2 + j] = cand[2 + j] * P.wstd[Q + j]; bs[0] = fma(cand[2 + j], P.wstd[j], bs[0]); }
    // a k-mer carried by most samples is taken by its complement (x_1 -> 1 - x_1: b_0 += b_1, b_1 = -b_1): the k-row of I is then a sum over the
    // minority, and I is as well conditioned as the data allows (F, I's determinant and the fitted values do not change)
    int carriers = 0;
    for (int sb = h; sb < P.NB64; sb += 2) carriers += __popcll(T[(int64_t)sb * Vpad + v]);
    carriers += __shfl_xor(carriers, 32);
    const bool flip = 2 * carriers > N;
    if (flip) { bs[0] += bs[1]; bs[1] = -bs[1]; }
    const uint32_t flipm = flip ? 0xffu : 0u;

    ff_v16f acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    RT nU[PC], Ik0 = 0;                                    // -score = sum (mu - y) x;  I11 = sum w k
    RT bsr[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) { nU[a] = 0; bsr[a] = (RT)bs[a]; }
    double apos = 0.0, seta = 0.0, prod = 1.0;                              // (apos, seta: twice the sum of max(s eta, 0), as sum |eta| + sum s eta)
    int pexp = 0;
    // The run's tables reach the block through LDS: per 16-sample group the NTA x 2 fragment blocks of the monomial table (1 KB each: one
    // wave-wide LDS-DMA) and the group's 16 sample records (2 KB slot), double buffered; iteration g of the loop below reads records g and
    // the fragments of group g - 1 (the MFMAs trail the samples by one group).
    constexpr int NPIECE = (NTA * 2 + 2 + 3) & ~3, STAGE = NPIECE * 1024;
    // (the single-precision pass takes the tables' hi halves only: one product per tile, 2^-11 on sums that need 1e-5)
    constexpr int NFETCH = F32 ? NTA + 2 : NTA * 2 + 2, NPW = (NFETCH + 3) / 4;     // 1 KB pieces, NPW per wavefront (the last ones padding)
    // Ring of FOUR stages, copies issued THREE iterations ahead (round 6; was three / two).  Iteration g reads stage g and -- its last pair's
    // read-ahead: the next group's first record and first fragment -- stage g + 1, so stage g + 1 must have landed for EVERY wavefront when g
    // starts: the end of iteration g - 1 waits for this wavefront's share of everything but the copies issued last (stage g + 2) and then takes
    // the barrier.  With copies only two ahead that wait covered stage g, and stage g + 1 was read on the strength of having been in flight for a
    // whole iteration: 5 400 cycles at Q = 10 in fp64, but ~1 500 in the single-precision pass at Q <= 3 -- less than a loaded memory system's
    // latency: 5 of 60 repeated calls differed (tools/gpu_firth_determinism.py, profiles/r06/firth_determinism.txt).
    constexpr int NRING = 4;
    char *const lds = (char *)xw_lds;
    const char *const tab_g = (const char *)P.ff_tab;
    const char *const rec_g = F32 ? (const char *)P.ff_rec32 : (const char *)P.ff_rec;
    auto dma = [&](int gi) {                                                    // everything iteration gi reads, into buffer gi % NRING
        char *const buf = lds + (gi % NRING) * STAGE;
        const int gt = max(gi - 1, 0), gr = min(gi, NG - 1);
#pragma unroll
        for (int q2 = 0; q2 < NPW; ++q2) {
            const int pf = q2 * 4 + wave;                                       // the pf-th piece fetched = piece pc of the stage
            const int pc = !F32 ? pf : pf < NTA ? 2 * pf : NTA + pf;
            const char *src = pc < NTA * 2 ? tab_g + ((int64_t)gt * NTA * 2 + pc) * 1024 : rec_g + (int64_t)gr * (16 * RS * 8) + (pc - NTA * 2) * 1024;
            // (inline assembly, not __builtin_amdgcn_global_load_lds: the compiler cannot tell which LDS bytes a DMA in flight will write, so with
            // the builtin it puts s_waitcnt vmcnt(0) in front of the next LDS read -- the whole latency of the copy just issued, every
            // iteration.  The waits for these copies are the explicit ones at the end of the iteration.)
            const char *gsrc = src + lane * 16;
            const uint32_t ldst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(buf + pc * 1024);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
        }
    };
    uint32_t Bw[2][4], Bc[2][4], Bk[2][4], Bx[2][4];                        // this group's B operands (w - w0, c, c k, (w - w0) k): [hi, lo][4 x half2]
    uint32_t Pw[2][4], Pc[2][4], Pk[2][4], Px[2][4];                        // the previous group's: its MFMAs are issued among this group's samples
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int e = 0; e < 4; ++e) { Pw[q2][e] = 0u; Pc[q2][e] = 0u; Pk[q2][e] = 0u; Px[q2][e] = 0u; }

    // the MFMAs of tile `ta` of the A table (ta < T2: degree 2, against w and c k; else degree 3, against c) for the operands in P*
    auto tile_mfma = [&](int ta, const ff_v4u &ah, const ff_v4u &al) {
        const ff_v8h Ah = __builtin_bit_cast(ff_v8h, ah), Al = __builtin_bit_cast(ff_v8h, al);
        auto three = [&](ff_v16f &a, const uint32_t (&B)[2][4]) {
            const ff_v4u bh = {B[0][0], B[0][1], B[0][2], B[0][3]}, bl = {B[1][0], B[1][1], B[1][2], B[1][3]};
            const ff_v8h Bh = __builtin_bit_cast(ff_v8h, bh), Bl = __builtin_bit_cast(ff_v8h, bl);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, a, 0, 0, 0);
            if constexpr (!F32) {
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, a, 0, 0, 0);
            }
        };
        if (FF_ABL & 1) { acc[ta < T2 ? ta : T2 + (ta - T2)][0] += __builtin_bit_cast(float, ah[0] ^ al[1]); return; }
        if (ta < T2) { three(acc[ta], Pw); three(acc[T2 + T3 + ta], Pk); if (ta == 0) three(acc[C::AWK], Px); }
        else three(acc[T2 + (ta - T2)], Pc);
    };
    // one sample: returns the scaled weights (w - w0, c) as floats.  (c k and (w - w0) k are not formed per sample: k is 0 / 1, so their
    // halves are the halves of c and w - w0 under the pair's presence bits -- pair_mask below; round 5: 14 -> 7 instructions per pair)
    // (tail: the group may hold slots behind sample N -- only the last group does; everywhere else the masks `lv` are dropped.  Round 5)
    auto sample = [&](const RT (&rc)[RS], uint32_t bit, float &wf, float &cf, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        if constexpr (F32) {
            // the FIRST pass (at the start vector, some 1e-2 from the fit) in single precision: its step need not be better than the 1e-5 the
            // next point is from the fit anyway; no log-likelihood (nothing to compare F with yet)
            const float xd = (float)bit;
            float eta = fmaf(bsr[1], xd, bsr[0]);
#pragma unroll
            for (int j = 0; j < Q; ++j) eta = fmaf(bsr[2 + j], rc[j], eta);
            const float lv = rc[Q + 1];
            float t = __builtin_amdgcn_exp2f(fabsf(eta) * -1.4426950408889634f);
            if (TAIL) t *= lv;
            const float u = 1.0f + t;
            float inv = __builtin_amdgcn_rcpf(u);
            inv = fmaf(fmaf(-u, inv, 1.0f), inv, inv);
            const float wgt = (t * inv) * inv;
            const float hm = copysignf(fmaf(-0.5f, t, 0.5f) * inv, eta);
            float r = fmaf(0.5f, rc[Q], hm);
            if (TAIL) r *= lv;
            nU[0] += r; nU[1] = fmaf(xd, r, nU[1]);
            Ik0 = fmaf(wgt, xd, Ik0);
#pragma unroll
            for (int j = 0; j < Q; ++j) nU[2 + j] = fmaf(rc[j], r, nU[2 + j]);
            wf = fmaf(wgt, (float)FF_SCALE, rc[Q + 2]);                          // (the record holds -2^12 w0)
            cf = -(wgt * hm) * (float)FF_SCALE;
        } else {
        const double xd = (double)bit;
        double eta = fma(bs[1], xd, bs[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fma(bs[2 + j], rc[j], eta);
        const double lv = rc[Q + 1];
        double t = ff_exp_neg(fabs(eta));
        if (TAIL) t *= lv;
        const double u = 1.0 + t;
        double inv = __builtin_amdgcn_rcp(u);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        inv = fma(fma(-u, inv, 1.0), inv, inv);
        const double wgt = (t * inv) * inv;                                     // mu (1 - mu) = t / (1 + t)^2   (0 behind sample N)
        const double hm = copysign(fma(-0.5, t, 0.5) * inv, eta);              // mu - 1/2
        // -log-likelihood term = max(s eta, 0) + log(1 + t), and max(s eta, 0) = (|eta| + s eta) / 2 for s = +-1: two sums, halved at the end
        // (an add with the |.| modifier and an fma instead of multiply, max, add).  s = 0 behind N: the tail keeps the max form.
        if (TAIL) apos += 2.0 * fmax(rc[Q] * eta, 0.0);
        else { apos += fabs(eta); seta = fma(rc[Q], eta, seta); }
        prod *= u;
        double r = fma(0.5, rc[Q], hm);                                         // mu - y = (mu - 1/2) + s / 2
        if (TAIL) r *= lv;
        nU[0] += r; nU[1] = fma(xd, r, nU[1]);
        Ik0 = fma(wgt, xd, Ik0);                                                // I11 = sum w k: the reference's bse^2, kept exact (k is 0 / 1: the product is)
#pragma unroll
        for (int j = 0; j < Q; ++j) nU[2 + j] = fma(rc[j], r, nU[2 + j]);
        wf = (float)fma(wgt, FF_SCALE, rc[Q + 2]);                              // 2^12 (w - w0) (the record holds -2^12 w0: one rounding either way): I = I(null model) + sum (w - w0) m2, the sum an order of magnitude smaller than I
        cf = (float)(-(wgt * hm) * FF_SCALE);                                   // c = w (1/2 - mu)
        }
    };
    // 0xffff in the half of the even / the odd sample of pair pp that carries the variant (bits 2 pp, 2 pp + 1 of the group's byte)
    auto pair_mask = [&](uint32_t byte, int pp) -> uint32_t {
        const uint32_t ev = (uint32_t)((int32_t)(byte << (31 - 2 * pp)) >> 31), od = (uint32_t)((int32_t)(byte << (30 - 2 * pp)) >> 31);
        return (ev & 0xffffu) | (od & 0xffff0000u);
    };
    auto stash = [&](uint32_t (&B)[2][4], int e, float a, float b) {
        const ff_v2h hh = __builtin_convertvector(ff_v2f{a, b}, ff_v2h);
        if constexpr (F32) { B[0][e] = __builtin_bit_cast(uint32_t, hh); B[1][e] = 0u; return; }
        const ff_v2f back = __builtin_convertvector(hh, ff_v2f);
        const ff_v2h ll = __builtin_convertvector(ff_v2f{a - back.x, b - back.y}, ff_v2h);
        B[0][e] = __builtin_bit_cast(uint32_t, hh); B[1][e] = __builtin_bit_cast(uint32_t, ll);
    };
    // sample j (0..7) of this lane's half of the group in buffer `buf`
    auto fetch_rec = [&](const char *buf, int j, RT (&rc)[RS]) {
        if constexpr (F32) {                                                    // (single-precision records: the same slots, half filled)
            const ff_v2f *r = (const ff_v2f *)(buf + NTA * 2048 + (8 * h + j) * (RS * 8));
#pragma unroll
            for (int k2 = 0; k2 < RS / 2; ++k2) { const ff_v2f x = r[k2]; rc[2 * k2] = x.x; rc[2 * k2 + 1] = x.y; }
        } else {
            const ff_v2d *r = (const ff_v2d *)(buf + NTA * 2048 + (8 * h + j) * (RS * 8));
#pragma unroll
            for (int k2 = 0; k2 < RS / 2; ++k2) { const ff_v2d x = r[k2]; rc[2 * k2] = x.x; rc[2 * k2 + 1] = x.y; }
        }
    };
    // A group = 8 samples per lane = four pairs; the MFMAs of the PREVIOUS group's operands are issued AMONG the pairs' arithmetic (the matrix
    // core works beside the vector ALU only if the instruction stream alternates: sched_group_barrier), table tiles pp, pp + 4, pp + 8 ...
    // behind pair pp (an even share of the MFMAs: a degree-2 tile carries six, a degree-3 tile three), their A fragments read from LDS ahead of
    // the pair.  The tables of iterations g + 2 and g + 3 are in flight while g computes (ring of NRING buffers): the end of an iteration waits for its
    // wavefront's share of g + 1 and g + 2 only.
    constexpr int PER = (NTA + 3) / 4;
    RT ra[RS], rb[RS];
    uint64_t w64 = 0, wnext = T[v];
    // fragments of the slot's tiles q2 = lo..hi-1 (tile q2 * 4 + pp)
    auto slot_load = [&](const char *buf, int pp, int lo, int hi, ff_v4u (&ah)[PER], ff_v4u (&al)[PER]) {
#pragma unroll
        for (int q2 = 0; q2 < PER; ++q2) {
            const int ta = q2 * 4 + pp;
            if (q2 >= lo && q2 < hi && ta < NTA) {
                ah[q2] = *(const ff_v4u *)(buf + (ta * 2 + 0) * 1024 + lane * 16);
                if constexpr (!F32) al[q2] = *(const ff_v4u *)(buf + (ta * 2 + 1) * 1024 + lane * 16);   // (the single-precision pass never reads al)
            }
        }
    };
    auto slot_mfma = [&](int pp, int lo, int hi, const ff_v4u (&ah)[PER], const ff_v4u (&al)[PER]) {
#pragma unroll
        for (int q2 = 0; q2 < PER; ++q2) {
            const int ta = q2 * 4 + pp;
            if (q2 >= lo && q2 < hi && ta < NTA) tile_mfma(ta, ah[q2], F32 ? ah[q2] : al[q2]);
        }
    };
    dma(0);
    dma(1);
    dma(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");         // (stages 0 and 1 have landed; stage 2 may be in flight)
    __syncthreads();
    fetch_rec(lds, 0, ra);
    ff_v4u ah[PER], al[PER];                                                    // the slot's A fragments: [0] read a sample ahead of the rest
    slot_load(lds, 0, 0, 1, ah, al);
    auto group = [&](int g, auto tail) {
        if ((g & 3) == 0) {                                                     // the variant's next 64 presence bits, fetched four groups ahead
            w64 = wnext;
            wnext = T[(int64_t)min((g >> 2) + 1, P.NB64 - 1) * Vpad + v];
        }
        const uint32_t byte = ((uint32_t)(w64 >> (16 * (g & 3) + 8 * h)) & 0xffu) ^ flipm;
        if (!(FF_ABL & 2)) dma(min(g + 3, NG));
#if FF_STAGGER
        // the four wavefronts leave the barrier together and would hit the LDS with their reads at the same moments of every pair: a quarter of
        // a pair's time apart they find it free
        if (wave == 1) __builtin_amdgcn_s_sleep(FF_STAGGER); else if (wave == 2) __builtin_amdgcn_s_sleep(2 * FF_STAGGER); else if (wave == 3) __builtin_amdgcn_s_sleep(3 * FF_STAGGER);
#endif
        const char *const buf = lds + (g % NRING) * STAGE;
        const char *const bufn = lds + ((g + 1) % NRING) * STAGE;               // (landed: the end of the previous iteration waited for everything but stage g + 2)
        // A pair of samples = one slot.  LDS reads are issued a whole sample ahead of their use and the fences keep them there (left alone the
        // compiler clusters them in front of their first use: 22 exposed LDS latencies per group, a third of the kernel's time):
        //   [record of the odd sample, fragments 1..]  |  even sample + the MFMAs of fragment 0  |  [record of the next even sample, the next
        //   slot's fragment 0]  |  odd sample + the MFMAs of fragments 1..
        // so that the slot's MFMAs are spread over both samples (the matrix core is busy 32 cycles per MFMA: all fifteen among one sample's
        // hundred vector instructions and the MFMA issue waits for the pipe)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            float w0, c0, w1, c1;
            fetch_rec(buf, 2 * pp + 1, rb);
            slot_load(buf, pp, 1, PER, ah, al);
            __builtin_amdgcn_sched_barrier(0);
            if (FF_ABL & 4) { w0 = (float)ra[0]; c0 = (float)ra[1]; nU[pp] += ra[3]; }
            else sample(ra, (byte >> (2 * pp)) & 1u, w0, c0, tail);
            slot_mfma(pp, 0, 1, ah, al);
#if FF_SCHED
#pragma unroll
            for (int k2 = 0; k2 < 9; ++k2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, FF_SCHED, 0);
            }
#endif
            // (the fences order instructions with side effects; plain arithmetic is placed wherever its operands allow.  Empty volatile asms that
            // "define" the even sample's results pin its arithmetic in front of the next fence, i.e. UNDER the LDS reads issued above)
            if constexpr (F32) asm volatile("" : "+v"(w0), "+v"(c0), "+v"(Ik0));
            else asm volatile("" : "+v"(w0), "+v"(c0), "+v"(apos), "+v"(seta), "+v"(prod), "+v"(Ik0));
#pragma unroll
            for (int a = 0; a < PC; ++a) asm volatile("" : "+v"(nU[a]));
            __builtin_amdgcn_sched_barrier(0);
            if (pp < 3) { fetch_rec(buf, 2 * pp + 2, ra); slot_load(buf, pp + 1, 0, 1, ah, al); }
            else { fetch_rec(bufn, 0, ra); slot_load(bufn, 0, 0, 1, ah, al); }
            __builtin_amdgcn_sched_barrier(0);
            if (FF_ABL & 4) { w1 = (float)rb[0]; c1 = (float)rb[1]; nU[pp] += rb[3]; }
            else sample(rb, (byte >> (2 * pp + 1)) & 1u, w1, c1, tail);
            stash(Bw, pp, w0, w1); stash(Bc, pp, c0, c1);
            {
                const uint32_t pm = pair_mask(byte, pp);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) { Bk[q2][pp] = Bc[q2][pp] & pm; Bx[q2][pp] = Bw[q2][pp] & pm; }
            }
            slot_mfma(pp, 1, PER, ah, al);
#if FF_SCHED
            // one MFMA, then a run of vector instructions, and so on through the sample
#pragma unroll
            for (int k2 = 0; k2 < 9; ++k2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, FF_SCHED, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
            for (int e = 0; e < 4; ++e) { Pw[q2][e] = Bw[q2][e]; Pc[q2][e] = Bc[q2][e]; Pk[q2][e] = Bk[q2][e]; Px[q2][e] = Bx[q2][e]; }
        if (!F32 && (g & 3) == 3) { int e2; prod = frexp(prod, &e2); pexp += e2; }
        // this wavefront's share of the next TWO iterations' tables has landed (the copies issued last, stage g + 3, may still be in flight), then the bare barrier:
        // everyone's has, and everyone is done reading this iteration's buffer.  (__syncthreads() would add a vmcnt(0) and wait for both)
        if (!(FF_ABL & 2)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPW) : "memory");
    };
#pragma unroll 1
    for (int g = 0; g < NG - 1; ++g) group(g, std::false_type{});
    group(NG - 1, std::true_type{});
    {                                                                           // the MFMAs of the last group (slot 0's first fragment is in hand)
        const char *const buf = lds + (NG % NRING) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            slot_load(buf, pp, 0, PER, ah, al);
            slot_mfma(pp, 0, PER, ah, al);
        }
    }
    // ---- the two halves' partial sums ----------------------------------------------------------------------------------------------------
    double nUd[PC], Ik0d;
#pragma unroll
    for (int a = 0; a < PC; ++a) { const double x = (double)nU[a]; nUd[a] = x + ff_xor32(x); }
    { const double x = (double)Ik0; Ik0d = x + ff_xor32(x); }
    double ll = 0.0;                                                        // (the single-precision pass has no likelihood: nothing to compare it with)
    if constexpr (!F32) {
        int e2; prod = frexp(prod, &e2); pexp += e2;
        double lp = fma((double)pexp, 0.6931471805599453, log(prod));
        apos = 0.5 * (apos + seta);
        lp += ff_xor32(lp); apos += ff_xor32(apos);
        ll = -(apos + lp);
    }
    // ---- I in the standardised basis (design order 0 = 1, 1 = k, 2.. = z), its factor, V ----------------------------------------------------
    const float unscale = (float)(1.0 / FF_SCALE);
    double I[NH];
#pragma unroll
    for (int a = 0; a < NH; ++a) I[a] = 0.0;
    double rho = 0.0;
    ff_all2i<Q>(acc, h, unscale, P.ff_inull, I, rho, std::make_integer_sequence<int, T2>{});
    I[sidx(1, 0)] = Ik0d; I[sidx(1, 1)] = Ik0d;
    {   // the k-row: sum w k z_j = (the null model's part: the carrier sums of w0 z_j, k_glm_bitdot; of the complement: the totals minus them)
        // + the matrix-core sum of (w - w0) k z_j = rows 1..Q of the degree-2 table's tile 0
        float row[32];
        ff_rows(acc[C::AWK], h, row);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const double cs = P.ch_bd[(int64_t)(1 + j) * Vpad + v];
            I[sidx(2 + j, 1)] = (flip ? P.null_h[1 + j] - cs : cs) + (double)(row[1 + j] * unscale);
        }
    }
    double A[NH], det;
#pragma unroll
    for (int a = 0; a < NH; ++a) A[a] = I[a];
    const bool okf = ldl_factor<PC>(A, 1e-8, &det) && det > 0.0;
    const double F = -(ll + 0.5 * log(det));                                // the one-pass F (standardised basis: a constant away from the reference's)
    int iter = fw.iter[s];
    const double Fcur = fw.st[(int64_t)fw_fcur<PC>() * cap + s];
    // leave for the exact rounds at the last accepted point: a real increase of F, a failed pivot, anything not finite
    const bool rise = iter >= 0 && !(F <= Fcur + FF_TAU);
    if (!okf || rise || !isfinite(F)) {
        if (live && h == 0) {
            if (iter >= 0) {
#pragma unroll
                for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
            }
            fw.st[(int64_t)fw_snp<PC>() * cap + s] = INFINITY;
            fw.iter[s] = -1; fw.halv[s] = 0;
            list_push(true, exact_list, exact_count, s);
        }
        return;
    }
    double Vm[NH];                                                           // V = I^-1, packed lower
#pragma unroll
    for (int col = 0; col < PC; ++col) {
        double e[PC];
#pragma unroll
        for (int a = 0; a < PC; ++a) e[a] = (a == col) ? 1.0 : 0.0;
        ldl_solve<PC>(A, e);
#pragma unroll
        for (int a = col; a < PC; ++a) Vm[sidx(a, col)] = e[a];
    }
    // ---- penalty: g_a = sum_bc V_bc T_abc;  U* = -nU + g;  step = V U* ------------------------------------------------------------------
    double gp[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) gp[a] = 0.0;
    ff_all3<Q>(acc, h, unscale, Vm, gp, std::make_integer_sequence<int, T3>{});
    ff_all2k<Q>(acc, h, unscale, Vm, gp, std::make_integer_sequence<int, T2>{});
    double U[PC], d[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) U[a] = gp[a] - nUd[a];
#pragma unroll
    for (int a = 0; a < PC; ++a) {
        double acc2 = 0.0;
#pragma unroll
        for (int b = 0; b < PC; ++b) acc2 = fma(a >= b ? Vm[sidx(a, b)] : Vm[sidx(b, a)], U[b], acc2);
        d[a] = acc2;
    }
    double ud = 0.0, dmax = 0.0;                                             // U . d (the same number in every basis), for F one step on
#pragma unroll
    for (int a = 0; a < PC; ++a) ud = fma(U[a], d[a], ud);
    const double i00 = I[sidx(0, 0)];
    // the step in the columns as given: undo the complement (d_0 += d_1, d_1 = -d_1), then d_(2+j) /= s_j, d_0 -= sum m_j d_(2+j)
    if (flip) { d[0] += d[1]; d[1] = -d[1]; }
    bool fin = isfinite(d[0]) && isfinite(d[1]);
#pragma unroll
    for (int j = 0; j < Q; ++j) { d[2 + j] = d[2 + j] / P.wstd[Q + j]; d[0] = fma(-d[2 + j], P.wstd[j], d[0]); fin = fin && isfinite(d[2 + j]); }
#pragma unroll
    for (int a = 0; a < PC; ++a) dmax = fmax(dmax, fabs(d[a]));
    if (!live || h != 0) return;
    // accepted: beta <- cand.  The stop rule tests the PREVIOUS step (model.py:477-479): with iter accepted steps so far and sn the norm of
    // the step that led here, the candidate formed now is the fit's last iff iter > 0 and sn < 1e-4 ... and that is known before it is evaluated
    double sn = 0.0;
    if (iter >= 0) {
#pragma unroll
        for (int a = 0; a < PC; ++a) { const double dd = cand[a] - fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s]; sn = fma(dd, dd, sn); }
        sn = sqrt(sn);
    }
    iter = iter < 0 ? 0 : iter + 1;
#pragma unroll
    for (int a = 0; a < PC; ++a) {
        fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s] = cand[a];
        fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = fin ? cand[a] + d[a] : cand[a];
    }
    fw.iter[s] = iter; fw.halv[s] = 0;
    const bool last = iter > 0 && sn < 1e-4;                                // the candidate just formed is the fit's result
    // The fit's LAST likelihood pass, saved (as k_firth_step2 does, firth_rounds.hip): the result is a step of ~1e-8 away, where F(beta + d) =
    // F(beta) - U.d / 2 (third-order terms ~ N |d|^3 < 1e-17) and I11(beta + d) = I11(beta) (1 + O(|d|)).  F here is the one-pass F: exact
    // log-likelihood, I = I(null) + the matrix-core sum of (w - w0) m2 (~1e-7 on log det), in the reference's columns (det I = det I_s prod s_j^2);
    // I11 = sum w k in fp64 (the complement's: I00 - I11').  A last step above 1e-7 keeps the exact evaluation, and so does a fit whose weights
    // have moved far from the null model's (rho > FF_RHO: a strong effect): the matrix-core sum's 1e-6 then shows in log det I at ~1e-6 rho.
    if (last && fin && dmax <= FF_FIN_TOL && rho <= FF_RHO && !(FF_ABL & 8)) {
        double lsd = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) lsd += log(P.wstd[Q + j]);
        const double fitll = -(F - 0.5 * ud) + lsd;
        const double i11 = flip ? i00 - Ik0d : Ik0d;
        const double lrstat = -2.0 * (P.null_firth - fitll);
        double pval = 1.0; if (lrstat > 0.0) pval = sh_chi2_sf1(lrstat);      // model.py:366-369
        const double b1 = cand[1] + d[1];
        uint32_t fl = flags[v] | firth_sensitive(iter + 1, sn);
        out[V + v] = pval; out[2 * V + v] = b1; out[3 * V + v] = sqrt(i11); out[4 * V + v] = cand[0] + d[0];   // bse = sqrt(I11), model.py:491
#pragma unroll
        for (int j = 0; j < Q; ++j) out[(5 + j) * V + v] = cand[2 + j] + d[2 + j];
        if (pval > P.lrtt || !isfinite(pval) || !isfinite(b1)) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;
        flags[v] = fl;
        return;
    }
    if (last || !fin || iter >= FF_MAXIT) {
        // k_firth_eval2 on this slot: compares F(cand) with fcur (here +inf: the step is accepted, as the noise rule accepts a last step of
        // ~1e-8), tests snp < 1e-4 with iter > 0 (true for `last`: the outputs are written), otherwise carries on with the exact rounds
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = INFINITY;
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = last ? sn : INFINITY;
        if (!last) { fw.iter[s] = -1; }
        list_push(true, exact_list, exact_count, s);
    } else {
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = F32 ? (double)INFINITY : F;
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = sn;
        list_push(true, next_fast, next_fast_count, s);
    }
}

template <int Q, bool F32>
static hipError_t launch_firth_fast(hipStream_t st, int64_t n, const uint64_t *T, int64_t Vpad, GlmParams P, FirthWork fw, const int *in_list,
                                    const int *in_count, int *next_fast, int *next_fast_count, int *exact_list, int *exact_count, int64_t V, double *out,
                                    uint32_t *flags)
{
    if (n <= 0) return hipSuccess;
    constexpr size_t lds = 4 * (size_t)((FFC<Q>::NTA * 2 + 2 + 3) & ~3) * 1024;      // NRING stages
    hipLaunchKernelGGL((k_firth_fast<Q, F32>), dim3((unsigned)((n + 127) / 128)), dim3(256), lds, st, T, Vpad, P, fw, in_list, in_count, next_fast, next_fast_count,
                       exact_list, exact_count, V, out, flags);
    return hipGetLastError();
}

// Two translation units from this file (the kernel is large, and there are two per Q): firth_fast.o holds the fp64 kernels and the entry points,
// firth_fast32.o (-DFF_F32_TU) the single-precision first pass.
#ifdef FF_ONLY_Q
#define FF_ALL_Q(X) X(FF_ONLY_Q)
#else
#define FF_ALL_Q(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#endif
#define FF_LAUNCH_ARGS hipStream_t st, int Q, int64_t n, const uint64_t *T, int64_t Vpad, GlmParams P, double *fst, int *fiter, int *fhalv, int *fvar, \
                       int64_t fcap, const int *in_list, const int *in_count, int *next_fast, int *next_fast_count, int *exact_list, int *exact_count, \
                       int64_t V, double *out, uint32_t *flags
#ifdef FF_F32_TU
extern "C" hipError_t shk_firth_fast_launch32(FF_LAUNCH_ARGS)
{
    FirthWork fw{fst, fiter, fhalv, fvar, fcap, nullptr, nullptr, nullptr, nullptr};
#define FF_CASE(q) case q: return launch_firth_fast<q, true>(st, n, T, Vpad, P, fw, in_list, in_count, next_fast, next_fast_count, exact_list, exact_count, V, out, flags);
    switch (Q) {
        FF_ALL_Q(FF_CASE)
    default: return hipErrorInvalidValue;
    }
#undef FF_CASE
}
#else
extern "C" hipError_t shk_firth_fast_launch32(FF_LAUNCH_ARGS);
extern "C" hipError_t shk_firth_fast_launch(FF_LAUNCH_ARGS, int f32)
{
    if (f32) return shk_firth_fast_launch32(st, Q, n, T, Vpad, P, fst, fiter, fhalv, fvar, fcap, in_list, in_count, next_fast, next_fast_count, exact_list,
                                            exact_count, V, out, flags);
    FirthWork fw{fst, fiter, fhalv, fvar, fcap, nullptr, nullptr, nullptr, nullptr};
#define FF_CASE(q) case q: return launch_firth_fast<q, false>(st, n, T, Vpad, P, fw, in_list, in_count, next_fast, next_fast_count, exact_list, exact_count, V, out, flags);
    switch (Q) {
        FF_ALL_Q(FF_CASE)
    default: return hipErrorInvalidValue;
    }
#undef FF_CASE
}
#endif
