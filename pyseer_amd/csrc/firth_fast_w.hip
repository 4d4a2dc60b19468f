// firth_fast_w.hip -- the one-pass Firth iteration's FIRST pass (single precision, firth_fast.hip: k_firth_fast<Q, true>) at TWO wavefronts
// per SIMD (round 5).
//
// Why.  k_firth_fast holds the whole third-moment tensor of 32 variants in one wavefront's accumulators (256 AGPRs at Q = 10), i.e. ONE
// wavefront per SIMD -- and one wavefront issues a vector instruction every 5.2-6 cycles whatever its kind (tools/ubench/valu_rate.hip,
// profiles/r05/valu_rate.txt: v_fma_f32 5.2 cycles at one wavefront per SIMD, 2.8 at two; conversions 8 against 4.5), with nothing to
// run under its LDS-DMA waits and barriers (1.4 of the pass' 4.8 ms, ablations in profiles/r05/firth_pass_ablations.txt).  The
// single-precision pass is all fp32 / integer instructions: a second wavefront on the SIMD nearly doubles its issue rate.
//
// How.  The accumulators are halved by giving a wavefront 16 variants instead of 32: v_mfma_f32_16x16x32_f16 (16 monomial rows x 32
// samples) x (32 samples x 16 variants), four accumulator registers per 16-row tile, 29 tiles at Q = 10 = 116 AGPRs.  Lane (n = lane & 15,
// kq = lane >> 4) takes samples 32 g + 8 kq + j (j = 0..7) of variant n: exactly the B operand's layout, as in firth_fast.hip.  A block is
// 8 wavefronts = 128 variants sharing the LDS copies of the run's tables (25 KB per 32-sample group, ring of three).  What a lane ends
// with is a quarter of its variant's sums, so the per-fit arithmetic (I, its factor, V, the penalty's contraction, the step, the routing) is
// not done here redundantly on four lanes: the sums are handed over through a workspace (FFW<Q>::ROWS floats per fit, rows x fits) to
// k_firth_fastw_fin, one lane per fit, which runs firth_fast.hip's epilogue on them.
//
// Same mathematics as k_firth_fast<Q, true> (same start, same sums up to the order of the additions, same routing); the pass exists to put
// the iterate 1e-5 from the fit, the fp64 passes that follow are firth_fast.hip's (model.py:414-504).
#include "firth_fast_common.h"

typedef float ffw_v4f __attribute__((ext_vector_type(4)));
#ifndef FFW_ABL
#define FFW_ABL 0                  /* timing ablations of k_firth_fastw_fin (results meaningless): 1 = no carrier count, 2 = no contraction */
#endif

#ifndef FFW_RHO2
#define FFW_RHO2 0.03
#define FFW_NOISE 1.5e-7
#define FFW_KB_REL 1e-7
#endif
template <int Q> struct FFW {
    typedef FFC<Q> C;
    static constexpr int T2H = (C::N2 + 15) / 16, T3H = (C::N3 + 15) / 16, NTA = T2H + T3H;   // 16-row tiles of the degree-2 / degree-3 tables
    static constexpr int NACC = 2 * T2H + T3H + 1, AWK = 2 * T2H + T3H;                        // (w - w0).m2, c.m3, c k.m2, (w - w0) k.(1, z)
    static constexpr int RS = (Q + 2 + 3) & ~3;            // floats per sample record: z_s[Q], s = 1 - 2 y (0 behind sample N), -2^12 w0; padded to 16 bytes
    static constexpr int ROWS = NACC * 16 + 32;            // hand-over rows per fit: the accumulator tiles' rows, then -score[PC] and I11 (hi), then their lo parts
    static constexpr int REC_PIECES = (32 * RS * 4 + 1023) / 1024, NPIECE = NTA + REC_PIECES, STAGE = NPIECE * 1024;
    static constexpr int NPW = (NPIECE + 7) / 8;           // 1 KB LDS-DMA pieces per wavefront and stage (8 wavefronts; the last ones repeat the last piece)
    static constexpr int NRING = 3;
};

extern "C" int shk_firth_fastw_layout(int Q, int *t2h, int *t3h, int *rs, int *rows)
{
    const int z1 = Q + 1, n2 = ff_tri(z1), n3 = ff_tet(z1);
    *t2h = (n2 + 15) / 16; *t3h = (n3 + 15) / 16; *rs = (Q + 2 + 3) & ~3;
    *rows = (2 * *t2h + *t3h + 1) * 16 + 32;
    return 0;
}

// ---- the sample pass ------------------------------------------------------------------------------------------------------------------------
//   tab16: [32-sample group][NTA tiles][64 lanes] x 16 bytes: lane (m = lane & 15, kq = lane >> 4) holds row m of the tile for samples
//          32 g + 8 kq .. + 7, as halves (the tables' hi parts: one product per tile, firth_fast.hip "the first pass")
//   rec16: per sample FFW<Q>::RS floats
//   ws:    [ROWS][ws_cap] floats; column = position in fast_list
template <int Q>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_firth_fast32w(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P, FirthWork fw, const int *__restrict__ fast_list,
                     const int *__restrict__ fast_count, const uint16_t *__restrict__ tab16, const float *__restrict__ rec16,
                     float *__restrict__ ws, int64_t ws_cap)
{
    typedef FFC<Q> C;
    typedef FFW<Q> W;
    constexpr int PC = C::PC, RS = W::RS, T2H = W::T2H, T3H = W::T3H, NTA = W::NTA, NACC = W::NACC, STAGE = W::STAGE, NPW = W::NPW, NRING = W::NRING;
    const int cnt = *fast_count;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int li = ((int)blockIdx.x * 8 + wave) * 16 + n;
    // (no early exit: the eight wavefronts of a block share the LDS copies of the run's tables and each issues an eighth of them; a lane past
    // the end of the list walks the samples of the list's last slot and writes nothing)
    const bool live = li < cnt;
    const int s = fast_list[min(li, cnt - 1)];
    const int vraw = fw.var[s];
    const int64_t v = vraw < 0 ? ~vraw : vraw;
    const int64_t cap = fw.cap;
    const int N = P.N;
    const int NG = (N + 31) >> 5;
    // beta (columns as given) -> the standardised basis x_s = (1, k, (z - m) / s):  b_s0 = b0 + sum m_j b_j,  b_s(2+j) = s_j b_(2+j)
    // a k-mer carried by most samples is taken by its complement (x_1 -> 1 - x_1: b_0 += b_1, b_1 = -b_1), as in firth_fast.hip
    int carriers = 0;
#pragma unroll 5
    for (int sb = kq; sb < P.NB64; sb += 4) carriers += __popcll(T[(int64_t)sb * Vpad + v]);
    carriers += __shfl_xor(carriers, 16); carriers += __shfl_xor(carriers, 32);
    const bool flip = 2 * carriers > N;
    const uint32_t flipm = flip ? 0xffu : 0u;
    float bs[PC];
    {
        double cand[PC], b0;
#pragma unroll
        for (int a = 0; a < PC; ++a) cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s];
        b0 = cand[0];
#pragma unroll
        for (int j = 0; j < Q; ++j) { bs[2 + j] = (float)(cand[2 + j] * P.wstd[Q + j]); b0 = fma(cand[2 + j], P.wstd[j], b0); }
        double b1 = cand[1];
        if (flip) { b0 += b1; b1 = -b1; }
        bs[0] = (float)b0; bs[1] = (float)b1;
    }

    ffw_v4f acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = ffw_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    float nU[PC], Ik0 = 0.0f;                               // -score = sum (mu - y) x;  I11 = sum w k
    // (round 6) summed in single precision within a group (8 samples of this lane) and in double precision across groups: 1 250 float additions
    // per lane at a running sum of ~20 put 1e-7 on the step of a rare variant (V_11 ~ bse^2 = 0.04) -- nothing for a FIRST pass, whose point is
    // 1e-5 from the fit anyway, but a second single-precision pass (firth_first32=2) must land within 1e-7 for the fp64 pass behind it to finish the fit
    double nUd[PC], Ik0d = 0.0;
#pragma unroll
    for (int a = 0; a < PC; ++a) { nU[a] = 0.0f; nUd[a] = 0.0; }

    extern __shared__ __attribute__((aligned(16))) char ffw_lds[];
    char *const lds = ffw_lds;
    const char *const tab_g = (const char *)tab16;
    const char *const rec_g = (const char *)rec16;
    // stage gi (buffer gi % NRING) = the table tiles of group gi - 1 (their MFMAs are issued among the samples of group gi) + the records of group gi
    auto dma = [&](int gi) {
        char *const buf = lds + (gi % NRING) * STAGE;
        const int gt = max(gi - 1, 0), gr = min(gi, NG - 1);
#pragma unroll
        for (int q2 = 0; q2 < NPW; ++q2) {
            const int pc = min(q2 * 8 + wave, W::NPIECE - 1);
            const char *src = pc < NTA ? tab_g + ((int64_t)gt * NTA + pc) * 1024 : rec_g + (int64_t)gr * (32 * RS * 4) + (pc - NTA) * 1024;
            const char *gsrc = src + lane * 16;
            const uint32_t ldst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(buf + pc * 1024);
            uint32_t keep;
            // (inline assembly: with the builtin the compiler waits vmcnt(0) in front of every LDS read -- firth_fast.hip)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
        }
    };
    uint32_t Pw[4], Pc[4], Pk[4], Px[4];                   // the previous group's B operands (w - w0, c, c k, (w - w0) k): 4 x half2 each
#pragma unroll
    for (int e = 0; e < 4; ++e) { Pw[e] = 0u; Pc[e] = 0u; Pk[e] = 0u; Px[e] = 0u; }
    auto frag = [&](const char *buf, int ta) { return __builtin_bit_cast(ff_v8h, *(const ff_v4u *)(buf + ta * 1024 + lane * 16)); };
    auto mfma = [&](ffw_v4f &a, const ff_v8h &A, const uint32_t (&B)[4]) {
        const ff_v4u b = {B[0], B[1], B[2], B[3]};
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, __builtin_bit_cast(ff_v8h, b), a, 0, 0, 0);
    };
    // the previous group's MFMAs for table tiles [lo, hi)
    auto tiles = [&](const char *buf, int lo, int hi) {
#pragma unroll
        for (int ta = 0; ta < NTA; ++ta) {
            if (ta < lo || ta >= hi) continue;
            const ff_v8h A = frag(buf, ta);
            if (ta < T2H) { mfma(acc[ta], A, Pw); mfma(acc[T2H + T3H + ta], A, Pk); if (ta == 0) mfma(acc[W::AWK], A, Px); }
            else mfma(acc[T2H + (ta - T2H)], A, Pc);
        }
    };
    auto pair_mask = [&](uint32_t byte, int pp) -> uint32_t {
        const uint32_t ev = (uint32_t)((int32_t)(byte << (31 - 2 * pp)) >> 31), od = (uint32_t)((int32_t)(byte << (30 - 2 * pp)) >> 31);
        return (ev & 0xffffu) | (od & 0xffff0000u);
    };
    // (tail: only the last group can hold slots behind sample N; everywhere else the mask `lv` is dropped: 2 of a sample's 55 instructions)
    auto sample = [&](const float (&rc)[RS], uint32_t bit, float &wf, float &cf, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        const float xd = (float)bit;
        float eta = fmaf(bs[1], xd, bs[0]);
#pragma unroll
        for (int j = 0; j < Q; ++j) eta = fmaf(bs[2 + j], rc[j], eta);
        const float sg = rc[Q], lv = fabsf(sg);                                 // s = +-1; 0 behind sample N
        float t = __builtin_amdgcn_exp2f(fabsf(eta) * -1.4426950408889634f);
        if (TAIL) t *= lv;
        const float u = 1.0f + t;
        float inv = __builtin_amdgcn_rcpf(u);
        inv = fmaf(fmaf(-u, inv, 1.0f), inv, inv);
        const float wgt = (t * inv) * inv;                                      // mu (1 - mu) = t / (1 + t)^2
        const float hm = copysignf(fmaf(-0.5f, t, 0.5f) * inv, eta);           // mu - 1/2
        float r = fmaf(0.5f, sg, hm);                                          // mu - y
        if (TAIL) r *= lv;
        nU[0] += r; nU[1] = fmaf(xd, r, nU[1]);
        Ik0 = fmaf(wgt, xd, Ik0);
#pragma unroll
        for (int j = 0; j < Q; ++j) nU[2 + j] = fmaf(rc[j], r, nU[2 + j]);
        wf = fmaf(wgt, (float)FF_SCALE, rc[Q + 1]);                             // 2^12 (w - w0): the record holds -2^12 w0 (firth_fast.hip: I = I(null model) + sum (w - w0) m2)
        cf = -(wgt * hm) * (float)FF_SCALE;                                     // c = w (1/2 - mu)
    };
    auto fetch_rec = [&](const char *buf, int j, float (&rc)[RS]) {
        const ffw_v4f *r = (const ffw_v4f *)(buf + NTA * 1024 + (8 * kq + j) * (RS * 4));
#pragma unroll
        for (int k2 = 0; k2 < RS / 4; ++k2) { const ffw_v4f x = r[k2]; rc[4 * k2] = x[0]; rc[4 * k2 + 1] = x[1]; rc[4 * k2 + 2] = x[2]; rc[4 * k2 + 3] = x[3]; }
    };
    auto pack = [](float a, float b) -> uint32_t { return __builtin_bit_cast(uint32_t, __builtin_convertvector(ff_v2f{a, b}, ff_v2h)); };

    dma(0);
    dma(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    __syncthreads();
    constexpr int TPP = (NTA + 3) / 4;                     // table tiles whose MFMAs follow each pair of samples
    // The variant's presence bits, 64 samples per load, fetched a pair of groups ahead BY INLINE ASSEMBLY: a load the compiler knows of gets
    // `s_waitcnt vmcnt(0)` in front of its first use -- which also waits for every table copy in flight (the compiler does not count them),
    // i.e. for the stage issued a moment ago, every other iteration.  Here the load is issued in front of this iteration's copies and is
    // complete at the iteration's own counted wait (it is older than the copies that may stay in flight); its register is read by nothing
    // but the move behind that wait.
    // (32 bits per group and iteration, two groups ahead: no branch in the loop)
    const uint32_t *const T32 = (const uint32_t *)T;
    auto bits_at = [&](int gi) { const int gc = min(gi, 2 * P.NB64 - 1); return T32 + (((int64_t)(gc >> 1) * Vpad + v) * 2 + (gc & 1)); };
    uint32_t wcur = *bits_at(0), wnext = *bits_at(1), wraw = 0;
    asm volatile("" : "+v"(wcur), "+v"(wnext));            // (the compiler's wait for these two loads sits HERE, not in the loop's header)
    auto group = [&](int g, auto tail) {
        {
            const uint32_t *wp = bits_at(g + 2);
            asm volatile("global_load_dword %0, %1, off" : "=&v"(wraw) : "v"(wp));
        }
        const uint32_t byte = ((wcur >> (8 * kq)) & 0xffu) ^ flipm;
        dma(min(g + 2, NG));
        const char *const buf = lds + (g % NRING) * STAGE;
        uint32_t Bw[4], Bc[4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            float ra[RS], rb[RS], w0, c0, w1, c1;
            fetch_rec(buf, 2 * pp, ra);
            fetch_rec(buf, 2 * pp + 1, rb);
            sample(ra, (byte >> (2 * pp)) & 1u, w0, c0, tail);
            sample(rb, (byte >> (2 * pp + 1)) & 1u, w1, c1, tail);
            Bw[pp] = pack(w0, w1); Bc[pp] = pack(c0, c1);
            tiles(buf, pp * TPP, min((pp + 1) * TPP, NTA));               // (the previous group's operands against the previous group's tiles)
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const uint32_t pm = pair_mask(byte, pp);
            Pw[pp] = Bw[pp]; Pc[pp] = Bc[pp]; Pk[pp] = Bc[pp] & pm; Px[pp] = Bw[pp] & pm;
        }
#pragma unroll
        for (int a = 0; a < PC; ++a) { nUd[a] += (double)nU[a]; nU[a] = 0.0f; }
        Ik0d += (double)Ik0; Ik0 = 0.0f;
        // this wavefront's share of the next stage has landed (the one after it may be in flight); the bare barrier: everyone's has, and
        // everyone is done reading this stage
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPW) : "memory");
        wcur = wnext;
        asm volatile("v_mov_b32 %0, %1" : "=v"(wnext) : "v"(wraw));
    };
#pragma unroll 1
    for (int g = 0; g < NG - 1; ++g) group(g, std::false_type{});
    group(NG - 1, std::true_type{});
    {                                                       // the MFMAs of the last group
        const char *const buf = lds + (NG % NRING) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tiles(buf, 0, NTA);
    }
    // ---- hand-over: a lane's registers are rows 4 kq .. + 3 of every 16-row tile of its variant; the four quarters' vector sums are added here
#pragma unroll
    for (int a = 0; a < PC; ++a) { nUd[a] += __shfl_xor(nUd[a], 16); nUd[a] += __shfl_xor(nUd[a], 32); }
    Ik0d += __shfl_xor(Ik0d, 16); Ik0d += __shfl_xor(Ik0d, 32);
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[(int64_t)(t * 16 + 4 * kq + r) * ws_cap + li] = acc[t][r];
    if (kq == 0) {                                          // hi + lo floats (the workspace is a float array)
#pragma unroll
        for (int a = 0; a <= PC; ++a) {
            const double x = a < PC ? nUd[a] : Ik0d;
            const float hi = (float)x;
            ws[(int64_t)(NACC * 16 + a) * ws_cap + li] = hi;
            ws[(int64_t)(NACC * 16 + 16 + a) * ws_cap + li] = (float)(x - (double)hi);
        }
    }
}

// ---- per fit: I, its factor, V, the penalty's contraction, the step, the routing (firth_fast.hip's epilogue on the handed-over sums) -------------
// A tile's 32 loads are issued together and waited for ONCE: left alone the compiler (under the register pressure of I, V and the gradient)
// sinks every load to its use -- ~450 serialised global loads per lane, 3.4 ms for 2^18 fits on the first build.
#define FFW_ROWS_LANDED(row)                                                          \
    do {                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 32; ++r_) row[r_] *= unscale;         \
    } while (0)
template <int Q, int... Ts>
__device__ __forceinline__ void ffw_all2i(const float *__restrict__ ws, int64_t ws_cap, int64_t col, float unscale, const double *__restrict__ inull,
                                          double (&I)[FFC<Q>::NH], double &rho, std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) row[r] = (TL * 32 + r < FFW<Q>::T2H * 16) ? ws[(int64_t)(TL * 32 + r) * ws_cap + col] : 0.0f;
        FFW_ROWS_LANDED(row);
        ff_tile2i<Q, TL>(row, inull, I, rho, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}
template <int Q, int... Ts>
__device__ __forceinline__ void ffw_all3(const float *__restrict__ ws, int64_t ws_cap, int64_t col, float unscale, const double (&V)[FFC<Q>::NH],
                                         double (&g)[FFC<Q>::PC], std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) row[r] = (TL * 32 + r < FFW<Q>::T3H * 16) ? ws[(int64_t)(FFW<Q>::T2H * 16 + TL * 32 + r) * ws_cap + col] : 0.0f;
        FFW_ROWS_LANDED(row);
        ff_tile3<Q, TL>(row, V, g, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}
template <int Q, int... Ts>
__device__ __forceinline__ void ffw_all2k(const float *__restrict__ ws, int64_t ws_cap, int64_t col, float unscale, const double (&V)[FFC<Q>::NH],
                                          double (&g)[FFC<Q>::PC], std::integer_sequence<int, Ts...>)
{
    auto one = [&](auto tile) {
        constexpr int TL = decltype(tile)::value;
        float row[32];
#pragma unroll
        for (int r = 0; r < 32; ++r)
            row[r] = (TL * 32 + r < FFW<Q>::T2H * 16) ? ws[(int64_t)((FFW<Q>::T2H + FFW<Q>::T3H) * 16 + TL * 32 + r) * ws_cap + col] : 0.0f;
        FFW_ROWS_LANDED(row);
        ff_tile2k<Q, TL>(row, V, g, std::make_integer_sequence<int, 32>{});
    };
    (one(std::integral_constant<int, Ts>{}), ...);
}

// One lane per fit of fast_list.  State on entry and routing exactly as k_firth_fast<Q, true> (firth_fast.hip): the single-precision pass has no
// likelihood (F = the determinant's half only, never compared: fcur = +inf is what the next pass finds).
template <int Q>
__global__ __launch_bounds__(64) void k_firth_fastw_fin(const uint64_t *__restrict__ T, int64_t Vpad, GlmParams P, FirthWork fw, const int *__restrict__ fast_list,
                                                        const int *__restrict__ fast_count, int *__restrict__ next_fast, int *__restrict__ next_fast_count,
                                                        int *__restrict__ exact_list, int *__restrict__ exact_count, const float *__restrict__ ws, int64_t ws_cap)
{
    typedef FFC<Q> C;
    typedef FFW<Q> W;
    constexpr int PC = C::PC, NH = C::NH, T2 = C::T2, T3 = C::T3;
    const int cnt = *fast_count;
    const int64_t li = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (li >= cnt) return;
    const int s = fast_list[li];
    const int vraw = fw.var[s];
    const int64_t v = vraw < 0 ? ~vraw : vraw;
    const int64_t cap = fw.cap;
    const int N = P.N;
    double cand[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) cand[a] = fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s];
    int carriers = 0;
#if !(FFW_ABL & 1)
    for (int sb = 0; sb < P.NB64; ++sb) carriers += __popcll(T[(int64_t)sb * Vpad + v]);
#endif
    const bool flip = 2 * carriers > N;
    double nUd[PC];
#pragma unroll
    for (int a = 0; a < PC; ++a) nUd[a] = (double)ws[(int64_t)(W::NACC * 16 + a) * ws_cap + li] + (double)ws[(int64_t)(W::NACC * 16 + 16 + a) * ws_cap + li];
    const double Ik0d = (double)ws[(int64_t)(W::NACC * 16 + PC) * ws_cap + li] + (double)ws[(int64_t)(W::NACC * 16 + 16 + PC) * ws_cap + li];
    // ---- I in the standardised basis (design order 0 = 1, 1 = k, 2.. = z), its factor, V ----------------------------------------------------
    const float unscale = (float)(1.0 / FF_SCALE);
    double I[NH];
#pragma unroll
    for (int a = 0; a < NH; ++a) I[a] = 0.0;
    double rho = 0.0;
    ffw_all2i<Q>(ws, ws_cap, li, unscale, P.ff_inull, I, rho, std::make_integer_sequence<int, T2>{});
    I[sidx(1, 0)] = Ik0d; I[sidx(1, 1)] = Ik0d;
#pragma unroll
    for (int j = 0; j < Q; ++j) {                          // the k-row: the null model's carrier sums + the matrix-core sum of (w - w0) k z_j
        const double cs = P.ch_bd[(int64_t)(1 + j) * Vpad + v];
        I[sidx(2 + j, 1)] = (flip ? P.null_h[1 + j] - cs : cs) + (double)(ws[(int64_t)(W::AWK * 16 + 1 + j) * ws_cap + li] * unscale);
    }
    double A[NH], det;
#pragma unroll
    for (int a = 0; a < NH; ++a) A[a] = I[a];
    const bool okf = ldl_factor<PC>(A, 1e-8, &det) && det > 0.0;
    const double F = -(0.5 * log(det));
    int iter = fw.iter[s];
    const double Fcur = fw.st[(int64_t)fw_fcur<PC>() * cap + s];
    const bool rise = iter >= 0 && !(F <= Fcur + FF_TAU);
    if (!okf || rise || !isfinite(F)) {                    // leave for the exact rounds at the last accepted point
        if (iter >= 0) {
#pragma unroll
            for (int a = 0; a < PC; ++a) fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s];
        }
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = INFINITY;
        fw.iter[s] = -1; fw.halv[s] = 0;
        list_push(true, exact_list, exact_count, s);
        return;
    }
    double Vm[NH];                                          // V = I^-1, packed lower
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
#if !(FFW_ABL & 2)
    ffw_all3<Q>(ws, ws_cap, li, unscale, Vm, gp, std::make_integer_sequence<int, T3>{});
    ffw_all2k<Q>(ws, ws_cap, li, unscale, Vm, gp, std::make_integer_sequence<int, T2>{});
#endif
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
    // the step in the columns as given: undo the complement (d_0 += d_1, d_1 = -d_1), then d_(2+j) /= s_j, d_0 -= sum m_j d_(2+j)
    if (flip) { d[0] += d[1]; d[1] = -d[1]; }
    bool fin = isfinite(d[0]) && isfinite(d[1]);
#pragma unroll
    for (int j = 0; j < Q; ++j) { d[2 + j] = d[2 + j] / P.wstd[Q + j]; d[0] = fma(-d[2 + j], P.wstd[j], d[0]); fin = fin && isfinite(d[2 + j]); }
    // accepted: beta <- cand (model.py:477-479: the stop rule tests the PREVIOUS step)
    double sn = 0.0;
    if (iter >= 0) {
#pragma unroll
        for (int a = 0; a < PC; ++a) { const double dd = cand[a] - fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s]; sn = fma(dd, dd, sn); }
        sn = sqrt(sn);
    }
    // A SECOND single-precision pass (iter == 0 on entry) leaves ~1e-7 of its own arithmetic in the point it forms, and the fit's remaining
    // passes shrink that by the iteration's linear rate each -- 1e-3 .. 2e-2 for most fits, so that the ONE fp64 pass behind it ends within
    // 1e-9 of the exact rounds' answer.  A fit that contracts slowly (quasi-separated: rate 0.1 .. 0.3) would keep 3e-8 of it: where this
    // pass' step is more than FFW_RHO2 of the previous one the pass is discarded -- state untouched, the fp64 pass takes it over from the
    // same candidate, as in round 5.
    // The same for a fit whose kbeta is so small that what is left shows RELATIVE to it (the reference-made rows hold kbeta to 1e-6 relative
    // with no absolute slack: kbeta = 7.5e-4 allows 7.5e-10): this pass leaves ~FFW_NOISE sqrt(V_11) on kbeta (every carrier's mu - y rounded to
    // float, summed over the carriers, through V_11 ~ 1 / (carriers w)), the fp64 pass a fraction max(rate, 0.02) of that; where that exceeds
    // FFW_KB_REL |kbeta| -- one fit in fifty -- the pass is discarded too.
    if (iter == 0 && fin) {
        double dn = 0.0;
#pragma unroll
        for (int a = 0; a < PC; ++a) dn = fma(d[a], d[a], dn);
        const double rate = sqrt(dn) / fmax(sn, 1e-300);
        const double v11 = fabs(Vm[sidx(1, 1)]);
        const bool slow = !(rate <= FFW_RHO2);
        const bool small = fmax(rate, 0.02) * FFW_NOISE * sqrt(v11) > FFW_KB_REL * fabs(cand[1] + d[1]);
        if (slow || small) { list_push(true, next_fast, next_fast_count, s); return; }
    }
    iter = iter < 0 ? 0 : iter + 1;
    // The pass evaluated the model at cand ROUNDED to float (in the standardised basis: k_firth_fast32w's bs), and the step it found leads
    // to the fit from THERE: the new candidate is that rounded point + d.  (cand + d carried the rounding, 6e-8 |beta|, into the next point --
    // nothing after a first pass, 1e-7 too much after a second one.)  The same arithmetic as the pass' prologue, then its inverse.
    double base[PC];
    {
        double b0 = cand[0];
#pragma unroll
        for (int j = 0; j < Q; ++j) { base[2 + j] = (double)(float)(cand[2 + j] * P.wstd[Q + j]) / P.wstd[Q + j]; b0 = fma(cand[2 + j], P.wstd[j], b0); }
        double b1 = cand[1];
        if (flip) { b0 += b1; b1 = -b1; }
        const double r0 = (double)(float)b0, r1 = (double)(float)b1;
        base[1] = flip ? -r1 : r1;
        double o0 = flip ? r0 - base[1] : r0;
#pragma unroll
        for (int j = 0; j < Q; ++j) o0 = fma(-base[2 + j], P.wstd[j], o0);
        base[0] = o0;
    }
#pragma unroll
    for (int a = 0; a < PC; ++a) {
        fw.st[(int64_t)(fw_beta<PC>() + a) * cap + s] = cand[a];
        fw.st[(int64_t)(fw_cand<PC>() + a) * cap + s] = fin ? base[a] + d[a] : cand[a];
    }
    fw.iter[s] = iter; fw.halv[s] = 0;
    const bool last = iter > 0 && sn < 1e-4;
    // (the single-precision pass finishes no fit: a candidate that meets the stop rule goes to k_firth_eval2, as in k_firth_fast<Q, true>)
    if (last || !fin || iter >= FF_MAXIT) {
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = INFINITY;
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = last ? sn : INFINITY;
        if (!last) fw.iter[s] = -1;
        list_push(true, exact_list, exact_count, s);
    } else {
        fw.st[(int64_t)fw_fcur<PC>() * cap + s] = INFINITY;
        fw.st[(int64_t)fw_snp<PC>() * cap + s] = sn;
        list_push(true, next_fast, next_fast_count, s);
    }
}

template <int Q>
static hipError_t launch_fastw(hipStream_t st, int64_t n, const uint64_t *T, int64_t Vpad, GlmParams P, FirthWork fw, const int *in_list, const int *in_count,
                               int *next_fast, int *next_fast_count, int *exact_list, int *exact_count, const uint16_t *tab16, const float *rec16,
                               float *ws, int64_t ws_cap)
{
    if (n <= 0) return hipSuccess;
    constexpr size_t lds = (size_t)FFW<Q>::NRING * FFW<Q>::STAGE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)k_firth_fast32w<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((k_firth_fast32w<Q>), dim3((unsigned)((n + 127) / 128)), dim3(512), lds, st, T, Vpad, P, fw, in_list, in_count, tab16, rec16, ws, ws_cap);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_firth_fastw_fin<Q>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, T, Vpad, P, fw, in_list, in_count, next_fast, next_fast_count,
                       exact_list, exact_count, ws, ws_cap);
    return hipGetLastError();
}

#ifdef FF_ONLY_Q
#define FFW_ALL_Q(X) X(FF_ONLY_Q)
#else
#define FFW_ALL_Q(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#endif
extern "C" hipError_t shk_firth_fastw_launch(hipStream_t st, int Q, int64_t n, const uint64_t *T, int64_t Vpad, GlmParams P, double *fst, int *fiter, int *fhalv,
                                             int *fvar, int64_t fcap, const int *in_list, const int *in_count, int *next_fast, int *next_fast_count,
                                             int *exact_list, int *exact_count, const uint16_t *tab16, const float *rec16, float *ws, int64_t ws_cap)
{
    FirthWork fw{fst, fiter, fhalv, fvar, fcap, nullptr, nullptr, nullptr, nullptr};
#define FFW_CASE(q) case q: return launch_fastw<q>(st, n, T, Vpad, P, fw, in_list, in_count, next_fast, next_fast_count, exact_list, exact_count, tab16, rec16, ws, ws_cap);
    switch (Q) {
        FFW_ALL_Q(FFW_CASE)
    default: return hipErrorInvalidValue;
    }
#undef FFW_CASE
}
