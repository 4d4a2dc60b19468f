// lmm_kernels.hip -- FaST-LMM per-variant test on MI355X (gfx950).
//
// Reference path restated here (see DESIGN.md §LMM for the algebra):
//   pyseer/lmm.py:228-260            fit_lmm_block
//   pyseer/fastlmm/lmm_cov.py:165-194 rotate  (+ Linreg.regress :874-880)
//   pyseer/fastlmm/lmm_cov.py:597-838 nLLeval / nLLcore, computeAKA/AKB :885-916
//
// For a 0/1 variant x the reference needs  xKx = sum_r (U~^T x)_r^2 / Sd_r  and  xKy = sum_r (U~^T x)_r (U~^T y)_r / Sd_r
// with U~ = P U (P = covariate residual maker).  With M = U~ diag(1/Sd) U~^T (N x N, per-run constant):
//     xKy = x . v          v = U~ diag(1/Sd) U~^T y                      -> O(N) masked sum   (k_lmm_linear)
//     xKx = sum_i x_i M_ii + x^T G x,   G = 2*strict_lower(M)             -> the dense contraction (k_lmm_quadform_i8)
// x is exactly representable in int8, and G is held as L balanced base-256 int8 limbs of a fixed-point number, so the
// contraction runs on v_mfma_i32_32x32x32_i8 with EXACT int32 accumulation; limbs are recombined in fp64.
#include <utility>
#include "common.h"
#include "lmm_params.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// bits (variant-major rows) -> T[sb][v] : one uint64 per (64-sample block, variant); coalesced for every consumer.
// ---------------------------------------------------------------------------------------------
// Complemented rows (LMM only): with the intercept in the covariate span, U~^T 1 = 0, so x and 1 - x have the same x^T M x and opposite
// x . v.  A row with more than N/2 carriers is stored complemented (flip[v] = 1): at most half of the bytes of the variant operand are then
// non-zero (less switching in the matrix pipe = higher sustained clock), and the quadratic form no longer cancels for allele
// frequencies near 1 (sum_i x_i M_ii and x^T G x were both ~ m c for a result ~ (N - m) c).  k_lmm_finalize undoes the flip in the
// counts and the sign of x^T K^-1 y.  flip == nullptr: rows are stored as given (fixed effects, de-duplication, similarity).

// fallback for rows too long to stage 64 of them in 64 KB of LDS (more than 8192 samples): strided 8-byte gather per thread
__global__ __launch_bounds__(256) void k_row_flip(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V, int N, uint8_t *__restrict__ flip)
{
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // one wavefront per row
    const int lane = threadIdx.x & 63;
    if (v >= V) return;
    const uint8_t *p = bits + v * row_bytes;
    const int nb = N >> 3;
    int m = 0;
    for (int b = lane; b < nb; b += 64) m += __popc((unsigned)p[b]);
    if (lane == 0 && (N & 7)) m += __popc((unsigned)p[nb] & ((1u << (N & 7)) - 1u));
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    if (lane == 0) flip[v] = (2 * m > N) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_repack_bits_gather(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V,
                                                            int64_t Vpad, int N, uint64_t *__restrict__ T, const uint8_t *__restrict__ flip)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int sb = blockIdx.y;
    uint64_t w = 0;
    if (v < V) {
        const int64_t off = (int64_t)sb * 8;
        const uint8_t *p = bits + v * row_bytes + off;
        int64_t nbytes = row_bytes - off; if (nbytes > 8) nbytes = 8;
        if (nbytes == 8 && ((reinterpret_cast<uintptr_t>(p) & 7) == 0)) w = *reinterpret_cast<const uint64_t *>(p);
        else for (int b = 0; b < nbytes; b++) w |= (uint64_t)p[b] << (8 * b);
        if (flip && flip[v]) w = ~w;
        const int valid = N - sb * 64;
        if (valid <= 0) w = 0; else if (valid < 64) w &= ((1ull << valid) - 1ull);
    }
    T[(int64_t)sb * Vpad + v] = w;
}

__global__ __launch_bounds__(256) void k_repack_bits(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V,
                                                     int64_t Vpad, int N, int NB64p, uint64_t *__restrict__ T, uint8_t *__restrict__ flip)
{
    // One block = 64 consecutive variants.  Their rows are one contiguous span of 64 * row_bytes bytes: it is copied to LDS
    // with coalesced 16-byte loads, then wave w writes the 64-sample words sb = w, w+4, ... (one variant per lane, 512 contiguous
    // bytes per store).  A per-thread strided 8-byte gather of the same data runs at a fifth of this.
    extern __shared__ __attribute__((aligned(16))) uint8_t rp_rows[];
    __shared__ int rp_cnt[4][64];
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nrows = min((int64_t)64, V - v0);
    const int64_t span = nrows > 0 ? nrows * row_bytes : 0;
    const uint8_t *src = bits + v0 * row_bytes;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int64_t n16 = span >> 4;
        for (int64_t i = tid; i < n16; i += 256) reinterpret_cast<uint4 *>(rp_rows)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (int64_t i = (n16 << 4) + tid; i < span; i += 256) rp_rows[i] = src[i];
    } else {
        for (int64_t i = tid; i < span; i += 256) rp_rows[i] = src[i];
    }
    __syncthreads();
    auto word = [&](int sb) -> uint64_t {
        uint64_t w = 0;
        const int64_t off = (int64_t)sb * 8;
        if (lane < nrows && off < row_bytes) {
            const uint8_t *p = rp_rows + lane * row_bytes + off;
            int64_t nbytes = row_bytes - off; if (nbytes > 8) nbytes = 8;
            if (nbytes == 8 && ((row_bytes & 7) == 0)) w = *reinterpret_cast<const uint64_t *>(p);
            else for (int b = 0; b < nbytes; b++) w |= (uint64_t)p[b] << (8 * b);
            const int valid = N - sb * 64;
            if (valid <= 0) w = 0; else if (valid < 64) w &= ((1ull << valid) - 1ull);
        }
        return w;
    };
    bool fl = false;
    if (flip) {                                                         // block-uniform
        int m = 0;
        for (int sb = wave; sb < NB64p; sb += 4) m += __popcll(word(sb));
        rp_cnt[wave][lane] = m;
        __syncthreads();
        m = rp_cnt[0][lane] + rp_cnt[1][lane] + rp_cnt[2][lane] + rp_cnt[3][lane];
        fl = lane < nrows && 2 * m > N;
        if (wave == 0 && lane < nrows) flip[v0 + lane] = fl ? 1 : 0;
    }
    for (int sb = wave; sb < NB64p; sb += 4) {
        uint64_t w = word(sb);
        if (fl) {
            const int valid = N - sb * 64;
            w = ~w;
            if (valid <= 0) w = 0; else if (valid < 64) w &= ((1ull << valid) - 1ull);
        }
        T[(int64_t)sb * Vpad + v0 + lane] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// O(N) per-variant terms: table counts (a1), x.v, x.diag(M), Welch sums, covariate-residual variance (rotate's
// "explained by the covariates" test, lmm_cov.py:179-181).  One variant per lane; sample loop is wave-uniform so the
// per-sample constants arrive through the scalar cache.
//   Qb: N x DP row-major orthonormal basis of the covariate space (only when D > 1; DP = D padded), else DP = 0.
// ---------------------------------------------------------------------------------------------

template <int DP>
__global__ __launch_bounds__(256) void k_lmm_linear(const uint64_t *__restrict__ T, int64_t Vpad, int N, int NB64,
                                                    const double *__restrict__ vv, const double *__restrict__ mdiag,
                                                    const double *__restrict__ yc, const double *__restrict__ Qb,
                                                    const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                    int continuous, LmmLinOut o)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double xky = 0, dg = 0, s1 = 0, q1 = 0;
    double c[DP > 0 ? DP : 1];
#pragma unroll
    for (int d = 0; d < DP; d++) c[d] = 0;
    int t11 = 0, t01 = 0, m = 0;
    for (int sb = 0; sb < NB64; sb++) {
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        m += __popcll(w); t11 += __popcll(w & y1[sb]); t01 += __popcll(w & y0[sb]);
        const int nb = min(64, N - sb * 64);
        for (int b = 0; b < nb; b++) {
            const int i = sb * 64 + b;
            const double xd = (double)(unsigned)((w >> b) & 1ull);
            xky = fma(xd, vv[i], xky);
            dg = fma(xd, mdiag[i], dg);
            if (continuous) { const double t = yc[i]; s1 = fma(xd, t, s1); q1 = fma(xd, t * t, q1); }
#pragma unroll
            for (int d = 0; d < DP; d++) c[d] = fma(xd, Qb[(int64_t)i * DP + d], c[d]);
        }
    }
    double rss;
    if (DP > 0) {
        rss = 0;
        for (int sb = 0; sb < NB64; sb++) {
            const uint64_t w = T[(int64_t)sb * Vpad + v];
            const int nb = min(64, N - sb * 64);
            for (int b = 0; b < nb; b++) {
                const int i = sb * 64 + b;
                double r = (double)(unsigned)((w >> b) & 1ull);
#pragma unroll
                for (int d = 0; d < DP; d++) r = fma(-Qb[(int64_t)i * DP + d], c[d], r);
                rss = fma(r, r, rss);
            }
        }
    } else {
        rss = (double)m * (double)(N - m) / (double)N;           // ||x - mean(x)||^2, exact
    }
    o.t11[v] = t11; o.t01[v] = t01; o.m[v] = m;
    o.xky[v] = xky; o.dg[v] = dg; o.rss[v] = rss; o.s1[v] = s1; o.q1[v] = q1;
}

// ---------------------------------------------------------------------------------------------
// The same O(N) terms for the common case (D = 1, binary phenotype) by table lookup: for every 4-sample nibble of a variant's
// word, tab[sb][nib][value] = (sum of vv, sum of mdiag) over the set bits, built once per run (k_lmm_build_tab, 4 KB per
// 64-sample block).  16 conflict-free ds_read_b128 + 32 fp64 adds per word instead of 64 x (extract, convert, 2 FMAs).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lmm_build_tab(const double *__restrict__ vv, const double *__restrict__ mdiag, int N, int NB64,
                                                       double2 *__restrict__ tab)
{
    const int e = blockIdx.x * 256 + threadIdx.x;              // entry = (sb * 16 + nib) * 16 + value
    if (e >= NB64 * 256) return;
    const int value = e & 15, base = (e >> 4) * 4;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 4; ++k)
        if (((value >> k) & 1) && base + k < N) { a += vv[base + k]; b += mdiag[base + k]; }
    tab[e] = make_double2(a, b);
}

__global__ __launch_bounds__(256) void k_lmm_linear_tab(const uint64_t *__restrict__ T, int64_t Vpad, int N, int NB64,
                                                        const double2 *__restrict__ tab, const uint64_t *__restrict__ y1,
                                                        const uint64_t *__restrict__ y0, LmmLinOut o)
{
    __shared__ double2 lt[2][256];
    const int tid = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;
    double xky = 0, dg = 0;
    int t11 = 0, t01 = 0, m = 0;
    lt[0][tid] = tab[tid];
    for (int sb = 0; sb < NB64; sb++) {
        const double2 nxt = (sb + 1 < NB64) ? tab[(sb + 1) * 256 + tid] : make_double2(0.0, 0.0);
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        __syncthreads();                                          // slice sb is in lt[sb & 1]; nobody reads lt[(sb+1) & 1] any more
        m += __popcll(w); t11 += __popcll(w & y1[sb]); t01 += __popcll(w & y0[sb]);
        const double2 *cur = lt[sb & 1];
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
#pragma unroll
        for (int nib = 0; nib < 8; ++nib) {
            const double2 e0 = cur[nib * 16 + ((lo >> (4 * nib)) & 15u)];
            const double2 e1 = cur[(8 + nib) * 16 + ((hi >> (4 * nib)) & 15u)];
            xky += e0.x; dg += e0.y; xky += e1.x; dg += e1.y;
        }
        lt[(sb + 1) & 1][tid] = nxt;
    }
    o.t11[v] = t11; o.t01[v] = t01; o.m[v] = m;
    o.xky[v] = xky; o.dg[v] = dg; o.rss[v] = (double)m * (double)(N - m) / (double)N; o.s1[v] = 0.0; o.q1[v] = 0.0;
}

// The general form of the table kernel: NE doubles per entry = (vv, diag M) [+ (yc, yc^2) for the Welch prefilter of a
// continuous phenotype] [+ the DP columns of the covariate basis Qb].  rss = ||x - Qb Qb^T x||^2 only feeds the "explained by the
// covariates" test (std <= 1e-10, lmm_cov.py:179-181): it is m - sum_d c_d^2 in exact arithmetic, and only a wavefront that holds
// a variant where that difference cancels (<= 1e-8 m) walks the samples again for the exact residual.
template <int NE>
__global__ __launch_bounds__(256) void k_lmm_build_tabn(const double *__restrict__ vv, const double *__restrict__ mdiag,
                                                        const double *__restrict__ yc, const double *__restrict__ Qb, int DP, int cont,
                                                        int N, int NB64, double *__restrict__ tab)
{
    const int e = blockIdx.x * 256 + threadIdx.x;              // entry = (sb * 16 + nib) * 16 + value
    if (e >= NB64 * 256) return;
    const int value = e & 15, base = (e >> 4) * 4;
    for (int c = 0; c < NE; ++c) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) {
            const int i = base + k;
            if (!((value >> k) & 1) || i >= N) continue;
            double x;
            if (c == 0) x = vv[i]; else if (c == 1) x = mdiag[i];
            else if (cont && c == 2) x = yc[i]; else if (cont && c == 3) x = yc[i] * yc[i];
            else x = Qb[(int64_t)i * DP + (c - 2 - 2 * cont)];
            a += x;
        }
        tab[(int64_t)e * NE + c] = a;
    }
}

template <int NE>
__global__ __launch_bounds__(256) void k_lmm_linear_tabn(const uint64_t *__restrict__ T, int64_t Vpad, int N, int NB64,
                                                         const double *__restrict__ tab, const uint64_t *__restrict__ y1,
                                                         const uint64_t *__restrict__ y0, const double *__restrict__ Qb, int DP,
                                                         int cont, LmmLinOut o)
{
    __shared__ __attribute__((aligned(16))) double lt[2][256 * NE];
    const int tid = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;
    double acc[NE];
#pragma unroll
    for (int c = 0; c < NE; ++c) acc[c] = 0.0;
    int t11 = 0, t01 = 0, m = 0;
#pragma unroll
    for (int c = 0; c < NE; ++c) lt[0][tid * NE + c] = tab[(int64_t)tid * NE + c];
    for (int sb = 0; sb < NB64; sb++) {
        double nxt[NE];
        const bool more = sb + 1 < NB64;
#pragma unroll
        for (int c = 0; c < NE; ++c) nxt[c] = more ? tab[((int64_t)(sb + 1) * 256 + tid) * NE + c] : 0.0;
        const uint64_t w = T[(int64_t)sb * Vpad + v];
        __syncthreads();                                          // slice sb is in lt[sb & 1]; nobody reads lt[(sb+1) & 1] any more
        m += __popcll(w); t11 += __popcll(w & y1[sb]); t01 += __popcll(w & y0[sb]);
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
    const int c0 = 2 + 2 * cont;
    double rss;
    if (DP > 0) {
        double sc = 0.0;
        for (int d = 0; d < NE - c0; ++d) sc = fma(acc[c0 + d], acc[c0 + d], sc);
        rss = (double)m - sc;
        const bool susp = rss <= 1e-8 * fmax((double)m, 1.0);
        if (__any(susp)) {
            double ex = 0.0;
            for (int sb = 0; sb < NB64; sb++) {
                const uint64_t w = T[(int64_t)sb * Vpad + v];
                const int nb = min(64, N - sb * 64);
                for (int b = 0; b < nb; b++) {
                    const int i = sb * 64 + b;
                    double r = (double)(unsigned)((w >> b) & 1ull);
                    for (int d = 0; d < NE - c0; ++d) r = fma(-Qb[(int64_t)i * DP + d], acc[c0 + d], r);
                    ex = fma(r, r, ex);
                }
            }
            if (susp) rss = ex;
        }
    } else {
        rss = (double)m * (double)(N - m) / (double)N;           // ||x - mean(x)||^2, exact
    }
    o.t11[v] = t11; o.t01[v] = t01; o.m[v] = m;
    o.xky[v] = acc[0]; o.dg[v] = acc[1]; o.rss[v] = rss;
    o.s1[v] = cont ? acc[2] : 0.0; o.q1[v] = cont ? acc[3] : 0.0;
}

template __global__ void k_lmm_linear<0>(const uint64_t *, int64_t, int, int, const double *, const double *, const double *, const double *, const uint64_t *, const uint64_t *, int, LmmLinOut);
template __global__ void k_lmm_linear<4>(const uint64_t *, int64_t, int, int, const double *, const double *, const double *, const double *, const uint64_t *, const uint64_t *, int, LmmLinOut);
template __global__ void k_lmm_linear<8>(const uint64_t *, int64_t, int, int, const double *, const double *, const double *, const double *, const uint64_t *, const uint64_t *, int, LmmLinOut);
template __global__ void k_lmm_linear<16>(const uint64_t *, int64_t, int, int, const double *, const double *, const double *, const double *, const uint64_t *, const uint64_t *, int, LmmLinOut);
template __global__ void k_lmm_linear<32>(const uint64_t *, int64_t, int, int, const double *, const double *, const double *, const double *, const uint64_t *, const uint64_t *, int, LmmLinOut);

// ---------------------------------------------------------------------------------------------
// THE hot kernel:  q[v] = sum_{i>j} x_i x_j Gq_ij   (exact integers; Gq = fixed-point 2*M_ij in L int8 limbs)
//
// Block = 512 variants x one 128-sample row tile at a time; 512 threads = 8 waves, wave w owns variants [64w, 64w+64) and
// ALL 128 rows of the current row tile: a 128 x 64 int32 accumulator = 4 x 2 MFMA 32x32 tiles = 128 VGPRs.
// For every (row tile I, limb l) the K loop runs over sample columns j < (I+1)*128 (G is strictly lower triangular; tiles
// above the diagonal are never visited), then the accumulator is folded against the variants' own bits for the rows of
// tile I (x_i * (G x)_i) into an fp64 running sum with weight 256^l.
//
//  * A operand (G): 128 x 64 int8 tiles stored in HBM in LDS-image order: byte (r, c) at r*64 + 16*((c>>4) ^ ((r>>2)&3)) + (c&15),
//    so (a) the HBM->LDS copy is a linear LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write) and (b) the
//    16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots (no bank conflicts).  A stage = two consecutive tiles
//    (128 deep in K, 16 KB), double-buffered; every G byte fetched feeds 512 variants.
//  * B operand (variants): never touches LDS.  Each lane expands its own fragment (16 samples x 1 variant -> 16 int8) from the
//    64-bit word of packed bits it loaded for that K tile: ((bits4 * 0x00204081) & 0x01010101) per 4 samples.
//  * The (I, l, stage) nest is flattened into one stream with the next stage's DMA always in flight, so segment boundaries
//    cost an epilogue, not a pipeline refill.
//
// History (V = 262144, N = 5000, L = 5; ms per launch): v1 LDS-staged B + register-staged A, 256x256 block: 20.3 ;
// v2 register B + LDS-DMA A: 14.9 ; v3 128-deep stages: 14.0 (ablation: DMA 2.5 ms, bit expansion 0.8 ms, MFMA-only 9.7 ms) ;
// v4 128 x 512 block + flattened (I,l,stage) stream, still one stage of DMA lookahead: 16.1 (LDS-DMA latency ~1.1 us > a stage) ;
// v5 = this kernel: 4-slot LDS ring, every load in the loop is an LDS-DMA (G tiles AND the packed bits), counted vmcnt + raw
// s_barrier so three stages of DMA stay in flight across every barrier: 14.3 (ablation: DMA 3.2, LDS fragment reads 1.8, bit
// expansion 1.2, MFMA-only 9.65; `nt` on the bits DMA -4 %, s_setprio around the MFMAs +-0).
// ---------------------------------------------------------------------------------------------
#define QF_TILE_BYTES 8192          // 128 rows x 64 int8
#define QF_BN 512                   // variants per block

__device__ __forceinline__ int qf_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ uint4 qf_expand16(uint32_t bits16)
{
    uint4 r;
    r.x = (((bits16) & 0xFu) * 0x00204081u) & 0x01010101u;
    r.y = (((bits16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
    r.z = (((bits16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
    r.w = (((bits16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
    return r;
}

#ifndef QF_G_AUX
#define QF_G_AUX 0
#endif
#ifndef QF_JT
#define QF_JT 2                      // 32-variant sub-tiles per wave: 2 (8 waves, two per SIMD) or 4 (4 waves, one per SIMD:
                                     // 256 accumulators in AGPRs halve the LDS fragment reads per MFMA, but hipcc then spills 187
                                     // VGPRs inside the MFMA loop -- correct, 5x slower; kept as a build option, not used)
#endif
#define QF_WAVES (16 / QF_JT)
#define QF_DMA_PER_WAVE (24 / QF_WAVES)  // LDS-DMA wave-instructions per stage per wave: 16 pieces of G + 8 of packed bits
#ifndef QF_DIAG_TRIM
#define QF_DIAG_TRIM 0               // skip the MFMAs of the all-zero quarter of a diagonal block: 1.2 % fewer MFMAs, but the branch costs the
#endif                               // software pipeline more (45.1 vs 43.6 ms per 2^20 variants, same box); kept as a build option
#ifndef QF_NST
#define QF_NST 5
#endif                               // LDS ring depth (stages): s (computing), s+1 (landed, prefetched from), s+2..s+4 in flight
#define QF_AHEAD (QF_NST - 1)        // stages issued ahead of the compute cursor
#define QF_STAGE_BYTES (2 * QF_TILE_BYTES + 2 * QF_BN * 8)      // 16 KB of G + 8 KB of packed bits = 24 KB

// one stage HBM/L2 -> LDS by LDS-DMA, 3 wave-instructions of 1 KB per wave: two for the G tiles, one for the block's packed
// bits of the two 64-sample blocks this stage spans.
__device__ __forceinline__ void qf_dma_stage(const int8_t *gsrc, const uint64_t *tsrc, int64_t Vpad, char *slot, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < 16 / QF_WAVES; ++j) {
        const int piece = wave * (16 / QF_WAVES) + j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(slot + piece * 1024), 16, 0, QF_G_AUX);
    }
    // bits: 8 pieces of 1 KB = (row of the stage's two sample blocks) x (128 variants of the block), 2 words per lane
#pragma unroll
    for (int j = 0; j < 8 / QF_WAVES; ++j) {
        const int pb = wave * (8 / QF_WAVES) + j;
        const uint64_t *src = tsrc + (int64_t)(pb >> 2) * Vpad + (pb & 3) * 128 + lane * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(slot + 2 * QF_TILE_BYTES + pb * 1024), 16, 0, 0);
    }
}

template <int N> __device__ __forceinline__ void qf_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL: timing ablations (results meaningless): 1 = no DMA, 2 = no LDS fragment reads, 4 = no bit expansion
template <int ABL>
__global__ __launch_bounds__(64 * QF_WAVES, QF_JT == 2 ? 2 : 1) void k_lmm_quadform_i8(const int8_t *__restrict__ G, const uint64_t *__restrict__ T,
                                                            int64_t Vpad, int NR, int L, int lsplit, double *__restrict__ qout,
                                                            const int *__restrict__ nlimit)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];       // QF_NST slots x 24 KB (the ONLY LDS object)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // Block -> (variant tile, limb group).  With lsplit = L every block owns ONE limb of one 512-variant tile and the L blocks of
    // a tile are neighbours on the same XCD (block b runs on XCD b % 8): they re-read the same packed bits, which then stay in that
    // XCD's L2 instead of being re-fetched from the Infinity Cache for every (row tile, limb) segment.
    const int ntiles = (int)(Vpad / QF_BN);
    int tile, lgrp;
    if ((ntiles & 7) == 0) { const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3; tile = (w / lsplit) * 8 + xcd; lgrp = w % lsplit; }
    else { tile = blockIdx.x / lsplit; lgrp = blockIdx.x % lsplit; }
    const int nl = (L - lgrp + lsplit - 1) / lsplit;                    // limbs handled here: lgrp, lgrp + lsplit, ...
    const int64_t v0 = (int64_t)tile * QF_BN;
    if (nlimit && v0 >= (int64_t)*nlimit) return;                       // extra-limb pass: the launch is sized for the whole batch, the list is short
    const bool big = NR * 128 > 65535;                                  // |acc| can reach 2^23: no 24-bit multiply in the epilogue
    const int64_t TL = (int64_t)NR * (NR + 1);                          // tiles per limb: sum_I 2(I+1)
    const int total = nl * (NR * (NR + 1) / 2);                         // stages in the flattened stream
    int aoff[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) { aoff[it][0] = qf_off(it * 32 + l31, lh); aoff[it][1] = qf_off(it * 32 + l31, 2 + lh); }
    const int boff = 2 * QF_TILE_BYTES + (wave * (32 * QF_JT) + l31) * 8;   // this lane's first variant column in the bits area

    double tot[QF_JT];
#pragma unroll
    for (int jt = 0; jt < QF_JT; ++jt) tot[jt] = 0.0;
    v16i acc[4][QF_JT];
    int pI = 0, pl = 0, pst = 0, pslot = 0;                              // DMA cursor (QF_NST-1 stages ahead)
    int cI = 0, cl = 0, cst = 0;                                        // compute cursor
    double scale0 = 1.0, step = 1.0;
    for (int i = 0; i < lgrp; ++i) scale0 *= 256.0;
    for (int i = 0; i < lsplit; ++i) step *= 256.0;
    double scale_l = scale0;

    auto fetch = [&]() {
        const int8_t *g = G + ((int64_t)(lgrp + pl * lsplit) * TL + (int64_t)pI * (pI + 1) + 2 * pst) * QF_TILE_BYTES;
        const uint64_t *t = T + (int64_t)(2 * pst) * Vpad + v0;
        if (ABL & 8) { g = G; t = T + v0; }                          // ablation: every stage re-reads the same 24 KB (cache-resident)
        if (!(ABL & 1)) qf_dma_stage(g, t, Vpad, smem + pslot * QF_STAGE_BYTES, wave, lane);
        if (++pslot == QF_NST) pslot = 0;
        if (++pst == pI + 1) { pst = 0; if (++pl == nl) { pl = 0; ++pI; } }
    };

#pragma unroll 1
    for (int s = 0; s < QF_AHEAD && s < total; ++s) fetch();

    // barrier_s: every wave's pieces of stages <= s+1 have landed and nobody reads stage s-1 any more, so its slot is refilled
    // with stage s + QF_AHEAD.  Landing one stage EARLY lets the last sub-step of stage s prefetch the first fragments and bit
    // words of stage s+1, so no wave starts a stage by waiting on LDS.  (Tried and measured slower, +1.4 % / +3 %: taking the
    // barrier half a stage apart on the two waves that share a SIMD; the stalls are not a phase-alignment effect.  The barrier
    // itself: a timing build without it (SEERHIP_ROUTE qf=46) runs 4.7 % faster, but one barrier per TWO stages over a 6-slot ring gains
    // only 0.25 %: what costs is the waves waiting for each other, not the instruction.)
    auto sync_refill = [&](int s) {
        const int newer = max(0, min(total - 2 - s, QF_AHEAD - 2));   // stages allowed to be still in flight: s+2 .. s+QF_AHEAD-1
        if (!(ABL & 1)) {
            if (newer >= 3) qf_wait_vm<3 * QF_DMA_PER_WAVE>(); else if (newer == 2) qf_wait_vm<2 * QF_DMA_PER_WAVE>();
            else if (newer == 1) qf_wait_vm<QF_DMA_PER_WAVE>(); else qf_wait_vm<0>();
        }
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();            // ABL 16: timing ablation without the barrier (results garbage)
        __builtin_amdgcn_sched_barrier(0);
        if (s + QF_AHEAD < total) fetch();
    };
    auto read_bits = [&](const char *base, uint64_t (&w)[2][QF_JT]) {
#pragma unroll
        for (int jt = 0; jt < QF_JT; ++jt) {
            w[0][jt] = *reinterpret_cast<const uint64_t *>(base + boff + 256 * jt);
            w[1][jt] = *reinterpret_cast<const uint64_t *>(base + boff + QF_BN * 8 + 256 * jt);
        }
    };
    // Variant fragment of sub-step (tile tl, k-half kh): lane half lh owns samples 32 lh .. 32 lh + 31 of the tile's 64-sample word, and
    // dword d of its fragment is ((half >> (4 kh + d)) & 0x01010101), i.e. byte b of dword d = sample 32 lh + 8 b + 4 kh + d.  The G tiles are
    // stored with the same permutation of their 64 columns (k_lmm_quantize), so a fragment costs 7-8 VALU (shift + and) instead of the 13 of
    // a nibble multiply ((bits4 * 0x00204081) & 0x01010101 per dword).
    auto expand = [&](const uint64_t (&w)[2][QF_JT], int sub, v4i (&b)[QF_JT]) {
        const int tl = sub >> 1, kh = sub & 1;
#pragma unroll
        for (int jt = 0; jt < QF_JT; ++jt) {
            const uint32_t h = lh ? (uint32_t)(w[tl][jt] >> 32) : (uint32_t)w[tl][jt];
            uint4 e;
            if (ABL & 4) e = make_uint4(h, kh, jt, 1);
            else e = make_uint4((h >> (4 * kh)) & 0x01010101u, (h >> (4 * kh + 1)) & 0x01010101u,
                                (h >> (4 * kh + 2)) & 0x01010101u, (h >> (4 * kh + 3)) & 0x01010101u);
            b[jt] = (v4i){(int)e.x, (int)e.y, (int)e.z, (int)e.w};
        }
    };

    // Software pipeline over the 32-deep sub-steps (sub = tile(0/1) * 2 + k-half(0/1)), continuous ACROSS stages: the A fragments
    // and the expanded variant fragments of the next sub-step are produced while the 8 MFMAs of the current one issue;
    // sched_barrier(0) closes each sub-step so that hipcc cannot sink the LDS reads next to their consumers.
    uint64_t wb[2][QF_JT], wbn[2][QF_JT];
    v4i a_cur[4], a_nxt[4], b_cur[QF_JT], b_nxt[QF_JT];
    sync_refill(0);
    read_bits(smem, wb);
#pragma unroll
    for (int it = 0; it < 4; ++it) a_cur[it] = (ABL & 2) ? (v4i){it, lane, 0, 1} : *reinterpret_cast<const v4i *>(smem + aoff[it][0]);
    expand(wb, 0, b_cur);

    int slot = 0;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
        if (s > 0) sync_refill(s);
        if ((ABL & 256) ? s == 0 : cst == 0) {                     // ABL 256: timing ablation without the per-segment zeroing
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < QF_JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0;
        }
        const char *a_base = smem + slot * QF_STAGE_BYTES;
        const char *n_base = smem + (slot + 1 == QF_NST ? 0 : slot + 1) * QF_STAGE_BYTES;     // stage s+1 (landed as of barrier_s)
        const bool diag = (cst == cI);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub < 3) {
                const int nt = (sub + 1) >> 1, nk = (sub + 1) & 1;
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    a_nxt[it] = (ABL & 2) ? (v4i){it, lane, nt, nk}
                                          : *reinterpret_cast<const v4i *>(a_base + nt * QF_TILE_BYTES + aoff[it][nk]);
                if (sub == 2) read_bits(n_base, wbn);               // garbage (never used) behind the last stage
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    a_nxt[it] = (ABL & 2) ? (v4i){it, lane, s, 1} : *reinterpret_cast<const v4i *>(n_base + aoff[it][0]);
            }
            // the last stage of a segment is the diagonal block of strictly-lower G: its second 64 columns are zero for rows 0..63
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (QF_DIAG_TRIM && sub >= 2 && it < 2 && diag) continue;
#pragma unroll
                for (int jt = 0; jt < QF_JT; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[it], b_cur[jt], acc[it][jt], 0, 0, 0);
            }
            if (sub < 3) expand(wb, sub + 1, b_nxt); else expand(wbn, 0, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 4; ++it) a_cur[it] = a_nxt[it];
#pragma unroll
            for (int jt = 0; jt < QF_JT; ++jt) b_cur[jt] = b_nxt[jt];
        }
        if (++slot == QF_NST) slot = 0;
        if (++cst == cI + 1) {
            // ---- segment epilogue: sum_i x_i * acc_i over the 128 rows of tile cI, per variant column.  The last stage of a
            // segment spans exactly the row tile's own samples, so its bit words ARE the epilogue mask.  |acc| <= 128 * N < 2^23 for
            // N <= 65535, so the masked sum is a chain of 24-bit multiply-adds by the 0/1 bit (v_bfe_u32 + v_mad_i32_i24 per register);
            // the partial sum of 64 of them, |part| <= 2^29 (N <= 65535) or <= 64 * 128 * N < 2^31 for N < 2^18, stays in int32.
#pragma unroll
            for (int jt = 0; jt < QF_JT; ++jt) {
                int part[4];
                if (ABL & 128) { tot[jt] += (double)(acc[0][jt][0] + acc[1][jt][1] + acc[2][jt][2] + acc[3][jt][3]); continue; }   // timing ablation: no epilogue
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const uint32_t w = (uint32_t)(wb[it >> 1][jt] >> ((it & 1) * 32 + 4 * lh));
                    part[it] = 0;
                    if (!big) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2);            // C/D layout of the 32x32 MFMA (+ 4*lh, pre-shifted)
                            const int bit = (int)__builtin_amdgcn_ubfe(w, row, 1);
                            asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(part[it]) : "v"(acc[it][jt][r]), "v"(bit));
                        }
                    } else {                                                   // more than 65535 samples: |acc| <= 128 N needs the full 32 bits
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2);
                            part[it] += ((w >> row) & 1u) ? acc[it][jt][r] : 0;
                        }
                    }
                }
                tot[jt] = fma((double)((part[0] + part[1]) + (part[2] + part[3])), scale_l, tot[jt]);
            }
            cst = 0; scale_l *= step;
            if (++cl == nl) { cl = 0; scale_l = scale0; ++cI; }
        }
#pragma unroll
        for (int jt = 0; jt < QF_JT; ++jt) { wb[0][jt] = wbn[0][jt]; wb[1][jt] = wbn[1][jt]; }
    }
    // lanes l and l^32 hold different rows of the same variant
#pragma unroll
    for (int jt = 0; jt < QF_JT; ++jt) tot[jt] += __shfl_xor(tot[jt], 32, 64);
    if (lh == 0) {
        double *qo = qout + (int64_t)lgrp * Vpad + v0 + wave * (32 * QF_JT) + l31;
#pragma unroll
        for (int jt = 0; jt < QF_JT; ++jt) qo[32 * jt] = tot[jt];
    }
}

#include "lmm_quadform_wide.inc"

// ---------------------------------------------------------------------------------------------
// Per-variant finalisation: a1 prefilter + A5 statistics + a7 filters (pyseer/lmm.py:160-217, 244-258).
// ---------------------------------------------------------------------------------------------

// the linear terms of variant v with the complement of k_repack_bits undone: counts and Welch sums of the ORIGINAL row, x . v of the
// original row (= sum v - x' . v); dg, rss and the quadratic form belong to the stored row (x'^T M x' = x^T M x when M 1 = 0)
struct LinV { int m, t11, t01, mst; double xky, dg, rss, s1, q1; };
__device__ __forceinline__ LinV lin_load(const LmmLinOut &li, const LmmFinParams &P, int64_t v)
{
    LinV r;
    r.mst = li.m[v]; r.m = r.mst; r.t11 = li.t11[v]; r.t01 = li.t01[v];
    r.xky = li.xky[v]; r.dg = li.dg[v]; r.rss = li.rss[v]; r.s1 = li.s1[v]; r.q1 = li.q1[v];
    if (li.flip && li.flip[v]) {
        r.m = P.N - r.mst; r.t11 = P.n1 - r.t11; r.t01 = P.n0 - r.t01;
        r.xky = P.sumv - r.xky; r.s1 = P.yc_sum - r.s1; r.q1 = P.yc_sq - r.q1;
    }
    return r;
}

// returns the relative bound on xKx of this variant (0 when it has no statistics); *want_refine: the bound exceeds P.tol
__device__ __forceinline__ double lmm_fin_one(int64_t v, int64_t V, int64_t Vpad, int nq, const LmmLinOut &li, const double *__restrict__ q,
                                              const LmmFinParams &P, double q_extra, bool refined, double *__restrict__ out,
                                              uint32_t *__restrict__ flags, bool *want_refine)
{
    const double nanv = NAN;
    double prep = nanv, pval = nanv, beta = nanv, bse = nanv, frac = nanv, bound = 0.0;
    uint32_t fl = 0;
    const LinV L = lin_load(li, P, v);
    const int m = L.m;
    bool go = true;
    *want_refine = false;
    if (P.af_on) {
        const double af = (double)m / (double)P.N;
        if (!(P.min_af <= af && af <= P.max_af)) { fl = SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER; go = false; }
    }
    if (go) {
        bool bad = false;
        if (P.continuous) {
            const double n1 = (double)m, n0 = (double)(P.N - m);
            prep = sh_prefilter_welch(n1, L.s1, L.q1, n0, P.yc_sum - L.s1, P.yc_sq - L.q1);
        } else {
            prep = sh_prefilter_binary(L.t11, P.n1 - L.t11, L.t01, P.n0 - L.t01, &bad);
        }
        if (bad) fl |= SH_NOTE_BAD_CHISQ;
        if (prep >= P.pret || !isfinite(prep)) fl |= SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER;      // lmm.py:174 (>=)
    }
    // With a prefilter requested (filter_pvalue < 1) a pre-filtered variant is never fitted (lmm.py:174-185): no statistics, and its quadratic
    // form is not even computed (k_af_keep).  Without one every AF-passing variant carries its statistics, as fit_lmm_block gives them; the
    // masking of the rows that only fail `prep >= 1` or have no finite prep is the caller's (lmm.py:176-217).
    if (go && !((fl & SH_FLAG_PREFILTER) && P.pret < 1.0)) {
        const bool zeroed = sqrt(L.rss / (double)P.N) <= 1e-10;                                  // lmm_cov.py:179-181
        double qs = 0.0;
        for (int a = 0; a < nq; ++a) qs += q[(int64_t)a * Vpad + v];          // per-limb-group partial sums, fixed order
        const double xKx = zeroed ? 0.0 : (L.dg + (qs * P.inv_scale + q_extra));
        const double xKy = zeroed ? 0.0 : L.xky;
        if (!zeroed && xKx > 0.0) {
            bound = (refined ? P.err_norm * (P.inv_scale_low / P.inv_scale) : P.err_norm) * (double)L.mst / xKx;
            if (!refined && P.tol > 0.0 && bound > P.tol) *want_refine = true;
        }
        double b = xKy / xKx;
        if (isnan(b) && xKy == 0.0) b = 0.0;                                                      // lmm_cov.py:802-805
        const double veb = xKy * b, r2 = P.yKy - veb;
        const double var = r2 / ((double)(P.N - P.D) - 1.0) / xKx;                                // lmm_cov.py:813
        const double chi2 = b * b / var;                                                          // lmm.py:248
        pval = sh_f_sf_1(chi2, (double)(P.N - (P.D + 1)));                                        // lmm.py:251-253
        beta = b; bse = sqrt(var); frac = sqrt(veb / P.yKy);
        if (!(fl & SH_FLAG_PREFILTER)) {
            if (pval >= P.lrtt || !isfinite(pval)) fl |= SH_NOTE_LRT_FILTER | SH_FLAG_FILTER;      // lmm.py:201 (>=)
        }
    }
    out[v] = prep; out[V + v] = pval; out[2 * V + v] = beta; out[3 * V + v] = bse; out[4 * V + v] = frac;
    flags[v] = fl;
    return bound;
}

__device__ __forceinline__ void lmm_bound_max(double bound, unsigned long long *bound_max)
{
    for (int o = 32; o > 0; o >>= 1) bound = fmax(bound, __shfl_xor(bound, o, 64));
    if ((threadIdx.x & 63) == 0 && bound > 0.0 && bound_max) atomicMax(bound_max, (unsigned long long)__double_as_longlong(bound));
}

__global__ __launch_bounds__(256) void k_lmm_finalize(int64_t V, int64_t Vpad, int nq, LmmLinOut li, const double *__restrict__ q,
                                                      LmmFinParams P, double *__restrict__ out, uint32_t *__restrict__ flags, LmmRefine R)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool ref = false;
    double bound = 0.0;
    if (v < V) bound = lmm_fin_one(v, V, Vpad, nq, li, q, P, 0.0, false, out, flags, &ref);
    // variants whose bound exceeds the tolerance are listed for the extra-limb pass (one atomic per wavefront); their bound is
    // recorded after that pass
    const unsigned long long rm = __ballot(ref);
    if (rm && R.list) {
        int base = 0;
        if (lane == 0) base = atomicAdd(R.count, __popcll(rm));
        base = __shfl(base, 0);
        if (ref) R.list[base + __popcll(rm & ((1ull << lane) - 1ull))] = (int)v;
    }
    lmm_bound_max((ref && R.list) ? 0.0 : bound, R.bound_max);
}

// the listed variants again, with the contribution of the extra limbs (q3: per-limb partial sums, column j of the gathered image)
__global__ __launch_bounds__(256) void k_lmm_refine_fix(int64_t V, int64_t Vpad, int nq, int nq3, LmmLinOut li, const double *__restrict__ q,
                                                        const double *__restrict__ q3, LmmFinParams P, double *__restrict__ out,
                                                        uint32_t *__restrict__ flags, LmmRefine R)
{
    const int n = *R.count;
    for (int64_t j0 = (int64_t)blockIdx.x * 256; j0 < n; j0 += (int64_t)gridDim.x * 256) {
        const int64_t j = j0 + threadIdx.x;
        double bound = 0.0;
        if (j < n) {
            const int64_t v = R.list[j];
            double qe = 0.0;
            for (int a = 0; a < nq3; ++a) qe += q3[(int64_t)a * Vpad + j];
            bool dummy;
            bound = lmm_fin_one(v, V, Vpad, nq, li, q, P, qe * P.inv_scale_low, true, out, flags, &dummy);
        }
        lmm_bound_max(bound, R.bound_max);
    }
}

// columns R.list[0 .. *R.count) of T -> dense image T3 (same column stride), zero columns up to the next multiple of 512
__global__ __launch_bounds__(256) void k_gather_T_list(const uint64_t *__restrict__ T, int64_t Vpad, uint64_t *__restrict__ T3, LmmRefine R)
{
    const int n = *R.count;
    const int64_t npad = ((int64_t)n + 511) / 512 * 512;
    const int64_t sb = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < npad; j += (int64_t)gridDim.x * 256)
        T3[sb * Vpad + j] = j < n ? T[sb * Vpad + R.list[j]] : 0ull;
}

// ---- AF compaction: the quadratic form is the whole cost of a variant and an AF-filtered variant never reads it (k_lmm_finalize),
// so when enough of a batch is filtered the kept columns of T are gathered into a dense image and only those are contracted.
__global__ __launch_bounds__(256) void k_af_keep(int64_t V, LmmLinOut li, LmmFinParams P, int *__restrict__ idx, int *__restrict__ n)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool keep = false;
    if (v < V) {                                                   // the decisions of k_lmm_finalize, on the same numbers
        const LinV L = lin_load(li, P, v);
        const int m = L.m;
        keep = true;
        if (P.af_on) { const double af = (double)m / (double)P.N; keep = (P.min_af <= af && af <= P.max_af); }
        if (keep && P.pret < 1.0) {
            bool bad = false; double prep;
            if (P.continuous) prep = sh_prefilter_welch((double)m, L.s1, L.q1, (double)(P.N - m), P.yc_sum - L.s1, P.yc_sq - L.q1);
            else prep = sh_prefilter_binary(L.t11, P.n1 - L.t11, L.t01, P.n0 - L.t01, &bad);
            keep = !(prep >= P.pret || !isfinite(prep));
        }
    }
    const unsigned long long km = __ballot(keep);                          // one atomic per wavefront, not per variant
    int base = 0;
    if (lane == 0 && km) base = atomicAdd(n, __popcll(km));
    base = __shfl(base, 0);
    if (keep) idx[base + __popcll(km & ((1ull << lane) - 1ull))] = (int)v;   // any order of wavefronts: results are scattered back
}

__global__ __launch_bounds__(256) void k_gather_T(const uint64_t *__restrict__ T, int64_t Vpad, uint64_t *__restrict__ T2, int64_t Vpad2,
                                                  const int *__restrict__ idx, int nk)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= Vpad2) return;
    const int64_t sb = blockIdx.y;
    T2[sb * Vpad2 + j] = j < nk ? T[sb * Vpad + idx[j]] : 0ull;            // padding columns: empty variants
}

__global__ __launch_bounds__(256) void k_scatter_q(const double *__restrict__ q2, int64_t Vpad2, double *__restrict__ q, int64_t Vpad,
                                                   const int *__restrict__ idx, int nk)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nk) return;
    q[(int64_t)blockIdx.y * Vpad + idx[j]] = q2[(int64_t)blockIdx.y * Vpad2 + j];
}

extern "C" hipError_t shk_af_compact(hipStream_t st, int which, int64_t V, LmmLinOut li, LmmFinParams P, int *idx, int *n,
                                     const uint64_t *T, int64_t Vpad, uint64_t *T2, int64_t Vpad2, int rows, int nk,
                                     const double *q2, double *q)
{
    if (which == 0) hipLaunchKernelGGL(k_af_keep, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, li, P, idx, n);
    else if (which == 1) hipLaunchKernelGGL(k_gather_T, dim3((unsigned)((Vpad2 + 255) / 256), (unsigned)rows), dim3(256), 0, st, T, Vpad, T2, Vpad2, idx, nk);
    else hipLaunchKernelGGL(k_scatter_q, dim3((unsigned)((nk + 255) / 256), (unsigned)rows), dim3(256), 0, st, q2, Vpad2, q, Vpad, idx, nk);
    return hipGetLastError();
}

// =============================================================================================
// Setup kernels (once per run)
// =============================================================================================

// M = W diag(sgn) W^T on fp64 MFMA (v_mfma_f64_16x16x4_f64), lower 128x128 tiles only.  W: Np x kp row-major
// (rows >= N and columns >= k are zero).  Block = 4 waves, each a 64x64 sub-tile (16 MFMA tiles, 64 f64 acc / lane).
// mirror != 0: the off-diagonal tiles are also written transposed, so that M is the full symmetric matrix (the squaring chain of
// shk_spectral_bound feeds M back in as W); sgn == nullptr: all signs +1.
__global__ __launch_bounds__(256) void k_syrk_f64(const double *__restrict__ W, const double *__restrict__ sgn,
                                                  int Np, int kp, double *__restrict__ M, int mirror = 0)
{
    // blockIdx.x enumerates (bi >= bj)
    int t = blockIdx.x, bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = bi * 128 + (wave >> 1) * 64, j0 = bj * 128 + (wave & 1) * 64;
    const int lr = lane & 15, lk = lane >> 4;
    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
    const double *Ai = W + (int64_t)(i0 + lr) * kp + lk;
    const double *Bj = W + (int64_t)(j0 + lr) * kp + lk;
    for (int kk = 0; kk < kp; kk += 4) {
        const double sg = sgn ? sgn[kk + lk] : 1.0;
        double a[4], b[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { a[x] = Ai[(int64_t)x * 16 * kp + kk] * sg; b[x] = Bj[(int64_t)x * 16 * kp + kk]; }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
    // C/D layout of the f64 16x16 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                M[(int64_t)(i0 + x * 16 + lk + 4 * r) * Np + (j0 + y * 16 + lr)] = acc[x][y][r];
                if (mirror && bi != bj) M[(int64_t)(j0 + y * 16 + lr) * Np + (i0 + x * 16 + lk + 4 * r)] = acc[x][y][r];
            }
}

// max |2 M_ij| over i > j (i, j < N)  ->  *amax (bit pattern of a non-negative double orders like uint64)
__global__ __launch_bounds__(256) void k_lower_absmax(const double *__restrict__ M, int N, int Np,
                                                      unsigned long long *__restrict__ amax)
{
    const int i = blockIdx.x;
    double mx = 0;
    for (int j = threadIdx.x; j < i; j += 256) mx = fmax(mx, fabs(2.0 * M[(int64_t)i * Np + j]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(amax, (unsigned long long)__double_as_longlong(mx));
}

__global__ void k_extract_diag(const double *__restrict__ M, int N, int Np, double *__restrict__ mdiag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) mdiag[i] = M[(int64_t)i * Np + i];
}

// fixed-point limbs in LDS-image order.  grid = (tile index within a limb), block = 128 threads (one tile row each).
// Lt limbs are stored (the main pass contracts the top L = Lt - E of them, the extra-limb pass the E lowest).  The 64 columns of a tile
// are stored in the order the kernel's variant fragments are built in (see `expand` in k_lmm_quadform_i8): byte c of 16-byte chunk ch
// holds column 32 (ch & 1) + 8 (c & 3) + 4 (ch >> 1) + (c >> 2).  Ef (optional): the quantisation error of the MAIN pass, e/2 at (i,j)
// and (j,i) in units of the main pass' ulp -- the symmetric matrix whose spectral norm bounds |x^T (G - Gq/s) x| / |x|^2.
__global__ __launch_bounds__(128) void k_lmm_quantize(const double *__restrict__ M, int N, int Np, int NR, int Lt, int E,
                                                      const unsigned long long *__restrict__ amax_bits,
                                                      int8_t *__restrict__ G, float *__restrict__ Ef)
{
    // decode (I, ks) from the linear tile id: tiles of 128-row tile I start at I*(I+1) and there are 2(I+1) of them
    int t = blockIdx.x, I = 0;
    while ((I + 1) * (I + 2) <= t) ++I;
    const int ks = t - I * (I + 1);
    const int r = threadIdx.x;
    const int i = I * 128 + r;
    const double amax = __longlong_as_double((long long)*amax_bits);
    double p256 = 1.0;
    for (int l = 0; l < Lt; ++l) p256 *= 256.0;
    const double scale = amax > 0 ? 0.49 * p256 / amax : 0.0;
    const int64_t TL = (int64_t)NR * (NR + 1);
    for (int ch = 0; ch < 4; ++ch) {
        long long qv[16];
        double fr[16];
        int jj[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int j = ks * 64 + 32 * (ch & 1) + 8 * (c & 3) + 4 * (ch >> 1) + (c >> 2);
            double g = 0.0;
            if (i < N && j < i) g = 2.0 * M[(int64_t)i * Np + j];
            const double gs = g * scale;
            qv[c] = llrint(gs); fr[c] = gs - (double)qv[c]; jj[c] = j;
        }
        for (int l = 0; l < Lt; ++l) {
            if (l == E && Ef) {                                         // what the main pass drops: low digits + rounding, in its own ulp
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    // after E digit extractions qv[c] * 256^E + (digits so far) = the full fixed-point number
                    const int j = jj[c];
                    if (i < N && j < i) {
                        const float e = (float)(fr[c] * 0.5);           // fr now holds (low digits + rounding) / 256^E
                        Ef[(int64_t)i * Np + j] = e; Ef[(int64_t)j * Np + i] = e;
                    }
                }
            }
            uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                long long d = ((qv[c] + 128) & 255) - 128;          // balanced digit in [-128, 127]
                qv[c] = (qv[c] - d) >> 8;
                if (l < E) fr[c] = (fr[c] + (double)d) * (1.0 / 256.0);   // Horner from the lowest digit up: sum_{l<E} d_l 256^l / 256^E
                pk[c >> 2] |= ((uint32_t)(d & 255)) << (8 * (c & 3));
            }
            uint4 o; o.x = pk[0]; o.y = pk[1]; o.z = pk[2]; o.w = pk[3];
            *reinterpret_cast<uint4 *>(G + ((int64_t)l * TL + t) * QF_TILE_BYTES + qf_off(r, ch)) = o;
        }
    }
}

// ---- spectral norm of the symmetric error matrix by power iteration (set-up, once per run)
__global__ __launch_bounds__(256) void k_symv_f32(const float *__restrict__ A, int N, int Np, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double red[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) s = fma((double)A[(int64_t)i * Np + j], x[j], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- a CERTIFIED upper bound on the spectral norm of a symmetric matrix: |A|_2 <= trace(A^(2p))^(1/(2p)) ------------------------------
// (sum of lambda_i^(2p) >= lambda_max^(2p)).  A^(2p) by repeated squaring on the fp64 matrix pipe (k_syrk_f64, A symmetric so A A^T = A^2),
// trace(A^(2p)) = |A^p|_F^2.  For the Wigner-like spectrum of a quantisation-error matrix the bound exceeds the norm by
// (N C_p / 4^p)^(1/(2p)) (C_p the Catalan number): 34 % after 3 squarings (p = 8), 12 % after 4, 4.4 % after 5 (p = 32, N = 5000); a
// spectrum with a few dominant eigenvalues is bounded tighter still.  Unlike the power iteration (which approaches the norm from below and
// stalls between two near-equal or opposite eigenvalues) it cannot under-estimate: the only inexactness is fp64 rounding in the products
// (relative N 2^-53 per squaring), covered by the factor the caller applies.
__global__ __launch_bounds__(256) void k_f32_to_f64_sq(const float *__restrict__ A, int N, int Np, double *__restrict__ X)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Np * Np) return;
    const int r = (int)(i / Np), c = (int)(i % Np);
    X[i] = (r < N && c < N) ? (double)A[i] : 0.0;
}

__global__ __launch_bounds__(256) void k_frob2(const double *__restrict__ X, int64_t n, double *__restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = fma(X[i], X[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// X *= 2^e (exact): keeps the squaring chain inside the fp64 range whatever the scale of A
__global__ __launch_bounds__(256) void k_scale_pow2(double *__restrict__ X, int64_t n, const double *__restrict__ frob2, int *__restrict__ exps, int step)
{
    // e = -floor(log2(|X|_F)) recomputed by every thread from the same value: block-uniform and exact
    int e2; (void)frexp(sqrt(*frob2), &e2);
    const int e = -(e2 - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) exps[step] = e;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) X[i] = ldexp(X[i], e);
}

// nrm[it] = |y| ; x = y / |y|     (one block)
__global__ __launch_bounds__(1024) void k_pow_norm(const double *__restrict__ y, double *__restrict__ x, int N, double *__restrict__ nrm, int it)
{
    __shared__ double red[16];
    __shared__ double tot;
    double s = 0.0;
    for (int j = threadIdx.x; j < N; j += 1024) s = fma(y[j], y[j], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += red[w]; tot = sqrt(t); nrm[it] = tot; }
    __syncthreads();
    const double inv = tot > 0.0 ? 1.0 / tot : 0.0;
    for (int j = threadIdx.x; j < N; j += 1024) x[j] = y[j] * inv;
}

__global__ void k_pow_init(double *__restrict__ x, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = ((double)(h & 0xFFFF) - 32767.5) / 32768.0; }
}

// ---------------------------------------------------------------------------------------------
// host-callable launch wrappers (called from api.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" {

hipError_t shk_repack_bits(hipStream_t st, const uint8_t *bits, int64_t row_bytes, int64_t V, int64_t Vpad, int N,
                           int NB64, uint64_t *T, uint8_t *flip)
{
    const size_t lds = (size_t)64 * row_bytes;
    if (lds > 64 * 1024) {                                                   // more than 8192 samples: per-thread gather (needs Vpad % 256 == 0)
        if (Vpad & 255) return hipErrorInvalidValue;
        if (flip) hipLaunchKernelGGL(k_row_flip, dim3((unsigned)((V + 3) / 4)), dim3(256), 0, st, bits, row_bytes, V, N, flip);
        hipLaunchKernelGGL(k_repack_bits_gather, dim3((unsigned)(Vpad / 256), (unsigned)NB64), dim3(256), 0, st, bits, row_bytes, V, Vpad, N, T, flip);
        return hipGetLastError();
    }
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_repack_bits), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_repack_bits, dim3((unsigned)(Vpad / 64)), dim3(256), lds, st, bits, row_bytes, V, Vpad, N, NB64, T, flip);
    return hipGetLastError();
}

// number of doubles per table entry for (continuous, DP), or 0 when the table kernels do not cover the configuration
static int lin_tab_ne(int DP, int continuous) { return (DP == 0 || DP == 4 || DP == 8) ? 2 + 2 * (continuous ? 1 : 0) + DP : 0; }

hipError_t shk_lmm_build_tab(hipStream_t st, const double *vv, const double *mdiag, const double *yc, const double *Qb, int DP,
                             int continuous, int N, int NB64, double *tab)
{
    const dim3 g((unsigned)NB64), b(256);
    const int cont = continuous ? 1 : 0;
    switch (lin_tab_ne(DP, continuous)) {
    case 2: hipLaunchKernelGGL(k_lmm_build_tab, g, b, 0, st, vv, mdiag, N, NB64, reinterpret_cast<double2 *>(tab)); break;
    case 4: hipLaunchKernelGGL(k_lmm_build_tabn<4>, g, b, 0, st, vv, mdiag, yc, Qb, DP, cont, N, NB64, tab); break;
    case 6: hipLaunchKernelGGL(k_lmm_build_tabn<6>, g, b, 0, st, vv, mdiag, yc, Qb, DP, cont, N, NB64, tab); break;
    case 8: hipLaunchKernelGGL(k_lmm_build_tabn<8>, g, b, 0, st, vv, mdiag, yc, Qb, DP, cont, N, NB64, tab); break;
    case 10: hipLaunchKernelGGL(k_lmm_build_tabn<10>, g, b, 0, st, vv, mdiag, yc, Qb, DP, cont, N, NB64, tab); break;
    case 12: hipLaunchKernelGGL(k_lmm_build_tabn<12>, g, b, 0, st, vv, mdiag, yc, Qb, DP, cont, N, NB64, tab); break;
    default: return hipSuccess;                                // no table for this configuration
    }
    return hipGetLastError();
}

hipError_t shk_lmm_linear(hipStream_t st, int DP, const uint64_t *T, int64_t Vpad, int N, int NB64, const double *vv,
                          const double *mdiag, const double *yc, const double *Qb, const uint64_t *y1,
                          const uint64_t *y0, int continuous, const double *tab, LmmLinOut o)
{
    dim3 grid((unsigned)(Vpad / 256)), blk(256);
    const int cont = continuous ? 1 : 0;
    if (tab) {
        switch (lin_tab_ne(DP, continuous)) {
        case 2: hipLaunchKernelGGL(k_lmm_linear_tab, grid, blk, 0, st, T, Vpad, N, NB64, reinterpret_cast<const double2 *>(tab), y1, y0, o); return hipGetLastError();
        case 4: hipLaunchKernelGGL(k_lmm_linear_tabn<4>, grid, blk, 0, st, T, Vpad, N, NB64, tab, y1, y0, Qb, DP, cont, o); return hipGetLastError();
        case 6: hipLaunchKernelGGL(k_lmm_linear_tabn<6>, grid, blk, 0, st, T, Vpad, N, NB64, tab, y1, y0, Qb, DP, cont, o); return hipGetLastError();
        case 8: hipLaunchKernelGGL(k_lmm_linear_tabn<8>, grid, blk, 0, st, T, Vpad, N, NB64, tab, y1, y0, Qb, DP, cont, o); return hipGetLastError();
        case 10: hipLaunchKernelGGL(k_lmm_linear_tabn<10>, grid, blk, 0, st, T, Vpad, N, NB64, tab, y1, y0, Qb, DP, cont, o); return hipGetLastError();
        case 12: hipLaunchKernelGGL(k_lmm_linear_tabn<12>, grid, blk, 0, st, T, Vpad, N, NB64, tab, y1, y0, Qb, DP, cont, o); return hipGetLastError();
        default: break;
        }
    }
    switch (DP) {
    case 0: hipLaunchKernelGGL(k_lmm_linear<0>, grid, blk, 0, st, T, Vpad, N, NB64, vv, mdiag, yc, Qb, y1, y0, continuous, o); break;
    case 4: hipLaunchKernelGGL(k_lmm_linear<4>, grid, blk, 0, st, T, Vpad, N, NB64, vv, mdiag, yc, Qb, y1, y0, continuous, o); break;
    case 8: hipLaunchKernelGGL(k_lmm_linear<8>, grid, blk, 0, st, T, Vpad, N, NB64, vv, mdiag, yc, Qb, y1, y0, continuous, o); break;
    case 16: hipLaunchKernelGGL(k_lmm_linear<16>, grid, blk, 0, st, T, Vpad, N, NB64, vv, mdiag, yc, Qb, y1, y0, continuous, o); break;
    case 32: hipLaunchKernelGGL(k_lmm_linear<32>, grid, blk, 0, st, T, Vpad, N, NB64, vv, mdiag, yc, Qb, y1, y0, continuous, o); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t shk_lmm_quadform(hipStream_t st, int variant, const int8_t *G, const uint64_t *T, int64_t Vpad, int NR, int L, int lsplit, double *q,
                            const int *nlimit, int rows_last, int64_t *macs_per_variant)
{
    // the wide kernel's conditions: one limb per block, at least four row tiles (its DMA cursor runs four stages past the stream), 32-bit offsets
    const bool wide_ok = lsplit == L && NR >= 4 && (uint64_t)NR * 2 * (uint64_t)Vpad * 8 < (1ull << 32) && (uint64_t)(L + 1) * NR * (NR + 1) * QF_TILE_BYTES < (1ull << 32);
    const bool wide_var = variant == 4 || variant == 6 || variant == 7 || variant == 8 || variant == 9 || variant == 50 || variant == 441 || variant == 162 || variant == 35;
    if (wide_var && !wide_ok) variant = 0;
    // the extra-limb pass (variant 64: a launch sized for the whole batch over a short list, `nlimit`) runs the same kernel under its own name
    // (template argument DM = 3)
    const bool refine_wide = true;
    if (variant == 64 && wide_ok && refine_wide) variant = 464;
    const bool wide = (wide_var && wide_ok) || variant == 464;       // 128 x 128 wave tiles, one wavefront per SIMD
    // rows_last = valid rows (samples) of the last 128-row tile; the wide kernel contracts only the 32-row sub-tiles that hold any (4, 2, 1, or none:
    // NR = 2 NT may exceed ceil(N / 128) by one).  The int8 MACs actually issued per variant go back to the caller (sh_lmm_info, the bench's roofline).
    const int nit_last = !wide ? 4 : rows_last <= 0 ? 0 : rows_last <= 32 ? 1 : rows_last <= 64 ? 2 : 4;
    if (macs_per_variant)
        *macs_per_variant = (int64_t)L * ((int64_t)(NR - 1) * NR * 128 * 64 + (int64_t)2 * NR * 32 * nit_last * 64);
    const dim3 g((unsigned)(Vpad / QF_BN * lsplit)), b(wide ? 256 : 64 * QF_WAVES);
    const size_t lds = QF_NST * QF_STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<23>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<384>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8<407>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<0, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<16, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lmm_quadform_i8w<407>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    switch (variant) {          // 30 + mask = timing ablations (results meaningless); 64 = the extra-limb pass (same code, its own name in profiles)
    case 31: hipLaunchKernelGGL(k_lmm_quadform_i8<1>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 32: hipLaunchKernelGGL(k_lmm_quadform_i8<2>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 34: hipLaunchKernelGGL(k_lmm_quadform_i8<4>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 53: hipLaunchKernelGGL(k_lmm_quadform_i8<23>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 46: hipLaunchKernelGGL(k_lmm_quadform_i8<16>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 38: hipLaunchKernelGGL(k_lmm_quadform_i8<8>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 37: hipLaunchKernelGGL(k_lmm_quadform_i8<7>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 64: hipLaunchKernelGGL(k_lmm_quadform_i8<64>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    case 158: hipLaunchKernelGGL(k_lmm_quadform_i8<128>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;   // no epilogue
    case 286: hipLaunchKernelGGL(k_lmm_quadform_i8<256>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;   // no zeroing
    case 414: hipLaunchKernelGGL(k_lmm_quadform_i8<384>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;   // neither
    case 464: hipLaunchKernelGGL((k_lmm_quadform_i8w<0, 3>), g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;   // extra-limb pass
    case 4: hipLaunchKernelGGL(k_lmm_quadform_i8w<0>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;        // wide wave tile
    case 6: hipLaunchKernelGGL((k_lmm_quadform_i8w<0, 0, 1>), g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break; // wide, s_memtime profile
    case 7: hipLaunchKernelGGL((k_lmm_quadform_i8w<0, 0, 2>), g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break; // wide, s_memtime profile 2
    case 8: hipLaunchKernelGGL((k_lmm_quadform_i8w<16, 0, 1>), g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break; // profile, no barrier
    case 9: hipLaunchKernelGGL((k_lmm_quadform_i8w<1, 0, 1>), g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;  // profile, no DMA
    case 50: hipLaunchKernelGGL(k_lmm_quadform_i8w<16>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;      // wide, no barrier
    case 35: hipLaunchKernelGGL(k_lmm_quadform_i8w<1>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;       // wide, no DMA
    case 162: hipLaunchKernelGGL(k_lmm_quadform_i8w<128>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;    // wide, no epilogue
    case 441: hipLaunchKernelGGL(k_lmm_quadform_i8w<407>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit, nit_last); break;    // wide, MFMAs only
    case 437: hipLaunchKernelGGL(k_lmm_quadform_i8<407>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;   // MFMAs only
    default: hipLaunchKernelGGL(k_lmm_quadform_i8<0>, g, b, lds, st, G, T, Vpad, NR, L, lsplit, q, nlimit); break;
    }
    return hipGetLastError();
}

hipError_t shk_lmm_finalize(hipStream_t st, int64_t V, int64_t Vpad, int nq, LmmLinOut li, const double *q, LmmFinParams P, double *out,
                            uint32_t *flags, LmmRefine R)
{
    hipLaunchKernelGGL(k_lmm_finalize, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, Vpad, nq, li, q, P, out, flags, R);
    return hipGetLastError();
}

// The extra-limb pass over the variants k_lmm_finalize listed: every launch is sized for the whole batch and reads the list length on the
// device (no host round trip); with an empty list the three kernels return at once.
hipError_t shk_lmm_refine(hipStream_t st, int64_t V, int64_t Vpad, int nq, int E, int NR, int NB64p, const int8_t *Glow, const uint64_t *T,
                          uint64_t *T3, double *q3, LmmLinOut li, const double *q, LmmFinParams P, double *out, uint32_t *flags, LmmRefine R)
{
    hipLaunchKernelGGL(k_gather_T_list, dim3(64, (unsigned)NB64p), dim3(256), 0, st, T, Vpad, T3, R);
    hipError_t e = shk_lmm_quadform(st, 64, Glow, T3, Vpad, NR, E, E, q3, R.count, 128, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lmm_refine_fix, dim3(64), dim3(256), 0, st, V, Vpad, nq, E, li, q, q3, P, out, flags, R);
    return hipGetLastError();
}

// M = W sgn W^T, its diagonal, max |G|, the Lt limbs, and (Ef != null) the spectral norm of the main pass' quantisation error:
// nrm[0 .. npow) = |A x_k| of the power iteration on Ef (in units of the main pass' ulp; nrm[npow-1] is the estimate)
hipError_t shk_lmm_build_G(hipStream_t st, const double *W, const double *sgn, int N, int Np, int kp, int NR, int Lt, int E,
                           double *M, double *mdiag, unsigned long long *amax, int8_t *G, float *Ef, double *px, double *py, double *nrm, int npow)
{
    const int nb = Np / 128;
    (void)hipMemsetAsync(amax, 0, sizeof(unsigned long long), st);
    hipLaunchKernelGGL(k_syrk_f64, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, st, W, sgn, Np, kp, M);
    hipLaunchKernelGGL(k_lower_absmax, dim3((unsigned)N), dim3(256), 0, st, M, N, Np, amax);
    hipLaunchKernelGGL(k_extract_diag, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, M, N, Np, mdiag);
    if (Ef) (void)hipMemsetAsync(Ef, 0, sizeof(float) * (size_t)Np * Np, st);
    hipLaunchKernelGGL(k_lmm_quantize, dim3((unsigned)(NR * (NR + 1))), dim3(128), 0, st, M, N, Np, NR, Lt, E, amax, G, Ef);
    if (Ef) {
        hipLaunchKernelGGL(k_pow_init, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, px, N);
        for (int it = 0; it < npow; ++it) {
            hipLaunchKernelGGL(k_symv_f32, dim3((unsigned)N), dim3(256), 0, st, Ef, N, Np, px, py);
            hipLaunchKernelGGL(k_pow_norm, dim3(1), dim3(1024), 0, st, py, px, N, nrm, it);
        }
    }
    return hipGetLastError();
}

// the power iteration alone (the estimate reported beside the certificate; also sh_spectral_bound_f32)
hipError_t shk_power_norm(hipStream_t st, const float *A, int N, int Np, double *px, double *py, double *nrm, int npow)
{
    hipLaunchKernelGGL(k_pow_init, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, px, N);
    for (int it = 0; it < npow; ++it) {
        hipLaunchKernelGGL(k_symv_f32, dim3((unsigned)N), dim3(256), 0, st, A, N, Np, px, py);
        hipLaunchKernelGGL(k_pow_norm, dim3(1), dim3(1024), 0, st, py, px, N, nrm, it);
    }
    return hipGetLastError();
}

// Certified bound on |A|_2 for the symmetric N x N matrix A (fp32, row stride Np): nsq squarings.  X, Y: Np x Np doubles of scratch; work: 2 doubles +
// (nsq + 1) ints on the device.  On return work[1] = |A_scaled^(2^nsq)|_F^2 and the ints hold the power-of-two exponents applied before each
// squaring; the caller assembles  bound = (work[1])^(1 / 2^(nsq+1)) * 2^(-sum_k e_k / 2^k)  (shk_spectral_bound_value).
hipError_t shk_spectral_bound(hipStream_t st, const float *A, int N, int Np, double *X, double *Y, int nsq, double *work, int *exps)
{
    const int nb = Np / 128;
    const int64_t n2 = (int64_t)Np * Np;
    hipLaunchKernelGGL(k_f32_to_f64_sq, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, A, N, Np, X);
    double *cur = X, *nxt = Y;
    for (int k = 0; k <= nsq; ++k) {
        (void)hipMemsetAsync(work, 0, sizeof(double), st);
        hipLaunchKernelGGL(k_frob2, dim3(1024), dim3(256), 0, st, cur, n2, work);
        if (k == nsq) break;
        hipLaunchKernelGGL(k_scale_pow2, dim3(1024), dim3(256), 0, st, cur, n2, work, exps, k);      // |cur|_F in [1, 2)
        hipLaunchKernelGGL(k_syrk_f64, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, st, cur, (const double *)nullptr, Np, Np, nxt, 1);
        std::swap(cur, nxt);
    }
    (void)hipMemcpyAsync(work + 1, work, sizeof(double), hipMemcpyDeviceToDevice, st);
    return hipGetLastError();
}

}  // extern "C"
