// job_kernels.hip -- the device side of the result sink (round 5): which rows of a block does the run print, how many were pre-filtered,
// and their statistics compacted in print order, so that the host touches only PRINTED rows.
//
// Reference: the print loops pyseer/__main__.py:571-593 (fixed effects) and :805-827 (LMM) walk every Seer / LMM tuple of a block:
//     if x.prefilter: prefilter += 1 (printed only with --print-filtered) else: tested += 1; if x.filter and not --print-filtered: skip
//     else: printed += 1; print(format_output(x, ...))
// and pyseer/lmm.py:160-217 leaves NaN in the fields a filtered LMM tuple never received (af / pre-filtered: everything but prep;
// lrt-filtered: beta, bse, frac_h2) and returns the filtered tuples of a block first.  Until round 4 that walk was numpy over every tested row
// (masks, selects, counters: 16 ns of host CPU per row, more than a row's share of the GPU at 33 M rows/s); here it is two small kernels
// over the flags (HBM-bound: 4 B read per row, 12 + 8 ncol bytes written per PRINTED row straight into host-mapped memory).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/seerhip.h"

#define JOB_ROWS_PER_BLOCK 1024          // 256 threads x 4 rows, row = base + i * 256 + t: coalesced flag reads, print order = (i, t) order

// which rows a pass takes (pred) -- the passes of one block run in print order:
//   0 PASS     not pre-filtered and not filtered                      (the default run: one pass)
//   1 ALL      every row                                              (--print-filtered, fixed effects: input order)
//   2 PF       pre-filtered rows     } --print-filtered, LMM: fit_lmm returns a block's filtered tuples first (lmm.py:160-217)
//   3 NOT_PF   the others            }
__device__ __forceinline__ bool job_take(uint32_t f, int pred)
{
    const bool pf = (f & SH_FLAG_PREFILTER) != 0, ft = (f & SH_FLAG_FILTER) != 0;
    return pred == 0 ? (!pf && !ft) : pred == 1 ? true : pred == 2 ? pf : !pf;
}

// per workgroup: how many of its rows the pass takes -> bcount[b]; hdr[1] += pre-filtered rows (pass 0 of a block only: count_pf)
extern "C" __global__ void __launch_bounds__(256) k_job_count(const uint32_t *__restrict__ flags, int64_t V, int pred, int count_pf,
                                                               int *__restrict__ bcount, long long *__restrict__ hdr)
{
    __shared__ int s_n[4], s_pf[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * JOB_ROWS_PER_BLOCK;
    int n = 0, npf = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t v = base + i * 256 + t;
        const uint32_t f = v < V ? flags[v] : (SH_FLAG_PREFILTER | SH_FLAG_FILTER | 0x80000000u);
        const bool live = v < V;
        n += __popcll(__ballot(live && job_take(f, pred)));
        npf += __popcll(__ballot(live && (f & SH_FLAG_PREFILTER) != 0));
    }
    if (lane == 0) { s_n[w] = n; s_pf[w] = npf; }
    __syncthreads();
    if (t == 0) {
        bcount[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        const int p = s_pf[0] + s_pf[1] + s_pf[2] + s_pf[3];
        if (count_pf && p) atomicAdd((unsigned long long *)&hdr[1], (unsigned long long)p);
    }
}

// the rows the pass takes, in order, behind the `*base_from` rows of the passes before it: row index, flags and the nrow statistics of each
// (column-major with stride `cap`), NaN-masked like fit_lmm's tuples when lmm != 0.  out[a * V + v] = statistic a of row v (the SoA the
// batch kernels leave).  The last workgroup leaves the count so far in *total_out (never the word another
// workgroup of the same launch reads its base from).
extern "C" __global__ void __launch_bounds__(256) k_job_scatter(const uint32_t *__restrict__ flags, const double *__restrict__ out, int64_t V, int nrow,
                                                                 int pred, int lmm, const int *__restrict__ bcount, const long long *__restrict__ base_from,
                                                                 long long *__restrict__ total_out, int32_t *__restrict__ r_idx, uint32_t *__restrict__ r_flags,
                                                                 double *__restrict__ r_cols, int64_t cap, int32_t *__restrict__ d_sel)
{
    __shared__ long long s_base; __shared__ int s_part[256]; __shared__ int s_w[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // rows taken by the workgroups before this one (at most a few thousand counts: every workgroup adds them up itself)
    int acc = 0;
    for (int b = t; b < (int)blockIdx.x; b += 256) acc += bcount[b];
    s_part[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) s_part[t] += s_part[t + s]; __syncthreads(); }
    if (t == 0) s_base = (base_from ? *base_from : 0) + s_part[0];
    __syncthreads();
    long long pos = s_base;
    const int64_t base = (int64_t)blockIdx.x * JOB_ROWS_PER_BLOCK;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (int i = 0; i < 4; ++i) {
        const int64_t v = base + i * 256 + t;
        const uint32_t f = v < V ? flags[v] : 0u;
        const bool take = v < V && job_take(f, pred);
        const unsigned long long m = __ballot(take);
        if (lane == 0) s_w[w] = __popcll(m);
        __syncthreads();
        long long p = pos;
        for (int k = 0; k < w; ++k) p += s_w[k];
        if (take) {
            p += __popcll(m & ((1ull << lane) - 1ull));
            r_idx[p] = (int32_t)v; r_flags[p] = f;
            if (d_sel) d_sel[p] = (int32_t)v;                           // (a device copy of the list: the lineage fits of the printed rows)
            const bool pre = (f & SH_FLAG_PREFILTER) != 0, af = (f & SH_NOTE_AF_FILTER) != 0, lrt = (f & SH_NOTE_LRT_FILTER) != 0;
            for (int a = 0; a < nrow; ++a) {
                double x = out[(size_t)a * (size_t)V + (size_t)v];
                if (lmm) {                                              // pyseer/lmm.py:160-217 (pyseer_amd/lmm.py mask_like_fit_lmm)
                    if (a == 0) { if (af) x = qnan; }
                    else if (pre || (a >= 2 && lrt)) x = qnan;
                }
                r_cols[(size_t)a * (size_t)cap + (size_t)p] = x;
            }
        }
        pos += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && t == 0) *total_out = pos;              // (= *base_from + this pass's total: the count so far)
}

extern "C" hipError_t shk_job_select(hipStream_t st, const uint32_t *flags, const double *out, int64_t V, int nrow, int lmm, int print_filtered,
                                     int *bcount, long long *hdr_dev, long long *hdr, int32_t *r_idx, uint32_t *r_flags, double *r_cols, int64_t cap, int32_t *d_sel)
{
    if (V <= 0) return hipSuccess;
    const unsigned nb = (unsigned)((V + JOB_ROWS_PER_BLOCK - 1) / JOB_ROWS_PER_BLOCK);
    hipError_t e = hipMemsetAsync(hdr_dev, 0, 4 * sizeof(long long), st); if (e != hipSuccess) return e;
    const int npass = (print_filtered && lmm) ? 2 : 1;
    for (int ps = 0; ps < npass; ++ps) {
        const int pred = !print_filtered ? 0 : (!lmm ? 1 : (ps == 0 ? 2 : 3));
        hipLaunchKernelGGL(k_job_count, dim3(nb), dim3(256), 0, st, flags, V, pred, ps == 0 ? 1 : 0, bcount, hdr_dev);
        hipLaunchKernelGGL(k_job_scatter, dim3(nb), dim3(256), 0, st, flags, out, V, nrow, pred, lmm, bcount, ps == 0 ? (const long long *)nullptr : hdr_dev + 2,
                           ps == npass - 1 ? hdr_dev : hdr_dev + 2, r_idx, r_flags, r_cols, cap, d_sel);
    }
    e = hipGetLastError(); if (e != hipSuccess) return e;
    // the two counts go to the host-mapped header last (a plain copy on the stream: the host reads them after the block's event)
    return hipMemcpyAsync(hdr, hdr_dev, 4 * sizeof(long long), hipMemcpyDeviceToHost, st);
}

// ---- --output-patterns inside the stream (round 6) ----------------------------------------------------------------------------------------
// The reference writes, for every TESTED variant, hash_pattern(k) = base64(md5(k.view(uint8))) + "\n" with k the variant's presence vector
// as int64 (pyseer/input.py:710-723; print loops pyseer/__main__.py:584-585, 794-795, 818-819): an md5 over 8 N bytes per variant -- 60 us
// of a CPU core per variant at N = 5000, four orders of magnitude from the rate of the fits.  md5 is sequential within a message and
// independent between messages: one lane per variant.  A 64-byte block of the message is eight samples, each a little-endian int64 0 / 1:
// word 2 i of the block = the bit of sample 8 b + i, the odd words are zero (the unrolled rounds fold them away); the last block carries the
// 0x80 marker and the length (64 N bits), in a block of its own when N % 8 == 7.  T: the repacked rows ([64-sample word][Vpad]).
__device__ __forceinline__ uint32_t md5_rotl(uint32_t x, int s) { return __builtin_amdgcn_alignbit(x, x, 32 - s); }
__device__ __forceinline__ void md5_block(uint32_t (&st)[4], const uint32_t (&M)[16])
{
    constexpr uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be,
        0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
        0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    constexpr int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                           4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        uint32_t f; int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        const uint32_t t = a + f + K[i] + M[g];
        a = d; d = c; c = b; b = b + md5_rotl(t, S[i]);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

// digest[v] (4 words) of every row that is not pre-filtered (the others are never written out)
extern "C" __global__ void __launch_bounds__(256) k_job_md5(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N, const uint32_t *__restrict__ flags,
                                                             uint32_t *__restrict__ digest)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V || (flags[v] & SH_FLAG_PREFILTER)) return;
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint32_t M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) M[i] = 0u;
    const int nfull = N >> 3;                                            // whole blocks: eight samples each
    uint64_t w = 0;
    for (int b = 0; b < nfull; ++b) {
        if ((b & 7) == 0) w = T[(int64_t)(b >> 3) * Vpad + v];
        const uint32_t eight = (uint32_t)(w >> (8 * (b & 7))) & 0xFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) M[2 * i] = (eight >> i) & 1u;
        md5_block(st, M);
    }
    // the tail: r = N % 8 samples, the marker byte 0x80 behind them, the length in bits in the last two words (of the next block when r == 7)
    const int r = N & 7;
    uint32_t eight = 0;
    if (r) eight = (uint32_t)(T[(int64_t)(nfull >> 3) * Vpad + v] >> (8 * (nfull & 7))) & ((1u << r) - 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) M[2 * i] = i < r ? ((eight >> i) & 1u) : (i == r ? 0x80u : 0u);
    const uint64_t nbits = (uint64_t)N * 64ull;
    if (r < 7) { M[14] = (uint32_t)nbits; M[15] = (uint32_t)(nbits >> 32); md5_block(st, M); }
    else {
        md5_block(st, M);
#pragma unroll
        for (int i = 0; i < 16; ++i) M[i] = 0u;
        M[14] = (uint32_t)nbits; M[15] = (uint32_t)(nbits >> 32);
        md5_block(st, M);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) digest[(size_t)v * 4 + i] = st[i];
}

// base64 of the digests of the rows that are not pre-filtered, 24 characters + "\n" each, in row order (binascii.b2a_base64): text[25 p ..]
// for the p-th such row.  bcount: k_job_count's counts for pred = 3.
extern "C" __global__ void __launch_bounds__(256) k_job_pat_text(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ digest, int64_t V,
                                                                  const int *__restrict__ bcount, char *__restrict__ text)
{
    __shared__ long long s_base; __shared__ int s_part[256]; __shared__ int s_w[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int acc = 0;
    for (int b = t; b < (int)blockIdx.x; b += 256) acc += bcount[b];
    s_part[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) s_part[t] += s_part[t + s]; __syncthreads(); }
    if (t == 0) s_base = s_part[0];
    __syncthreads();
    long long pos = s_base;
    const int64_t base = (int64_t)blockIdx.x * JOB_ROWS_PER_BLOCK;
    for (int i = 0; i < 4; ++i) {
        const int64_t v = base + i * 256 + t;
        const bool take = v < V && job_take(flags[v < V ? v : 0], 3);
        const unsigned long long m = __ballot(take);
        if (lane == 0) s_w[w] = __popcll(m);
        __syncthreads();
        long long p = pos;
        for (int k = 0; k < w; ++k) p += s_w[k];
        if (take) {
            p += __popcll(m & ((1ull << lane) - 1ull));
            uint8_t d[18];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t x = digest[(size_t)v * 4 + k]; d[4 * k] = (uint8_t)x; d[4 * k + 1] = (uint8_t)(x >> 8); d[4 * k + 2] = (uint8_t)(x >> 16); d[4 * k + 3] = (uint8_t)(x >> 24); }
            d[16] = 0; d[17] = 0;
            char *o = text + p * 25;
            auto b64 = [](uint32_t x) -> char { return (char)(x < 26 ? 'A' + x : x < 52 ? 'a' + (x - 26) : x < 62 ? '0' + (x - 52) : x == 62 ? '+' : '/'); };
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t x = ((uint32_t)d[3 * k] << 16) | ((uint32_t)d[3 * k + 1] << 8) | (uint32_t)d[3 * k + 2];
                o[4 * k] = b64(x >> 18); o[4 * k + 1] = b64((x >> 12) & 63u); o[4 * k + 2] = b64((x >> 6) & 63u); o[4 * k + 3] = b64(x & 63u);
            }
            o[22] = '='; o[23] = '='; o[24] = '\n';                       // (16 bytes: five whole groups and one byte)
        }
        pos += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// digests + text of a block's tested rows (the rows that are not pre-filtered, in row order); `text` (device, 25 V bytes) -> `h_text` (pinned)
extern "C" hipError_t shk_job_patterns(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, int N, const uint32_t *flags, int *bcount,
                                       long long *hdr_scratch, uint32_t *digest, char *text, char *h_text)
{
    if (V <= 0) return hipSuccess;
    const unsigned nb = (unsigned)((V + JOB_ROWS_PER_BLOCK - 1) / JOB_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(k_job_md5, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, T, Vpad, V, N, flags, digest);
    hipLaunchKernelGGL(k_job_count, dim3(nb), dim3(256), 0, st, flags, V, 3, 0, bcount, hdr_scratch);
    hipLaunchKernelGGL(k_job_pat_text, dim3(nb), dim3(256), 0, st, flags, digest, V, bcount, text);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    return hipMemcpyAsync(h_text, text, (size_t)V * 25, hipMemcpyDeviceToHost, st);
}

