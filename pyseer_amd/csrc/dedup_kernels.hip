// dedup_kernels.hip -- pattern de-duplication in front of the per-variant tests.
//
// Many k-mers share one presence/absence pattern (pyseer's --output-patterns exists for exactly that reason,
// docs/usage.rst:578; scripts/count_patterns.py).  Identical patterns have identical statistics, so each distinct pattern is
// tested once and its result fanned back out.  Exact: equality is decided on the packed rows themselves, the 64-bit hash
// only finds candidates (a colliding, different row simply becomes its own representative).
#include "common.h"

#define DD_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t dd_mix(uint64_t h, uint64_t w)
{
    h ^= w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 31;
    return h;
}

// one lane per variant, T[sb][v] layout (coalesced)
__global__ __launch_bounds__(256) void k_dd_hash(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int NB64,
                                                 uint64_t *__restrict__ h)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    uint64_t x = 0x243F6A8885A308D3ull;
    for (int sb = 0; sb < NB64; ++sb) x = dd_mix(x, T[(int64_t)sb * Vpad + v]);
    if (x == DD_EMPTY) x = 0;
    h[v] = x;
}

// insert: the table remembers, per hash, the SMALLEST variant index carrying it (deterministic representative)
__global__ __launch_bounds__(256) void k_dd_insert(const uint64_t *__restrict__ h, int64_t V, uint64_t cap_mask,
                                                   unsigned long long *__restrict__ keys, int *__restrict__ idx)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const uint64_t hv = h[v];
    uint64_t s = hv & cap_mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[s], (unsigned long long)DD_EMPTY, (unsigned long long)hv);
        if (prev == DD_EMPTY || prev == hv) { atomicMin(&idx[s], (int)v); return; }
        s = (s + 1) & cap_mask;
    }
}

// rep[v] = representative of v's pattern (v itself when its row differs from the hash's first owner)
__global__ __launch_bounds__(256) void k_dd_resolve(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int NB64,
                                                    const uint64_t *__restrict__ h, uint64_t cap_mask,
                                                    const unsigned long long *__restrict__ keys, const int *__restrict__ idx,
                                                    int *__restrict__ rep, int *__restrict__ slot_of, int *__restrict__ nuniq)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const uint64_t hv = h[v];
    uint64_t s = hv & cap_mask;
    while (keys[s] != hv) s = (s + 1) & cap_mask;
    int r = idx[s];
    if (r != (int)v) {
        bool same = true;
        for (int sb = 0; sb < NB64 && same; ++sb) same = T[(int64_t)sb * Vpad + v] == T[(int64_t)sb * Vpad + r];
        if (!same) r = (int)v;
    }
    rep[v] = r;
    if (r == (int)v) slot_of[v] = atomicAdd(nuniq, 1);
}

// compact the representatives' packed rows (one wavefront per row, 64-bit words when the row width allows)
__global__ __launch_bounds__(256) void k_dd_gather_rows(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V,
                                                        const int *__restrict__ rep, const int *__restrict__ slot_of,
                                                        uint8_t *__restrict__ bits_u)
{
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);         // one wavefront per row
    if (v >= V || rep[v] != (int)v) return;
    const int lane = threadIdx.x & 63;
    const uint8_t *src = bits + v * row_bytes;
    uint8_t *dst = bits_u + (int64_t)slot_of[v] * row_bytes;
    if ((row_bytes & 7) == 0 && (((uintptr_t)bits | (uintptr_t)bits_u) & 7) == 0) {
        const uint64_t *s64 = reinterpret_cast<const uint64_t *>(src); uint64_t *d64 = reinterpret_cast<uint64_t *>(dst);
        for (int64_t w = lane; w < (row_bytes >> 3); w += 64) d64[w] = s64[w];
    } else {
        for (int64_t b = lane; b < row_bytes; b += 64) dst[b] = src[b];
    }
}

// fan the unique results back out: out is (nrow, V) SoA, out_u is (nrow, Vu)
__global__ __launch_bounds__(256) void k_dd_scatter(int64_t V, int64_t Vu, int nrow, const int *__restrict__ rep,
                                                    const int *__restrict__ slot_of, const double *__restrict__ out_u,
                                                    const uint32_t *__restrict__ flags_u, double *__restrict__ out,
                                                    uint32_t *__restrict__ flags)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V || rep[v] < 0) return;                              // rep < 0: the row's output was written directly (k_pf_slots)
    const int64_t s = slot_of[rep[v]];
    for (int a = 0; a < nrow; ++a) out[(int64_t)a * V + v] = out_u[(int64_t)a * Vu + s];
    flags[v] = flags_u[s];
}

// ---- AF compaction for the fixed-effects path: rows outside [min_af, max_af] all produce the same output (NaN statistics, af-filter
// note), so one of them stands for all (rep = the lowest such row) and only the kept rows + that one go through the kernels.
// Reuses the gather / scatter of the pattern de-duplication (rep[v] == v rows are gathered to slot_of[v]).
__global__ __launch_bounds__(256) void k_af_rows(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V, int N,
                                                 double min_af, double max_af, int *__restrict__ rep, int *__restrict__ slot_of,
                                                 int *__restrict__ cnt /* [0] kept rows, [1] lowest filtered row */)
{
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const int lane = threadIdx.x & 63, nb = (N + 7) >> 3;
    const uint8_t *row = bits + v * row_bytes;
    int m = 0;
    if ((row_bytes & 7) == 0 && ((uintptr_t)bits & 7) == 0) {                   // the engine's own row width: 64-bit words
        const uint64_t *r64 = reinterpret_cast<const uint64_t *>(row);
        const int nw = (N + 63) >> 6;
        for (int w = lane; w < nw; w += 64) {
            uint64_t x = r64[w];
            if (w == nw - 1 && (N & 63)) x &= (1ull << (N & 63)) - 1ull;
            m += __popcll(x);
        }
    } else {
        for (int b = lane; b < nb; b += 64) {
            unsigned x = row[b];
            if (b == nb - 1 && (N & 7)) x &= (1u << (N & 7)) - 1u;
            m += __popc(x);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m += __shfl_xor(m, o);
    if (lane == 0) rep[v] = m;                                     // carrier count; k_af_slots turns it into rep / slot_of
}

// one thread per row; one atomic per wavefront (one per row serialises on the two counters: 4.5 ms for 393 216 rows)
__global__ __launch_bounds__(256) void k_af_slots(int64_t V, int N, double min_af, double max_af, int *__restrict__ rep,
                                                  int *__restrict__ slot_of, int *__restrict__ cnt)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = v < V;
    bool keep = false;
    if (valid) { const double af = (double)rep[v] / (double)N; keep = (min_af <= af && af <= max_af); }
    const unsigned long long km = __ballot(keep), fm = __ballot(valid && !keep);
    int base = 0;
    if (lane == 0 && km) base = atomicAdd(&cnt[0], __popcll(km));
    base = __shfl(base, 0);
    if (lane == 0 && fm) atomicMin(&cnt[1], (int)(v + __ffsll((long long)fm) - 1));
    if (!valid) return;
    if (keep) { rep[v] = (int)v; slot_of[v] = base + __popcll(km & ((1ull << lane) - 1ull)); }
    else rep[v] = -1;
}

// ---- the same with the binary prefilter (model.py:31-70, 266): a row whose 2x2 chi-square p-value exceeds --filter-pvalue is never
// fitted; its output (prep, notes, NaN statistics) is written here and only the rows that will be fitted go through the kernels.  With
// --filter-pvalue 1e-3 that is ~99 % of a batch; inside the regression kernel those rows are idle lanes of wavefronts that still run.
__global__ __launch_bounds__(256) void k_pf_rows(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V, int N,
                                                 const uint64_t *__restrict__ y1, const uint64_t *__restrict__ y0,
                                                 int *__restrict__ m_out, int *__restrict__ t11_out, int *__restrict__ t01_out)
{
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const int lane = threadIdx.x & 63;
    const uint8_t *row = bits + v * row_bytes;
    int m = 0, t11 = 0, t01 = 0;
    const int nw = (N + 63) >> 6;
    for (int w = lane; w < nw; w += 64) {
        uint64_t x = 0;
        if ((row_bytes & 7) == 0 && ((uintptr_t)bits & 7) == 0) x = reinterpret_cast<const uint64_t *>(row)[w];
        else for (int b = 0; b < 8 && w * 8 + b < row_bytes; ++b) x |= (uint64_t)row[w * 8 + b] << (8 * b);
        if (w == nw - 1 && (N & 63)) x &= (1ull << (N & 63)) - 1ull;
        m += __popcll(x); t11 += __popcll(x & y1[w]); t01 += __popcll(x & y0[w]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m += __shfl_xor(m, o); t11 += __shfl_xor(t11, o); t01 += __shfl_xor(t01, o); }
    if (lane == 0) { m_out[v] = m; t11_out[v] = t11; t01_out[v] = t01; }
}

__global__ __launch_bounds__(256) void k_pf_slots(int64_t V, int N, double min_af, double max_af, int af_on, int n1, int n0, double pret,
                                                  int nrow, const int *__restrict__ m_in, const int *__restrict__ t11_in,
                                                  const int *__restrict__ t01_in, int *__restrict__ rep, int *__restrict__ slot_of,
                                                  int *__restrict__ cnt, double *__restrict__ out, uint32_t *__restrict__ flags)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = v < V;
    int cls = 3;                                                   // 0 keep, 1 AF-filtered, 2 prefilter failed, 3 out of range
    double prep = NAN; bool bad = false;
    if (valid) {
        const int m = m_in[v];
        const double af = (double)m / (double)N;
        if (af_on && !(min_af <= af && af <= max_af)) cls = 1;
        else {
            const int t11 = t11_in[v], t01 = t01_in[v];
            prep = sh_prefilter_binary(t11, n1 - t11, t01, n0 - t01, &bad);
            cls = (prep > pret || !isfinite(prep)) ? 2 : 0;                            // model.py:266 (>)
        }
    }
    const unsigned long long km = __ballot(cls == 0);
    int base = 0;
    if (lane == 0 && km) base = atomicAdd(&cnt[0], __popcll(km));
    base = __shfl(base, 0);
    if (!valid) return;
    if (cls == 0) { rep[v] = (int)v; slot_of[v] = base + __popcll(km & ((1ull << lane) - 1ull)); }
    else {                                                         // never fitted: the output is known here (k_glm_fast writes the same)
        rep[v] = -2;
        out[v] = prep;                                             // NaN for an AF-filtered row
        for (int a = 1; a < nrow; ++a) out[(int64_t)a * V + v] = NAN;
        flags[v] = (cls == 1) ? (SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER)
                              : ((bad ? SH_NOTE_BAD_CHISQ : 0u) | SH_NOTE_PRE_FILTER | SH_FLAG_PREFILTER);
    }
}

__global__ __launch_bounds__(256) void k_af_rep(int64_t V, int *__restrict__ rep, int *__restrict__ slot_of, const int *__restrict__ cnt)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int R = cnt[1];
    if (rep[v] == -1) { rep[v] = R; if ((int)v == R) slot_of[v] = cnt[0]; }
}

extern "C" {

hipError_t shk_dd_find(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, int NB64, uint64_t *h, uint64_t cap,
                       unsigned long long *keys, int *idx, int *rep, int *slot_of, int *nuniq)
{
    const dim3 g((unsigned)((V + 255) / 256)), b(256);
    hipMemsetAsync(keys, 0xFF, cap * sizeof(unsigned long long), st);
    hipMemsetAsync(idx, 0x7F, cap * sizeof(int), st);
    hipMemsetAsync(nuniq, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_dd_hash, g, b, 0, st, T, Vpad, V, NB64, h);
    hipLaunchKernelGGL(k_dd_insert, g, b, 0, st, h, V, cap - 1, keys, idx);
    hipLaunchKernelGGL(k_dd_resolve, g, b, 0, st, T, Vpad, V, NB64, h, cap - 1, keys, idx, rep, slot_of, nuniq);
    return hipGetLastError();
}

hipError_t shk_af_rows(hipStream_t st, int which, const uint8_t *bits, int64_t row_bytes, int64_t V, int N, double min_af, double max_af,
                       int *rep, int *slot_of, int *cnt)
{
    if (which == 0) {
        hipLaunchKernelGGL(k_af_rows, dim3((unsigned)((V + 3) / 4)), dim3(256), 0, st, bits, row_bytes, V, N, min_af, max_af, rep, slot_of, cnt);
        hipLaunchKernelGGL(k_af_slots, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, min_af, max_af, rep, slot_of, cnt);
    } else hipLaunchKernelGGL(k_af_rep, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, rep, slot_of, cnt);
    return hipGetLastError();
}

hipError_t shk_pf_rows(hipStream_t st, const uint8_t *bits, int64_t row_bytes, int64_t V, int N, const uint64_t *y1, const uint64_t *y0,
                       double min_af, double max_af, int af_on, int n1, int n0, double pret, int nrow, int *m, int *t11, int *t01,
                       int *rep, int *slot_of, int *cnt, double *out, uint32_t *flags)
{
    hipLaunchKernelGGL(k_pf_rows, dim3((unsigned)((V + 3) / 4)), dim3(256), 0, st, bits, row_bytes, V, N, y1, y0, m, t11, t01);
    hipLaunchKernelGGL(k_pf_slots, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, N, min_af, max_af, af_on, n1, n0, pret, nrow, m, t11, t01,
                       rep, slot_of, cnt, out, flags);
    return hipGetLastError();
}

hipError_t shk_dd_gather(hipStream_t st, const uint8_t *bits, int64_t row_bytes, int64_t V, const int *rep, const int *slot_of,
                         uint8_t *bits_u)
{
    hipLaunchKernelGGL(k_dd_gather_rows, dim3((unsigned)((V + 3) / 4)), dim3(256), 0, st, bits, row_bytes, V, rep, slot_of, bits_u);
    return hipGetLastError();
}

hipError_t shk_dd_scatter(hipStream_t st, int64_t V, int64_t Vu, int nrow, const int *rep, const int *slot_of,
                          const double *out_u, const uint32_t *flags_u, double *out, uint32_t *flags)
{
    hipLaunchKernelGGL(k_dd_scatter, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, Vu, nrow, rep, slot_of, out_u,
                       flags_u, out, flags);
    return hipGetLastError();
}

}  // extern "C"
