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

// compact the representatives' packed rows: one wave-quarter per row would be nicer; rows are only ~640 B, keep it simple
__global__ __launch_bounds__(256) void k_dd_gather_rows(const uint8_t *__restrict__ bits, int64_t row_bytes, int64_t V,
                                                        const int *__restrict__ rep, const int *__restrict__ slot_of,
                                                        uint8_t *__restrict__ bits_u)
{
    const int64_t v = blockIdx.x;
    if (v >= V || rep[v] != (int)v) return;
    const uint8_t *src = bits + v * row_bytes;
    uint8_t *dst = bits_u + (int64_t)slot_of[v] * row_bytes;
    for (int64_t b = threadIdx.x; b < row_bytes; b += 256) dst[b] = src[b];
}

// fan the unique results back out: out is (nrow, V) SoA, out_u is (nrow, Vu)
__global__ __launch_bounds__(256) void k_dd_scatter(int64_t V, int64_t Vu, int nrow, const int *__restrict__ rep,
                                                    const int *__restrict__ slot_of, const double *__restrict__ out_u,
                                                    const uint32_t *__restrict__ flags_u, double *__restrict__ out,
                                                    uint32_t *__restrict__ flags)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int64_t s = slot_of[rep[v]];
    for (int a = 0; a < nrow; ++a) out[(int64_t)a * V + v] = out_u[(int64_t)a * Vu + s];
    flags[v] = flags_u[s];
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

hipError_t shk_dd_gather(hipStream_t st, const uint8_t *bits, int64_t row_bytes, int64_t V, const int *rep, const int *slot_of,
                         uint8_t *bits_u)
{
    hipLaunchKernelGGL(k_dd_gather_rows, dim3((unsigned)V), dim3(256), 0, st, bits, row_bytes, V, rep, slot_of, bits_u);
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
