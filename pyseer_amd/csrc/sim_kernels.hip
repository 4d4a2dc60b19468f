// sim_kernels.hip -- sample co-occurrence (kinship) matrix K = G G^T from packed presence rows.
//
// Reference: pyseer/similarity.py:99-113 concatenates the N x block matrices of load_var_block (AF/missing-filtered variants
// stay all-zero columns, pyseer/input.py:690-697) and calls np.matmul(G, G.T).  G is 0/1, so K[i][j] = number of kept variants
// carried by both samples: integer work, done here as AND + popcount on bit-planes (exact; no MFMA reshaping).
//
// Layout: T[sb][Vpad] (k_repack_bits: one 64-sample word per variant) -> S[vw][NS] (one 64-VARIANT word per sample, variant-word
// major so that both the transpose's writes and the pair kernel's tile loads are contiguous).  NS = N rounded up to 128.
#include <hip/hip_runtime.h>
#include <cstdint>

#define SIM_TILE 128            // samples per tile side
#define SIM_KW 16               // variant words (of 64 variants) per LDS chunk

// keep[vw]: bit l set <=> variant 64*vw + l exists (v < V) and passes the allele-frequency filter (input.py:690: af < min or af > max)
__global__ __launch_bounds__(256) void k_sim_keep(const uint64_t *__restrict__ T, int64_t Vpad, int64_t V, int N, int NB64,
                                                  double min_af, double max_af, int af_on, uint64_t *__restrict__ keep)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    for (int sb = 0; sb < NB64; ++sb) cnt += __popcll(T[(int64_t)sb * Vpad + v]);
    bool k = v < V;
    if (k && af_on) { const double af = (double)cnt / (double)N; k = (min_af <= af) && (af <= max_af); }
    const uint64_t m = __ballot(k);
    if ((threadIdx.x & 63) == 0) keep[v >> 6] = m;
}

// 64 x 64 bit transpose per wavefront: lane l holds the 64-sample word of variant 64*vw + l; sample s collects bit s of every lane.
__global__ __launch_bounds__(256) void k_sim_transpose(const uint64_t *__restrict__ T, int64_t Vpad, int NB64, int64_t VW, int64_t VWp,
                                                       int NS, const uint64_t *__restrict__ keep, uint64_t *__restrict__ S)
{
    const int lane = threadIdx.x & 63;
    const int64_t vw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int sb = blockIdx.y;
    if (vw >= VWp) return;
    uint64_t out = 0;
    if (vw < VW && sb < NB64) {
        const uint64_t w = T[(int64_t)sb * Vpad + vw * 64 + lane];
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
#pragma unroll
        for (int s = 0; s < 32; ++s) { const uint64_t b = __ballot((lo >> s) & 1u); if (lane == s) out = b; }
#pragma unroll
        for (int s = 0; s < 32; ++s) { const uint64_t b = __ballot((hi >> s) & 1u); if (lane == 32 + s) out = b; }
        out &= keep[vw];
    }
    S[vw * NS + sb * 64 + lane] = out;
}

// K[I-tile][J-tile] += sum_w popc(S[w][i] & S[w][j]) for J <= I; blockIdx.y splits the variant words (integer atomics: order-free).
// 256 threads, 8 x 8 sample pairs per thread: thread (tx, ty) owns samples {32m + 2t + {0,1}, m = 0..3} on each side, so that the
// 16 lanes of a ds_read_b128 group read 256 contiguous bytes.
__global__ __launch_bounds__(256) void k_sim_pairs(const uint64_t *__restrict__ S, int NS, int64_t VWp, int chunks_per_split,
                                                   unsigned long long *__restrict__ Kacc)
{
    __shared__ __attribute__((aligned(16))) uint64_t sa[SIM_KW][SIM_TILE];
    __shared__ __attribute__((aligned(16))) uint64_t sj[SIM_KW][SIM_TILE];
    // blockIdx.x -> (I, J), J <= I
    int I = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
    while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
    const int J = (int)blockIdx.x - I * (I + 1) / 2;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t nchunk = VWp / SIM_KW;
    const int64_t c0 = (int64_t)blockIdx.y * chunks_per_split;
    int64_t c1 = c0 + chunks_per_split; if (c1 > nchunk) c1 = nchunk;
    uint32_t acc[8][8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[p][q] = 0;
    // each thread moves 4 + 4 16-byte pieces per chunk (register-staged: the next chunk is in flight during the popcounts)
    uint4 ra0, ra1, ra2, ra3, rj0, rj1, rj2, rj3;
    const int fw = tid >> 6, fcol = (tid & 63) * 2;                // piece r: word row fw + 4 r, u64 column fcol
#define SIM_FETCH(c)                                                                                     \
    {                                                                                                    \
        const uint64_t *row = S + ((c) * SIM_KW + fw) * NS;                                              \
        ra0 = *reinterpret_cast<const uint4 *>(row + I * SIM_TILE + fcol);                               \
        rj0 = *reinterpret_cast<const uint4 *>(row + J * SIM_TILE + fcol);                               \
        ra1 = *reinterpret_cast<const uint4 *>(row + 4 * (int64_t)NS + I * SIM_TILE + fcol);             \
        rj1 = *reinterpret_cast<const uint4 *>(row + 4 * (int64_t)NS + J * SIM_TILE + fcol);             \
        ra2 = *reinterpret_cast<const uint4 *>(row + 8 * (int64_t)NS + I * SIM_TILE + fcol);             \
        rj2 = *reinterpret_cast<const uint4 *>(row + 8 * (int64_t)NS + J * SIM_TILE + fcol);             \
        ra3 = *reinterpret_cast<const uint4 *>(row + 12 * (int64_t)NS + I * SIM_TILE + fcol);            \
        rj3 = *reinterpret_cast<const uint4 *>(row + 12 * (int64_t)NS + J * SIM_TILE + fcol);            \
    }
    if (c0 < c1) SIM_FETCH(c0)
    for (int64_t c = c0; c < c1; ++c) {
        __syncthreads();
        *reinterpret_cast<uint4 *>(&sa[fw][fcol]) = ra0;      *reinterpret_cast<uint4 *>(&sj[fw][fcol]) = rj0;
        *reinterpret_cast<uint4 *>(&sa[fw + 4][fcol]) = ra1;  *reinterpret_cast<uint4 *>(&sj[fw + 4][fcol]) = rj1;
        *reinterpret_cast<uint4 *>(&sa[fw + 8][fcol]) = ra2;  *reinterpret_cast<uint4 *>(&sj[fw + 8][fcol]) = rj2;
        *reinterpret_cast<uint4 *>(&sa[fw + 12][fcol]) = ra3; *reinterpret_cast<uint4 *>(&sj[fw + 12][fcol]) = rj3;
        __syncthreads();
        if (c + 1 < c1) SIM_FETCH(c + 1)
#pragma unroll 2
        for (int w = 0; w < SIM_KW; ++w) {
            uint64_t a[8], b[8];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint4 x = *reinterpret_cast<const uint4 *>(&sa[w][32 * m + 2 * ty]);
                const uint4 y = *reinterpret_cast<const uint4 *>(&sj[w][32 * m + 2 * tx]);
                a[2 * m] = (uint64_t)x.x | ((uint64_t)x.y << 32); a[2 * m + 1] = (uint64_t)x.z | ((uint64_t)x.w << 32);
                b[2 * m] = (uint64_t)y.x | ((uint64_t)y.y << 32); b[2 * m + 1] = (uint64_t)y.z | ((uint64_t)y.w << 32);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[p][q] += (uint32_t)__popcll(a[p] & b[q]);
        }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int i = I * SIM_TILE + 32 * (p >> 1) + 2 * ty + (p & 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = J * SIM_TILE + 32 * (q >> 1) + 2 * tx + (q & 1);
            if (acc[p][q]) atomicAdd(&Kacc[(int64_t)i * NS + j], (unsigned long long)acc[p][q]);
        }
    }
}

// Kacc holds the tiles with J <= I in full; mirror into the dense N x N fp64 matrix the reference prints.
__global__ __launch_bounds__(256) void k_sim_finish(const unsigned long long *__restrict__ Kacc, int NS, int N, double *__restrict__ K)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * N) return;
    const int i = (int)(e / N), j = (int)(e % N);
    const bool lower = (i / SIM_TILE) >= (j / SIM_TILE);
    K[e] = (double)(lower ? Kacc[(int64_t)i * NS + j] : Kacc[(int64_t)j * NS + i]);
}

extern "C" {

hipError_t shk_sim_accumulate(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, int N, int NB64, double min_af, double max_af,
                              int af_on, uint64_t *keep, uint64_t *S, int NS, unsigned long long *Kacc)
{
    const int64_t VW = Vpad / 64, VWp = (VW + SIM_KW - 1) / SIM_KW * SIM_KW;
    hipLaunchKernelGGL(k_sim_keep, dim3((unsigned)(Vpad / 256)), dim3(256), 0, st, T, Vpad, V, N, NB64, min_af, max_af, af_on, keep);
    hipLaunchKernelGGL(k_sim_transpose, dim3((unsigned)((VWp + 3) / 4), (unsigned)(NS / 64)), dim3(256), 0, st, T, Vpad, NB64, VW, VWp, NS, keep, S);
    const int nt = NS / SIM_TILE, ntp = nt * (nt + 1) / 2;
    const int64_t nchunk = VWp / SIM_KW;
    // enough blocks to fill 256 CUs a few times over; every split walks at least 8 chunks (8192 variants)
    int ksplit = (int)((2048 + ntp - 1) / ntp);
    if (ksplit > nchunk / 8) ksplit = (int)(nchunk / 8);
    if (ksplit < 1) ksplit = 1;
    const int cps = (int)((nchunk + ksplit - 1) / ksplit);
    ksplit = (int)((nchunk + cps - 1) / cps);
    hipLaunchKernelGGL(k_sim_pairs, dim3((unsigned)ntp, (unsigned)ksplit), dim3(256), 0, st, S, NS, VWp, cps, Kacc);
    return hipGetLastError();
}

hipError_t shk_sim_finish(hipStream_t st, const unsigned long long *Kacc, int NS, int N, double *K)
{
    const int64_t n = (int64_t)N * N;
    hipLaunchKernelGGL(k_sim_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Kacc, NS, N, K);
    return hipGetLastError();
}

}  // extern "C"
