// bitdot_i8.hip -- carrier sums of per-run vectors on the int8 matrix cores.
//
// For every variant v and column c < NE:   out[c][v] = sum over the carriers i of v of vals[i][c]
// (vals = (w0_i, w0_i z_i0 .. w0_i z_i,Q-1, r0_i) at the null model; pyseer/model.py:316-344 fits the variant's model from there, and the
// information matrix of the finishing kernels is a difference FROM these sums -- glm_kernels.hip: k_glm_first_step, k_glm_dpass, k_glm_finish;
// firth_fast.hip takes the k-row of I from them).  Round 3 did this with nibble tables in LDS on the vector ALU (k_glm_bitdot, 0.5 ms per 2^18
// variants at N = 5000, LDS-bound); here it is a GEMM  bits (V x N, 0/1) x digits (N x 8 NE, int8):
//   * every column is scaled by a power of two to |x| < 2^62, rounded to an integer and written in EIGHT balanced base-256 digits in
//     [-128, 127]  (bitdot_i8_digits, host, once per run);  v_mfma_i32_32x32x32_i8 accumulates bit x digit exactly in int32
//     (|sum| <= 128 N), and the digits are recombined in fp64 by Horner's rule (exact until the value passes 2^53, one rounding per step
//     after that): the result is the exactly rounded sum of the 62-bit fixed-point values -- tighter than an fp64 running sum.
//   * the digit table is the A operand (M = 32 of the 8 NE digit columns), the variants' bits the B operand (N = 32 variants), expanded from
//     16 bits to 16 bytes in registers (nibble * 0x00204081 & 0x01010101); a wavefront owns 64 variants (two B tiles) and all digit columns
//     (NT = ceil(NE / 4) A tiles): 2 NT accumulator tiles in registers.  The 4 wavefronts of a workgroup share the A tiles through LDS
//     (a stage = 4 K-steps of 32 samples, double-buffered; one barrier per stage).
//   * digit columns are ordered so that the 16 accumulator registers of a lane are the 8 digits of TWO columns of ONE variant
//     (C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): the recombination needs no cross-lane traffic.
// Algorithmic bytes per variant: N / 8 of bits + 8 NE of output; the digit table (16 NE N bytes... 96 N at Q = 10) is read from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {
constexpr int KC = 4;                                                  // K-steps (of 32 samples) per LDS stage
}

// which (column-in-tile, digit) the A-tile row i carries: see the header (rows of one lane's registers = digits of two columns)
static inline void bitdot_i8_row(int i, int &c_local, int &limb)
{
    const int hi = (i >> 2) & 1, r = (i & 3) | ((i >> 3) << 2);
    c_local = 2 * hi + (r >> 3); limb = r & 7;
}

// Host: the digit table in the order the kernel's lanes load it, and the scale that turns the recombined integer back.
//   atab: nstage * KC * NT * 64 lanes * 16 bytes;  v4i index ((ks * NT + t) * 64 + lane), lane = h * 32 + i: bytes b < 16 = digit of sample
//   ks * 32 + h * 16 + b in the (column, digit) of row i of tile t.
extern "C" int64_t shk_bitdot_i8_table_bytes(int NE, int NB64) { return (int64_t)((2 * NB64 + KC - 1) / KC) * KC * ((NE + 3) / 4) * 64 * 16; }
extern "C" void shk_bitdot_i8_digits(const double *vals, int N, int NE, int NB64, int8_t *atab, double *scale /* 4 * ceil(NE / 4) */)
{
    const int NT = (NE + 3) / 4, nks = 2 * NB64;
    memset(atab, 0, (size_t)shk_bitdot_i8_table_bytes(NE, NB64));
    for (int c = 0; c < 4 * NT; ++c) scale[c] = 0.0;
    std::vector<int> ex(NE, 0);
    for (int c = 0; c < NE; ++c) {
        double m = 0.0;
        for (int i = 0; i < N; ++i) m = std::fmax(m, std::fabs(vals[(size_t)i * NE + c]));
        ex[c] = (m > 0.0 && std::isfinite(m)) ? std::ilogb(m) + 1 : 0;  // m < 2^ex
        scale[c] = std::ldexp(1.0, ex[c] - 62);
    }
    std::vector<int8_t> dig((size_t)N * NE * 8);
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < NE; ++c) {
            const double x = vals[(size_t)i * NE + c];
            long long X = std::isfinite(x) ? std::llrint(std::ldexp(x, 62 - ex[c])) : 0;      // |X| <= 2^62
            for (int l = 0; l < 8; ++l) {
                const int8_t d = l < 7 ? (int8_t)(uint8_t)(X & 0xFF) : (int8_t)X;                // balanced digit: the low byte read as signed
                dig[((size_t)i * NE + c) * 8 + l] = d;
                X = (X - d) >> 8;                                                                 // exact (X - d is a multiple of 256)
            }
        }
    for (int ks = 0; ks < nks; ++ks)
        for (int t = 0; t < NT; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                int cl, limb; bitdot_i8_row(i, cl, limb);
                const int c = 4 * t + cl;
                if (c >= NE) continue;
                int8_t *dst = atab + (((size_t)ks * NT + t) * 64 + lane) * 16;
                for (int b = 0; b < 16; ++b) {
                    const int s = ks * 32 + h * 16 + b;
                    if (s < N) dst[b] = dig[((size_t)s * NE + c) * 8 + limb];
                }
            }
}

template <int NT>
__global__ __launch_bounds__(256) void k_bitdot_i8(const uint64_t *__restrict__ T, int64_t Vpad, int NB64, int NE, const v4i *__restrict__ atab,
                                                  const double *__restrict__ scale, double *__restrict__ out)
{
    constexpr int STAGE = KC * NT * 64;                               // v4i per stage = 256 NT: NT per thread
    __shared__ v4i lds[2][STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t v0 = (int64_t)blockIdx.x * 256 + wave * 64;         // B tile b: variants v0 + 32 b + j
    v16i acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0;
    const int nks = 2 * NB64, nstage = (nks + KC - 1) / KC;
    v4i nxt[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) lds[0][q * 256 + tid] = atab[q * 256 + tid];
    __syncthreads();
    uint64_t w0 = 0, w1 = 0;
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nstage;
        if (more) {
#pragma unroll
            for (int q = 0; q < NT; ++q) nxt[q] = atab[(int64_t)(s + 1) * STAGE + q * 256 + tid];
        }
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int ks = s * KC + kk;
            if ((kk & 1) == 0) {                                       // (KC is even: a stage starts on a word boundary)
                const int wd = ks >> 1;
                if (wd < NB64) { w0 = T[(int64_t)wd * Vpad + v0 + j]; w1 = T[(int64_t)wd * Vpad + v0 + 32 + j]; } else { w0 = 0; w1 = 0; }
            }
            const int sh = (kk & 1) * 32 + h * 16;
            const uint32_t b0 = (uint32_t)(w0 >> sh) & 0xFFFFu, b1 = (uint32_t)(w1 >> sh) & 0xFFFFu;
            v4i B0, B1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                B0[q] = (int)((((b0 >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u);
                B1[q] = (int)((((b1 >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const v4i a = lds[cur][(kk * NT + t) * 64 + lane];
                acc[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1, acc[t][1], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < NT; ++q) lds[cur ^ 1][q * 256 + tid] = nxt[q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = 4 * t + 2 * h + half;
                if (c < NE) {
                    double S = (double)acc[t][b][half * 8 + 7];
#pragma unroll
                    for (int l = 6; l >= 0; --l) S = fma(S, 256.0, (double)acc[t][b][half * 8 + l]);
                    out[(int64_t)c * Vpad + v0 + 32 * b + j] = S * scale[c];
                }
            }
}

// T: NB64 words x Vpad variants (Vpad a multiple of 256, rows past the last variant zero); out: NE x Vpad doubles.
extern "C" hipError_t shk_bitdot_i8(hipStream_t st, const uint64_t *T, int64_t Vpad, int NB64, int NE, const void *atab, const double *scale, double *out)
{
    const dim3 grid((unsigned)(Vpad / 256)), blk(256);
    const int NT = (NE + 3) / 4;
    if (NT == 1) hipLaunchKernelGGL(k_bitdot_i8<1>, grid, blk, 0, st, T, Vpad, NB64, NE, (const v4i *)atab, scale, out);
    else if (NT == 2) hipLaunchKernelGGL(k_bitdot_i8<2>, grid, blk, 0, st, T, Vpad, NB64, NE, (const v4i *)atab, scale, out);
    else if (NT == 3) hipLaunchKernelGGL(k_bitdot_i8<3>, grid, blk, 0, st, T, Vpad, NB64, NE, (const v4i *)atab, scale, out);
    else if (NT == 4) hipLaunchKernelGGL(k_bitdot_i8<4>, grid, blk, 0, st, T, Vpad, NB64, NE, (const v4i *)atab, scale, out);   // Q = 11 .. 14 (GLM_MAXQ)
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
