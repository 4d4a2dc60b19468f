// glm_device.h -- device helpers shared by the register-resident fixed-effects kernels (glm_kernels.hip) and the Firth round kernels
// (firth_rounds.hip): packed-symmetric LDL^T, sample-split blocks, the scalar-load pipeline fence, exp(-x), the Firth state layout.
#pragma once
#include <algorithm>
#include "common.h"
#include "glm_common.h"

// ---- LDL^T of a packed symmetric P x P matrix, in place (no pivoting; tolerates indefinite matrices) ----------------
// returns false when a pivot is exactly zero (or NaN-free tiny relative to its diagonal when rel_tol > 0).
template <int P>
__device__ __forceinline__ bool ldl_factor(double (&A)[P * (P + 1) / 2], double rel_tol, double *det)
{
    bool ok = true;
    double dt = 1.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double v[P];
        const double ajj = A[sidx(j, j)];
        double d = ajj;
#pragma unroll
        for (int k = 0; k < j; ++k) { v[k] = A[sidx(j, k)] * A[sidx(k, k)]; d = fma(-A[sidx(j, k)], v[k], d); }
        if (d == 0.0 || fabs(d) <= rel_tol * fabs(ajj)) ok = false;
        A[sidx(j, j)] = d;
        dt *= d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = A[sidx(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-A[sidx(i, k)], v[k], s);
            A[sidx(i, j)] = s * inv;
        }
    }
    *det = dt;
    return ok;
}

template <int P>
__device__ __forceinline__ void ldl_solve(const double (&A)[P * (P + 1) / 2], double (&b)[P])
{
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] = fma(-A[sidx(i, k)], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) b[i] = b[i] / A[sidx(i, i)];
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
#pragma unroll
        for (int k = i + 1; k < P; ++k) b[i] = fma(-A[sidx(k, i)], b[k], b[i]);
    }
}


// ---- sample-split blocks: S wavefronts share the same 64 variants (lane = variant) and each walks every S-th 64-sample word --------
// The list-driven kernels (Firth rounds, the fp64 restart of the Newton iteration) see anything from one to 10^5 variants per
// launch; with one wavefront per 64 variants a short list is bound by the latency of one lane walking all N samples (3 ms per
// pass at N = 5000, however few variants there are).  S depends only on N (glm_split_waves), never on the length of a list, so a
// variant's partial sums are combined in the same order whatever else is in its batch: results do not depend on batch composition.
// Partial sums go through LDS in chunks of XW_CH accumulators: wave 0 adds waves 1..S-1 in that order.
#define XW_CH 16
extern __shared__ double xw_lds[];
struct XWave { int w, lane, S; };
// readfirstlane: the wavefront index is uniform within a wavefront, but the compiler cannot know it; without this every address
// derived from it (the sample index, hence the covariate rows) is treated as divergent and loaded per lane instead of through SGPRs
__device__ __forceinline__ XWave xwave()
{
    return XWave{__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), (int)(blockDim.x >> 6)};
}

// wavefronts per 64 variants in the list-driven kernels: a function of the sample count only, so that results do not depend
// on what else is in a batch.
static int glm_split_waves(int NB64)
{
    return NB64 >= 64 ? 8 : NB64 >= 8 ? 4 : NB64 >= 4 ? 2 : 1;      // measured at N = 500 ... 5000
}
static size_t glm_split_lds(int S) { return S > 1 ? (size_t)((S - 1) * XW_CH + 2) * 64 * sizeof(double) : 0; }
__device__ __forceinline__ int glm_split_lds_dev(int S) { return S > 1 ? ((S - 1) * XW_CH + 2) * 64 * (int)sizeof(double) : 0; }

template <int NA>
__device__ __forceinline__ void xw_sum(const XWave &x, double (&a)[NA])            // block-uniform call; total valid in wave 0
{
    if (x.S == 1) return;
#pragma unroll
    for (int c0 = 0; c0 < NA; c0 += XW_CH) {
        if (x.w > 0) {
#pragma unroll
            for (int k = 0; k < XW_CH; ++k) if (c0 + k < NA) xw_lds[((x.w - 1) * XW_CH + k) * 64 + x.lane] = a[c0 + k];
        }
        __syncthreads();
        if (x.w == 0) {
            for (int ww = 1; ww < x.S; ++ww) {
#pragma unroll
                for (int k = 0; k < XW_CH; ++k) if (c0 + k < NA) a[c0 + k] += xw_lds[((ww - 1) * XW_CH + k) * 64 + x.lane];
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void xw_sum_max(const XWave &x, double &sum, double &mx)  // sum and max of two scalars, as xw_sum
{
    if (x.S == 1) return;
    if (x.w > 0) { xw_lds[((x.w - 1) * XW_CH + 0) * 64 + x.lane] = sum; xw_lds[((x.w - 1) * XW_CH + 1) * 64 + x.lane] = mx; }
    __syncthreads();
    if (x.w == 0) {
        for (int ww = 1; ww < x.S; ++ww) {
            sum += xw_lds[((ww - 1) * XW_CH + 0) * 64 + x.lane];
            mx = fmax(mx, xw_lds[((ww - 1) * XW_CH + 1) * 64 + x.lane]);
        }
    }
    __syncthreads();
}

template <int NA>
__device__ __forceinline__ void xw_bcast(const XWave &x, double (&a)[NA], bool &flag)   // wave 0 -> every wave (NA <= XW_CH)
{
    static_assert(NA <= XW_CH, "broadcast area is one chunk plus the flag row");
    if (x.S == 1) return;
    if (x.w == 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) xw_lds[k * 64 + x.lane] = a[k];
        xw_lds[NA * 64 + x.lane] = flag ? 1.0 : 0.0;
    }
    __syncthreads();
    if (x.w > 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) a[k] = xw_lds[k * 64 + x.lane];
        flag = xw_lds[NA * 64 + x.lane] != 0.0;
    }
    __syncthreads();
}

// Scalar loads return out of order, so the only wait the compiler can place for them is "all of them" (s_waitcnt lgkmcnt(0)).  A software
// pipeline over two wave-uniform buffers therefore needs the wait for buffer A to sit BEFORE the loads of buffer B are issued; left alone,
// the scheduler hoists B's loads above A's first use and the wait then covers both.  pipe_zero(x) is a zero the compiler cannot see
// through, computed from one of A's SGPRs: added to B's index, it orders B's loads behind A's arrival and keeps them scalar.
// (s_and_b32 writes SCC: declared, or a compare scheduled across the asm loses its result)
__device__ __forceinline__ int pipe_zero(double x) { int z; asm("s_and_b32 %0, %1, 0" : "=s"(z) : "s"(__double2loint(x)) : "scc"); return z; }
// pipe_zero that stays BEHIND a piece of arithmetic: the plain form has no side effects, and the scheduler may hoist it up to the load it
// reads from -- the wait it was meant to place late then sits right behind the load.  `after` is an unused vector operand: a result of the
// arithmetic the fence has to follow.  Being volatile it also takes the "never clobbered" property from the kernel's global loads, so the
// records it orders must be read through the constant address space (firth_rounds.hip) or they turn into per-lane vector loads.  Tried on
// k_glm_pass32 / k_glm_dpass_pk / k_glm_score / k_glm_ll in round 3: no measurable change (3.150 vs 3.145 ms), those kernels are not
// waiting for their records; not kept there.
__device__ __forceinline__ int pipe_zero_after(double x, double after)
{
    int z; asm volatile("s_and_b32 %0, %1, 0" : "=s"(z) : "s"(__double2loint(x)), "v"(after) : "scc"); return z;
}
__device__ __forceinline__ int pipe_zero(float x) { int z; asm("s_and_b32 %0, %1, 0" : "=s"(z) : "s"(__float_as_int(x)) : "scc"); return z; }

// e^-x for x >= 0: k = rint(-x log2 e), r = -x - k ln 2 (two-part ln 2), |r| <= 0.3466, Taylor to degree 13 (remainder 4e-18), 2^k by v_ldexp
__device__ __forceinline__ double exp_neg(double x)
{
    const double u = -fmin(x, 800.0);
    const double kf = rint(u * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, u);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                       // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                     // 1/12!
    p = fma(p, r, 2.505210838544172e-08);                    // 1/11!
    p = fma(p, r, 2.755731922398589e-07);                    // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);                   // 1/9!
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

// wave-aggregated append: one atomic per wavefront; callable from divergent code (the leader is one of the active lanes)
__device__ __forceinline__ void list_push(bool p, int *__restrict__ list, int *__restrict__ count, int v)
{
    const unsigned long long m = __ballot(p);
    if (!m) return;
    const int lane = threadIdx.x & 63, lead = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == lead) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, lead);
    if (p) list[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

// workgroup-aggregated append: ONE atomic per workgroup.  Every thread of the workgroup must call it (uniform control flow; blockDim.x a
// multiple of 64, at most 1024); lds = 17 ints of shared memory, free again when the call returns.  With one wavefront per workgroup a
// 2^20-variant round issues 16 384 same-address atomics per list, and those serialise in the L2: they, not the arithmetic, set the time of
// the per-variant kernels at N = 1000.
__device__ __forceinline__ void list_push_block(bool p, int *__restrict__ list, int *__restrict__ count, int v, int *lds)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long m = __ballot(p);
    if (lane == 0) lds[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < nw; ++i) { const int c = lds[i]; lds[i] = tot; tot += c; }
        lds[16] = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (p) list[lds[16] + lds[w] + __popcll(m & ((1ull << lane) - 1ull))] = v;
    __syncthreads();
}

struct FirthWork {
    double *st;                 // [FW_ND(PC)][cap]
    int *iter, *halv, *var;     // [cap] accepted steps (-1 = initial evaluation pending), halvings of the current step, variant index
    int64_t cap;
    int *blk_list, *blk_count;  // slots handed to k_firth_blk after GlmParams.firth_handoff accepted steps
    int *s64_list, *s64_count;  // slots whose score pass stays all-fp64 (k_firth_step<Q, true>): pivot guard of k_firth_eval2; may be null
};
// A variant still iterating after `firth_handoff` accepted steps leaves the rounds and is finished by one workgroup (k_firth_blk).
// The rule looks at the variant alone, so which kernel finishes a variant -- and hence the order of its sums -- does not depend
// on what else is in the batch.  Ordinary variants converge in 5-14 steps; (quasi-)separated ones need hundreds.
// (GlmParams.firth_handoff: 16, or 0 for the routed variants of an ordinary run at N >= 768, see sh_glm_setup)
template <int PC> __host__ __device__ constexpr int fw_beta() { return 0; }
template <int PC> __host__ __device__ constexpr int fw_cand() { return PC; }
template <int PC> __host__ __device__ constexpr int fw_fac() { return 2 * PC; }
template <int PC> __host__ __device__ constexpr int fw_fcur() { return 2 * PC + PC * (PC + 1) / 2; }
template <int PC> __host__ __device__ constexpr int fw_i11() { return fw_fcur<PC>() + 1; }
template <int PC> __host__ __device__ constexpr int fw_snp() { return fw_fcur<PC>() + 2; }
template <int PC> __host__ __device__ constexpr int fw_nd() { return fw_fcur<PC>() + 3; }

