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
