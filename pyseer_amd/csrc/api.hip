// api.hip -- host side of libseerhip.so: the flat C ABI declared in include/seerhip.h.
// One-off per-run setup is done here in plain C++ (it is O(N^2 D) at most); everything per-variant is a HIP kernel.
#include "common.h"
#include "route.h"
#include "host_pool.h"
#include <atomic>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <string>
#include <vector>
#include <thread>
#include <future>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <algorithm>

// ---- kernel launchers (lmm_kernels.hip / glm_kernels.hip) -----------------------------------------------------
#include "lmm_params.h"
extern "C" {
hipError_t shk_sim_accumulate(hipStream_t st, const uint64_t *T, int64_t Vpad, int64_t V, int N, int NB64, double min_af, double max_af,
                              int af_on, uint64_t *keep, uint64_t *S, int NS, unsigned long long *Kacc);
hipError_t shk_sim_finish(hipStream_t st, const unsigned long long *Kacc, int NS, int N, double *K);
hipError_t shk_repack_bits(hipStream_t, const uint8_t *, int64_t, int64_t, int64_t, int, int, uint64_t *, uint8_t *);
hipError_t shk_lmm_linear(hipStream_t, int, const uint64_t *, int64_t, int, int, const double *, const double *,
                          const double *, const double *, const uint64_t *, const uint64_t *, int, const double *, LmmLinOut);
hipError_t shk_lmm_build_tab(hipStream_t, const double *, const double *, const double *, const double *, int, int, int, int, double *);
hipError_t shk_lmm_quadform(hipStream_t, int, const int8_t *, const uint64_t *, int64_t, int, int, int, double *, const int *, int, int64_t *);
hipError_t shk_lmm_refine(hipStream_t, int64_t, int64_t, int, int, int, int, const int8_t *, const uint64_t *, uint64_t *, double *, LmmLinOut,
                          const double *, LmmFinParams, double *, uint32_t *, LmmRefine);
hipError_t shk_af_compact(hipStream_t, int, int64_t, LmmLinOut, LmmFinParams, int *, int *, const uint64_t *, int64_t, uint64_t *,
                          int64_t, int, int, const double *, double *);
hipError_t shk_lmm_finalize(hipStream_t, int64_t, int64_t, int, LmmLinOut, const double *, LmmFinParams, double *, uint32_t *, LmmRefine);
hipError_t shk_spectral_bound(hipStream_t, const float *, int, int, double *, double *, int, double *, int *);
hipError_t shk_power_norm(hipStream_t, const float *, int, int, double *, double *, double *, int);
hipError_t shk_lmm_build_G(hipStream_t, const double *, const double *, int, int, int, int, int, int, double *, double *,
                           unsigned long long *, int8_t *, float *, double *, double *, double *, int);
hipError_t shk_dd_find(hipStream_t, const uint64_t *, int64_t, int64_t, int, uint64_t *, uint64_t, unsigned long long *, int *, int *, int *, int *);
hipError_t shk_af_rows(hipStream_t, int, const uint8_t *, int64_t, int64_t, int, double, double, int *, int *, int *);
hipError_t shk_pf_rows(hipStream_t, const uint8_t *, int64_t, int64_t, int, const uint64_t *, const uint64_t *, double, double, int, int, int,
                       double, int, int *, int *, int *, int *, int *, int *, double *, uint32_t *);
hipError_t shk_dd_gather(hipStream_t, const uint8_t *, int64_t, int64_t, const int *, const int *, uint8_t *);
hipError_t shk_dd_scatter(hipStream_t, int64_t, int64_t, int, const int *, const int *, const double *, const uint32_t *, double *, uint32_t *);
hipError_t shk_job_select(hipStream_t, const uint32_t *, const double *, int64_t, int, int, int, int *, long long *, long long *, int32_t *, uint32_t *, double *, int64_t, int32_t *);
int64_t sh_format_records(const char *, const int64_t *, const int32_t *, int, const int32_t *, int64_t, const double *const *, int, const double *, int64_t, int,
                          const uint8_t *, const int32_t *, const char *const *, int, const uint32_t *, const char **);
// (writer.cpp: sh_format_records + the two sample lists of --print-samples from the printed rows' host bits)
int64_t format_records_samples(const char *, const int64_t *, const int32_t *, int, const int32_t *, int64_t, const double *const *, int, const double *, int64_t, int,
                               const uint8_t *, const int32_t *, const char *const *, int, const uint32_t *, const char **, const uint8_t *, int64_t, const char *,
                               const int64_t *, const int32_t *);
hipError_t shk_job_patterns(hipStream_t, const uint64_t *, int64_t, int64_t, int, const uint32_t *, int *, long long *, uint32_t *, char *, char *);
#define JOB_ROWS_PER_BLOCK_HOST 1024
}
#include "glm_api.inc"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SH_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// SEERHIP_DEBUG=backtrace: a native backtrace on stderr when the process aborts (SIGABRT: an assertion of a runtime underneath, glibc's heap
// checks, std::terminate) or faults -- Python's faulthandler shows the interpreter's frames only.
#include <execinfo.h>
#include <signal.h>
static void sh_abort_backtrace(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "[seerhip] fatal signal; native backtrace of the signalled thread:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
namespace { struct ShBacktraceInit { ShBacktraceInit() { if (sh_debug("backtrace")) { signal(SIGABRT, sh_abort_backtrace); signal(SIGSEGV, sh_abort_backtrace); signal(SIGBUS, sh_abort_backtrace); } } } sh_backtrace_init; }

// (SEERHIP_DEBUG=poison: every allocation filled with 0xFF bytes -- NaN as a float or a double, -1 as an index -- so that a kernel that reads what
// nobody wrote fails a test at once instead of when the allocator happens to hand out a dirty page: tools/gpu_poison.sh)
template <typename T> static hipError_t dmalloc(T **p, size_t n)
{
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
    if (e == hipSuccess && n && sh_debug("poison")) {               // (complete before anybody's stream can write the buffer)
        const hipError_t m = hipMemset(*p, 0xFF, n * sizeof(T));
        return m != hipSuccess ? m : hipStreamSynchronize(nullptr);
    }
    return e;
}

struct sh_lanes;
// what sh_glm_setup was called with: the lanes of the context (lanes_api.inc) are set up from it
struct GlmSetupArgs { std::vector<double> y, W; int q = 0, continuous = 0, force_firth = 0; double null_llf = 0, null_firth = 0, pret = 1, lrtt = 1;
                      std::string route; };   // (route: the SEERHIP_ROUTE string the set-up ran under)
struct sh_ctx {
    int device = 0, N = 0;
    hipStream_t stream = nullptr;
    int *h_rb = nullptr;                 // host-mapped pinned ints for readback_sync
    // ---- lanes (sh_glm_batch_dev_async, the job stream): worker threads with their own sub-context and stream, lanes_api.inc
    sh_lanes *lanes = nullptr; int n_lanes = 3; bool is_lane = false; GlmSetupArgs glm_args;
    double min_af = 0.0, max_af = 1.0; int af_on = 0;
    // common per-run constants
    int NT = 0, Np = 0, NB64 = 0, NB64p = 0;
    // ---- LMM state
    bool lmm_ready = false;
    int k = 0, D = 0, L = 5, DP = 0;
    LmmFinParams fin{};
    double *d_vv = nullptr, *d_mdiag = nullptr, *d_yc = nullptr, *d_Qb = nullptr, *d_tab = nullptr;   // d_tab: nibble tables of (vv, mdiag)
    uint64_t *d_y1 = nullptr, *d_y0 = nullptr;
    int8_t *d_G = nullptr;
    double quant_scale = 0.0;
    size_t g_bytes = 0, tab_doubles = 0;
    int64_t macs_issued = 0;      // int8 MACs per variant of the last main-pass contraction (sh_lmm_info)
    int E = 0;                    // extra (low) limbs stored below the L of the main pass; contracted only for variants whose bound exceeds lmm_tol
    bool complement = false;      // rows with more than N/2 carriers are stored complemented (needs the intercept in the covariate span)
    double err_norm_ulp = 0.0, err_norm_est_ulp = 0.0, lmm_tol = 1e-8, trace_M = 0.0; int err_norm_squarings = 0;
    uint8_t *d_flip = nullptr; uint64_t *d_T3 = nullptr; double *d_q3 = nullptr; int *d_rlist = nullptr, *d_rcount = nullptr;
    unsigned long long *d_bmax = nullptr; int64_t cap_ref = 0;
    int qf_split = 1;             // 1: one block per (variant tile, limb) (0: one block per tile loops over the limbs; not selectable any more)
    int qf_variant = 4;           // hot-kernel variant: 4 = k_lmm_quadform_i8w (one wavefront per SIMD; the launcher falls back to 0 where its conditions do not
                                  // hold), SEERHIP_ROUTE qf=0 = k_lmm_quadform_i8 (two wavefronts per SIMD), other values = timing ablations (lmm_kernels.hip)
    // ---- GLM state
    GlmState glm;
    // ---- per-batch workspace (grown on demand)
    int64_t capV = 0;
    uint64_t *d_T = nullptr;
    int *d_t11 = nullptr, *d_t01 = nullptr, *d_m = nullptr;
    double *d_xky = nullptr, *d_dg = nullptr, *d_rss = nullptr, *d_s1 = nullptr, *d_q1 = nullptr, *d_q = nullptr;
    uint64_t *d_T2 = nullptr; double *d_q2 = nullptr; int *d_keep = nullptr, *d_nkeep = nullptr; int64_t cap_keep = 0;   // AF compaction
    int af_compact = 1; int64_t last_kept = -1;
    int *h_nkeep = nullptr; hipEvent_t keep_ev = nullptr; bool keep_pending = false; int64_t keep_V = 0; double filtered_hint = 0.0;
    // ---- optional timing of the dominant kernel (sh_set_timing / sh_get_timing)
    int timing = 0;
    int lin_tab = 1;                                  // 0 would be the per-sample k_lmm_linear instead of the nibble tables
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;
    // ---- pattern de-duplication (sh_set_dedup)
    int dedup = 0; int64_t dd_cap = 0, dd_capV = 0, dd_last_unique = -1;
    uint64_t *dd_h = nullptr; unsigned long long *dd_keys = nullptr; int *dd_idx = nullptr, *dd_rep = nullptr, *dd_slot = nullptr, *dd_n = nullptr;
    uint8_t *dd_bits = nullptr; double *dd_out = nullptr; uint32_t *dd_flags = nullptr; int64_t dd_cap_bits = 0, dd_cap_out = 0, dd_cap_flags = 0;
    // row-level AF compaction of the fixed-effects path (af_wrap): its own buffers, because the wrapped call may de-duplicate
    int *af_rep = nullptr, *af_slot = nullptr, *af_cnt = nullptr, *h_af_cnt = nullptr; int64_t af_capV = 0;
    int *af_m = nullptr;                                          // [3][af_capV]: carrier count and the two case/control cells of each row
    uint8_t *af_bits = nullptr; double *af_out = nullptr; uint32_t *af_flags = nullptr; int64_t af_cap_bits = 0, af_cap_out = 0, af_cap_flags = 0;
    hipEvent_t af_ev = nullptr; bool af_pending = false; int64_t af_pending_V = 0; double af_hint = 0.0; int64_t af_last_rows = -1; unsigned af_tick = 0;
    // ---- similarity accumulation (sim_kernels.hip)
    unsigned long long *sim_K = nullptr; uint64_t *sim_S = nullptr, *sim_keep = nullptr; double *sim_out = nullptr; int64_t sim_capV = 0; int NS = 0;
    // ---- pipelined host-pointer batches (host_batch)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    uint8_t *hb_bits[2] = {nullptr, nullptr}; double *hb_out[2] = {nullptr, nullptr}; uint32_t *hb_flags[2] = {nullptr, nullptr};
    int64_t hb_cap_bits = 0, hb_cap_out = 0, hb_cap_flags = 0;
    uint8_t *hp_bits[2] = {nullptr, nullptr}; int64_t hp_cap = 0;     // pinned host staging (pageable user rows are copied in by several threads)
    // the chunk whose kernels are queued and whose results have not been copied back yet: within a call the previous chunk, across
    // *_batch_async calls the last chunk of the previous call (its result pointers are the caller's, alive until sh_wait / the next call)
    struct HostPending { bool valid = false; int b = 0, q = 0; int64_t s = 0, n = 0; double *outs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
                         double *betas = nullptr; uint32_t *flags = nullptr; } pend;
    uint64_t hb_seq = 0;                                              // chunks queued so far: chunk k uses staging set k & 1
    // rows announced for the NEXT host-pointer batch (sh_prefetch_rows) and, once the current batch's copy thread has uploaded their first
    // chunk, where that chunk sits
    const uint8_t *next_bits = nullptr; int64_t next_row_bytes = 0, next_V = 0;
    const uint8_t *pre_bits = nullptr; int64_t pre_row_bytes = 0, pre_n = 0; int pre_set = -1;
    // ---- staging for the host-pointer entry points
    int64_t cap_bits = 0, cap_out = 0, cap_flags = 0;
    uint8_t *d_bits = nullptr; double *d_out = nullptr; uint32_t *d_flags = nullptr;
};

// ---- small device -> host read-backs (list lengths, counts) WITHOUT the copy engine ------------------------------------------------------------
// hipMemcpyAsync(device -> host) of four bytes goes through the same SDMA queue as the job stream's uploads: behind a burst of four 166 MB
// block copies (2.9 ms each) a lane's list-length read-back waited 11 ms (tools/gpu_e2e_job_copytrace.sh: the gaps inside a batch), and the
// fixed-effects job ran at 26 M rows/s with the device idle a quarter of the time.  A one-thread kernel on the batch's own stream writes the
// values into host-mapped pinned memory instead; the host reads them after the stream's (or an event's) wait.
__global__ void k_readback4(const int *__restrict__ a, const int *__restrict__ b, const int *__restrict__ c2, const int *__restrict__ d, int *__restrict__ out)
{
    if (threadIdx.x == 0) { out[0] = *a; if (b) out[1] = *b; if (c2) out[2] = *c2; if (d) out[3] = *d; __threadfence_system(); }
}
static hipError_t readback_launch(hipStream_t st, const int *a, const int *b, const int *c2, const int *d, int *host_mapped)
{
    hipLaunchKernelGGL(k_readback4, dim3(1), dim3(64), 0, st, a, b, c2, d, host_mapped);
    return hipGetLastError();
}
static int ensure_rb(sh_ctx *c);
// values of up to four device ints, after everything queued on st: launches the kernel, waits for the stream
static int readback_sync(sh_ctx *c, hipStream_t st, const int *a, const int *b, const int *c2, const int *d, int *dst);

static void lanes_destroy(sh_ctx *c);
static int lanes_wait(sh_ctx *c);
static void lanes_set_timing(sh_ctx *c, int on);
static int lanes_add_timing(sh_ctx *c, double *total_ms, int64_t *launches);
static void free_ws(sh_ctx *c)
{
    hipFree(c->d_T); hipFree(c->d_t11); hipFree(c->d_t01); hipFree(c->d_m); hipFree(c->d_xky); hipFree(c->d_dg);
    hipFree(c->d_rss); hipFree(c->d_s1); hipFree(c->d_q1); hipFree(c->d_q);
    c->d_T = nullptr; c->d_t11 = c->d_t01 = c->d_m = nullptr;
    c->d_xky = c->d_dg = c->d_rss = c->d_s1 = c->d_q1 = c->d_q = nullptr; c->capV = 0;
    hipFree(c->d_T2); hipFree(c->d_q2); hipFree(c->d_keep); hipFree(c->d_nkeep);
    c->d_T2 = nullptr; c->d_q2 = nullptr; c->d_keep = c->d_nkeep = nullptr; c->cap_keep = 0;
    hipFree(c->d_flip); hipFree(c->d_T3); hipFree(c->d_q3); hipFree(c->d_rlist);
    c->d_flip = nullptr; c->d_T3 = nullptr; c->d_q3 = nullptr; c->d_rlist = nullptr; c->cap_ref = 0;
}

static int ensure_rb(sh_ctx *c)
{
    if (!c->h_rb) HIPCHK(hipHostMalloc((void **)&c->h_rb, 16 * sizeof(int), hipHostMallocMapped));
    return SH_OK;
}
static int readback_sync(sh_ctx *c, hipStream_t st, const int *a, const int *b, const int *c2, const int *d, int *dst)
{
    int rc = ensure_rb(c); if (rc) return rc;
    HIPCHK(readback_launch(st, a, b, c2, d, c->h_rb));
    HIPCHK(hipStreamSynchronize(st));
    dst[0] = c->h_rb[0]; if (b) dst[1] = c->h_rb[1]; if (c2) dst[2] = c->h_rb[2]; if (d) dst[3] = c->h_rb[3];
    return SH_OK;
}

static int ensure_ws(sh_ctx *c, int64_t Vpad)
{
    if (Vpad <= c->capV) return SH_OK;
    free_ws(c);
    HIPCHK(dmalloc(&c->d_T, (size_t)Vpad * c->NB64p));
    HIPCHK(dmalloc(&c->d_t11, Vpad)); HIPCHK(dmalloc(&c->d_t01, Vpad)); HIPCHK(dmalloc(&c->d_m, Vpad));
    HIPCHK(dmalloc(&c->d_xky, Vpad)); HIPCHK(dmalloc(&c->d_dg, Vpad)); HIPCHK(dmalloc(&c->d_rss, Vpad));
    HIPCHK(dmalloc(&c->d_s1, Vpad)); HIPCHK(dmalloc(&c->d_q1, Vpad)); HIPCHK(dmalloc(&c->d_q, Vpad * 8));
    c->capV = Vpad;
    return SH_OK;
}

// Runs `inner` (sh_lmm_batch_dev / sh_glm_batch_dev body) on the distinct patterns only and fans the results out.
static int ensure_staging(sh_ctx *c, int64_t bits_bytes, int64_t out_doubles, int64_t V)
{
    if (bits_bytes > c->cap_bits) { hipFree(c->d_bits); c->d_bits = nullptr; HIPCHK(hipMalloc((void **)&c->d_bits, bits_bytes)); c->cap_bits = bits_bytes; }
    if (out_doubles > c->cap_out) { hipFree(c->d_out); c->d_out = nullptr; HIPCHK(dmalloc(&c->d_out, out_doubles)); c->cap_out = out_doubles; }
    if (V > c->cap_flags) { hipFree(c->d_flags); c->d_flags = nullptr; HIPCHK(dmalloc(&c->d_flags, V)); c->cap_flags = V; }   // its own capacity: rows, not rows x columns
    return SH_OK;
}

// user rows -> pinned staging: the calling thread and the idle workers of the process-wide pool (host_pool.h), at most this stream's share of
// the CPU budget.  (Round 4 started up to 16 std::threads per chunk per context, sized from hardware_concurrency(): 256 on the GPU boxes.)
static void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    shost::CpuScope cs(shost::ST_STAGE_COPY);
    const size_t piece = (size_t)4 << 20, np = (n + piece - 1) / piece;
    if (np <= 1) { std::memcpy(dst, src, n); return; }
    const std::function<void(int64_t)> cp = [&](int64_t i) { const size_t lo = (size_t)i * piece, hi = std::min(n, lo + piece); std::memcpy(dst + lo, src + lo, hi - lo); };
    shost::pool().run((int64_t)np, 1, cp, shost::ST_STAGE_COPY, shost::per_stream_cpus(2));
}

// Host-pointer batches, pipelined: the packed rows of chunk i+1 cross PCIe on a copy stream while chunk i runs its kernels
// on the compute stream (two device staging sets; the results of chunk i-1 are copied back after chunk i has been queued).
// outs[a] receives row a of the (nrow x V) SoA result; rows >= 5 of the GLM result are delivered row-major through `betas`.
// With `async` the last chunk is left on the device (c->pend) and copied back by the next call -- after that call has queued its own
// first chunk, so the device never waits for the host between calls -- or by sh_wait.
// covariate slopes of a chunk, (q x n) as the kernels leave them -> (n x q) as the caller's `betas` rows are laid out: one contiguous
// device-to-host copy instead of a strided scatter on the host (5 ms per 131 072-row chunk at q = 10, more than the kernels of the chunk)
__global__ void k_betas_rows(const double *__restrict__ src, double *__restrict__ dst, int64_t n, int q)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * q) return;
    const int64_t v = i / q; const int j = (int)(i - v * q);
    dst[i] = src[(size_t)j * n + v];
}

// An announcement (sh_prefetch_rows) and a first chunk uploaded ahead belong to the call sequence they were made in: a set-up, a change of
// stream or a reallocation of the staging buffers forgets them (the next batch then uploads its own first chunk).
static void forget_announced(sh_ctx *c) { c->next_bits = nullptr; c->pre_bits = nullptr; c->pre_set = -1; }

static int drain_pending(sh_ctx *c)
{
    if (!c->pend.valid) return SH_OK;
    HIPCHK(hipSetDevice(c->device));
    const sh_ctx::HostPending p = c->pend;
    c->pend.valid = false;
    const int b = p.b;
    const int64_t s = p.s, n = p.n;
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_done[b], 0));      // chunk's results: copy stream, after its kernels (not behind the next chunk's)
    for (int a = 0; a < 5; ++a)
        HIPCHK(hipMemcpyAsync(p.outs[a] + s, c->hb_out[b] + (size_t)a * n, sizeof(double) * n, hipMemcpyDeviceToHost, c->copy_stream));
    if (p.q > 0)                                                       // rows [s, s + n) of the caller's (V x q) array, transposed on the device (k_betas_rows)
        HIPCHK(hipMemcpyAsync(p.betas + (size_t)s * p.q, c->hb_out[b] + (size_t)(5 + p.q) * n, sizeof(double) * p.q * n, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipMemcpyAsync(p.flags + s, c->hb_flags[b], sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    return SH_OK;
}

template <typename F>
static int host_batch(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V, int nrow, double *const *outs, double *betas,
                      int q, uint32_t *flags, bool async, F inner_dev)
{
    const double t_entry = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    HIPCHK(hipSetDevice(c->device));
    const int64_t CH = 1 << 18;
    const int64_t cap = std::min(CH, V);
    if (!c->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) { HIPCHK(hipEventCreateWithFlags(&c->ev_h2d[b], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming)); }
    }
    if (cap * row_bytes > c->hb_cap_bits || cap * (nrow + q) > c->hb_cap_out || cap > c->hb_cap_flags) {     // each buffer against its own capacity
        int rc = drain_pending(c); if (rc) return rc;                  // (the staging sets are about to be replaced)
        const int64_t nb = std::max(cap * row_bytes, c->hb_cap_bits), no = std::max(cap * (nrow + q), c->hb_cap_out), nf = std::max(cap, c->hb_cap_flags);   // + q rows: the slopes row-major
        for (int b = 0; b < 2; ++b) {
            hipFree(c->hb_bits[b]); hipFree(c->hb_out[b]); hipFree(c->hb_flags[b]);
            c->hb_bits[b] = nullptr; c->hb_out[b] = nullptr; c->hb_flags[b] = nullptr;
            HIPCHK(hipMalloc((void **)&c->hb_bits[b], nb)); HIPCHK(dmalloc(&c->hb_out[b], no)); HIPCHK(dmalloc(&c->hb_flags[b], nf));
        }
        c->hb_cap_bits = nb; c->hb_cap_out = no; c->hb_cap_flags = nf;
        c->pre_bits = nullptr; c->pre_set = -1;                        // a first chunk uploaded ahead lived in the buffers just freed
    }
    if (cap * row_bytes > c->hp_cap) {
        int rc = drain_pending(c); if (rc) return rc;
        for (int b = 0; b < 2; ++b) { if (c->hp_bits[b]) hipHostFree(c->hp_bits[b]); c->hp_bits[b] = nullptr; HIPCHK(hipHostMalloc((void **)&c->hp_bits[b], cap * row_bytes, hipHostMallocDefault)); }
        c->hp_cap = cap * row_bytes;
        c->pre_bits = nullptr; c->pre_set = -1;
    }
    // chunk boundaries: a shorter first chunk (2^17) so that the kernels start after 83 MB of upload instead of 166 MB; 2^17 and
    // 2^18 variants are whole numbers of block rounds for the LMM kernel (tiles x 5 limbs on 256 CUs)
    std::vector<int64_t> cut{0};
    for (int64_t step = 1 << 17; cut.back() < V; step = CH) cut.push_back(std::min(V, cut.back() + step));
    const int64_t nchunk = (int64_t)cut.size() - 1;
    const bool dbg = sh_debug("host");
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint8_t *const nb_bits = c->next_bits; const int64_t nb_rb = c->next_row_bytes, nb_V = c->next_V;   // (an announcement serves one batch)
    c->next_bits = nullptr;
    const double t00 = now();
    // rows of chunk i -> pinned staging set b -> device, on the copy stream (asynchronous past the staging copy)
    auto upload = [&](int64_t i, int b) -> int {
        const int64_t s = cut[i], n = cut[i + 1] - s;
        parallel_copy(c->hp_bits[b], bits + s * row_bytes, (size_t)(n * row_bytes));
        HIPCHK(hipMemcpyAsync(c->hb_bits[b], c->hp_bits[b], n * row_bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(hipEventRecord(c->ev_h2d[b], c->copy_stream));
        return SH_OK;
    };
    // While chunk i's kernels are queued / run on this thread (the fixed-effects rounds read list lengths back: their launch code blocks),
    // a side thread copies back the chunk before (c->pend: chunk i-1, or the previous call's last chunk) and then stages and uploads
    // chunk i+1 into the staging set that drain has just freed.  Without it a fixed-effects chunk was upload + kernels + download in a row.
    for (int64_t i = 0; i < nchunk; ++i) {
        const int b = (int)(c->hb_seq++ & 1);
        const int64_t s = cut[i], n = cut[i + 1] - s;
        const double t0 = now();
        if (i == 0) {                                                  // (later chunks were uploaded by the side thread of the chunk before)
            if (c->pend.valid && c->pend.b == b) { int rc = drain_pending(c); if (rc) return rc; }   // (cannot happen while the sets alternate)
            // ... and so was this one if the caller announced these rows before the previous batch (sh_prefetch_rows)
            bool have = c->pre_bits == bits && c->pre_row_bytes == row_bytes && c->pre_n == n && c->pre_set == b;
            if (have) {                                                // same pointer: and still the same rows at both ends of the chunk?
                const size_t nbytes = (size_t)(n * row_bytes), probe = std::min<size_t>(nbytes, 4096);
                have = std::memcmp(c->hp_bits[b], bits, probe) == 0 && std::memcmp(c->hp_bits[b] + nbytes - probe, bits + nbytes - probe, probe) == 0;
            }
            c->pre_bits = nullptr; c->pre_set = -1;
            if (!have) { int rc = upload(0, b); if (rc) return rc; }
        }
        const double t1 = now();
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_h2d[b], 0));
        std::string side_err;
        std::future<int> side = std::async(std::launch::async, [&]() -> int {
            if (hipSetDevice(c->device) != hipSuccess) { side_err = "hipSetDevice failed on the copy thread"; return SH_EHIP; }
            int rc = drain_pending(c);
            if (!rc && i + 1 < nchunk) rc = upload(i + 1, b ^ 1);
            else if (!rc && nb_bits && nb_bits != bits) {              // last chunk: the first chunk of the announced next batch, into the set just freed
                const int64_t n0 = std::min<int64_t>((int64_t)1 << 17, nb_V);
                if (n0 * nb_rb <= c->hp_cap && n0 * nb_rb <= c->hb_cap_bits) {
                    parallel_copy(c->hp_bits[b ^ 1], nb_bits, (size_t)(n0 * nb_rb));
                    if (hipMemcpyAsync(c->hb_bits[b ^ 1], c->hp_bits[b ^ 1], n0 * nb_rb, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
                        hipEventRecord(c->ev_h2d[b ^ 1], c->copy_stream) != hipSuccess) { g_err = "prefetch upload failed"; rc = SH_EHIP; }
                    else { c->pre_bits = nb_bits; c->pre_row_bytes = nb_rb; c->pre_n = n0; c->pre_set = b ^ 1; }
                }
            }
            if (rc) side_err = g_err;                                  // (the message is thread-local)
            return rc;
        });
        int rc = inner_dev(c->hb_bits[b], n, c->hb_out[b], c->hb_flags[b]);
        if (!rc && q > 0) {
            hipLaunchKernelGGL(k_betas_rows, dim3((unsigned)((n * q + 255) / 256)), dim3(256), 0, c->stream, c->hb_out[b] + (size_t)5 * n,
                               c->hb_out[b] + (size_t)(5 + q) * n, n, q);
            if (hipGetLastError() != hipSuccess) rc = fail(SH_EHIP, "k_betas_rows launch failed");
        }
        if (!rc && hipEventRecord(c->ev_done[b], c->stream) != hipSuccess) rc = fail(SH_EHIP, "hipEventRecord failed");
        const double t2 = now();
        const int rc_side = side.get();
        if (rc) { c->pend.valid = false; return rc; }
        if (rc_side) { c->pend.valid = false; return fail(rc_side, side_err); }
        c->pend.valid = true; c->pend.b = b; c->pend.q = q; c->pend.s = s; c->pend.n = n; c->pend.betas = betas; c->pend.flags = flags;
        for (int a = 0; a < 5; ++a) c->pend.outs[a] = outs[a];
        if (dbg) fprintf(stderr, "[host_batch] chunk %lld n=%lld: first upload %.2f ms, kernels queued (fixed effects: run) %.2f ms, wait for the copy thread %.2f ms, t=%.2f\n", (long long)i, (long long)n, t1 - t0, t2 - t1, now() - t2, now() - t00);
    }
    const int rc_end = async ? SH_OK : drain_pending(c);
    if (dbg) fprintf(stderr, "[host_batch] call of %lld variants: %.2f ms in all (%.2f ms before the first chunk)\n", (long long)V, now() - t_entry, t00 - t_entry);
    return rc_end;
}

static int lmm_batch_dev_inner(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags);
static int glm_batch_dev_inner(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags);

template <typename F>
static int dedup_wrap(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags, int nrow, F inner)
{
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t Vpad = (V + 511) / 512 * 512;
    int rc = ensure_ws(c, Vpad); if (rc) return rc;
    if (V > c->dd_capV) {
        hipFree(c->dd_h); hipFree(c->dd_keys); hipFree(c->dd_idx); hipFree(c->dd_rep); hipFree(c->dd_slot); hipFree(c->dd_n);
        c->dd_h = nullptr; c->dd_keys = nullptr; c->dd_idx = c->dd_rep = c->dd_slot = c->dd_n = nullptr;
        uint64_t cap = 1024; while (cap < (uint64_t)V * 2) cap <<= 1;
        HIPCHK(dmalloc(&c->dd_h, V)); HIPCHK(dmalloc(&c->dd_keys, cap)); HIPCHK(dmalloc(&c->dd_idx, cap));
        HIPCHK(dmalloc(&c->dd_rep, V)); HIPCHK(dmalloc(&c->dd_slot, V)); HIPCHK(dmalloc(&c->dd_n, 1));
        c->dd_cap = (int64_t)cap; c->dd_capV = V;
    }
    if (V * row_bytes > c->dd_cap_bits) { hipFree(c->dd_bits); c->dd_bits = nullptr; HIPCHK(hipMalloc((void **)&c->dd_bits, V * row_bytes)); c->dd_cap_bits = V * row_bytes; }
    if (V * nrow > c->dd_cap_out) { hipFree(c->dd_out); c->dd_out = nullptr; HIPCHK(dmalloc(&c->dd_out, V * nrow)); c->dd_cap_out = V * nrow; }
    if (V > c->dd_cap_flags) { hipFree(c->dd_flags); c->dd_flags = nullptr; HIPCHK(dmalloc(&c->dd_flags, V)); c->dd_cap_flags = V; }
    HIPCHK(shk_repack_bits(st, (const uint8_t *)d_bits, row_bytes, V, Vpad, c->N, c->NB64p, c->d_T, nullptr));
    HIPCHK(shk_dd_find(st, c->d_T, Vpad, V, c->NB64, c->dd_h, (uint64_t)c->dd_cap, c->dd_keys, c->dd_idx, c->dd_rep, c->dd_slot, c->dd_n));
    int nu = 0;
    { const int rcb = readback_sync(c, st, c->dd_n, nullptr, nullptr, nullptr, &nu); if (rcb) return rcb; }   // the number of distinct patterns sizes the launches below
    c->dd_last_unique = nu;
    HIPCHK(shk_dd_gather(st, (const uint8_t *)d_bits, row_bytes, V, c->dd_rep, c->dd_slot, c->dd_bits));
    rc = inner(c->dd_bits, (int64_t)nu, c->dd_out, c->dd_flags); if (rc) return rc;
    HIPCHK(shk_dd_scatter(st, V, nu, nrow, c->dd_rep, c->dd_slot, c->dd_out, c->dd_flags, (double *)d_out, (uint32_t *)d_flags));
    return SH_OK;
}

// Rows outside the AF window all give the same output, so one of them stands for all: only the kept rows (+ that one) go through
// `inner`.  Same adaptive rule as in lmm_batch_dev_inner: a stream that does not filter is only counted asynchronously.
struct PrefilterRows { const uint64_t *y1 = nullptr, *y0 = nullptr; int n1 = 0, n0 = 0; double pret = 1.0; };   // binary phenotype, --filter-pvalue < 1
template <typename F>
static int af_wrap(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags, int nrow, F inner,
                   PrefilterRows pf = PrefilterRows())
{
    c->af_last_rows = -1;
    const bool use_pf = pf.y1 != nullptr && pf.pret < 1.0;
    if (!((c->af_on || use_pf) && c->af_compact) || (V < 1024 && c->af_compact != 2)) return inner(d_bits, V, d_out, d_flags);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (V > c->af_capV) {
        hipFree(c->af_rep); hipFree(c->af_slot); hipFree(c->af_m); c->af_rep = c->af_slot = c->af_m = nullptr;
        HIPCHK(dmalloc(&c->af_rep, V)); HIPCHK(dmalloc(&c->af_slot, V)); HIPCHK(dmalloc(&c->af_m, 3 * V)); c->af_capV = V;
    }
    if (!c->af_cnt) {
        HIPCHK(dmalloc(&c->af_cnt, 2)); HIPCHK(hipHostMalloc((void **)&c->h_af_cnt, 2 * sizeof(int)));
        HIPCHK(hipEventCreateWithFlags(&c->af_ev, hipEventDisableTiming));
    }
    if (c->af_pending && hipEventQuery(c->af_ev) == hipSuccess) {
        c->af_hint = 1.0 - (double)c->h_af_cnt[0] / (double)c->af_pending_V; c->af_pending = false;
    }
    bool compact = false; int nk = 0, R = 0;
    // a stream that does not filter is sampled: every fourth batch is counted
    // (a wavefront holds 64 variants and a batch is a few rounds of wavefronts, so dropping < 10 % of the rows rarely removes a round)
    if (!c->af_pending && (c->af_hint >= 0.10 || c->af_compact == 2 || (c->af_tick++ & 3) == 0)) {
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->af_cnt, 0, 1, st));
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(c->af_cnt + 1), 2147483647, 1, st));
        if (use_pf)
            HIPCHK(shk_pf_rows(st, (const uint8_t *)d_bits, row_bytes, V, c->N, pf.y1, pf.y0, c->min_af, c->max_af, c->af_on, pf.n1, pf.n0, pf.pret, nrow,
                               c->af_m, c->af_m + c->af_capV, c->af_m + 2 * c->af_capV, c->af_rep, c->af_slot, c->af_cnt, (double *)d_out,
                               (uint32_t *)d_flags));
        else
            HIPCHK(shk_af_rows(st, 0, (const uint8_t *)d_bits, row_bytes, V, c->N, c->min_af, c->max_af, c->af_rep, c->af_slot, c->af_cnt));
        HIPCHK(readback_launch(st, c->af_cnt, c->af_cnt + 1, nullptr, nullptr, c->h_af_cnt));      // (h_af_cnt is pinned: written by the kernel, no copy engine)
        if (c->af_hint >= 0.10 || c->af_compact == 2) {
            HIPCHK(hipStreamSynchronize(st));
            nk = c->h_af_cnt[0]; R = c->h_af_cnt[1];
            c->af_hint = 1.0 - (double)nk / (double)V;
            compact = (int64_t)nk * 100 <= V * 90 || (c->af_compact == 2 && (int64_t)nk * 100 <= V * 97);
        } else {
            HIPCHK(hipEventRecord(c->af_ev, st)); c->af_pending = true; c->af_pending_V = V;
        }
    }
    if (!compact) return inner(d_bits, V, d_out, d_flags);
    const int64_t nu = (int64_t)nk + (R != 2147483647 ? 1 : 0);
    if (nu == 0) { c->af_last_rows = 0; return SH_OK; }            // every row's output was written by the classification kernel
    if (nu * row_bytes > c->af_cap_bits) { hipFree(c->af_bits); c->af_bits = nullptr; HIPCHK(hipMalloc((void **)&c->af_bits, nu * row_bytes)); c->af_cap_bits = nu * row_bytes; }
    if (nu * nrow > c->af_cap_out) { hipFree(c->af_out); c->af_out = nullptr; HIPCHK(dmalloc(&c->af_out, nu * nrow)); c->af_cap_out = nu * nrow; }
    if (nu > c->af_cap_flags) { hipFree(c->af_flags); c->af_flags = nullptr; HIPCHK(dmalloc(&c->af_flags, nu)); c->af_cap_flags = nu; }
    HIPCHK(shk_af_rows(st, 1, nullptr, 0, V, 0, 0, 0, c->af_rep, c->af_slot, c->af_cnt));
    HIPCHK(shk_dd_gather(st, (const uint8_t *)d_bits, row_bytes, V, c->af_rep, c->af_slot, c->af_bits));
    c->af_last_rows = nu;
    int rc = inner(c->af_bits, nu, c->af_out, c->af_flags); if (rc) return rc;
    HIPCHK(shk_dd_scatter(st, V, nu, nrow, c->af_rep, c->af_slot, c->af_out, c->af_flags, (double *)d_out, (uint32_t *)d_flags));
    return SH_OK;
}

extern "C" {

int sh_abi_version(void) { return SH_ABI_VERSION; }
const char *sh_last_error(void) { return g_err.c_str(); }

int sh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

__global__ void k_noop() {}

// How host threads wait for the device (sh_set_wait_mode): 1 = asleep.  The HIP runtime's waits (hipStreamSynchronize, hipEventSynchronize,
// the device-wide wait inside hipHostUnregister) spin by default -- a CPU per waiting stream, whatever the event's flags; with
// hipDeviceScheduleBlockingSync set on the device BEFORE its first use they sleep on an interrupt (tools/ubench/host_feed_probe.hip lines F:
// 0.0705 -> 0.0008 CPU-s for a 70 ms wait).  A job of eight device streams on a 16-CPU quota cannot afford eight spinning threads; a
// benchmark driven through torch (which initialises the device first) keeps the default.
static std::atomic<int> g_wait_mode{0};
static std::atomic<uint64_t> g_wait_applied{0};                     // bit d: the flag was offered to device d
static void apply_wait_mode(int device)
{
    if (g_wait_mode.load() != 1 || device < 0 || device >= 64) return;
    const uint64_t bit = 1ull << device;
    if (g_wait_applied.fetch_or(bit) & bit) return;
    if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();   // (a device somebody initialised already keeps its mode)
}
void sh_set_wait_mode(int sleeping) { g_wait_mode.store(sleeping ? 1 : 0); }

int sh_warmup(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(SH_ENODEV, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    apply_wait_mode(device);
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return SH_OK;
}

sh_ctx *sh_create(int device, int n_samples)
{
    int n = 0;
    if (const std::string bad = sh_route_unknown(); !bad.empty()) { g_err = "SEERHIP_ROUTE: unknown item '" + bad + "' (keys: csrc/route.h)"; return nullptr; }
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_err = "no HIP device: libseerhip has no CPU fallback"; return nullptr; }
    if (device < 0 || device >= n) { g_err = "bad device index"; return nullptr; }
    if (n_samples < 2) { g_err = "n_samples must be >= 2"; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_err = "hipGetDeviceProperties failed"; return nullptr; }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_err = std::string("device is ") + prop.gcnArchName + ", libseerhip is built for gfx950 only"; return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    apply_wait_mode(device);
    sh_ctx *c = new sh_ctx();
    c->device = device; c->N = n_samples;
    if (const char *qv = sh_route("qf")) c->qf_variant = std::atoi(qv);
    if (const char *lv = sh_route("lanes")) c->n_lanes = std::max(1, std::min(8, std::atoi(lv)));
    if (const char *ac = sh_route("afcompact")) c->af_compact = std::atoi(ac);
    c->NT = (n_samples + 255) / 256; c->Np = c->NT * 256;
    c->NB64 = (n_samples + 63) / 64; c->NB64p = c->NT * 4;
    return c;
}

void sh_destroy(sh_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    lanes_destroy(c);
    free_ws(c);
    if (c->h_nkeep) hipHostFree(c->h_nkeep);
    if (c->h_rb) hipHostFree(c->h_rb);
    if (c->h_af_cnt) hipHostFree(c->h_af_cnt);
    if (c->af_ev) hipEventDestroy(c->af_ev);
    hipFree(c->af_rep); hipFree(c->af_slot); hipFree(c->af_m); hipFree(c->af_cnt); hipFree(c->af_bits); hipFree(c->af_out); hipFree(c->af_flags);
    if (c->keep_ev) hipEventDestroy(c->keep_ev);
    hipFree(c->d_vv); hipFree(c->d_mdiag); hipFree(c->d_yc); hipFree(c->d_Qb); hipFree(c->d_y1); hipFree(c->d_y0); hipFree(c->d_tab);
    hipFree(c->d_G); hipFree(c->d_bits); hipFree(c->d_out); hipFree(c->d_flags); hipFree(c->d_rcount); hipFree(c->d_bmax);
    for (int b = 0; b < 2; ++b) { hipFree(c->hb_bits[b]); hipFree(c->hb_out[b]); hipFree(c->hb_flags[b]); if (c->ev_h2d[b]) hipEventDestroy(c->ev_h2d[b]); if (c->ev_done[b]) hipEventDestroy(c->ev_done[b]); }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    for (int b = 0; b < 2; ++b) if (c->hp_bits[b]) hipHostFree(c->hp_bits[b]);
    for (auto &p : c->tev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    hipFree(c->dd_h); hipFree(c->dd_keys); hipFree(c->dd_idx); hipFree(c->dd_rep); hipFree(c->dd_slot); hipFree(c->dd_n); hipFree(c->dd_bits); hipFree(c->dd_out); hipFree(c->dd_flags);
    hipFree(c->sim_K); hipFree(c->sim_S); hipFree(c->sim_keep); hipFree(c->sim_out);
    glm_free(&c->glm);
    delete c;
}

int sh_set_stream(sh_ctx *c, void *s) { if (!c) return fail(SH_EINVAL, "null ctx"); forget_announced(c); c->stream = (hipStream_t)s; return SH_OK; }

int sh_synchronize(sh_ctx *c)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

int sh_set_timing(sh_ctx *c, int on)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    for (auto &p : c->tev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    c->tev.clear(); c->timing = on;
    if (c->lanes) { const int rc = lanes_wait(c); if (rc) return rc; lanes_set_timing(c, on); }
    return SH_OK;
}

int sh_get_timing(sh_ctx *c, double *total_ms, int64_t *launches)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    double tot = 0;
    for (auto &p : c->tev) { HIPCHK(hipEventSynchronize(p.second)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, p.first, p.second)); tot += ms; }
    int64_t nl = (int64_t)c->tev.size();
    // (batches of the lanes: every batch's own span on its lane's stream -- spans of different lanes overlap in time)
    if (c->lanes) { const int rc = lanes_wait(c); if (rc) return rc; const int r2 = lanes_add_timing(c, &tot, &nl); if (r2) return r2; }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = nl;
    return SH_OK;
}

int sh_set_dedup(sh_ctx *c, int on) { if (!c) return fail(SH_EINVAL, "null ctx"); c->dedup = on; return SH_OK; }

int sh_dedup_info(sh_ctx *c, int64_t *unique_last_batch)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    if (unique_last_batch) *unique_last_batch = c->dd_last_unique;
    return SH_OK;
}

int sh_set_af_filter(sh_ctx *c, double min_af, double max_af)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    c->min_af = min_af; c->max_af = max_af; c->af_on = !(min_af <= 0.0 && max_af >= 1.0);
    return SH_OK;
}

// -------------------------------------------------------------------------------------------------------------
// LMM setup: pyseer/lmm.py:26-122 hands over (U, S, y, covariates, h2); everything derived from them that the per-
// variant kernels need is built here (rotate / getUY: lmm_cov.py:165-218; Sd, yKy: lmm_cov.py:665, 734).
// -------------------------------------------------------------------------------------------------------------
// bound = (F2)^(1 / 2^(nsq+1)) * 2^(-sum_k e_k / 2^k): see shk_spectral_bound (lmm_kernels.hip)
static double spectral_bound_value(double f2, const int *e, int nsq)
{
    if (!(f2 > 0.0)) return 0.0;
    double lg = std::log2(f2) / std::ldexp(1.0, nsq + 1);
    for (int k = 0; k < nsq; ++k) lg -= (double)e[k] / std::ldexp(1.0, k);
    return std::exp2(lg);
}

int sh_lmm_setup(sh_ctx *c, const double *U, const double *S, int k, const double *y, const double *C, int D,
                 double h2, int continuous, double pret, double lrtt, int n_limbs)
{
    if (!c || !U || !S || !y || !C) return fail(SH_EINVAL, "null argument");
    if (k < 1 || D < 1) return fail(SH_ESHAPE, "k and D must be >= 1");
    if (h2 < 0.0 || h2 >= 1.0 || std::isnan(h2)) return fail(SH_EH2, "h2 outside [0,1): reference returns no 'beta' (KeyError)");
    { const int rc0 = drain_pending(c); if (rc0) return rc0; }        // a batch still in flight belongs to the model being replaced
    forget_announced(c);
    if (n_limbs == 0) {
        // Automatic limb count: the smallest L in {4, 5} whose TYPICAL a-posteriori bound (an AF-0.5 variant, section 3 of DESIGN.md) is at most a
        // quarter of lmm_tol, so that the extra-limb pass stays the exception; a variant whose own bound exceeds lmm_tol gets the extra limbs
        // whichever L was chosen.  The base-256 digits do not depend on where the main pass stops, only how many of them it contracts does:
        // L = 4 is 20 % less int8 work than L = 5.  SEERHIP_ROUTE lmm_limbs=n forces a count.
        const char *fe = sh_route("lmm_limbs");
        if (fe && std::atoi(fe) > 0) n_limbs = std::atoi(fe);
        else {
            const int rc4 = sh_lmm_setup(c, U, S, k, y, C, D, h2, continuous, pret, lrtt, 4);
            if (rc4 != SH_OK) return rc4;
            const double typ = c->trace_M > 0 ? c->fin.err_norm * (0.5 * c->N) / (0.25 * c->trace_M) : 0.0;
            if (c->E > 0 && typ <= 0.25 * c->lmm_tol) return SH_OK;
            n_limbs = 5;
        }
    }
    if (n_limbs < 3 || n_limbs > 7) return fail(SH_EINVAL, "n_limbs must be 3..7");
    if (c->N > 262143) return fail(SH_EINVAL, "more than 262143 samples: the int32 partial sums of k_lmm_quadform_i8 could overflow");
    HIPCHK(hipSetDevice(c->device));
    const int N = c->N, Np = c->Np;
    const int kp = (k + 3) & ~3;
    c->lmm_ready = false; c->macs_issued = 0;

    // 1. orthonormal basis of the covariate space (modified Gram-Schmidt, twice): P = I - Qb Qb^T = I - X pinv(X)
    std::vector<double> Qb((size_t)N * D, 0.0);
    int r = 0;
    for (int d = 0; d < D; ++d) {
        // intercept (last column of C) first, so that rank 1 == "intercept only"
        const int src = (d == 0) ? D - 1 : d - 1;
        std::vector<double> col(N);
        double n0 = 0;
        for (int i = 0; i < N; ++i) { col[i] = C[(size_t)i * D + src]; n0 += col[i] * col[i]; }
        for (int pass = 0; pass < 2; ++pass)
            for (int e = 0; e < r; ++e) {
                double dot = 0; for (int i = 0; i < N; ++i) dot += Qb[(size_t)i * D + e] * col[i];
                for (int i = 0; i < N; ++i) col[i] -= dot * Qb[(size_t)i * D + e];
            }
        double n1 = 0; for (int i = 0; i < N; ++i) n1 += col[i] * col[i];
        if (!(n1 > 1e-24 * n0) || n0 == 0) continue;        // numerically dependent column (pinv drops it as well)
        const double inv = 1.0 / std::sqrt(n1);
        for (int i = 0; i < N; ++i) Qb[(size_t)i * D + r] = col[i] * inv;
        ++r;
    }
    if (r < 1) return fail(SH_EINVAL, "covariate matrix has rank 0");
    // Is the constant vector in the covariate span?  (pyseer always appends the intercept, lmm.py:95-99; a caller of the C ABI may not.)
    // Only then U~^T 1 = 0 and a row may be stored complemented (k_repack_bits).
    bool ones_in_span;
    {
        std::vector<double> one(N, 1.0);
        for (int e = 0; e < r; ++e) { double dot = 0; for (int i = 0; i < N; ++i) dot += Qb[(size_t)i * D + e] * one[i]; for (int i = 0; i < N; ++i) one[i] -= dot * Qb[(size_t)i * D + e]; }
        double res = 0; for (int i = 0; i < N; ++i) res += one[i] * one[i];
        ones_in_span = res <= 1e-20 * (double)N;
    }
    bool intercept_only = (r == 1);
    if (intercept_only) { const double q0 = Qb[0]; for (int i = 0; i < N; ++i) if (std::fabs(Qb[(size_t)i * D] - q0) > 1e-12 * std::fabs(q0)) intercept_only = false; }
    int DP = 0;
    if (!intercept_only) { DP = 4; while (DP < r) DP *= 2; if (DP > 32) return fail(SH_EINVAL, "more than 32 independent covariates are not supported"); }

    // 2. residualised phenotype (getUY -> rotate(Y)), with rotate's zeroing rule
    std::vector<double> yt(N);
    {
        std::vector<double> cy(r, 0.0);
        for (int e = 0; e < r; ++e) { double s = 0; for (int i = 0; i < N; ++i) s += Qb[(size_t)i * D + e] * y[i]; cy[e] = s; }
        double mean = 0;
        for (int i = 0; i < N; ++i) { double s = y[i]; for (int e = 0; e < r; ++e) s -= Qb[(size_t)i * D + e] * cy[e]; yt[i] = s; mean += s; }
        mean /= N; double q = 0; for (int i = 0; i < N; ++i) q += (yt[i] - mean) * (yt[i] - mean);
        if (std::sqrt(q / N) <= 1e-10) std::fill(yt.begin(), yt.end(), 0.0);
    }
    // 3. U~ = P U  (a no-op for a fresh fit where U is already orthogonal to the covariates; required for --load-lmm with
    //    different covariates, run_test.sh:47), Sd, uy, yKy, v, W
    std::vector<double> Sd(k), uy(k, 0.0), sgn(kp, 0.0);
    for (int rr = 0; rr < k; ++rr) Sd[rr] = h2 * S[rr] + (1.0 - h2);                    // lmm_cov.py:665
    std::vector<double> W((size_t)Np * kp, 0.0);
    {
        std::vector<double> QtU((size_t)r * k, 0.0);
        for (int i = 0; i < N; ++i) {
            const double *u = U + (size_t)i * k;
            for (int e = 0; e < r; ++e) { const double q = Qb[(size_t)i * D + e]; double *o = &QtU[(size_t)e * k]; for (int rr = 0; rr < k; ++rr) o[rr] += q * u[rr]; }
        }
        for (int i = 0; i < N; ++i) {
            const double *u = U + (size_t)i * k; double *w = &W[(size_t)i * kp];
            for (int rr = 0; rr < k; ++rr) w[rr] = u[rr];
            for (int e = 0; e < r; ++e) { const double q = Qb[(size_t)i * D + e]; const double *o = &QtU[(size_t)e * k]; for (int rr = 0; rr < k; ++rr) w[rr] -= q * o[rr]; }
            const double yi = yt[i];
            if (yi != 0.0) for (int rr = 0; rr < k; ++rr) uy[rr] += w[rr] * yi;         // uy = U~^T y
        }
    }
    double yKy = 0;
    std::vector<double> uys(k);
    for (int rr = 0; rr < k; ++rr) { yKy += uy[rr] * uy[rr] / Sd[rr]; uys[rr] = uy[rr] / Sd[rr]; sgn[rr] = Sd[rr] < 0 ? -1.0 : 1.0; }
    std::vector<double> vv(N);
    for (int i = 0; i < N; ++i) {
        double *w = &W[(size_t)i * kp]; double s = 0;
        for (int rr = 0; rr < k; ++rr) s += w[rr] * uys[rr];
        vv[i] = s;                                                                       // v = U~ diag(1/Sd) uy
        for (int rr = 0; rr < k; ++rr) w[rr] *= 1.0 / std::sqrt(std::fabs(Sd[rr]));
    }
    // 4. prefilter constants
    std::vector<uint64_t> y1(c->NB64p, 0), y0(c->NB64p, 0);
    int n1 = 0, n0 = 0; double ymean = 0;
    for (int i = 0; i < N; ++i) { if (y[i] == 1.0) { y1[i >> 6] |= 1ull << (i & 63); ++n1; } else if (y[i] == 0.0) { y0[i >> 6] |= 1ull << (i & 63); ++n0; } ymean += y[i]; }
    ymean /= N;
    std::vector<double> yc(N); double ycs = 0, ycq = 0;
    for (int i = 0; i < N; ++i) { yc[i] = y[i] - ymean; ycs += yc[i]; ycq += yc[i] * yc[i]; }
    std::vector<double> Qbp;
    if (DP) { Qbp.assign((size_t)N * DP, 0.0); for (int i = 0; i < N; ++i) for (int e = 0; e < r; ++e) Qbp[(size_t)i * DP + e] = Qb[(size_t)i * D + e]; }

    // 5. device side: M = W sgn W^T (fp64 MFMA), diagonal, limbs
    hipFree(c->d_vv); hipFree(c->d_mdiag); hipFree(c->d_yc); hipFree(c->d_Qb); hipFree(c->d_y1); hipFree(c->d_y0); hipFree(c->d_G); hipFree(c->d_tab);
    c->d_vv = c->d_mdiag = c->d_yc = c->d_Qb = c->d_tab = nullptr; c->d_y1 = c->d_y0 = nullptr; c->d_G = nullptr;
    const int NT = c->NT, L = n_limbs;
    const int NR = 2 * NT;                                   // 128-sample row tiles
    // E extra limbs below the L of the main pass (at most 7 in all: 0.49 * 256^7 is where fp64 itself ends); contracted only for the
    // variants whose a-posteriori bound exceeds lmm_tol
    int E = std::min(2, 7 - L);
    const int Lt = L + E;
    const size_t gbytes = (size_t)Lt * NR * (NR + 1) * 8192;
    double *d_W = nullptr, *d_sgn = nullptr, *d_M = nullptr; unsigned long long *d_amax = nullptr;
    float *d_Ef = nullptr; double *d_px = nullptr, *d_py = nullptr, *d_nrm = nullptr;
    const int NPOW = 48;
    HIPCHK(dmalloc(&c->d_vv, N)); HIPCHK(dmalloc(&c->d_mdiag, N)); HIPCHK(dmalloc(&c->d_yc, N));
    HIPCHK(dmalloc(&c->d_y1, c->NB64p)); HIPCHK(dmalloc(&c->d_y0, c->NB64p));
    if (DP) HIPCHK(dmalloc(&c->d_Qb, (size_t)N * DP));
    HIPCHK(hipMalloc((void **)&c->d_G, gbytes + (256u << 10)));       // + 256 KB: k_lmm_quadform_i8w's DMA cursor runs four stages past the last limb
    HIPCHK(dmalloc(&d_W, (size_t)Np * kp)); HIPCHK(dmalloc(&d_sgn, kp)); HIPCHK(dmalloc(&d_M, (size_t)Np * Np)); HIPCHK(dmalloc(&d_amax, 1));
    hipStream_t st = c->stream;
    HIPCHK(hipMemcpyAsync(d_W, W.data(), sizeof(double) * (size_t)Np * kp, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sgn, sgn.data(), sizeof(double) * kp, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_vv, vv.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_yc, yc.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_y1, y1.data(), sizeof(uint64_t) * c->NB64p, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_y0, y0.data(), sizeof(uint64_t) * c->NB64p, hipMemcpyHostToDevice, st));
    if (DP) HIPCHK(hipMemcpyAsync(c->d_Qb, Qbp.data(), sizeof(double) * (size_t)N * DP, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_M, 0, sizeof(double) * (size_t)Np * Np, st));
    HIPCHK(dmalloc(&d_Ef, (size_t)Np * Np)); HIPCHK(dmalloc(&d_px, Np)); HIPCHK(dmalloc(&d_py, Np)); HIPCHK(dmalloc(&d_nrm, NPOW));
    HIPCHK(shk_lmm_build_G(st, d_W, d_sgn, N, Np, kp, NR, Lt, E, d_M, c->d_mdiag, d_amax, c->d_G, d_Ef, d_px, d_py, d_nrm, NPOW));
    // certified bound on the spectral norm of the quantisation error (shk_spectral_bound: trace of its 2^(nsq+1)-th power by repeated squaring on
    // the fp64 matrix pipe; d_M is free again once its diagonal and the limbs are taken)
    const int nsq = Np <= 16384 ? 5 : Np <= 32768 ? 4 : 3;
    double *d_X = nullptr, *d_work = nullptr; int *d_exps = nullptr;
    HIPCHK(dmalloc(&d_X, (size_t)Np * Np)); HIPCHK(dmalloc(&d_work, 2)); HIPCHK(dmalloc(&d_exps, 8));
    HIPCHK(shk_spectral_bound(st, d_Ef, N, Np, d_X, d_M, nsq, d_work, d_exps));
    double cert_f2 = 0.0; int cert_e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(&cert_f2, d_work + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(cert_e, d_exps, sizeof(cert_e), hipMemcpyDeviceToHost, st));
    c->tab_doubles = (size_t)c->NB64 * 256 * (2 + 2 + 8); c->g_bytes = gbytes;
    HIPCHK(dmalloc(&c->d_tab, c->tab_doubles));                                // up to 12 doubles per nibble entry
    HIPCHK(shk_lmm_build_tab(st, c->d_vv, c->d_mdiag, c->d_yc, c->d_Qb, DP, continuous, N, c->NB64, c->d_tab));
    unsigned long long amax_bits = 0;
    HIPCHK(hipMemcpyAsync(&amax_bits, d_amax, sizeof(amax_bits), hipMemcpyDeviceToHost, st));
    std::vector<double> nrm(NPOW, 0.0), mdiag_h(N, 0.0);
    HIPCHK(hipMemcpyAsync(nrm.data(), d_nrm, sizeof(double) * NPOW, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(mdiag_h.data(), c->d_mdiag, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    hipFree(d_W); hipFree(d_sgn); hipFree(d_M); hipFree(d_amax); hipFree(d_Ef); hipFree(d_px); hipFree(d_py); hipFree(d_nrm);
    hipFree(d_X); hipFree(d_work); hipFree(d_exps);
    double amax; std::memcpy(&amax, &amax_bits, sizeof(double));
    const double p256 = std::pow(256.0, L);
    c->quant_scale = amax > 0 ? 0.49 * p256 / amax : 0.0;                 // of the main pass (its top L limbs)
    // |x^T (G - Gq/s) x| <= ||Esym||_2 |x|^2 = err_norm * (carriers of the stored row).  err_norm is a CERTIFIED upper bound since round 3:
    // ||E||_2 <= trace(E^(2p))^(1/(2p)), 2p = 2^(nsq+1) (64 at N <= 16384: within 4.4 % of the norm for this matrix' Wigner-like spectrum),
    // inflated by 1e-6 for the fp64 rounding of the squarings.  The power iteration of rounds 1-2 (48 steps, which approach the norm from below)
    // is kept as the reported estimate and as a cross-check: a bound below it would be a bug.
    c->err_norm_est_ulp = nrm[NPOW - 1];
    c->err_norm_squarings = nsq;
    c->err_norm_ulp = spectral_bound_value(cert_f2, cert_e, nsq) * (1.0 + 1e-6);
    if (!(c->err_norm_ulp >= c->err_norm_est_ulp * (1.0 - 1e-9)) || !std::isfinite(c->err_norm_ulp))
        return fail(SH_EHIP, "spectral-norm certificate below the power-iteration estimate");
    double sumv = 0; for (int i = 0; i < N; ++i) sumv += vv[i];
    c->trace_M = 0; for (int i = 0; i < N; ++i) c->trace_M += mdiag_h[i];

    c->k = k; c->D = D; c->L = L; c->DP = DP; c->E = E;
    c->complement = ones_in_span && !(sh_route("complement") && std::atoi(sh_route("complement")) == 0);
    if (const char *tv = sh_route("lmm_tol")) c->lmm_tol = std::atof(tv);
    LmmFinParams &P = c->fin;
    P.N = N; P.D = D; P.continuous = continuous; P.n1 = n1; P.n0 = n0; P.yc_sum = ycs; P.yc_sq = ycq;
    P.yKy = yKy; P.inv_scale = amax > 0 ? 1.0 / c->quant_scale : 0.0; P.pret = pret; P.lrtt = lrtt;
    P.sumv = sumv; P.err_norm = c->err_norm_ulp * P.inv_scale; P.tol = E > 0 ? c->lmm_tol : 0.0;
    P.inv_scale_low = P.inv_scale / std::pow(256.0, E);
    c->lmm_ready = true;
    return SH_OK;
}

int sh_lmm_info(sh_ctx *c, int *n_limbs, int64_t *macs, double *qscale)
{
    if (!c || !c->lmm_ready) return fail(SH_EINVAL, "sh_lmm_setup has not run");
    if (n_limbs) *n_limbs = c->L;
    // executed int8 MACs per variant: as issued by the last main-pass launch (k_lmm_quadform_i8w drops the 32-row sub-tiles of the last row tile
    // that are padding), else the full tiles of k_lmm_quadform_i8: L * sum_I 2(I+1) tiles * (128 rows * 64), NR = 2*NT row tiles
    if (macs) *macs = c->macs_issued > 0 ? c->macs_issued : (int64_t)c->L * (2 * c->NT) * (2 * c->NT + 1) * 128 * 64;
    if (qscale) *qscale = c->quant_scale;
    return SH_OK;
}

// The per-run LMM state of `src` (limbs of G, v, diag M, nibble tables, covariate basis, phenotype masks and every scalar) copied device
// to device into `dst`: one sh_lmm_setup per run instead of one per GPU ("optional set-up time broadcast").  94 MB at
// N = 5000 over xGMI or within one device; no collective, nothing on the data path.
int sh_lmm_share(sh_ctx *dst, sh_ctx *src)
{
    if (!dst || !src || !src->lmm_ready) return fail(SH_EINVAL, "sh_lmm_share: the source context has no LMM set-up");
    if (dst == src) return SH_OK;
    if (dst->N != src->N) return fail(SH_ESHAPE, "sh_lmm_share: contexts of different sample counts");
    HIPCHK(hipSetDevice(src->device));
    HIPCHK(hipStreamSynchronize(src->stream));
    HIPCHK(hipSetDevice(dst->device));
    sh_ctx *c = dst;
    const int N = c->N;
    hipFree(c->d_vv); hipFree(c->d_mdiag); hipFree(c->d_yc); hipFree(c->d_Qb); hipFree(c->d_y1); hipFree(c->d_y0); hipFree(c->d_G); hipFree(c->d_tab);
    c->d_vv = c->d_mdiag = c->d_yc = c->d_Qb = c->d_tab = nullptr; c->d_y1 = c->d_y0 = nullptr; c->d_G = nullptr; c->lmm_ready = false;
    HIPCHK(dmalloc(&c->d_vv, N)); HIPCHK(dmalloc(&c->d_mdiag, N)); HIPCHK(dmalloc(&c->d_yc, N));
    HIPCHK(dmalloc(&c->d_y1, c->NB64p)); HIPCHK(dmalloc(&c->d_y0, c->NB64p));
    if (src->DP) HIPCHK(dmalloc(&c->d_Qb, (size_t)N * src->DP));
    HIPCHK(hipMalloc((void **)&c->d_G, src->g_bytes + (256u << 10))); HIPCHK(dmalloc(&c->d_tab, src->tab_doubles));
    auto cp = [&](void *d, const void *s_, size_t n) { return hipMemcpyPeerAsync(d, dst->device, s_, src->device, n, dst->stream); };
    HIPCHK(cp(c->d_vv, src->d_vv, sizeof(double) * N)); HIPCHK(cp(c->d_mdiag, src->d_mdiag, sizeof(double) * N));
    HIPCHK(cp(c->d_yc, src->d_yc, sizeof(double) * N));
    HIPCHK(cp(c->d_y1, src->d_y1, sizeof(uint64_t) * c->NB64p)); HIPCHK(cp(c->d_y0, src->d_y0, sizeof(uint64_t) * c->NB64p));
    if (src->DP) HIPCHK(cp(c->d_Qb, src->d_Qb, sizeof(double) * (size_t)N * src->DP));
    HIPCHK(cp(c->d_G, src->d_G, src->g_bytes)); HIPCHK(cp(c->d_tab, src->d_tab, sizeof(double) * src->tab_doubles));
    HIPCHK(hipStreamSynchronize(dst->stream));
    c->k = src->k; c->D = src->D; c->L = src->L; c->DP = src->DP; c->E = src->E; c->complement = src->complement;
    c->quant_scale = src->quant_scale; c->err_norm_ulp = src->err_norm_ulp; c->err_norm_est_ulp = src->err_norm_est_ulp; c->err_norm_squarings = src->err_norm_squarings; c->trace_M = src->trace_M; c->lmm_tol = src->lmm_tol;
    c->g_bytes = src->g_bytes; c->tab_doubles = src->tab_doubles; c->fin = src->fin;
    c->lmm_ready = true;
    return SH_OK;
}

int sh_set_lmm_tol(sh_ctx *c, double tol)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    if (!(tol >= 0.0)) return fail(SH_EINVAL, "tol must be >= 0 (0 = never refine)");
    c->lmm_tol = tol;
    if (c->lmm_ready) c->fin.tol = c->E > 0 ? tol : 0.0;
    return SH_OK;
}

int sh_lmm_bound(sh_ctx *c, double *err_norm_ulp, double *ulp, double *tol, int *extra_limbs, double *bound_typical,
                 double *bound_max_last, int64_t *refined_last)
{
    if (!c || !c->lmm_ready) return fail(SH_EINVAL, "sh_lmm_setup has not run");
    HIPCHK(hipSetDevice(c->device));
    if (err_norm_ulp) *err_norm_ulp = c->err_norm_ulp;
    if (ulp) *ulp = c->fin.inv_scale;
    if (tol) *tol = c->fin.tol;
    if (extra_limbs) *extra_limbs = c->E;
    // a variant carried by half of the samples, independent of the population structure: x^T M x ~ (N/4) * trace(M)/N
    if (bound_typical) *bound_typical = c->trace_M > 0 ? c->fin.err_norm * (0.5 * c->N) / (0.25 * c->trace_M) : 0.0;
    unsigned long long bm = 0; int nr = 0;
    if (c->d_bmax) {
        HIPCHK(hipMemcpyAsync(&bm, c->d_bmax, sizeof(bm), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&nr, c->d_rcount, sizeof(nr), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    double b; std::memcpy(&b, &bm, sizeof(double));
    if (bound_max_last) *bound_max_last = b;
    if (refined_last) *refined_last = nr;
    return SH_OK;
}

int sh_lmm_bound_estimate(sh_ctx *c, double *power_iteration_ulp, int *squarings)
{
    if (!c || !c->lmm_ready) return fail(SH_EINVAL, "sh_lmm_setup has not run");
    if (power_iteration_ulp) *power_iteration_ulp = c->err_norm_est_ulp;
    if (squarings) *squarings = c->err_norm_squarings;
    return SH_OK;
}

int sh_spectral_bound_f32(sh_ctx *c, const float *A, int n, int squarings, double *upper, double *power_iteration)
{
    if (!c || !A || n < 1 || squarings < 1 || squarings > 6) return fail(SH_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    const int Np = (n + 127) / 128 * 128, NPOW = 48;
    float *d_A = nullptr; double *d_X = nullptr, *d_Y = nullptr, *d_work = nullptr, *d_px = nullptr, *d_py = nullptr, *d_nrm = nullptr; int *d_exps = nullptr;
    HIPCHK(dmalloc(&d_A, (size_t)Np * Np)); HIPCHK(dmalloc(&d_X, (size_t)Np * Np)); HIPCHK(dmalloc(&d_Y, (size_t)Np * Np));
    HIPCHK(dmalloc(&d_work, 2)); HIPCHK(dmalloc(&d_exps, 8)); HIPCHK(dmalloc(&d_px, Np)); HIPCHK(dmalloc(&d_py, Np)); HIPCHK(dmalloc(&d_nrm, NPOW));
    hipStream_t st = c->stream;
    HIPCHK(hipMemsetAsync(d_A, 0, sizeof(float) * (size_t)Np * Np, st));
    HIPCHK(hipMemcpy2DAsync(d_A, sizeof(float) * Np, A, sizeof(float) * n, sizeof(float) * n, n, hipMemcpyHostToDevice, st));
    HIPCHK(shk_spectral_bound(st, d_A, n, Np, d_X, d_Y, squarings, d_work, d_exps));
    HIPCHK(shk_power_norm(st, d_A, n, Np, d_px, d_py, d_nrm, NPOW));
    double f2 = 0.0, nrm[48]; int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(&f2, d_work + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(e, d_exps, sizeof(e), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(nrm, d_nrm, sizeof(nrm), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    hipFree(d_A); hipFree(d_X); hipFree(d_Y); hipFree(d_work); hipFree(d_exps); hipFree(d_px); hipFree(d_py); hipFree(d_nrm);
    if (upper) *upper = spectral_bound_value(f2, e, squarings) * (1.0 + 1e-6);
    if (power_iteration) *power_iteration = nrm[NPOW - 1];
    return SH_OK;
}

int sh_lmm_batch_dev(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags)
{
    if (!c || !c->lmm_ready) return fail(SH_EINVAL, "sh_lmm_setup has not run");
    if (V <= 0) return SH_OK;
    if (row_bytes * 8 < c->N) return fail(SH_ESHAPE, "row_bytes*8 < n_samples: shape mismatch between snps and Y");
    if (c->dedup)
        return dedup_wrap(c, d_bits, row_bytes, V, d_out, d_flags, 5,
                          [&](const void *b, int64_t n, void *o, void *f) { return lmm_batch_dev_inner(c, b, row_bytes, n, o, f); });
    return lmm_batch_dev_inner(c, d_bits, row_bytes, V, d_out, d_flags);
}

static int lmm_batch_dev_inner(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags)
{
    if (V <= 0) return SH_OK;
    if (row_bytes * 8 < c->N) return fail(SH_ESHAPE, "row_bytes*8 < n_samples: shape mismatch between snps and Y");
    HIPCHK(hipSetDevice(c->device));
    const int64_t Vpad = (V + 511) / 512 * 512;
    int rc = ensure_ws(c, Vpad); if (rc) return rc;
    hipStream_t st = c->stream;
    if (Vpad > c->cap_ref || (c->E > 0 && !c->d_T3)) {                // (a context set up again with extra limbs after batches without them has no T3 / q3 yet)
        hipFree(c->d_flip); hipFree(c->d_T3); hipFree(c->d_q3); hipFree(c->d_rlist);
        c->d_flip = nullptr; c->d_T3 = nullptr; c->d_q3 = nullptr; c->d_rlist = nullptr;
        HIPCHK(dmalloc(&c->d_flip, Vpad)); HIPCHK(dmalloc(&c->d_rlist, Vpad));
        if (c->E > 0) { HIPCHK(dmalloc(&c->d_T3, (size_t)Vpad * c->NB64p)); HIPCHK(dmalloc(&c->d_q3, Vpad * 2)); }
        c->cap_ref = Vpad;
    }
    if (!c->d_rcount) { HIPCHK(dmalloc(&c->d_rcount, 1)); HIPCHK(dmalloc(&c->d_bmax, 1)); }
    LmmLinOut lo{c->d_t11, c->d_t01, c->d_m, c->d_xky, c->d_dg, c->d_rss, c->d_s1, c->d_q1, c->complement ? c->d_flip : nullptr};
    HIPCHK(shk_repack_bits(st, (const uint8_t *)d_bits, row_bytes, V, Vpad, c->N, c->NB64p, c->d_T, c->complement ? c->d_flip : nullptr));
    HIPCHK(shk_lmm_linear(st, c->DP, c->d_T, Vpad, c->N, c->NB64, c->d_vv, c->d_mdiag, c->d_yc, c->d_Qb, c->d_y1, c->d_y0,
                          c->fin.continuous, c->lin_tab ? c->d_tab : nullptr, lo));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    const int lsplit = c->qf_split ? c->L : 1;
    const int8_t *Gmain = c->d_G + (size_t)c->E * (2 * c->NT) * (2 * c->NT + 1) * 8192;                   // the main pass contracts the top L limbs
    // AF-filtered variants never read their quadratic form: contract only the kept columns when >= 3 % of the batch is filtered.
    // Knowing the count costs a host round trip, which would serialise back-to-back batches; so a stream whose last counted batch
    // had < 3 % filtered is only counted asynchronously (pinned counter + event, looked at when the next batch arrives) and runs the
    // plain path, and a stream that does filter pays the round trip and saves the work.  Results are identical either way.
    int64_t nk = V;
    bool compact = false;
    LmmFinParams KP = c->fin; KP.min_af = c->min_af; KP.max_af = c->max_af; KP.af_on = c->af_on;
    if ((c->af_on || KP.pret < 1.0) && c->af_compact) {
        if (Vpad > c->cap_keep) {
            hipFree(c->d_T2); hipFree(c->d_q2); hipFree(c->d_keep); hipFree(c->d_nkeep);
            c->d_T2 = nullptr; c->d_q2 = nullptr; c->d_keep = c->d_nkeep = nullptr;
            HIPCHK(dmalloc(&c->d_T2, (size_t)Vpad * c->NB64p)); HIPCHK(dmalloc(&c->d_q2, Vpad * 8)); HIPCHK(dmalloc(&c->d_keep, Vpad));
            HIPCHK(dmalloc(&c->d_nkeep, 1)); c->cap_keep = Vpad;
        }
        if (!c->h_nkeep) { HIPCHK(hipHostMalloc((void **)&c->h_nkeep, sizeof(int))); HIPCHK(hipEventCreateWithFlags(&c->keep_ev, hipEventDisableTiming)); }
        if (c->keep_pending && hipEventQuery(c->keep_ev) == hipSuccess) {
            c->filtered_hint = 1.0 - (double)*c->h_nkeep / (double)c->keep_V; c->keep_pending = false;
        }
        if (!c->keep_pending) {
            HIPCHK(hipMemsetAsync(c->d_nkeep, 0, sizeof(int), st));
            HIPCHK(shk_af_compact(st, 0, V, lo, KP, c->d_keep, c->d_nkeep, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr));
            HIPCHK(readback_launch(st, c->d_nkeep, nullptr, nullptr, nullptr, c->h_nkeep));
            if (c->filtered_hint >= 0.03 || c->af_compact == 2) {                 // SEERHIP_ROUTE afcompact=2: always count (tests)
                HIPCHK(hipStreamSynchronize(st));
                nk = *c->h_nkeep; c->filtered_hint = 1.0 - (double)nk / (double)V;
                compact = nk * 100 <= V * 97;
            } else {
                HIPCHK(hipEventRecord(c->keep_ev, st)); c->keep_pending = true; c->keep_V = V;
            }
        }
    }
    c->last_kept = compact ? nk : -1;
    if (compact && nk > 0) {
        const int64_t Vpad2 = (nk + 511) / 512 * 512;
        HIPCHK(shk_af_compact(st, 1, V, lo, KP, c->d_keep, nullptr, c->d_T, Vpad, c->d_T2, Vpad2, c->NB64p, (int)nk, nullptr, nullptr));
        HIPCHK(shk_lmm_quadform(st, c->qf_variant, Gmain, c->d_T2, Vpad2, 2 * c->NT, c->L, lsplit, c->d_q2, nullptr, c->N - (2 * c->NT - 1) * 128, &c->macs_issued));
        HIPCHK(shk_af_compact(st, 2, V, lo, KP, c->d_keep, nullptr, nullptr, Vpad, nullptr, Vpad2, lsplit, (int)nk, c->d_q2, c->d_q));
    } else if (!compact) {
        HIPCHK(shk_lmm_quadform(st, c->qf_variant, Gmain, c->d_T, Vpad, 2 * c->NT, c->L, lsplit, c->d_q, nullptr, c->N - (2 * c->NT - 1) * 128, &c->macs_issued));
    }
    if (c->timing) { HIPCHK(hipEventRecord(e1, st)); c->tev.emplace_back(e0, e1); }
    LmmFinParams P = c->fin; P.min_af = c->min_af; P.max_af = c->max_af; P.af_on = c->af_on;
    const bool refine = c->E > 0 && P.tol > 0.0;
    LmmRefine R{refine ? c->d_rlist : nullptr, c->d_rcount, c->d_bmax};
    HIPCHK(hipMemsetAsync(c->d_rcount, 0, sizeof(int), st));
    HIPCHK(hipMemsetAsync(c->d_bmax, 0, sizeof(unsigned long long), st));
    HIPCHK(shk_lmm_finalize(st, V, Vpad, lsplit, lo, c->d_q, P, (double *)d_out, (uint32_t *)d_flags, R));
    if (refine)
        HIPCHK(shk_lmm_refine(st, V, Vpad, lsplit, c->E, 2 * c->NT, c->NB64p, c->d_G, c->d_T, c->d_T3, c->d_q3, lo, c->d_q, P, (double *)d_out,
                              (uint32_t *)d_flags, R));
    return SH_OK;
}

static int lmm_batch_host(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V, double *prep, double *pvalue,
                          double *beta, double *bse, double *frac_h2, uint32_t *flags, bool async)
{
    if (!c || !c->lmm_ready) return fail(SH_EINVAL, "sh_lmm_setup has not run");
    if (!bits || !prep || !pvalue || !beta || !bse || !frac_h2 || !flags) return fail(SH_EINVAL, "null argument");
    if (V <= 0) { HIPCHK(hipSetDevice(c->device)); return drain_pending(c); }      // (an empty batch still completes the one before it)
    double *outs[5] = {prep, pvalue, beta, bse, frac_h2};
    return host_batch(c, bits, row_bytes, V, 5, outs, nullptr, 0, flags, async,
                      [&](const void *b, int64_t n, void *o, void *f) { return sh_lmm_batch_dev(c, b, row_bytes, n, o, f); });
}

int sh_lmm_batch(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V, double *prep, double *pvalue,
                 double *beta, double *bse, double *frac_h2, uint32_t *flags)
{
    return lmm_batch_host(c, bits, row_bytes, V, prep, pvalue, beta, bse, frac_h2, flags, false);
}

int sh_lmm_batch_async(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V, double *prep, double *pvalue,
                       double *beta, double *bse, double *frac_h2, uint32_t *flags)
{
    return lmm_batch_host(c, bits, row_bytes, V, prep, pvalue, beta, bse, frac_h2, flags, true);
}

int sh_prefetch_rows(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    if (!bits || V <= 0 || row_bytes <= 0) { c->next_bits = nullptr; return SH_OK; }
    c->next_bits = bits; c->next_row_bytes = row_bytes; c->next_V = V;
    return SH_OK;
}

int sh_wait(sh_ctx *c)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    const int rc = drain_pending(c);
    const int rl = lanes_wait(c);                                      // (batches handed to the lanes: sh_glm_batch_dev_async)
    return rc ? rc : rl;
}

// -------------------------------------------------------------------------------------------------------------
// Similarity (kinship) matrix: pyseer/similarity.py:99-113  K = G G^T over the AF-kept variants.
// -------------------------------------------------------------------------------------------------------------
int sh_sim_begin(sh_ctx *c)
{
    if (!c) return fail(SH_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    c->NS = (c->N + 127) / 128 * 128;
    if (!c->sim_K) HIPCHK(dmalloc(&c->sim_K, (size_t)c->NS * c->NS));
    HIPCHK(hipMemsetAsync(c->sim_K, 0, sizeof(unsigned long long) * (size_t)c->NS * c->NS, c->stream));
    return SH_OK;
}

int sh_sim_accumulate_dev(sh_ctx *c, const void *d_bits, int64_t row_bytes, int64_t V)
{
    if (!c || !c->sim_K) return fail(SH_EINVAL, "sh_sim_begin has not run");
    if (!d_bits) return fail(SH_EINVAL, "null argument");
    if (row_bytes * 8 < c->N) return fail(SH_ESHAPE, "row_bytes too small for n_samples");
    if (V <= 0) return SH_OK;
    if (V > ((int64_t)1 << 31)) return fail(SH_EINVAL, "at most 2^31 variants per call (32-bit partial counts)");
    HIPCHK(hipSetDevice(c->device));
    const int64_t Vpad = (V + 511) / 512 * 512;
    int rc = ensure_ws(c, Vpad); if (rc) return rc;
    if (Vpad > c->sim_capV) {
        hipFree(c->sim_S); hipFree(c->sim_keep); c->sim_S = nullptr; c->sim_keep = nullptr;
        const int64_t VWp = (Vpad / 64 + 15) / 16 * 16;
        HIPCHK(dmalloc(&c->sim_S, (size_t)VWp * c->NS)); HIPCHK(dmalloc(&c->sim_keep, (size_t)VWp));
        c->sim_capV = Vpad;
    }
    HIPCHK(shk_repack_bits(c->stream, (const uint8_t *)d_bits, row_bytes, V, Vpad, c->N, c->NB64p, c->d_T, nullptr));
    HIPCHK(shk_sim_accumulate(c->stream, c->d_T, Vpad, V, c->N, c->NB64, c->min_af, c->max_af, c->af_on, c->sim_keep, c->sim_S, c->NS, c->sim_K));
    return SH_OK;
}

int sh_sim_accumulate(sh_ctx *c, const uint8_t *bits, int64_t row_bytes, int64_t V)
{
    if (!c || !c->sim_K) return fail(SH_EINVAL, "sh_sim_begin has not run");
    if (!bits) return fail(SH_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const int64_t CH = 1 << 20;
    for (int64_t s = 0; s < V; s += CH) {
        const int64_t n = std::min(CH, V - s);
        int rc = ensure_staging(c, n * row_bytes, 0, 0); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->d_bits, bits + s * row_bytes, n * row_bytes, hipMemcpyHostToDevice, c->stream));
        rc = sh_sim_accumulate_dev(c, c->d_bits, row_bytes, n); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SH_OK;
}

int sh_sim_finish(sh_ctx *c, double *K)
{
    if (!c || !c->sim_K) return fail(SH_EINVAL, "sh_sim_begin has not run");
    if (!K) return fail(SH_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const size_t n = (size_t)c->N * c->N;
    if (!c->sim_out) HIPCHK(dmalloc(&c->sim_out, n));
    HIPCHK(shk_sim_finish(c->stream, c->sim_K, c->NS, c->N, c->sim_out));
    HIPCHK(hipMemcpyAsync(K, c->sim_out, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

#include "glm_api_impl.inc"
#include "lanes_api.inc"
#include "job_api.inc"
#include "job_run.inc"

}  // extern "C"
