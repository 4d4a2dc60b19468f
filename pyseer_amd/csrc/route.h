// Test hook: ONE environment variable that forces the routes a run would otherwise choose from its data, so that the tests can put every
// fallback kernel under the same inputs (tests/test_glm_gpu.py::test_logistic_fit_paths_agree and friends).  Not a tuning interface.
//   SEERHIP_ROUTE="key=value,key=value,..."      unknown keys are an error (sh_create / sh_reader_open refuse)
// and ONE for diagnostics on stderr:
//   SEERHIP_DEBUG="firth,host,cli,glm"            (any subset; sh_debug("firth") ...)
// Round 5 folded the remaining single-purpose variables into the two (SEERHIP_FIRTH_LITERAL / _STRICT / _FAST, SEERHIP_LMM_LIMBS / _TOL,
// SEERHIP_QF, SEERHIP_READER / _THREADS, SEERHIP_WAIT / _JOB / _DMA -> route keys; SEERHIP_FIRTH_DEBUG / _HOST_DEBUG / _CLI_TIMING ->
// SEERHIP_DEBUG items).  A context reads its routes at set-up; the lanes of a context (lanes_api.inc) are set up under the route string
// their parent saw (sh_route_override), whatever the environment holds by then.
// Keys (default in brackets):
//   chord=0          [1]  logistic: fp64-score Newton passes instead of the chord rounds (the form a run without covariates takes)
//   chord_n32=K      [5]  single-precision Newton rounds before the chord rounds (1..8); stragglers restart in fp64
//   chord_enter=X    [5e-3] step below which a variant enters the chord rounds
//   bitdot=0         [1]  first Newton step by a sample pass, k_glm_final instead of the finishing kernels;  2: carrier sums by the nibble-table
//                    kernel on the vector ALU (k_glm_bitdot) instead of the int8 matrix cores (bitdot_i8.hip)
//   first_bordered=0 [1]  first Newton step through the general (Q+2)^2 solve
//   pk=0             [1]  the passes without the packed single-precision records
//   warm=0           [1]  no null-model warm start (the reference's start vector)
//   fin_rounds=0     [1]  k_glm_final for every variant
//   ll_first=0       [1]  score pass first, likelihood pass last
//   newton=1         [0]  every variant through the all-fp64 kernel (the reference's trajectory)
//   firth_last=0     [1]  Firth rounds: the last likelihood pass taken instead of carried over the last step
//   firth_first32=n  [2]  one-pass Firth: how many of the first passes run in single precision (0: all fp64; 1: round 5; 2: then ONE fp64 pass finishes)
//   afcompact=M      [1]  0: never compact AF / prefilter-rejected rows before the kernels, 2: always
//   complement=0     [1]  LMM: rows with more than N/2 carriers stored as given
//   reader_slab=B, reader_pad=B   reader: bytes per decoded slab / carried over between slabs
//   reader_depth=D   [10] slabs decoded ahead of the parser;  reader_helpers=H  translating threads beside the producer [reader threads / 2, <= 8];
//                    reader_target=B  text bytes a region is sized for [12e6]
//   firth_literal=1  [0]  Firth step halving by the reference's literal F(new) > F(old) on the rounds' own evaluation order
//   firth_strict=1   [0]  literal rule + the reference's start vector + one log per sample (the A/B closest to the reference's own order)
//   firth_fast=0     [1]  forced Firth at N >= 4096 through the exact two-pass rounds instead of the one-pass kernel (firth_fast.hip)
//   firth_w=0        [1]  the one-pass kernel's single-precision first pass at one wavefront per SIMD (firth_fast.hip) instead of two (firth_fast_w.hip)
//   lmm_limbs=L      [auto] int8 limbs of the LMM contraction;  lmm_tol=X [1e-8] bound above which a variant is contracted with the extra limbs
//   qf=V             [4]  LMM contraction kernel: 0 = k_lmm_quadform_i8 (two wavefronts per SIMD), 3x = timing ablations
//   lanes=n          [3]  lanes of a fixed-effects context (sh_set_lanes overrides)
//   reader=serial|zlib [par] container decoder of the native reader;  reader_threads=T  its parser workers
//   wait=spin        [sleep], job=0 / py [1], dma=0 [1], procs=0 [1]   command line (pyseer_amd/__main__.py, input.py): host threads spin on the device; the
//                    Python block loop instead of the job stream; rows through pinned slabs instead of DMA from the registered cache mapping
//   dma_window=B     [1 GB] bytes of the packed-cache mapping registered for DMA at a time by the library's block loop (job_run.inc; small values:
//                    many windows in a small cache)
//   reader_chunk=B   [4 MB] most compressed bytes per region of the parallel gzip decoder (regions are sized for ~12 MB of text; small values:
//                    many regions in a small file);  reader_workers=W  its decoding threads [2/3 of the reader's threads, at most 32]
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

static inline const char *const *sh_route_keys()
{
    static const char *const keys[] = {"chord", "chord_n32", "chord_enter", "bitdot", "first_bordered", "pk", "warm", "fin_rounds", "ll_first", "newton",
                                       "firth_last", "firth_first32", "afcompact", "complement", "reader_slab", "reader_pad", "reader_chunk", "reader_workers", "reader_depth", "reader_helpers", "reader_target",
                                       "firth_literal", "firth_strict", "firth_fast", "firth_w", "lmm_limbs", "lmm_tol", "qf", "lanes", "reader", "reader_threads", "wait", "job", "dma", "procs", "dma_window", nullptr};
    return keys;
}
// the route string in force for the calling thread: the environment's, or the one a parent context was set up under (lanes_api.inc)
static inline const char *&sh_route_override() { static thread_local const char *p = nullptr; return p; }
static inline const char *sh_route_string() { const char *o = sh_route_override(); return o ? o : std::getenv("SEERHIP_ROUTE"); }
// the value of `key` in SEERHIP_ROUTE, or nullptr (the returned string lives until the calling thread's next sh_route)
static inline const char *sh_route(const char *key)
{
    const char *e = sh_route_string();
    if (!e) return nullptr;
    static thread_local std::string val;
    const size_t kl = std::strlen(key);
    for (const char *p = e; *p;) {
        const char *end = std::strchr(p, ','); if (!end) end = p + std::strlen(p);
        const char *eq = (const char *)std::memchr(p, '=', (size_t)(end - p));
        if (eq && (size_t)(eq - p) == kl && std::memcmp(p, key, kl) == 0) { val.assign(eq + 1, end); return val.c_str(); }
        p = *end ? end + 1 : end;
    }
    return nullptr;
}
// empty when every key of SEERHIP_ROUTE is known, else the offending item
static inline std::string sh_route_unknown()
{
    const char *e = sh_route_string();
    if (!e) return std::string();
    for (const char *p = e; *p;) {
        const char *end = std::strchr(p, ','); if (!end) end = p + std::strlen(p);
        const char *eq = (const char *)std::memchr(p, '=', (size_t)(end - p));
        bool ok = false;
        if (eq) for (const char *const *k = sh_route_keys(); *k; ++k) if (std::strlen(*k) == (size_t)(eq - p) && std::memcmp(*k, p, (size_t)(eq - p)) == 0) ok = true;
        if (!ok && end > p) return std::string(p, end);
        p = *end ? end + 1 : end;
    }
    return std::string();
}

// is `item` in SEERHIP_DEBUG (a comma-separated list)?
static inline bool sh_debug(const char *item)
{
    const char *e = std::getenv("SEERHIP_DEBUG");
    if (!e) return false;
    const size_t kl = std::strlen(item);
    for (const char *p = e; *p;) {
        const char *end = std::strchr(p, ','); if (!end) end = p + std::strlen(p);
        if ((size_t)(end - p) == kl && std::memcmp(p, item, kl) == 0) return true;
        p = *end ? end + 1 : end;
    }
    return false;
}
