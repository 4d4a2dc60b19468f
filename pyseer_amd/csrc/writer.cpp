// writer.cpp -- native result sink: array-backed results -> the reference's TSV rows.
//
// Replaces the per-row Python of pyseer/utils.py:39-105 (format_output) and the print loop of pyseer/__main__.py:805-827:
//   variant \t af \t filter-pvalue \t lrt-pvalue \t beta \t beta-std-err \t (variant_h2 | intercept [\t covariate betas...])
//   [\t lineage] \t notes
// numbers as '%.2E' (utils.py:60-75: '%.2E' % Decimal(x), i.e. the correctly rounded 3-significant-digit form, which glibc's
// printf also produces), non-finite -> empty field; lineage label or NA; notes joined by ',' in the flag-bit order of
// include/seerhip.h.  Rows are formatted in parts by the calling thread and the process-wide host pool (host_pool.h) and concatenated in order.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <functional>
#include "../../include/seerhip.h"
#include "host_pool.h"

static const char *const kNotes[9] = {"af-filter", "pre-filtering-failed", "bad-chisq", "high-bse", "perfectly-separable-data",
                                      "matrix-inversion-error", "firth-fail", "missing-data-error", "lrt-filtering-failed"};

// '%.2E' without printf and without floating point: d.ddE+xx, the correctly rounded 3-significant-digit decimal (round-half-even on the EXACT
// binary value, as glibc does).  |x| = m * 2^q (m the 53-bit significand) times 10^(2-e) ~ P * 2^s (P a 64-bit significand, relative error
// < 2^-62) is one 64 x 64 -> 128-bit product Z; the three digits are Z >> t and the rounding decision is the fraction below them against one
// half.  Z carries an absolute error below 2^56; whenever the fraction is closer than 2^58 to the half (which includes every exact tie of a
// short decimal such as 1.125) or to either end, and for subnormals, the answer is left to snprintf, so the fast path never decides a case
// its arithmetic could get wrong.  History: snprintf cost 1.2 us per 6-number row (the sink ran at 7 M rows/s, below the engine); an 80-bit
// x87 version of this function 0.5 us; this one ~0.1 us.  tests/test_sink_cpu.py sweeps it against '%.2E'.
#include <cfloat>
// the table below takes the 64-bit significand of 10^k from long double arithmetic: exact to 2^-63 only where long double is the 80-bit x87
// format (LDBL_MANT_DIG = 64).  Anywhere else the margin argument of put_num does not hold and every number goes through snprintf.
#if LDBL_MANT_DIG >= 64
static constexpr bool kFastNum = true;
#else
static constexpr bool kFastNum = false;
#endif
struct Pow10 { uint64_t m; int e; };
static Pow10 g_pow10[700];
static double g_pow10d[700];                       // 10^k as a double: the branch-free first guess of the decimal exponent
static uint32_t g_dig[1000];                       // "d.dd" of 100 .. 999
static uint64_t g_exp[700]; static uint8_t g_explen[700];   // "E+05" / "E-123" (little-endian bytes) and its length, by e + 345
static const bool g_pow10_init = [] {
    for (int k = -345; k < 355; ++k) {
        int ex; const long double fr = frexpl(powl(10.0L, (long double)k), &ex);        // 10^k = fr * 2^ex, fr in [0.5, 1)
        g_pow10[k + 345].m = (uint64_t)ldexpl(fr, 64); g_pow10[k + 345].e = ex - 64;
        g_pow10d[k + 345] = k < -323 ? 0.0 : (k > 308 ? INFINITY : (double)powl(10.0L, (long double)k));
        char b[8] = {0}; const int n = snprintf(b, sizeof b, "E%c%02d", k < 0 ? '-' : '+', k < 0 ? -k : k);
        uint64_t v = 0; memcpy(&v, b, 8); g_exp[k + 345] = v; g_explen[k + 345] = (uint8_t)n;
    }
    for (int d = 100; d < 1000; ++d) { const char b[4] = {(char)('0' + d / 100), '.', (char)('0' + (d / 10) % 10), (char)('0' + d % 10)}; memcpy(&g_dig[d], b, 4); }
    return true; }();

static char *put_num_slow(char *w, double x) { return w + snprintf(w, 32, "%.2E", x); }

// writes at most 16 bytes at w (12 of them text), returns the new end.  The common path has no data-dependent branch: sign, exponent
// guess, rounding and the digit / exponent text are selects and table stores.
static inline char *put_num(char *w, double x)
{
    uint64_t bits; memcpy(&bits, &x, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    if (__builtin_expect(be == 0x7ff, 0)) return w;                                    // nan / inf: empty field
    if (!kFastNum) return put_num_slow(w, x);
    const uint64_t frac52 = bits & ((1ull << 52) - 1);
    if (__builtin_expect(be == 0, 0)) {
        if (frac52 == 0) { if (bits >> 63) *w++ = '-'; memcpy(w, "0.00E+00", 8); return w + 8; }
        return put_num_slow(w, x);                                                     // subnormal
    }
    const uint64_t m = frac52 | (1ull << 52);
    const int q = be - 1075;                                                           // |x| = m * 2^q
    double ax; { const uint64_t ab = bits & ~(1ull << 63); memcpy(&ax, &ab, 8); }
    int e = ((be - 1023) * 78913) >> 18;                                               // floor((be - 1023) * log10 2): floor(log10 |x|) or one less
    e += (ax >= g_pow10d[e + 1 + 345]) ? 1 : 0;
    uint64_t D = 0; unsigned __int128 fr = 0; int t = 0;
    for (int tries = 0; tries < 3; ++tries) {                                          // (a second turn only next to a power of ten)
        const Pow10 &p = g_pow10[(2 - e) + 345];
        const unsigned __int128 Z = (unsigned __int128)m * p.m;                        // |x| * 10^(2 - e) = Z * 2^(q + p.e)
        t = -(q + p.e);
        if (__builtin_expect(t < 100 || t > 116, 0)) return put_num_slow(w, x);     // (D in [100, 1000) implies 105 <= t <= 110)
        const uint64_t hi = (uint64_t)(Z >> 64);
        D = hi >> (t - 64);
        fr = Z & ((((unsigned __int128)1) << t) - 1);
        if (__builtin_expect(D >= 1000, 0)) { ++e; continue; }
        if (__builtin_expect(D < 100, 0)) { --e; continue; }
        break;
    }
    if (__builtin_expect(D < 100 || D >= 1000, 0)) return put_num_slow(w, x);
    // the rounding decision on the top 64 bits of the fraction: the error of Z (< 2^56) is < 2^(56 + 64 - t) <= 2^20 units of that 64-bit
    // fraction at t >= 100; a margin of 2^24 units is kept around one half and at both ends
    const uint64_t f64 = (uint64_t)(fr >> (t - 64));                                   // fraction as a 64-bit fixed-point number in [0, 1)
    const uint64_t half = 1ull << 63, margin = 1ull << 24;
    if (__builtin_expect(f64 < margin || f64 > ~margin || (f64 > half - margin && f64 < half + margin), 0)) return put_num_slow(w, x);
    int d = (int)D + (int)(f64 >> 63);
    if (__builtin_expect(d == 1000, 0)) { d = 100; ++e; }
    *w = '-'; w += bits >> 63;
    memcpy(w, &g_dig[d], 4); w += 4;
    memcpy(w, &g_exp[e + 345], 8);
    return w + g_explen[e + 345];
}

// one part's output: a malloc'ed buffer grown by doubling (no zero fill, no per-append capacity check: the row loop asks for a row's
// upper bound once).  The buffers outlive the call: a block's 10 - 20 MB of text would otherwise be page-faulted in afresh on every call,
// under the process's one mmap lock with all threads at it.  They belong to the CALLING thread (thread_local): until round 4 they were one
// process-wide array behind a mutex held for the whole call, so the sinks of eight device streams formatted one after the other.
struct alignas(128) Part {                          // (a cache line pair of its own: the row loop updates n per row)
    char *p = nullptr; size_t n = 0, cap = 0;
    ~Part() { free(p); }
    inline char *room(size_t need) {
        if (n + need > cap) { cap = std::max(cap * 2, n + need + (1u << 16)); p = (char *)realloc(p, cap); if (!p) abort(); }
        return p + n;
    }
};
static constexpr int kMaxParts = 32;
struct Parts { Part a[kMaxParts]; };
static Parts &my_parts() { static thread_local Parts p; return p; }
static Part &my_gather() { static thread_local Part g; return g; }     // the calling thread's gathered text (format_core with own_text)

// The block loop of the library (csrc/job_run.inc) hands a block's gathered text to a writer thread instead of copying it: the calling thread's
// gather buffer is detached (text must be that buffer; NULL otherwise -- e.g. the one-part case, whose text is the part itself) and a
// buffer the writer is done with is attached in its place (its pages are already faulted in).
extern "C" char *format_gather_detach(const char *text, size_t *cap)
{
    Part &g = my_gather();
    if (!text || g.p != text) return nullptr;
    char *p = g.p; *cap = g.cap;
    g.p = nullptr; g.n = 0; g.cap = 0;
    return p;
}
extern "C" void format_gather_attach(char *p, size_t cap)
{
    Part &g = my_gather();
    free(g.p);
    g.p = p; g.n = 0; g.cap = cap;
}

// how many sinks are inside the formatter right now (tests/test_sink_cpu.py: two sinks format concurrently)
static std::atomic<int> g_inside{0}, g_inside_max{0};
struct Inside { Inside() { const int n = ++g_inside; int m = g_inside_max.load(); while (n > m && !g_inside_max.compare_exchange_weak(m, n)) {} } ~Inside() { --g_inside; } };

// The rows are cut into `nth` parts, part i into the caller's parts[i]; the parts are shared out between the calling thread and whatever
// workers of the process-wide pool are idle (host_pool.h: one CPU budget for readers, stagers and sinks; nothing is started per call).
// Row r prints the values at position vp = sel ? sel[r] : r of cols / betas / flags / lineage and the name (and carrier count) of variant
// np = name_idx ? name_idx[r] : vp.  af_counts != NULL: a first column count / n_samples (the reference's af, input.py:446) before cols.
static int64_t format_core(const char *names, const int64_t *name_off, const int64_t *sel, const int32_t *name_idx, int64_t nsel,
                           const int32_t *af_counts, int n_samples, const double *const *cols, int ncol, int64_t betas_stride,
                           const double *betas, int q, const uint8_t *betas_valid, const int32_t *lineage, const char *const *lineage_labels,
                           int n_labels, const uint32_t *flags, char *out, int64_t cap, const char **own_text,
                           const uint8_t *smp_bits = nullptr, int64_t smp_row_bytes = 0, const char *smp_names = nullptr, const int64_t *smp_off = nullptr,
                           const int32_t *smp_order = nullptr)
{
    Inside in_;
    int nth = std::min(shost::host_cpus(), kMaxParts);
    if (nsel < 4096) nth = 1;
    size_t lab_max = 2;                                                    // "NA"
    std::vector<size_t> lab_len((size_t)std::max(n_labels, 0));
    for (int l = 0; l < n_labels; ++l) { lab_len[(size_t)l] = strlen(lineage_labels[l]); lab_max = std::max(lab_max, lab_len[(size_t)l]); }
    // --print-samples (utils.py:96-98): the carriers' names, then the others', comma-joined, in the reference's (sorted) order, from the row's bits
    size_t smp_room = 0;
    if (smp_bits) smp_room = (size_t)smp_off[n_samples] + (size_t)n_samples + 4;
    const size_t fixed = (size_t)(ncol + 1 + std::max(q, 0)) * 13 + lab_max + 1 + 160 + 2 + 8 + smp_room;  // all but the name: numbers <= 12 + tab (put_num stores 8 bytes at its tail), notes <= 149
    Part *const parts = my_parts().a;
    const std::function<void(int64_t)> body = [&](int64_t part) {
        const int64_t lo = nsel * part / nth, hi = nsel * (part + 1) / nth;
        Part &s = parts[(size_t)part];
        s.n = 0;
        s.room((size_t)(hi - lo) * 96 + fixed);
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t v = sel ? sel[r] : r;
            const int64_t nv = name_idx ? (int64_t)name_idx[r] : v;
            const size_t nl = (size_t)(name_off[nv + 1] - name_off[nv]);
            char *w = s.room(nl + fixed);
            memcpy(w, names + name_off[nv], nl); w += nl;
            if (af_counts) { *w++ = '\t'; w = put_num(w, (double)af_counts[nv] / (double)n_samples); }
            for (int c = 0; c < ncol; ++c) { *w++ = '\t'; w = put_num(w, cols[c][v]); }
            if (q > 0 && betas_valid[v])
                for (int j = 0; j < q; ++j) { *w++ = '\t'; w = put_num(w, betas[(size_t)v * (size_t)(betas_stride == 1 ? q : 1) + (size_t)j * (size_t)betas_stride]); }
            if (lineage) {
                *w++ = '\t';
                const int32_t l = lineage[v];
                if (l >= 0 && l < n_labels) { memcpy(w, lineage_labels[l], lab_len[(size_t)l]); w += lab_len[(size_t)l]; }
                else { *w++ = 'N'; *w++ = 'A'; }
            }
            if (smp_bits) {
                const uint8_t *row = smp_bits + (size_t)nv * (size_t)smp_row_bytes;
                for (int pass = 1; pass >= 0; --pass) {                   // carriers first (kstrains), then the rest (nkstrains)
                    *w++ = '\t';
                    bool first = true;
                    for (int i = 0; i < n_samples; ++i) {
                        const int32_t sidx = smp_order[i];
                        if ((int)((row[sidx >> 3] >> (sidx & 7)) & 1u) != pass) continue;
                        if (!first) *w++ = ',';
                        const size_t k = (size_t)(smp_off[sidx + 1] - smp_off[sidx]);
                        memcpy(w, smp_names + smp_off[sidx], k); w += k; first = false;
                    }
                }
            }
            *w++ = '\t';
            const uint32_t f = flags[v] & 0x1ffu;
            if (f) {
                bool first = true;
                for (int b = 0; b < 9; ++b)
                    if ((f >> b) & 1u) { if (!first) *w++ = ','; const size_t k = strlen(kNotes[b]); memcpy(w, kNotes[b], k); w += k; first = false; }
            }
            *w++ = '\n';
            s.n = (size_t)(w - s.p);
        }
    };
    {
        shost::CpuScope cs(shost::ST_FORMAT);
        shost::pool().run(nth, 1, body, shost::ST_FORMAT);
    }
    if (nth == 1 && own_text) { *own_text = parts[0].p; return (int64_t)parts[0].n; }      // (the caller reads the part itself: no second copy)
    std::vector<int64_t> start((size_t)nth + 1, 0);
    for (int i = 0; i < nth; ++i) start[(size_t)i + 1] = start[(size_t)i] + (int64_t)parts[(size_t)i].n;
    const int64_t total = start[(size_t)nth];
    if (own_text) {                                                        // gather into a buffer of the calling thread's own
        Part &gather = my_gather();
        gather.n = 0; gather.room((size_t)total + 16);
        out = gather.p; cap = total; *own_text = gather.p;
    }
    if (!out || total > cap) return -(total + 1);                      // caller retries with at least `total` bytes
    const std::function<void(int64_t)> cat = [&](int64_t part) { memcpy(out + start[(size_t)part], parts[(size_t)part].p, parts[(size_t)part].n); };
    {
        shost::CpuScope cs(shost::ST_FORMAT);
        shost::pool().run(nth, 1, cat, shost::ST_FORMAT);
    }
    return total;
}

extern "C" int64_t sh_format_rows(const char *names, const int64_t *name_off, const int64_t *sel, int64_t nsel,
                                  const double *const *cols, int ncol, const double *betas, int q, const uint8_t *betas_valid,
                                  const int32_t *lineage, const char *const *lineage_labels, int n_labels,
                                  const uint32_t *flags, char *out, int64_t cap)
{
    if (!names || !name_off || !sel || !cols || !flags || nsel < 0 || ncol < 1) return -1;
    if (q > 0 && (!betas || !betas_valid)) return -1;
    return format_core(names, name_off, sel, nullptr, nsel, nullptr, 0, cols, ncol, 1, betas, q, betas_valid, lineage, lineage_labels, n_labels, flags,
                       out, cap, nullptr);
}

// The job stream's sink (csrc/job_api.inc): `nsel` compacted records as the device left them -- idx[r] = the variant's row in its block, its
// flags, its statistics column by column (cols[c][r]; the covariate slopes column-major too: betas[j * stride + r]) -- into text owned by the
// calling thread (valid until its next call).  Returns the number of bytes.
int64_t sh_format_records(const char *names, const int64_t *name_off, const int32_t *counts, int n_samples, const int32_t *idx, int64_t nsel,
                          const double *const *cols, int ncol, const double *betas, int64_t betas_stride, int q, const uint8_t *betas_valid,
                          const int32_t *lineage, const char *const *lineage_labels, int n_labels, const uint32_t *flags, const char **text)
{
    return format_core(names, name_off, nullptr, idx, nsel, counts, n_samples, cols, ncol, betas_stride, betas, q, betas_valid, lineage, lineage_labels,
                       n_labels, flags, nullptr, 0, text);
}

// sh_format_records with the two sample lists of --print-samples (the job stream: csrc/job_api.inc sh_job_set_samples)
extern "C" int64_t format_records_samples(const char *names, const int64_t *name_off, const int32_t *counts, int n_samples, const int32_t *idx, int64_t nsel,
                                          const double *const *cols, int ncol, const double *betas, int64_t betas_stride, int q, const uint8_t *betas_valid,
                                          const int32_t *lineage, const char *const *lineage_labels, int n_labels, const uint32_t *flags, const char **text,
                                          const uint8_t *smp_bits, int64_t smp_row_bytes, const char *smp_names, const int64_t *smp_off, const int32_t *smp_order)
{
    return format_core(names, name_off, nullptr, idx, nsel, counts, n_samples, cols, ncol, betas_stride, betas, q, betas_valid, lineage, lineage_labels,
                       n_labels, flags, nullptr, 0, text, smp_bits, smp_row_bytes, smp_names, smp_off, smp_order);
}

extern "C" int sh_format_concurrency_max(int reset)
{
    const int m = g_inside_max.load();
    if (reset) g_inside_max.store(0);
    return m;
}
