// writer.cpp -- native result sink: array-backed results -> the reference's TSV rows.
//
// Replaces the per-row Python of pyseer/utils.py:39-105 (format_output) and the print loop of pyseer/__main__.py:805-827:
//   variant \t af \t filter-pvalue \t lrt-pvalue \t beta \t beta-std-err \t (variant_h2 | intercept [\t covariate betas...])
//   [\t lineage] \t notes
// numbers as '%.2E' (utils.py:60-75: '%.2E' % Decimal(x), i.e. the correctly rounded 3-significant-digit form, which glibc's
// printf also produces), non-finite -> empty field; lineage label or NA; notes joined by ',' in the flag-bit order of
// include/seerhip.h.  Rows are formatted in parallel (OpenMP) into per-thread buffers and concatenated in order.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/seerhip.h"

static const char *const kNotes[9] = {"af-filter", "pre-filtering-failed", "bad-chisq", "high-bse", "perfectly-separable-data",
                                      "matrix-inversion-error", "firth-fail", "missing-data-error", "lrt-filtering-failed"};

// '%.2E' without printf: d.ddE+xx, the correctly rounded 3-significant-digit decimal (round-half-even on the EXACT binary value, as glibc
// does).  |x| is scaled to [100, 1000) with an 80-bit power of ten and rounded; whenever the scaled value is closer than 1e-6 to a rounding
// boundary (x.5 ties, which are exactly representable for short decimals such as 1.125, and the 99.9999... / 999.5 edges) the answer is
// left to snprintf, so the fast path never decides a case its arithmetic could get wrong (relative error of the scaling < 1e-17).
// snprintf cost 1.2 us per 6-number row: the whole sink ran at 7 M rows/s, below the engine.  Checked against snprintf on 2e8 values
// (tests/test_sink_cpu.py runs a smaller sweep).
static long double g_pow10[700];
static const bool g_pow10_init = [] { for (int k = -345; k < 355; ++k) g_pow10[k + 345] = powl(10.0L, (long double)k); return true; }();

static inline void put_num(std::string &s, double x)
{
    if (!std::isfinite(x)) return;
    char b[40];
    if (x == 0.0) { s.append(std::signbit(x) ? "-0.00E+00" : "0.00E+00"); return; }
    const double ax = std::fabs(x);
    int e2; (void)std::frexp(ax, &e2);
    int e = (int)std::floor((e2 - 1) * 0.30102999566398120);              // floor(log10) within one
    long double sc = (long double)ax * g_pow10[(2 - e) + 345];
    if (sc >= 1000.0L) { ++e; sc = (long double)ax * g_pow10[(2 - e) + 345]; }
    else if (sc < 100.0L) { --e; sc = (long double)ax * g_pow10[(2 - e) + 345]; }
    const long double fl = floorl(sc), fr = sc - fl;
    const bool edge = sc < 100.000001L || sc > 999.499999L || (fr > 0.499999L && fr < 0.500001L) || ax < 1e-300 || e < -340 || e > 340;
    if (edge) { const int n = snprintf(b, sizeof b, "%.2E", x); s.append(b, (size_t)n); return; }
    int m = (int)fl + (fr > 0.5L ? 1 : 0);                                 // 100 .. 999 (1000 is an edge case above)
    char *w = b;
    if (x < 0) *w++ = '-';
    *w++ = (char)('0' + m / 100); *w++ = '.'; *w++ = (char)('0' + (m / 10) % 10); *w++ = (char)('0' + m % 10);
    *w++ = 'E';
    int ea = e;
    if (ea < 0) { *w++ = '-'; ea = -ea; } else *w++ = '+';
    if (ea >= 100) { *w++ = (char)('0' + ea / 100); ea %= 100; }
    *w++ = (char)('0' + ea / 10); *w++ = (char)('0' + ea % 10);
    s.append(b, (size_t)(w - b));
}

// threads worth starting: the cgroup CPU quota when there is one (a GPU box shows 256 CPUs under a quota of 16; an OpenMP team of 256
// spinning threads then only steals time from the threads that feed the GPU), at most 32
static int format_threads()
{
    static const int n = [] {
        int t = 1;
#ifdef _OPENMP
        t = omp_get_max_threads();
#endif
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32]; long per = 0;
            if (fscanf(f, "%31s %ld", q, &per) == 2 && q[0] != 'm' && per > 0) { const long qq = atol(q); if (qq > 0) t = std::min<long>(t, (qq + per - 1) / per); }
            fclose(f);
        }
        return std::max(1, std::min(t, 32));
    }();
    return n;
}

extern "C" int64_t sh_format_rows(const char *names, const int64_t *name_off, const int64_t *sel, int64_t nsel,
                                  const double *const *cols, int ncol, const double *betas, int q, const uint8_t *betas_valid,
                                  const int32_t *lineage, const char *const *lineage_labels, int n_labels,
                                  const uint32_t *flags, char *out, int64_t cap)
{
    if (!names || !name_off || !sel || !cols || !flags || nsel < 0 || ncol < 1) return -1;
    if (q > 0 && (!betas || !betas_valid)) return -1;
    int nth = 1;
    nth = format_threads();
    if (nsel < 4096) nth = 1;
    std::vector<std::string> parts((size_t)nth);
#pragma omp parallel num_threads(nth)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        const int64_t lo = nsel * t / nth, hi = nsel * (t + 1) / nth;
        std::string &s = parts[(size_t)t];
        s.reserve((size_t)(hi - lo) * 96);
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t v = sel[r];
            s.append(names + name_off[v], (size_t)(name_off[v + 1] - name_off[v]));
            for (int c = 0; c < ncol; ++c) { s.push_back('\t'); put_num(s, cols[c][v]); }
            if (q > 0 && betas_valid[v])
                for (int j = 0; j < q; ++j) { s.push_back('\t'); put_num(s, betas[(size_t)v * q + j]); }
            if (lineage) {
                s.push_back('\t');
                const int32_t l = lineage[v];
                if (l >= 0 && l < n_labels) s.append(lineage_labels[l]); else s.append("NA");
            }
            s.push_back('\t');
            const uint32_t f = flags[v];
            bool first = true;
            for (int b = 0; b < 9; ++b)
                if ((f >> b) & 1u) { if (!first) s.push_back(','); s.append(kNotes[b]); first = false; }
            s.push_back('\n');
        }
    }
    int64_t total = 0;
    for (auto &p : parts) total += (int64_t)p.size();
    if (!out || total > cap) return -(total + 1);       // caller retries with at least `total` bytes
    char *w = out;
    for (auto &p : parts) { memcpy(w, p.data(), p.size()); w += p.size(); }
    return total;
}
