// reader.cpp -- native k-mer text reader / packer: gzip or plain "KMER | sample:count sample:count ..."
// lines -> packed presence rows in phenotype order, the input format of sh_lmm_batch / sh_glm_batch.
//
// Replaces the per-line Python of pyseer/input.py:301-454 (read_variant, k-mer branch) for the GPU feed:
//   var_name = first whitespace token;  strains = segment between the first and the second '|', whitespace-split,
//   each token cut at ':' (input.py:377-388);  presence over the phenotyped samples in phenotype order (input.py:438-452).
// One thread inflates (gzip is sequential), lines of a chunk are parsed in parallel (OpenMP).
#include <zlib.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include "../../include/seerhip.h"

// open-addressing table keyed by the sample name bytes (no per-token allocation; FNV-1a)
struct NameTable {
    std::vector<int32_t> slot;           // index into names, or -1
    std::vector<std::string> names;
    uint32_t mask = 0;
    static inline uint32_t hash(const char *s, size_t n) { uint32_t h = 2166136261u; for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 16777619u; } return h; }
    void build(const char *const *nm, int n) {
        uint32_t cap = 16; while (cap < (uint32_t)n * 4u) cap <<= 1;
        slot.assign(cap, -1); mask = cap - 1; names.resize(n);
        for (int i = 0; i < n; ++i) {
            names[i] = nm[i];
            uint32_t h = hash(names[i].data(), names[i].size()) & mask;
            bool dup = false;
            while (slot[h] >= 0) { if (names[slot[h]] == names[i]) { dup = true; break; } h = (h + 1) & mask; }
            if (!dup) slot[h] = i;       // first occurrence wins
        }
    }
    inline int find(const char *s, size_t n) const {
        uint32_t h = hash(s, n) & mask;
        for (;;) {
            const int32_t i = slot[h];
            if (i < 0) return -1;
            const std::string &c = names[i];
            if (c.size() == n && memcmp(c.data(), s, n) == 0) return i;
            h = (h + 1) & mask;
        }
    }
};

struct sh_reader {
    gzFile gz = nullptr;                 // zlib reads plain files transparently as well
    int n = 0;
    NameTable index;
    std::vector<char> buf;               // unconsumed text
    size_t pos = 0;                      // first unconsumed byte of buf
    bool eof = false;
    int64_t names_needed = 0;            // bytes of variant names the last refused call (-2) would have written
    std::string err;
};

static thread_local std::string g_rerr;

extern "C" {

const char *sh_reader_error(void) { return g_rerr.c_str(); }

sh_reader *sh_reader_open(const char *path, const char *const *sample_names, int n_samples)
{
    if (!path || !sample_names || n_samples < 1) { g_rerr = "bad argument"; return nullptr; }
    sh_reader *r = new sh_reader();
    r->gz = gzopen(path, "rb");
    if (!r->gz) { g_rerr = std::string("cannot open ") + path; delete r; return nullptr; }
    gzbuffer(r->gz, 1 << 20);
    r->n = n_samples;
    r->index.build(sample_names, n_samples);
    return r;
}

void sh_reader_close(sh_reader *r)
{
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    delete r;
}

// Parses up to max_variants lines.  bits: max_variants * row_bytes (zeroed here); counts[v] = carriers among the phenotyped
// samples (af = counts/n, input.py:446); names: concatenated variant names, name_off[v]..name_off[v+1].
// Returns the number of variants parsed (0 at end of file), or -1 on error (sh_reader_error()).
int64_t sh_reader_next(sh_reader *r, int64_t max_variants, uint8_t *bits, int64_t row_bytes, int32_t *counts,
                       char *names, int64_t names_cap, int64_t *name_off)
{
    if (!r || !bits || !counts || !names || !name_off || max_variants < 1) { g_rerr = "bad argument"; return -1; }
    if (row_bytes * 8 < r->n) { g_rerr = "row_bytes too small"; return -1; }
    auto t0 = std::chrono::steady_clock::now();
    // release the text handed out by earlier calls: line offsets are only recorded from here on, so compacting is safe, and the buffer
    // never holds more than one call's lines plus one slab (the reference streams line by line; a k-mer file is tens of GB inflated)
    if (r->pos > 0) { r->buf.erase(r->buf.begin(), r->buf.begin() + r->pos); r->pos = 0; }
    const size_t pos0 = r->pos;
    // ---- collect up to max_variants complete lines in the buffer
    std::vector<std::pair<size_t, size_t>> lines;            // [begin, end) without the newline
    size_t scan = r->pos;
    for (;;) {
        while ((int64_t)lines.size() < max_variants) {
            const char *base = r->buf.data();
            const void *nl = scan < r->buf.size() ? memchr(base + scan, '\n', r->buf.size() - scan) : nullptr;
            if (!nl) break;
            const size_t e = (const char *)nl - base;
            lines.emplace_back(r->pos, e);
            r->pos = scan = e + 1;
        }
        if ((int64_t)lines.size() >= max_variants || r->eof) break;
        // refill: read another slab behind what is buffered
        const size_t old = r->buf.size();
        const size_t slab = 8u << 20;
        r->buf.resize(old + slab);
        const int got = gzread(r->gz, r->buf.data() + old, (unsigned)slab);
        if (got < 0) { int en; g_rerr = gzerror(r->gz, &en); return -1; }
        r->buf.resize(old + (size_t)got);
        if (got == 0) r->eof = true;
        scan = std::max(scan, r->pos);
        // re-scan from where we stopped (lines found so far keep their offsets because we only compact when none are pending)
    }
    if (r->eof && r->pos < r->buf.size() && (int64_t)lines.size() < max_variants) {   // last line without a trailing newline
        lines.emplace_back(r->pos, r->buf.size());
        r->pos = r->buf.size();
    }
    const int64_t nv = (int64_t)lines.size();
    if (nv == 0) return 0;
    auto t1 = std::chrono::steady_clock::now();
    // ---- names (serial: offsets), presence (parallel).  Unitig names run to tens of kilobases: the total is measured BEFORE anything is
    // written, and a call that does not fit is refused without consuming its lines (-2; sh_reader_names_needed() says how much to bring)
    const char *base = r->buf.data();
    std::vector<std::pair<const char *, const char *>> nm((size_t)nv);
    int64_t need = 0;
    for (int64_t v = 0; v < nv; ++v) {
        const char *p = base + lines[v].first, *e = base + lines[v].second;
        while (p < e && (*p == ' ' || *p == '\t')) ++p;
        const char *q = p;
        while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
        nm[v] = {p, q}; need += q - p;
    }
    if (need > names_cap) { r->names_needed = need; r->pos = pos0; g_rerr = "names buffer too small"; return -2; }
    memset(bits, 0, (size_t)nv * row_bytes);
    int64_t off = 0;
    for (int64_t v = 0; v < nv; ++v) { name_off[v] = off; memcpy(names + off, nm[v].first, nm[v].second - nm[v].first); off += nm[v].second - nm[v].first; }
    name_off[nv] = off;
    const auto &index = r->index;
#pragma omp parallel
    {
#pragma omp for schedule(dynamic, 16)
        for (int64_t v = 0; v < nv; ++v) {
            const char *p = base + lines[v].first, *e = base + lines[v].second;
            const char *bar = (const char *)memchr(p, '|', e - p);
            uint8_t *row = bits + v * row_bytes;
            int cnt = 0;
            if (bar) {
                const char *s = bar + 1;
                const char *bar2 = (const char *)memchr(s, '|', e - s);
                const char *end = bar2 ? bar2 : e;
                while (s < end) {
                    while (s < end && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;
                    const char *t = s;
                    while (t < end && *t != ' ' && *t != '\t' && *t != '\r') ++t;
                    if (t > s) {
                        const char *colon = (const char *)memchr(s, ':', t - s);
                        const int i = index.find(s, (colon ? colon : t) - s);
                        if (i >= 0) {
                            if (!((row[i >> 3] >> (i & 7)) & 1)) { row[i >> 3] |= (uint8_t)(1u << (i & 7)); ++cnt; }
                        }
                    }
                    s = t;
                }
            }
            counts[v] = cnt;
        }
    }
    if (std::getenv("SEERHIP_READER_DEBUG")) {
        auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[reader] %lld lines: read+split %.3fs parse %.3fs\n", (long long)nv,
                std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    }
    return nv;
}

int64_t sh_reader_names_needed(sh_reader *r) { return r ? r->names_needed : 0; }
int64_t sh_reader_buffered(sh_reader *r) { return r ? (int64_t)r->buf.size() : 0; }

}  // extern "C"
