// inflate_par.h -- ONE gzip member decoded by several threads (reader.cpp: produce_gzip_parallel).
//
// A DEFLATE stream is sequential twice over: a block can only be found by decoding the one before it, and a match may copy from the 32 KB
// of text before its block.  Both are worked around the way pugz does it (Kerbiriou & Chikhi 2019, "Parallel decompression of gzip-compressed
// files and random access to DNA sequences"), restated here for the k-mer text:
//   1. the compressed bytes are cut into chunks; in every chunk but the first a thread LOOKS for the head of a dynamic-Huffman block, bit
//      offset by bit offset: the 17 header bits, a complete code-length code, complete literal/length and distance codes, and a first stretch
//      of symbols that decodes to printable text.  Chance passes all of that about never; if it does, step 3 finds out.
//   2. every chunk is decoded from its block head to the next chunk's, into 16-bit symbols: a literal is its byte, and a copy that reaches
//      back before the chunk's first byte copies MARKERS -- the decoder starts with 32 768 of them in front of its output, marker k standing
//      for byte k of the unknown window -- so matches are plain 16-bit copies and need no special case.
//   3. in stream order: chunk j is accepted iff chunk j-1 was accepted and ENDED on the very bit chunk j started from (so a false block head
//      costs its chunk, nothing else); its window is the last 32 KB of the text before it, known once chunk j-1's last 32 KB are resolved;
//      the symbols of all accepted chunks are then translated to bytes in parallel.
// The first chunk of a round starts from the exact position the previous round ended on, with the real window in front of it: same code,
// no markers.  The member's CRC-32 and ISIZE are checked by the caller as for the one-thread decoder.  Test infrastructure: zlib
// (tests/test_reader_cpu.py decodes the same files through python's gzip and through SEERHIP_ROUTE reader=serial).
#pragma once
#include <vector>
#include <new>
#include <cstdlib>
#include <emmintrin.h>
#include "inflate_fast.h"

namespace shinf {

static const uint32_t PAR_WIN = 32768;

static inline uint64_t par_bitpos(const Decoder &d, const uint8_t *base) { return (uint64_t)(d.in - base) * 8 - (uint64_t)d.bitcnt; }
static inline void par_seek(Decoder &d, const uint8_t *base, const uint8_t *end, uint64_t bit)
{
    d.in = base + (bit >> 3); d.in_end = end; d.bitbuf = 0; d.bitcnt = 0; d.err = nullptr;
    d.refill(); d.drop((int)(bit & 7));
    d.state = Decoder::BLOCK_HEAD; d.raw = true; d.last_block = false;
}

// the symbols of a chunk: grown by realloc (large blocks move by page remapping: no copy, no zero fill, no second round of page faults --
// a std::vector grown to 900 MB of symbols decoded at 88 MB/s, its second use at 940)
struct SymBuf {
    uint16_t *p = nullptr; size_t cap = 0;
    SymBuf() = default; SymBuf(const SymBuf &) = delete; SymBuf &operator=(const SymBuf &) = delete;
    ~SymBuf() { free(p); }
    size_t size() const { return cap; }
    uint16_t *data() { return p; }
    const uint16_t *data() const { return p; }
    uint16_t &operator[](size_t i) { return p[i]; }
    void resize(size_t n) { if (n <= cap) return; void *q = realloc(p, n * sizeof(uint16_t)); if (!q) throw std::bad_alloc(); p = (uint16_t *)q; cap = n; }
};

struct ParChunk {
    uint64_t start_bit = 0, stop_bit = ~0ull;     // decode from a block head at start_bit up to the first block head at or after stop_bit
    bool exact = false;                            // the window in front is real text (sym[0 .. PAR_WIN) hold bytes where known)
    SymBuf sym;                                    // PAR_WIN window entries, then the chunk's text as symbols
    size_t n = 0;                                  // symbols of text (sym[PAR_WIN .. PAR_WIN + n))
    uint64_t end_bit = 0; bool hit_final = false, ok = false; const char *err = nullptr;
    size_t max_syms = 0;                           // a searching thread's bound on the text of its region (0 = none): a false head in a highly compressible
                                                   // stretch could otherwise grow the buffer to gigabytes before stop_bit; beyond it the decode gives up
                                                   // (std::bad_alloc, which the caller takes as "not a head") and the producer decodes the region itself
};

static inline bool par_is_text(const uint16_t *p, const uint16_t *e)
{
    for (; p < e; ++p) {
        const uint16_t s = *p;
        // literals: printable ASCII, or a byte >= 0x80 (sample names in UTF-8 / latin-1: until round 5 such a file never yielded an accepted
        // region and every region was decoded twice); markers (>= 0x8000) stand for unknown bytes of the window
        if (s < 0x80 && !(s == '\n' || s == '\t' || s == '\r' || (s >= 32 && s < 127))) return false;
    }
    return true;
}

// Decode blocks into c.sym from c.start_bit.  `probe`: stop after `probe_syms` symbols or one block, and insist on printable literals
// (the search for a block head); returns false if anything is wrong.
static bool par_decode(Decoder &d, const uint8_t *base, const uint8_t *end, ParChunk &c, bool probe = false, size_t probe_syms = 0)
{
    par_seek(d, base, end, c.start_bit);
    if (c.sym.size() < PAR_WIN + (1u << 16)) c.sym.resize(PAR_WIN + (1u << 16));
    uint16_t *buf = c.sym.data();
    uint16_t *out = buf + PAR_WIN, *lim = buf + c.sym.size();
    auto grow = [&](size_t need) {
        const size_t at = (size_t)(out - buf);
        if ((size_t)(lim - out) >= need) return;
        if (c.max_syms && at + need > PAR_WIN + c.max_syms) throw std::bad_alloc();
        c.sym.resize(std::max(c.sym.size() + need, c.sym.size() + c.sym.size() / 2));
        buf = c.sym.data(); out = buf + at; lim = buf + c.sym.size();
    };
    int blocks = 0;
    for (;;) {
        if (d.state == Decoder::BLOCK_HEAD) {
            const uint64_t pos = par_bitpos(d, base);
            if (!probe && pos >= c.stop_bit) { c.end_bit = pos; c.ok = true; break; }
            if (probe && blocks == 1) { c.end_bit = pos; c.ok = true; break; }
            if (!d.read_block_head()) { c.err = d.err; return false; }
            ++blocks;
            continue;
        }
        if (d.state == Decoder::STORED) {
            size_t left = d.stored_left;
            if (left > (size_t)(d.in_end - d.in)) { c.err = "truncated stored block"; return false; }
            grow(left + 8);
            for (size_t i = 0; i < left; ++i) out[i] = d.in[i];
            out += left; d.in += left; d.stored_left = 0;
            if (d.last_block) { c.end_bit = par_bitpos(d, base); c.hit_final = true; c.ok = true; break; }
            d.state = Decoder::BLOCK_HEAD;
            continue;
        }
        if (d.state != Decoder::CODES) { c.err = d.err ? d.err : "unexpected decoder state"; return false; }
        // ---- the symbols of one block (inflate_fast.h's loop, 16-bit stores) ----
        bool eob = false;
        while (!eob) {
            if ((size_t)(lim - out) < 1200) grow(1u << 20);             // (strictly more than the inner loop's margin: it must make progress)
            if (probe && (size_t)(out - (buf + PAR_WIN)) >= probe_syms) {
                if (!par_is_text(buf + PAR_WIN, out)) { c.err = "not text"; return false; }
                c.end_bit = par_bitpos(d, base); c.ok = true; c.n = (size_t)(out - (buf + PAR_WIN)); return true;
            }
            uint16_t *const stop = lim - 600;
            while (out < stop) {
                const bool full = d.refill();
                uint32_t e = d.ll[d.peek(LL_BITS)];
                if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; d.drop(r); e = d.ll[(e >> 16) + d.peek(sb)]; }
                if (!full) {
                    if (d.bitcnt < (int)(e & 0x1F)) { c.err = "truncated deflate stream"; return false; }
                    if (e & F_LIT) { d.drop(e & 0x1F); *out++ = (uint16_t)((e >> 16) & 0xFF); continue; }
                } else if (e & F_LIT) {
                    d.drop(e & 0x1F); *out++ = (uint16_t)((e >> 16) & 0xFF);
                    e = d.ll[d.peek(LL_BITS)];
                    if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; d.drop(r); e = d.ll[(e >> 16) + d.peek(sb)]; }
                    if (e & F_LIT) {
                        d.drop(e & 0x1F); *out++ = (uint16_t)((e >> 16) & 0xFF);
                        e = d.ll[d.peek(LL_BITS)];
                        if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; d.drop(r); e = d.ll[(e >> 16) + d.peek(sb)]; }
                        if (e & F_LIT) { d.drop(e & 0x1F); *out++ = (uint16_t)((e >> 16) & 0xFF); continue; }
                    }
                    d.refill();
                }
                d.drop(e & 0x1F);
                if (e & F_EOB) {
                    if (d.bitcnt < 0) { c.err = "truncated deflate stream"; return false; }
                    eob = true; break;
                }
                if ((e >> 16) == 0) { c.err = "invalid literal/length symbol"; return false; }
                const int xb = (e >> 8) & 0x1F;
                const uint32_t len = (e >> 16) + d.peek(xb); d.drop(xb);
                if (d.bitcnt < 32) d.refill();
                uint32_t dc = d.dd[d.peek(D_BITS)];
                if (dc & F_SUB) { const int r = dc & 0x1F, sb = (dc >> 8) & 0x1F; d.drop(r); dc = d.dd[(dc >> 16) + d.peek(sb)]; }
                if ((dc >> 16) == 0) { c.err = "invalid distance symbol"; return false; }
                d.drop(dc & 0x1F);
                const int dxb = (dc >> 8) & 0x1F;
                const uint32_t dist = (dc >> 16) + d.peek(dxb); d.drop(dxb);
                if (d.bitcnt < 0) { c.err = "truncated deflate stream"; return false; }
                const uint16_t *src = out - dist;                       // (dist <= 32768 = the window in front of the output: always inside the buffer)
                uint16_t *dst = out;
                out += len;
                if (dist >= 4) { do { memcpy(dst, src, 8); dst += 4; src += 4; } while (dst < out); }
                else { do { *dst++ = *src++; } while (dst < out); }
            }
        }
        if (probe && !par_is_text(buf + PAR_WIN, out)) { c.err = "not text"; return false; }     // a block head found by search must give text
        if (d.last_block) { c.end_bit = par_bitpos(d, base); c.hit_final = true; c.ok = true; break; }
        d.state = Decoder::BLOCK_HEAD;
    }
    c.n = (size_t)(out - (buf + PAR_WIN));
    return true;
}

// Kraft sums of four 3-bit code lengths at a time: low 16 bits = sum of 128 >> l over the non-zero lengths, high 16 = how many are non-zero
struct Kraft4 { uint32_t t[4096]; Kraft4() { for (int v = 0; v < 4096; ++v) { uint32_t sum = 0, cnt = 0; for (int k = 0; k < 4; ++k) { const int l = (v >> (3 * k)) & 7; if (l) { sum += 128u >> l; ++cnt; } } t[v] = sum | (cnt << 16); } } };

// The first bit offset in [from, to) at which a dynamic block head stands the tests of step 1, or ~0.
static uint64_t par_find_block(const uint8_t *base, const uint8_t *end, uint64_t from, uint64_t to, Decoder &d, ParChunk &scratch)
{
    static const Kraft4 K4;
    const uint64_t total_bits = (uint64_t)(end - base) * 8;
    if (to > total_bits - 64 * 8) to = total_bits > 64 * 8 ? total_bits - 64 * 8 : 0;       // (the quick test reads 16 bytes ahead)
    for (uint64_t byte = from >> 3; byte * 8 < to; ++byte) {
        const uint8_t *p = base + byte;
        const uint64_t A0 = load64(p), B0 = load64(p + 7);              // stream bits 0 .. 63 and 56 .. 119 from this byte
        // BTYPE = 2 at bit offset sh: bits sh + 1, sh + 2 of the byte pair = 0, 1.  All eight offsets at once: m bit sh set <=> a candidate
        const uint32_t lo = (uint32_t)A0 & 0x3FFu;
        uint32_t m = (~(lo >> 1) & (lo >> 2)) & 0xFFu;
        if (byte * 8 < from) m &= 0xFFu << (from & 7);
        while (m) {
            const int sh = __builtin_ctz(m); m &= m - 1;
            const uint64_t bit = byte * 8 + (uint64_t)sh;
            if (bit >= to) break;
            const uint64_t A = A0 >> sh;                                // >= 57 bits of the stream from `bit`
            // BFINAL (either), BTYPE = 2, HLIT <= 29, HDIST <= 29
            const uint32_t hlit = (uint32_t)(A >> 3) & 31, hdist = (uint32_t)(A >> 8) & 31, hclen = (uint32_t)(A >> 13) & 15;
            if (hlit > 29 || hdist > 29) continue;
            // the code-length code: (hclen + 4) x 3 bits from stream bit 17, must be complete (Kraft sum 128 / 128) or a single code
            const int nc = (int)hclen + 4;
            uint64_t L = ((A >> 17) & ((1ull << 39) - 1)) | ((B0 >> sh) << 39);       // stream bits 17 .. 80
            L &= (1ull << (3 * nc)) - 1;                                 // (nc <= 19: 57 bits)
            const uint32_t k = K4.t[L & 4095] + K4.t[(L >> 12) & 4095] + K4.t[(L >> 24) & 4095] + K4.t[(L >> 36) & 4095] + K4.t[(L >> 48) & 4095];
            if ((k & 0xFFFFu) != 128u && (k >> 16) != 1u) continue;
            if ((k & 0xFFFFu) > 128u) continue;
            // the full tests: both codes complete, a first stretch of printable text
            scratch.start_bit = bit; scratch.ok = false; scratch.err = nullptr; scratch.hit_final = false;
            if (par_decode(d, base, end, scratch, true, 4096) && scratch.ok) return bit;
        }
    }
    return ~0ull;
}

// sym -> bytes with the chunk's window (win[0 .. PAR_WIN), of which the last `avail` are real); false if a marker points before them.
// Sixteen symbols at a time while none of them is a marker (SSE2: or, movemask of the top bits, packus); markers one by one.
static bool par_translate(const uint16_t *sym, size_t n, const uint8_t *win, uint32_t avail, uint8_t *dst)
{
    bool ok = true;
    const uint32_t first = PAR_WIN - avail;
    size_t i = 0;
    while (i < n) {
        while (i + 16 <= n) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(sym + i)), b = _mm_loadu_si128((const __m128i *)(sym + i + 8));
            if (_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) break;                  // a marker among the sixteen
            _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
            i += 16;
        }
        const size_t e = std::min(n, i + 16);
        for (; i < e; ++i) {
            const uint16_t s = sym[i];
            if (s < 0x8000) dst[i] = (uint8_t)s;
            else { const uint32_t k = s & 0x7FFF; if (k < first) ok = false; dst[i] = win[k]; }
        }
    }
    return ok;
}

}  // namespace shinf
