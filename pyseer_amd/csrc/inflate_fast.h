// inflate_fast.h -- a resumable DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the k-mer reader (reader.cpp).
//
// Why not zlib's inflate: a pyseer k-mer file is ~30 KB of "sample:count" tokens per line, gzip-compressed ~5:1, and the reference
// (pyseer/input.py:301-454, gzip.open) as well as zlib decode it at 250-400 MB/s on one core -- 6-10 k k-mers/s at N = 5000, three
// orders of magnitude below the engine.  This decoder is built for that text: a 64-bit bit buffer refilled by one unaligned load,
// an 11-bit first-level table whose entries carry literals ready to store and length/extra-bit fields ready to add, two literals per
// refill, and match copies in 8-byte words (the matches ARE the repeated sample names).  Output goes to a caller-provided buffer that
// keeps the last 32 KB as history; the decoder can stop when the buffer is nearly full and resume mid-block, so a file of any size
// streams through a fixed buffer.  Members of a multi-member file are decoded one after another; BGZF files (bgzip: independent
// members that announce their compressed size) are decoded member-parallel by reader.cpp.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace shinf {

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// table entry (32 bits):  bits 0..4  = number of code bits to consume (first level: the whole code, or the index width for a sub-table pointer)
//                         bits 8..12 = number of extra bits (lengths / distances)
//                         bit  13    = literal          bit 14 = end of block        bit 15 = sub-table pointer
//                         bits 16..31 = literal value | length or distance base | sub-table start
enum : uint32_t { F_LIT = 1u << 13, F_EOB = 1u << 14, F_SUB = 1u << 15 };
static const int LL_BITS = 11, D_BITS = 8;
static const int LL_ENOUGH = 2400, D_ENOUGH = 450;       // > zlib's ENOUGH bounds for (288, 11, 15) and (32, 8, 15): 2342 / 400 entries

struct Decoder {
    // input
    const uint8_t *in = nullptr, *in_end = nullptr;
    uint64_t bitbuf = 0; int bitcnt = 0;
    // state
    enum State { HEADER, BLOCK_HEAD, STORED, CODES, TRAILER, DONE, ERROR } state = HEADER;
    bool last_block = false;
    uint32_t stored_left = 0;
    uint32_t ll[LL_ENOUGH], dd[D_ENOUGH];
    const char *err = nullptr;
    uint64_t member_out = 0;              // bytes produced by the current member (ISIZE check)
    bool raw = false;                     // raw deflate stream (no gzip header / trailer): BGZF member payloads
    int members = 0;
    // every gzip member that ended during run(): where its output ended and the CRC-32 its trailer announces (the caller checks it --
    // over whole slabs, in parallel; this decoder does not touch the bytes a second time)
    struct MemberEnd { const uint8_t *at; uint32_t crc; };
    MemberEnd ends[64]; int n_ends = 0;

    void begin(const uint8_t *p, const uint8_t *e, bool raw_deflate = false)
    { in = p; in_end = e; bitbuf = 0; bitcnt = 0; state = raw_deflate ? BLOCK_HEAD : HEADER; raw = raw_deflate; last_block = false; err = nullptr; member_out = 0; members = 0; }

    inline bool refill()
    {   // >= 56 valid bits when 8 input bytes remain (true); otherwise whatever is left, byte by byte (false)
        if (in + 8 <= in_end) { bitbuf |= load64(in) << bitcnt; in += (63 - bitcnt) >> 3; bitcnt |= 56; return true; }
        while (bitcnt <= 56 && in < in_end) { bitbuf |= (uint64_t)*in++ << bitcnt; bitcnt += 8; }
        return false;
    }
    // hand the whole bytes still in the bit buffer back to the input and clear it (byte-aligned data follows)
    inline void unread_bytes() { in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0; }
    inline uint32_t peek(int n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1)); }
    inline void drop(int n) { bitbuf >>= n; bitcnt -= n; }
    bool fail(const char *m) { err = m; state = ERROR; return false; }

    // canonical Huffman decoding table: lens[n] code lengths (0 = unused), first-level width `root`
    static bool build(const uint8_t *lens, int n, int root, uint32_t *tab, int tab_cap, const uint16_t *base, const uint8_t *extra, int kind)
    {
        int count[16] = {0}, offs[16];
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        int maxlen = 15; while (maxlen > 0 && count[maxlen] == 0) --maxlen;
        if (maxlen == 0) {                                   // no codes at all: every lookup is an error (only legal for distances)
            for (int i = 0; i < (1 << root); ++i) tab[i] = 0;
            return true;
        }
        int left = 1;
        for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0 && !(kind == 1 && maxlen == 1 && count[1] == 1)) {   // incomplete: allowed only for a single distance code
            if (!(maxlen == 1 && count[1] == 1)) return false;
        }
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        uint16_t sorted[320];
        for (int i = 0; i < n; ++i) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        for (int i = 0; i < (1 << root); ++i) tab[i] = 0;
        int next_sub = 1 << root;
        uint32_t code = 0;                                    // canonical code, MSB-first; table index = bit-reversed
        int si = 0;
        int cur_sub_prefix = -1, cur_sub_bits = 0, cur_sub_start = 0;
        for (int len = 1; len <= maxlen; ++len) {
            for (int c = 0; c < count[len]; ++c, ++si) {
                const int sym = sorted[si];
                uint32_t e;
                if (kind == 0) {                               // literal/length alphabet
                    if (sym < 256) e = F_LIT | ((uint32_t)sym << 16);
                    else if (sym == 256) e = F_EOB;
                    else if (sym < 286) e = ((uint32_t)base[sym - 257] << 16) | ((uint32_t)extra[sym - 257] << 8);
                    else e = 0;                                // 286, 287: invalid when decoded
                } else {
                    if (sym < 30) e = ((uint32_t)base[sym] << 16) | ((uint32_t)extra[sym] << 8);
                    else e = 0;
                }
                // bit-reverse the code
                uint32_t rev = 0;
                for (int b = 0; b < len; ++b) rev |= ((code >> b) & 1u) << (len - 1 - b);
                if (len <= root) {
                    const uint32_t ent = e | (uint32_t)len;
                    if (e == 0 && !(kind == 0 && sym >= 286) && !(kind == 1 && sym >= 30)) { /* symbol with zero entry cannot happen */ }
                    for (uint32_t i = rev; i < (1u << root); i += 1u << len) tab[i] = ent ? ent : (uint32_t)len;   // invalid symbols: length only
                } else {
                    const int prefix = (int)(rev & ((1u << root) - 1));
                    if (prefix != cur_sub_prefix) {
                        // width of this sub-table: enough for the longest code that shares the prefix
                        int bits = len - root, lft = 1 << bits;
                        int l2 = len, cleft = count[len] - c;
                        // canonical codes are consecutive: walk the remaining counts until the sub-table is full
                        lft -= cleft;
                        while (lft > 0 && l2 < maxlen) { ++l2; ++bits; lft = (lft << 1) - count[l2]; }
                        cur_sub_prefix = prefix; cur_sub_bits = bits; cur_sub_start = next_sub;
                        next_sub += 1 << bits;
                        if (next_sub > tab_cap) return false;
                        for (int i = 0; i < (1 << bits); ++i) tab[cur_sub_start + i] = 0;
                        tab[prefix] = F_SUB | ((uint32_t)cur_sub_start << 16) | ((uint32_t)bits << 8) | (uint32_t)root;
                    }
                    const uint32_t ent = (e ? e : 0u) | (uint32_t)(len - root);
                    const uint32_t hi = rev >> root;
                    for (uint32_t i = hi; i < (1u << cur_sub_bits); i += 1u << (len - root)) tab[cur_sub_start + i] = ent;
                }
                ++code;
            }
            code <<= 1;
        }
        return true;
    }

    bool read_gzip_header()
    {
        // needs the whole header in the input (the reader maps the file)
        if (members > 0 && (in_end - in < 2 || in[0] != 0x1f || in[1] != 0x8b)) {
            // after the last member only zero padding is accepted (python's gzip module, which the reference reads through, skips zero bytes
            // and raises on anything else)
            for (const uint8_t *p = in; p < in_end; ++p) if (*p) return fail("trailing garbage after the last gzip member");
            state = DONE; return false;
        }
        if (in_end - in < 18) return fail("truncated gzip header");
        if (in[0] != 0x1f || in[1] != 0x8b) return fail("not a gzip member");
        if (in[2] != 8) return fail("unknown gzip compression method");
        const int flg = in[3];
        const uint8_t *p = in + 10;
        if (flg & 4) { if (in_end - p < 2) return fail("truncated gzip header"); const int xl = p[0] | (p[1] << 8); p += 2; if (in_end - p < xl) return fail("truncated gzip header"); p += xl; }
        if (flg & 8) { while (p < in_end && *p) ++p; if (p >= in_end) return fail("truncated gzip header"); ++p; }
        if (flg & 16) { while (p < in_end && *p) ++p; if (p >= in_end) return fail("truncated gzip header"); ++p; }
        if (flg & 2) { if (in_end - p < 2) return fail("truncated gzip header"); p += 2; }
        in = p; bitbuf = 0; bitcnt = 0; member_out = 0; ++members;
        state = BLOCK_HEAD;
        return true;
    }

    bool read_block_head()
    {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        refill();
        if (bitcnt < 3) return fail("truncated deflate stream");
        last_block = peek(1); drop(1);
        const int type = peek(2); drop(2);
        if (type == 0) {
            // stored: skip to the byte boundary; LEN, NLEN
            drop(bitcnt & 7);
            unread_bytes();
            if (in_end - in < 4) return fail("truncated stored block");
            const uint32_t len = (uint32_t)in[0] | ((uint32_t)in[1] << 8), nlen = (uint32_t)in[2] | ((uint32_t)in[3] << 8);
            in += 4;
            if ((len ^ 0xFFFFu) != nlen) return fail("stored block length check failed");
            stored_left = len;
            state = STORED;
            return true;
        }
        uint8_t lens[320];
        int nl, nd;
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            nl = 288; nd = 32;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else if (type == 2) {
            refill();
            if (bitcnt < 14) return fail("truncated dynamic block header");
            nl = (int)peek(5) + 257; drop(5);
            nd = (int)peek(5) + 1; drop(5);
            const int nc = (int)peek(4) + 4; drop(4);
            if (nl > 286 || nd > 30) return fail("too many length or distance symbols");
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < nc; ++i) { refill(); if (bitcnt < 3) return fail("truncated dynamic block header"); cl[order[i]] = (uint8_t)peek(3); drop(3); }
            uint32_t ct[1 << 7];
            // code-length code: at most 7 bits, decoded through a flat 128-entry table
            {
                int count[8] = {0}; for (int i = 0; i < 19; ++i) count[cl[i]]++;
                count[0] = 0;
                int left = 1; for (int l = 1; l <= 7; ++l) { left <<= 1; left -= count[l]; if (left < 0) return fail("over-subscribed code-length code"); }
                int nused = 0; for (int l = 1; l <= 7; ++l) nused += count[l];
                if (left > 0 && nused != 1) return fail("incomplete code-length code");
                for (int i = 0; i < 128; ++i) ct[i] = 0xFFFFFFFFu;
                uint32_t code = 0;
                for (int len = 1; len <= 7; ++len) {
                    for (int sym = 0; sym < 19; ++sym) if (cl[sym] == len) {
                        uint32_t rev = 0; for (int b = 0; b < len; ++b) rev |= ((code >> b) & 1u) << (len - 1 - b);
                        for (uint32_t i = rev; i < 128; i += 1u << len) ct[i] = ((uint32_t)sym << 8) | (uint32_t)len;
                        ++code;
                    }
                    code <<= 1;
                }
            }
            int i = 0;
            while (i < nl + nd) {
                refill();
                const uint32_t e = ct[peek(7)];
                if (e == 0xFFFFFFFFu) return fail("invalid code-length symbol");
                const int len = e & 0xFF, sym = e >> 8;
                if (bitcnt < len + 7) return fail("truncated dynamic block header");     // code + the widest repeat field
                drop(len);
                if (sym < 16) lens[i++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (i == 0) return fail("repeat with no previous length"); val = lens[i - 1]; rep = 3 + (int)peek(2); drop(2); }
                    else if (sym == 17) { rep = 3 + (int)peek(3); drop(3); }
                    else { rep = 11 + (int)peek(7); drop(7); }
                    if (i + rep > nl + nd) return fail("code lengths overrun");
                    while (rep--) lens[i++] = (uint8_t)val;
                }
            }
            if (lens[256] == 0) return fail("no end-of-block code");
            // distances follow the literal/length lengths
            memmove(lens + 288, lens + nl, nd);
            for (int k = nl; k < 288; ++k) lens[k] = 0;
            for (int k = nd; k < 32; ++k) lens[288 + k] = 0;
            nl = 288; nd = 32;
        } else return fail("invalid block type");
        if (!build(lens, 288, LL_BITS, ll, LL_ENOUGH, lbase, lext, 0)) return fail("invalid literal/length code");
        if (!build(lens + 288, 32, D_BITS, dd, D_ENOUGH, dbase, dext, 1)) return fail("invalid distance code");
        state = CODES;
        return true;
    }

    // Decode into [out, out_lim); `hist` = first byte that may be referenced (out - 32768 at most).  Returns the new `out`.
    // Stops when fewer than 300 bytes of room remain, at the end of the stream (state DONE) or on error.
    uint8_t *run(uint8_t *out, uint8_t *out_lim, const uint8_t *hist)
    {
        for (;;) {
            switch (state) {
            case HEADER:
                if (in >= in_end) { state = DONE; return out; }
                if (!read_gzip_header()) return out;
                break;
            case BLOCK_HEAD:
                if (!read_block_head()) return out;
                break;
            case STORED: {
                if (stored_left) {                                     // (the bit buffer is empty: read_block_head handed its bytes back)
                    if (out >= out_lim) return out;
                    size_t n = stored_left;
                    if (n > (size_t)(in_end - in)) return fail("truncated stored block"), out;
                    if (n > (size_t)(out_lim - out)) n = (size_t)(out_lim - out);
                    memcpy(out, in, n); out += n; in += n; stored_left -= (uint32_t)n; member_out += n;
                    if (stored_left) return out;
                }
                state = last_block ? TRAILER : BLOCK_HEAD;
                break;
            }
            case CODES: {
                uint8_t *const o0 = out;
                bool room = true;
                for (;;) {
                    if (out_lim - out < 300) { room = false; break; }
                    const bool full = refill();
                    uint32_t e = ll[peek(LL_BITS)];
                    if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; drop(r); e = ll[(e >> 16) + peek(sb)]; }
                    if (!full) {                                               // the last few bytes of the input: one symbol per refill, checked
                        if (bitcnt < (int)(e & 0x1F)) { fail("truncated deflate stream"); goto out_codes; }
                        if (e & F_LIT) { drop(e & 0x1F); *out++ = (uint8_t)(e >> 16); continue; }
                    } else if (e & F_LIT) {
                        drop(e & 0x1F); *out++ = (uint8_t)(e >> 16);
                        // a second and a third literal from the same refill: >= 56 bits were valid, three codes take <= 45
                        e = ll[peek(LL_BITS)];
                        if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; drop(r); e = ll[(e >> 16) + peek(sb)]; }
                        if (e & F_LIT) {
                            drop(e & 0x1F); *out++ = (uint8_t)(e >> 16);
                            e = ll[peek(LL_BITS)];
                            if (e & F_SUB) { const int r = e & 0x1F, sb = (e >> 8) & 0x1F; drop(r); e = ll[(e >> 16) + peek(sb)]; }
                            if (e & F_LIT) { drop(e & 0x1F); *out++ = (uint8_t)(e >> 16); continue; }
                        }
                        refill();                                              // e was looked up in bits that stay where they are
                    }
                    drop(e & 0x1F);
                    if (e & F_EOB) {
                        if (bitcnt < 0) { fail("truncated deflate stream"); goto out_codes; }      // an end-of-block code read from bits past the input
                        state = last_block ? TRAILER : BLOCK_HEAD; goto out_codes;
                    }
                    if ((e >> 16) == 0) { fail("invalid literal/length symbol"); goto out_codes; }
                    {
                        const int xb = (e >> 8) & 0x1F;
                        uint32_t len = (e >> 16) + peek(xb); drop(xb);
                        if (bitcnt < 32) refill();
                        uint32_t d = dd[peek(D_BITS)];
                        if (d & F_SUB) { const int r = d & 0x1F, sb = (d >> 8) & 0x1F; drop(r); d = dd[(d >> 16) + peek(sb)]; }
                        if ((d >> 16) == 0) { fail("invalid distance symbol"); goto out_codes; }
                        drop(d & 0x1F);
                        const int dxb = (d >> 8) & 0x1F;
                        const uint32_t dist = (d >> 16) + peek(dxb); drop(dxb);
                        if (bitcnt < 0) { fail("truncated deflate stream"); goto out_codes; }
                        if ((size_t)(out - hist) < dist) { fail("distance too far back"); goto out_codes; }
                        const uint8_t *src = out - dist;
                        uint8_t *dst = out;
                        out += len;
                        if (dist >= 8) {                              // whole words; may write up to 7 bytes past `out` (room is guaranteed)
                            do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < out);
                        } else if (dist == 1) {
                            memset(dst, *src, len);
                        } else {
                            do { *dst++ = *src++; } while (dst < out);
                        }
                    }
                }
            out_codes:
                member_out += (uint64_t)(out - o0);
                if (state == ERROR) return out;
                if (!room) return out;
                break;
            }
            case TRAILER: {
                if (raw) { state = DONE; return out; }
                drop(bitcnt & 7);                                      // to the byte boundary
                // give unread whole bytes back to the input
                while (bitcnt >= 8) { --in; bitcnt -= 8; }
                bitbuf = 0; bitcnt = 0;
                if (in_end - in < 8) return fail("truncated gzip trailer"), out;
                const uint32_t isize = (uint32_t)in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24);
                if (isize != (uint32_t)member_out) return fail("gzip length check failed"), out;
                ends[n_ends].at = out;
                ends[n_ends].crc = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
                ++n_ends;
                in += 8;
                state = HEADER;                                        // next member, or the end of the input
                if (n_ends == 64) return out;                          // let the caller drain the list
                break;
            }
            case DONE:
            case ERROR:
                return out;
            }
        }
    }
};

}  // namespace shinf
