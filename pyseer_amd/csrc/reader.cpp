// reader.cpp -- native k-mer text reader / packer: gzip or plain "KMER | sample:count sample:count ..."
// lines -> packed presence rows in phenotype order, the input format of sh_lmm_batch / sh_glm_batch.
//
// Replaces the per-line Python of pyseer/input.py:301-454 (read_variant, k-mer branch) for the GPU feed:
//   var_name = first whitespace token;  strains = segment between the first and the second '|', whitespace-split,
//   each token cut at ':' (input.py:377-388);  presence over the phenotyped samples in phenotype order (input.py:438-452).
// A producer thread decodes the file a few slabs ahead (inflate_fast.h; BGZF member-parallel), lines of a block are parsed in parallel (OpenMP).
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
#include <emmintrin.h>
#include "route.h"
#include "host_pool.h"
#include "inflate_par.h"
#include "crc32_clmul.h"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <functional>
#include <memory>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "../../include/seerhip.h"
#include "inflate_fast.h"

// open-addressing table keyed by the sample name bytes (no per-token allocation; FNV-1a)
struct NameTable {
    std::vector<int32_t> slot;           // index into names, or -1
    std::vector<std::string> names;
    uint32_t mask = 0;
    // eight bytes per multiplication (a byte-wise FNV-1a was 15 dependent multiplications for "sample_00012:1", a quarter of the parser's time)
    static inline uint32_t hash(const char *s, size_t n)
    {
        uint64_t h = (uint64_t)n * 0x9E3779B97F4A7C15ull;
        while (n >= 8) { uint64_t w; memcpy(&w, s, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; s += 8; n -= 8; }
        if (n) { uint64_t w = 0; memcpy(&w, s, n); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }      // (little-endian: the bytes that are there, zeros above)
        return (uint32_t)(h ^ (h >> 29));
    }
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
        build_keys();
    }
    // names of up to 16 bytes (the usual case) a second time as zero-padded 16-byte keys: one vector compare instead of a memcmp call.
    // `wide`: 16 bytes may be read at s (the caller knows its line does not end before)
    std::vector<uint8_t> key16; std::vector<uint8_t> len16;          // len16[i] = 0: longer than 16, compare the string
    void build_keys() {
        key16.assign(names.size() * 16, 0); len16.assign(names.size(), 0);
        for (size_t i = 0; i < names.size(); ++i) if (names[i].size() <= 16 && !names[i].empty()) { memcpy(&key16[i * 16], names[i].data(), names[i].size()); len16[i] = (uint8_t)names[i].size(); }
    }
    inline int find16(const char *s, size_t n) const {               // n <= 16, s[0 .. 16) readable
        static const uint8_t ones[32] = {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const __m128i x = _mm_and_si128(_mm_loadu_si128((const __m128i *)s), _mm_loadu_si128((const __m128i *)(ones + 16 - n)));
        uint64_t w0, w1; memcpy(&w0, &x, 8); memcpy(&w1, (const char *)&x + 8, 8);
        uint64_t h = (uint64_t)n * 0x9E3779B97F4A7C15ull;
        h = (h ^ w0) * 0xFF51AFD7ED558CCDull; h ^= h >> 32;
        if (n > 8) { h = (h ^ w1) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
        uint32_t hh = (uint32_t)(h ^ (h >> 29)) & mask;
        for (;;) {
            const int32_t i = slot[hh];
            if (i < 0) return -1;
            if (len16[i] == n && _mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_loadu_si128((const __m128i *)&key16[(size_t)i * 16]))) == 0xFFFF) return i;
            hh = (hh + 1) & mask;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Text source: the (memory-mapped) file decoded by a producer thread into slabs, a few slabs ahead of the parser.
//   plain text      -> slabs are windows of the mapping itself (no copy)
//   gzip            -> inflate_fast.h on one thread (the format is sequential), ~4x zlib on k-mer text, straight into the slab
//   BGZF (bgzip)    -> members are independent and announce their compressed size: a slab's worth of members is decoded by a pool of
//                      workers, each member by one of them (raw deflate + its own CRC)
// Every slab is preceded, in the same allocation, by the tail of the text before it (PAD bytes: the decoder's 32 KB history lives there
// anyway), so a line that straddles two slabs is contiguous in the second one and the parser never copies text.  The CRC-32 of every
// gzip member is verified by the consumer's worker pool over whole slabs (zlib's crc32 + crc32_combine), not by the decoding thread.
// SEERHIP_ROUTE reader=zlib selects zlib's gzread instead (A/B, and a fallback should a stream ever disagree).
// ---------------------------------------------------------------------------------------------------------------------------------
// A small pool of SLEEPING workers (condition variables) instead of an OpenMP team: libgomp's workers spin after every parallel region,
// and on the 256-thread GPU host 128-256 spinning threads starve the decoding thread (measured: gzip 10.5 k k-mers/s with 128 OpenMP
// threads, 27.5 k with 32, plain text 48 k vs 226 k).  run(n, fn) calls fn(i) for i in [0, n) with dynamic scheduling and returns when done.
using shost::ParPool;          // (host_pool.h: the fixed-size fork-join pool; its size comes from the process-wide CPU budget)

struct MemberEndAt { size_t at; uint32_t crc; };
struct Slab {
    const char *data = nullptr; size_t len = 0, pad = 0;       // data[-pad .. len): pad bytes of the preceding text, then this slab's
    char *mem = nullptr;                                        // owned buffer to recycle (nullptr: a window of the mapping)
    bool last = false; std::string err;
    std::vector<MemberEndAt> ends;                              // gzip members that end inside this slab: offset, CRC-32 of the trailer
    std::vector<uint32_t> nl; bool indexed = false;             // offsets of the newlines in data[0 .. len) (index_newlines, on first use)
};

struct sh_reader {
    int fd = -1; const uint8_t *map = nullptr; size_t map_len = 0;
    gzFile gz = nullptr;                 // SEERHIP_ROUTE reader=zlib
    std::vector<char> zbuf;              //   its text (unconsumed part), as in round 1
    size_t zpos = 0;
    int n = 0;
    NameTable index;
    bool eof = false;
    int64_t names_needed = 0;            // bytes of variant names the last refused call (-2) would have written
    // producer
    std::thread producer;
    std::mutex mu; std::condition_variable cv_put, cv_get, cv_free;
    std::deque<Slab> queue; std::vector<char *> free_bufs; std::vector<char *> all_bufs; bool stop = false;
    size_t slab_bytes = 16u << 20, pad_bytes = 1u << 20;
    size_t depth = 10;                   // slabs decoded ahead of the parser: a 4 096-line block at N = 5000 is 9 slabs of text, and with 3 the decoder stood
                                         // still while the block before it was parsed (gzip 122 k k-mers/s, decode and parse taking turns)
    std::atomic<int64_t> par_accepted{0}; // chunks accepted from a searched block head (sh_reader_par_chunks)
    size_t par_chunk = 4u << 20;         // most compressed bytes per region of the parallel gzip decoder (inflate_par.h)
    int par_workers = 0;                 //   its decoding threads
    double par_target = 12.0e6;          //   text bytes a region is sized for (the producer hands a region's text on in one fork-join of its helpers:
                                         //   3 MB regions spent the whole run there, profiles/r04/bench_reader_sweep.jsonl)
    int mode = 0;                        // 0 plain, 1 gzip, 2 BGZF
    std::unique_ptr<ParPool> pool, pool_bgzf;   // parser / CRC workers; member-parallel BGZF decoding (the producer's)
    // consumer
    std::deque<Slab> held; size_t cur = 0; const char *ptr = nullptr;     // slabs in hand, the one being read, next unread byte
    std::vector<char> bridge;            // a line longer than the pad that straddles two slabs (copied, rare)
    uint32_t crc_run = 0;
    ~sh_reader() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_put.notify_all(); cv_get.notify_all(); cv_free.notify_all();
        if (producer.joinable()) producer.join();
        for (char *b : all_bufs) free(b);
        if (map && map_len) munmap((void *)map, map_len);
        if (fd >= 0) close(fd);
        if (gz) gzclose(gz);
    }
};

static thread_local std::string g_rerr;

static bool put_slab(sh_reader *r, Slab &&sl)
{
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv_put.wait(lk, [&] { return r->stop || r->queue.size() < r->depth; });
    if (r->stop) return false;
    r->queue.push_back(std::move(sl));
    lk.unlock(); r->cv_get.notify_one();
    return true;
}

static char *get_buf(sh_reader *r)                                // nullptr when the reader is being closed
{
    std::unique_lock<std::mutex> lk(r->mu);
    // Grow on demand, without a cap: sh_reader_next keeps every slab of the block it is cutting until it returns, so a block whose text spans
    // more slabs than any fixed number of buffers would leave the producer waiting for a buffer and the consumer waiting for a slab (round 2
    // capped at 64: --block_size 65536 at N = 5000, ~2 GB of text per block, hung).  What bounds the memory is the block the caller asked
    // for plus the look-ahead of `depth` slabs (put_slab), not this function.
    if (r->free_bufs.empty()) {
        char *b = (char *)malloc(r->pad_bytes + r->slab_bytes + 1024);
        if (b) { r->all_bufs.push_back(b); return b; }
    }
    r->cv_free.wait(lk, [&] { return r->stop || !r->free_bufs.empty(); });
    if (r->stop) return nullptr;
    char *b = r->free_bufs.back(); r->free_bufs.pop_back();
    return b;
}

static void release_slab(sh_reader *r, Slab &sl)
{
    if (!sl.mem) return;
    { std::lock_guard<std::mutex> lk(r->mu); r->free_bufs.push_back(sl.mem); }
    sl.mem = nullptr;
    r->cv_free.notify_one();
}

// CRC-32 of [p, p + n) by the worker pool: 1 MB pieces, combined in order
static uint32_t crc_parallel(ParPool *pool, uint32_t crc, const uint8_t *p, size_t n)
{
    const size_t CH = 1u << 20;
    if (n <= 2 * CH || !pool) return shcrc::crc32(crc, p, n);
    const int64_t nch = (int64_t)((n + CH - 1) / CH);
    std::vector<uint32_t> part((size_t)nch);
    pool->run(nch, 1, [&](int64_t i) {
        const size_t lo = (size_t)i * CH, len = std::min(CH, n - lo);
        part[(size_t)i] = shcrc::crc32(0u, p + lo, len);
    });
    for (int64_t i = 0; i < nch; ++i) {
        const size_t lo = (size_t)i * CH, len = std::min(CH, n - lo);
        crc = (uint32_t)crc32_combine(crc, part[(size_t)i], (z_off_t)len);
    }
    return crc;
}

static void produce_plain(sh_reader *r)
{
    size_t off = 0;
    for (;;) {
        Slab sl;
        sl.len = std::min(r->slab_bytes, r->map_len - off);
        sl.data = (const char *)r->map + off; sl.pad = std::min(r->pad_bytes, off);
        off += sl.len; sl.last = off >= r->map_len;
        const bool last = sl.last;
        if (!put_slab(r, std::move(sl)) || last) return;
    }
}

// tail <- the last (at most cap) bytes of (tail + data)
static void keep_tail(std::vector<char> &tail, size_t cap, const char *data, size_t len)
{
    if (len >= cap) { tail.assign(data + len - cap, data + len); return; }
    const size_t keep = std::min(tail.size(), cap - len);
    tail.erase(tail.begin(), tail.end() - keep);
    tail.insert(tail.end(), data, data + len);
}

static void produce_gzip(sh_reader *r)
{
    const size_t PAD = r->pad_bytes;                               // >= 32 KB: also the decoder's history
    shinf::Decoder d; d.begin(r->map, r->map + r->map_len);
    std::vector<char> tail;
    for (;;) {
        char *buf = get_buf(r);
        if (!buf) return;
        memcpy(buf + PAD - tail.size(), tail.data(), tail.size());
        uint8_t *const start = (uint8_t *)buf + PAD, *out = start, *const lim = start + r->slab_bytes;
        const uint8_t *hist = start - tail.size();
        Slab sl; sl.mem = buf; sl.data = buf + PAD; sl.pad = tail.size();
        for (;;) {
            out = d.run(out, lim, hist);
            for (int i = 0; i < d.n_ends; ++i) sl.ends.push_back(MemberEndAt{(size_t)(d.ends[i].at - start), d.ends[i].crc});
            d.n_ends = 0;
            if (d.state == shinf::Decoder::DONE || d.state == shinf::Decoder::ERROR) break;
            if ((size_t)(lim - out) < 300) break;
        }
        if (d.state == shinf::Decoder::ERROR) sl.err = std::string("gzip: ") + (d.err ? d.err : "error");
        sl.len = (size_t)(out - start);
        sl.last = d.state == shinf::Decoder::DONE || !sl.err.empty();
        keep_tail(tail, PAD, sl.data, sl.len);
        const bool last = sl.last;
        if (!put_slab(r, std::move(sl)) || last) return;
    }
}

// One gzip member on several threads (inflate_par.h), as a continuous pipeline.  The compressed bytes are cut into regions on a grid that is
// fixed as it is handed out; `W` decoding threads take regions in order: find the first block head in the region, decode from it up to the
// first block head at or after the region's end, into 16-bit symbols with window markers.  The producer thread is the ACCEPTOR: it walks the
// regions in order knowing the exact bit position the stream has reached; a region whose decode started on that very bit is accepted, any
// other (no head found, a false head, a member boundary inside) is decoded again by the acceptor itself from the exact position -- so a wrong
// guess costs time, never bytes.  Accepted symbols are translated against the 32 KB of text before them (kept by the acceptor) into slab
// buffers by a few helper threads and handed to the parser; decoding runs ahead by at most 2 W regions.
static void produce_gzip_parallel(sh_reader *r)
{
    using namespace shinf;
    const uint8_t *const base = r->map, *const end = r->map + r->map_len;
    const size_t LEN = r->map_len, PAD = r->pad_bytes, SB = r->slab_bytes, CHMAX = r->par_chunk, CHMIN = std::min<size_t>(65536, r->par_chunk);
    const int W = std::max(1, r->par_workers), INFL = 2 * W;
    ParPool *tpool = r->pool_bgzf.get();                              // the acceptor's translation helpers
    struct Task { ParChunk c; uint64_t from_bit = 0, to_bit = 0; int64_t id = -1; bool done = false; };
    std::vector<Task> slots((size_t)INFL);
    std::mutex mu; std::condition_variable cv_w, cv_a;
    int64_t next_id = 0, consumed = 0;                                // regions handed out / regions the acceptor is through with
    size_t next_from = 0, ch_cur = std::min<size_t>(CHMAX, std::max<size_t>(CHMIN, 1u << 18));
    int infl_cur = INFL;                                              // regions in flight now: fewer when the text of a region is large (ratio)
    bool exhausted = LEN == 0, quit = false;
    // where the time goes (SEERHIP_DEBUG=host: one line on stderr when the stream ends), seconds summed over the threads of a kind
    const bool dbg = sh_debug("host");
    std::atomic<int64_t> ns_wslot{0}, ns_find{0}, ns_dec{0}, n_redo{0}, n_acc{0};
    std::atomic<bool> give_up{false};                                 // the searched heads keep being wrong (stored / fixed blocks, an alphabet the text test does not know):
                                                                      // the searching threads stop searching and the producer decodes alone, as produce_gzip does
    int64_t ns_await = 0, ns_redo = 0, ns_trans = 0, ns_put = 0;
    uint32_t vhist = 0; int nhist = 0, serial_since = 0;               // (the producer's own: verdicts of the last regions, 1 = decoded again)
    auto now_ns = [] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int64_t t_begin = now_ns();
    std::vector<std::thread> workers;
    for (int w = 0; w < W; ++w) workers.emplace_back([&] {
        std::unique_ptr<Decoder> dec(new Decoder);
        for (;;) {
            Task *t;
            const int64_t tw0 = dbg ? now_ns() : 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_w.wait(lk, [&] { return quit || exhausted || next_id < consumed + infl_cur; });
                if (quit || exhausted) return;
                t = &slots[(size_t)(next_id % INFL)];
                t->id = next_id++; t->done = false;
                size_t to = next_from + ch_cur;
                if (to >= LEN || LEN - to < ch_cur / 2 + 4096) { to = LEN; exhausted = true; }
                t->from_bit = (uint64_t)next_from * 8; t->to_bit = to == LEN ? ~0ull : (uint64_t)to * 8;
                next_from = to;
            }
            ParChunk &c = t->c;
            if (c.sym.size() < PAR_WIN + 65536) c.sym.resize(PAR_WIN + 65536);
            for (uint32_t k = 0; k < PAR_WIN; ++k) c.sym[k] = (uint16_t)(0x8000u | k);
            c.ok = false; c.hit_final = false; c.err = nullptr; c.n = 0; c.exact = false;
            const int64_t tf0 = dbg ? now_ns() : 0;
            const uint64_t region_bytes = ((t->to_bit == ~0ull ? (uint64_t)LEN * 8 : t->to_bit) - t->from_bit) / 8;
            c.max_syms = (size_t)std::max<uint64_t>(16u << 20, 64 * region_bytes);
            const uint64_t start = give_up.load(std::memory_order_relaxed) ? ~0ull
                                 : par_find_block(base, end, t->from_bit, t->to_bit == ~0ull ? (uint64_t)LEN * 8 : t->to_bit, *dec, c);
            const int64_t tf1 = dbg ? now_ns() : 0;
            c.ok = false; c.hit_final = false; c.err = nullptr; c.n = 0;
            if (start != ~0ull) {
                c.start_bit = start; c.stop_bit = t->to_bit;
                try { par_decode(*dec, base, end, c); } catch (const std::bad_alloc &) { c.ok = false; c.err = "out of memory"; }   // (the producer decides: it decodes the region again)
            }
            if (dbg) { const int64_t tf2 = now_ns(); ns_wslot += tf0 - tw0; ns_find += tf1 - tf0; ns_dec += tf2 - tf1; }
            { std::lock_guard<std::mutex> lk(mu); t->done = true; }
            cv_a.notify_all();
        }
    });
    struct Joiner { std::mutex &mu; bool &quit; std::condition_variable &cv; std::vector<std::thread> &th;
                    ~Joiner() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); for (auto &t : th) t.join(); } } joiner{mu, quit, cv_w, workers};

    std::vector<char> tail;                                          // the last PAD bytes of text handed on (slab pads)
    std::vector<uint8_t> hist(PAR_WIN, 0), hist2(PAR_WIN, 0);        // the 32 KB of text before the next chunk, right-aligned; `avail` of them real
    uint32_t avail = 0;
    Slab cur; size_t fill = 0; bool have = false;                    // the slab being filled
    auto flush = [&](bool last, const std::string &err) -> bool {    // hands `cur` on (an empty one if there is none and something must be said)
        if (!have) { char *buf = get_buf(r); if (!buf) return false; cur = Slab(); cur.mem = buf; cur.data = buf + PAD; fill = 0; }
        have = false;
        cur.len = fill;
        memcpy(cur.mem + PAD - tail.size(), tail.data(), tail.size());
        cur.pad = tail.size();
        keep_tail(tail, PAD, cur.data, cur.len);
        cur.err = err; cur.last = last || !err.empty();
        return put_slab(r, std::move(cur));
    };
    Decoder hd; hd.begin(base, end);
    std::unique_ptr<Decoder> redo_dec(new Decoder);
    ParChunk redo;
    uint64_t member_out = 0, pos_bit = 0;
    bool in_member = false;
    int64_t j = 0;                                                   // the region the exact position lies in (or an earlier one, not yet let go)
    struct Piece { size_t off, cnt; uint8_t *dst; };
    std::vector<Piece> pieces;
    for (;;) {
        if (!in_member) {
            hd.state = Decoder::HEADER;
            if (!hd.read_gzip_header()) {
                if (dbg) {
                    const double wall = (double)(now_ns() - t_begin) * 1e-9;
                    fprintf(stderr, "[reader debug] parallel gzip: wall %.3f s, %d decoding threads: waiting for a slot %.3f, searching %.3f, decoding %.3f s (summed); "
                                    "acceptor: waiting for a region %.3f, decoding again %.3f (%lld of %lld regions), translating + handing on %.3f (of it blocked on the parser %.3f) s\n",
                            wall, W, (double)ns_wslot * 1e-9, (double)ns_find * 1e-9, (double)ns_dec * 1e-9, (double)ns_await * 1e-9, (double)ns_redo * 1e-9,
                            (long long)n_redo.load(), (long long)(n_redo.load() + n_acc.load()), (double)ns_trans * 1e-9, (double)ns_put * 1e-9);
                }
                if (hd.state == Decoder::DONE) flush(true, std::string());
                else flush(true, std::string("gzip: ") + (hd.err ? hd.err : "error"));
                return;
            }
            in_member = true; member_out = 0; avail = 0; pos_bit = (uint64_t)(hd.in - base) * 8;
        }
        // ---- the region the position lies in --------------------------------------------------------------------------------------------
        Task *t = nullptr;
        const int64_t ta0 = dbg ? now_ns() : 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            if (exhausted && j >= next_id) break;                      // (cannot happen: the last region runs to the end of the file)
            t = &slots[(size_t)(j % INFL)];
            cv_a.wait(lk, [&] { return t->id == j && t->done; });
            if (t->to_bit != ~0ull && t->to_bit <= pos_bit) { ++j; consumed = j; lk.unlock(); cv_w.notify_all(); t = nullptr; continue; }
            break;
        }
        if (!t) { flush(true, "gzip: internal error (no region for the position)"); return; }
        ParChunk *c = &t->c;
        const int64_t ta1 = dbg ? now_ns() : 0;
        ns_await += ta1 - ta0;
        // the verdicts of the last 16 regions decide whether searching is worth its threads (a sliding window: an atypical head of the file
        // must not turn the search off for the rest of it -- ADVICE r05); turned off, it is tried again after 32 regions decoded alone
        const bool accepted = c->ok && c->start_bit == pos_bit;
        if (give_up.load(std::memory_order_relaxed)) {
            if (++serial_since >= 32) { give_up.store(false); vhist = 0; nhist = 0; serial_since = 0; }
        } else {
            vhist = ((vhist << 1) | (accepted ? 0u : 1u)) & 0xFFFFu; nhist = std::min(nhist + 1, 16);
            if (nhist >= 8 && __builtin_popcount(vhist & ((1u << nhist) - 1u)) * 4 > nhist * 3) { give_up.store(true); serial_since = 0; }
        }
        if (accepted) { ++r->par_accepted; ++n_acc; }
        else {
            ++n_redo;
            redo.start_bit = pos_bit; redo.stop_bit = t->to_bit;
            if (redo.sym.size() < PAR_WIN + 65536) redo.sym.resize(PAR_WIN + 65536);
            for (uint32_t k = 0; k < PAR_WIN; ++k) redo.sym[k] = (uint16_t)(0x8000u | k);
            redo.ok = false; redo.hit_final = false; redo.err = nullptr; redo.n = 0;
            try { par_decode(*redo_dec, base, end, redo); } catch (const std::bad_alloc &) { redo.ok = false; redo.err = "out of memory decoding a region"; }
            if (!redo.ok) { flush(true, std::string("gzip: ") + (redo.err ? redo.err : "error")); return; }
            c = &redo;
        }
        const int64_t ta2 = dbg ? now_ns() : 0;
        ns_redo += ta2 - ta1;
        // ---- its text: the next window first (in order, 32 K symbols), then everything into slabs by the helpers ---------------------------
        const uint16_t *sym = c->sym.data() + PAR_WIN;
        bool refs_ok = true;
        if (c->n >= PAR_WIN) refs_ok &= par_translate(sym + c->n - PAR_WIN, PAR_WIN, hist.data(), avail, hist2.data());
        else {
            memcpy(hist2.data(), hist.data() + c->n, PAR_WIN - c->n);
            refs_ok &= par_translate(sym, c->n, hist.data(), avail, hist2.data() + PAR_WIN - c->n);
        }
        std::string err;
        size_t off = 0;
        while (off < c->n) {
            if (!have) { char *buf = get_buf(r); if (!buf) return; cur = Slab(); cur.mem = buf; cur.data = buf + PAD; fill = 0; have = true; }
            const size_t cnt = std::min(c->n - off, SB - fill);
            pieces.clear();
            const size_t PIECE = 1u << 18;
            for (size_t o = 0; o < cnt; o += PIECE) pieces.push_back(Piece{off + o, std::min(PIECE, cnt - o), (uint8_t *)cur.mem + PAD + fill + o});
            std::atomic<int> bad{0};
            tpool->run((int64_t)pieces.size(), 1, [&](int64_t i) {
                const Piece &pc = pieces[(size_t)i];
                if (!par_translate(sym + pc.off, pc.cnt, hist.data(), avail, pc.dst)) ++bad;
            });
            if (bad) refs_ok = false;
            fill += cnt; off += cnt;
            if (fill == SB && !refs_ok) break;
            if (fill == SB) { const int64_t tp0 = dbg ? now_ns() : 0; const bool okp = flush(false, std::string()); if (dbg) ns_put += now_ns() - tp0; if (!okp) return; }
        }
        if (dbg) ns_trans += now_ns() - ta2;
        if (!refs_ok) { flush(true, "gzip: distance too far back"); return; }
        hist.swap(hist2);
        avail = (uint32_t)std::min<uint64_t>(PAR_WIN, (uint64_t)avail + c->n);
        member_out += c->n;
        pos_bit = c->end_bit;
        if (c->n > 0 && c->end_bit > c->start_bit) {                  // the next regions sized for ~3 MB of text each
            const double ratio = (double)c->n / ((double)(c->end_bit - c->start_bit) / 8.0);
            const size_t want = (size_t)std::min<double>((double)CHMAX, std::max<double>((double)CHMIN, r->par_target / std::max(1.0, ratio)));
            // at most ~768 MB of symbols in flight (a text that compresses 1000 : 1 would otherwise hold 64 regions of 128 MB)
            const double per = std::max(1.0, ratio) * (double)want * 2.0;
            const int infl = (int)std::min<double>((double)INFL, std::max<double>(2.0, 768.0e6 / (double)shost::host_streams() / per));
            { std::lock_guard<std::mutex> lk(mu); ch_cur = want; infl_cur = infl; }
            cv_w.notify_all();
        }
        if (c->hit_final) {
            // the trailer: CRC-32 (checked by the consumer over whole slabs), ISIZE; then the next member's header, or the end
            const uint8_t *tr = base + ((c->end_bit + 7) >> 3);
            if (end - tr < 8) { flush(true, "gzip: truncated gzip trailer"); return; }
            const uint32_t crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
            const uint32_t isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
            if (isize != (uint32_t)member_out) { flush(true, "gzip: gzip length check failed"); return; }
            if (!have) { char *buf = get_buf(r); if (!buf) return; cur = Slab(); cur.mem = buf; cur.data = buf + PAD; fill = 0; have = true; }
            cur.ends.push_back(MemberEndAt{fill, crc});
            hd.in = tr + 8; hd.bitbuf = 0; hd.bitcnt = 0; in_member = false;
        }
    }
}

// BGZF member at p: total compressed size, or 0 if p is not a BGZF member header
static size_t bgzf_member(const uint8_t *p, const uint8_t *end)
{
    if (end - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const int xlen = p[10] | (p[11] << 8);
    const uint8_t *x = p + 12, *xe = x + xlen;
    if (xe > end) return 0;
    while (x + 4 <= xe) {
        const int slen = x[2] | (x[3] << 8);
        if (x[0] == 'B' && x[1] == 'C' && slen == 2 && x + 6 <= xe) return (size_t)(x[4] | (x[5] << 8)) + 1;
        x += 4 + slen;
    }
    return 0;
}

static void produce_bgzf(sh_reader *r)
{
    const size_t PAD = r->pad_bytes;
    const uint8_t *p = r->map, *end = r->map + r->map_len;
    std::vector<char> tail;
    for (;;) {
        struct Mem { const uint8_t *cdata; size_t clen; uint32_t isize, crc; size_t off; };
        std::vector<Mem> mem;
        size_t total = 0;
        std::string err;
        while (p < end && total + 65536 <= r->slab_bytes) {
            const size_t bs = bgzf_member(p, end);
            if (bs < 26 || p + bs > end) { err = "BGZF: bad member header"; break; }
            const int xlen = p[10] | (p[11] << 8);
            const uint8_t *cd = p + 12 + xlen, *tr = p + bs - 8;
            Mem m{cd, (size_t)(tr - cd), (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24),
                  (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24), total};
            if (m.isize > 65536) { err = "BGZF: member larger than 64 KB"; break; }
            total += m.isize; mem.push_back(m); p += bs;
        }
        char *buf = get_buf(r);
        if (!buf) return;
        memcpy(buf + PAD - tail.size(), tail.data(), tail.size());
        Slab sl; sl.mem = buf; sl.data = buf + PAD; sl.pad = tail.size(); sl.len = total;
        std::atomic<int> bad{0};
        r->pool_bgzf->run((int64_t)mem.size(), 4, [&](int64_t i) {
            static thread_local std::vector<uint8_t> tmp(65536 + 512);
            static thread_local shinf::Decoder d;
            const Mem &m = mem[(size_t)i];
            d.begin(m.cdata, m.cdata + m.clen, true);
            uint8_t *o = d.run(tmp.data(), tmp.data() + tmp.size(), tmp.data());
            if (d.state != shinf::Decoder::DONE || (size_t)(o - tmp.data()) != m.isize || shcrc::crc32(0u, tmp.data(), m.isize) != m.crc) ++bad;
            else memcpy(buf + PAD + m.off, tmp.data(), m.isize);
        });
        if (bad && err.empty()) err = "BGZF: a member failed to decode or its CRC-32 check";
        sl.err = err; sl.last = p >= end || !err.empty();
        keep_tail(tail, PAD, sl.data, sl.len);
        const bool last = sl.last;
        if (!put_slab(r, std::move(sl)) || last) return;
    }
}

// Parser / CRC workers per reader: the CPUs this process may really use -- hardware threads cut by the cgroup CPU quota (a GPU box shows 256
// CPUs under a quota of 16) -- shared between the readers that run at once (sh_reader_set_concurrency: `--kmers a.gz b.gz ...` opens one
// reader per file, and 8 x 47 workers on 16 CPUs was the oversubscription the writer and the bench already avoid), at most 48, at least 2.
static int reader_threads()
{
    return std::max(2, std::min(48, shost::per_stream_cpus(2)));
}

extern "C" {

const char *sh_reader_error(void) { return g_rerr.c_str(); }

void sh_reader_set_concurrency(int n_readers) { shost::streams().store(n_readers < 1 ? 1 : n_readers); }      // (= sh_set_host_streams)

sh_reader *sh_reader_open(const char *path, const char *const *sample_names, int n_samples)
{
    if (!path || !sample_names || n_samples < 1) { g_rerr = "bad argument"; return nullptr; }
    if (const std::string bad = sh_route_unknown(); !bad.empty()) { g_rerr = "SEERHIP_ROUTE: unknown item '" + bad + "' (keys: csrc/route.h)"; return nullptr; }
    sh_reader *r = new sh_reader();
    r->n = n_samples;
    r->index.build(sample_names, n_samples);
    int nt = reader_threads();
    if (const char *te = sh_route("reader_threads")) nt = std::max(1, std::atoi(te));
    r->pool.reset(new ParPool(nt - 1, shost::ST_READER_PARSE));
    const char *sel = sh_route("reader");
    if (sel && std::string(sel) == "zlib") {
        r->gz = gzopen(path, "rb");
        if (!r->gz) { g_rerr = std::string("cannot open ") + path; delete r; return nullptr; }
        gzbuffer(r->gz, 1 << 20);
        return r;
    }
    r->fd = open(path, O_RDONLY);
    struct stat stt;
    if (r->fd < 0 || fstat(r->fd, &stt) != 0) { g_rerr = std::string("cannot open ") + path; delete r; return nullptr; }
    r->map_len = (size_t)stt.st_size;
    if (r->map_len) {
        void *m = mmap(nullptr, r->map_len, PROT_READ, MAP_PRIVATE, r->fd, 0);
        if (m == MAP_FAILED) { g_rerr = std::string("cannot map ") + path; r->map_len = 0; delete r; return nullptr; }
        r->map = (const uint8_t *)m;
        madvise(m, r->map_len, MADV_SEQUENTIAL);
    }
    if (const char *sb = sh_route("reader_slab")) r->slab_bytes = std::min<size_t>(2047u << 20, std::max<size_t>(70000, (size_t)std::atoll(sb)));   // (newline offsets are 32-bit)
    if (const char *pb = sh_route("reader_pad")) r->pad_bytes = std::max<size_t>(32768, (size_t)std::atoll(pb));
    if (r->map_len == 0) { r->eof = true; return r; }
    r->mode = (r->map_len >= 2 && r->map[0] == 0x1f && r->map[1] == 0x8b) ? (bgzf_member(r->map, r->map + r->map_len) ? 2 : 1) : 0;
    if (r->mode == 2) r->pool_bgzf.reset(new ParPool(std::max(1, std::min(32, std::max(2, nt / 2)) - 1), shost::ST_READER_DECODE));
    // one gzip member on several threads (inflate_par.h) unless SEERHIP_ROUTE reader=serial, or there is nothing to share out
    r->depth = std::max<size_t>(3, r->depth / (size_t)shost::host_streams());       // several readers at once share the memory as they share the CPUs
    if (const char *cd = sh_route("reader_depth")) r->depth = std::max<size_t>(1, (size_t)std::atoll(cd));
    size_t par_min = 1u << 20;                                      // files below this go through the one-thread decoder
    if (const char *cb = sh_route("reader_chunk")) { r->par_chunk = std::max<size_t>(4096, (size_t)std::atoll(cb)); par_min = 2 * r->par_chunk; }
    const bool par = r->mode == 1 && nt >= 3 && !(sel && std::string(sel) == "serial") && r->map_len >= par_min;
    if (par) {
        r->par_workers = std::max(2, std::min(32, nt * 2 / 3));
        if (const char *cw = sh_route("reader_workers")) r->par_workers = std::max(1, std::atoi(cw));
        int helpers = std::max(2, std::min(8, nt / 2));
        if (const char *ch = sh_route("reader_helpers")) helpers = std::max(1, std::atoi(ch));
        if (const char *ct = sh_route("reader_target")) r->par_target = std::max(65536.0, std::atof(ct));
        r->pool_bgzf.reset(new ParPool(helpers, shost::ST_READER_DECODE));
    }
    r->producer = std::thread([r, par] {
        if (r->mode == 0) produce_plain(r); else if (r->mode == 1) { if (par) produce_gzip_parallel(r); else produce_gzip(r); } else produce_bgzf(r);
    });
    return r;
}

void sh_reader_close(sh_reader *r) { delete r; }

// the next slab from the producer, CRCs of the gzip members that end in it verified; false at the end of the text or on error
static bool next_slab(sh_reader *r, bool *failed)
{
    *failed = false;
    if (r->eof) return false;
    Slab sl;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        r->cv_get.wait(lk, [&] { return !r->queue.empty(); });
        sl = std::move(r->queue.front()); r->queue.pop_front();
    }
    r->cv_put.notify_one();
    if (r->mode == 1 && sl.err.empty()) {
        size_t p = 0;
        for (const auto &e : sl.ends) {
            r->crc_run = crc_parallel(r->pool.get(), r->crc_run, (const uint8_t *)sl.data + p, e.at - p);
            if (r->crc_run != e.crc) { sl.err = "gzip: CRC-32 check failed"; break; }
            r->crc_run = 0; p = e.at;
        }
        if (sl.err.empty()) r->crc_run = crc_parallel(r->pool.get(), r->crc_run, (const uint8_t *)sl.data + p, sl.len - p);
    }
    if (!sl.err.empty()) { g_rerr = sl.err; *failed = true; r->eof = true; release_slab(r, sl); return false; }
    if (sl.last) r->eof = true;
    r->held.push_back(std::move(sl));
    return true;
}

// The newlines of a slab, found by all parser threads at once the first time a call walks into it (one memchr loop over the 150 MB of a
// 4096-line block at N = 5000 took as long as parsing them on 48 threads: plain text 230 k -> see profiles/r04/bench_reader.json).
static void index_newlines(sh_reader *r, Slab &s)
{
    if (s.indexed) return;
    s.indexed = true;
    const size_t PIECE = 1u << 17, np = (s.len + PIECE - 1) / PIECE;
    if (np <= 1) {
        for (const char *p = s.data, *e = s.data + s.len; p < e;) { const void *q = memchr(p, '\n', (size_t)(e - p)); if (!q) break; s.nl.push_back((uint32_t)((const char *)q - s.data)); p = (const char *)q + 1; }
        return;
    }
    std::vector<std::vector<uint32_t>> part(np);
    r->pool->run((int64_t)np, 1, [&](int64_t i) {
        const char *p = s.data + (size_t)i * PIECE, *e = s.data + std::min(s.len, (size_t)(i + 1) * PIECE);
        auto &v = part[(size_t)i];
        while (p < e) { const void *q = memchr(p, '\n', (size_t)(e - p)); if (!q) break; v.push_back((uint32_t)((const char *)q - s.data)); p = (const char *)q + 1; }
    });
    size_t tot = 0; for (auto &v : part) tot += v.size();
    s.nl.reserve(tot);
    for (auto &v : part) s.nl.insert(s.nl.end(), v.begin(), v.end());
}

// Parses up to max_variants lines.  bits: max_variants * row_bytes (zeroed here); counts[v] = carriers among the phenotyped
// samples (af = counts/n, input.py:446); names: concatenated variant names, name_off[v]..name_off[v+1].
// Returns the number of variants parsed (0 at end of file), or -1 on error (sh_reader_error()).
int64_t sh_reader_next(sh_reader *r, int64_t max_variants, uint8_t *bits, int64_t row_bytes, int32_t *counts,
                       char *names, int64_t names_cap, int64_t *name_off)
{
    if (!r || !bits || !counts || !names || !name_off || max_variants < 1) { g_rerr = "bad argument"; return -1; }
    if (row_bytes * 8 < r->n) { g_rerr = "row_bytes too small"; return -1; }
    std::vector<std::pair<const char *, const char *>> lines;   // [begin, end) without the newline; they point into slabs held until the end of the call
    const char *ptr0 = nullptr;
    if (r->gz) {
        // ---- zlib path (SEERHIP_ROUTE reader=zlib): one growing buffer, as in round 1
        if (r->zpos > 0) { r->zbuf.erase(r->zbuf.begin(), r->zbuf.begin() + r->zpos); r->zpos = 0; }
        size_t scan = 0, lstart = 0;
        std::vector<std::pair<size_t, size_t>> lo;
        for (;;) {
            while ((int64_t)lo.size() < max_variants) {
                const char *base = r->zbuf.data();
                const void *nl = scan < r->zbuf.size() ? memchr(base + scan, '\n', r->zbuf.size() - scan) : nullptr;
                if (!nl) { scan = r->zbuf.size(); break; }
                const size_t e = (const char *)nl - base;
                lo.emplace_back(lstart, e); lstart = scan = e + 1;
            }
            if ((int64_t)lo.size() >= max_variants || r->eof) break;
            const size_t old = r->zbuf.size(), slab = 8u << 20;
            r->zbuf.resize(old + slab);
            const int got = gzread(r->gz, r->zbuf.data() + old, (unsigned)slab);
            if (got < 0) { int en; g_rerr = gzerror(r->gz, &en); return -1; }
            r->zbuf.resize(old + (size_t)got);
            if (got == 0) r->eof = true;
        }
        if (r->eof && lstart < r->zbuf.size() && (int64_t)lo.size() < max_variants) { lo.emplace_back(lstart, r->zbuf.size()); lstart = r->zbuf.size(); }
        for (auto &l : lo) lines.emplace_back(r->zbuf.data() + l.first, r->zbuf.data() + l.second);
        r->zpos = lstart;                                             // committed below unless the call is refused
    } else {
        // ---- slabs: drop the ones read completely by earlier calls, then walk
        while (r->cur > 0) { release_slab(r, r->held.front()); r->held.pop_front(); --r->cur; }
        bool failed = false;
        if (r->held.empty()) { if (!next_slab(r, &failed)) { if (failed) return -1; return 0; } r->cur = 0; r->ptr = r->held[0].data; }
        ptr0 = r->ptr;
        r->bridge.clear();
        const char *lstart = r->ptr;                                   // start of the line being scanned (may lie in the pad of held[cur])
        const char *scan = r->ptr;
        size_t cur_indexed = (size_t)-1, nli = 0;                      // the slab `nli` walks, the next entry of its newline index
        while ((int64_t)lines.size() < max_variants) {
            Slab &s = r->held[r->cur];
            const char *send = s.data + s.len;
            if (cur_indexed != r->cur) {
                index_newlines(r, s);
                const uint32_t from = scan > s.data ? (uint32_t)(scan - s.data) : 0u;
                nli = (size_t)(std::lower_bound(s.nl.begin(), s.nl.end(), from) - s.nl.begin());
                cur_indexed = r->cur;
            }
            if (nli < s.nl.size()) { const char *nl = s.data + s.nl[nli++]; lines.emplace_back(lstart, nl); lstart = scan = nl + 1; continue; }
            // the slab is exhausted with `part` bytes of an unfinished line
            const size_t part = (size_t)(send - lstart);
            if (r->cur + 1 >= r->held.size()) {
                if (r->eof) { if (part) { lines.emplace_back(lstart, send); lstart = send; } break; }
                if (!next_slab(r, &failed)) {
                    if (failed) return -1;
                    if (part) { lines.emplace_back(lstart, send); lstart = send; }      // the text ended without a newline
                    break;
                }
            }
            const Slab &nx = r->held[r->cur + 1];
            ++r->cur;
            if (part <= nx.pad) { lstart = nx.data - part; scan = nx.data; }            // the unfinished line is contiguous in the next slab's pad
            else {
                // longer than the pad (a unitig name of megabases): copy it out.  One such line per call at most keeps `bridge` stable.
                if (!r->bridge.empty()) { --r->cur; break; }                             // finish this call here; the next one bridges again
                const char *nsend = nx.data + nx.len;
                const void *nl2 = memchr(nx.data, '\n', nx.len);
                if (!nl2 && !(nx.last)) { g_rerr = "a line longer than a whole read slab: raise SEERHIP_ROUTE reader_slab"; return -1; }
                const char *le = nl2 ? (const char *)nl2 : nsend;
                r->bridge.assign(lstart, send); r->bridge.insert(r->bridge.end(), nx.data, le);
                lines.emplace_back(r->bridge.data(), r->bridge.data() + r->bridge.size());
                lstart = scan = nl2 ? le + 1 : le;
            }
        }
        r->ptr = lstart;
    }
    const size_t cur_after = r->cur;
    (void)cur_after;
    const int64_t nv = (int64_t)lines.size();
    if (nv == 0) return 0;
    // ---- names (serial: offsets), presence (parallel).  Unitig names run to tens of kilobases: the total is measured BEFORE anything is
    // written, and a call that does not fit is refused without consuming its lines (-2; sh_reader_names_needed() says how much to bring)
    std::vector<std::pair<const char *, const char *>> nm((size_t)nv);
    int64_t need = 0;
    for (int64_t v = 0; v < nv; ++v) {
        const char *p = lines[v].first, *e = lines[v].second;
        while (p < e && (*p == ' ' || *p == '\t')) ++p;
        const char *q = p;
        while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
        nm[v] = {p, q}; need += q - p;
    }
    if (need > names_cap) {                                           // nothing is consumed: the slabs of this call stay in hand
        r->names_needed = need; g_rerr = "names buffer too small";
        if (r->gz) r->zpos = 0; else { r->cur = 0; r->ptr = ptr0; }
        return -2;
    }
    memset(bits, 0, (size_t)nv * row_bytes);
    int64_t off = 0;
    for (int64_t v = 0; v < nv; ++v) { name_off[v] = off; memcpy(names + off, nm[v].first, nm[v].second - nm[v].first); off += nm[v].second - nm[v].first; }
    name_off[nv] = off;
    const auto &index = r->index;
    r->pool->run(nv, 8, [&](int64_t v) {
        const char *p = lines[v].first, *e = lines[v].second;
        const char *bar = (const char *)memchr(p, '|', e - p);
        uint8_t *row = bits + v * row_bytes;
        int cnt = 0;
        if (bar) {
            const char *s = bar + 1;
            const char *bar2 = (const char *)memchr(s, '|', e - s);
            const char *end = bar2 ? bar2 : e;
            const __m128i c_sp = _mm_set1_epi8(' '), c_tb = _mm_set1_epi8('\t'), c_cr = _mm_set1_epi8('\r'), c_co = _mm_set1_epi8(':');
            while (s < end) {
                while (s < end && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;
                // the token's end and its first colon, 16 bytes at a time ("sample_00012:1 " is one step)
                const char *t = s, *colon = nullptr;
                const bool wide = s + 16 <= end;
                for (;;) {
                    if (t + 16 <= end) {
                        const __m128i x = _mm_loadu_si128((const __m128i *)t);
                        const unsigned mw = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, c_sp), _mm_cmpeq_epi8(x, c_tb)), _mm_cmpeq_epi8(x, c_cr)));
                        const unsigned mc = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, c_co));
                        if (!colon && mc) { const int pc = __builtin_ctz(mc); if (!mw || pc < __builtin_ctz(mw)) colon = t + pc; }
                        if (mw) { t += __builtin_ctz(mw); break; }
                        t += 16;
                    } else {
                        while (t < end && *t != ' ' && *t != '\t' && *t != '\r') { if (*t == ':' && !colon) colon = t; ++t; }
                        break;
                    }
                }
                if (t > s) {
                    const size_t nlen = (size_t)((colon ? colon : t) - s);
                    const int i = (wide && nlen <= 16 && nlen > 0) ? index.find16(s, nlen) : index.find(s, nlen);
                    if (i >= 0) {
                        if (!((row[i >> 3] >> (i & 7)) & 1)) { row[i >> 3] |= (uint8_t)(1u << (i & 7)); ++cnt; }
                    }
                }
                s = t;
            }
        }
        counts[v] = cnt;
    });
    return nv;
}

uint32_t shk_crc32(uint32_t crc, const uint8_t *p, int64_t n) { return shcrc::crc32(crc, p, (size_t)n); }   // (internal: tests/test_reader_cpu.py)
int64_t sh_reader_names_needed(sh_reader *r) { return r ? r->names_needed : 0; }
int64_t sh_reader_par_chunks(sh_reader *r) { return r ? r->par_accepted.load() : 0; }
int64_t sh_reader_buffered(sh_reader *r)
{
    if (!r) return 0;
    if (r->gz) return (int64_t)r->zbuf.size();
    std::lock_guard<std::mutex> lk(r->mu);
    return (int64_t)(r->all_bufs.size() * (r->pad_bytes + r->slab_bytes));     // every slab buffer ever allocated (they are recycled)
}

}  // extern "C"
