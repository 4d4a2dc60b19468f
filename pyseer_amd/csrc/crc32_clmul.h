// crc32_clmul.h -- CRC-32 (the gzip polynomial, reflected 0xEDB88320) by carry-less multiplication.
//
// zlib 1.2.11's crc32 runs at ~1 GB/s per core; with the text decoded at the same rate the member check of a gzip / BGZF k-mer file cost as
// much host time as inflating it (the GPU boxes grant 16 CPUs: DESIGN.md section 5.5).  This is the folding scheme of Gopal et al., "Fast CRC
// Computation for Generic Polynomials Using PCLMULQDQ Instruction" (Intel, 2009) for the reflected polynomial: four 128-bit lanes folded
// across 64-byte blocks with x^(4*128 +- 32) mod P, folded together with x^(128 +- 32) mod P, reduced 128 -> 64 -> 32 bits with x^64 mod P
// and a Barrett step.  Constants are the paper's for this polynomial.  The fold body follows the widely published PCLMULQDQ routine for this
// polynomial as it appears in Chromium's zlib (third_party/zlib/crc32_simd.c, crc32_sse42_simd_: the same x0..x8 / y5..y8 register
// schedule and the same k1k2 / k3k4 / k5k0 / poly constants) -- third-party code this header is modelled on, not part of the reference
// (pyseer reads gzip through Python's gzip module, input.py:271-276).  Same value as zlib's crc32 for every input
// (tests/test_reader_cpu.py::test_clmul_crc32_equals_zlib); zlib's own routine where the CPU lacks PCLMULQDQ and for the last < 16 bytes.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <immintrin.h>
#include <zlib.h>

namespace shcrc {

__attribute__((target("pclmul,sse4.1"))) static inline uint32_t fold(const uint8_t *buf, size_t len, uint32_t state)
{   // len >= 64, a multiple of 16; state = the running register (zlib's value complemented); returns the register after buf
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);                        // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                               // whole 16-byte blocks that are left
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                           // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);                        // Barrett reduction 64 -> 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

static inline bool have_clmul() { static const bool h = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return h; }

// zlib's crc32(crc, p, n) for any n
static inline uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    if (n >= 64 && have_clmul()) {
        const size_t m = n & ~(size_t)15;
        crc = ~fold(p, m, ~crc);
        p += m; n -= m;
    }
    while (n > 0) { const size_t c = n > (1u << 30) ? (1u << 30) : n; crc = (uint32_t)::crc32(crc, p, (uInt)c); p += c; n -= c; }
    return crc;
}

}  // namespace shcrc
