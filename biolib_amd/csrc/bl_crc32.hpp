// bl_crc32.hpp — CRC-32 (the gzip polynomial) by carry-less multiplication where the CPU has it, zlib's table walk elsewhere.
//
// The method is the folding one of Intel's white paper "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ
// Instruction" (Gopal et al., 2009), for the bit-reflected polynomial 0x1DB710641: four 128-bit accumulators take 64 bytes per
// round, each multiplied by x^(512+-) mod P and added to the next 16 bytes; then 4 -> 1, 128 -> 64 -> 32 bits (Barrett).  The
// constants are the paper's (x^N mod P for the fold distances, the polynomial and its inverse).  Checked against zlib on
// every length and offset that matter (tests/test_pgzip.py through tests/emu/pgzip_check.cpp).
#pragma once
#include <zlib.h>

#include <cstddef>
#include <cstdint>

#if defined(__x86_64__)
#include <immintrin.h>

namespace blcrc {

// `crc` and the result are the register's value (zlib's crc32 takes and returns its complement); n >= 64 and n % 16 == 0
__attribute__((target("pclmul,sse4.1"))) inline uint32_t fold(uint32_t crc, const uint8_t* p, size_t n)
{
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};  // x^(4*128+32), x^(4*128-32) mod P
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};  // x^(128+32), x^(128-32) mod P
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};  // x^64 mod P
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};  // P, floor(x^64 / P)
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
    p += 64;
    n -= 64;
    while (n >= 64) {  // 64 bytes per round
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x00));
        y6 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x20));
        y8 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        p += 64;
        n -= 64;
    }
    // four accumulators into one
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (n >= 16) {  // whole 16-byte blocks that are left
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett: 64 -> 32 bits
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

inline bool have_clmul()
{
    static const bool yes = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return yes;
}

}  // namespace blcrc
#endif

// drop-in for zlib's crc32(crc, p, n) on buffers of any size
inline uint32_t bl_crc32(uint32_t crc, const uint8_t* p, size_t n)
{
#if defined(__x86_64__)
    if (n >= 64 && blcrc::have_clmul()) {
        const size_t body = n & ~(size_t)15;
        crc = ~blcrc::fold(~crc, p, body);
        p += body;
        n -= body;
    }
#endif
    while (n) {  // (zlib takes a 32-bit length)
        const size_t step = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)step);
        p += step;
        n -= step;
    }
    return crc;
}
