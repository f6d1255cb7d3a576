// TEST INFRASTRUCTURE ONLY — not part of the product path.
//
// extern "C" shim over the UNMODIFIED reference headers, compiled from where they
// lie under /root/reference (see oracle/Makefile, target `ref`).  The output goes to
// oracle/_ref/libbiolib_ref.so, is git-ignored and is only ever loaded by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  No reference source is
// copied into this repository: this file only *calls* the reference's public
// templates:
//   wrapper::kmer_view            include/kmer_view.hpp:25-83
//   hash::hash64 / double_hash64  include/hash.hpp:11-71
//   hash::minimizer_position_extractor  include/kmer_view.hpp:250-283, src/kmer_view.cpp:7-15
//   sampler::syncmer_sampler      include/syncmer_sampler.hpp:9-137
// Build flags pin the reference's uninitialised kmer_buffer UB (kmer_view.hpp:54) to
// its intended zero-init meaning (-ftrivial-auto-var-init=zero).
#include <cstdint>
#include <cstddef>
#include <cassert>
#include <vector>
#include <optional>
#include <iterator>
#include <string>

#include "kmer_view.hpp"
#include "hash.hpp"
#include "syncmer_sampler.hpp"

extern "C" {

uint64_t ref_hash64_u64(uint64_t val, uint64_t seed) { return hash::hash64::hash(val, seed); }

void ref_double_hash64_u64(uint64_t val, uint64_t seed, uint64_t out[2])
{
    auto h = hash::double_hash64::hash(val, seed);
    out[0] = h[0];
    out[1] = h[1];
}

uint64_t ref_hash64_bytes(const uint8_t* key, uint32_t len, uint32_t seed) { return hash::hash64::hash(key, len, seed); }

uint64_t ref_hash64_u128(uint64_t lo, uint64_t hi, uint64_t seed)
{
    __uint128_t v = (static_cast<__uint128_t>(hi) << 64) | lo;
    return hash::hash64::hash(v, seed);
}

uint64_t ref_hash64_u32(uint32_t val, uint64_t seed) { return hash::hash64::hash(val, seed); }

uint64_t ref_remix(uint64_t z) { return hash::remix(z); }

// Iterate a kmer_view<uint64_t> with the reference's canonical idiom
// `for (it = cbegin(); it != cend(); ++it)`.  If `complete` is non-zero the item that
// is still readable as *it after the loop (the final k-mer the idiom omits, quirk Q1,
// SURVEY.md §8a-a3) is appended when it is a real k-mer.  Only call this on inputs in
// the reference's defined domain (length >= k, no break followed by 1..k-1 valid
// bases at the tail) — outside it the reference reads out of bounds.
// Returns the number of items; writes at most `cap`.
size_t ref_kmer_view(const char* s, size_t n, uint8_t k, int canonical, int complete,
                     uint64_t* values, uint8_t* is_null, uint64_t* positions, uint64_t* ids, size_t cap)
{
    auto view = wrapper::kmer_view_from_cstr<uint64_t>(s, n, k, canonical != 0);
    size_t cnt = 0;
    auto put = [&](wrapper::kmer_context_t<uint64_t> const& it) {
        if (cnt < cap) {
            values[cnt] = it.value ? *it.value : 0;
            is_null[cnt] = it.value ? 0 : 1;
            positions[cnt] = it.position;
            ids[cnt] = it.id;
        }
        ++cnt;
    };
    auto itr = view.cbegin();
    for (; itr != view.cend(); ++itr) put(*itr);
    if (complete) {
        auto last = *itr;
        // the post-loop state is a real k-mer iff the sequence ended inside a valid run
        if (last.value && n >= k && last.position + k == n) put(last);
    }
    return cnt;
}

// minimizer_position_extractor applied to every item the idiom yields.
size_t ref_minpos(const char* s, size_t n, uint8_t k, uint8_t m, int canonical, int complete, uint64_t* minpos, size_t cap)
{
    auto view = wrapper::kmer_view_from_cstr<uint64_t>(s, n, k, canonical != 0);
    hash::minimizer_position_extractor ex(k, m);
    size_t cnt = 0;
    auto itr = view.cbegin();
    for (; itr != view.cend(); ++itr) {
        if (cnt < cap) minpos[cnt] = ex(*itr);
        ++cnt;
    }
    if (complete) {
        auto last = *itr;
        if (last.value && n >= k && last.position + k == n) {
            if (cnt < cap) minpos[cnt] = ex(last);
            ++cnt;
        }
    }
    return cnt;
}

// Count of syncmers through the reference sampler itself (iteration only; operator*
// of the sampler does not compile for kmer_view iterators).
uint64_t ref_syncmer_count(const char* s, size_t n, uint8_t k, uint8_t m, uint16_t soff, uint16_t eoff, int canonical)
{
    using view_t = wrapper::kmer_view<uint64_t, char_iterator>;
    auto view = wrapper::kmer_view_from_cstr<uint64_t>(s, n, k, canonical != 0);
    hash::minimizer_position_extractor ex(k, m);
    sampler::syncmer_sampler<view_t::const_iterator, hash::minimizer_position_extractor> smp(view.cbegin(), view.cend(), ex, soff, eoff);
    uint64_t cnt = 0;
    for (auto it = smp.cbegin(); it != smp.cend(); ++it) ++cnt;
    return cnt;
}

// Timed legs for bench.py's cpu_baseline ("kind": "reference"): whole loops inside the
// reference's own code so that call overhead is the reference's, not ctypes'.
uint64_t ref_scan_kmers_xor(const char* s, size_t n, uint8_t k, int canonical)
{
    auto view = wrapper::kmer_view_from_cstr<uint64_t>(s, n, k, canonical != 0);
    uint64_t x = 0;
    for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
        auto item = *itr;
        if (item.value) x ^= *item.value;
    }
    return x;
}

uint64_t ref_scan_kmer_hash_xor(const char* s, size_t n, uint8_t k, int canonical, uint64_t seed)
{
    auto view = wrapper::kmer_view_from_cstr<uint64_t>(s, n, k, canonical != 0);
    uint64_t x = 0;
    for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
        auto item = *itr;
        if (item.value) x ^= hash::hash64::hash(*item.value, seed);
    }
    return x;
}

} // extern "C"
