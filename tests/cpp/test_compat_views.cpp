// GPU test of the drop-in C++ headers (include/compat/): written the way the reference's own test
// drivers use the views (tests/test_kmer_view.cpp:35-41, tests/test_minimizer_view.cpp:37-43,
// tests/test_super_kmer_view.cpp:29-34 in the reference tree), but with expected values: the
// reference-generated tiny-string vectors of SURVEY.md §8c and the CPU oracle on seeded inputs.
// Needs an MI355X: run through tests/test_gpu_cpp_compat.py (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <optional>
#include <string>
#include <vector>

#include "kmer_view.hpp"
#include "minimizer_view.hpp"
#include "super_kmer_view.hpp"
#include "syncmer_sampler.hpp"
#include "hash_sampler.hpp"
#include "minimizer_sampler.hpp"

extern "C" {
#include "../../oracle/bl_oracle.h"
}

typedef uint64_t kmer_t;
typedef uint64_t mmer_t;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (g_fail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } ++g_fail; } } while (0)

struct item { long position, id, value; };  // value -1 = null

static std::vector<item> collect(std::string const& s, uint8_t k, bool canonical, bool complete)
{
    std::vector<item> out;
    auto view = wrapper::kmer_view_from_cstr<kmer_t>(s.c_str(), s.size(), k, canonical);
    auto itr = view.cbegin();
    for (; itr != view.cend(); ++itr) {
        auto v = *itr;
        out.push_back({(long)v.position, (long)v.id, v.value ? (long)*v.value : -1});
    }
    if (complete) {  // the k-mer the idiom leaves behind stays readable (quirk Q1)
        auto v = *itr;
        if (v.value && s.size() >= k && v.position + k == s.size()) out.push_back({(long)v.position, (long)v.id, (long)*v.value});
    }
    return out;
}

static void expect_items(std::string const& s, uint8_t k, bool canonical, std::vector<item> const& exp)
{
    auto got = collect(s, k, canonical, false);
    CHECK(got.size() == exp.size(), "%s k=%d c=%d: %zu items, expected %zu", s.c_str(), k, canonical, got.size(), exp.size());
    for (size_t i = 0; i < got.size() && i < exp.size(); ++i)
        CHECK(got[i].position == exp[i].position && got[i].id == exp[i].id && got[i].value == exp[i].value,
              "%s k=%d c=%d item %zu: (%ld,%ld,%ld) expected (%ld,%ld,%ld)", s.c_str(), k, canonical, i, got[i].position, got[i].id, got[i].value,
              exp[i].position, exp[i].id, exp[i].value);
}

int main()
{
    // ---- kmer_view: reference-generated vectors, k = 3
    expect_items("ACGTTGCA", 3, false, {{0, 0, 6}, {1, 1, 27}, {2, 2, 47}, {3, 3, 62}, {4, 4, 57}});
    expect_items("ACGTTGCA", 3, true, {{0, 0, 6}, {1, 1, 6}, {2, 2, 1}, {3, 3, 16}, {4, 4, 36}});
    expect_items("ACGNTGCAT", 3, false, {{0, 0, 6}, {1, 1, -1}, {4, 2, 57}, {5, 3, 36}});
    expect_items("ACGNTGCAT", 3, true, {{0, 0, 6}, {1, 1, -1}, {4, 2, 36}, {5, 3, 36}});
    expect_items("ACGTN", 3, false, {{0, 0, 6}, {1, 1, 27}});
    expect_items("NACGT", 3, false, {{1, 0, 6}});
    expect_items("NNACGTA", 3, false, {{2, 0, 6}, {3, 1, 27}});
    expect_items("ACNACGTA", 3, false, {{3, 0, 6}, {4, 1, 27}});
    expect_items("ACGNNNTGCAT", 3, false, {{0, 0, 6}, {1, 1, -1}, {6, 2, 57}, {7, 3, 36}});
    expect_items("ACGTNACNGTACGA", 3, false, {{0, 0, 6}, {1, 1, 27}, {2, 2, -1}, {8, 3, 44}, {9, 4, 49}, {10, 5, 6}});
    {   // k = 21 on the survey's 24-mer (zero-initialised buffers)
        auto c = collect("ACGTTGCAGGATCCATTTACGGCA", 21, true, false);
        CHECK(c.size() == 3 && c[0].value == 479200104390 && c[1].value == 1563612248326 && c[2].value == 2589926317633, "k21 canonical");
        auto f = collect("ACGTTGCAGGATCCATTTACGGCA", 21, false, false);
        CHECK(f.size() == 3 && f[0].value == 479200104390 && f[1].value == 1916800417562 && f[2].value == 3269155159145, "k21 forward");
    }
    // outside the reference's defined domain: terminate cleanly (no items, no crash)
    CHECK(collect("AC", 3, false, true).empty(), "short input");
    CHECK(collect("", 3, false, true).empty(), "empty input");

    // ---- kmer_view vs the oracle's protocol on seeded sequences with breaks
    for (int seed = 0; seed < 4; ++seed) {
        std::string s(5000 + 13 * seed, 'A');
        blo_synth(100 + seed, 0, s.size(), s.data());
        if (seed & 1) for (size_t p = 97; p + 300 < s.size(); p += 611) s[p] = "NnRY"[p % 4];
        for (uint8_t k : {(uint8_t)1, (uint8_t)15, (uint8_t)21, (uint8_t)31, (uint8_t)32})
            for (int canon = 0; canon < 2; ++canon)
                for (int complete = 0; complete < 2; ++complete) {
                    std::vector<uint64_t> v(s.size() + 2), p(s.size() + 2), id(s.size() + 2);
                    std::vector<uint8_t> nul(s.size() + 2);
                    size_t n = blo_kmer_items(s.data(), s.size(), k, canon, complete, v.data(), nul.data(), p.data(), id.data(), v.size());
                    auto got = collect(s, k, canon, complete);
                    CHECK(got.size() == n, "seed %d k %d c %d complete %d: %zu items vs %zu", seed, k, canon, complete, got.size(), n);
                    for (size_t i = 0; i < n && i < got.size(); ++i)
                        CHECK(got[i].position == (long)p[i] && got[i].id == (long)id[i] && got[i].value == (nul[i] ? -1 : (long)v[i]), "seed %d k %d item %zu", seed, k, i);
                }
    }

    // ---- a1 over the WHOLE byte domain (reference constants.hpp:12-21; it indexes the table with a signed char, kmer_view.hpp:191, so
    // bytes >= 0x80 are outside its table: this build's contract for them is "break"): every byte value 1..255 at every position
    // of a 16-byte load (0 would end the C string), through the view, against the oracle's item protocol
    {
        std::string s(255 * 16 * 48, 'A');
        blo_synth(4242, 0, s.size(), s.data());
        for (int v = 1; v < 256; ++v)
            for (int o = 0; o < 16; ++o) s[(size_t)((v - 1) * 16 + o) * 48 + 16 + o] = (char)v;
        for (uint8_t k : {(uint8_t)4, (uint8_t)31})
            for (int canon = 0; canon < 2; ++canon) {
                std::vector<uint64_t> v(s.size() + 2), p(s.size() + 2), id(s.size() + 2);
                std::vector<uint8_t> nul(s.size() + 2);
                size_t n = blo_kmer_items(s.data(), s.size(), k, canon, 0, v.data(), nul.data(), p.data(), id.data(), v.size());
                auto got = collect(s, k, canon, false);
                CHECK(got.size() == n, "all bytes k %d c %d: %zu items vs %zu", k, canon, got.size(), n);
                for (size_t i = 0; i < n && i < got.size(); ++i)
                    CHECK(got[i].position == (long)p[i] && got[i].id == (long)id[i] && got[i].value == (nul[i] ? -1 : (long)v[i]), "all bytes k %d item %zu", k, i);
            }
    }

    // ---- hash functors: reference KATs
    CHECK(hash::hash64::hash<uint64_t>(0, 0) == 0x28df63b7cc57c3cbULL, "hash64(0,0)");
    CHECK(hash::hash64::hash<uint64_t>(0x0123456789abcdefULL, 42) == 0xccbfe31ee09a27dbULL, "hash64 seed 42");
    CHECK(hash::hash64::hash<uint64_t>(1, 0x10000002AULL) == hash::hash64::hash<uint64_t>(1, 0x2A), "seed truncation");
    CHECK(hash::double_hash64::hash<uint64_t>(0, 0)[1] == 0xf2557dfcc4e8fe52ULL, "double_hash64 h2");
    CHECK(hash::hash64::hash<uint32_t>(0xdeadbeefu, 7) == 0xf6a411dad5c661d5ULL, "u32 key");
    CHECK(hash::remix(1) == 0x5692161d100b05e5ULL, "remix");
    CHECK(bl_hash64_u64(27, 0) == 0xbfe4bdbaa6420f03ULL, "device-equivalent hash");

    // ---- minimizer_view / super_kmer_view / syncmer_sampler vs the oracle
    for (int seed = 0; seed < 3; ++seed) {
        std::string s(20000 + 7 * seed, 'A');
        blo_synth(7 + seed, 0, s.size(), s.data());
        if (seed == 1) for (size_t p = 500; p + 400 < s.size(); p += 1777) s[p] = 'N';
        const uint64_t offs[2] = {0, s.size()};
        {   // the reference test's parameters (k=15, m=10, seed 42, non canonical) and the BASELINE ones (k=41, m=31 <=> unit 31, w 11)
            const int params[3][3] = {{15, 10, 0}, {41, 31, 1}, {31, 15, 1}};
            for (auto& pr : params) {
                auto view = wrapper::minimizer_view_from_cstr<kmer_t, mmer_t, hash::hash64>(s.c_str(), s.size(), (uint8_t)pr[0], (uint8_t)pr[1], 42, pr[2]);
                std::vector<uint64_t> ov(s.size()), op(s.size()), oh(s.size());
                size_t n = blo_minimizers(s.data(), offs, 1, pr[1], pr[0] - pr[1] + 1, 42, pr[2], 1, ov.data(), op.data(), oh.data(), ov.size());
                size_t i = 0;
                for (auto itr = view.cbegin(); itr != view.cend(); ++itr, ++i) {
                    auto val = *itr;
                    if (i < n) CHECK(val.value == ov[i] && val.position == op[i] && val.id == op[i], "minimizer_view(%d,%d) item %zu", pr[0], pr[1], i);
                }
                CHECK(i == n, "minimizer_view(%d,%d): %zu items vs %zu", pr[0], pr[1], i, n);
            }
        }
        {
            wrapper::super_kmer_view<kmer_t, mmer_t, hash::hash64> view(s.c_str(), s.size(), 31, 15, true, 42);
            std::vector<uint64_t> om(s.size()), of(s.size()), oh(s.size());
            std::vector<uint8_t> op(s.size()), os(s.size());
            size_t n = blo_super_kmers(s.data(), offs, 1, 31, 15, 42, 1, om.data(), of.data(), op.data(), os.data(), oh.data(), om.size());
            size_t i = 0, kmers = 0;
            for (auto itr = view.cbegin(); itr != view.cend(); ++itr, ++i) {
                auto const& sk = *itr;
                kmers += sk.size;
                if (i < n) CHECK(sk.minimizer == om[i] && sk.mm_pos == op[i] && sk.size == os[i] && sk.position == of[i], "super_kmer_view group %zu", i);
            }
            CHECK(i == n, "super_kmer_view: %zu groups vs %zu", i, n);
            if (seed != 1) CHECK(kmers == s.size() - 30, "every k-mer belongs to exactly one super-k-mer");
        }
        for (int canon = 0; canon < 2; ++canon) {
            using view_t = wrapper::kmer_view<kmer_t, char_iterator>;
            auto view = wrapper::kmer_view_from_cstr<kmer_t>(s.c_str(), s.size(), 31, canon);
            hash::minimizer_position_extractor ex(31, 11);
            sampler::syncmer_sampler<view_t::const_iterator, hash::minimizer_position_extractor> smp(view.cbegin(), view.cend(), ex, 0, 20);
            std::vector<uint64_t> op(s.size());
            size_t n = blo_syncmers(s.data(), offs, 1, 31, 11, 0, 20, canon, /*drop_last=*/1, 1, op.data(), op.size());
            size_t i = 0;
            for (auto it = smp.cbegin(); it != smp.cend(); ++it, ++i)
                if (i < n) CHECK(it.position() == op[i] && *it == view.values()[op[i]], "syncmer %zu", i);
            CHECK(i == n && smp.count() == n, "syncmer_sampler canon %d: %zu vs %zu", canon, i, n);
            // the scalar extractor agrees with the oracle's on every k-mer of the view
            size_t checked = 0;
            for (auto itr = view.cbegin(); itr != view.cend() && checked < 3000; ++itr, ++checked) {
                auto kc = *itr;
                if (kc.value) CHECK(ex(kc) == blo_minimizer_position(*kc.value, 31, 11), "extractor at %zu", kc.position);
                else CHECK(ex(kc) == 32, "null item extractor");
            }
        }
        for (double rate : {1.0, 0.3, 0.0}) {  // hash_sampler over kmer_view, hash64: GPU path
            using view_t = wrapper::kmer_view<kmer_t, char_iterator>;
            auto view = wrapper::kmer_view_from_cstr<kmer_t>(s.c_str(), s.size(), 21, true);
            sampler::hash_sampler<view_t::const_iterator, hash::hash64> smp(view.cbegin(), view.cend(), hash::hash64(), 42, rate);
            std::vector<uint64_t> exp;
            const uint64_t thr = rate >= 1.0 ? ~0ULL : (uint64_t)(rate * 18446744073709551615.0);
            for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {  // the reference's own filter, item by item
                auto kc = *itr;
                if (kc.value && hash::hash64::hash(*kc.value, 42) < thr) exp.push_back(*kc.value);
            }
            size_t i = 0;
            for (auto it = smp.cbegin(); it != smp.cend(); ++it, ++i)
                if (i < exp.size()) CHECK(*it == exp[i], "hash_sampler rate %.1f item %zu", rate, i);
            CHECK(i == exp.size(), "hash_sampler rate %.1f: %zu vs %zu", rate, i, exp.size());
        }
        for (int canon = 0; canon < 2; ++canon) {  // minimizer_sampler(w = 11) over kmer_view(k = 31), hash64: the literal "k=31, w=11" entry point
            using view_t = wrapper::kmer_view<kmer_t, char_iterator>;
            auto view = wrapper::kmer_view_from_cstr<kmer_t>(s.c_str(), s.size(), 31, canon);
            sampler::minimizer_sampler<view_t::const_iterator, hash::hash64> smp(view.cbegin(), view.cend(), hash::hash64(), 42, 11);
            CHECK(smp.get_w() == 11, "get_w");
            // the idiom never reaches the k-mer that ends the sequence (Q1) = the sequence without its last base
            std::vector<uint64_t> ov(s.size()), op(s.size()), oh(s.size());
            const uint64_t cut[2] = {0, s.size() - 1};
            size_t n = blo_minimizers(s.data(), cut, 1, 31, 11, 42, canon, 1, ov.data(), op.data(), oh.data(), ov.size());
            size_t i = 0;
            for (auto it = smp.cbegin(); it != smp.cend(); ++it, ++i) {
                auto const& kc = *it;
                if (i < n) CHECK(kc.value && *kc.value == ov[i] && kc.position == op[i], "minimizer_sampler canon %d item %zu", canon, i);
            }
            CHECK(i == n && n > 0, "minimizer_sampler canon %d: %zu vs %zu", canon, i, n);
        }
        {   // the host form of the sampler over optional items with a break in the middle: leftmost minimum, window restarts
            struct idhash { typedef uint64_t hash_type; hash_type operator()(uint64_t v, uint64_t) const {return v;} };
            std::vector<std::optional<uint64_t>> items = {5, 3, 3, 9, std::nullopt, 7, 1, 8, 1, 1};
            sampler::minimizer_sampler<std::vector<std::optional<uint64_t>>::const_iterator, idhash> smp(items.cbegin(), items.cend(), idhash(), 0, 3);
            std::vector<uint64_t> got;
            for (auto it = smp.cbegin(); it != smp.cend(); ++it) got.push_back(**it);
            CHECK((got == std::vector<uint64_t>{3, 1, 1}), "host minimizer_sampler");  // windows 5 3 3 | 3 3 9 (same leftmost 3) || 7 1 8 | 1 8 1 (same 1) | 8 1 1 (the first 1 has left)
        }
        {   // generic path of the sampler: any iterator + any extractor (here: even numbers)
            struct even_extractor { using value_type = int; std::size_t operator()(int v) const {return v % 2;} };
            std::vector<int> nums = {1, 2, 3, 4, 6, 7, 8};
            even_extractor ee;
            sampler::syncmer_sampler<std::vector<int>::const_iterator, even_extractor> smp(nums.cbegin(), nums.cend(), ee, 0, 0);
            std::vector<int> got;
            for (auto it = smp.cbegin(); it != smp.cend(); ++it) got.push_back(*it);
            CHECK((got == std::vector<int>{2, 4, 6, 8}), "generic sampler");
        }
    }
    if (g_fail) { std::printf("test_compat_views: %d failures\n", g_fail); return 1; }
    std::printf("test_compat_views: OK\n");
    return 0;
}
