// Throughput of the UNCHANGED per-read loop of the reference's driver (tests/test_kmer_view.cpp:30-42 in the reference tree)
// through the drop-in kmer_view:  one view per read, iterate, use every k-mer.
//   pooled    records come from a biolib_amd::read_pool: one upload + one scan per batch of reads, views index into it
//   per_view  records come from a plain host buffer: every view uploads, scans and downloads by itself (first N reads only)
// usage: bench_view_loop file.fq k canonical [per_view_reads]   -> one JSON line
//        bench_view_loop file.fq k canonical per_view_reads m seed
//            the same for the reference's minimizer driver (tests/test_minimizer_view.cpp:37-43): minimizer_view_from_cstr per read
//        bench_view_loop file.fq k canonical per_view_reads m seed super
//            and for its super-k-mer driver (tests/test_super_kmer_view.cpp:30-36): one super_kmer_view per read
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kmer_view.hpp"
#include "minimizer_view.hpp"
#include "super_kmer_view.hpp"

typedef uint64_t kmer_t;
typedef uint64_t mmer_t;

// the minimizer driver's loop: count, XOR of values, XOR of (hash-free) positions rebased to the read, over pooled / plain memory
struct mini_digest {
    uint64_t count = 0, xor_value = 0, sum_pos = 0;
};
static bool g_super = false;  // digest super-k-mers instead: value = minimizer, position = first k-mer + mm_pos + 256 * size
static void add_view(mini_digest& d, char const* s, std::size_t l, uint8_t k, uint8_t m, uint64_t seed, bool canonical)
{
    if (g_super) {
        wrapper::super_kmer_view<kmer_t, mmer_t, hash::hash64> view(s, l, k, m, canonical, seed);
        for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
            auto const& sk = *itr;
            ++d.count;
            d.xor_value ^= sk.minimizer;
            d.sum_pos += sk.position + sk.mm_pos + 256ull * sk.size;
        }
        return;
    }
    auto view = wrapper::minimizer_view_from_cstr<kmer_t, mmer_t, hash::hash64>(s, l, k, m, seed, canonical);
    for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
        auto val = *itr;
        ++d.count;
        d.xor_value ^= val.value;
        d.sum_pos += val.position;
    }
}
static int minimizer_loop(const std::string& path, uint8_t k, uint8_t m, uint64_t seed, bool canonical, size_t per_view_reads)
{
    using clock = std::chrono::steady_clock;
    (void)biolib_amd::context::get();
    mini_digest pooled, pooled_head, plain;
    uint64_t reads = 0, bases = 0, scans = 0;
    std::vector<std::string> keep;
    const auto t0 = clock::now();
    {
        biolib_amd::read_pool pool(path);
        char const* s; std::size_t l;
        while (pool.next(s, l)) {
            add_view(pooled, s, l, k, m, seed, canonical);
            if (keep.size() < per_view_reads) {
                add_view(pooled_head, s, l, k, m, seed, canonical);
                keep.emplace_back(s, l);
            }
            ++reads;
            bases += l;
        }
        scans = pool.batches_scanned();
    }
    const double pooled_s = std::chrono::duration<double>(clock::now() - t0).count();
    const auto t1 = clock::now();
    for (auto const& r : keep) add_view(plain, r.c_str(), r.size(), k, m, seed, canonical);
    const double per_view_s = std::chrono::duration<double>(clock::now() - t1).count();
    std::printf("{\"reads\": %llu, \"bases\": %llu, \"minimizers\": %llu, \"xor_values\": %llu, \"sum_positions\": %llu, \"batch_scans\": %llu, \"pooled_seconds\": %.4f, "
                "\"pooled_reads_per_s\": %.0f, \"per_view_reads\": %zu, \"per_view_seconds\": %.4f, \"per_view_reads_per_s\": %.0f, "
                "\"head_pooled\": [%llu, %llu, %llu], \"head_per_view\": [%llu, %llu, %llu]}\n",
                (unsigned long long)reads, (unsigned long long)bases, (unsigned long long)pooled.count, (unsigned long long)pooled.xor_value, (unsigned long long)pooled.sum_pos,
                (unsigned long long)scans, pooled_s, reads / pooled_s, keep.size(), per_view_s, keep.size() / per_view_s, (unsigned long long)pooled_head.count,
                (unsigned long long)pooled_head.xor_value, (unsigned long long)pooled_head.sum_pos, (unsigned long long)plain.count, (unsigned long long)plain.xor_value,
                (unsigned long long)plain.sum_pos);
    return 0;
}

int main(int argc, char* argv[])
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s file k canonical [per_view_reads]\n", argv[0]); return 2; }
    const std::string path = argv[1];
    const uint8_t k = (uint8_t)std::atoi(argv[2]);
    const bool canonical = std::atoi(argv[3]) != 0;
    const size_t per_view_reads = argc > 4 ? (size_t)std::atoll(argv[4]) : 2000;
    g_super = argc > 7 and std::string(argv[7]) == "super";
    if (argc > 6) return minimizer_loop(path, k, (uint8_t)std::atoi(argv[5]), std::strtoull(argv[6], nullptr, 10), canonical, per_view_reads);
    using clock = std::chrono::steady_clock;
    (void)biolib_amd::context::get();  // context creation is not part of either loop

    uint64_t reads = 0, kmers = 0, x = 0, bases = 0;
    std::vector<std::string> keep;  // the first reads again, for the per-view loop
    const auto t0 = clock::now();
    uint64_t scans = 0;
    {
        biolib_amd::read_pool pool(path);
        char const* s; std::size_t l;
        while (pool.next(s, l)) {
            auto view = wrapper::kmer_view_from_cstr<kmer_t>(s, l, k, canonical);
            for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
                if ((*itr).value) { x ^= *((*itr).value); ++kmers; }
            }
            ++reads;
            bases += l;
            if (keep.size() < per_view_reads) keep.emplace_back(s, l);
        }
        scans = pool.batches_scanned();
    }
    const double pooled_s = std::chrono::duration<double>(clock::now() - t0).count();

    uint64_t x2 = 0, kmers2 = 0;
    const auto t1 = clock::now();
    for (auto const& r : keep) {
        auto view = wrapper::kmer_view_from_cstr<kmer_t>(r.c_str(), r.size(), k, canonical);
        for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
            if ((*itr).value) { x2 ^= *((*itr).value); ++kmers2; }
        }
    }
    const double per_view_s = std::chrono::duration<double>(clock::now() - t1).count();
    std::printf("{\"reads\": %llu, \"bases\": %llu, \"kmers\": %llu, \"xor_values\": %llu, \"batch_scans\": %llu, \"pooled_seconds\": %.4f, \"pooled_reads_per_s\": %.0f, "
                "\"pooled_Mbp_per_s\": %.1f, \"per_view_reads\": %zu, \"per_view_kmers\": %llu, \"per_view_xor\": %llu, \"per_view_seconds\": %.4f, \"per_view_reads_per_s\": %.0f}\n",
                (unsigned long long)reads, (unsigned long long)bases, (unsigned long long)kmers, (unsigned long long)x, (unsigned long long)scans, pooled_s, reads / pooled_s,
                bases / pooled_s / 1e6, keep.size(), (unsigned long long)kmers2, (unsigned long long)x2, per_view_s, keep.size() / per_view_s);
    return 0;
}
