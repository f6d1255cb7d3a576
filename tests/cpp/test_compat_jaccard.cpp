// GPU test of a Jaccard workflow written against the drop-in headers: the k-mers of two FASTA files go into two
// emem::external_memory_vector (small RAM budget: several run files each), sampler::ordered_unique_sampler walks both in sorted
// order, and the intersection / union sizes are checked four ways — std::set on the host, <algorithm> on the samplers' output,
// algorithm::jaccard and the device form algorithm::jaccard_device.  The reference's own tool for this is tests/test_jaccard.cpp
// (same headers, same signatures: that is what "drop-in" means here); this caller is ours.  Also checks that the run files are the
// ones the reference would leave on disk.
// usage: test_compat_jaccard first.fa second.fa tmp_dir
#include <algorithm>
#include <cstdio>
#include <iterator>
#include <set>
#include <string>
#include <vector>

#include "external_memory_vector.hpp"
#include "jaccard.hpp"
#include "kmer_view.hpp"
#include "ordered_unique_sampler.hpp"

using kmer_t = uint64_t;
using emem_vec = emem::external_memory_vector<kmer_t>;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (g_fail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } ++g_fail; } } while (0)

// every non-null k-mer of every record of `path`, into the spill vector and into the host-side set
static void collect(const std::string& path, uint16_t k, bool canonical, emem_vec& out, std::set<kmer_t>& seen)
{
    biolib_amd::read_pool pool(path);
    char const* seq = nullptr;
    std::size_t len = 0;
    while (pool.next(seq, len)) {
        auto view = wrapper::kmer_view_from_cstr<kmer_t>(seq, len, k, canonical);
        for (auto it = view.cbegin(); it != view.cend(); ++it) {
            const auto item = *it;
            if (!item.value) continue;
            out.push_back(*item.value);
            seen.insert(*item.value);
        }
    }
}

// sorted with duplicates, all elements there, spilled to several files named as the reference names them
static void check_spill(emem_vec& v, const std::string& tmp_dir, const char* tag, uint64_t ram_budget)
{
    std::size_t n = 0;
    kmer_t prev = 0;
    for (auto it = v.cbegin(); it != v.cend(); ++it, ++n) {
        CHECK(n == 0 || prev <= *it, "%s: order at %zu", tag, n);
        prev = *it;
    }
    CHECK(n == v.size(), "%s: iterated %zu of %zu", tag, n, v.size());
    if (v.size() * sizeof(kmer_t) > 2 * ram_budget) CHECK(v.run_files().size() > 1, "%s: expected several run files", tag);
    uint64_t total = 0;
    for (auto const& f : v.run_files()) {
        uint64_t cnt = 0;
        CHECK(bl_file_count_u64(f.c_str(), 0, &cnt) == BL_OK, "run file %s", f.c_str());
        total += cnt;
    }
    CHECK(total == v.size(), "%s: run files hold %llu of %zu elements", tag, (unsigned long long)total, v.size());
    if (!v.run_files().empty()) CHECK(v.run_files()[0] == tmp_dir + "/tmp.run_" + tag + "_0.bin", "%s: reference naming of run files", tag);
}

template <class Sampler>
static std::vector<kmer_t> drain(Sampler& s)
{
    std::vector<kmer_t> out;
    for (auto it = s.cbegin(); it != s.cend(); ++it) out.push_back(*it);
    return out;
}

int main(int argc, char* argv[])
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s first.fa second.fa tmp_dir\n", argv[0]); return 2; }
    const std::string fasta_a = argv[1], fasta_b = argv[2], tmp_dir = argv[3];
    const uint64_t ram_budget = 16000;  // small on purpose: several run files per vector
    for (uint16_t k : {(uint16_t)21, (uint16_t)11}) {
        for (bool canonical : {false, true}) {
            emem_vec vec_a(ram_budget, tmp_dir, "first"), vec_b(ram_budget, tmp_dir, "second");
            std::set<kmer_t> set_a, set_b;
            collect(fasta_a, k, canonical, vec_a, set_a);
            collect(fasta_b, k, canonical, vec_b, set_b);
            std::fprintf(stderr, "k %u canonical %d: %zu and %zu k-mers\n", k, (int)canonical, vec_a.size(), vec_b.size());
            check_spill(vec_a, tmp_dir, "first", ram_budget);
            check_spill(vec_b, tmp_dir, "second", ram_budget);

            // expected, from the host sets
            std::size_t exp_inter = 0;
            for (kmer_t v : set_a) exp_inter += set_b.count(v);
            const std::size_t exp_union = set_a.size() + set_b.size() - exp_inter;

            // the samplers' output is each vector's distinct elements in order
            sampler::ordered_unique_sampler uniq_a(vec_a.cbegin(), vec_a.cend());
            sampler::ordered_unique_sampler uniq_b(vec_b.cbegin(), vec_b.cend());
            const std::vector<kmer_t> da = drain(uniq_a), db = drain(uniq_b);
            CHECK(da.size() == set_a.size() && std::equal(da.begin(), da.end(), set_a.begin()), "first: sampler output != std::set");
            CHECK(db.size() == set_b.size() && std::equal(db.begin(), db.end(), set_b.begin()), "second: sampler output != std::set");
            CHECK(uniq_a.size() && *uniq_a.size() == set_a.size(), "sampler size() after a full walk");
            std::vector<kmer_t> both;
            std::set_intersection(da.begin(), da.end(), db.begin(), db.end(), std::back_inserter(both));
            CHECK(both.size() == exp_inter, "k %u c %d: intersection %zu vs %zu", k, (int)canonical, both.size(), exp_inter);

            // the library's two forms
            const auto j = algorithm::jaccard(uniq_a.cbegin(), uniq_a.cend(), uniq_b.cbegin(), uniq_b.cend());
            CHECK(std::get<0>(j) == exp_inter && std::get<1>(j) == exp_union && std::get<2>(j) == set_a.size() && std::get<3>(j) == set_b.size(),
                  "algorithm::jaccard: %zu/%zu vs %zu/%zu", std::get<0>(j), std::get<1>(j), exp_inter, exp_union);
            const auto d = algorithm::jaccard_device(vec_a, vec_b);
            CHECK(d == j, "algorithm::jaccard_device: %zu/%zu", std::get<0>(d), std::get<1>(d));
            std::printf("Jaccard k=%u canonical=%d: %zu/%zu = %.6f\n", k, (int)canonical, exp_inter, exp_union, exp_union ? double(exp_inter) / exp_union : 0.0);
        }
    }
    if (g_fail) { std::printf("test_compat_jaccard: %d failures\n", g_fail); return 1; }
    std::printf("test_compat_jaccard: OK\n");
    return 0;
}
