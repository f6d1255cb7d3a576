// GPU test of the Jaccard workflow through the drop-in headers: the body of the reference's tool
// (tests/test_jaccard.cpp:55-130 in the reference tree) — k-mers of two FASTA files pushed into two
// emem::external_memory_vector, ordered_unique_sampler over both, the two-finger count — with the file reader swapped for
// the library's (the reference reads with a third-party header that is not part of this repository) and the printed
// result checked: against a std::set computation on the host, against algorithm::jaccard and against the device form
// algorithm::jaccard_device.  Also checks that the run files are the ones the reference would leave on disk.
// usage: test_compat_jaccard first.fa second.fa tmp_dir
#include <cstdio>
#include <iostream>
#include <set>
#include <string>

#include "external_memory_vector.hpp"
#include "jaccard.hpp"
#include "kmer_view.hpp"
#include "ordered_unique_sampler.hpp"

typedef uint64_t kmer_t;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (g_fail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } ++g_fail; } } while (0)

int main(int argc, char* argv[])
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s first.fa second.fa tmp_dir\n", argv[0]); return 2; }
    std::string first_fasta = argv[1];
    std::string second_fasta = argv[2];
    std::string tmp_dir = argv[3];
    for (uint16_t k : {(uint16_t)21, (uint16_t)11}) {
        for (int c = 0; c < 2; ++c) {
            bool canonical = c != 0;
            uint64_t max_ram_bytes = 16000;  // small on purpose: several run files per vector

            emem::external_memory_vector<kmer_t> kmer_vector_first(max_ram_bytes, tmp_dir, "first");
            emem::external_memory_vector<kmer_t> kmer_vector_second(max_ram_bytes, tmp_dir, "second");
            std::set<kmer_t> set_first, set_second;

            {
                biolib_amd::read_pool pool(first_fasta);
                char const* s; std::size_t l;
                while (pool.next(s, l)) {
                    auto view = wrapper::kmer_view_from_cstr<kmer_t>(s, l, k, canonical);
                    for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
                        if ((*itr).value) kmer_vector_first.push_back(*((*itr).value));
                        if ((*itr).value) set_first.insert(*((*itr).value));
                    }
                }
            }
            {
                biolib_amd::read_pool pool(second_fasta);
                char const* s; std::size_t l;
                while (pool.next(s, l)) {
                    auto view = wrapper::kmer_view_from_cstr<kmer_t>(s, l, k, canonical);
                    for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
                        if ((*itr).value) kmer_vector_second.push_back(*((*itr).value));
                        if ((*itr).value) set_second.insert(*((*itr).value));
                    }
                }
            }

            std::cerr << "found " << kmer_vector_first.size() << " and " << kmer_vector_second.size() << " k-mers" << std::endl;
            // iteration yields the elements in sorted order, duplicates kept
            {
                kmer_t prev = 0; std::size_t n = 0;
                for (auto it = kmer_vector_first.cbegin(); it != kmer_vector_first.cend(); ++it, ++n) { CHECK(n == 0 || prev <= *it, "order at %zu", n); prev = *it; }
                CHECK(n == kmer_vector_first.size(), "iterated %zu of %zu", n, kmer_vector_first.size());
                CHECK(kmer_vector_first.run_files().size() > 1, "expected several run files");
                uint64_t cnt = 0, total = 0;
                for (auto const& f : kmer_vector_first.run_files()) { CHECK(bl_file_count_u64(f.c_str(), 0, &cnt) == BL_OK, "run file %s", f.c_str()); total += cnt; }
                CHECK(total == kmer_vector_first.size() && kmer_vector_first.run_files()[0] == tmp_dir + "/tmp.run_first_0.bin", "run files hold the elements, reference naming");
            }

            sampler::ordered_unique_sampler unique_kmers_first(kmer_vector_first.cbegin(), kmer_vector_first.cend());
            sampler::ordered_unique_sampler unique_kmers_second(kmer_vector_second.cbegin(), kmer_vector_second.cend());
            std::size_t unique_kmers_size_first = 0;
            std::size_t unique_kmers_size_second = 0;
            auto itr_first = unique_kmers_first.cbegin();
            auto itr_second = unique_kmers_second.cbegin();
            auto first_ok = [&unique_kmers_first, &itr_first]() {
                return itr_first != unique_kmers_first.cend();
            };
            auto second_ok = [&unique_kmers_second, &itr_second]() {
                return itr_second != unique_kmers_second.cend();
            };
            std::size_t unione, intersection;
            unione = intersection = 0;
            while(first_ok() or second_ok())
            {
                if(first_ok() and second_ok() and *itr_first == *itr_second)
                {
                    unione += 1;
                    intersection += 1;
                    ++itr_first;
                    ++itr_second;
                    ++unique_kmers_size_first;
                    ++unique_kmers_size_second;
                }
                else if (first_ok() and (not second_ok() or *itr_first < *itr_second))
                {
                    unione += 1;
                    ++itr_first;
                    ++unique_kmers_size_first;
                }
                else if (second_ok() and (not first_ok() or *itr_second < *itr_first))
                {
                    unione += 1;
                    ++itr_second;
                    ++unique_kmers_size_second;
                }
            }
            std::cout << "Jaccard : " << intersection << "/" << unione << " = " << double(intersection) / unione << "\n";

            std::size_t exp_inter = 0;
            for (kmer_t v : set_first) exp_inter += set_second.count(v);
            const std::size_t exp_union = set_first.size() + set_second.size() - exp_inter;
            CHECK(intersection == exp_inter && unione == exp_union, "k %u c %d: %zu/%zu vs %zu/%zu", k, c, intersection, unione, exp_inter, exp_union);
            CHECK(unique_kmers_size_first == set_first.size() && unique_kmers_size_second == set_second.size(), "distinct counts");
            CHECK(unique_kmers_first.size() && *unique_kmers_first.size() == set_first.size(), "sampler size() after a full walk");
            auto j = algorithm::jaccard(unique_kmers_first.cbegin(), unique_kmers_first.cend(), unique_kmers_second.cbegin(), unique_kmers_second.cend());
            CHECK(std::get<0>(j) == exp_inter && std::get<1>(j) == exp_union && std::get<2>(j) == set_first.size() && std::get<3>(j) == set_second.size(), "algorithm::jaccard");
            auto d = algorithm::jaccard_device(kmer_vector_first, kmer_vector_second);
            CHECK(d == j, "algorithm::jaccard_device: %zu/%zu", std::get<0>(d), std::get<1>(d));
        }
    }
    if (g_fail) { std::printf("test_compat_jaccard: %d failures\n", g_fail); return 1; }
    std::printf("test_compat_jaccard: OK\n");
    return 0;
}
