// GPU test of the C++ multi-device driver (include/compat/multi_gpu.hpp) and of bl_count_allreduce (RCCL's C API): run with
// every visible device — ONE on the test box, so what is exercised is the whole flow (threads, shards, ncclCommInitAll,
// ncclAllReduce) at world size 1; N > 1 needs hardware this box does not have.
#include <zlib.h>

#include <cstdio>
#include <string>
#include <vector>

#include "multi_gpu.hpp"

extern "C" {
#include "../../oracle/bl_oracle.h"
}

// a BGZF file (SAM spec 4.1: gzip members with the 'BC' extra field) holding `text`, members of `block` bytes of text
static void write_bgzf(const std::string& path, const std::string& text, size_t block)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    for (size_t a = 0; a <= text.size(); a += block) {
        const size_t n = a < text.size() ? std::min(block, text.size() - a) : 0;  // the last, empty member is the end-of-file marker
        std::vector<unsigned char> body(compressBound(n) + 64);
        z_stream z{};
        deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        z.next_in = reinterpret_cast<unsigned char*>(const_cast<char*>(text.data() + a));
        z.avail_in = (uInt)n;
        z.next_out = body.data();
        z.avail_out = (uInt)body.size();
        deflate(&z, Z_FINISH);
        const size_t nb = z.total_out;
        deflateEnd(&z);
        const unsigned bsize = (unsigned)(12 + 6 + nb + 8 - 1);
        const unsigned long crc = crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char*>(text.data() + a), (uInt)n);
        const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
        std::fwrite(head, 1, 18, f);
        std::fwrite(body.data(), 1, nb, f);
        const unsigned char tail[8] = {(unsigned char)crc, (unsigned char)(crc >> 8), (unsigned char)(crc >> 16), (unsigned char)(crc >> 24),
                                       (unsigned char)n, (unsigned char)(n >> 8), (unsigned char)(n >> 16), (unsigned char)(n >> 24)};
        std::fwrite(tail, 1, 8, f);
        if (n == 0) break;
    }
    std::fclose(f);
}

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (g_fail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } ++g_fail; } } while (0)

int main()
{
    biolib_amd::multi_gpu node;
    std::printf("devices: %d\n", node.devices());
    // shard_of: contiguous, covering, balanced
    {
        std::vector<uint64_t> offs = {0};
        for (int i = 0; i < 1000; ++i) offs.push_back(offs.back() + 50 + (i * 37) % 400);
        for (int n : {1, 2, 3, 8}) {
            uint64_t expect_lo = 0;
            for (int g = 0; g < n; ++g) {
                auto [lo, hi] = biolib_amd::multi_gpu::shard_of(offs.data(), 1000, g, n);
                CHECK(lo == expect_lo && hi >= lo, "shard %d of %d: [%llu, %llu)", g, n, (unsigned long long)lo, (unsigned long long)hi);
                const double share = double(offs[hi] - offs[lo]) / double(offs.back());
                CHECK(share > 0.8 / n && share < 1.2 / n, "shard %d of %d holds %.3f of the bases", g, n, share);
                expect_lo = hi;
            }
            CHECK(expect_lo == 1000, "shards cover the reads");
        }
    }
    // base_range: contiguous, covering, balanced to one base
    for (uint64_t total : {0ull, 1ull, 1000ull, 400000000007ull})
        for (int n : {1, 2, 3, 8}) {
            uint64_t expect_lo = 0, least = ~0ull, most = 0;
            for (int g = 0; g < n; ++g) {
                auto [lo, hi] = biolib_amd::multi_gpu::base_range(total, g, n);
                CHECK(lo == expect_lo && hi >= lo, "piece %d of %d", g, n);
                least = std::min(least, hi - lo); most = std::max(most, hi - lo);
                expect_lo = hi;
            }
            CHECK(expect_lo == total && most - least <= 1, "pieces cover %llu bases evenly", (unsigned long long)total);
        }
    // ONE contig with breaks, cut by bases into 1, 3 and 7 pieces per device (halo of (unit-1)+(w-1) bases, windows
    // owned by the piece that holds their first base, positions global): all four digests, the position digest included, equal
    // the oracle's over the whole contig
    {
        const uint64_t n = 5000000;
        std::string s(n, 'A');
        blo_synth(21, 0, n, s.data());
        for (uint64_t p = 777; p < n; p += 10007) s[p] = "NnRY-"[p % 5];
        const uint64_t offs[2] = {0, n};
        uint64_t dg[4];
        blo_minimizer_digest(s.data(), offs, 1, 31, 11, 42, 1, 1, dg);
        std::vector<uint64_t> pos(n);
        const uint64_t cnt = blo_syncmers(s.data(), offs, 1, 31, 11, 0, 20, 1, 0, 1, pos.data(), pos.size());
        uint64_t xp = 0;
        for (uint64_t i = 0; i < cnt; ++i) xp ^= pos[i];
        for (int pieces : {1, 3, 7}) {
            node.set_pieces_per_device(pieces);
            auto got = node.minimizers(s.data(), offs, 1, 31, 11, 42, true);
            CHECK(got.count == dg[0] && got.xor_value == dg[1] && got.xor_hash == dg[2] && got.xor_pos == dg[3], "one contig in %d piece(s) per device: %llu vs %llu",
                  pieces, (unsigned long long)got.count, (unsigned long long)dg[0]);
            auto sy = node.syncmers(s.data(), offs, 1, 31, 11, 0, 20, true);
            CHECK(sy.count == cnt && sy.xor_pos == xp, "syncmers of one contig in %d piece(s): %llu vs %llu", pieces, (unsigned long long)sy.count, (unsigned long long)cnt);
        }
        node.set_pieces_per_device(1);
    }
    // host reads: ragged lengths; the digests — positions are global — equal the oracle's over the whole input for any number of
    // devices and pieces
    {
        const uint64_t n = 3000000;
        std::string s(n, 'A');
        blo_synth(9, 0, n, s.data());
        for (uint64_t p = 1000; p < n; p += 7919) s[p] = 'N';
        std::vector<uint64_t> offs = {0};
        while (offs.back() < n) offs.push_back(std::min<uint64_t>(n, offs.back() + 100 + (offs.size() * 131) % 300));
        const uint64_t n_seqs = offs.size() - 1;
        uint64_t dg[4];
        blo_minimizer_digest(s.data(), offs.data(), n_seqs, 31, 11, 42, 1, 1, dg);
        auto got = node.minimizers(s.data(), offs.data(), n_seqs, 31, 11, 42, true);
        CHECK(got.count == dg[0] && got.xor_value == dg[1] && got.xor_hash == dg[2], "minimizers over %d device(s): %llu vs %llu", node.devices(),
              (unsigned long long)got.count, (unsigned long long)dg[0]);
        CHECK(got.xor_pos == dg[3], "position digest");
        node.set_pieces_per_device(5);
        auto cut = node.minimizers(s.data(), offs.data(), n_seqs, 31, 11, 42, true);
        CHECK(cut.count == dg[0] && cut.xor_value == dg[1] && cut.xor_hash == dg[2] && cut.xor_pos == dg[3], "reads cut into 5 pieces per device");
        node.set_pieces_per_device(1);
        std::vector<uint64_t> pos(n);
        const uint64_t cnt = blo_syncmers(s.data(), offs.data(), n_seqs, 31, 11, 0, 20, 1, 0, 1, pos.data(), pos.size());
        auto sy = node.syncmers(s.data(), offs.data(), n_seqs, 31, 11, 0, 20, true);
        CHECK(sy.count == cnt, "syncmers: %llu vs %llu", (unsigned long long)sy.count, (unsigned long long)cnt);
    }
    // the same reads as a bgzip'ed FASTQ file that the devices read between them (three parts per device here, so that the cutting is
    // exercised on one GPU too): same count and XOR digests as the scan of the reads in memory
    {
        const uint64_t n_reads = 30000, L = 150;
        std::string s(n_reads * L, 'A');
        blo_synth(11, 0, n_reads * L, s.data());
        for (uint64_t p = 500; p < s.size(); p += 4099) s[p] = 'N';
        std::vector<uint64_t> offs;
        std::string text;
        for (uint64_t i = 0; i <= n_reads; ++i) offs.push_back(i * L);
        for (uint64_t i = 0; i < n_reads; ++i) {
            text += "@read" + std::to_string(i) + "\n";
            text.append(s, i * L, L);
            text += "\n+\n" + std::string(L, i % 3 ? 'F' : '@') + "\n";
        }
        const std::string path = "/tmp/bl_multi_gpu_test.fq.gz";
        write_bgzf(path, text, 60000);
        uint64_t dg[4];
        blo_minimizer_digest(s.data(), offs.data(), n_reads, 31, 11, 42, 1, 1, dg);
        for (int parts : {1, 3}) {
            auto got = node.minimizers_file(path, 31, 11, 42, true, parts);
            CHECK(got.count == dg[0] && got.xor_value == dg[1] && got.xor_hash == dg[2] && node.bases_read() == n_reads * L,
                  "file in %d part(s) per device: %llu minimizers vs %llu, %llu bases", parts, (unsigned long long)got.count, (unsigned long long)dg[0],
                  (unsigned long long)node.bases_read());
        }
        std::remove(path.c_str());
        // the same reads as a plain file: the parts are byte ranges that meet at record starts
        const std::string plain = "/tmp/bl_multi_gpu_test.fq";
        {
            FILE* f = std::fopen(plain.c_str(), "wb");
            CHECK(f && std::fwrite(text.data(), 1, text.size(), f) == text.size(), "write %s", plain.c_str());
            if (f) std::fclose(f);
        }
        for (int parts : {1, 4}) {
            auto got = node.minimizers_file(plain, 31, 11, 42, true, parts);
            CHECK(got.count == dg[0] && got.xor_value == dg[1] && got.xor_hash == dg[2] && node.bases_read() == n_reads * L,
                  "plain file in %d part(s) per device: %llu minimizers vs %llu, %llu bases", parts, (unsigned long long)got.count, (unsigned long long)dg[0],
                  (unsigned long long)node.bases_read());
        }
        std::remove(plain.c_str());
    }
    // synthetic shards on the devices (BASELINE C5's shape, small): per-device counts add up; device 0's shard = the oracle's seed
    {
        const uint64_t L = 10000, per = 2000 * L;
        auto sy = node.syncmers_synth(42, per, L, 31, 11, 0, 20, true);
        uint64_t sum = 0;
        for (auto const& d : node.per_device()) sum += d.count;
        CHECK(sy.count == sum && sum > 0, "all-reduced count %llu vs sum of shards %llu", (unsigned long long)sy.count, (unsigned long long)sum);
        std::string s(per, 'A');
        blo_synth(42, 0, per, s.data());
        std::vector<uint64_t> offs;
        for (uint64_t p = 0; p <= per; p += L) offs.push_back(p);
        const uint64_t cnt = blo_syncmers(s.data(), offs.data(), offs.size() - 1, 31, 11, 0, 20, 1, 0, 1, nullptr, 0);
        CHECK(node.per_device()[0].count == cnt, "device 0's shard: %llu vs oracle %llu", (unsigned long long)node.per_device()[0].count, (unsigned long long)cnt);
        auto mm = node.minimizers_synth(42, 1500000, 150, 31, 11, 42, true);
        CHECK(mm.count > 0 && mm.count == [&] { uint64_t t = 0; for (auto const& d : node.per_device()) t += d.count; return t; }(), "minimizer shards add up");
    }
    // the collective itself, several counters, wrap-around sums; bad arguments are refused
    {
        bl_ctx* c0 = nullptr;
        biolib_amd::check(bl_ctx_create(0, &c0), "bl_ctx_create");
        uint64_t counters[3] = {5, ~0ull, 1234567890123ull};
        bl_ctx* one[1] = {c0};
        CHECK(bl_count_allreduce(one, 1, counters, 3) == BL_OK, "allreduce: %s", bl_last_error());
        CHECK(counters[0] == 5 && counters[1] == ~0ull && counters[2] == 1234567890123ull, "world 1: sums = inputs");
        bl_ctx* twice[2] = {c0, c0};
        uint64_t two[2] = {1, 2};
        CHECK(bl_count_allreduce(twice, 2, two, 1) == BL_ERR_INVALID, "two ranks on one device must be refused");
        CHECK(bl_count_allreduce(one, 0, counters, 3) == BL_ERR_INVALID && bl_count_allreduce(nullptr, 1, counters, 3) == BL_ERR_INVALID, "bad arguments");
        bl_ctx_destroy(c0);
    }
    if (g_fail) { std::printf("test_multi_gpu: %d failures\n", g_fail); return 1; }
    std::printf("test_multi_gpu: OK\n");
    return 0;
}
