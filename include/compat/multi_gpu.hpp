// multi_gpu.hpp — the host side of the path on the GPUs of one node, in C++ (north_star: "host side stays C++ ... input
// sequences shard by read/contig across the 8 GPUs of one node, with RCCL over xGMI used only for the optional final k-mer-count
// reduction").  No counterpart in biolib, which is single-threaded and streams a contig of any length through one view
// (kmer_view.hpp:46-54, 181-202).
//
// biolib_amd::multi_gpu owns one context and one host thread per device.  Host sequences are cut BY BASES: device
// g owns the windows whose first base lies in its share [lo, hi) of the concatenation, whether that cuts a contig or not.  It
// uploads the piece [lo - 1, hi + (unit-1)+(w-1)) — the halo its last window reads, and the one base in front that decides
// whether its first window opens a minimizer occurrence — tells the batch where the piece lies in the whole
// (bl_batch_set_origin: reported positions are global) and scans the range of the piece that is its own in calls of <= 1.5 Gbp.
// The devices' records, in device order, are the records of one scan over the whole; the digests say so: count, xor_value,
// xor_hash AND xor_pos equal the single-device ones for any number of devices.  Counts are summed across the devices with
// bl_count_allreduce (ncclAllReduce over RCCL), XOR digests — for which RCCL has no reduction — are folded on the host.
// There is no data-path collective.
#ifndef BIOLIB_AMD_COMPAT_MULTI_GPU_HPP
#define BIOLIB_AMD_COMPAT_MULTI_GPU_HPP

#include <algorithm>
#include <exception>
#include <functional>
#include <thread>

#include "biolib_amd_runtime.hpp"

namespace biolib_amd {

struct scan_digest {
    uint64_t count = 0, xor_value = 0, xor_hash = 0, xor_pos = 0;
};

class multi_gpu
{
    public:
        // n_devices <= 0: every visible device
        explicit multi_gpu(int n_devices = 0)
        {
            int visible = 0;
            check(bl_device_count(&visible), "bl_device_count");
            const int n = n_devices <= 0 ? visible : std::min(n_devices, visible);
            if (n < 1) throw std::runtime_error("[biolib_amd] no GPU visible: there is no CPU fallback");
            for (int d = 0; d < n; ++d) {
                bl_ctx* c = nullptr;
                check(bl_ctx_create(d, &c), "bl_ctx_create");
                ctxs.push_back(c);
            }
        }
        ~multi_gpu() {for (bl_ctx* c : ctxs) bl_ctx_destroy(c);}
        multi_gpu(multi_gpu const&) = delete;
        int devices() const noexcept {return static_cast<int>(ctxs.size());}

        // contiguous range of sequences [first, end) for shard g of n: about the same number of bases each
        static std::pair<uint64_t, uint64_t> shard_of(uint64_t const* offsets, uint64_t n_seqs, int g, int n)
        {
            const uint64_t total = offsets[n_seqs];
            auto cut = [&](int i) -> uint64_t {
                if (i <= 0) return 0;
                if (i >= n) return n_seqs;
                const uint64_t want = total / n * i + total % n * i / n;
                return static_cast<uint64_t>(std::lower_bound(offsets, offsets + n_seqs + 1, want) - offsets);
            };
            const uint64_t lo = std::min(cut(g), n_seqs), hi = std::min(std::max(cut(g + 1), lo), n_seqs);
            return {lo, hi};
        }

        // [lo, hi) of `total` bases for piece g of n: contiguous, balanced to one base
        static std::pair<uint64_t, uint64_t> base_range(uint64_t total, int g, int n)
        {
            auto cut = [&](int i) -> uint64_t { return total / (uint64_t)n * (uint64_t)i + total % (uint64_t)n * (uint64_t)i / (uint64_t)n; };
            return {cut(g), g + 1 >= n ? total : cut(g + 1)};
        }
        // pieces every device cuts its share into (default 1).  More than one exercises the cutting on a box with a single GPU.
        void set_pieces_per_device(int pieces) noexcept {ppd = pieces < 1 ? 1 : pieces;}

        // minimizer scan (unit-mers, window w) of host sequences; digest over the whole, positions global
        scan_digest minimizers(char const* bases, uint64_t const* offsets, uint64_t n_seqs, uint32_t unit, uint32_t w, uint64_t seed, bool canonical)
        {
            return run(bases, offsets, n_seqs, unit + w - 1, 0, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_minimizers(c, b, first, n, unit, w, seed, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, nullptr, nullptr, 0, r);
            });
        }
        // syncmer count (BASELINE C5): k-mers whose minimum s-mer sits at one of the two offsets
        scan_digest syncmers(char const* bases, uint64_t const* offsets, uint64_t n_seqs, uint32_t k, uint32_t s, uint32_t start_offset, uint32_t end_offset,
                             bool canonical)
        {
            return run(bases, offsets, n_seqs, k, 0, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_syncmers(c, b, first, n, k, s, start_offset, end_offset, 0, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, 0, r);
            });
        }
        // the same over synthetic shards generated on the devices (SURVEY.md §8d generator, seed + device index), read_len-base reads
        scan_digest syncmers_synth(uint64_t seed, uint64_t bases_per_device, uint64_t read_len, uint32_t k, uint32_t s, uint32_t start_offset, uint32_t end_offset,
                                   bool canonical)
        {
            return run_synth(seed, bases_per_device, read_len, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_syncmers(c, b, first, n, k, s, start_offset, end_offset, 0, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, 0, r);
            });
        }
        scan_digest minimizers_synth(uint64_t seed, uint64_t bases_per_device, uint64_t read_len, uint32_t unit, uint32_t w, uint64_t hash_seed, bool canonical)
        {
            return run_synth(seed, bases_per_device, read_len, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_minimizers(c, b, first, n, unit, w, hash_seed, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, nullptr, nullptr, 0, r);
            });
        }
        // the same scans over a plain or bgzip'ed FASTA / FASTQ FILE that the devices read between them: every device thread opens its own
        // part(s) of the file (bl_reader_open_shard: parts begin and end at record starts that each reader recognises by itself),
        // inflates and parses them on its GPU and scans batch after batch.  xor_pos is folded over positions relative to each
        // batch and means nothing here; count, xor_value and xor_hash do not depend on how the file was cut.
        scan_digest minimizers_file(std::string const& path, uint32_t unit, uint32_t w, uint64_t seed, bool canonical, int parts_per_device = 1)
        {
            return run_file(path, parts_per_device, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_minimizers(c, b, first, n, unit, w, seed, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, nullptr, nullptr, 0, r);
            });
        }
        scan_digest syncmers_file(std::string const& path, uint32_t k, uint32_t s, uint32_t start_offset, uint32_t end_offset, bool canonical, int parts_per_device = 1)
        {
            return run_file(path, parts_per_device, [=](bl_ctx* c, bl_batch* b, uint64_t first, uint64_t n, bl_result* r) {
                return bl_scan_syncmers(c, b, first, n, k, s, start_offset, end_offset, 0, canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u, nullptr, 0, r);
            });
        }
        std::vector<scan_digest> const& per_device() const noexcept {return last;}
        uint64_t bases_read() const noexcept {return file_bases;}

    private:
        using scan_fn = std::function<int(bl_ctx*, bl_batch*, uint64_t, uint64_t, bl_result*)>;
        static constexpr uint64_t RANGE = 1500000000ull;
        std::vector<bl_ctx*> ctxs;
        std::vector<scan_digest> last;
        uint64_t file_bases = 0;
        int ppd = 1;

        // scan the range [first, first + n_bases) of a batch in calls that end on read boundaries for fixed-length reads
        static scan_digest scan_batch(bl_ctx* c, bl_batch* b, uint64_t n_bases, uint64_t align, scan_fn const& scan, uint64_t first = 0)
        {
            std::vector<bl_result> res;
            const uint64_t step = align ? std::max<uint64_t>(align, RANGE / align * align) : RANGE;
            res.reserve(n_bases / step + 1);
            for (uint64_t a = 0; a < n_bases; a += step) {
                res.emplace_back();
                check(scan(c, b, first + a, std::min(step, n_bases - a), &res.back()), "scan");
            }
            check(bl_ctx_sync(c), "bl_ctx_sync");
            scan_digest d;
            for (bl_result const& r : res) {
                if (r.status != BL_OK) throw std::runtime_error("[biolib_amd] a scan range failed");
                d.count += r.count;
                d.xor_value ^= r.xor_value;
                d.xor_hash ^= r.xor_hash;
                d.xor_pos ^= r.xor_pos;
            }
            return d;
        }
        scan_digest reduce()
        {
            // counts: summed across the devices over RCCL; XOR digests: folded on the host (no XOR reduction in RCCL)
            const int n = devices();
            std::vector<uint64_t> counters(n);
            for (int g = 0; g < n; ++g) counters[g] = last[g].count;
            check(bl_count_allreduce(ctxs.data(), n, counters.data(), 1), "bl_count_allreduce");
            scan_digest total;
            total.count = counters[0];
            for (int g = 0; g < n; ++g) {
                if (counters[g] != counters[0]) throw std::runtime_error("[biolib_amd] the all-reduce left different sums on different devices");
                total.xor_value ^= last[g].xor_value;
                total.xor_hash ^= last[g].xor_hash;
                total.xor_pos ^= last[g].xor_pos;
            }
            return total;
        }
        void in_parallel(std::function<void(int)> const& work)
        {
            const int n = devices();
            std::vector<std::exception_ptr> errors(n);
            std::vector<std::thread> threads;
            for (int g = 0; g < n; ++g)
                threads.emplace_back([&, g] {
                    try { work(g); } catch (...) { errors[g] = std::current_exception(); }
                });
            for (auto& t : threads) t.join();
            for (auto const& e : errors)
                if (e) std::rethrow_exception(e);
        }
        // span: bases of one window (unit + w - 1; k for k-mer scans); tail: 1 for scans that drop the k-mer that ends its sequence
        scan_digest run(char const* bases, uint64_t const* offsets, uint64_t n_seqs, uint64_t span, uint64_t tail, scan_fn scan)
        {
            const int n = devices(), pieces = n * ppd;
            const uint64_t total = offsets[n_seqs];
            last.assign(n, scan_digest());
            in_parallel([&](int g) {
                for (int piece = g * ppd; piece < (g + 1) * ppd; ++piece) {
                    auto [lo, hi] = base_range(total, piece, pieces);
                    if (hi <= lo) continue;
                    // the piece: one base in front of lo (does the window at lo open an occurrence?), the last window's bases behind hi
                    const uint64_t piece_lo = lo ? lo - 1 : 0, piece_hi = std::min(total, hi + span - 1 + tail);
                    std::vector<uint64_t> local = {0};  // sequence starts strictly inside the piece; its first base opens a sequence
                    for (uint64_t const* q = std::upper_bound(offsets, offsets + n_seqs + 1, piece_lo); q < offsets + n_seqs + 1 && *q < piece_hi; ++q)
                        local.push_back(*q - piece_lo);
                    local.push_back(piece_hi - piece_lo);
                    bl_batch* b = nullptr;
                    check(bl_batch_upload(ctxs[g], bases + piece_lo, piece_hi - piece_lo, local.data(), local.size() - 1, &b), "bl_batch_upload");
                    scan_digest d;
                    try {
                        check(bl_batch_set_origin(b, piece_lo), "bl_batch_set_origin");
                        d = scan_batch(ctxs[g], b, hi - lo, 0, scan, lo - piece_lo);
                    } catch (...) { bl_batch_destroy(b); throw; }
                    bl_batch_destroy(b);
                    last[g].count += d.count;
                    last[g].xor_value ^= d.xor_value;
                    last[g].xor_hash ^= d.xor_hash;
                    last[g].xor_pos ^= d.xor_pos;
                }
            });
            return reduce();
        }
        scan_digest run_file(std::string const& path, int parts_per_device, scan_fn scan)
        {
            const int n = devices(), ppd = parts_per_device < 1 ? 1 : parts_per_device;
            last.assign(n, scan_digest());
            std::vector<uint64_t> bases(n, 0);
            in_parallel([&](int g) {
                for (int part = g * ppd; part < (g + 1) * ppd; ++part) {
                    bl_reader* reader = nullptr;
                    check(bl_reader_open_shard(path.c_str(), (uint32_t)part, (uint32_t)(n * ppd), &reader), "bl_reader_open_shard");
                    try {
                        for (;;) {
                            bl_batch* b = nullptr;
                            uint64_t n_seqs = 0, n_bases = 0;
                            check(bl_reader_next_batch_device(ctxs[g], reader, 0, &b, &n_seqs, &n_bases), "bl_reader_next_batch_device");
                            if (not b) break;
                            scan_digest d;
                            try { d = scan_batch(ctxs[g], b, n_bases, 0, scan); } catch (...) { bl_batch_destroy(b); throw; }
                            bl_batch_destroy(b);
                            last[g].count += d.count;
                            last[g].xor_value ^= d.xor_value;
                            last[g].xor_hash ^= d.xor_hash;
                            bases[g] += n_bases;
                        }
                    } catch (...) { bl_reader_close(reader); throw; }
                    bl_reader_close(reader);
                }
            });
            file_bases = 0;
            for (uint64_t b : bases) file_bases += b;
            return reduce();
        }
        scan_digest run_synth(uint64_t seed, uint64_t bases_per_device, uint64_t read_len, scan_fn scan)
        {
            const int n = devices();
            last.assign(n, scan_digest());
            in_parallel([&](int g) {
                bl_batch* b = nullptr;
                check(bl_batch_synth(ctxs[g], seed + g, bases_per_device, read_len, &b), "bl_batch_synth");
                try { last[g] = scan_batch(ctxs[g], b, bases_per_device, read_len, scan); } catch (...) { bl_batch_destroy(b); throw; }
                bl_batch_destroy(b);
            });
            return reduce();
        }
};

}  // namespace biolib_amd

#endif
