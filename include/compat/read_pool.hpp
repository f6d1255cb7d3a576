// read_pool.hpp — batched front end for the drop-in views (no counterpart in biolib: it exists because of how biolib's own
// drivers use the views).  The reference builds one view per sequence inside its read loop (tests/test_kmer_view.cpp:30-42):
//     while (kseq_read(seq) >= 0) { auto view = wrapper::kmer_view_from_cstr<kmer_t>(seq->seq.s, seq->seq.l, k, canonical); for (...) ... }
// On short reads a GPU round trip per view costs far more than the work it carries.  A read_pool reads AHEAD: it parses the next
// records of the file into one arena, uploads them as one batch, and hands out pointers into that arena.  When a view is then
// built on such a pointer, kmer_view finds the arena through the pool registry and takes its k-mers from ONE scan of the whole
// batch (run when the first view of the batch asks, with that view's k and canonical flag): the loop body stays unchanged,
//     biolib_amd::read_pool pool(path);
//     while (pool.next(s, len)) { auto view = wrapper::kmer_view_from_cstr<kmer_t>(s, len, k, canonical); for (...) ... }
// and pays one upload + one scan + one download per batch (default 32 Mbases) instead of per read.  The device array the scan
// writes and the page-locked host array it is downloaded into belong to the pool and are reused from batch to batch (a fresh
// 256 MB vector per batch costs a page fault per 4 KiB and a device-wide wait per hipFree: more than the scan).
#ifndef BIOLIB_AMD_COMPAT_READ_POOL_HPP
#define BIOLIB_AMD_COMPAT_READ_POOL_HPP

#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <utility>

#include "biolib_amd_runtime.hpp"

namespace biolib_amd {

class read_pool
{
    public:
        explicit read_pool(std::string const& path, uint64_t max_bases_per_batch = uint64_t(32) << 20) : max_bases(max_bases_per_batch)
        {
            check(bl_reader_open(path.c_str(), &reader), "bl_reader_open");
            next_in_chain = head();
            head() = this;
        }
        ~read_pool()
        {
            for (read_pool** p = &head(); *p; p = &(*p)->next_in_chain)
                if (*p == this) { *p = next_in_chain; break; }
            drop_batch();
            bl_reader_close(reader);
            if (d_values) bl_device_free(context::get(), d_values);
            if (h_values) bl_host_free(context::get(), h_values);
            for (uint64_t* d : {d_mvalues, d_mpositions, d_mhashes})
                if (d) bl_device_free(context::get(), d);
            for (uint64_t* h : {mscan.values, mscan.positions, mscan.hashes})
                if (h) bl_host_free(context::get(), h);
        }
        read_pool(read_pool const&) = delete;

        // next record of the file; the pointers stay valid until next() has returned the last record of the current batch and is called again
        bool next(char const*& seq, std::size_t& len, char const** name = nullptr)
        {
            if (at == n_seqs and not refill()) return false;
            seq = bases + offsets[at];
            len = static_cast<std::size_t>(offsets[at + 1] - offsets[at]);
            if (name) *name = bl_reader_last_name(reader, at);
            ++at;
            return true;
        }
        uint64_t batches_scanned() const noexcept {return scans;}

        // used by kmer_view: if [p, p + len) lies in the current batch of a live pool of this thread, the k-mer values of that stretch
        static uint64_t const* lookup(char const* p, std::size_t len, unsigned k, bool canonical)
        {
            for (read_pool* q = head(); q; q = q->next_in_chain)
                if (q->bases and p >= q->bases and p + len <= q->bases + q->n_bases) return q->values(k, canonical) + (p - q->bases);
            return nullptr;
        }

        // used by minimizer_view: if [p, p + len) lies in the current batch of a live pool of this thread, the minimizer records
        // (value, position relative to p, hash) of the windows inside that stretch, taken from ONE scan of the whole batch
        static bool lookup_minimizers(char const* p, std::size_t len, unsigned k, unsigned m, uint64_t seed, bool canonical, std::vector<uint64_t>& values,
                                      std::vector<uint64_t>& positions, std::vector<uint64_t>& hashes)
        {
            for (read_pool* q = head(); q; q = q->next_in_chain)
                if (q->bases and p >= q->bases and p + len <= q->bases + q->n_bases) {
                    q->minimizers(static_cast<uint64_t>(p - q->bases), len, k, m, seed, canonical, values, positions, hashes);
                    return true;
                }
            return false;
        }

        // used by super_kmer_view, in the same way: the super-k-mers that begin inside [p, p + len).  That view keeps its own
        // copy of the sequence and may outlive the batch it was cut from, so `copy` (its bytes) must still be what the pool holds.
        static bool lookup_super_kmers(char const* p, char const* copy, std::size_t len, unsigned k, unsigned m, uint64_t seed, bool canonical,
                                       std::vector<uint64_t>& minimizers, std::vector<uint64_t>& first_pos, std::vector<uint64_t>& hashes, std::vector<uint8_t>& mm_pos,
                                       std::vector<uint8_t>& sizes)
        {
            for (read_pool* q = head(); q; q = q->next_in_chain)
                if (q->bases and p >= q->bases and p + len <= q->bases + q->n_bases and std::memcmp(p, copy, len) == 0) {
                    q->super_kmers(static_cast<uint64_t>(p - q->bases), len, k, m, seed, canonical, minimizers, first_pos, hashes, mm_pos, sizes);
                    return true;
                }
            return false;
        }

    private:
        // the batch's super-k-mers for one (k, m, seed, canonical)
        struct super_kmer_scan {
            std::vector<uint64_t> minimizers, first_pos, hashes;
            std::vector<uint8_t> mm_pos, sizes;
            bool valid = false;
            std::tuple<unsigned, unsigned, uint64_t, bool> key {0u, 0u, 0ull, false};
        };
        super_kmer_scan sscan;

        void super_kmers(uint64_t first, std::size_t len, unsigned k, unsigned m, uint64_t seed, bool canonical, std::vector<uint64_t>& minimizers,
                         std::vector<uint64_t>& first_pos, std::vector<uint64_t>& hashes, std::vector<uint8_t>& mm_pos, std::vector<uint8_t>& sizes)
        {
            auto key = std::make_tuple(k, m, seed, canonical);
            if (not sscan.valid or sscan.key != key) {
                const std::size_t cap = n_bases + 64;
                device_array<uint64_t> dm(cap), df(cap), dh(cap);
                device_array<uint8_t> dp(cap), ds(cap);
                bl_result res;
                check(bl_scan_super_kmers(context::get(), batch, 0, 0, k, m, seed, (canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, dm.d, df.d, dp.d, ds.d, dh.d,
                                          cap, &res), "bl_scan_super_kmers");
                sscan.minimizers = dm.to_host(res.count);
                sscan.first_pos = df.to_host(res.count);
                sscan.hashes = dh.to_host(res.count);
                sscan.mm_pos = dp.to_host(res.count);
                sscan.sizes = ds.to_host(res.count);
                sscan.key = key;
                sscan.valid = true;
                ++scans;
            }
            auto lo = std::lower_bound(sscan.first_pos.begin(), sscan.first_pos.end(), first);
            auto hi = std::lower_bound(lo, sscan.first_pos.end(), first + static_cast<uint64_t>(len));
            const std::size_t a = static_cast<std::size_t>(lo - sscan.first_pos.begin()), n = static_cast<std::size_t>(hi - lo);
            minimizers.assign(sscan.minimizers.begin() + a, sscan.minimizers.begin() + a + n);
            hashes.assign(sscan.hashes.begin() + a, sscan.hashes.begin() + a + n);
            mm_pos.assign(sscan.mm_pos.begin() + a, sscan.mm_pos.begin() + a + n);
            sizes.assign(sscan.sizes.begin() + a, sscan.sizes.begin() + a + n);
            first_pos.resize(n);
            for (std::size_t i = 0; i < n; ++i) first_pos[i] = sscan.first_pos[a + i] - first;
        }

        // the batch's minimizer records for one (k, m, seed, canonical): page-locked host arrays that live with the pool
        struct minimizer_scan {
            uint64_t *values = nullptr, *positions = nullptr, *hashes = nullptr;
            uint64_t count = 0, cap = 0;
            bool valid = false;
            std::tuple<unsigned, unsigned, uint64_t, bool> key {0u, 0u, 0ull, false};
        };
        minimizer_scan mscan;
        uint64_t *d_mvalues = nullptr, *d_mpositions = nullptr, *d_mhashes = nullptr;
        uint64_t d_mcap = 0;

        void minimizers(uint64_t first, std::size_t len, unsigned k, unsigned m, uint64_t seed, bool canonical, std::vector<uint64_t>& values,
                        std::vector<uint64_t>& positions, std::vector<uint64_t>& hashes)
        {
            auto key = std::make_tuple(k, m, seed, canonical);
            if (not mscan.valid or mscan.key != key) {  // the first view of this batch (or other parameters): one scan for all its reads
                bl_ctx* c = context::get();
                const uint64_t cap = n_bases + 64;  // at most one record per position
                if (cap > d_mcap) {
                    for (uint64_t** d : {&d_mvalues, &d_mpositions, &d_mhashes}) {
                        if (*d) check(bl_device_free(c, *d), "bl_device_free");
                        void* p = nullptr;
                        check(bl_device_alloc(c, (cap + cap / 8) * sizeof(uint64_t), &p), "bl_device_alloc");
                        *d = static_cast<uint64_t*>(p);
                    }
                    d_mcap = cap + cap / 8;
                }
                bl_result res;
                check(bl_scan_minimizers(c, batch, 0, 0, m, k - m + 1, seed, (canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, d_mvalues, d_mpositions,
                                         d_mhashes, d_mcap, &res), "bl_scan_minimizers");
                if (res.count > mscan.cap) {
                    for (uint64_t** h : {&mscan.values, &mscan.positions, &mscan.hashes}) {
                        if (*h) check(bl_host_free(c, *h), "bl_host_free");
                        void* p = nullptr;
                        check(bl_host_alloc(c, (res.count + res.count / 8 + 64) * sizeof(uint64_t), &p), "bl_host_alloc");
                        *h = static_cast<uint64_t*>(p);
                    }
                    mscan.cap = res.count + res.count / 8 + 64;
                }
                check(bl_copy_to_host(c, mscan.values, d_mvalues, res.count * sizeof(uint64_t)), "bl_copy_to_host");
                check(bl_copy_to_host(c, mscan.positions, d_mpositions, res.count * sizeof(uint64_t)), "bl_copy_to_host");
                check(bl_copy_to_host(c, mscan.hashes, d_mhashes, res.count * sizeof(uint64_t)), "bl_copy_to_host");
                mscan.count = res.count;
                mscan.key = key;
                mscan.valid = true;
                ++scans;
            }
            // records are in position order: those of this read are the ones that start inside it (a window never spans two reads)
            uint64_t const* const all_end = mscan.positions + mscan.count;
            uint64_t const* lo = std::lower_bound(static_cast<uint64_t const*>(mscan.positions), all_end, first);
            uint64_t const* hi = std::lower_bound(lo, all_end, first + static_cast<uint64_t>(len));
            const std::size_t a = static_cast<std::size_t>(lo - mscan.positions), n = static_cast<std::size_t>(hi - lo);
            values.assign(mscan.values + a, mscan.values + a + n);
            hashes.assign(mscan.hashes + a, mscan.hashes + a + n);
            positions.resize(n);
            for (std::size_t i = 0; i < n; ++i) positions[i] = mscan.positions[a + i] - first;
        }

        bl_reader* reader = nullptr;
        bl_batch* batch = nullptr;
        char const* bases = nullptr;
        uint64_t const* offsets = nullptr;
        uint64_t n_seqs = 0, n_bases = 0, at = 0, max_bases, scans = 0;
        // per-position k-mer values of the batch: the first (k, canonical) asked for lives in the pool's own arrays, any further
        // one (unusual: two kinds of view over the same reads) in a vector of its own
        uint64_t* d_values = nullptr;
        uint64_t* h_values = nullptr;
        uint64_t values_cap = 0;
        bool have_primary = false;
        std::pair<unsigned, bool> primary_key {0u, false};
        std::map<std::pair<unsigned, bool>, std::vector<uint64_t>> cache;
        read_pool* next_in_chain = nullptr;

        static read_pool*& head() {thread_local read_pool* h = nullptr; return h;}

        void drop_batch()
        {
            if (batch) bl_batch_destroy(batch);
            batch = nullptr;
            bases = nullptr;
            cache.clear();
            have_primary = false;
            mscan.valid = false;
            sscan.valid = false;
            n_seqs = n_bases = at = 0;
        }
        bool refill()
        {
            drop_batch();
            uint64_t ns = 0, nb = 0;
            check(bl_reader_next_batch(context::get(), reader, max_bases, &batch, &ns, &nb), "bl_reader_next_batch");
            if (not batch) return false;
            check(bl_reader_last_batch(reader, &bases, &offsets, &n_seqs), "bl_reader_last_batch");
            n_bases = nb;
            return n_seqs != 0;
        }
        uint64_t const* values(unsigned k, bool canonical)
        {
            auto key = std::make_pair(k, canonical);
            if (have_primary and key == primary_key) return h_values;
            if (not have_primary) {
                if (n_bases > values_cap) {
                    if (d_values) check(bl_device_free(context::get(), d_values), "bl_device_free");
                    if (h_values) check(bl_host_free(context::get(), h_values), "bl_host_free");
                    d_values = h_values = nullptr;
                    values_cap = n_bases + n_bases / 8;
                    void* p = nullptr;
                    check(bl_device_alloc(context::get(), values_cap * sizeof(uint64_t), &p), "bl_device_alloc");
                    d_values = static_cast<uint64_t*>(p);
                    check(bl_host_alloc(context::get(), values_cap * sizeof(uint64_t), &p), "bl_host_alloc");
                    h_values = static_cast<uint64_t*>(p);
                }
                bl_result res;
                check(bl_scan_kmers(context::get(), batch, 0, 0, k, 0, (canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, d_values, nullptr, nullptr, &res),
                      "bl_scan_kmers");
                check(bl_copy_to_host(context::get(), h_values, d_values, n_bases * sizeof(uint64_t)), "bl_copy_to_host");
                have_primary = true;
                primary_key = key;
                ++scans;
                return h_values;
            }
            auto it = cache.find(key);
            if (it == cache.end()) {  // the first view of this batch with these parameters: one scan for all its reads
                device_array<uint64_t> d_values(n_bases);
                bl_result res;
                check(bl_scan_kmers(context::get(), batch, 0, 0, k, 0, (canonical ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, d_values.d, nullptr, nullptr, &res),
                      "bl_scan_kmers");
                it = cache.emplace(key, d_values.to_host(n_bases)).first;
                ++scans;
            }
            return it->second.data();
        }
};

}  // namespace biolib_amd

#endif
