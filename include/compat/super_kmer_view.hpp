// super_kmer_view.hpp — drop-in for biolib's include/super_kmer_view.hpp on top of the MI355X scan library.
//
// REFUSED AT COMPILE TIME, like minimizer_view.hpp and for the same reason: a HashFunction other than hash::hash64, a
// MinimizerType wider than 64 bits.  There is no host evaluation behind this view.
//
// Same public surface (reference super_kmer_view.hpp:11-58): wrapper::super_kmer_view<KmerType, MinimizerType,
// HashFunction>(contig, len, k, m, canonical) / (std::string, k, m, canonical), cbegin/cend/get_k/get_m,
// value_type super_kmer_t{minimizer, mm_pos, size}.  The reference header does not compile;
// this implements its intent (:121-135): maximal groups of consecutive k-mers that share one minimizer
// occurrence; mm_pos = offset of the minimizer in the group's first k-mer (:132), size = number of k-mers
// (:133).  A trailing `seed` argument (default 0) is added because the reference forwards none to its
// minimizer view.  Extra fields position (first k-mer) and hash are carried for bulk consumers.
#ifndef BIOLIB_AMD_COMPAT_SUPER_KMER_VIEW_HPP
#define BIOLIB_AMD_COMPAT_SUPER_KMER_VIEW_HPP

#include <string>
#include <type_traits>

#include "biolib_amd_runtime.hpp"
#include "hash.hpp"
#include "read_pool.hpp"

namespace wrapper {

template <typename KmerType, typename MinimizerType, typename HashFunction>
class super_kmer_view
{
    static_assert(std::is_same<HashFunction, hash::hash64>::value, "the GPU path implements hash::hash64");
    static_assert(sizeof(MinimizerType) <= 8, "minimizers are packed in 64 bits (m <= 32)");

    public:
        class const_iterator
        {
            public:
                struct super_kmer_t {
                    MinimizerType minimizer;  // 2-bit packed minimizer
                    uint8_t mm_pos;           // position of the minimizer in the first k-mer
                    uint8_t size;             // super k-mer size (number of k-mers)
                    std::size_t position;     // start of the first k-mer (extension)
                };
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = super_kmer_t;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(super_kmer_view const* view) : parent_view(view), idx(0) {view->materialise(); load();}
                const_iterator(super_kmer_view const* view, int /*dummy_end*/) : parent_view(view), idx(view->materialise()->minimizers.size()) {}
                super_kmer_t const& operator*() const {return current_sk;}
                const_iterator const& operator++() {++idx; load(); return *this;}
                const_iterator operator++(int) {auto res = *this; operator++(); return res;}

            private:
                super_kmer_view const* parent_view;
                std::size_t idx;
                super_kmer_t current_sk{};
                void load()
                {
                    auto const* m = parent_view->cache.get();
                    if (idx < m->minimizers.size())
                        current_sk = super_kmer_t{static_cast<MinimizerType>(m->minimizers[idx]), m->mm_pos[idx], m->sizes[idx], static_cast<std::size_t>(m->first_pos[idx])};
                }
                friend bool operator==(const_iterator const& a, const_iterator const& b) {return a.parent_view == b.parent_view and a.idx == b.idx;}
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        super_kmer_view(char const* contig, std::size_t contig_len, uint8_t k, uint8_t m, bool canonical = false, uint64_t seed = 0)
            : seq(contig, contig_len), origin(contig), klen(k), mlen(m), canon(canonical), mseed(seed) {validate();}
        super_kmer_view(std::string const& contig, uint8_t k, uint8_t m, bool canonical = false, uint64_t seed = 0)
            : seq(contig), klen(k), mlen(m), canon(canonical), mseed(seed) {validate();}
        const_iterator cbegin() const {return const_iterator(this);}
        const_iterator cend() const {return const_iterator(this, 0);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        uint8_t get_k() const noexcept {return klen;}
        uint8_t get_m() const noexcept {return mlen;}

    private:
        struct materialised {
            std::vector<uint64_t> minimizers, first_pos, hashes;
            std::vector<uint8_t> mm_pos, sizes;
        };
        std::string seq;
        char const* origin = nullptr;  // where the contig was handed over from: a read_pool's arena, perhaps
        uint8_t klen, mlen;
        bool canon;
        uint64_t mseed;
        mutable std::shared_ptr<materialised> cache;

        void validate() const
        {
            if (mlen == 0 or mlen > 32 or klen < mlen or klen - mlen + 1 > 64) throw std::runtime_error("[super-k-mer view] need 1 <= m <= 32, m <= k, k-m+1 <= 64");
        }

        materialised const* materialise() const
        {
            if (cache) return cache.get();
            auto out = std::make_shared<materialised>();
            const std::size_t n = seq.size();
            // a record a read_pool handed out?  then ONE scan of the pool's whole batch holds its super-k-mers
            if (origin and biolib_amd::read_pool::lookup_super_kmers(origin, seq.data(), n, klen, mlen, mseed, canon, out->minimizers, out->first_pos, out->hashes, out->mm_pos,
                                                                     out->sizes)) {
                cache = out;
                return cache.get();
            }
            if (n >= klen) {
                biolib_amd::batch_handle batch(seq.data(), n);
                const std::size_t cap = n - klen + 1;
                biolib_amd::device_array<uint64_t> dm(cap), df(cap), dh(cap);
                biolib_amd::device_array<uint8_t> dp(cap), ds(cap);
                bl_result res;
                biolib_amd::check(bl_scan_super_kmers(biolib_amd::context::get(), batch.b, 0, 0, klen, mlen, mseed, (canon ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC,
                                                      dm.d, df.d, dp.d, ds.d, dh.d, cap, &res), "bl_scan_super_kmers");
                out->minimizers = dm.to_host(res.count);
                out->first_pos = df.to_host(res.count);
                out->hashes = dh.to_host(res.count);
                out->mm_pos = dp.to_host(res.count);
                out->sizes = ds.to_host(res.count);
            }
            cache = out;
            return cache.get();
        }
};

}  // namespace wrapper

#endif
