// minimizer_view.hpp — drop-in for biolib's include/minimizer_view.hpp on top of the MI355X scan library.
//
// REFUSED AT COMPILE TIME: a HashFunction other than hash::hash64 (the reference's template takes any functor with a static
// hash(T, uint64_t), minimizer_view.hpp:88-92) and a MinimizerType wider than 64 bits.  The device hashes with
// MurmurHash3_x64_128 and nothing else, and this header has no host evaluation to fall back on — by design: a view that silently
// ran on the CPU would be a different product.  Code that needs another functor has the same windows through
// sampler::minimizer_sampler over a kmer_view of the m-mers (minimizer_sampler.hpp evaluates any functor element by element
// on the host, as hash_sampler and syncmer_sampler do for pairings the GPU does not implement).  INTEGRATION.md lists this.
//
// Same public surface (reference minimizer_view.hpp:14-122): wrapper::minimizer_view<KmerType, MinimizerType,
// HashFunction, Iterator>(start, stop, k, m, seed, canonical), cbegin/cend/get_k/get_m,
// const_iterator::{operator* -> minimizer_context_t const&, ++, ++(int), break_offset},
// minimizer_view_from_string / minimizer_view_from_cstr.
// Semantics = the reference's INTENDED ones (its own iterator yields nothing): units are
// (canonical) m-mers hashed with HashFunction::hash(mmer, seed); a window is k-m+1 consecutive m-mers
// (= one k-mer); the minimizer is the leftmost minimum hash (strict '<', :283,374); one item each time the
// minimizer occurrence changes; a break restarts the window.  `position` is the 0-based start of the
// m-mer in the view (the reference codes a 1-based value that is never observable),
// `id` = that same m-mer ordinal.  HashFunction must be hash::hash64 (the device hash).
#ifndef BIOLIB_AMD_COMPAT_MINIMIZER_VIEW_HPP
#define BIOLIB_AMD_COMPAT_MINIMIZER_VIEW_HPP

#include <string>
#include <type_traits>

#include "biolib_amd_runtime.hpp"
#include "constants.hpp"
#include "hash.hpp"
#include "read_pool.hpp"

namespace wrapper {

template <typename KmerType, typename MinimizerType, typename HashFunction, typename Iterator>
class minimizer_view
{
    static_assert(std::is_same<HashFunction, hash::hash64>::value, "the GPU path implements hash::hash64 (MurmurHash3_x64_128, low 64 bits)");
    static_assert(sizeof(MinimizerType) <= 8, "minimizers are packed in 64 bits (m <= 32)");

    public:
        class const_iterator
        {
            public:
                struct minimizer_context_t {
                    MinimizerType value;   // 2-bit packed minimizer
                    std::size_t position;  // minimizer index from start
                    std::size_t id;        // unique id for current view
                };
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = minimizer_context_t;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(minimizer_view const* view) : parent_view(view), idx(0) {view->materialise(); load();}
                const_iterator(minimizer_view const* view, int /*dummy_end*/) : parent_view(view), idx(view->materialise()->values.size()) {}
                value_type const& operator*() const noexcept {return current;}
                const_iterator const& operator++() {++idx; load(); return *this;}
                const_iterator operator++(int) {auto res = *this; operator++(); return res;}
                // bases since the last break at the current minimizer's window (the reference exposes its rolling counter)
                std::size_t break_offset() const noexcept {return current.position;}

            private:
                minimizer_view const* parent_view;
                std::size_t idx;
                value_type current{};
                void load()
                {
                    auto const* m = parent_view->cache.get();
                    if (idx < m->values.size()) current = value_type{static_cast<MinimizerType>(m->values[idx]), static_cast<std::size_t>(m->positions[idx]), static_cast<std::size_t>(m->positions[idx])};
                }
                friend bool operator==(const_iterator const& a, const_iterator const& b) {return a.parent_view == b.parent_view and a.idx == b.idx;}
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        minimizer_view(Iterator start, Iterator stop, uint8_t k, uint8_t m, uint64_t seed, bool canonical = false)
            : itr_start(start), itr_stop(stop), klen(k), mlen(m), mseed(seed), canon(canonical)
        {
            if (m == 0 or m > 32 or k < m or k - m + 1 > 64) throw std::runtime_error("[minimizer view] need 1 <= m <= 32, m <= k, k-m+1 <= 64");
        }
        const_iterator cbegin() const {return const_iterator(this);}
        const_iterator cend() const {return const_iterator(this, 0);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        uint8_t get_k() const noexcept {return klen;}
        uint8_t get_m() const noexcept {return mlen;}

        // bulk access
        std::vector<uint64_t> const& values() const {return materialise()->values;}
        std::vector<uint64_t> const& positions() const {return materialise()->positions;}
        std::vector<uint64_t> const& hashes() const {return materialise()->hashes;}

    private:
        struct materialised {
            std::vector<uint64_t> values, positions, hashes;
        };
        Iterator itr_start;
        Iterator itr_stop;
        uint8_t klen;
        uint8_t mlen;
        uint64_t mseed;
        bool canon;
        mutable std::shared_ptr<materialised> cache;

        materialised const* materialise() const
        {
            if (cache) return cache.get();
            auto out = std::make_shared<materialised>();
            if constexpr (std::is_same<Iterator, char_iterator>::value) {
                // contiguous memory: is it a record a read_pool handed out?  then ONE scan of the pool's whole batch holds its minimizers
                const std::size_t len = static_cast<std::size_t>(itr_stop.base() - itr_start.base());
                if (biolib_amd::read_pool::lookup_minimizers(itr_start.base(), len, klen, mlen, mseed, canon, out->values, out->positions, out->hashes)) {
                    cache = out;
                    return cache.get();
                }
            }
            std::string chars;
            for (Iterator it = itr_start; it != itr_stop; ++it) chars.push_back(*it);
            const std::size_t n = chars.size();
            if (n >= klen) {
                biolib_amd::batch_handle batch(chars.data(), n);
                const std::size_t cap = n - klen + 1;  // at most one record per window
                biolib_amd::device_array<uint64_t> dv(cap), dp(cap), dh(cap);
                bl_result res;
                biolib_amd::check(bl_scan_minimizers(biolib_amd::context::get(), batch.b, 0, 0, mlen, klen - mlen + 1, mseed,
                                                     (canon ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, dv.d, dp.d, dh.d, cap, &res), "bl_scan_minimizers");
                out->values = dv.to_host(res.count);
                out->positions = dp.to_host(res.count);
                out->hashes = dh.to_host(res.count);
            }
            cache = out;
            return cache.get();
        }
};

template <typename KmerType, typename MmerType, typename HashFunction>
minimizer_view<KmerType, MmerType, HashFunction, std::string::const_iterator> minimizer_view_from_string(
    const std::string& s, uint8_t k, uint8_t m, uint64_t seed, bool canonical)
{
    return minimizer_view<KmerType, MmerType, HashFunction, std::string::const_iterator>(s.cbegin(), s.cend(), k, m, seed, canonical);
}

template <typename KmerType, typename MmerType, typename HashFunction>
minimizer_view<KmerType, MmerType, HashFunction, char_iterator> minimizer_view_from_cstr(
    char const* s, std::size_t len, uint8_t k, uint8_t m, uint64_t seed, bool canonical)
{
    return minimizer_view<KmerType, MmerType, HashFunction, char_iterator>(char_iterator(s), char_iterator(s + len), k, m, seed, canonical);
}

}  // namespace wrapper

#endif
