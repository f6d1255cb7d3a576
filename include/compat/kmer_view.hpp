// kmer_view.hpp — drop-in for biolib's include/kmer_view.hpp on top of the MI355X scan library.
//
// Same public surface (reference kmer_view.hpp:18-23, 25-94, 250-283):
//   wrapper::kmer_context_t<KmerType>, wrapper::kmer_view<KmerType, Iterator> with cbegin/cend/begin/end/get_k,
//   const_iterator::{operator*, ++, ++(int), get_mask, ==, !=}, kmer_view_from_string / kmer_view_from_cstr,
//   hash::minimizer_position_extractor.
// How it works: the first cbegin() of a view uploads the sequence and runs bl_scan_kmers (2-bit pack,
// rolling canonical k-mers on the GPU); the iterator then replays the reference's ITERATION PROTOCOL
// over the result — null items for breaks, position / id bookkeeping, and quirk Q1 (`it != cend()` stops
// before the last k-mer, which stays readable as *it; kmer_view.hpp:57,172-202).  Inputs on which the
// reference reads out of bounds (length < k, break followed by fewer than k bases at the end; Q2)
// terminate cleanly instead.  KmerType must fit 64 bits and k <= 32 (the reference's own limit for
// uint64_t, kmer_view.hpp:195).
// Views built on memory handed out by a biolib_amd::read_pool (read_pool.hpp) do not go to the GPU one by one: they
// index into the single scan of the pool's current batch.
#ifndef BIOLIB_AMD_COMPAT_KMER_VIEW_HPP
#define BIOLIB_AMD_COMPAT_KMER_VIEW_HPP

#include <cassert>
#include <limits>
#include <optional>
#include <string>
#include <type_traits>

#include "biolib_amd_runtime.hpp"
#include "read_pool.hpp"
#include "constants.hpp"
#include "hash.hpp"

namespace wrapper {

template <typename KmerType>
struct kmer_context_t {
    std::optional<KmerType> value;
    std::size_t position;  // position from start
    std::size_t id;        // unique id for current view
};

template <typename KmerType, class Iterator>
class kmer_view
{
    static_assert(std::is_same<typename Iterator::value_type, char>::value, "kmer_view iterates over char");
    static_assert(sizeof(KmerType) <= 8, "the GPU path packs k-mers in 64 bits (k <= 32)");

    struct materialised {
        std::string own_chars;             // host copy of [start, stop) (empty for pooled views)
        std::vector<uint64_t> own_values;  // per position: packed (canonical) k-mer, 0 where none starts
        char const* chars = nullptr;       // what the iterators read: the copies above, or a read_pool's arena and its batch scan
        uint64_t const* values = nullptr;
        std::size_t n = 0;
    };

    public:
        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = kmer_context_t<KmerType>;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(kmer_view const* view) noexcept(false) : parent_view(view), data(view->materialise()), consumed(0), run(0), kmer_count(0), dead(false)
                {
                    find_first_good_kmer();
                }
                const_iterator(kmer_view const* view, int /*dummy_end*/) noexcept : parent_view(view), data(nullptr), consumed(view->length()), run(0), kmer_count(0), dead(false) {}

                value_type operator*() const noexcept
                {
                    const std::size_t k = parent_view->klen;
                    if (run == 0) return value_type{std::nullopt, consumed - k, kmer_count};
                    return value_type{static_cast<KmerType>(data->values[consumed - k]), consumed - k, kmer_count};
                }

                const_iterator const& operator++()
                {
                    ++kmer_count;  // ids count null items too (reference :186)
                    if (run == 0) {
                        find_first_good_kmer();
                        return *this;
                    }
                    push();
                    return *this;
                }
                const_iterator operator++(int) {auto res = *this; operator++(); return res;}

                KmerType get_mask() const noexcept
                {
                    const unsigned k = parent_view->klen;
                    if (2 * k != sizeof(KmerType) * 8) return static_cast<KmerType>((KmerType(1) << (2 * k)) - 1);
                    return std::numeric_limits<KmerType>::max();
                }

                // position of the underlying char iterator, for samplers that run the GPU path themselves
                kmer_view const* view() const noexcept {return parent_view;}
                std::size_t chars_consumed() const noexcept {return consumed;}

            private:
                kmer_view const* parent_view;
                materialised const* data;
                std::size_t consumed;    // chars read so far (the reference's `position`)
                std::size_t run;         // bases_since_last_break
                std::size_t kmer_count;  // id
                bool dead;

                void push()
                {
                    const auto c = constants::seq_nt4_table[static_cast<uint8_t>(data->chars[consumed++])];
                    if (c < 4) ++run; else run = 0;
                }
                void find_first_good_kmer()
                {
                    const std::size_t n = parent_view->length(), k = parent_view->klen;
                    while (consumed != n && run < k) push();
                    if (run < k) {  // reference would step past the end here (Q2): finish cleanly
                        run = 0;
                        consumed = n;
                        dead = true;
                    }
                }
                // the reference compares the char iterators only (:57)
                friend bool operator==(const_iterator const& a, const_iterator const& b) {return a.parent_view == b.parent_view and a.consumed == b.consumed;}
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        kmer_view(Iterator start, Iterator stop, uint8_t k, bool canonical = false) : itr_start(start), itr_stop(stop), klen(k), canon(canonical)
        {
            if (k == 0 or k > 32) throw std::runtime_error("[k-mer view] k must be in [1, 32] for 64-bit k-mers");
        }
        const_iterator cbegin() const {return const_iterator(this);}
        const_iterator cend() const noexcept {return const_iterator(this, 0);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const noexcept {return cend();}
        uint8_t get_k() const noexcept {return klen;}
        bool is_canonical() const noexcept {return canon;}

        // bulk access: the whole view as arrays (what a GPU-aware caller should use instead of iterating)
        std::string const& chars() const
        {
            auto const* m = materialise();
            if (m->own_chars.size() != m->n) cache->own_chars.assign(m->chars, m->n);  // pooled view: copy on demand
            return cache->own_chars;
        }
        std::vector<uint64_t> const& values() const
        {
            auto const* m = materialise();
            if (m->own_values.size() != m->n) cache->own_values.assign(m->values, m->values + m->n);
            return cache->own_values;
        }

    private:
        Iterator itr_start;
        Iterator itr_stop;
        uint8_t klen;
        bool canon;
        mutable std::shared_ptr<materialised> cache;

        std::size_t length() const
        {
            if (cache) return cache->n;
            if constexpr (std::is_same<Iterator, char_iterator>::value) return static_cast<std::size_t>(itr_stop.base() - itr_start.base());
            std::size_t n = 0;
            for (Iterator it = itr_start; it != itr_stop; ++it) ++n;
            return n;
        }

        materialised const* materialise() const
        {
            if (cache) return cache.get();
            auto m = std::make_shared<materialised>();
            if constexpr (std::is_same<Iterator, char_iterator>::value) {
                // contiguous memory: is it a record a read_pool handed out?  then the batch scan already holds its k-mers
                const std::size_t n = static_cast<std::size_t>(itr_stop.base() - itr_start.base());
                if (uint64_t const* pooled = biolib_amd::read_pool::lookup(itr_start.base(), n, klen, canon)) {
                    m->chars = itr_start.base();
                    m->values = pooled;
                    m->n = n;
                    cache = m;
                    return cache.get();
                }
            }
            for (Iterator it = itr_start; it != itr_stop; ++it) m->own_chars.push_back(*it);
            const std::size_t n = m->own_chars.size();
            m->own_values.assign(n, 0);
            if (n >= klen) {
                biolib_amd::batch_handle batch(m->own_chars.data(), n);
                biolib_amd::device_array<uint64_t> d_values(n);
                bl_result res;
                biolib_amd::check(bl_scan_kmers(biolib_amd::context::get(), batch.b, 0, 0, klen, 0, (canon ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC,
                                                d_values.d, nullptr, nullptr, &res), "bl_scan_kmers");
                m->own_values = d_values.to_host(n);
            }
            m->chars = m->own_chars.data();
            m->values = m->own_values.data();
            m->n = n;
            cache = m;
            return cache.get();
        }

        friend bool operator==(kmer_view const& a, kmer_view const& b)
        {
            return a.itr_start == b.itr_start and a.itr_stop == b.itr_stop and a.klen == b.klen and a.canon == b.canon;
        }
        friend bool operator!=(kmer_view const& a, kmer_view const& b) {return not (a == b);}
};

template <typename KmerType>
kmer_view<KmerType, std::string::const_iterator> kmer_view_from_string(const std::string& s, uint8_t k, bool canonical)
{
    return kmer_view<KmerType, std::string::const_iterator>(s.cbegin(), s.cend(), k, canonical);
}

template <typename KmerType>
kmer_view<KmerType, char_iterator> kmer_view_from_cstr(char const* s, std::size_t len, uint8_t k, bool canonical)
{
    return kmer_view<KmerType, char_iterator>(char_iterator(s), char_iterator(s + len), k, canonical);
}

}  // namespace wrapper

namespace hash {

// The syncmer predicate on ONE packed k-mer (reference kmer_view.hpp:250-283, src/kmer_view.cpp:7-15):
// offset from the k-mer's left end of its leftmost minimum-hash m-mer, hash64 with seed 0; klen + 1
// for a null k-mer.  A scalar host function by nature (it takes a single value); the bulk form over a
// whole sequence is sampler::syncmer_sampler / bl_scan_syncmers.
class minimizer_position_extractor
{
    public:
        using value_type = uint64_t;
        minimizer_position_extractor(uint8_t k, uint8_t m) : klen(k), mlen(m)
        {
            assert(m <= k);
            mask = (2 * m != 64) ? ((uint64_t(1) << (2 * m)) - 1) : std::numeric_limits<uint64_t>::max();
        }
        template <typename KmerType>
        std::size_t operator()(wrapper::kmer_context_t<KmerType> const& kmer) const noexcept
        {
            if (!kmer.value) return klen + 1;
            // the k-mer packs its first base in the most significant pair: the m-mer at offset o from the left end is the 2m bits
            // that start 2 * (k - m - o) bits up.  Left to right with a strict '<': the leftmost of equal minima stays.
            const uint64_t packed = static_cast<uint64_t>(*kmer.value);
            const unsigned last = static_cast<unsigned>(klen - mlen);
            unsigned best = 0;
            uint64_t best_hash = bl_hash64_u64((packed >> (2 * last)) & mask, 0);
            for (unsigned o = 1; o <= last; ++o) {
                const uint64_t h = bl_hash64_u64((packed >> (2 * (last - o))) & mask, 0);
                if (h < best_hash) {
                    best_hash = h;
                    best = o;
                }
            }
            return best;
        }
        uint8_t get_k() const noexcept {return klen;}
        uint8_t get_m() const noexcept {return mlen;}

    private:
        uint8_t klen;
        uint8_t mlen;
        uint64_t mask;
};

}  // namespace hash

#endif
