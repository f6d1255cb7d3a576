// syncmer_sampler.hpp — drop-in for biolib's include/syncmer_sampler.hpp on top of the MI355X scan library.
//
// Same public surface (reference syncmer_sampler.hpp:9-73): sampler::syncmer_sampler<Iterator, PropertyExtractor>
// (start, stop, extractor, start_offset, end_offset), cbegin/cend/begin/end/get_offsets; elements are kept when
// extractor(*it) equals one of the two offsets (:130-137).
//  * Generic Iterator / PropertyExtractor: the filter is applied element by element, exactly as the reference does.
//  * Iterator = wrapper::kmer_view<K,It>::const_iterator with PropertyExtractor = hash::minimizer_position_extractor
//    (the only pairing the reference's own code targets): the whole range is evaluated on the GPU with
//    bl_scan_syncmers (2 hashes per base instead of (k-m+1)+1 per k-mer) and the sampler walks the resulting
//    position list.  Like the reference, a range [cbegin(), cend()) of a kmer_view never includes the k-mer that
//    ends the sequence (quirk Q1).  operator* yields the k-mer VALUE (PropertyExtractor::value_type); in the
//    reference that expression does not compile for this pairing (:104-108).
#ifndef BIOLIB_AMD_COMPAT_SYNCMER_SAMPLER_HPP
#define BIOLIB_AMD_COMPAT_SYNCMER_SAMPLER_HPP

#include <optional>
#include <type_traits>
#include <utility>
#include <vector>

#include "kmer_view.hpp"

namespace sampler {

namespace detail {
// kmer_view iterators advertise themselves through a member function
template <typename It, typename = void> struct has_view : std::false_type {};
template <typename It> struct has_view<It, std::void_t<decltype(std::declval<It const&>().view()), decltype(std::declval<It const&>().chars_consumed())>> : std::true_type {};
}  // namespace detail

template <class Iterator, typename PropertyExtractor>
class syncmer_sampler
{
    static constexpr bool gpu_path = detail::has_view<Iterator>::value and std::is_same<PropertyExtractor, hash::minimizer_position_extractor>::value;

    public:
        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = typename PropertyExtractor::value_type;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(syncmer_sampler const& sampler, Iterator const& start, bool at_end) : parent_sampler(&sampler), itr_start(start), idx(0)
                {
                    if constexpr (gpu_path) {
                        auto const& pos = sampler.positions();
                        idx = at_end ? pos.size() : 0;
                    } else {
                        (void)at_end;
                        find_first_syncmer();
                    }
                }
                value_type operator*() const
                {
                    if constexpr (gpu_path) {
                        auto const* view = parent_sampler->itr_start.view();
                        return static_cast<value_type>(view->values()[parent_sampler->positions()[idx]]);
                    } else {
                        return optional_unwrap(*itr_start);
                    }
                }
                // position of the current syncmer in the sequence (GPU path only)
                std::size_t position() const {static_assert(gpu_path, "positions exist on the kmer_view path"); return parent_sampler->positions()[idx];}
                const_iterator const& operator++()
                {
                    if constexpr (gpu_path) ++idx;
                    else { ++itr_start; find_first_syncmer(); }
                    return *this;
                }
                const_iterator operator++(int) {auto current = *this; operator++(); return current;}

            private:
                syncmer_sampler const* parent_sampler;
                Iterator itr_start;
                std::size_t idx;

                void find_first_syncmer() noexcept
                {
                    std::size_t pos;
                    while (itr_start != parent_sampler->itr_stop and
                           ((pos = parent_sampler->extor(*itr_start)) != parent_sampler->soffset and pos != parent_sampler->eoffset)) ++itr_start;
                }
                template <typename T> static T optional_unwrap(T const& val) noexcept {return val;}
                template <typename T> static T optional_unwrap(std::optional<T> const& opt) noexcept {return *opt;}

                friend bool operator==(const_iterator const& a, const_iterator const& b)
                {
                    if constexpr (gpu_path) return a.parent_sampler == b.parent_sampler and a.idx == b.idx;
                    else return a.parent_sampler == b.parent_sampler and a.itr_start == b.itr_start;
                }
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        syncmer_sampler(Iterator const& start, Iterator const& stop, PropertyExtractor const& extractor, uint16_t start_offset, uint16_t end_offset)
            : itr_start(start), itr_stop(stop), extor(extractor), soffset(start_offset), eoffset(end_offset) {}
        const_iterator cbegin() const {return const_iterator(*this, itr_start, false);}
        const_iterator cend() const {return const_iterator(*this, itr_stop, true);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        std::pair<uint16_t, uint16_t> get_offsets() const {return std::make_pair(soffset, eoffset);}

        // GPU path: number of syncmers in the range without walking it
        std::size_t count() const {static_assert(gpu_path, "count() exists on the kmer_view path"); return positions().size();}

    private:
        Iterator const itr_start;
        Iterator const itr_stop;
        PropertyExtractor const& extor;  // held by reference, as in the reference (:58)
        uint16_t soffset;
        uint16_t eoffset;
        mutable std::shared_ptr<std::vector<uint64_t>> cache;

        // positions of the syncmers among the k-mers the range [itr_start, itr_stop) covers
        std::vector<uint64_t> const& positions() const
        {
            if (cache) return *cache;
            auto out = std::make_shared<std::vector<uint64_t>>();
            if constexpr (gpu_path) {
                auto const* view = itr_start.view();
                std::string const& chars = view->chars();
                const unsigned k = extor.get_k(), m = extor.get_m();
                if (view->get_k() != k) throw std::runtime_error("[syncmer sampler] extractor k differs from the k-mer view's k");
                // the iterator's current k-mer starts at chars_consumed - k; the stop iterator's k-mer is excluded (Q1)
                const std::size_t first = itr_start.chars_consumed() >= k ? itr_start.chars_consumed() - k : 0;
                const std::size_t stop = itr_stop.chars_consumed() >= k ? itr_stop.chars_consumed() - k : 0;
                if (chars.size() >= k and stop > first) {
                    biolib_amd::batch_handle batch(chars.data(), chars.size());
                    const std::size_t cap = stop - first;
                    biolib_amd::device_array<uint64_t> dp(cap);
                    bl_result res;
                    biolib_amd::check(bl_scan_syncmers(biolib_amd::context::get(), batch.b, first, stop - first, k, m, soffset, eoffset, 0,
                                                       (view->is_canonical() ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, dp.d, cap, &res), "bl_scan_syncmers");
                    *out = dp.to_host(res.count);
                }
            }
            cache = out;
            return *cache;
        }

        friend bool operator==(syncmer_sampler const& a, syncmer_sampler const& b) {return a.itr_start == b.itr_start and a.itr_stop == b.itr_stop;}
        friend bool operator!=(syncmer_sampler const& a, syncmer_sampler const& b) {return not (a == b);}
};

}  // namespace sampler

#endif
