// hash_sampler.hpp — drop-in for biolib's include/hash_sampler.hpp on top of the MI355X scan library.
//
// Same public surface (reference hash_sampler.hpp:9-78): sampler::hash_sampler<Iterator, HashFunctionFamily>
// (start, stop, hash, seed, sampling_rate), cbegin/cend/begin/end/get_sampling_rate; an element is kept when
// hash(*it, seed) < threshold, threshold = sampling_rate * max(hash_type) (:73-78, :136-141).
//  * Iterator = wrapper::kmer_view<K,It>::const_iterator with HashFunctionFamily = hash::hash64: evaluated on the
//    GPU (bl_scan_hash_sample); the sampler walks the resulting list.  As with the reference, the k-mer that ends the
//    sequence is outside [cbegin(), cend()) of a kmer_view (quirk Q1).
//  * any other pairing: element-by-element filter, exactly as the reference does.
#ifndef BIOLIB_AMD_COMPAT_HASH_SAMPLER_HPP
#define BIOLIB_AMD_COMPAT_HASH_SAMPLER_HPP

#include <limits>
#include <optional>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "kmer_view.hpp"

namespace sampler {

template <class Iterator, typename HashFunctionFamily>
class hash_sampler
{
    template <typename It, typename = void> struct has_view : std::false_type {};
    template <typename It> struct has_view<It, std::void_t<decltype(std::declval<It const&>().view()), decltype(std::declval<It const&>().chars_consumed())>> : std::true_type {};
    static constexpr bool gpu_path = has_view<Iterator>::value and std::is_same<HashFunctionFamily, hash::hash64>::value;
    using hash_type = typename HashFunctionFamily::hash_type;

    public:
        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = std::conditional_t<gpu_path, uint64_t, typename std::iterator_traits<Iterator>::value_type>;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(hash_sampler const& sampler, Iterator const& start, bool at_end) : parent_sampler(&sampler), itr_start(start), idx(0)
                {
                    if constexpr (gpu_path) idx = at_end ? sampler.kept().size() : 0;
                    else { (void)at_end; find_first_kmer(); }
                }
                auto operator*() const
                {
                    if constexpr (gpu_path) return parent_sampler->kept()[idx];
                    else return optional_unwrap(*itr_start);
                }
                const_iterator const& operator++()
                {
                    if constexpr (gpu_path) ++idx;
                    else { ++itr_start; find_first_kmer(); }
                    return *this;
                }
                const_iterator operator++(int) {auto current = *this; operator++(); return current;}

            private:
                hash_sampler const* parent_sampler;
                Iterator itr_start;
                std::size_t idx;
                void find_first_kmer() noexcept
                {
                    while (itr_start != parent_sampler->itr_stop and parent_sampler->mhash(*itr_start, parent_sampler->mseed) >= parent_sampler->threshold) ++itr_start;
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

        hash_sampler(Iterator const& start, Iterator const& stop, HashFunctionFamily const hash, uint64_t seed, double sampling_rate)
            : itr_start(start), itr_stop(stop), mhash(hash), mseed(seed), srate(sampling_rate)
        {
            if (srate > 1 or srate < 0) throw std::invalid_argument("[hash_sampler] Invalid sampling rate");
            // the reference casts rate * max to hash_type (:77), which overflows for rate = 1: saturate instead
            const double t = srate * static_cast<double>(std::numeric_limits<hash_type>::max());
            threshold = t >= static_cast<double>(std::numeric_limits<hash_type>::max()) ? std::numeric_limits<hash_type>::max() : static_cast<hash_type>(t);
        }
        const_iterator cbegin() const {return const_iterator(*this, itr_start, false);}
        const_iterator cend() const {return const_iterator(*this, itr_stop, true);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        double get_sampling_rate() const {return srate;}

    private:
        Iterator const itr_start;
        Iterator const itr_stop;
        HashFunctionFamily const mhash;
        uint64_t mseed;
        double srate;
        hash_type threshold;
        mutable std::shared_ptr<std::vector<uint64_t>> cache;

        std::vector<uint64_t> const& kept() const
        {
            if (cache) return *cache;
            auto out = std::make_shared<std::vector<uint64_t>>();
            if constexpr (gpu_path) {
                auto const* view = itr_start.view();
                std::string const& chars = view->chars();
                const unsigned k = view->get_k();
                const std::size_t first = itr_start.chars_consumed() >= k ? itr_start.chars_consumed() - k : 0;
                const std::size_t stop = itr_stop.chars_consumed() >= k ? itr_stop.chars_consumed() - k : 0;
                if (chars.size() >= k and stop > first) {
                    biolib_amd::batch_handle batch(chars.data(), chars.size());
                    const std::size_t cap = stop - first;
                    biolib_amd::device_array<uint64_t> dv(cap);
                    bl_result res;
                    biolib_amd::check(bl_scan_hash_sample(biolib_amd::context::get(), batch.b, first, stop - first, k, mseed, threshold,
                                                          (view->is_canonical() ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, dv.d, nullptr, nullptr, cap, &res),
                                      "bl_scan_hash_sample");
                    *out = dv.to_host(res.count);
                }
            }
            cache = out;
            return *cache;
        }

        friend bool operator==(hash_sampler const& a, hash_sampler const& b) {return a.itr_start == b.itr_start and a.itr_stop == b.itr_stop;}
        friend bool operator!=(hash_sampler const& a, hash_sampler const& b) {return not (a == b);}
};

}  // namespace sampler

#endif
