// minimizer_sampler.hpp — drop-in for biolib's include/minimizer_sampler.hpp on top of the MI355X scan library.
//
// Same public surface (reference minimizer_sampler.hpp:12-70): sampler::minimizer_sampler<Iterator, HashFunctionFamily>
// (start, stop, hash, seed, window_size), cbegin/cend/get_w; the literal "k = 31, w = 11 over a k-mer iterator" entry point of
// the path: the sampled elements are the window minimizers of the underlying range under hash(item, seed).
// Semantics = the reference's INTENDED ones (its header does not compile, :46): a window is `window_size` consecutive non-null
// items, a null item (sequence break) restarts it (:118-119,143-147), the minimum is replaced on strict '>' only, so the
// LEFTMOST minimum wins ties (:121,162-165); one element each time the minimizer occurrence changes.
//  * Iterator = wrapper::kmer_view<K,It>::const_iterator with HashFunctionFamily = hash::hash64: the whole range is evaluated
//    on the GPU by bl_scan_minimizers(unit = k, w = window_size); operator* yields the minimizer's kmer_context_t
//    (value, position of the k-mer in the view, id = position).  As with the reference, the k-mer that ends the sequence is
//    outside [cbegin(), cend()) of a kmer_view (quirk Q1) and takes no part in any window.
//  * any other pairing: evaluated element by element on the host with the same rules (items must be optional-like).
#ifndef BIOLIB_AMD_COMPAT_MINIMIZER_SAMPLER_HPP
#define BIOLIB_AMD_COMPAT_MINIMIZER_SAMPLER_HPP

#include <iterator>
#include <optional>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "kmer_view.hpp"

namespace sampler {

template <class Iterator, typename HashFunctionFamily>
class minimizer_sampler
{
    template <typename It, typename = void> struct has_view : std::false_type {};
    template <typename It> struct has_view<It, std::void_t<decltype(std::declval<It const&>().view()), decltype(std::declval<It const&>().chars_consumed())>> : std::true_type {};
    static constexpr bool gpu_path = has_view<Iterator>::value and std::is_same<HashFunctionFamily, hash::hash64>::value;

    public:
        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = typename std::iterator_traits<Iterator>::value_type;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(minimizer_sampler const& sampler, bool at_end) : parent_sampler(&sampler), idx(at_end ? sampler.sampled().size() : 0) {}
                value_type const& operator*() const {return parent_sampler->sampled()[idx];}
                const_iterator const& operator++() {++idx; return *this;}
                const_iterator operator++(int) {auto current = *this; operator++(); return current;}

            private:
                minimizer_sampler const* parent_sampler;
                std::size_t idx;
                friend bool operator==(const_iterator const& a, const_iterator const& b) {return a.parent_sampler == b.parent_sampler and a.idx == b.idx;}
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        minimizer_sampler(Iterator const& start, Iterator const& stop, HashFunctionFamily hash, uint64_t seed, uint16_t window_size)
            : itr_start(start), itr_stop(stop), mhash(hash), mseed(seed), w(window_size)
        {
            if (w == 0 or w > 64) throw std::invalid_argument("[minimizer_sampler] window size must be in [1, 64]");
        }
        const_iterator cbegin() const {return const_iterator(*this, false);}
        const_iterator cend() const {return const_iterator(*this, true);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        uint16_t get_w() const {return w;}

    private:
        using item_type = typename std::iterator_traits<Iterator>::value_type;
        Iterator const itr_start;
        Iterator const itr_stop;
        HashFunctionFamily mhash;
        uint64_t mseed;
        uint16_t w;
        mutable std::shared_ptr<std::vector<item_type>> cache;

        std::vector<item_type> const& sampled() const
        {
            if (cache) return *cache;
            auto out = std::make_shared<std::vector<item_type>>();
            if constexpr (gpu_path) {
                auto const* view = itr_start.view();
                std::string const& chars = view->chars();
                const unsigned k = view->get_k();
                // k-mers of the range: start positions [first, stop); windows must lie inside it entirely
                const std::size_t first = itr_start.chars_consumed() >= k ? itr_start.chars_consumed() - k : 0;
                const std::size_t stop = itr_stop.chars_consumed() >= k ? itr_stop.chars_consumed() - k : 0;
                if (chars.size() >= k and stop >= first + w) {
                    // the scan reads units beyond its range as windows need them, so the batch ends where the range ends:
                    // the last k-mer taken starts at stop - 1 and ends at stop + k - 2
                    biolib_amd::batch_handle batch(chars.data(), stop + k - 1);
                    const std::size_t cap = stop - first;
                    biolib_amd::device_array<uint64_t> dv(cap), dp(cap);
                    bl_result res;
                    biolib_amd::check(bl_scan_minimizers(biolib_amd::context::get(), batch.b, first, stop - first, k, w, mseed,
                                                         (view->is_canonical() ? (uint32_t)BL_FLAG_CANONICAL : 0u) | BL_FLAG_SYNC, dv.d, dp.d, nullptr, cap, &res),
                                      "bl_scan_minimizers");
                    auto values = dv.to_host(res.count);
                    auto positions = dp.to_host(res.count);
                    out->reserve(res.count);
                    for (std::size_t i = 0; i < res.count; ++i)
                        out->push_back(item_type{static_cast<typename decltype(item_type::value)::value_type>(values[i]), static_cast<std::size_t>(positions[i]),
                                                 static_cast<std::size_t>(positions[i])});
                }
            } else {
                // host evaluation of the same rules over optional-like items
                std::vector<item_type> window;
                std::vector<typename HashFunctionFamily::hash_type> hashes;
                std::size_t last_emitted = static_cast<std::size_t>(-1), index = 0, run_start = 0;
                for (Iterator it = itr_start; it != itr_stop; ++it, ++index) {
                    item_type item = *it;
                    if (not item) {
                        window.clear();
                        hashes.clear();
                        run_start = index + 1;
                        continue;
                    }
                    window.push_back(item);
                    hashes.push_back(mhash(*item, mseed));
                    if (window.size() < w) continue;
                    const std::size_t lo = window.size() - w;
                    std::size_t arg = lo;
                    for (std::size_t j = lo + 1; j < window.size(); ++j)
                        if (hashes[arg] > hashes[j]) arg = j;  // strict: the leftmost minimum stays
                    const std::size_t global = run_start + arg;
                    if (global != last_emitted) {
                        out->push_back(window[arg]);
                        last_emitted = global;
                    }
                }
            }
            cache = out;
            return *cache;
        }

        friend bool operator==(minimizer_sampler const& a, minimizer_sampler const& b) {return a.itr_start == b.itr_start and a.itr_stop == b.itr_stop;}
        friend bool operator!=(minimizer_sampler const& a, minimizer_sampler const& b) {return not (a == b);}
};

}  // namespace sampler

#endif
