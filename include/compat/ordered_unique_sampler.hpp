// ordered_unique_sampler.hpp — drop-in for biolib's include/ordered_unique_sampler.hpp.
//
// Same public surface (reference ordered_unique_sampler.hpp:11-65): sampler::ordered_unique_sampler<Iterator>(start, stop),
// cbegin/cend/begin/end/size; over a SORTED range it yields every distinct value once (:115-130): operator++ skips the
// elements equal to the current one, and size() is known once an iterator has reached the end.
// This sampler is a host-side cursor by nature; the work that matters sits below it: ranges that come from the drop-in
// emem::external_memory_vector (external_memory_vector.hpp) were sorted and merged on the GPU, and
// algorithm::jaccard_device (jaccard.hpp) runs unique + intersection there without walking the elements at all.
#ifndef BIOLIB_AMD_COMPAT_ORDERED_UNIQUE_SAMPLER_HPP
#define BIOLIB_AMD_COMPAT_ORDERED_UNIQUE_SAMPLER_HPP

#include <iterator>
#include <mutex>
#include <optional>

namespace sampler {

template <class Iterator>
class ordered_unique_sampler
{
    public:
        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = typename Iterator::value_type;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(ordered_unique_sampler const& sampler, Iterator const& start) : parent_sampler(&sampler), itr_start(start), unique_count(0) {}
                value_type const& operator*() const {return *itr_start;}
                const_iterator const& operator++()
                {
                    if (itr_start != parent_sampler->itr_stop) {
                        const value_type seen = *itr_start;
                        do ++itr_start; while (itr_start != parent_sampler->itr_stop and *itr_start == seen);
                        ++unique_count;
                    }
                    if (itr_start == parent_sampler->itr_stop) {
                        std::lock_guard<std::mutex> lock(parent_sampler->size_guard);
                        parent_sampler->known_size = unique_count;
                    }
                    return *this;
                }
                const_iterator operator++(int) {auto current = *this; operator++(); return current;}

            private:
                ordered_unique_sampler const* parent_sampler;
                Iterator itr_start;
                std::size_t unique_count;
                friend bool operator==(const_iterator const& a, const_iterator const& b) {return a.parent_sampler == b.parent_sampler and a.itr_start == b.itr_start;}
                friend bool operator!=(const_iterator const& a, const_iterator const& b) {return not (a == b);}
        };

        ordered_unique_sampler(Iterator const& start, Iterator const& stop) : itr_start(start), itr_stop(stop) {}
        const_iterator cbegin() const {return const_iterator(*this, itr_start);}
        const_iterator cend() const {return const_iterator(*this, itr_stop);}
        const_iterator begin() const {return cbegin();}
        const_iterator end() const {return cend();}
        std::optional<std::size_t> size() const {return known_size;}

    private:
        Iterator const itr_start;
        Iterator const itr_stop;
        mutable std::optional<std::size_t> known_size;
        mutable std::mutex size_guard;

        friend bool operator==(ordered_unique_sampler const& a, ordered_unique_sampler const& b) {return a.itr_start == b.itr_start and a.itr_stop == b.itr_stop;}
        friend bool operator!=(ordered_unique_sampler const& a, ordered_unique_sampler const& b) {return not (a == b);}
};

}  // namespace sampler

#endif
