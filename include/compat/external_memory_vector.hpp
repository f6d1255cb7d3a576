// external_memory_vector.hpp — drop-in for biolib's include/external_memory_vector.hpp for the case the path uses:
// emem::external_memory_vector<uint64_t> (sorted, default order), the container the reference's tools push k-mers into
// (tests/test_jaccard.cpp:55-79) and then iterate in sorted order.
//
// Same public surface (reference external_memory_vector.hpp:29-136): (available_space_bytes, tmp_dir, name), push_back,
// cbegin/cend (forward iterator over the elements in sorted order, duplicates kept), size, minimize; and the same FILES:
// when the buffer is full it is sorted and written as a run file <tmp_dir>/tmp.run[_<name>]_<id>.bin of raw 8-byte values
// (:243-262), removed by the destructor — so a process built on the reference can read the runs this one spills and vice
// versa.  What differs is where the work happens: the buffer is sorted on the GPU (bl_sort_u64) and iteration does not run a
// heap over memory-mapped runs (:265-347) — the runs are merged on the device (bl_merge_runs_u64) and the iterator walks the
// merged array.  to_device() hands that array to device consumers (algorithm::jaccard_device) without a host round trip.
#ifndef BIOLIB_AMD_COMPAT_EXTERNAL_MEMORY_VECTOR_HPP
#define BIOLIB_AMD_COMPAT_EXTERNAL_MEMORY_VECTOR_HPP

#include <cstdio>
#include <iterator>
#include <string>
#include <type_traits>
#include <vector>

#include "biolib_amd_runtime.hpp"

namespace emem {

template <typename T, bool sorted = true>
class external_memory_vector
{
    static_assert(std::is_same<T, uint64_t>::value and sorted, "the GPU path keeps sorted vectors of uint64_t (packed k-mers)");

    public:
        using value_type = T;

        class const_iterator
        {
            public:
                using iterator_category = std::forward_iterator_tag;
                using difference_type   = std::ptrdiff_t;
                using value_type        = T;
                using pointer           = value_type*;
                using reference         = value_type&;

                const_iterator(external_memory_vector const* vec) : v(vec), merged(vec->merged_host()), idx(0) {}
                const_iterator(external_memory_vector const* vec, int /*dummy_end*/) : v(vec), idx(vec->size()) {}
                T const& operator*() const {return (*merged)[idx];}
                const_iterator const& operator++() {++idx; return *this;}
                const_iterator operator++(int) {auto current = *this; ++idx; return current;}
                bool operator==(const_iterator const& other) const {return v == other.v and idx == other.idx;}
                bool operator!=(const_iterator const& other) const {return not operator==(other);}

            private:
                external_memory_vector const* v;
                std::shared_ptr<std::vector<T>> merged;
                std::size_t idx;
        };

        external_memory_vector(uint64_t available_space_bytes, std::string tmp_dir, std::string name = "")
            : m_total_elems(0), m_tmp_dirname(tmp_dir), m_prefix(name)
        {
            if (available_space_bytes / sizeof(T) == 0) throw std::runtime_error("[EMV] Insufficient memory");
            m_buffer_size = available_space_bytes / sizeof(T) + 1;
            m_buffer.reserve(m_buffer_size);
        }
        external_memory_vector(external_memory_vector&&) = default;
        ~external_memory_vector() {for (auto const& f : m_tmp_files) std::remove(f.c_str());}

        void push_back(T const& elem)
        {
            m_buffer.push_back(elem);
            ++m_total_elems;
            m_merged.reset();
            if (m_buffer.size() >= m_buffer_size) sort_and_flush();
        }
        const_iterator cbegin() const
        {
            const_cast<external_memory_vector*>(this)->minimize();
            return const_iterator(this);
        }
        const_iterator cend() const {return const_iterator(this, 0);}
        std::size_t size() const {return m_total_elems;}
        void minimize()
        {
            if (not m_buffer.empty()) sort_and_flush();
            m_buffer.shrink_to_fit();
        }
        std::vector<std::string> const& run_files() const {return m_tmp_files;}

        // all elements, sorted (duplicates kept), in device memory: the k-way merge of the run files done on the GPU
        std::shared_ptr<biolib_amd::device_array<uint64_t>> to_device() const
        {
            const_cast<external_memory_vector*>(this)->minimize();
            auto out = std::make_shared<biolib_amd::device_array<uint64_t>>(m_total_elems ? m_total_elems : 1);
            std::vector<char const*> paths;
            for (auto const& f : m_tmp_files) paths.push_back(f.c_str());
            uint64_t total = 0;
            biolib_amd::check(bl_merge_runs_u64(biolib_amd::context::get(), paths.data(), static_cast<uint32_t>(paths.size()), out->d, m_total_elems, &total), "bl_merge_runs_u64");
            if (total != m_total_elems) throw std::runtime_error("[EMV] run files do not hold the elements pushed");
            return out;
        }

    private:
        std::size_t m_buffer_size;
        std::size_t m_total_elems;
        std::string m_tmp_dirname;
        std::string m_prefix;
        std::vector<std::string> m_tmp_files;
        std::vector<T> m_buffer;
        mutable std::shared_ptr<std::vector<T>> m_merged;

        std::shared_ptr<std::vector<T>> merged_host() const
        {
            if (not m_merged) m_merged = std::make_shared<std::vector<T>>(to_device()->to_host(m_total_elems));
            return m_merged;
        }
        void sort_and_flush()
        {
            biolib_amd::device_array<uint64_t> d(m_buffer.size());
            biolib_amd::check(bl_copy_to_device(biolib_amd::context::get(), d.d, m_buffer.data(), m_buffer.size() * sizeof(T)), "bl_copy_to_device");
            biolib_amd::check(bl_sort_u64(biolib_amd::context::get(), d.d, m_buffer.size()), "bl_sort_u64");
            char name[4096];
            biolib_amd::check(bl_run_file_name(m_tmp_dirname.c_str(), m_prefix.c_str(), m_tmp_files.size(), name, sizeof(name)), "bl_run_file_name");
            biolib_amd::check(bl_write_run_u64(biolib_amd::context::get(), d.d, m_buffer.size(), name), "bl_write_run_u64");
            m_tmp_files.push_back(name);
            m_buffer.clear();
        }
};

}  // namespace emem

#endif
