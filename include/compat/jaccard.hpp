// jaccard.hpp — drop-in for biolib's include/jaccard.hpp, plus the device form of the same computation.
//
// algorithm::jaccard(start1, stop1, start2, stop2) (reference jaccard.hpp:8-37): over two SORTED duplicate-free ranges,
// returns (|A n B|, |A u B|, |A|, |B|) — a two-finger walk, kept as a host template for arbitrary iterators.
// algorithm::jaccard_device(a, b): the same four numbers for two emem::external_memory_vector<uint64_t> (the containers the
// reference's Jaccard tool fills, tests/test_jaccard.cpp:55-130) computed on the GPU — run files merged on the device
// (bl_merge_runs_u64), duplicates removed (bl_sort_unique_u64), intersection counted (bl_jaccard_sorted_u64) — without ever
// walking the k-mers on the host.
#ifndef BIOLIB_AMD_COMPAT_JACCARD_HPP
#define BIOLIB_AMD_COMPAT_JACCARD_HPP

#include <tuple>

#include "external_memory_vector.hpp"

namespace algorithm {

template <typename Iterator1, typename Iterator2>
std::tuple<std::size_t, std::size_t, std::size_t, std::size_t> jaccard(Iterator1 start1, Iterator1 stop1, Iterator2 start2, Iterator2 stop2)
{
    std::size_t both = 0, size1 = 0, size2 = 0;
    while (start1 != stop1 and start2 != stop2) {
        auto const& x = *start1;
        auto const& y = *start2;
        if (x < y) {++start1; ++size1;}
        else if (y < x) {++start2; ++size2;}
        else {++start1; ++start2; ++size1; ++size2; ++both;}
    }
    for (; start1 != stop1; ++start1) ++size1;
    for (; start2 != stop2; ++start2) ++size2;
    return std::make_tuple(both, size1 + size2 - both, size1, size2);
}

inline std::tuple<std::size_t, std::size_t, std::size_t, std::size_t> jaccard_device(emem::external_memory_vector<uint64_t> const& a,
                                                                                       emem::external_memory_vector<uint64_t> const& b)
{
    auto da = a.to_device();  // sorted, duplicates kept
    auto db = b.to_device();
    uint64_t ua = 0, ub = 0, inter = 0, uni = 0;
    biolib_amd::check(bl_sort_unique_u64(biolib_amd::context::get(), da->d, a.size(), &ua), "bl_sort_unique_u64");
    biolib_amd::check(bl_sort_unique_u64(biolib_amd::context::get(), db->d, b.size(), &ub), "bl_sort_unique_u64");
    biolib_amd::check(bl_jaccard_sorted_u64(biolib_amd::context::get(), da->d, ua, db->d, ub, &inter, &uni), "bl_jaccard_sorted_u64");
    return std::make_tuple(static_cast<std::size_t>(inter), static_cast<std::size_t>(uni), static_cast<std::size_t>(ua), static_cast<std::size_t>(ub));
}

}  // namespace algorithm

#endif
