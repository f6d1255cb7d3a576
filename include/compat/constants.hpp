// constants.hpp — drop-in for biolib's include/constants.hpp (reference lines 12-23, 85-109).
// Same names and meaning: constants::seq_nt4_table (A/a 0, C/c 1, G/g 2, T/t/U/u 3, else 4),
// constants::bases, and the char_iterator adaptor.  Written from the reference's documented
// behaviour; the table is generated, not copied.
#ifndef BIOLIB_AMD_COMPAT_CONSTANTS_HPP
#define BIOLIB_AMD_COMPAT_CONSTANTS_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <iterator>

namespace constants {

namespace detail {
constexpr std::array<uint8_t, 256> make_nt4()
{
    std::array<uint8_t, 256> t{};
    for (auto& x : t) x = 4;
    t['A'] = t['a'] = 0;
    t['C'] = t['c'] = 1;
    t['G'] = t['g'] = 2;
    t['T'] = t['t'] = t['U'] = t['u'] = 3;
    return t;
}
}  // namespace detail

inline constexpr std::array<uint8_t, 256> seq_nt4_table = detail::make_nt4();
inline constexpr std::array<char, 4> bases = {'A', 'C', 'G', 'T'};

}  // namespace constants

class char_iterator
{
    public:
        using iterator_category = std::random_access_iterator_tag;
        using difference_type   = std::ptrdiff_t;
        using value_type        = char;
        using pointer           = value_type*;
        using reference         = value_type&;

        char_iterator(char const* ptr) : internal(ptr) {}
        value_type operator*() const noexcept {return *internal;}
        char_iterator const& operator++() {++internal; return *this;}
        char_iterator operator++(int) {auto res = *this; ++internal; return res;}
        char const* base() const noexcept {return internal;}
    private:
        char const* internal;
        friend bool operator==(char_iterator const& a, char_iterator const& b) {return a.internal == b.internal;}
        friend bool operator!=(char_iterator const& a, char_iterator const& b) {return a.internal != b.internal;}
};

#endif
