// Host-only check of hash::minimizer_position_extractor from the drop-in header (no GPU: the class is scalar host code over
// bl_hash64_u64).  stdin: "k m n" then n lines, each a packed k-mer as a decimal u64 or "-" for a null item; stdout: one
// offset per line.  tests/test_compat_extractor_host.py compares the output with the offsets the reference's own extractor
// produced (tests/golden/arrays.npz: minpos_*).
#include <cstdio>
#include <cstring>

#include "kmer_view.hpp"

int main()
{
    unsigned k, m;
    unsigned long long n;
    if (std::scanf("%u %u %llu", &k, &m, &n) != 3) return 2;
    hash::minimizer_position_extractor ex((uint8_t)k, (uint8_t)m);
    if (ex.get_k() != k || ex.get_m() != m) return 3;
    char word[64];
    for (unsigned long long i = 0; i < n; ++i) {
        if (std::scanf("%63s", word) != 1) return 4;
        wrapper::kmer_context_t<uint64_t> item{std::nullopt, (std::size_t)i, (std::size_t)i};
        if (std::strcmp(word, "-") != 0) item.value = std::strtoull(word, nullptr, 10);
        std::printf("%zu\n", ex(item));
    }
    return 0;
}
