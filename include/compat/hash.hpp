// hash.hpp — drop-in for biolib's include/hash.hpp (reference lines 11-85) and bundled/MurmurHash3.hpp:33.
// hash::double_hash64 / hash::hash64 keep the reference's static hash(key,len,seed) / hash<T>(val,seed)
// and operator() overloads; hash64 of an 8-byte value is bit-identical to the device hash used by the
// scans (bl_hash64_u64).  MurmurHash3_x64_128 here is a host-side restatement of Appleby's public-domain
// algorithm for arbitrary key lengths (the GPU path only ever hashes 8-byte keys).
#ifndef BIOLIB_AMD_COMPAT_HASH_HPP
#define BIOLIB_AMD_COMPAT_HASH_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <optional>

inline void MurmurHash3_x64_128(const void* key, const int len, const uint32_t seed, void* out)
{
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto fmix = [](uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
        return k;
    };
    const uint8_t* data = static_cast<const uint8_t*>(key);
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int i = 0; i < nblocks; ++i) {
        uint64_t k1, k2;
        std::memcpy(&k1, data + 16 * i, 8);
        std::memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t* tail = data + 16 * nblocks;
    const int rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for (int i = rem - 1; i >= 8; --i) k2 ^= static_cast<uint64_t>(tail[i]) << (8 * (i - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 ^= static_cast<uint64_t>(tail[i]) << (8 * i);
    if (rem > 0) { k1 *= c1; k1 = rotl(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= static_cast<uint64_t>(len); h2 ^= static_cast<uint64_t>(len);
    h1 += h2; h2 += h1;
    h1 = fmix(h1); h2 = fmix(h2);
    h1 += h2; h2 += h1;
    static_cast<uint64_t*>(out)[0] = h1;
    static_cast<uint64_t*>(out)[1] = h2;
}

namespace hash {

class double_hash64
{
    public:
        typedef uint64_t hash_type;

        static std::array<uint64_t, 2> hash(uint8_t const* key, uint32_t keylen, uint32_t seed) noexcept
        {
            std::array<uint64_t, 2> hval;
            MurmurHash3_x64_128(reinterpret_cast<const void*>(key), static_cast<int>(keylen), seed, hval.data());
            return hval;
        }

        template <typename T>
        static std::array<uint64_t, 2> hash(T val, uint64_t seed) noexcept
        {   // raw bytes of val, seed truncated to 32 bits (reference hash.hpp:23-27 -> :16)
            return hash(reinterpret_cast<uint8_t*>(&val), sizeof(T), static_cast<uint32_t>(seed));
        }

        std::array<uint64_t, 2> operator()(uint8_t const* key, uint32_t keylen, uint32_t seed) const noexcept {return hash(key, keylen, seed);}
        template <typename T> std::array<uint64_t, 2> operator()(T val, uint64_t seed) const noexcept {return hash(val, seed);}
};

class hash64
{
    public:
        typedef uint64_t hash_type;

        static uint64_t hash(uint8_t const* key, uint32_t keylen, uint32_t seed) noexcept {return double_hash64::hash(key, keylen, seed)[0];}

        template <typename T>
        static uint64_t hash(T val, uint64_t seed) noexcept
        {
            return hash(reinterpret_cast<uint8_t*>(&val), sizeof(T), static_cast<uint32_t>(seed));
        }

        uint64_t operator()(uint8_t const* key, uint32_t keylen, uint32_t seed) const noexcept {return hash(key, keylen, seed);}
        template <typename T> uint64_t operator()(T val, uint64_t seed) const noexcept {return hash(val, seed);}
};

// Stafford mix13 (reference hash.hpp:81-85)
inline uint64_t remix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

}  // namespace hash

#endif
