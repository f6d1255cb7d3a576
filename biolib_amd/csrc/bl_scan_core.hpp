// bl_scan_core.hpp — the per-thread building blocks of the fused k-mer / minimizer scan.
//
// Everything here is written once and compiled two ways:
//   * by hipcc for gfx950 (bl_kernels.hip): BL_DEV = __device__ __forceinline__
//   * by a host compiler for the CPU emulation harness under tests/emu/ (BL_CPU_EMU), where
//     the same phase functions are run thread-by-thread under AddressSanitizer.  The harness
//     is test infrastructure; the product path is the HIP build only.
//
// Reference semantics reproduced (file:line in /root/reference):
//   constants.hpp:12-21            ASCII -> 2-bit code table (encode16)
//   kmer_view.hpp:162-170,190-199  mask/shift, rolling forward / reverse-complement registers
//   hash.hpp:50-59 + bundled/MurmurHash3.cpp:263-341  hash64 of an 8-byte key, 32-bit seed
//   minimizer_view.hpp:283,374     strict '<'  => leftmost minimum wins ties
//   kmer_view.hpp:266-283          syncmer predicate (rightmost tie when the reverse strand is canonical)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(BL_CPU_EMU)
#define BL_DEV __host__ __device__ __forceinline__
#define BL_UNROLL _Pragma("unroll")
#else
#define BL_DEV static inline
#define BL_UNROLL
#endif

// Branches that synthetic / clean data never take (hash-prefix ties, breaks, ragged batch ends).  tools/valu_model.py builds
// the kernels with BL_CENSUS_HOT, which compiles them out, to count the instructions of the hot path alone; no product build
// defines it.
#ifdef BL_CENSUS_HOT
#define BL_COLD(c) (false)
#else
#define BL_COLD(c) (c)
#endif

// The instruction scheduler may not move anything across this point.  Used between groups of independent hashes: left alone, the
// scheduler interleaves all of a lane's hashes for instruction-level parallelism and keeps every one's temporaries alive at once,
// which is what decides whether a kernel fits the registers of four waves per SIMD.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU) && !defined(BL_NO_SCHED_FENCE)
#define BL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BL_SCHED_FENCE() do {} while (0)
#endif

namespace bl {

constexpr int TPB = 256;            // threads per workgroup (4 wave64)
constexpr int S = 16;               // window/unit start positions owned by one thread (= one 16-byte load)
constexpr int H = TPB * S;          // positions hashed per tile
constexpr int NCHUNK = 512;         // 16-base chunks a tile may stage (position-tiled scans use H / 16 + 8 of them, read-tiled ones more)
constexpr int NCHUNK_POS = H / 16 + 8;  // position-tiled scans: one chunk per thread plus a halo of up to 128 bases
constexpr int MAX_UNIT = 32;        // KmerType = uint64_t only
constexpr int MAX_W = 64;
constexpr int NSHARD = 256;         // digest accumulator shards (one 64-byte line each)
constexpr int SCAN_BLK = 2048;      // tiles per block of the tile-count prefix scan
constexpr int NWAVE_CORE = TPB / 64;

enum ScanMode { MODE_MINIMIZER = 0, MODE_SUPERKMER = 1, MODE_SYNCMER = 2 };

struct ScanParams {
    const uint8_t* bases;        // ASCII, 1 byte per base, 16-byte aligned
    int64_t n_bases;
    const uint32_t* start_bits;  // bit p set <=> p is the first base of a sequence (nullptr: one sequence)
    int64_t win_first, win_end;  // only windows whose first base lies in [win_first, win_end) are reported
    int64_t origin;              // first hashed position of wave 0 of tile 0 (multiple of 16, may be negative)
    int64_t pos_base;            // added to every reported position: where base 0 of the batch lies in the caller's whole (bl_batch_set_origin)
    int32_t n_tiles;
    int32_t stride;              // owned positions per workgroup tile = NWAVE * (1024 - 16*ceil(w/16))
    int32_t unit, w;
    uint32_t seed;
    int32_t canonical;
    int32_t soff, eoff;          // syncmer offsets
    int32_t drop_last;           // syncmer / k-mer scans (and w = 1 unit lists): skip the k-mer that ends a sequence (quirk Q1)
    int32_t use_threshold;       // w = 1 only: keep units with hash < hash_below (hash_sampler.hpp:136-141)
    uint64_t hash_below;
    // outputs (device pointers, nullable)
    uint64_t* out_value;
    uint64_t* out_pos;
    uint64_t* out_hash;
    uint64_t* out_first;         // super-k-mer: position of the first k-mer of the group
    uint8_t* out_mmpos;          // super-k-mer: minimizer offset inside the first k-mer
    uint8_t* out_size;           // super-k-mer: number of k-mers of the group (last k-mer - first k-mer + 1, super_kmer_view.hpp:133)
    uint64_t* out_records;       // super-k-mer: the group's bases packed into 16 bytes (bl_superkmer.hip's record), two words per group
    uint64_t capacity;           // records the output arrays can hold
    // two-pass ordered compaction (no inter-workgroup communication inside a kernel):
    //   pass 1 (scan_count_kernel) writes per tile its record counts and its compacted u16 lists,
    //   a prefix scan over the tile counts gives every tile its global record offset,
    //   pass 2 (scan_emit_kernel) rebuilds the records of its tile and stores them at that offset.
    unsigned long long* tile_counts;  // [n_tiles] starts | ends << 32
    unsigned long long* tile_base;    // [n_tiles] exclusive prefix inside its scan block
    unsigned long long* block_base;   // [ceil(n_tiles / SCAN_BLK)] exclusive prefix of the scan blocks
    uint16_t* slots_a;                // [n_tiles][stride] list_a of every tile (first `starts` entries valid)
    uint16_t* slots_j;                // super-k-mer: list_j
    uint16_t* slots_e;                // super-k-mer: list_e (first `ends` entries valid)
    uint32_t* slots_c;                // [n_tiles][slot_chunks] the tile's 2-bit codes (pass 2 rebuilds unit values from them)
    int32_t slot_chunks;              // chunks a tile stages and hands to pass 2 (<= NCHUNK)
    unsigned long long* shards;       // [NSHARD][8] digest accumulators
    // closed-syncmer scans: tiles in which a comparison met equal high dwords are listed here by pass 1 and counted again, in the
    // exact argmin form, by scan_redo_kernel before the prefix scan (nullptr: never happens for other scans)
    int32_t exact_windows;            // bl_ctx_set_exact_windows: no pass 1 on murmur64_top, no closed-syncmer form (checks and A/B runs)
    uint32_t* redo_list;              // [n_tiles]
    unsigned long long* redo_count;
    // Read-tiled layout (bl_scan_frl.hpp): batches of FIXED-LENGTH short reads, range aligned to reads.  A wave takes
    // rpw whole reads, lpr lanes per read, ns consecutive unit starts per lane: the unit-1 tail positions of a read,
    // which cannot start a unit, are never rolled or hashed, sequence starts are arithmetic (no start_bits), and there
    // is no halo between wave tiles.  origin = first base of the range, stride = NWAVE * rpw * read_len bases per tile.
    int32_t frl;                      // 0: position-tiled layout
    int32_t read_len;
    int32_t lpr, rpw, ns;
    int32_t nwin;                     // windows per read = read_len - unit - w + 2
    uint32_t lpr_inv;                 // ceil(65536 / lpr): lane / lpr == (lane * lpr_inv) >> 16 for lane < 64
    int64_t n_reads;                  // reads in the range
};

// widths the read-tiled kernels are built for (bl_kernels.hip: launch_count_frl; the emulation harness instantiates the same set)
BL_DEV bool frl_width_built(int mode, int w)
{
    return (mode == MODE_MINIMIZER && (w == 11 || w == 5 || w == 10 || w == 19)) || (mode == MODE_SUPERKMER && w == 17);
}

// Read-tiled plan for a range of fixed-length reads.  ns_fixed: units per lane the kernel was compiled for (0: choose).
// Returns false when the layout does not apply (long reads, windows that do not fit, LDS bound) or would hash MORE
// positions per useful window than the position-tiled layout.
BL_DEV bool plan_scan_frl(int64_t first, int64_t end, int64_t n_bases, int64_t read_len, int unit, int w, int ns_fixed, ScanParams& p)
{
    if (read_len <= 0 || read_len > 4096 || first % read_len != 0) return false;
    if (end % read_len != 0 || end > n_bases || end <= first) return false;
    const int L = (int)read_len;
    const int nu = L - unit + 1, nwin = nu - w + 1;
    if (nwin < 1 || w < 2) return false;
    const int lpr = (nu + S - 1) / S;
    if (lpr > 64) return false;
    const int ns = ns_fixed ? ns_fixed : (nu + lpr - 1) / lpr;
    if (ns * lpr < nu || ns > S || w - 1 > 3 * ns) return false;     // halo of at most three lanes
    int rpw = 64 / lpr;
    const int cap = (NCHUNK * 16 - 64 - 32) / NWAVE_CORE / L;        // every staged chunk of a tile must fit NCHUNK
    if (rpw > cap) rpw = cap;
    if (rpw < 1) return false;
    // useful windows per hashed lane-slot, both layouts (64 lanes x 16 slots per wave tile)
    const double frl_eff = (double)(rpw * nwin) / (64.0 * ns);
    const double pos_eff = (double)nwin / L * (double)(64 * S - 16 * ((w + 15) / 16)) / (64.0 * S);
    if (frl_eff <= pos_eff) return false;
    p.frl = 1;
    p.read_len = L;
    p.lpr = lpr;
    p.rpw = rpw;
    p.ns = ns;
    p.nwin = nwin;
    p.lpr_inv = (65536u + (uint32_t)lpr - 1) / (uint32_t)lpr;
    p.n_reads = (end - first) / L;
    p.win_first = first;
    p.win_end = end;
    p.origin = first;
    p.stride = NWAVE_CORE * rpw * L;
    p.n_tiles = (int32_t)((p.n_reads + NWAVE_CORE * rpw - 1) / (NWAVE_CORE * rpw));
    p.slot_chunks = (15 + p.stride + 15) / 16 + 3;
    return true;
}

// The read-tiled plan the C ABI and the emulation harness use: the BASELINE shape (canonical 31-mers, window 11) takes the units per lane
// the read length asks for when kernels are built for it (14, 15, 16: launch_count_frl), everything else S per lane.
BL_DEV bool plan_scan_frl_for(int mode, int64_t first, int64_t end, int64_t n_bases, int64_t read_len, int unit, int w, bool canonical, ScanParams& p)
{
    if (mode == MODE_MINIMIZER && w == 11 && unit == 31 && canonical) {
        ScanParams q = p;
        if (plan_scan_frl(first, end, n_bases, read_len, unit, w, 0, q) && q.ns >= 14 && q.ns <= 16) {
            p = q;
            return true;
        }
    }
    return plan_scan_frl(first, end, n_bases, read_len, unit, w, S, p);
}

// a contiguous group of tiles handled by one stage of the software pipeline
struct GroupRange {
    uint32_t first, count;
};

// Tile plan shared by the C ABI (bl_capi.hip) and the emulation harness: which positions tile 0
// starts at, how many positions a tile owns and how many tiles cover the range [first, end).
BL_DEV int64_t align_down16(int64_t x) { return x >= 0 ? (x & ~15LL) : -(((-x) + 15) & ~15LL); }

BL_DEV void plan_scan(int mode, int64_t first, int64_t end, int w, ScanParams& p)
{
    p.win_first = first;
    p.win_end = end;
    p.stride = (TPB / 64) * (64 * S - 16 * ((w + 15) / 16));  // NWAVE wave tiles, each owning 1024 - halo positions
    // minimizer modes: the owner of position i decides window i+1, so tile 0 starts one position early
    p.origin = align_down16(mode == MODE_SYNCMER ? first : first - 1);
    p.n_tiles = end > first ? (int32_t)((end - 1 - p.origin) / p.stride + 1) : 0;
    p.frl = 0;
    p.slot_chunks = NCHUNK_POS;
}

// ------------------------------------------------------------------------------------------------
// MurmurHash3_x64_128 of one 8-byte little-endian key, low 64 bits (h1).  Closed form of
// bundled/MurmurHash3.cpp:263-341 for len == 8: nblocks = 0, tail case 8 builds k1 = key.
BL_DEV uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// rotl64(x, 31) as two v_alignbit_b32 (the generic form costs two 64-bit shifts and two ORs)
BL_DEV uint64_t rotl64_31(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 1), nl = __builtin_amdgcn_alignbit(lo, hi, 1);
    return ((uint64_t)nh << 32) | nl;
#else
    return rotl64(x, 31);
#endif
}

// a * c mod 2^64 for a constant c.  Written so that gfx950 gets v_mul_lo_u32 + 2 x v_mad_u64_u32 + v_mov (the cross terms
// ride in as the 64-bit addend of the multiply-adds): 15.0 issue cycles per wave against 16.5 for the compiler's own
// lowering of `a * c` (v_mad_u64_u32 + 2 x v_mul_lo_u32 + v_add3_u32; tools/ubench_valu.hip prices both).
template <bool SPELLED = false>
BL_DEV uint64_t mul64c_as(uint64_t a, uint64_t c)
{
    const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
    const uint32_t clo = (uint32_t)c, chi = (uint32_t)(c >> 32);
    uint32_t cross = alo * chi + ahi * clo;  // v_mul_lo_u32, then v_mad_u64_u32 with the first product as its addend
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    asm("" : "+v"(cross));  // opaque: left visible, the optimizer folds the three products back into `a * c`
    if (SPELLED) {
        // the last multiply-add spelled out, for murmur64_top: written in C++, a caller that goes on with the two dwords separately
        // gets a v_mul_lo_u32 of its own for the low one, beside the v_mad_u64_u32 that already holds it.  (Everywhere else the
        // spelled form only adds register moves.)
        uint64_t d, carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(alo), "s"(clo), "v"((uint64_t)cross << 32));
        return d;
    }
#endif
    return (uint64_t)alo * clo + ((uint64_t)cross << 32);
}
BL_DEV uint64_t mul64c(uint64_t a, uint64_t c) { return mul64c_as<false>(a, c); }

BL_DEV uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k = mul64c(k, 0xff51afd7ed558ccdULL);
    k ^= k >> 33;
    k = mul64c(k, 0xc4ceb9fe1a85ec53ULL);
    k ^= k >> 33;
    return k;
}

BL_DEV uint64_t murmur64(uint64_t key, uint32_t seed)
{
    uint64_t k1 = mul64c(key, 0x87c37b91114253d5ULL);
    k1 = rotl64_31(k1);
    k1 = mul64c(k1, 0x4cf5ad432745937fULL);
    uint64_t h1 = (uint64_t)seed ^ k1;
    uint64_t h2 = (uint64_t)seed;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// NEARLY the high dword of murmur64(key, seed), for the window comparisons of pass 1: T = murmur64(...) >> 32 is S or S + 1
// (mod 2^32), S the value returned here.  The hash is fmix64(h1) + fmix64(h2); its high dword is the sum of the two high
// dwords plus the carry of the low ones, and the carry is what S leaves out — with it go the low halves of both final
// multiplies, the two `k ^= k >> 33` that only touch those halves, and half of the 64-bit add: 8 instructions of 45.
// What the callers make of it (window_argmin_packed<..., APPROX>): two keys whose 26-bit prefixes differ by 2 or more are
// ordered like their hashes; anything closer counts as a tie and the tile is decided again with murmur64 itself, and so
// is a tile that holds an S with all prefix bits set, the one place where S + 1 could wrap to the SMALLEST prefix.
// a * b + c as v_mad_u64_u32 says it, for chains of which only the low dword is wanted in the end (the compiler, seeing that,
// writes v_mul_lo_u32 + v_add per link): the high dword is carried along, never looked at.  mad_lo0: the first link, c = 0 (or 1).
BL_DEV uint64_t mad_lo(uint32_t a, uint32_t b, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    return d;
#else
    return (uint64_t)(uint32_t)(a * b + (uint32_t)c);
#endif
}
template <bool PLUS_ONE = false>
BL_DEV uint64_t mad_lo0(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    uint64_t d, carry;
    if (PLUS_ONE) asm("v_mad_u64_u32 %0, %1, %2, %3, 1" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    return d;
#else
    return (uint64_t)(uint32_t)(a * b + (PLUS_ONE ? 1u : 0u));
#endif
}
// PLUS_ONE: S + 1 instead (the hash's high dword is then the value or one BELOW it, and the value that could wrap is 0): the 1 rides in
// as the addend of the chain's first link.
template <bool PLUS_ONE = false>
BL_DEV uint32_t murmur64_top(uint64_t key, uint32_t seed)
{
    uint64_t k1 = mul64c(key, 0x87c37b91114253d5ULL);
    k1 = rotl64_31(k1);
    k1 = mul64c(k1, 0x4cf5ad432745937fULL);
    uint64_t h1 = (uint64_t)seed ^ k1;
    uint64_t h2 = (uint64_t)seed;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 ^= h1 >> 33;
    h1 = mul64c_as<true>(h1, 0xff51afd7ed558ccdULL);
    h1 ^= h1 >> 33;
    h2 ^= h2 >> 33;
    h2 = mul64c_as<true>(h2, 0xff51afd7ed558ccdULL);
    h2 ^= h2 >> 33;
    // high dwords of h1 * C and h2 * C, summed: the four cross products in one chain of multiply-adds, the two v_mul_hi on top
    const uint32_t clo = 0x1a85ec53u, chi = 0xc4ceb9feu;
    const uint32_t a1 = (uint32_t)h1, b1 = (uint32_t)(h1 >> 32), a2 = (uint32_t)h2, b2 = (uint32_t)(h2 >> 32);
    const uint32_t cross = (uint32_t)mad_lo(b2, clo, mad_lo(a2, chi, mad_lo(b1, clo, mad_lo0<PLUS_ONE>(a1, chi))));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    return __umulhi(a1, clo) + __umulhi(a2, clo) + cross;
#else
    return (uint32_t)(((uint64_t)a1 * clo) >> 32) + (uint32_t)(((uint64_t)a2 * clo) >> 32) + cross;
#endif
}

// The same hash with the compiler's own lowering of the six multiplies (v_mad_u64_u32 + 2 x v_mul_lo_u32 + v_add3_u32).  For the
// argmin form of the syncmer scan, which its 245 registers hold to two waves per SIMD: with so few waves to switch between, the
// dependent v_mul_lo -> v_mad -> v_mov -> v_mad chain of mul64c is exposed, and that kernel ran 10 % SLOWER with it (252 -> 228).
BL_DEV uint64_t fmix64_plain(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
BL_DEV uint64_t murmur64_plain(uint64_t key, uint32_t seed)
{
    uint64_t k1 = key * 0x87c37b91114253d5ULL;
    k1 = rotl64_31(k1);
    k1 *= 0x4cf5ad432745937fULL;
    uint64_t h1 = (uint64_t)seed ^ k1;
    uint64_t h2 = (uint64_t)seed;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix64_plain(h1); h2 = fmix64_plain(h2);
    return h1 + h2;
}

// ------------------------------------------------------------------------------------------------
// 16 ASCII bytes (4 little-endian dwords, first base in byte 0 of d[0]) ->
//   code : 2 bits per base, FIRST base in the most significant pair (kmer_view.hpp:194 order)
//   bad  : bit b set <=> base b is not one of ACGTUacgtu (constants.hpp:12-21 maps those to 4)
BL_DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t both = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t byte = s < 8 ? (uint32_t)((both >> (8 * s)) & 0xff) : 0;  // selectors >= 8 never used here
        r |= byte << (8 * i);
    }
    return r;
#endif
}

// true in every lane of the wave if the predicate holds in any of them (the CPU emulation decides per thread: every
// use below picks between a fast form and an exact form that agree wherever the fast form is valid)
BL_DEV bool wave_any(bool pred)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    return __builtin_amdgcn_ballot_w64(pred) != 0;
#else
    return pred;
#endif
}

// four bases: code8 = their 2-bit codes (first base most significant), diff = a word whose byte b is non-zero
// iff base b is not one of ACGTUacgtu
BL_DEV void encode4(uint32_t d, uint32_t& code8, uint32_t& diff)
{
    // per byte: x = (c >> 1) & 3; x ^= x >> 1   -> A/a 0, C/c 1, G/g 2, T/t/U/u 3
    uint32_t x = (d >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    // validity: rebuild the lowercase letter the code stands for and compare with (c | 0x20);
    // 'u' (0x75) differs from 't' (0x74) in bit 0 only, which is forgiven where code == 3
    const uint32_t lut = 0x74676361u;  // bytes 0..3 = 'a','c','g','t'
    const uint32_t expect = byte_perm(lut, lut, x);
    diff = ((d | 0x20202020u) ^ expect) & ~((x & (x >> 1)) & 0x01010101u);
    // gather the four 2-bit codes, byte 0 (first base) most significant
    code8 = (x * 0x40100401u) >> 24;
}

// bit b set iff byte b of diff is non-zero
BL_DEV uint32_t bad_bits4(uint32_t diff)
{
    const uint32_t nz = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;  // bit 7 of each non-zero byte
    return ((nz >> 7) * 0x00204081u >> 21) & 0xfu;  // gather bits 0,8,16,24 -> bits 0..3 (byte 0 -> bit 0)
}

BL_DEV void encode16(const uint32_t d[4], uint32_t& code, uint32_t& bad)
{
    uint32_t c0, c1, c2, c3, f0, f1, f2, f3;
    encode4(d[0], c0, f0);
    encode4(d[1], c1, f1);
    encode4(d[2], c2, f2);
    encode4(d[3], c3, f3);
    code = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    bad = 0;
    // breaks are rare in sequencing data: the per-base bit gather runs only for waves that hold one
    if (BL_COLD(wave_any((f0 | f1 | f2 | f3) != 0))) bad = bad_bits4(f0) | (bad_bits4(f1) << 4) | (bad_bits4(f2) << 8) | (bad_bits4(f3) << 12);
}

// reverse the order of the 32 two-bit pairs of x
BL_DEV uint64_t pairrev64(uint64_t x)
{
    uint64_t r = __builtin_bitreverse64(x);
    return ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
}

// ------------------------------------------------------------------------------------------------
// The rolling state of kmer_view.hpp:190-199 for one thread, started from packed codes instead of
// a warm-up loop.  c0:c1:c2 are the codes of the thread's 48 bases (big-endian pairs).
// Kept as explicit 32-bit halves: the 64-bit C++ forms of the two updates compile to 11 VALU instructions per
// base on gfx950 (64-bit shifts, a redundant low mask), the halves below to 8 (v_bfe, v_lshl_or, 2 x v_alignbit,
// v_and, v_lshrrev, v_xor, v_lshl_or) — 3 % of the whole scan.
struct Roller {
    uint32_t flo, fhi;   // forward unit, first base most significant (kmer_view.hpp:194)
    uint32_t rlo, rhi;   // reverse complement (kmer_view.hpp:195)
    uint32_t next16;     // the 16 bases following the first unit-1, big-endian pairs
    int unit;
};
BL_DEV uint64_t roller_fwd(const Roller& r) { return ((uint64_t)r.fhi << 32) | r.flo; }
BL_DEV uint64_t roller_rc(const Roller& r) { return ((uint64_t)r.rhi << 32) | r.rlo; }

// low 32 bits of (hi:lo) >> n, 0 < n < 32: one v_alignbit_b32
BL_DEV uint32_t funnel_shr(uint32_t hi, uint32_t lo, int n)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    return __builtin_amdgcn_alignbit(hi, lo, n);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> n);
#endif
}

BL_DEV void roller_start(Roller& r, uint32_t c0, uint32_t c1, uint32_t c2, int unit)
{
    const uint64_t A = ((uint64_t)c0 << 32) | c1;  // bases 0..31
    const uint64_t B = ((uint64_t)c1 << 32) | c2;  // bases 16..47
    const int pre = unit - 1;                      // bases already inside the registers
    uint64_t fwd = 0, rc = 0;
    if (pre != 0) {
        fwd = A >> (64 - 2 * pre);
        // reference state after `pre` bases: latest base at 2*(unit-1), older ones 2 bits lower each
        uint64_t little = pairrev64(fwd) >> (64 - 2 * pre);          // base i at bit 2i
        little ^= (pre == 32) ? ~0ULL : ((1ULL << (2 * pre)) - 1);   // complement (3 ^ c)
        rc = little << 2;
    }
    r.flo = (uint32_t)fwd;
    r.fhi = (uint32_t)(fwd >> 32);
    r.rlo = (uint32_t)rc;
    r.rhi = (uint32_t)(rc >> 32);
    r.next16 = pre < 16 ? (uint32_t)((A << (2 * pre)) >> 32) : (uint32_t)((B << (2 * (pre - 16))) >> 32);
    r.unit = unit;
}

BL_DEV void roller_step(Roller& r, int s)  // s = 0..15, compile-time after unrolling
{
    const int at = 30 - 2 * s;                       // bit position of base s inside next16
    const uint32_t c = (r.next16 >> at) & 3u;
    const int top = 2 * (r.unit - 1);                // where the complement of the new base enters rc
    if (r.unit > 16) {
        // fwd = ((fwd << 2) | c) & mask
        const uint32_t nhi = funnel_shr(r.fhi, r.flo, 30);
        r.flo = (r.flo << 2) | c;
        r.fhi = r.unit == 32 ? nhi : (nhi & ((1u << (2 * r.unit - 32)) - 1u));
        // rc = (rc >> 2) | ((3 ^ c) << top)
        r.rlo = funnel_shr(r.rhi, r.rlo, 2);
        r.rhi = ((c ^ 3u) << (top - 32)) | (r.rhi >> 2);
    } else {
        const uint32_t m = r.unit == 16 ? ~0u : ((1u << (2 * r.unit)) - 1u);
        r.flo = ((r.flo << 2) | c) & m;
        r.rlo = ((c ^ 3u) << top) | (r.rlo >> 2);
    }
}

// unit starting at tile-relative base `pos`, straight from the staged codes (used when a record is
// materialised; not on the per-base path)
BL_DEV uint64_t extract_unit(const uint32_t* codes, int pos, int unit, int canonical)
{
    const int ch = pos >> 4, off = pos & 15;
    const uint64_t A = ((uint64_t)codes[ch] << 32) | codes[ch + 1];
    // 96 bits of pairs starting at chunk ch; take `unit` pairs from pair offset `off`
    // 32 bases from `off`; the third chunk contributes its top 2*off bits (none for off = 0: shifted out in two steps,
    // which keeps the expression branch-free for the per-record loop of pass 2)
    uint64_t top = (A << (2 * off)) | (uint64_t)((codes[ch + 2] >> 1) >> (31 - 2 * off));
    uint64_t fwd = unit == 32 ? top : (top >> (64 - 2 * unit));
    if (!canonical) return fwd;
    uint64_t rc = pairrev64(fwd) >> (64 - 2 * unit);
    rc ^= unit == 32 ? ~0ULL : ((1ULL << (2 * unit)) - 1);
    return rc < fwd ? rc : fwd;  // numeric min, kmer_view.hpp:196
}

// The `nb` <= 59 bases from tile-relative base `pos` on, as the 16-byte super-k-mer record of bl_superkmer.hip: x = bases 0..31
// (first base in the two top bits), y = bases 32..58 in its 54 top bits | mm_pos << 5 | size - 1.  Straight from the staged
// codes; chunks beyond `last_chunk` are not looked at (their bases would be masked anyway).
BL_DEV void pack_group(const uint32_t* codes, int last_chunk, int pos, int nb, uint32_t mm_pos, int size, uint64_t& x, uint64_t& y)
{
    const int ch = pos >> 4, off = pos & 15;
    uint64_t c[5];
    BL_UNROLL
    for (int i = 0; i < 5; ++i) c[i] = codes[ch + i <= last_chunk ? ch + i : last_chunk];
    const uint64_t A = (c[0] << 32) | c[1], B = (c[2] << 32) | c[3], D = c[4] << 32;
    uint64_t hi = A, lo = B;
    if (off) {
        hi = (A << (2 * off)) | (B >> (64 - 2 * off));
        lo = (B << (2 * off)) | (D >> (64 - 2 * off));
    }
    if (nb < 32) hi &= nb > 0 ? ~0ULL << (64 - 2 * nb) : 0ULL;
    if (nb <= 32) lo = 0;
    else if (nb < 64) lo &= ~0ULL << (64 - 2 * (nb - 32));
    x = hi;
    y = (lo & ~0x3ffULL) | ((uint64_t)(mm_pos & 31u) << 5) | (uint64_t)((size - 1) & 31);
}

// ------------------------------------------------------------------------------------------------
// 128-bit bit-vectors for the validity masks (bit i = tile-relative position i0 + i)
struct Bits128 {
    uint64_t lo, hi;
};
BL_DEV Bits128 b128_shr(Bits128 a, int n)  // 0 <= n < 128
{
    Bits128 r;
    if (n == 0) return a;
    if (n >= 64) {
        r.lo = a.hi >> (n - 64);
        r.hi = 0;
    } else {
        r.lo = (a.lo >> n) | (a.hi << (64 - n));
        r.hi = a.hi >> n;
    }
    return r;
}
BL_DEV Bits128 b128_and(Bits128 a, Bits128 b) { return Bits128{a.lo & b.lo, a.hi & b.hi}; }

// bit i of result = AND of bits i .. i+len-1 of m   (len >= 1)
BL_DEV Bits128 and_run(Bits128 m, int len)
{
    Bits128 r = m;
    int have = 1;
    while (have < len) {
        int step = have < len - have ? have : len - have;
        r = b128_and(r, b128_shr(r, step));
        have += step;
    }
    return r;
}

// Window validity for the S+1 windows starting at the thread's positions 0..S:
// a window of `span` bases starting at i is valid iff all its bases are good and no base except
// the first is a sequence start.  good/start: bit i = base i0 + i.
BL_DEV uint32_t window_valid_mask(Bits128 good, Bits128 start, int span)
{
    if (span + S <= 63) {  // every bit we look at lives in the low words (k=31,w=11: 57 bits)
        const uint64_t link = good.lo & ~(start.lo >> 1);  // link[i] = good[i] && !start[i+1]
        uint64_t v = good.lo >> (span - 1);
        if (span > 1) {
            uint64_t r = link;  // r: AND of `have` consecutive link bits
            int have = 1;
            const int len = span - 1;
            while (have < len) {
                const int step = have < len - have ? have : len - have;
                r &= r >> step;
                have += step;
            }
            v &= r;
        }
        return (uint32_t)v & ((1u << (S + 1)) - 1);
    }
    Bits128 ns = b128_shr(start, 1);
    Bits128 link{good.lo & ~ns.lo, good.hi & ~ns.hi};
    Bits128 v = b128_shr(good, span - 1);
    if (span > 1) v = b128_and(v, and_run(link, span - 1));
    return (uint32_t)(v.lo & ((1u << (S + 1)) - 1));
}

// ------------------------------------------------------------------------------------------------
// Sliding-window argmin over registers (van Herk / Gil-Werman inside one thread): NW windows of W
// elements over e[0 .. NW+W-2]; a[i] = index of the minimum of e[i .. i+W-1].
// LEFT = true : leftmost minimum wins ties (minimizer_view.hpp:283,374)
// LEFT = false: rightmost minimum wins ties (reverse-strand syncmers)
template <int NW, int W, bool LEFT>
BL_DEV void window_argmin(const uint64_t* e, uint32_t* a)
{
    BL_UNROLL
    for (int base = 0; base < NW; base += W) {
        // suffix minima of block e[base .. base+W-1], right to left
        uint64_t sv[W];
        uint32_t si[W];
        sv[W - 1] = e[base + W - 1];
        si[W - 1] = (uint32_t)(base + W - 1);
        BL_UNROLL
        for (int i = W - 2; i >= 0; --i) {
            const uint64_t x = e[base + i];
            const bool take = LEFT ? (x <= sv[i + 1]) : (x < sv[i + 1]);
            sv[i] = take ? x : sv[i + 1];
            si[i] = take ? (uint32_t)(base + i) : si[i + 1];
        }
        a[base] = si[0];
        // prefix minima of the following block, combined on the fly
        uint64_t pv = 0;
        uint32_t pi = 0;
        BL_UNROLL
        for (int i = 1; i < W; ++i) {
            if (base + i >= NW) break;
            const uint64_t x = e[base + W - 1 + i];
            if (i == 1) {
                pv = x;
                pi = (uint32_t)(base + W);
            } else {
                const bool take = LEFT ? (x < pv) : (x <= pv);
                pv = take ? x : pv;
                pi = take ? (uint32_t)(base + W - 1 + i) : pi;
            }
            const bool right = LEFT ? (pv < sv[i]) : (pv <= sv[i]);
            a[base + i] = right ? pi : si[i];
        }
    }
}


// Fast form of the same argmin on 32-bit PACKED keys: key[x] = (top 26 bits of the hash) << 6 | tag, tag = x for
// LEFT (smaller tag wins a tie of the prefix = leftmost) and 63 - x otherwise.  One v_min_u32 per step gives value
// and position at once (the 64-bit form costs a compare, two value selects, and a re-compare + select for the
// index).  It is exact unless two operands of some step agree in their 26-bit prefix; every step therefore also
// folds (a ^ b) into a running minimum, and the caller re-runs the 64-bit form when that minimum is below 64
// (probability ~1e-6 per window on random hashes; certain on repeats, which is why the exact form stays).
// Returns the minimum xor distance seen.  NW + W - 1 <= 64 elements.
// RAW: leave the whole minimum key in a[] (position in its low 6 bits, hash bits above) for callers that mask anyway.
// APPROX: the keys come from murmur64_top; the distance folded is |a - b| instead of a ^ b and the caller's bound is 128 (prefixes
// that differ by less than 2) instead of 64 (equal prefixes).
BL_DEV uint32_t abs_diff(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return a < b ? b - a : a - b;
#endif
}
BL_DEV uint32_t key_distance(uint32_t a, uint32_t b, bool approx)
{
    if (!approx) return a ^ b;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return a < b ? b - a : a - b;
#endif
}
template <int NW, int W, bool LEFT, bool RAW = false, bool APPROX = false>
BL_DEV uint32_t window_argmin_packed(const uint32_t* key, uint32_t* a)
{
    uint32_t dmin = ~0u;
    BL_UNROLL
    for (int base = 0; base < NW; base += W) {
        uint32_t sv[W];
        sv[W - 1] = key[base + W - 1];
        BL_UNROLL
        for (int i = W - 2; i >= 0; --i) {
            const uint32_t x = key[base + i], d = key_distance(x, sv[i + 1], APPROX);
            sv[i] = x < sv[i + 1] ? x : sv[i + 1];
            dmin = d < dmin ? d : dmin;
        }
        a[base] = RAW ? sv[0] : (LEFT ? (sv[0] & 63u) : 63u - (sv[0] & 63u));
        uint32_t pv = 0;
        BL_UNROLL
        for (int i = 1; i < W; ++i) {
            if (base + i >= NW) break;
            const uint32_t x = key[base + W - 1 + i];
            if (i == 1) {
                pv = x;
            } else {
                const uint32_t d = key_distance(x, pv, APPROX);
                pv = x < pv ? x : pv;
                dmin = d < dmin ? d : dmin;
            }
            const uint32_t d2 = key_distance(pv, sv[i], APPROX);
            const uint32_t r = pv < sv[i] ? pv : sv[i];
            dmin = d2 < dmin ? d2 : dmin;
            a[base + i] = RAW ? r : (LEFT ? (r & 63u) : 63u - (r & 63u));
        }
    }
    return dmin;
}

// Runtime window size w, P < w <= 2P, P a power of two (template): a sparse table inside the lane.  log2(P) doubling
// levels turn key[i] into the minimum over [i, i+P); window i is then min(key[i], key[i + w - P]).  Keys are packed
// (top 25 hash bits | 7-bit position tag, leftmost wins a prefix tie); the return value is the smallest xor distance
// between two DIFFERENT operands, as in window_argmin_packed (below 128 = a prefix tie = not exact).
// key: NW + 2P - 1 entries; entries from NW + w - 1 on are never part of a wanted window and must be pads with
// distinct prefixes (pad_key).  a[i] = the whole minimum key (position in the low 7 bits).
// acc = min(acc, d), opaque to the optimizer: left as plain C++ the long chain of minima is reassociated into a tree
// whose leaves (one xor per step of a level) are then all alive at once — 155-256 VGPRs instead of ~90
BL_DEV void fold_min(uint32_t& acc, uint32_t d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    asm("v_min_u32_e32 %0, %1, %0" : "+v"(acc) : "v"(d));
#else
    acc = d < acc ? d : acc;
#endif
}

// acc = min(acc, d1, d2), equally opaque: one v_min3_u32
BL_DEV void fold_min3(uint32_t& acc, uint32_t d1, uint32_t d2)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    asm("v_min3_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(d1), "v"(d2));
#else
    const uint32_t d = d1 < d2 ? d1 : d2;
    acc = d < acc ? d : acc;
#endif
}

// one doubling level with a compile-time stride (a loop over q = 1, 2, 4 .. is not unrolled by the compiler, which
// turns key[] into a dynamically indexed array)
template <int NE, int Q>
BL_DEV void doubling_level(uint32_t* key, uint32_t& dmin)
{
    BL_UNROLL
    for (int i = 0; i + 2 * Q <= NE; ++i) {
        const uint32_t x = key[i], y = key[i + Q];
        key[i] = x < y ? x : y;
        fold_min(dmin, x ^ y);
    }
}

template <int N, int B>
BL_DEV void shift_stage(uint32_t* t, int sh)  // t[i] = t[i + B] for i < N when bit B of sh is set
{
    if (sh & B) {
        BL_UNROLL
        for (int i = 0; i < N; ++i) t[i] = t[i + B];
    }
}

template <int NW, int P>
BL_DEV uint32_t window_argmin_doubling(uint32_t* key, int w, uint32_t* a)
{
    constexpr int NE = NW + 2 * P - 1;
    uint32_t dmin = ~0u;
    // after level q key[i] covers [i, i + 2q): disjoint operands, tags differ
    if (P > 1) doubling_level<NE, 1>(key, dmin);
    if (P > 2) doubling_level<NE, 2>(key, dmin);
    if (P > 4) doubling_level<NE, 4>(key, dmin);
    if (P > 8) doubling_level<NE, 8>(key, dmin);
    if (P > 16) doubling_level<NE, 16>(key, dmin);
    // t[i] = key[i + (w - P - 1)]: the shift amount, 0 .. P-1, is applied bit by bit (uniform branches over register
    // moves); the first stage reads key[] directly so that t[] needs NW + P/2 entries only
    constexpr int H = P > 1 ? P / 2 : 0;
    constexpr int NT = NW + (H ? H : 1);
    uint32_t t[NT];
    const int sh = w - P - 1;
    if (H && (sh & H)) {
        BL_UNROLL
        for (int i = 0; i < NT; ++i) t[i] = key[i + H];
    } else {
        BL_UNROLL
        for (int i = 0; i < NT; ++i) t[i] = key[i];
    }
    // stage b keeps t[0 .. NW + b) valid: exactly what the later stages still read
    if (H > 8) shift_stage<NW + 8, 8>(t, sh);
    if (H > 4) shift_stage<NW + 4, 4>(t, sh);
    if (H > 2) shift_stage<NW + 2, 2>(t, sh);
    if (H > 1) shift_stage<NW + 1, 1>(t, sh);
    BL_UNROLL
    for (int i = 0; i < NW; ++i) {  // the two ranges overlap when w < 2P: the same element on both sides gives d = 0
        const uint32_t x = key[i], y = t[i + 1];
        a[i] = x < y ? x : y;
        fold_min(dmin, (x ^ y) - 1u);
    }
    return dmin;
}

BL_DEV uint32_t packed_key7(uint32_t hash_hi, int x) { return (hash_hi & ~127u) | (uint32_t)x; }
BL_DEV uint32_t pad_key7(int x) { return ((0x1ffffffu - (uint32_t)x) << 7) | (uint32_t)x; }

BL_DEV uint32_t packed_key(uint32_t hash_hi, int x, bool left) { return (hash_hi & ~63u) | (uint32_t)(left ? x : 63 - x); }

}  // namespace bl
