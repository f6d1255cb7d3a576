// bl_inflate_core.hpp — DEFLATE (RFC 1951) decoder for ONE wavefront per BGZF member, written from the format.
//
// A BGZF file (bgzip, SAM spec §4.1) is a chain of gzip members of at most 64 KiB of text each, every one an independent deflate
// stream: the members of a span are inflated side by side on the GPU, one wave each, so that compressed FASTA / FASTQ crosses
// PCIe compressed and the host only reads the file and walks the member headers (bl_ingest.cpp).
//
// How one wave decodes a stream.  Decoding is sequential (a symbol's position depends on the lengths of all before it), so the
// 64 lanes run the SAME decoder on the same bits: every lookup is an LDS broadcast and every branch is uniform.  The lanes
// earn their keep where the format has width:
//   * building a code's lookup table: one symbol per lane
//   * a match (length, distance): up to 258 bytes copied 64 at a time inside the window
//   * writing text to HBM: the window is flushed in 16 KiB halves, 16 bytes per lane per store
//   * the input: each lane holds one dword of the current 256-byte chunk; the decoder takes the next dword with a readlane,
//     and the following chunk is already on its way
// The last 32 KiB of text (the deflate window) live in LDS as a ring addressed by the text position plus the low four bits of
// the member's destination address, so that 16-byte pieces of ring and of HBM line up whatever the destination is.
//
// A damaged stream must end in an error, never in a fault or a hang: every output position is checked against the member's
// stated size, every distance against the text produced so far, every code set against the Kraft sum (over-subscribed and
// incomplete sets are refused the way zlib refuses them), and the symbol loop stops as soon as it has consumed more bits than
// the member holds.
//
// The same source compiles for the host (BL_INFLATE_EMU: lane loops instead of lanes, arrays instead of LDS) so that the
// decoder is tested on the CPU against zlib over fuzzed and damaged streams before it ever runs on a GPU (tests/emu/).
#pragma once
#include <cstdint>

#ifdef BL_INFLATE_EMU
#define BL_IDEV inline
#define BL_LANES(l) for (int l = 0; l < 64; ++l)
#define BL_WAVE_SYNC() ((void)0)
#define BL_UNI(x) (x)
#else
#define BL_IDEV __device__ __forceinline__
#define BL_LANES(l) for (int l = (int)(threadIdx.x & 63u), once_ = 1; once_; once_ = 0)
// A value every lane has read from the same LDS address is the same in every lane, but the compiler cannot know: told so, it
// keeps the decoder's state in scalar registers and its branches scalar.  (Left alone, one such value reaching the loop state
// makes the whole symbol loop "divergent": masks instead of conditions, several extra jumps per symbol.)
#define BL_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// LDS operations of one wave complete in program order; the fence keeps the compiler from moving accesses across the point
// where the lanes exchange roles (written by one lane, read by another)
#define BL_WAVE_SYNC()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)
#endif

namespace bl_inflate {

constexpr int WINDOW = 32768;  // bytes of text kept in LDS: the furthest a match may reach back
constexpr int FLUSH = 16384;   // the window goes to HBM in halves
constexpr int LL_ROOT = 11;    // literal/length codes up to this many bits decode with one lookup
constexpr int D_ROOT = 9;      // distance codes
constexpr int CL_ROOT = 7;     // code-length codes are never longer
constexpr int MAX_LL = 288, MAX_D = 32, MAX_CODES = MAX_LL + MAX_D;

enum Status : uint32_t {
    OK = 0,
    ERR_BLOCK_TYPE = 1,    // reserved block type
    ERR_STORED = 2,        // stored block: LEN / NLEN mismatch
    ERR_HEADER = 3,        // dynamic block: too many codes, bad repeat, no end-of-block code
    ERR_CODE_SET = 4,      // over-subscribed or incomplete code
    ERR_SYMBOL = 5,        // bits that are no code, or a reserved symbol
    ERR_DISTANCE = 6,      // match reaches in front of the member's text
    ERR_OVERRUN = 7,       // more text than the member says it holds
    ERR_INPUT = 8,         // ran past the end of the member's data
    ERR_SIZE = 9,          // stream ended with less text than stated
};

struct CodeTable {  // canonical decoder for the codes longer than the root: symbols ordered by (length, symbol)
    uint32_t count[16];
};

struct Shared {
    uint8_t ring[WINDOW];
    uint16_t ll_table[1 << LL_ROOT];  // (symbol << 4) | length, 0 = longer than the root (or no code)
    uint16_t d_table[1 << D_ROOT];
    uint16_t cl_table[1 << CL_ROOT];
    uint8_t lens[MAX_CODES + 16];      // code lengths: literal/length codes, then distance codes
    uint16_t code[MAX_CODES];          // canonical code of each symbol
    uint16_t ll_sorted[MAX_LL], d_sorted[MAX_D], cl_sorted[32];
    CodeTable ll, d, cl;
    uint32_t next[16];                 // table building: next free slot of each length among the sorted symbols
};

// Input: dwords of the member's deflate data.  `word(i)` may be asked for any i >= 0 in increasing order; beyond the data it
// returns zeros (the caller notices from the bit count).
#ifdef BL_INFLATE_EMU
struct Input {
    const uint8_t* data;
    uint32_t n_bytes;
    uint32_t taken = 0;  // dwords handed out
    Input(const uint8_t* d, uint32_t n) : data(d), n_bytes(n) {}
    uint32_t peek_word() const  // the next dword, not consumed yet
    {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) {
            const uint32_t at = taken * 4 + b;
            if (at < n_bytes) w |= (uint32_t)data[at] << (8 * b);
        }
        return w;
    }
    void advance(uint32_t n) { taken += n; }  // n = 0 or 1
    uint32_t next_word()
    {
        const uint32_t w = peek_word();
        advance(1);
        return w;
    }
    uint32_t words_taken() const { return taken; }
    // the symbol loop's form: nothing is counted per word; `over` rises (at the latest 64 words late) once more words have been
    // taken than `limit`
    uint32_t over = 0;
    uint32_t next_word_hot(uint32_t limit)
    {
        const uint32_t w = next_word();
        if ((taken & 63u) == 0 && taken > limit) over = 1;
        return w;
    }
    uint32_t lead_bits() const { return 0; }
};
#else
struct Input {
    const uint32_t* words;  // the data's first byte lies in words[0] (lead bytes in front of it are skipped by the bit reader)
    uint32_t max_word;      // highest index that may be loaded (inside the packed buffer)
    uint32_t lead;          // bytes of words[0] in front of the data
    uint32_t cur, nxt;      // this lane's dword of the current / next 64-dword chunk
    uint32_t chunk = 0, idx = 0;
    uint32_t over = 0;  // more words taken than the data holds (noticed when a chunk is finished: at most 64 words late)
    // (Tried: the scalar memory path — every lane wants the same dword — with the next dword's load issued as the current one
    // is taken.  The compiler loads into a temporary and waits for it at once to copy it into the loop-carried register, so
    // every refill paid a scalar-load latency; one wait per 64 dwords, as here, is cheaper.)
    __device__ __forceinline__ uint32_t load(uint32_t c) const
    {
        uint32_t i = c * 64u + (threadIdx.x & 63u);
        if (i > max_word) i = max_word;
        return __builtin_nontemporal_load(words + i);
    }
    __device__ __forceinline__ Input(const uint8_t* data, uint32_t n_bytes, const uint8_t* buffer_end)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(data);
        lead = (uint32_t)(a & 3u);
        words = reinterpret_cast<const uint32_t*>(a - lead);
        const uintptr_t last = (reinterpret_cast<uintptr_t>(buffer_end) & ~(uintptr_t)3) - 4;  // last whole dword inside the buffer
        max_word = last >= reinterpret_cast<uintptr_t>(words) ? (uint32_t)((last - reinterpret_cast<uintptr_t>(words)) >> 2) : 0u;
        (void)n_bytes;
        cur = load(0);
        nxt = load(1);
    }
    __device__ __forceinline__ uint32_t peek_word() const { return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)idx); }
    __device__ __forceinline__ void advance(uint32_t n)  // n = 0 or 1
    {
        idx += n;
        if (idx == 64u) {  // once per 256 bytes of input
            idx = 0;
            cur = nxt;
            ++chunk;
            nxt = load(chunk + 1);
        }
    }
    __device__ __forceinline__ uint32_t next_word()
    {
        const uint32_t w = peek_word();
        advance(1);
        return w;
    }
    __device__ __forceinline__ uint32_t words_taken() const { return chunk * 64u + idx; }
    __device__ __forceinline__ uint32_t next_word_hot(uint32_t limit)  // the symbol loop's form
    {
        const uint32_t w = peek_word();
        if (++idx == 64u) {
            idx = 0;
            cur = nxt;
            ++chunk;
            nxt = load(chunk + 1);
            over |= (limit - chunk * 64u) >> 31;  // (both far below 2^31)
        }
        return w;
    }
    __device__ __forceinline__ uint32_t lead_bits() const { return lead * 8u; }
};
#endif

// The two tables every symbol goes through leave LDS for registers once a block's codes are built, and grow on the way from
// (symbol, code length) to everything the symbol loop needs, so that the loop neither computes nor looks up anything else:
//   literal/length entry   bits 0-3 code length (0: longer than the root, or no code), 4-7 number of extra bits,
//                          8-15 the literal, 16-24 the base length, bit 31 literal, bit 30 end of block, bit 29 reserved symbol
//   distance entry         bits 0-3 code length, 4-7 number of extra bits, 8-22 the base distance, bit 29 reserved symbol
// Entry i lives in lane i % 64 of register i / 64 (32 registers for 2048 literal/length entries, 8 for 512 distance entries);
// a lookup is an indexed register move and a readlane.  One wave per SIMD is all the 32 KiB window leaves room for, and such a
// wave issues one instruction every four cycles at best: what the loop costs is its instruction count, and memory latency on
// top; both are what this layout removes.
constexpr uint32_t E_LITERAL = 1u << 31, E_END = 1u << 30, E_RESERVED = 1u << 29;

BL_IDEV uint32_t expand_ll(uint32_t sym, uint32_t code_len)
{
    if (sym < 256u) return E_LITERAL | (sym << 8) | code_len;
    if (sym == 256u) return E_END | code_len;
    if (sym > 285u) return E_RESERVED | code_len;
    const uint32_t c = sym - 257u;
    uint32_t base, extra;
    if (c < 8u) { base = c + 3u; extra = 0; }
    else if (c == 28u) { base = 258u; extra = 0; }
    else { extra = (c >> 2) - 1u; base = ((4u + (c & 3u)) << extra) + 3u; }
    return (base << 16) | (extra << 4) | code_len;
}
BL_IDEV uint32_t expand_d(uint32_t sym, uint32_t code_len)
{
    if (sym > 29u) return E_RESERVED | code_len;
    uint32_t base, extra;
    if (sym < 4u) { base = sym + 1u; extra = 0; }
    else { extra = (sym >> 1) - 1u; base = ((2u + (sym & 1u)) << extra) + 1u; }
    return (base << 8) | (extra << 4) | code_len;
}

#ifdef BL_INFLATE_EMU
template <int NREG, bool DIST>
struct RegTable {
    uint32_t e[NREG * 64];
    void load(const uint16_t* table)
    {
        for (int i = 0; i < NREG * 64; ++i) {
            const uint32_t t = table[i];
            e[i] = (t & 15u) ? (DIST ? expand_d(t >> 4, t & 15u) : expand_ll(t >> 4, t & 15u)) : 0u;
        }
    }
    uint32_t lookup(uint32_t i) const { return e[i]; }
};
#else
template <int NREG, bool DIST>
struct RegTable {
    typedef uint32_t regs_t __attribute__((ext_vector_type(NREG < 16 ? 16 : NREG)));  // (under 16 elements the compiler indexes a
    regs_t regs;                                                                       // vector with a chain of selects, not a move)
    __device__ __forceinline__ void load(const uint16_t* table)
    {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const uint32_t t = table[r * 64 + (int)(threadIdx.x & 63u)];
            regs[r] = (t & 15u) ? (DIST ? expand_d(t >> 4, t & 15u) : expand_ll(t >> 4, t & 15u)) : 0u;
        }
    }
    __device__ __forceinline__ uint32_t lookup(uint32_t i) const
    {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i >> 6)), l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i & 63u));
        return (uint32_t)__builtin_amdgcn_readlane((int)regs[r], (int)l);
    }
};
#endif
typedef RegTable<(1 << LL_ROOT) / 64, false> LLRegs;
typedef RegTable<(1 << D_ROOT) / 64, true> DRegs;

struct Bits {
    uint64_t bb = 0;
    int nb = 0;
    template <class In>
    BL_IDEV void start(In& in)
    {
        bb = in.next_word();
        nb = 32;
        const int skip = (int)in.lead_bits();
        bb >>= skip;
        nb -= skip;
    }
    template <class In>
    BL_IDEV bool need32(In& in, uint32_t word_limit)  // at least 33 bits afterwards; false: the input is used up (and more)
    {
        if (nb <= 32) {
            bb |= (uint64_t)in.next_word() << nb;
            nb += 32;
            if (in.words_taken() > word_limit) return false;
        }
        return true;
    }
    template <class In>
    BL_IDEV void refill(In& in)  // at least 33 bits afterwards, without a branch: the symbol loop's form of need32
    {
        const uint32_t take = nb <= 32 ? 1u : 0u;
        const uint64_t w = take ? (uint64_t)in.peek_word() : 0ull;
        bb |= w << (nb & 63);
        nb += (int)(take << 5);
        in.advance(take);
    }
    BL_IDEV uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    BL_IDEV void drop(int n)
    {
        bb >>= n;
        nb -= n;
    }
    BL_IDEV uint32_t take(int n)
    {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

BL_IDEV uint32_t reverse_bits(uint32_t v, int n)  // the low n bits of v, reversed
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    v = (v >> 16) | (v << 16);
    return v >> (32 - n);
}

// Build the decoder of one code from lens[0 .. n): `table` answers codes of at most `root` bits, `sorted` + `ct` the longer
// ones.  Returns false for an over-subscribed set and for an incomplete one — except, when `lone_code_ok`, a set of one 1-bit
// code or of no code at all (what zlib lets through for the literal/length and distance codes, not for the code-length code).
#ifndef BL_INFLATE_EMU
// The same on the device with one symbol per lane and register (up to 320 symbols = 5 registers): lengths are counted and the
// symbols of a length ranked with ballots, 75 of each, instead of two walks over the symbols in which every lane did the same
// thing 340 times.  (The host build keeps the walk: it is what the emulation tests check the tables against on the GPU, where
// tests/test_inflate.py sends thousands of different code sets through this version.)
__device__ __forceinline__ bool build_code(Shared& sh, const uint8_t* lens, int n, int root, uint16_t* table, uint16_t* sorted, CodeTable& ct, bool lone_code_ok)
{
    const int lane = (int)(threadIdx.x & 63u);
    for (int i = lane; i < (1 << root); i += 64) table[i] = 0;
    uint32_t my_len[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int s = r * 64 + lane;
        my_len[r] = s < n ? (uint32_t)(lens[s] & 15) : 0u;
    }
    uint32_t cnt[16];
    cnt[0] = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) c += (uint32_t)__popcll(__ballot(my_len[r] == (uint32_t)k));
        cnt[k] = c;
    }
    int left = 1, longest = 0;
    uint32_t first = 0, at = 0;  // canonical code / slot among the sorted symbols where each length starts
    uint32_t first_of[16], slot_of[16];
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - (int)cnt[l];
        if (left < 0) return false;  // over-subscribed
        if (cnt[l]) longest = l;
        first_of[l] = first;
        slot_of[l] = at;
        first = (first + cnt[l]) << 1;
        at += cnt[l];
    }
    if (left > 0 && !(lone_code_ok && (at == 0 || (longest == 1 && at == 1)))) return false;  // incomplete
    if (lane < 16) {
        uint32_t c = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l)
            if (l == lane) c = cnt[l];
        ct.count[lane] = c;
    }
    // the k-th symbol of a length (in symbol order: register by register, lane by lane) gets that length's k-th code and slot
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t my_code[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        uint32_t run = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const unsigned long long mask = __ballot(my_len[r] == (uint32_t)k);
            if (my_len[r] == (uint32_t)k) {
                const uint32_t rank = run + (uint32_t)__popcll(mask & below);
                sorted[slot_of[k] + rank] = (uint16_t)(r * 64 + lane);
                my_code[r] = first_of[k] + rank;
            }
            run += (uint32_t)__popcll(mask);
        }
    }
    BL_WAVE_SYNC();  // (the cleared table is in place)
    // every symbol of at most `root` bits fills the table entries whose low bits are its (bit-reversed) code
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const uint32_t l = my_len[r];
        if (l && l <= (uint32_t)root) {
            const uint32_t rev = reverse_bits(my_code[r], (int)l);
            const uint16_t e = (uint16_t)(((r * 64 + lane) << 4) | (int)l);
            for (uint32_t j = rev; j < (1u << root); j += (1u << l)) table[j] = e;
        }
    }
    BL_WAVE_SYNC();
    return true;
}
#else
BL_IDEV bool build_code(Shared& sh, const uint8_t* lens, int n, int root, uint16_t* table, uint16_t* sorted, CodeTable& ct, bool lone_code_ok)
{
    BL_LANES(lane)
    {
        if (lane < 16) ct.count[lane] = 0;
        for (int i = lane; i < (1 << root); i += 64) table[i] = 0;
    }
    BL_WAVE_SYNC();
    // how many codes of each length (every lane walks the same short list: a histogram by atomics would not be faster)
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int s = 0; s < n; ++s) {
        const int l = (int)BL_UNI(lens[s] & 15);
#pragma unroll
        for (int k = 1; k < 16; ++k) cnt[k] += (l == k);
    }
    int left = 1, longest = 0;
    uint32_t first = 0, at = 0;  // canonical code / slot among the sorted symbols where each length starts
    uint32_t first_of[16], slot_of[16];
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - (int)cnt[l];
        if (left < 0) return false;  // over-subscribed
        if (cnt[l]) longest = l;
        first_of[l] = first;
        slot_of[l] = at;
        first = (first + cnt[l]) << 1;
        at += cnt[l];
    }
    if (left > 0 && !(lone_code_ok && (at == 0 || (longest == 1 && at == 1)))) return false;  // incomplete
    BL_LANES(lane)
    {
        if (lane >= 1 && lane < 16) {
            uint32_t c = 0, f = 0;
#pragma unroll
            for (int l = 1; l < 16; ++l)
                if (l == lane) { c = cnt[l]; f = slot_of[l]; }
            ct.count[lane] = c;
            sh.next[lane] = f;
        }
    }
    BL_WAVE_SYNC();
    // symbols in order: the k-th symbol of a length gets that length's k-th code and k-th sorted slot
    for (int s = 0; s < n; ++s) {
        const int l = (int)BL_UNI(lens[s] & 15);
        if (l) {
            uint32_t f = 0, base = 0;
#pragma unroll
            for (int k = 1; k < 16; ++k)
                if (k == l) { f = first_of[k]; base = slot_of[k]; }
            const uint32_t slot = BL_UNI(sh.next[l]);  // (every lane does the same here: no exchange between lanes)
            sh.next[l] = slot + 1;
            sorted[slot] = (uint16_t)s;
            sh.code[s] = (uint16_t)(f + (slot - base));
        }
    }
    BL_WAVE_SYNC();
    // every symbol of at most `root` bits fills the table entries whose low bits are its (bit-reversed) code
    BL_LANES(lane)
    {
        for (int s = lane; s < n; s += 64) {
            const int l = lens[s] & 15;
            if (l && l <= root) {
                const uint32_t r = reverse_bits(sh.code[s], l);
                const uint16_t e = (uint16_t)((s << 4) | l);
                for (uint32_t j = r; j < (1u << root); j += (1u << l)) table[j] = e;
            }
        }
    }
    BL_WAVE_SYNC();
    return true;
}

#endif

// A code longer than its table's root (or bits that are no code at all): walk the canonical code one bit at a time.
// Returns (symbol << 4) | code length without dropping the bits, or 0 when the bits are no code.  (One loop with one exit, not
// unrolled: rare, and the symbol loop around it should stay small.)
BL_IDEV uint32_t decode_long(const Bits& b, const uint16_t* sorted, const CodeTable& ct)
{
    uint32_t code = 0, first = 0, index = 0, found = 0;
    uint32_t bits = (uint32_t)b.bb;
#pragma nounroll
    for (uint32_t l = 1; l < 16 && !found; ++l) {
        code |= bits & 1u;
        bits >>= 1;
        const uint32_t c = BL_UNI(ct.count[l]);
        if (code < first + c) found = (BL_UNI(sorted[index + (code - first)]) << 4) | l;
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return found;
}

// One symbol of the code-length code (used a few hundred times per block: its table stays in LDS).
BL_IDEV int decode_cl(Bits& b, const Shared& sh)
{
    const uint32_t e = BL_UNI(sh.cl_table[b.peek(CL_ROOT)]);
    if (e) {
        b.drop((int)(e & 15u));
        return (int)(e >> 4);
    }
    const uint32_t f = decode_long(b, sh.cl_sorted, sh.cl);
    if (!f) return -1;
    b.drop((int)(f & 15u));
    return (int)(f >> 4);
}

BL_IDEV void copy16(uint8_t* dst, const uint8_t* src)  // both 16-byte aligned
{
#ifdef BL_INFLATE_EMU
    for (int i = 0; i < 16; ++i) dst[i] = src[i];
#else
    *reinterpret_cast<uint4*>(__builtin_assume_aligned(dst, 16)) = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(src, 16));
#endif
}

// Text written so far goes out to HBM: ring positions [a, b) (b - a <= WINDOW), `g + q` being the address of ring position q.
BL_IDEV void flush_range(const Shared& sh, uint8_t* g, uint32_t a, uint32_t b)
{
    uint32_t a16 = (a + 15u) & ~15u, b16 = b & ~15u;
    if (a16 > b16) a16 = b16 = b;  // less than one aligned piece: bytes only
    BL_LANES(lane)
    {
        for (uint32_t i = a + lane; i < a16 && i < b; i += 64) g[i] = sh.ring[i & (WINDOW - 1)];
        for (uint32_t c = a16 + 16u * lane; c < b16; c += 16u * 64u) {
            copy16(g + c, sh.ring + (c & (WINDOW - 1)));
        }
        for (uint32_t i = (b16 > a ? b16 : a) + lane; i < b; i += 64)
            if (i >= a16) g[i] = sh.ring[i & (WINDOW - 1)];
    }
}

// Inflate one member: `in` delivers its deflate data (n_in bytes), the text (exactly `isize` bytes if the stream is sound)
// goes to out[0 .. isize).  Nothing outside out[0 .. isize) is written.  Returns a Status.
template <class In>
BL_IDEV uint32_t inflate_member(Shared& sh, In& in, uint32_t n_in, uint8_t* out, uint32_t isize)
{
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);
    uint8_t* const g = out - shift;  // g + q: where ring position q = text position + shift goes
    uint32_t q = shift, flushed = shift, next_flush = FLUSH;
    const uint32_t q_end = shift + isize;
    // the symbol loop looks at ONE bound after each symbol: the nearer of "time to flush" and "the member's text is complete"
    uint32_t q_stop = next_flush < q_end ? next_flush : q_end;
    const uint32_t word_limit = (n_in + in.lead_bits() / 8u + 3u) / 4u + 2u;  // dwords that hold data, and two to finish a symbol on
    Bits b;
    b.start(in);
    uint32_t status = OK;
    LLRegs ll_regs;
    DRegs d_regs;
#define BL_NEED32()                                    \
    if (!b.need32(in, word_limit)) {                   \
        status = ERR_INPUT;                            \
        break;                                         \
    }
    for (;;) {
        BL_NEED32();
        const uint32_t last = b.take(1), type = b.take(2);
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, ~LEN, LEN bytes
            b.drop(b.nb & 7);
            BL_NEED32();
            const uint32_t len = b.take(16);
            BL_NEED32();
            const uint32_t nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu) { status = ERR_STORED; break; }
            if (q + len > q_end) { status = ERR_OVERRUN; break; }
            for (uint32_t i = 0; i < len; ++i) {
                BL_NEED32();
                sh.ring[q & (WINDOW - 1)] = (uint8_t)b.take(8);
                ++q;
                if (q >= next_flush) {
                    BL_WAVE_SYNC();
                    flush_range(sh, g, flushed, next_flush);
                    flushed = next_flush;
                    next_flush += FLUSH;
                }
            }
            if (status != OK) break;
            q_stop = next_flush < q_end ? next_flush : q_end;
        } else if (type == 3) {
            status = ERR_BLOCK_TYPE;
            break;
        } else {
            int n_ll, n_d;
            if (type == 1) {
                n_ll = 288;
                n_d = 32;  // (the two last of each are never used in a sound stream; they take part in the code all the same)
                BL_LANES(lane)
                {
                    for (int s = lane; s < 288 + 32; s += 64) sh.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                }
                BL_WAVE_SYNC();
            } else {
                n_ll = (int)b.take(5) + 257;
                n_d = (int)b.take(5) + 1;
                const int n_cl = (int)b.take(4) + 4;
                if (n_ll > 286 || n_d > 30) { status = ERR_HEADER; break; }
                BL_LANES(lane)
                {
                    if (lane < 19) sh.lens[lane] = 0;
                }
                BL_WAVE_SYNC();
                for (int i = 0; i < n_cl && status == OK; ++i) {
                    if (!b.need32(in, word_limit)) status = ERR_INPUT;
                    const uint8_t v = (uint8_t)b.take(3);
                    // the order in which the lengths of the code-length code are sent: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                    const int at = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? (19 - i) >> 1 : 6 + (i >> 1);
                    sh.lens[at] = v;  // (every lane stores the same byte)
                }
                if (status != OK) break;
                BL_WAVE_SYNC();
                if (!build_code(sh, sh.lens, 19, CL_ROOT, sh.cl_table, sh.cl_sorted, sh.cl, false)) { status = ERR_CODE_SET; break; }
                // the lengths of the two codes, run-length coded with the code just built (whose decoder no longer reads sh.lens:
                // the array is free for them)
                const int total = n_ll + n_d;
                int have = 0;
                uint8_t prev = 0;
                uint8_t* const lens = sh.lens;
                while (have < total) {
                    BL_NEED32();
                    const int s = decode_cl(b, sh);
                    if (s < 0) { status = ERR_SYMBOL; break; }
                    int rep;
                    uint8_t v;
                    if (s < 16) {
                        rep = 1;
                        v = (uint8_t)s;
                        prev = v;
                    } else if (s == 16) {
                        if (have == 0) { status = ERR_HEADER; break; }
                        rep = 3 + (int)b.take(2);
                        v = prev;
                    } else if (s == 17) {
                        rep = 3 + (int)b.take(3);
                        v = 0;
                        prev = 0;
                    } else {
                        rep = 11 + (int)b.take(7);
                        v = 0;
                        prev = 0;
                    }
                    if (have + rep > total) { status = ERR_HEADER; break; }
                    BL_LANES(lane)
                    {
                        for (int j = lane; j < rep; j += 64) {
                            const int i = have + j;
                            lens[i < n_ll ? i : MAX_LL + (i - n_ll)] = v;  // distance lengths start at slot MAX_LL
                        }
                    }
                    have += rep;
                }
                if (status != OK) break;
                BL_WAVE_SYNC();
                if (BL_UNI(sh.lens[256]) == 0) { status = ERR_HEADER; break; }  // no end-of-block code
            }
            if (!build_code(sh, sh.lens, n_ll, LL_ROOT, sh.ll_table, sh.ll_sorted, sh.ll, true)) { status = ERR_CODE_SET; break; }
            if (!build_code(sh, sh.lens + MAX_LL, n_d, D_ROOT, sh.d_table, sh.d_sorted, sh.d, true)) { status = ERR_CODE_SET; break; }
            ll_regs.load(sh.ll_table);
            d_regs.load(sh.d_table);
            // The symbols of the block.  The loop is written for the instruction count (a lone wave issues one instruction per
            // four cycles at best, and pays for every taken jump): a literal is a straight run back to the top; whatever can go
            // wrong is gathered in `bad` / `in.over` and looked at together with the one bound on q, and a wrong path only ever
            // touches the ring (whose index is masked) before it is noticed.
            uint32_t bad = 0;
            // a match of one round whose bytes have been read and not yet written (one byte per lane)
#ifdef BL_INFLATE_EMU
            uint8_t pend_v[64] = {0};
#define BL_LANE_SLOT(l) (l)
#else
            uint8_t pend_v[1] = {0};
#define BL_LANE_SLOT(l) 0
#endif
            bool pend_on = false;
            uint32_t pend_q = 0, pend_len = 0;
            // Two loops: the inner one decodes symbols and changes nothing but the bit buffer, q and the error bits, so that the
            // bound, the flush positions and the status are constants to it (as one loop, every symbol paid a dozen register
            // copies for the rare path that moves them); it is left at the bound (or on an error bit) and at the end of the block.
            for (;;) {
                bool block_done = false;
                // the ONE number a symbol is checked against: the bound, or 0 as soon as an error bit is up (looked at after
                // every match; a run of literals reaches the bound soon enough)
                uint32_t q_limit = (bad | in.over) ? 0u : q_stop;
                for (;;) {
                    if (b.nb <= 32) {
                        b.bb |= (uint64_t)in.next_word_hot(word_limit) << b.nb;
                        b.nb += 32;
                    }
                    uint32_t e = ll_regs.lookup(b.peek(LL_ROOT));
                    if (__builtin_expect((e & 15u) == 0, 0)) {  // rare: a code longer than the root
                        const uint32_t f = decode_long(b, sh.ll_sorted, sh.ll);
                        e = f ? expand_ll(f >> 4, f & 15u) : (E_RESERVED | 1u);
                    }
                    b.drop((int)(e & 15u));
                    if (e & E_LITERAL) {
                        sh.ring[q & (WINDOW - 1)] = (uint8_t)(e >> 8);  // (the same store from every lane; one byte past the text's
                        ++q;                                             // end is caught at the bound before anything leaves the ring)
                        if (__builtin_expect(q >= q_limit, 0)) break;
                        continue;
                    }
                    if (__builtin_expect((e & (E_END | E_RESERVED)) != 0, 0)) {  // end of block, or a reserved symbol
                        bad |= e & E_RESERVED;
                        block_done = true;
                        break;
                    }
                    const uint32_t len = (e >> 16) + b.take((int)((e >> 4) & 15u));
                    if (b.nb <= 32) {
                        b.bb |= (uint64_t)in.next_word_hot(word_limit) << b.nb;
                        b.nb += 32;
                    }
                    uint32_t de = d_regs.lookup(b.peek(D_ROOT));
                    if (__builtin_expect((de & 15u) == 0, 0)) {
                        const uint32_t f = decode_long(b, sh.d_sorted, sh.d);
                        de = f ? expand_d(f >> 4, f & 15u) : (E_RESERVED | 1u);
                    }
                    b.drop((int)(de & 15u));
                    const uint32_t dist = ((de >> 8) & 0x7fffu) + b.take((int)((de >> 4) & 15u));
                    bad |= (de & E_RESERVED) | (uint32_t)(dist > q - shift);
#ifdef BL_INFLATE_HIST  // (host harness only: how far back matches reach; tests/emu/emu_inflate.cpp --hist)
                    BL_INFLATE_HIST(dist, len);
#endif
                    // the previous match's bytes go into the ring now (see below), then the literals written since are in place
                    BL_LANES(lane)
                    {
                        if (pend_on && (uint32_t)lane < pend_len) sh.ring[(pend_q + lane) & (WINDOW - 1)] = pend_v[BL_LANE_SLOT(lane)];
                    }
                    pend_on = false;
                    BL_WAVE_SYNC();
                    const uint32_t from = q - dist;
                    if (dist >= len && len <= 64u) {
                        // one round: the bytes are READ now and WRITTEN when the next match (or the end of the run of symbols)
                        // needs them in place — the symbols decoded meanwhile hide the LDS round trip
                        BL_LANES(lane)
                        {
                            pend_v[BL_LANE_SLOT(lane)] = sh.ring[(from + lane) & (WINDOW - 1)];
                        }
                        pend_on = true;
                        pend_q = q;
                        pend_len = len;
                    } else if (dist >= len || dist >= 64u) {
                        // the source of every byte is in place before the round of 64 that writes it (rounds run one after the
                        // other: LDS operations of a wave keep their order)
                        BL_LANES(lane)
                        {
                            if ((uint32_t)lane < len) sh.ring[(q + lane) & (WINDOW - 1)] = sh.ring[(from + lane) & (WINDOW - 1)];
                        }
                        for (uint32_t base = 64u; base < len; base += 64u) {
                            BL_LANES(lane)
                            {
                                const uint32_t i = base + (uint32_t)lane;
                                if (i < len) sh.ring[(q + i) & (WINDOW - 1)] = sh.ring[(from + i) & (WINDOW - 1)];
                            }
                        }
                    } else if (dist <= 1u) {
                        // a run of one byte (quality values, poly-A; dist is 0 only on a wrong path)
                        BL_LANES(lane)
                        {
                            const uint8_t v = sh.ring[from & (WINDOW - 1)];
                            for (uint32_t i = (uint32_t)lane; i < len; i += 64) sh.ring[(q + i) & (WINDOW - 1)] = v;
                        }
                    } else {
                        // the match runs into itself with a short period: its `dist` last bytes repeat.  Rounds of R = the largest
                        // multiple of the period within 64 bytes: byte `lane` of every round is the same pattern byte, read once
                        const uint32_t per_round = BL_UNI((uint32_t)(64.0f / (float)dist));  // exact: both small integers
                        const uint32_t round = per_round * dist;
                        BL_LANES(lane)
                        {
                            int m = (int)lane - (int)((uint32_t)((float)lane / (float)dist) * dist);  // lane % dist, up to a rounding step
                            if (m < 0) m += (int)dist;
                            if (m >= (int)dist) m -= (int)dist;
                            const uint8_t pattern = sh.ring[(from + (uint32_t)m) & (WINDOW - 1)];
                            if ((uint32_t)lane < round)
                                for (uint32_t i = (uint32_t)lane; i < len; i += round) sh.ring[(q + i) & (WINDOW - 1)] = pattern;
                        }
                    }
                    BL_WAVE_SYNC();
                    q += len;
                    q_limit = (bad | in.over) ? 0u : q_stop;
                    if (__builtin_expect(q >= q_limit, 0)) break;
                }
                // the bound, an error bit, or the end of the block: the ring is brought up to date first
                BL_LANES(lane)
                {
                    if (pend_on && (uint32_t)lane < pend_len) sh.ring[(pend_q + lane) & (WINDOW - 1)] = pend_v[BL_LANE_SLOT(lane)];
                }
                pend_on = false;
                if (in.over) status = ERR_INPUT;
                else if (bad) status = (bad & E_RESERVED) ? ERR_SYMBOL : ERR_DISTANCE;
                else if (q > q_end) status = ERR_OVERRUN;
                if (status != OK || block_done) break;
                if (q >= next_flush) {
                    BL_WAVE_SYNC();
                    flush_range(sh, g, flushed, next_flush);
                    flushed = next_flush;
                    next_flush += FLUSH;
                }
                // at the text's end only the end-of-block code may follow: the bound then sits one past it
                q_stop = q == q_end ? q_end + 1u : (next_flush < q_end ? next_flush : q_end);
            }
            if (status != OK) break;
        }
        if (last) break;
    }
#undef BL_NEED32
#undef BL_LANE_SLOT
    BL_WAVE_SYNC();
    if (q > q_end) q = q_end;  // (a literal one past the end was refused above; it stays in the ring)
    if (q > flushed) flush_range(sh, g, flushed, q);  // what a damaged stream produced before it failed is within [0, isize) too
    if (status == OK) {
        const uint32_t bits_used = in.words_taken() * 32u - (uint32_t)b.nb - in.lead_bits();
        if (bits_used > n_in * 8u) status = ERR_INPUT;
        else if (q != q_end) status = ERR_SIZE;
    }
    return status;
}

}  // namespace bl_inflate
