// bl_pgzip.hpp — the pieces of decoding ONE gzip stream on many host threads (bl_ingest.cpp: ByteSource::inflate_parallel).
//
// A deflate stream has no entry points: a block can begin at any bit, and its matches reach up to 32 KiB back into text that
// the blocks before it produced.  What is done here (the scheme published with pugz and rapidgzip, written from RFC 1951):
//   * find_block:   try every bit position from some byte on as the start of a non-final dynamic-Huffman block and take the
//                   first whose header is a valid one (complete code-length code, lengths that decode without overrun, an
//                   end-of-block code, complete literal/length and distance codes) — a false find is possible and is caught
//                   later, because the decoder of the part before never arrives at it;
//   * SymbolDecoder: inflate from such a position WITHOUT the window: the output is 16-bit symbols, a byte, or a marker
//                   0x8000 | i for "byte i of the 32 KiB before this part" — copied on by later matches like any other symbol;
//   * resolve:      once the part before is text, the markers are looked up in its last 32 KiB.
// The order of the parts, the check that one part ends at the bit where the next began, the CRC-32 of the members and
// everything unusual (zlib decodes whatever lies between a part's end and the next part's start, and so has the last word on
// damaged or odd streams) are the caller's.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace blpg {

constexpr uint32_t WINDOW = 32768;
constexpr uint64_t NPOS = ~0ull;
constexpr uint16_t MARK = 0x8000;

// LSB-first bit reader over the whole (memory-mapped) file; bytes beyond the end read as zero and are noticed afterwards
struct BitIn {
    const uint8_t* base = nullptr;
    uint64_t size = 0;
    const uint8_t* p = nullptr;  // next byte to load
    uint64_t buf = 0;
    unsigned cnt = 0;  // valid bits in buf

    void open(const uint8_t* b, uint64_t n, uint64_t bit)
    {
        base = b;
        size = n;
        p = b + (bit >> 3);
        buf = 0;
        cnt = 0;
        refill();
        drop((unsigned)(bit & 7));
    }
    uint64_t bitpos() const { return 8 * (uint64_t)(p - base) - cnt; }
    bool over() const { return bitpos() > 8 * size; }
    // at least 56 valid bits afterwards; false once the reader is well beyond the end of the file (a loop that decodes the
    // zeros there must stop)
    inline bool refill()
    {
        if (__builtin_expect(p + 8 <= base + size, 1)) {
            uint64_t w;
            std::memcpy(&w, p, 8);
            buf |= w << cnt;  // (bits above the whole bytes taken are the next byte's: the same bits arrive again later)
            p += (63 - cnt) >> 3;
            cnt |= 56;
            return true;
        }
        while (cnt <= 56) {
            if (p < base + size) buf |= (uint64_t)*p << cnt;
            ++p;
            cnt += 8;
        }
        return p <= base + size + 16;
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(unsigned n)
    {
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t take(unsigned n)
    {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    void align_to_byte() { drop(cnt & 7); }
};

// A canonical prefix code (RFC 1951 §3.2.2): a direct table for codes of up to BITS bits, the counting walk for longer ones.
template <int BITS, int MAXSYM>
struct Code {
    uint16_t table[1 << BITS];  // (symbol << 4) | length, 0: longer than BITS (or no such code)
    uint16_t count[16];
    uint16_t sorted[MAXSYM];
    // 0: complete; 1: incomplete with codes of one bit only (zlib allows that); -1: over-subscribed or incomplete otherwise
    int build(const uint8_t* lens, int n)
    {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        count[0] = 0;
        int left = 1, max = 0;
        for (int l = 1; l < 16; ++l) {
            left = 2 * left - count[l];
            if (left < 0) return -1;
            if (count[l]) max = l;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        std::memset(table, 0, sizeof(table));
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= BITS && l < 16; ++l) {
            for (int c = 0; c < count[l]; ++c, ++idx, ++code) {
                uint32_t rev = 0;  // codes are packed starting from their most significant bit
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)((sorted[idx] << 4) | l);
                for (uint32_t x = rev; x < (1u << BITS); x += 1u << l) table[x] = e;
            }
            code <<= 1;
        }
        if (left > 0) return max <= 1 ? 1 : -1;
        return 0;
    }
    // a code longer than BITS, from the bits at the front of `in` (>= 15 of them valid): (symbol << 4) | length, 0: none
    uint32_t walk(const BitIn& in) const
    {
        uint32_t code = 0, first = 0, index = 0;
        const uint64_t bits = in.buf;
        for (int l = 1; l < 16; ++l) {
            code |= (uint32_t)((bits >> (l - 1)) & 1u);
            const uint32_t c = count[l];
            if (code < first + c) return (uint32_t)(sorted[index + (code - first)] << 4) | (uint32_t)l;
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        return 0;
    }
    inline uint32_t decode(const BitIn& in) const
    {
        const uint32_t e = table[in.buf & ((1u << BITS) - 1)];
        return __builtin_expect(e != 0, 1) ? e : walk(in);
    }
};

typedef Code<12, 288> LitLenCode;
typedef Code<10, 32> DistCode;
typedef Code<7, 19> LensCode;

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t LENS_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The header of a dynamic block behind its three type bits (RFC 1951 §3.2.7): fills `lens` (literal/length then distance
// lengths), returns false on anything zlib would refuse.
inline bool read_dynamic_header(BitIn& in, uint8_t* lens, int& hlit, int& hdist, LensCode& lc)
{
    in.refill();
    hlit = (int)in.take(5) + 257;
    hdist = (int)in.take(5) + 1;
    const int hclen = (int)in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    in.refill();
    for (int i = 0; i < hclen; ++i) {
        if (i == 16) in.refill();
        cl[LENS_ORDER[i]] = (uint8_t)in.take(3);
    }
    uint32_t kraft = 0;
    for (int i = 0; i < 19; ++i)
        if (cl[i]) kraft += 128u >> cl[i];
    if (kraft != 128u) return false;  // zlib: the code-length code must be complete
    if (lc.build(cl, 19) != 0) return false;
    int n = 0;
    const int total = hlit + hdist;
    while (n < total) {
        if (!in.refill()) return false;
        const uint32_t e = lc.decode(in);
        if (!e) return false;
        in.drop(e & 15u);
        const uint32_t sym = e >> 4;
        if (sym < 16) {
            lens[n++] = (uint8_t)sym;
        } else {
            uint8_t v = 0;
            int rep;
            if (sym == 16) {
                if (n == 0) return false;
                v = lens[n - 1];
                rep = 3 + (int)in.take(2);
            } else if (sym == 17) {
                rep = 3 + (int)in.take(3);
            } else {
                rep = 11 + (int)in.take(7);
            }
            if (n + rep > total) return false;
            while (rep--) lens[n++] = v;
        }
    }
    if (lens[256] == 0) return false;  // no end-of-block code
    return !in.over();
}

// First bit position in [from_bit, to_bit) that opens a valid non-final dynamic block, NPOS if there is none.
inline uint64_t find_block(const uint8_t* base, uint64_t size, uint64_t from_bit, uint64_t to_bit)
{
    static thread_local LitLenCode ll;
    static thread_local DistCode dc;
    LensCode lc;
    uint8_t lens[320];
    const uint64_t last = 8 * size;
    if (to_bit > last) to_bit = last;
    for (uint64_t bit = from_bit; bit + 17 + 12 <= last && bit < to_bit; ++bit) {
        const uint64_t byte = bit >> 3;
        uint32_t w = 0;
        for (int i = 0; i < 4 && byte + i < size; ++i) w |= (uint32_t)base[byte + i] << (8 * i);
        const uint32_t x = w >> (bit & 7);
        if ((x & 7u) != 4u) continue;            // BFINAL = 0, BTYPE = 10
        if (((x >> 3) & 31u) > 29u) continue;    // HLIT
        if (((x >> 8) & 31u) > 29u) continue;    // HDIST
        BitIn in;
        in.open(base, size, bit + 3);
        int hlit, hdist;
        if (!read_dynamic_header(in, lens, hlit, hdist, lc)) continue;
        if (ll.build(lens, hlit) != 0) continue;
        // the distance code: complete, or one code at most
        int used = 0;
        for (int i = 0; i < hdist; ++i) used += lens[hlit + i] != 0;
        const int rc = dc.build(lens + hlit, hdist);
        if (rc < 0 || (rc != 0 && used > 1)) continue;
        return bit;
    }
    return NPOS;
}

// gzip member header at byte `at` (RFC 1952 §2.3): the offset of the deflate data behind it, NPOS if there is no good header
inline uint64_t skip_member_header(const uint8_t* p, uint64_t size, uint64_t at)
{
    if (at + 10 > size || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || (p[at + 3] & 0xe0)) return NPOS;
    const uint8_t flg = p[at + 3];
    uint64_t q = at + 10;
    if (flg & 4) {  // FEXTRA
        if (q + 2 > size) return NPOS;
        const uint64_t xlen = (uint64_t)p[q] | ((uint64_t)p[q + 1] << 8);
        q += 2 + xlen;
    }
    for (int bit = 8; bit <= 16; bit <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            while (q < size && p[q]) ++q;
            ++q;
        }
    if (flg & 2) q += 2;  // FHCRC
    return q <= size ? q : NPOS;
}

struct MemberEnd {
    uint64_t out_off;  // symbols of this part that belong to the member (and those before it)
    uint32_t crc, isize;
};

// Symbol storage that grows without being copied or cleared (realloc of a large block moves pages, it does not touch them)
struct SymbolBuffer {
    uint16_t* p = nullptr;
    uint64_t cap = 0;
    SymbolBuffer() = default;
    SymbolBuffer(const SymbolBuffer&) = delete;
    SymbolBuffer& operator=(const SymbolBuffer&) = delete;
    ~SymbolBuffer() { std::free(p); }
    bool reserve(uint64_t n)
    {
        if (n <= cap) return true;
        void* q = std::realloc(p, n * sizeof(uint16_t));
        if (!q) return false;
        p = static_cast<uint16_t*>(q);
        cap = n;
        return true;
    }
    // give memory back when a part needed far more than parts usually do (a run of highly compressible text): the slots live as
    // long as the stream is read, and with 16 workers their high-water marks used to add up to gigabytes
    void shrink_to(uint64_t n)
    {
        if (cap <= n) return;
        std::free(p);
        p = nullptr;
        cap = 0;
    }
};

// One part of the stream as symbols.
struct Part {
    uint64_t start_bit = NPOS, end_bit = NPOS;  // [first block's first bit, the bit where the next block begins)
    bool at_eof = false;                        // the stream ended (after a member's trailer) where this part ends
    SymbolBuffer sym;
    uint64_t n = 0;  // symbols in use
    std::vector<MemberEnd> ends;
};

class SymbolDecoder {
public:
    // Decode blocks from `start_bit` until one ends at or beyond `stop_bit` (or the stream ends, or the part holds `max_out`
    // symbols, or something is wrong: then the part ends with the last block that was fine).  false: not even one block.
    bool run(const uint8_t* base, uint64_t size, uint64_t start_bit, uint64_t stop_bit, uint64_t max_out, Part& part)
    {
        in_.open(base, size, start_bit);
        part.start_bit = start_bit;
        part.n = 0;
        part.ends.clear();
        part.at_eof = false;
        part_ = &part;
        no_memory_ = false;
        if (!part.sym.reserve(1u << 20)) return false;
        out_ = part.sym.p;
        cap_ = part.sym.cap;
        o_ = 0;
        uint64_t good_bit = start_bit, good_o = 0;
        size_t good_ends = 0;
        bool any = false;
        for (;;) {
            in_.refill();
            const uint32_t hdr = in_.take(3);
            bool ok;
            switch (hdr >> 1) {
                case 0: ok = stored(); break;
                case 1: ok = fixed(); break;
                case 2: ok = dynamic(); break;
                default: ok = false;
            }
            if (ok && (hdr & 1u)) ok = member_end(base, size);
            if (!ok || in_.over()) break;
            any = true;
            good_bit = in_.bitpos();
            good_o = o_;
            good_ends = part.ends.size();
            if (part.at_eof || good_bit >= stop_bit || o_ >= max_out) break;
        }
        part.end_bit = good_bit;
        part.n = good_o;
        part.ends.resize(good_ends);
        if (part.at_eof && good_bit != in_.bitpos()) part.at_eof = false;
        return any;
    }

private:
    BitIn in_;
    uint16_t* out_ = nullptr;
    uint64_t cap_ = 0, o_ = 0;
    Part* part_ = nullptr;
    LitLenCode ll_;
    DistCode dc_;
    LensCode lc_;
    LitLenCode fixed_ll_;
    DistCode fixed_dc_;
    bool have_fixed_ = false;

    bool no_memory_ = false;
    bool grow(uint64_t need)
    {
        uint64_t c = cap_;
        while (c < need) c += c / 2;
        if (!part_->sym.reserve(c)) {
            no_memory_ = true;
            return false;
        }
        out_ = part_->sym.p;
        cap_ = part_->sym.cap;
        return true;
    }
    // after a final block: trailer, then the next member's header or the end of the file
    bool member_end(const uint8_t* base, uint64_t size)
    {
        in_.align_to_byte();
        const uint64_t at = in_.bitpos() >> 3;
        if (in_.over() || at + 8 > size) return false;
        auto le32 = [&](uint64_t q) { return (uint32_t)base[q] | ((uint32_t)base[q + 1] << 8) | ((uint32_t)base[q + 2] << 16) | ((uint32_t)base[q + 3] << 24); };
        part_->ends.push_back(MemberEnd{o_, le32(at), le32(at + 4)});
        if (at + 8 == size) {
            part_->at_eof = true;
            in_.open(base, size, 8 * size);
            return true;
        }
        const uint64_t data = skip_member_header(base, size, at + 8);
        if (data == NPOS) return false;
        in_.open(base, size, 8 * data);
        return true;
    }
    bool stored()
    {
        in_.align_to_byte();
        in_.refill();
        const uint32_t len = in_.take(16), nlen = in_.take(16);
        if ((len ^ nlen) != 0xffffu) return false;
        const uint64_t at = in_.bitpos() >> 3;
        if (in_.over() || at + len > in_.size) return false;
        if (o_ + len > cap_ && !grow(o_ + len)) return false;
        for (uint32_t i = 0; i < len; ++i) out_[o_ + i] = in_.base[at + i];
        o_ += len;
        in_.open(in_.base, in_.size, 8 * (at + len));
        return true;
    }
    bool fixed()
    {
        if (!have_fixed_) {
            uint8_t lens[288];
            for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            fixed_ll_.build(lens, 288);
            uint8_t dl[32];
            for (int i = 0; i < 32; ++i) dl[i] = 5;
            fixed_dc_.build(dl, 32);
            have_fixed_ = true;
        }
        return symbols(fixed_ll_, fixed_dc_);
    }
    bool dynamic()
    {
        uint8_t lens[320];
        int hlit, hdist;
        if (!read_dynamic_header(in_, lens, hlit, hdist, lc_)) return false;
        if (ll_.build(lens, hlit) < 0) return false;
        if (dc_.build(lens + hlit, hdist) < 0) return false;
        return symbols(ll_, dc_);
    }
    // the symbols of one block, up to and including its end-of-block code
    bool symbols(const LitLenCode& ll, const DistCode& dc)
    {
        BitIn in = in_;
        uint16_t* out = out_;
        uint64_t o = o_, cap = cap_;
        bool ok = false;
        for (;;) {
            if (__builtin_expect(o + 300 > cap, 0)) {
                o_ = o;
                if (!grow(o + 300 + (1u << 16))) break;
                out = out_;
                cap = cap_;
            }
            if (__builtin_expect(!in.refill(), 0)) break;
            uint32_t e = ll.decode(in);
            if (__builtin_expect(e == 0, 0)) break;
            in.drop(e & 15u);
            uint32_t sym = e >> 4;
            if (sym < 256) {
                out[o++] = (uint16_t)sym;
                // a second literal from the same refill (>= 41 bits are left)
                e = ll.decode(in);
                if (__builtin_expect(e == 0, 0)) break;
                sym = e >> 4;
                if (sym >= 256) goto not_literal;
                in.drop(e & 15u);
                out[o++] = (uint16_t)sym;
                continue;
            not_literal:
                in.drop(e & 15u);
                in.refill();
            }
            if (sym == 256) {
                ok = true;
                break;
            }
            if (sym > 285) break;
            sym -= 257;
            const uint32_t len = LEN_BASE[sym] + in.take(LEN_EXTRA[sym]);
            const uint32_t de = dc.decode(in);
            if (__builtin_expect(de == 0, 0)) break;
            in.drop(de & 15u);
            const uint32_t ds = de >> 4;
            if (ds > 29) break;
            const uint32_t dist = DIST_BASE[ds] + in.take(DIST_EXTRA[ds]);
            if (__builtin_expect(dist <= o, 1)) {
                const uint16_t* s = out + (o - dist);
                uint16_t* d = out + o;
                if (dist >= len) {
                    std::memcpy(d, s, 2 * (size_t)len);
                } else {
                    for (uint32_t j = 0; j < len; ++j) d[j] = s[j];
                }
            } else {
                if (dist - o > WINDOW) break;  // further back than any window reaches
                for (uint32_t j = 0; j < len; ++j) {
                    const int64_t s = (int64_t)(o + j) - (int64_t)dist;
                    out[o + j] = s >= 0 ? out[s] : (uint16_t)(MARK | (uint32_t)(s + (int64_t)WINDOW));
                }
            }
            o += len;
        }
        in_ = in;
        o_ = o;
        return ok && !in.over();
    }
};

// Symbols -> bytes through one table: a byte stands for itself, the marker 0x8000 | i for byte i of the 32 KiB before the part
// (window[32767] is the last byte before it).
struct SymbolTable {
    uint8_t byte[65536];
    void set(const uint8_t* window)
    {
        for (int i = 0; i < 256; ++i) byte[i] = (uint8_t)i;
        std::memcpy(byte + MARK, window, WINDOW);
    }
};
inline void resolve(const uint16_t* sym, uint64_t n, const SymbolTable& t, uint8_t* dst)
{
    uint64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint8_t a = t.byte[sym[i]], b = t.byte[sym[i + 1]], c = t.byte[sym[i + 2]], d = t.byte[sym[i + 3]];
        dst[i] = a;
        dst[i + 1] = b;
        dst[i + 2] = c;
        dst[i + 3] = d;
    }
    for (; i < n; ++i) dst[i] = t.byte[sym[i]];
}
// Near the start of the stream only the last `known` bytes of the window exist: does every marker point at one of them?
inline bool markers_known(const uint16_t* sym, uint64_t n, uint32_t known)
{
    const uint16_t first = (uint16_t)(MARK | (WINDOW - known));
    bool good = true;
    for (uint64_t i = 0; i < n; ++i) good = good && (sym[i] < MARK || sym[i] >= first);
    return good;
}

}  // namespace blpg
