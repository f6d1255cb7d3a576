// TEST INFRASTRUCTURE: the wave-per-member DEFLATE decoder of biolib_amd/csrc/bl_inflate_core.hpp compiled for the host
// (lane loops instead of lanes) and checked against zlib: streams of every block type and compression level, texts with long
// and short matches, and thousands of damaged streams, which must end in an error exactly when zlib reports one and may never
// write outside the member's text.  Built with -fsanitize=address,undefined by tests/emu/Makefile.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define BL_INFLATE_EMU 1
// match statistics of the decoded streams (--hist): matches and matched bytes by distance, in 4-KiB bins of the 32-KiB window
static unsigned long long g_hist_n[8], g_hist_bytes[8];
#define BL_INFLATE_HIST(dist, len) do { const unsigned b_ = (unsigned)(((dist) - 1u) >> 12) & 7u; ++g_hist_n[b_]; g_hist_bytes[b_] += (len); } while (0)
#include "../../biolib_amd/csrc/bl_inflate_core.hpp"

namespace {

std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& text, int level, int strategy, int mem_level = 8)
{
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level, Z_DEFLATED, -15, mem_level, strategy) != Z_OK) std::abort();
    std::vector<uint8_t> out(deflateBound(&z, text.size()) + 64);
    z.next_in = const_cast<uint8_t*>(text.data());
    z.avail_in = (uInt)text.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) std::abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// zlib's verdict on a raw stream that should hold `isize` bytes: the text, or empty + false
bool zlib_inflate(const std::vector<uint8_t>& packed, uint32_t isize, std::vector<uint8_t>& text)
{
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) std::abort();
    text.assign(isize + 1, 0);
    z.next_in = const_cast<uint8_t*>(packed.data());
    z.avail_in = (uInt)packed.size();
    z.next_out = text.data();
    z.avail_out = isize;
    const int rc = inflate(&z, Z_FINISH);
    const bool good = rc == Z_STREAM_END && z.total_out == isize;
    inflateEnd(&z);
    text.resize(isize);
    return good;
}

struct Result {
    uint32_t status;
    std::vector<uint8_t> text;
    bool wrote_outside;
};

Result ours(const std::vector<uint8_t>& packed, uint32_t isize, unsigned misalign)
{
    static bl_inflate::Shared sh;  // 39 KiB: what a wave has in LDS
    std::vector<uint8_t> buf(isize + 64 + 32, 0xA5);
    uint8_t* out = buf.data() + 16 + misalign;  // any alignment of the destination must work
    bl_inflate::Input in(packed.data(), (uint32_t)packed.size());
    Result r;
    r.status = bl_inflate::inflate_member(sh, in, (uint32_t)packed.size(), out, isize);
    r.text.assign(out, out + isize);
    r.wrote_outside = false;
    for (size_t i = 0; i < buf.size(); ++i)
        if ((buf.data() + i < out || buf.data() + i >= out + isize) && buf[i] != 0xA5) r.wrote_outside = true;
    return r;
}

int failures = 0;
void expect(bool ok, const char* what, size_t id)
{
    if (!ok) {
        std::printf("FAIL %s (case %zu)\n", what, id);
        ++failures;
    }
}

std::vector<uint8_t> make_text(std::mt19937_64& rng, int kind, size_t n)
{
    std::vector<uint8_t> t(n);
    switch (kind) {
    case 0:  // FASTQ-like
    {
        size_t at = 0;
        unsigned rec = 0;
        while (at < n) {
            char head[64];
            const int h = std::snprintf(head, sizeof(head), "@read%u/1 lane=3\n", rec++);
            std::string s(head, head + h);
            const int L = 30 + (int)(rng() % 200);
            for (int i = 0; i < L; ++i) s.push_back("ACGTN"[rng() % 100 < 2 ? 4 : rng() % 4]);
            s += "\n+\n";
            for (int i = 0; i < L; ++i) s.push_back((char)('#' + (rng() % 8 == 0 ? rng() % 40 : 38)));
            s.push_back('\n');
            for (size_t i = 0; i < s.size() && at < n; ++i) t[at++] = (uint8_t)s[i];
        }
        break;
    }
    case 1:  // incompressible
        for (auto& c : t) c = (uint8_t)rng();
        break;
    case 2:  // long runs and short periods: matches that run into themselves
    {
        size_t at = 0;
        while (at < n) {
            const size_t period = 1 + rng() % 7, len = 1 + rng() % 3000;
            uint8_t pat[8];
            for (auto& c : pat) c = (uint8_t)('a' + rng() % 4);
            for (size_t i = 0; i < len && at < n; ++i) t[at++] = pat[i % period];
        }
        break;
    }
    case 3:  // far matches: blocks repeated from up to 32 KiB back
    {
        size_t at = 0;
        while (at < n) {
            if (at > 1000 && rng() % 3) {
                const size_t back = 1 + rng() % (at < 32768 ? at : 32768), len = 3 + rng() % 600;
                for (size_t i = 0; i < len && at < n; ++i, ++at) t[at] = t[at - back];
            } else {
                const size_t len = 1 + rng() % 300;
                for (size_t i = 0; i < len && at < n; ++i) t[at++] = (uint8_t)rng();
            }
        }
        break;
    }
    default:  // skewed alphabet with rare symbols: long codes beyond the table root
        for (auto& c : t) {
            const unsigned r = (unsigned)(rng() % 100000);
            c = r < 60000 ? 'A' : r < 85000 ? 'C' : r < 95000 ? 'G' : r < 99000 ? 'T' : (uint8_t)(rng() % 256);
        }
    }
    return t;
}

}  // namespace

static std::vector<uint8_t> read_file(const char* path)
{
    std::vector<uint8_t> v;
    FILE* f = std::fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    std::fclose(f);
    return v;
}

int main(int argc, char** argv)
{
    // emu_inflate --check RAW_DEFLATE_FILE TEXT_FILE: streams from encoders other than zlib (tests/test_inflate.py feeds it what GNU
    // gzip wrote); the text is at most 64 KiB, as in a BGZF member
    if (argc == 4 && std::string(argv[1]) == "--check") {
        const auto packed = read_file(argv[2]), text = read_file(argv[3]);
        if (text.size() > 65536) return 2;
        const Result r = ours(packed, (uint32_t)text.size(), 5);
        const bool ok = r.status == bl_inflate::OK && r.text == text && !r.wrote_outside;
        std::printf("%s status %u\n", ok ? "same" : "DIFFERENT", r.status);
        return ok ? 0 : 1;
    }
    // emu_inflate --hist LEVEL: a FASTQ of 150-bp reads (Illumina-style names, binned qualities), 65280-byte members deflated by zlib at
    // LEVEL, decoded here; prints where the matches reach — the reason the decoder's window is LDS and not the member's own output in HBM:
    // four letters make any 8-mer recur within a few KiB, and zlib takes those matches wherever in the window it finds them
    if (argc == 3 && std::string(argv[1]) == "--hist") {
        const int level = std::atoi(argv[2]);
        std::mt19937_64 rng(1);
        std::string text;
        for (int i = 0; text.size() < 64u * 65280u; ++i) {
            char name[96];
            std::snprintf(name, sizeof name, "@A00123:45:HXXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT\n", 1000 + i % 30000, 1000 + i / 7);
            text += name;
            for (int b = 0; b < 150; ++b) text += "ACGT"[rng() & 3];
            text += "\n+\n";
            for (int b = 0; b < 150; ++b) text += (rng() % 100 < 93) ? 'F' : ":,#"[rng() % 3];
            text += "\n";
        }
        unsigned long long text_bytes = 0;
        for (size_t a = 0; a + 65280 <= text.size(); a += 65280) {
            const std::vector<uint8_t> chunk(text.begin() + a, text.begin() + a + 65280);
            const auto packed = deflate_raw(chunk, level, Z_DEFAULT_STRATEGY);
            const Result r = ours(packed, 65280, 0);
            if (r.status != bl_inflate::OK || r.text != chunk) return 1;
            text_bytes += 65280;
        }
        unsigned long long n = 0, bytes = 0;
        for (int b = 0; b < 8; ++b) { n += g_hist_n[b]; bytes += g_hist_bytes[b]; }
        std::printf("{\"zlib_level\": %d, \"text_bytes\": %llu, \"matches\": %llu, \"matched_bytes\": %llu, \"by_distance_4KiB_bins\": [", level, text_bytes, n, bytes);
        for (int b = 0; b < 8; ++b) std::printf("%s{\"upto_KiB\": %d, \"matches_frac\": %.4f, \"bytes_frac_of_text\": %.4f}", b ? ", " : "", 4 * (b + 1), (double)g_hist_n[b] / (double)n, (double)g_hist_bytes[b] / (double)text_bytes);
        std::printf("]}\n");
        return 0;
    }
    const size_t n_sound = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 400;
    const size_t n_damaged = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 4000;
    std::mt19937_64 rng(12345);
    size_t id = 0, long_code_streams = 0;
    std::vector<std::vector<uint8_t>> keep_packed;
    std::vector<uint32_t> keep_isize;
    // sound streams
    for (size_t c = 0; c < n_sound; ++c, ++id) {
        const int kind = (int)(c % 5);
        const size_t sizes[] = {0, 1, 2, 17, 255, 4096, 16383, 16384, 16385, 40000, 65280, 65536};
        const size_t n = c < 60 ? sizes[c % 12] : 1 + rng() % 65536;
        const int level = (int)(rng() % 10);
        const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        const int strategy = strategies[rng() % 5];
        const int mem_level = 1 + (int)(rng() % 9);  // small: many blocks per stream
        const auto text = make_text(rng, kind, n);
        const auto packed = deflate_raw(text, level, strategy, mem_level);
        const Result r = ours(packed, (uint32_t)n, (unsigned)(rng() % 16));
        expect(r.status == bl_inflate::OK, "sound stream refused", id);
        expect(r.text == text, "text differs", id);
        expect(!r.wrote_outside, "wrote outside the member's text", id);
        if (kind == 4) ++long_code_streams;
        if (keep_packed.size() < 64 && n > 200) {
            keep_packed.push_back(packed);
            keep_isize.push_back((uint32_t)n);
        }
    }
    // the stated size is wrong: an error, and nothing written outside
    for (size_t c = 0; c < keep_packed.size(); ++c, ++id) {
        for (int delta : {-1, +1, -100}) {
            const uint32_t isize = (uint32_t)((int)keep_isize[c] + delta);
            const Result r = ours(keep_packed[c], isize, (unsigned)(rng() % 16));
            expect(r.status != bl_inflate::OK, "wrong size accepted", id);
            expect(!r.wrote_outside, "wrote outside the member's text", id);
        }
    }
    // damaged streams: bit flips, truncations, garbage — zlib is the judge
    size_t agree_ok = 0, agree_bad = 0;
    for (size_t c = 0; c < n_damaged; ++c, ++id) {
        const size_t which = rng() % keep_packed.size();
        std::vector<uint8_t> packed = keep_packed[which];
        const uint32_t want = keep_isize[which];
        const int how = (int)(rng() % 4);
        if (how == 0) {
            const int flips = 1 + (int)(rng() % 3);
            for (int f = 0; f < flips; ++f) packed[rng() % packed.size()] ^= (uint8_t)(1u << (rng() % 8));
        } else if (how == 1) {
            packed.resize(rng() % packed.size());
        } else if (how == 2) {
            const size_t at = rng() % packed.size(), len = 1 + rng() % 16;
            for (size_t i = at; i < at + len && i < packed.size(); ++i) packed[i] = (uint8_t)rng();
        } else {
            for (auto& b : packed) b = (uint8_t)rng();  // garbage from the first bit
            packed.resize(1 + rng() % packed.size());
        }
        std::vector<uint8_t> ztext;
        const bool zok = zlib_inflate(packed, want, ztext);
        const Result r = ours(packed, want, (unsigned)(rng() % 16));
        expect(!r.wrote_outside, "wrote outside the member's text", id);
        if (zok) {
            expect(r.status == bl_inflate::OK && r.text == ztext, "zlib accepts, we differ", id);
            ++agree_ok;
        } else {
            expect(r.status != bl_inflate::OK, "zlib refuses, we accept", id);
            ++agree_bad;
        }
    }
    std::printf("emu_inflate: %zu sound streams (%zu with a skewed alphabet), %zu damaged (%zu still sound, %zu refused by both), %d failures\n", n_sound,
                long_code_streams, n_damaged, agree_ok, agree_bad, failures);
    if (failures == 0) std::printf("emu_inflate: OK\n");
    return failures ? 1 : 0;
}
