// TEST INFRASTRUCTURE: the pieces of the many-threaded gzip decoder (biolib_amd/csrc/bl_pgzip.hpp) on one thread, against zlib,
// built with -fsanitize=address,undefined (tests/test_pgzip.py runs it on good and on damaged files).
//   pgzip_check file.gz part_bytes
// zlib inflates the whole file and notes every bit position where a block begins, with the text offset there.  Then, part by
// part as the reader does: find a block, decode it to symbols without the window, and — when the find is a true block start —
// turn the symbols into text with the 32 KiB zlib wrote before it: that must be zlib's text, and the part must end on a true
// block start too.  A find that is no block start is allowed (the reader notices: the part before never arrives there).
// Prints one line: "ok parts=<n> found=<n> true=<n> text=<bytes checked>" or "damaged ..." (zlib refused the file: the decoder only
// has to survive it), exit code 1 on a mismatch.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../biolib_amd/csrc/bl_crc32.hpp"
#include "../../biolib_amd/csrc/bl_pgzip.hpp"

// pgzip_check --crc: bl_crc32 against zlib's crc32 on every short length, start offset and two start values, and on long buffers
static int check_crc()
{
    std::vector<uint8_t> d((1u << 22) + 64);
    uint64_t s = 88172645463325252ull;
    for (auto& b : d) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        b = (uint8_t)s;
    }
    for (size_t n = 0; n < 600; ++n)
        for (size_t off = 0; off < 17; ++off)
            for (uint32_t c : {0u, 0xdeadbeefu})
                if ((uint32_t)crc32(c, d.data() + off, (uInt)n) != bl_crc32(c, d.data() + off, n)) { std::printf("bad: crc of %zu bytes at offset %zu\n", n, off); return 1; }
    for (size_t n : {1000ul, 4096ul, 65536ul, 1000003ul, (1ul << 22)})
        for (size_t off : {0ul, 1ul, 15ul})
            if ((uint32_t)crc32(7, d.data() + off, (uInt)n) != bl_crc32(7, d.data() + off, n)) { std::printf("bad: crc of %zu bytes\n", n); return 1; }
#if defined(__x86_64__)
    std::printf("ok crc clmul=%d\n", (int)blcrc::have_clmul());
#else
    std::printf("ok crc clmul=0\n");
#endif
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 2 && std::string(argv[1]) == "--crc") return check_crc();
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    uint8_t* data = static_cast<uint8_t*>(std::malloc(size > 0 ? (size_t)size : 1));  // exactly the file: a read beyond it is caught
    if (size > 0 && std::fread(data, 1, (size_t)size, f) != (size_t)size) return 2;
    std::fclose(f);
    const uint64_t part_bytes = std::strtoull(argv[2], nullptr, 10);

    // zlib's view: text and block starts
    std::vector<uint8_t> text;
    std::map<uint64_t, uint64_t> starts;  // bit -> text offset
    bool damaged = false;
    {
        z_stream z{};
        if (inflateInit2(&z, 15 + 16) != Z_OK) return 2;
        std::vector<uint8_t> out(1 << 16);
        uint64_t in_at = 0;
        z.next_in = data;
        z.avail_in = (uInt)size;
        bool open_member = false;
        while (in_at < (uint64_t)size || open_member) {
            z.next_out = out.data();
            z.avail_out = (uInt)out.size();
            const uint8_t* before = z.next_in;
            const int rc = inflate(&z, Z_BLOCK);
            in_at += (uint64_t)(z.next_in - before);
            text.insert(text.end(), out.data(), out.data() + (out.size() - z.avail_out));
            open_member = true;
            if (rc == Z_STREAM_END) {
                open_member = false;
                if (in_at == (uint64_t)size) break;
                if (inflateReset(&z) != Z_OK) { damaged = true; break; }
                continue;
            }
            if (rc != Z_OK) { damaged = true; break; }
            if ((z.data_type & 128) && !(z.data_type & 64)) starts[8 * in_at - (uint64_t)(z.data_type & 7)] = text.size();
            if (z.avail_in == 0 && z.avail_out != 0) { damaged = true; break; }  // ends inside a member
        }
        inflateEnd(&z);
    }

    blpg::SymbolDecoder decoder;
    blpg::Part part;
    static blpg::SymbolTable table;
    const uint64_t n_parts = ((uint64_t)size + part_bytes - 1) / part_bytes;
    uint64_t found = 0, truly = 0, checked = 0;
    std::vector<uint8_t> got;
    for (uint64_t i = 0; i < n_parts; ++i) {
        const uint64_t limit = 8 * (i + 1) * part_bytes;
        uint64_t at;
        if (i == 0) {
            const uint64_t d = blpg::skip_member_header(data, (uint64_t)size, 0);
            at = d == blpg::NPOS ? blpg::NPOS : 8 * d;
        } else {
            at = blpg::find_block(data, (uint64_t)size, 8 * i * part_bytes, limit);
        }
        for (int tries = 0; at != blpg::NPOS && tries < 16; ++tries) {
            if (decoder.run(data, (uint64_t)size, at, limit, 48ull << 20, part)) break;
            at = i == 0 ? blpg::NPOS : blpg::find_block(data, (uint64_t)size, at + 1, limit);
        }
        if (at == blpg::NPOS || part.start_bit != at) continue;
        ++found;
        if (damaged) continue;
        auto it = starts.find(at);
        if (it == starts.end()) continue;  // a false find
        ++truly;
        const uint64_t off = it->second;
        uint8_t window[blpg::WINDOW] = {0};
        const uint32_t known = off < blpg::WINDOW ? (uint32_t)off : blpg::WINDOW;
        std::memcpy(window + (blpg::WINDOW - known), text.data() + (off - known), known);
        got.resize(part.n);
        if (off + part.n > text.size()) { std::printf("bad: part %llu runs past zlib's text\n", (unsigned long long)i); return 1; }
        if (!blpg::markers_known(part.sym.p, part.n, known)) { std::printf("bad: part %llu points before the text\n", (unsigned long long)i); return 1; }
        table.set(window);
        blpg::resolve(part.sym.p, part.n, table, got.data());
        if (std::memcmp(got.data(), text.data() + off, part.n) != 0) { std::printf("bad: part %llu differs from zlib's text\n", (unsigned long long)i); return 1; }
        if (!part.at_eof && !starts.count(part.end_bit)) { std::printf("bad: part %llu ends where no block begins\n", (unsigned long long)i); return 1; }
        if (part.at_eof && off + part.n != text.size()) { std::printf("bad: part %llu claims the end of the stream too early\n", (unsigned long long)i); return 1; }
        checked += part.n;
    }
    std::printf("%s parts=%llu found=%llu true=%llu text=%llu\n", damaged ? "damaged" : "ok", (unsigned long long)n_parts, (unsigned long long)found, (unsigned long long)truly,
                (unsigned long long)checked);
    std::free(data);
    return 0;
}
