// bl_inflate.hip — BGZF members inflated on the device, one wave per member (decoder: bl_inflate_core.hpp), and their CRC-32
// checked there too, so that a bgzip'ed FASTA / FASTQ crosses PCIe compressed and the host's part of ingest is reading the
// file and walking the member headers (the host inflate pool of bl_ingest.cpp stays for the record
// calls and for plain gzip, whose single deflate stream has no entry points).
//
// Formats: RFC 1951 (deflate), RFC 1952 (gzip member: 10-byte header, extra field, deflate data, CRC-32, ISIZE), SAM
// specification §4.1 (BGZF: the extra field carries 'B','C', 2, BSIZE = member size - 1; members hold at most 64 KiB of text).
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/biolib_amd.h"
#include "bl_inflate_core.hpp"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);
extern int bl_ctx_device(bl_ctx* ctx);
int bl_bgzf_inflate_on(hipStream_t s, const void* d_packed, uint64_t packed_bytes, const bl_bgzf_member* d_members, uint64_t n_members, void* d_text,
                       uint64_t text_bytes, uint32_t* d_status);

namespace {

constexpr uint32_t CRC_POLY = 0xedb88320u;  // CRC-32 of gzip, bit-reflected: bit 31 of a word is x^0
constexpr uint32_t STATUS_CRC = 10;         // after the decoder's own codes (bl_inflate::Status)

// a(x) * b(x) mod P(x) in the reflected representation
__host__ __device__ constexpr uint32_t mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}

struct PowerTable {
    uint32_t x2n[32];  // x^(2^k) mod P
};
constexpr PowerTable make_powers()
{
    PowerTable t{};
    uint32_t p = 0x40000000u;  // x^1
    for (int k = 0; k < 32; ++k) {
        t.x2n[k] = p;
        p = mulmod(p, p);
    }
    return t;
}
__constant__ PowerTable g_powers = make_powers();

__global__ __launch_bounds__(64) void inflate_members_kernel(const uint8_t* packed, uint64_t packed_bytes, const bl_bgzf_member* members, uint32_t n_members,
                                                             uint8_t* text, uint64_t text_bytes, uint32_t* status)
{
    __shared__ bl_inflate::Shared sh;
    const uint32_t mi = blockIdx.x;
    if (mi >= n_members) return;
    const bl_bgzf_member m = members[mi];
    uint32_t st;
    // the host built the table from the headers it walked; a table that points outside the buffers is refused, not followed
    if (m.src_off > packed_bytes || m.src_len > packed_bytes - m.src_off || m.dst_off > text_bytes || m.isize > text_bytes - m.dst_off ||
        m.isize > 65536u || m.src_len == 0) {
        st = bl_inflate::ERR_INPUT;
    } else {
        // (the allocation behind `packed` is whole dwords: the last data byte's dword may be loaded)
        bl_inflate::Input in(packed + m.src_off, m.src_len, packed + ((packed_bytes + 3) & ~(uint64_t)3));
        st = bl_inflate::inflate_member(sh, in, m.src_len, text + m.dst_off, m.isize);
    }
    if ((threadIdx.x & 63u) == 0) status[mi] = st;
}

// CRC-32 of each member's text: the lanes take 64 consecutive slices, each its slice's own CRC; the CRC of a concatenation is
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B)  (mod P), so lane i's value is carried over the bytes behind its slice and the 64
// results are added (xor).
__global__ __launch_bounds__(64) void crc_members_kernel(const uint8_t* text, const bl_bgzf_member* members, uint32_t n_members, uint32_t* status)
{
    __shared__ uint32_t table[256];
    const uint32_t lane = threadIdx.x & 63u, mi = blockIdx.x;
    for (uint32_t e = lane; e < 256; e += 64) {
        uint32_t c = e;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        table[e] = c;
    }
    __syncthreads();
    if (mi >= n_members || status[mi] != bl_inflate::OK) return;
    const bl_bgzf_member m = members[mi];
    const uint32_t slice = (m.isize + 63u) / 64u;
    const uint32_t a = lane * slice < m.isize ? lane * slice : m.isize;
    const uint32_t b = a + slice < m.isize ? a + slice : m.isize;
    const uint8_t* p = text + m.dst_off;
    uint32_t c = 0xffffffffu;
    for (uint32_t i = a; i < b; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    c = a < b ? ~c : 0u;
    uint32_t power = 0x80000000u;  // x^0, then x^(8 * bytes behind the slice)
    uint32_t behind = m.isize - b;
    for (int k = 3; behind; ++k, behind >>= 1)
        if (behind & 1u) power = mulmod(g_powers.x2n[k & 31], power);
    uint32_t v = mulmod(power, c);
    for (int d = 32; d >= 1; d >>= 1) v ^= (uint32_t)__shfl_xor((int)v, d, 64);
    if (lane == 0 && v != m.crc) status[mi] = STATUS_CRC;
}

inline uint32_t le16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const unsigned char* p) { return le16(p) | (le16(p + 2) << 16); }

}  // namespace

extern "C" {

int bl_bgzf_walk(const void* bytes, uint64_t n_bytes, uint64_t src_base, uint64_t dst_base, bl_bgzf_member* members, uint64_t capacity, uint64_t* n_members,
                 uint64_t* consumed, uint64_t* text_bytes)
{
    if ((!bytes && n_bytes) || (!members && capacity) || !n_members || !consumed || !text_bytes) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    const unsigned char* p = static_cast<const unsigned char*>(bytes);
    uint64_t at = 0, n = 0, text = 0;
    while (n < capacity && n_bytes - at >= 18) {
        const unsigned char* h = p + at;
        // gzip header with the FEXTRA flag, whose extra field opens with the BGZF subfield 'B' 'C' (length 2)
        const uint32_t xlen = le16(h + 10);
        const bool good = h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && le16(h + 14) == 2 && xlen >= 6;
        if (!good) return bl_set_error(BL_ERR_INVALID, "not a BGZF member header");
        const uint64_t bsize = (uint64_t)le16(h + 16) + 1;
        if (bsize < 12 + (uint64_t)xlen + 8) return bl_set_error(BL_ERR_INVALID, "BGZF member shorter than its own header and trailer");
        if (n_bytes - at < bsize) break;  // the rest of this member is not here yet
        bl_bgzf_member& m = members[n++];
        m.src_off = src_base + at + 12 + xlen;
        m.src_len = (uint32_t)(bsize - 12 - xlen - 8);
        m.crc = le32(h + bsize - 8);
        m.isize = le32(h + bsize - 4);
        m.dst_off = dst_base + text;
        m.reserved = 0;
        if (m.isize > 65536u) return bl_set_error(BL_ERR_INVALID, "BGZF member claims more than 64 KiB of text");
        text += m.isize;
        at += bsize;
    }
    *n_members = n;
    *consumed = at;
    *text_bytes = text;
    return BL_OK;
}

int bl_bgzf_inflate(bl_ctx* ctx, const void* d_packed, uint64_t packed_bytes, const bl_bgzf_member* d_members, uint64_t n_members, void* d_text,
                    uint64_t text_bytes, uint32_t* d_status)
{
    if (!ctx) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (hipSetDevice(bl_ctx_device(ctx)) != hipSuccess) return bl_set_error(BL_ERR_HIP, "hipSetDevice failed");
    return bl_bgzf_inflate_on(bl_ctx_stream(ctx), d_packed, packed_bytes, d_members, n_members, d_text, text_bytes, d_status);
}

}  // extern "C"

// the same on a stream of the caller's (the reader inflates the next span on its own stream while the current one is parsed)
int bl_bgzf_inflate_on(hipStream_t s, const void* d_packed, uint64_t packed_bytes, const bl_bgzf_member* d_members, uint64_t n_members, void* d_text,
                       uint64_t text_bytes, uint32_t* d_status)
{
    if (!d_packed || !d_members || !d_status || (!d_text && text_bytes)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (reinterpret_cast<uintptr_t>(d_packed) & 3u) return bl_set_error(BL_ERR_INVALID, "d_packed must be 4-byte aligned");
    if (n_members == 0) return BL_OK;
    if (n_members > 0x7fffffffull) return bl_set_error(BL_ERR_INVALID, "too many members in one call");
    hipLaunchKernelGGL(inflate_members_kernel, dim3((unsigned)n_members), dim3(64), 0, s, static_cast<const uint8_t*>(d_packed), packed_bytes, d_members,
                       (uint32_t)n_members, static_cast<uint8_t*>(d_text), text_bytes, d_status);
    hipLaunchKernelGGL(crc_members_kernel, dim3((unsigned)n_members), dim3(64), 0, s, static_cast<const uint8_t*>(d_text), d_members, (uint32_t)n_members,
                       d_status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    return BL_OK;
}
