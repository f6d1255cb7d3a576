// bl_parse.hip — device-side FASTA / FASTQ text parser: the raw file bytes are copied to HBM once and turned into
// a batch THERE (newline index -> line classification -> prefix sums -> gather of the sequence lines), so ingest
// runs at PCIe speed instead of at the speed of one host thread (the host reader of
// bl_ingest.cpp stays the general path and the semantic reference, itself pinned against the reference's kseq).
// Accepted on this path: FASTQ with exactly four lines per record, FASTA with any line wrapping; LF or CRLF.
// Anything irregular is refused with BL_ERR_INVALID (never silently mis-parsed): use bl_reader_* for those files.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <string>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);
extern int bl_ctx_device(bl_ctx* ctx);
extern void* bl_ctx_scratch(bl_ctx* ctx, int slot, size_t bytes);  // bl_capi.hip: grow-only device scratch that lives with the context
extern void* bl_ctx_pool_alloc(bl_ctx* ctx, size_t bytes);          // bl_capi.hip: batch buffers, recycled between batches
extern void bl_ctx_pool_free(bl_ctx* ctx, void* p);
extern int bl_batch_adopt_device(bl_ctx* ctx, void* d_bases, uint64_t n_bases, uint64_t* d_offsets, uint64_t n_seqs, uint64_t fixed_len,
                                 bl_batch** out);  // bl_capi.hip

namespace {

constexpr int PB = 256;          // threads per block
constexpr int BYTES_PER_BLOCK = PB * 16;

__device__ __forceinline__ uint32_t newline_mask16(const uint8_t* text, uint64_t n, uint64_t at)
{
    uint32_t m = 0;
    if (at + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + at);  // text is 16-byte aligned, at is a multiple of 16
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = d[i] ^ 0x0a0a0a0au;                                        // zero byte <=> '\n'
            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;  // bit 7 of every zero byte
            m |= (((z >> 7) * 0x00204081u >> 21) & 0xfu) << (4 * i);
        }
    } else {
        for (int b = 0; b < 16 && at + b < n; ++b)
            if (text[at + b] == '\n') m |= 1u << b;
    }
    return m;
}

__global__ __launch_bounds__(PB) void count_newlines_kernel(const uint8_t* text, uint64_t n, unsigned long long* block_count)
{
    __shared__ unsigned int wsum[PB / 64];
    const uint64_t at = ((uint64_t)blockIdx.x * PB + threadIdx.x) * 16;
    unsigned int c = at < n ? __builtin_popcount(newline_mask16(text, n, at)) : 0;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// line_end[i] = byte offset of the i-th '\n'
__global__ __launch_bounds__(PB) void newline_positions_kernel(const uint8_t* text, uint64_t n, const unsigned long long* block_base,
                                                               unsigned long long* line_end)
{
    __shared__ unsigned int wsum[PB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t at = ((uint64_t)blockIdx.x * PB + threadIdx.x) * 16;
    uint32_t m = at < n ? newline_mask16(text, n, at) : 0;
    const unsigned int c = __builtin_popcount(m);
    unsigned int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned int before = 0;
    for (int i = 0; i < wv; ++i) before += wsum[i];
    unsigned long long idx = block_base[blockIdx.x] + before + incl - c;
    while (m) {
        const int b = __builtin_ctz(m);
        m &= m - 1;
        line_end[idx++] = at + b;
    }
}

enum { KIND_OTHER = 0, KIND_SEQ = 1, KIND_HEADER = 2 };
enum { ERR_FASTQ_HEADER = 1, ERR_FASTQ_PLUS = 2, ERR_FASTQ_QUAL = 4, ERR_FASTA_SEQLINE = 8 };

__device__ __forceinline__ void line_span(const uint8_t* text, const unsigned long long* line_end, uint64_t li, uint64_t& start, uint64_t& end)
{
    start = li ? line_end[li - 1] + 1 : 0;
    end = line_end[li];
    if (end > start && text[end - 1] == '\r') --end;
}

// one thread per line: kind, sequence length (0 unless a sequence line), header flag; format checks
__global__ void classify_lines_kernel(const uint8_t* text, const unsigned long long* line_end, uint64_t n_lines, int fastq,
                                      unsigned long long* seq_len, unsigned long long* hdr, unsigned int* err)
{
    const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_lines) return;
    uint64_t s, e;
    line_span(text, line_end, li, s, e);
    const uint8_t first = e > s ? text[s] : 0;
    unsigned long long len = 0, h = 0;
    if (fastq) {
        const int f = (int)(li & 3);
        if (f == 0) {
            h = 1;
            if (first != '@') atomicOr(err, (unsigned)ERR_FASTQ_HEADER);
        } else if (f == 1) {
            len = e - s;
        } else if (f == 2) {
            if (first != '+') atomicOr(err, (unsigned)ERR_FASTQ_PLUS);
        } else {
            uint64_t s2, e2;
            line_span(text, line_end, li - 2, s2, e2);
            if (e - s != e2 - s2) atomicOr(err, (unsigned)ERR_FASTQ_QUAL);
        }
    } else {
        if (first == '>') h = 1;
        else {
            len = e - s;  // masked later for lines in front of the first header
            if (first == '@' || first == '+') atomicOr(err, (unsigned)ERR_FASTA_SEQLINE);  // would end the record in the reference reader
        }
    }
    seq_len[li] = len;
    hdr[li] = h;
}

// FASTA: lines in front of the first header belong to no record
__global__ void mask_leading_lines_kernel(const unsigned long long* rec_incl, unsigned long long* seq_len, uint64_t n_lines)
{
    const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li < n_lines && rec_incl[li] == 0) seq_len[li] = 0;
}

// offsets[r] = first base of record r (r = rec_incl - 1 at its header line); offsets[n_records] = total
__global__ void record_offsets_kernel(const unsigned long long* hdr, const unsigned long long* rec_incl, const unsigned long long* dst,
                                      uint64_t n_lines, unsigned long long* offsets, uint64_t n_records, uint64_t total)
{
    const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li == 0) offsets[n_records] = total;
    if (li < n_lines && hdr[li]) offsets[rec_incl[li] - 1] = dst[li];
}

// do all records hold exactly `len` bases?  (flag |= 1 where one does not)
__global__ void uniform_length_kernel(const unsigned long long* offsets, uint64_t n_records, uint64_t len, unsigned int* flag)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_records && offsets[r] != r * len) atomicOr(flag, 1u);
}

// one thread per 16 output bytes: find the sequence line that holds output byte x (last line with dst <= x), gather
__global__ void gather_bases_kernel(const uint8_t* text, const unsigned long long* line_end, const unsigned long long* dst,
                                    const unsigned long long* seq_len, uint64_t n_lines, uint8_t* bases, uint64_t total)
{
    const uint64_t x0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (x0 >= total) return;
    uint64_t lo = 0, hi = n_lines;  // first line with dst > x0
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (dst[mid] <= x0) lo = mid + 1;
        else hi = mid;
    }
    uint64_t li = lo - 1;  // dst[0] = 0 <= x0, so lo >= 1; this line has seq_len > 0
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t x = x0;
    int filled = 0;
    while (filled < 16 && x < total) {
        while (li < n_lines && seq_len[li] == 0) ++li;  // header / quality lines in between
        if (li >= n_lines) break;                        // cannot happen while x < total; keeps a logic error from running away
        const uint64_t ls = li ? line_end[li - 1] + 1 : 0;
        const uint64_t in_line = x - dst[li];
        uint64_t take = seq_len[li] - in_line;
        if (take > (uint64_t)(16 - filled)) take = 16 - filled;
        const uint8_t* src = text + ls + in_line;
        for (uint64_t b = 0; b < take; ++b, ++filled) w[filled >> 2] |= (uint32_t)src[b] << (8 * (filled & 3));
        x += take;
        if (in_line + take == seq_len[li]) ++li;
    }
    *reinterpret_cast<uint4*>(bases + x0) = make_uint4(w[0], w[1], w[2], w[3]);  // bases has >= 16 bytes of slack
}

#define P_HIP(call)                                                                                                       \
    do {                                                                                                                  \
        hipError_t e_ = (call);                                                                                           \
        if (e_ != hipSuccess) { cleanup(); return bl_set_error(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e_)); } \
    } while (0)

}  // namespace

// The parser proper, over text that is in device memory already.  `ends` is the host's copy of the text's last `ends_n` bytes
// (all of it when the text is shorter than 64): the few decisions taken on the host look at the first byte and at the end only.
int bl_parse_device_text(bl_ctx* ctx, const uint8_t* d_text_in, uint64_t n_bytes, char first_byte, const char* ends, uint64_t ends_n, bl_batch** out,
                         uint64_t* n_seqs, uint64_t* n_bases);

extern "C" int bl_batch_from_text(bl_ctx* ctx, const char* text, uint64_t n_bytes, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !out || (n_bytes && !text)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_seqs) *n_seqs = 0;
    if (n_bases) *n_bases = 0;
    if (n_bytes == 0) return bl_batch_upload(ctx, "", 0, nullptr, 0, out);
    if (hipSetDevice(bl_ctx_device(ctx)) != hipSuccess) return bl_set_error(BL_ERR_HIP, "hipSetDevice failed");
    uint8_t* d_text = static_cast<uint8_t*>(bl_ctx_scratch(ctx, 3, n_bytes + 64));
    if (!d_text) return bl_set_error(BL_ERR_OOM, "device allocation failed (text)");
    const hipError_t e = hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, bl_ctx_stream(ctx));
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    const uint64_t ends_n = n_bytes < 64 ? n_bytes : 64;
    const int rc = bl_parse_device_text(ctx, d_text, n_bytes, text[0], text + (n_bytes - ends_n), ends_n, out, n_seqs, n_bases);
    if (rc != BL_OK) (void)hipStreamSynchronize(bl_ctx_stream(ctx));  // the copy out of the caller's buffer is over when we return
    return rc;
}

int bl_parse_device_text(bl_ctx* ctx, const uint8_t* d_text_in, uint64_t n_bytes, char first_byte, const char* ends, uint64_t ends_n, bl_batch** out,
                         uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !out || !d_text_in || !ends || ends_n == 0 || ends_n > n_bytes) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_seqs) *n_seqs = 0;
    if (n_bases) *n_bases = 0;
    auto last = [&](uint64_t back) { return ends[ends_n - 1 - back]; };  // the text's byte `back` places in front of its last one
    // the format is decided by the first character
    const bool fastq = first_byte == '@';
    if (!fastq && first_byte != '>') return bl_set_error(BL_ERR_INVALID, "text starts with neither '>' nor '@': use bl_reader_* for irregular files");
    // a '>' that is the very last byte of the file and alone on its line opens no record in the reference reader (kseq meets
    // end of file while looking for the name and reports end of input): drop it
    if (!fastq && last(0) == '>' && (n_bytes == 1 || last(1) == '\n')) {
        --n_bytes;
        --ends_n;
        if (n_bytes == 0) return bl_batch_upload(ctx, "", 0, nullptr, 0, out);
    }
    const bool open_last_line = last(0) != '\n';  // the last line has no terminator: a virtual one is added

    hipStream_t s = bl_ctx_stream(ctx);
    // Temporary arrays come from the context's grow-only scratch (slot 0: block counts, slot 1: per-line arrays, slot 2: scan
    // workspace; slot 3 holds the text when it came from the host): a file is parsed span after span, and a dozen hipMalloc /
    // hipFree per span cost more than the kernels.  Only the two arrays the batch keeps are allocated here.
    const uint8_t* d_text = nullptr;
    unsigned long long *d_blk = nullptr, *d_line_end = nullptr, *d_len = nullptr, *d_hdr = nullptr, *d_rec = nullptr, *d_dst = nullptr, *d_offsets = nullptr;
    unsigned int* d_err = nullptr;
    void* d_tmp = nullptr;
    uint8_t* d_bases = nullptr;
    bool handed_over = false;  // d_bases / d_offsets now belong to the batch
    auto cleanup = [&]() {     // every exit path: nothing may still be running on the scratch; the outputs go unless adopted
        (void)hipStreamSynchronize(s);
        if (!handed_over) {
            bl_ctx_pool_free(ctx, d_bases);
            bl_ctx_pool_free(ctx, d_offsets);
            d_bases = nullptr;
            d_offsets = nullptr;
        }
    };
    auto fail_free = [&](int code, const char* msg) {
        cleanup();
        return bl_set_error(code, msg);
    };
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };

    P_HIP(hipSetDevice(bl_ctx_device(ctx)));
    const uint64_t n = n_bytes;
    const unsigned n_blocks = (unsigned)((n + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK);
    const size_t blk_bytes = up256(2 * ((size_t)n_blocks + 1) * sizeof(unsigned long long));
    unsigned char* a0 = static_cast<unsigned char*>(bl_ctx_scratch(ctx, 0, blk_bytes + 256));
    if (!a0) return fail_free(BL_ERR_OOM, "device allocation failed (block scratch)");
    d_text = d_text_in;
    d_blk = reinterpret_cast<unsigned long long*>(a0);  // counts, then their exclusive prefix
    d_err = reinterpret_cast<unsigned int*>(a0 + blk_bytes);
    unsigned long long* d_blk_base = d_blk + n_blocks + 1;
    P_HIP(hipMemsetAsync(d_blk + n_blocks, 0, sizeof(unsigned long long), s));
    P_HIP(hipMemsetAsync(d_err, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(count_newlines_kernel, dim3(n_blocks), dim3(PB), 0, s, d_text, n, d_blk);

    // scans: one scratch buffer sized for the largest of them (n_lines <= n)
    size_t tmp_bytes = 0, need = 0;
    P_HIP(rocprim::exclusive_scan(nullptr, need, d_blk, d_blk_base, 0ull, (size_t)n_blocks + 1, rocprim::plus<unsigned long long>(), s));
    tmp_bytes = need;
    // the other two scans run over n_lines <= n_blocks * BYTES_PER_BLOCK items; their workspace is asked for again below
    d_tmp = bl_ctx_scratch(ctx, 2, tmp_bytes ? tmp_bytes : 16);
    if (!d_tmp) return fail_free(BL_ERR_OOM, "device allocation failed (scan scratch)");
    P_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_blk, d_blk_base, 0ull, (size_t)n_blocks + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long n_newlines = 0;
    P_HIP(hipMemcpyAsync(&n_newlines, d_blk_base + n_blocks, sizeof(n_newlines), hipMemcpyDeviceToHost, s));
    P_HIP(hipStreamSynchronize(s));
    const uint64_t n_lines_raw = n_newlines + (open_last_line ? 1 : 0);
    uint64_t n_lines = n_lines_raw;
    if (fastq && (n_lines & 3)) {  // blank lines after the last record are tolerated, anything else is not 4-line FASTQ
        uint64_t excess = n_lines & 3, blank = 0, pos = ends_n;  // positions inside `ends`; running out of it means "not blank"
        while (blank < excess && pos > 0) {  // walk back over empty lines ("\n" or "\r\n")
            if (open_last_line && blank == 0) break;  // the last line is not empty
            if (ends[pos - 1] != '\n') break;
            uint64_t q = pos - 1;
            if (q > 0 && ends[q - 1] == '\r') --q;
            if (q == 0 && ends_n < n_bytes) break;     // cannot see the byte in front
            if (q > 0 && ends[q - 1] != '\n') break;  // the line ending here has content
            ++blank;
            pos = q;
        }
        if (blank < excess) return fail_free(BL_ERR_INVALID, "FASTQ text is not made of 4-line records: use bl_reader_*");
        n_lines -= excess;
    }

    {
        const size_t per = up256((n_lines_raw + 2) * sizeof(unsigned long long));
        unsigned char* a1 = static_cast<unsigned char*>(bl_ctx_scratch(ctx, 1, 5 * per));
        if (!a1) return fail_free(BL_ERR_OOM, "device allocation failed (line scratch)");
        d_line_end = reinterpret_cast<unsigned long long*>(a1);
        d_len = reinterpret_cast<unsigned long long*>(a1 + per);
        d_hdr = reinterpret_cast<unsigned long long*>(a1 + 2 * per);
        d_rec = reinterpret_cast<unsigned long long*>(a1 + 3 * per);
        d_dst = reinterpret_cast<unsigned long long*>(a1 + 4 * per);
    }
    hipLaunchKernelGGL(newline_positions_kernel, dim3(n_blocks), dim3(PB), 0, s, d_text, n, d_blk_base, d_line_end);
    if (open_last_line) P_HIP(hipMemcpyAsync(d_line_end + n_newlines, &n, sizeof(unsigned long long), hipMemcpyHostToDevice, s));  // virtual '\n'
    P_HIP(hipMemsetAsync(d_len + n_lines, 0, sizeof(unsigned long long), s));
    const unsigned lb = (unsigned)((n_lines + 255) / 256);
    hipLaunchKernelGGL(classify_lines_kernel, dim3(lb), dim3(256), 0, s, d_text, d_line_end, n_lines, fastq ? 1 : 0, d_len, d_hdr, d_err);

    P_HIP(rocprim::inclusive_scan(nullptr, need, d_hdr, d_rec, (size_t)n_lines, rocprim::plus<unsigned long long>(), s));
    if (need > tmp_bytes) {
        P_HIP(hipStreamSynchronize(s));
        tmp_bytes = need;
        d_tmp = bl_ctx_scratch(ctx, 2, tmp_bytes);
        if (!d_tmp) return fail_free(BL_ERR_OOM, "device allocation failed (scan scratch)");
    }
    P_HIP(rocprim::inclusive_scan(d_tmp, need, d_hdr, d_rec, (size_t)n_lines, rocprim::plus<unsigned long long>(), s));
    if (!fastq) hipLaunchKernelGGL(mask_leading_lines_kernel, dim3(lb), dim3(256), 0, s, d_rec, d_len, n_lines);
    P_HIP(rocprim::exclusive_scan(nullptr, need, d_len, d_dst, 0ull, (size_t)n_lines + 1, rocprim::plus<unsigned long long>(), s));
    if (need > tmp_bytes) {
        P_HIP(hipStreamSynchronize(s));
        tmp_bytes = need;
        d_tmp = bl_ctx_scratch(ctx, 2, tmp_bytes);
        if (!d_tmp) return fail_free(BL_ERR_OOM, "device allocation failed (scan scratch)");
    }
    P_HIP(rocprim::exclusive_scan(d_tmp, need, d_len, d_dst, 0ull, (size_t)n_lines + 1, rocprim::plus<unsigned long long>(), s));

    unsigned long long total = 0, n_records = 0;
    unsigned int err = 0;
    P_HIP(hipMemcpyAsync(&total, d_dst + n_lines, sizeof(total), hipMemcpyDeviceToHost, s));
    P_HIP(hipMemcpyAsync(&n_records, d_rec + (n_lines - 1), sizeof(n_records), hipMemcpyDeviceToHost, s));
    P_HIP(hipMemcpyAsync(&err, d_err, sizeof(err), hipMemcpyDeviceToHost, s));
    P_HIP(hipStreamSynchronize(s));
    if (err) {
        const char* msg = (err & ERR_FASTQ_HEADER) ? "FASTQ record does not start with '@' every 4 lines: use bl_reader_*"
                          : (err & ERR_FASTQ_PLUS) ? "FASTQ separator line does not start with '+': use bl_reader_*"
                          : (err & ERR_FASTQ_QUAL) ? "FASTQ quality length differs from the sequence length"
                                                   : "FASTA sequence line starts with '@' or '+': use bl_reader_*";
        return fail_free(BL_ERR_INVALID, msg);
    }

    d_bases = static_cast<uint8_t*>(bl_ctx_pool_alloc(ctx, total + 64));
    d_offsets = static_cast<unsigned long long*>(bl_ctx_pool_alloc(ctx, (n_records + 1) * sizeof(unsigned long long)));
    if (!d_bases || !d_offsets) return fail_free(BL_ERR_OOM, "device allocation failed (batch)");
    P_HIP(hipMemsetAsync(d_bases + (total & ~15ull), 0, 64 + (total & 15ull), s));
    if (total) {
        const uint64_t threads = (total + 15) / 16;
        hipLaunchKernelGGL(gather_bases_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_text, d_line_end, d_dst, d_len, n_lines,
                           d_bases, (uint64_t)total);
    }
    hipLaunchKernelGGL(record_offsets_kernel, dim3(lb), dim3(256), 0, s, d_hdr, d_rec, d_dst, n_lines, d_offsets, (uint64_t)n_records, (uint64_t)total);
    // reads of one length (the usual short-read file) make a fixed-length batch: the read-tiled scan kernels, no start-bit vector
    uint64_t fixed_len = n_records > 1 && total % n_records == 0 ? total / n_records : 0;
    unsigned int ragged = 0;
    if (fixed_len) {
        P_HIP(hipMemsetAsync(d_err + 1, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(uniform_length_kernel, dim3((unsigned)((n_records + 256) / 256)), dim3(256), 0, s, d_offsets, (uint64_t)n_records, fixed_len, d_err + 1);
        P_HIP(hipMemcpyAsync(&ragged, d_err + 1, sizeof(ragged), hipMemcpyDeviceToHost, s));
    }
    P_HIP(hipGetLastError());
    P_HIP(hipStreamSynchronize(s));
    if (ragged) fixed_len = 0;
    handed_over = true;
    cleanup();
    int rc = bl_batch_adopt_device(ctx, d_bases, total, reinterpret_cast<uint64_t*>(d_offsets), n_records, fixed_len, out);  // takes ownership of both
    if (rc != BL_OK) return rc;
    if (n_seqs) *n_seqs = n_records;
    if (n_bases) *n_bases = total;
    return BL_OK;
}
