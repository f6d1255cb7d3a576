// Packed super-k-mer records for the multi-GPU bucket exchange.
//
// A super-k-mer (reference record: include/super_kmer_view.hpp:20-24 — minimizer, mm_pos, size) stands for `size`
// consecutive k-mers = size + k - 1 bases.  What travels between GPUs is its sequence, 2 bits per base, in a fixed
// 16-byte record, routed by the hash of its minimizer — every occurrence of a canonical k-mer has the same minimizer
// value, hence the same owner, so the owner can count k-mers exactly with no further exchange:
//
//   rec[0]  bases 0..31, first base in the most significant pair (kmer_view.hpp:194 packing)
//   rec[1]  bases 32..58 in bits 63..10 (same order), bits 9..5 = mm_pos (offset of the minimizer in the first k-mer,
//           super_kmer_view.hpp:132), bits 4..0 = size - 1 (number of k-mers, 1 .. k-m+1)
//
// so size + k - 1 <= 59 bases (k = 31, m = 15: at most 47).  8 B/k-mer become ~1.8 B/base on the links.  mm_pos makes a record
// self-describing: its owner can find the minimizer again (one m-mer extraction + one hash) and bucket the record by it,
// which is what bl_count_super_kmers does to count k-mers in LDS-sized buckets instead of sorting them all.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <vector>

#include "../../include/biolib_amd.h"
#include "bl_partition.hpp"
#include "bl_scan_core.hpp"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);
extern int bl_ctx_device(bl_ctx* ctx);
extern void* bl_ctx_scratch(bl_ctx* ctx, int slot, size_t bytes);  // bl_capi.hip: device scratch that lives with the context

namespace {

constexpr int MAX_PARTS = 64;
constexpr int MAX_BASES = 59;

#define SK_HIP(call)                                                                                                             \
    do {                                                                                                                         \
        hipError_t e_ = (call);                                                                                                  \
        if (e_ != hipSuccess) return bl_set_error(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e_));    \
    } while (0)

__device__ __forceinline__ unsigned code_of(unsigned char c) { return ((c >> 1) ^ (c >> 2)) & 3u; }  // A0 C1 G2 T/U3 (constants.hpp:12-21)

__device__ __forceinline__ unsigned base_at(unsigned long long hi, unsigned long long lo, int i)
{
    return i < 32 ? (unsigned)(hi >> (62 - 2 * i)) & 3u : (unsigned)(lo >> (62 - 2 * (i - 32))) & 3u;
}

__global__ __launch_bounds__(256) void pack_kernel(const unsigned char* __restrict__ bases, unsigned long long n_bases,
                                                   const unsigned long long* __restrict__ first_pos, const unsigned char* __restrict__ sizes,
                                                   const unsigned char* __restrict__ mm_pos, unsigned long long n, int k, ulonglong2* __restrict__ out)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const unsigned long long p = first_pos[g];
    const int size = sizes[g];
    int nb = size + k - 1;
    if (p + (unsigned long long)nb > n_bases) nb = p < n_bases ? (int)(n_bases - p) : 0;  // never read past the batch (a caller error; the record is then short)
    unsigned long long hi = 0, lo = 0;
    const int n_hi = nb < 32 ? nb : 32;
    for (int i = 0; i < n_hi; ++i) hi = (hi << 2) | code_of(bases[p + i]);
    if (n_hi < 32) hi <<= 2 * (32 - n_hi);
    for (int i = 32; i < nb; ++i) lo = (lo << 2) | code_of(bases[p + i]);
    if (nb > 32) lo <<= 64 - 2 * (nb - 32);
    const unsigned long long mp = mm_pos ? (unsigned long long)(mm_pos[g] & 31u) : 0ULL;
    out[g] = make_ulonglong2(hi, (lo & ~0x3ffULL) | (mp << 5) | (unsigned long long)((size - 1) & 31));
}

struct HashArrayOwner {
    const unsigned long long* hashes;
    __device__ uint32_t operator()(unsigned long long i, uint32_t parts) const { return blpart::bucket_of(hashes[i], parts); }
};

__global__ void sizes_kernel(const ulonglong2* recs, unsigned long long n, unsigned long long* sizes)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) sizes[g] = (recs[g].y & 31ULL) + 1;
}

__global__ __launch_bounds__(256) void expand_kernel(const ulonglong2* __restrict__ recs, const unsigned long long* __restrict__ offsets,
                                                     unsigned long long n, int k, int canonical, unsigned long long* __restrict__ out)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const unsigned long long hi = recs[g].x, lo = recs[g].y;
    const int size = (int)(lo & 31ULL) + 1;
    const unsigned long long mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int shift = 2 * (k - 1);
    unsigned long long fwd = 0, rc = 0;
    for (int i = 0; i < k - 1; ++i) {  // kmer_view.hpp:190-199, started from the packed codes
        const unsigned long long c = base_at(hi, lo, i);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3ULL ^ c) << shift);
    }
    unsigned long long* dst = out + offsets[g];
    for (int j = 0; j < size; ++j) {
        const unsigned long long c = base_at(hi, lo, k - 1 + j);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3ULL ^ c) << shift);
        dst[j] = canonical ? (fwd < rc ? fwd : rc) : fwd;
    }
}

}  // namespace

extern "C" {

int bl_pack_super_kmers(bl_ctx* ctx, const bl_batch* batch, const uint64_t* d_first_pos, const uint8_t* d_sizes, const uint8_t* d_mm_pos, uint64_t n_groups,
                        uint32_t k, uint32_t m, uint64_t* d_records)
{
    if (!ctx || !batch || (n_groups && (!d_first_pos || !d_sizes || !d_records))) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (k < 1 || k > 32 || m < 1 || m > k || 2 * k - m > MAX_BASES) return bl_set_error(BL_ERR_INVALID, "need 1 <= m <= k <= 32 and 2k - m <= 59 (bases per packed record)");
    if (n_groups == 0) return BL_OK;
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned char*>(bl_batch_device_bases(batch)),
                       (unsigned long long)bl_batch_n_bases(batch), reinterpret_cast<const unsigned long long*>(d_first_pos), d_sizes, d_mm_pos,
                       (unsigned long long)n_groups, (int)k, reinterpret_cast<ulonglong2*>(d_records));
    SK_HIP(hipGetLastError());
    return BL_OK;
}

int bl_partition_records(bl_ctx* ctx, const uint64_t* d_hashes, const uint64_t* d_records, uint64_t n, uint32_t parts, uint64_t* d_out, uint64_t* counts)
{
    if (!ctx || !counts || parts == 0 || parts > MAX_PARTS || (n && (!d_hashes || !d_records || !d_out)))
        return bl_set_error(BL_ERR_INVALID, "bad argument (1 <= parts <= 64)");
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    unsigned long long host[MAX_PARTS];
    const hipError_t e = blpart::partition(reinterpret_cast<const ulonglong2*>(d_records), (unsigned long long)n, parts,
                                           HashArrayOwner{reinterpret_cast<const unsigned long long*>(d_hashes)}, reinterpret_cast<ulonglong2*>(d_out), host,
                                           bl_ctx_stream(ctx));
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    for (uint32_t b = 0; b < parts; ++b) counts[b] = host[b];
    return BL_OK;
}

int bl_expand_super_kmers(bl_ctx* ctx, const uint64_t* d_records, uint64_t n_groups, uint32_t k, uint32_t flags, uint64_t* d_kmers, uint64_t capacity,
                          uint64_t* n_kmers)
{
    if (!ctx || !n_kmers || (n_groups && !d_records)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (k < 1 || k > 32) return bl_set_error(BL_ERR_INVALID, "need 1 <= k <= 32");
    *n_kmers = 0;
    if (n_groups == 0) return BL_OK;
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long *sizes = nullptr, *offsets = nullptr;
    void* tmp = nullptr;
    size_t bytes = 0;
    const unsigned blocks = (unsigned)((n_groups + 255) / 256);
    hipError_t e = hipMalloc(&sizes, 2 * n_groups * sizeof(unsigned long long));
    if (e != hipSuccess) return bl_set_error(BL_ERR_OOM, hipGetErrorString(e));
    offsets = sizes + n_groups;
    hipLaunchKernelGGL(sizes_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const ulonglong2*>(d_records), (unsigned long long)n_groups, sizes);
    e = rocprim::exclusive_scan(nullptr, bytes, sizes, offsets, 0ull, n_groups, rocprim::plus<unsigned long long>(), s);
    if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 16);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, bytes, sizes, offsets, 0ull, n_groups, rocprim::plus<unsigned long long>(), s);
    unsigned long long last[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&last[0], offsets + n_groups - 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], sizes + n_groups - 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int rc = BL_OK;
    if (e == hipSuccess) {
        *n_kmers = last[0] + last[1];
        if (*n_kmers > capacity || (!d_kmers && *n_kmers)) {
            rc = bl_set_error(BL_ERR_CAPACITY, "expanded k-mers exceed the capacity of d_kmers (n_kmers holds the need)");
        } else {
            hipLaunchKernelGGL(expand_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const ulonglong2*>(d_records), offsets, (unsigned long long)n_groups,
                               (int)k, (flags & BL_FLAG_CANONICAL) ? 1 : 0, reinterpret_cast<unsigned long long*>(d_kmers));
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    (void)hipFree(sizes);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    return rc;
}

}  // extern "C"
