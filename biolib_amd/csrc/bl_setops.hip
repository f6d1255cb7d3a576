// bl_setops.hip — what happens to k-mers right after the scan in biolib's own consumer
// (tests/test_jaccard.cpp:55-130 in the reference tree): sort, unique, and the sizes
// of intersection / union of two sorted unique sets (include/ordered_unique_sampler.hpp:115-130,
// include/jaccard.hpp:8-37).  The sort and the unique are rocPRIM device primitives (plain library
// plumbing, like a BLAS GEMM would be); the set intersection is a hand-written merge-path-free
// kernel: every element of the smaller... of A binary-searches B (both unique and sorted).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);  // bl_capi.hip
extern int bl_ctx_device(bl_ctx* ctx);

namespace {

__global__ void intersect_count_kernel(const unsigned long long* a, unsigned long long na, const unsigned long long* b, unsigned long long nb,
                                       unsigned long long* out)
{
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long key = a[i];
        unsigned long long lo = 0, hi = nb;  // first element of b that is >= key
        while (lo < hi) {
            const unsigned long long mid = (lo + hi) >> 1;
            if (b[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        local += (lo < nb && b[lo] == key) ? 1 : 0;
    }
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

#define SET_HIP(call)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return bl_set_error(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

int bl_sort_unique_u64(bl_ctx* ctx, uint64_t* d_keys, uint64_t n, uint64_t* n_unique)
{
    if (!ctx || !n_unique || (n && !d_keys)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *n_unique = 0;
    if (n == 0) return BL_OK;
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(d_keys);
    unsigned long long* tmp = nullptr;
    unsigned long long* d_count = nullptr;
    void* scratch = nullptr;
    size_t sort_bytes = 0, uniq_bytes = 0;
    SET_HIP(hipMalloc(&tmp, n * sizeof(unsigned long long)));
    SET_HIP(hipMalloc(&d_count, sizeof(unsigned long long)));
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, keys, tmp, n, 0, 64, s);
    if (e == hipSuccess) e = rocprim::unique(nullptr, uniq_bytes, tmp, keys, d_count, n, rocprim::equal_to<unsigned long long>(), s);
    const size_t bytes = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
    if (e == hipSuccess) e = hipMalloc(&scratch, bytes ? bytes : 16);
    if (e == hipSuccess) e = rocprim::radix_sort_keys(scratch, sort_bytes, keys, tmp, n, 0, 64, s);             // keys -> tmp (sorted)
    if (e == hipSuccess) e = rocprim::unique(scratch, uniq_bytes, tmp, keys, d_count, n, rocprim::equal_to<unsigned long long>(), s);  // tmp -> keys
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_count, sizeof(cnt), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    (void)hipFree(d_count);
    if (scratch) (void)hipFree(scratch);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    *n_unique = cnt;
    return BL_OK;
}

int bl_jaccard_sorted_u64(bl_ctx* ctx, const uint64_t* d_a, uint64_t na, const uint64_t* d_b, uint64_t nb, uint64_t* intersection,
                          uint64_t* union_size)
{
    if (!ctx || !intersection || !union_size || (na && !d_a) || (nb && !d_b)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long* d_out = nullptr;
    SET_HIP(hipMalloc(&d_out, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s);
    // search the larger set with the elements of the smaller one
    const bool a_small = na <= nb;
    const unsigned long long* x = reinterpret_cast<const unsigned long long*>(a_small ? d_a : d_b);
    const unsigned long long* y = reinterpret_cast<const unsigned long long*>(a_small ? d_b : d_a);
    const unsigned long long nx = a_small ? na : nb, ny = a_small ? nb : na;
    if (e == hipSuccess && nx && ny) {
        const unsigned blocks = (unsigned)((nx + 255) / 256 < 256 * 16 ? (nx + 255) / 256 : 256 * 16);
        hipLaunchKernelGGL(intersect_count_kernel, dim3(blocks), dim3(256), 0, s, x, nx, y, ny, d_out);
        e = hipGetLastError();
    }
    unsigned long long inter = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&inter, d_out, sizeof(inter), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_out);
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    *intersection = inter;
    *union_size = na + nb - inter;
    return BL_OK;
}

}  // extern "C"
