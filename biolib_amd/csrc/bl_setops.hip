// bl_setops.hip — what happens to k-mers right after the scan in biolib's own consumer
// (tests/test_jaccard.cpp:55-130 in the reference tree): sort, unique, and the sizes
// of intersection / union of two sorted unique sets (include/ordered_unique_sampler.hpp:115-130,
// include/jaccard.hpp:8-37).  The sort and the unique are rocPRIM device primitives (plain library
// plumbing, like a BLAS GEMM would be); the set intersection is a hand-written merge-path-free
// kernel: every element of the smaller... of A binary-searches B (both unique and sorted).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);  // bl_capi.hip
extern int bl_ctx_device(bl_ctx* ctx);
extern void* bl_ctx_scratch(bl_ctx* ctx, int slot, size_t bytes);  // bl_capi.hip: grow-only device scratch of the context (slots 4-6 are ours)

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
    // temporaries from the context's scratch (no hipMalloc / hipFree per call: they cost more than the kernels on small inputs
    // and a device-wide synchronisation each on large ones)
    tmp = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 4, n * sizeof(unsigned long long)));
    d_count = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 6, 64));
    hipError_t e = tmp && d_count ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, sort_bytes, keys, tmp, n, 0, 64, s);
    if (e == hipSuccess) e = rocprim::unique(nullptr, uniq_bytes, tmp, keys, d_count, n, rocprim::equal_to<unsigned long long>(), s);
    const size_t bytes = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
    if (e == hipSuccess) {
        scratch = bl_ctx_scratch(ctx, 5, bytes ? bytes : 16);
        if (!scratch) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess) e = rocprim::radix_sort_keys(scratch, sort_bytes, keys, tmp, n, 0, 64, s);             // keys -> tmp (sorted)
    if (e == hipSuccess) e = rocprim::unique(scratch, uniq_bytes, tmp, keys, d_count, n, rocprim::equal_to<unsigned long long>(), s);  // tmp -> keys
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_count, sizeof(cnt), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
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
    unsigned long long* d_out = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 6, 64));
    if (!d_out) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
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
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    *intersection = inter;
    *union_size = na + nb - inter;
    return BL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// k-mer counting pieces for the multi-GPU bucket exchange: split a key list into `parts`
// buckets by hash64(key, seed) % parts (the owner GPU of a k-mer), and run-length count a sorted list.
#include <rocprim/device/device_run_length_encode.hpp>
#include "bl_scan_core.hpp"

#include "bl_partition.hpp"

namespace {

struct KeyHashOwner {
    const unsigned long long* keys;
    uint32_t seed;
    __device__ uint32_t operator()(unsigned long long i, uint32_t parts) const { return blpart::bucket_of(bl::murmur64(keys[i], seed), parts); }
};

}  // namespace

extern "C" {

int bl_partition_u64(bl_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t parts, uint64_t seed, uint64_t* d_out, uint64_t* counts)
{
    if (!ctx || !counts || parts == 0 || parts > blpart::MAX_PARTS || (n && (!d_keys || !d_out))) return bl_set_error(BL_ERR_INVALID, "bad argument (1 <= parts <= 64)");
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(d_keys);
    unsigned long long host[blpart::MAX_PARTS];
    const hipError_t e = blpart::partition(keys, (unsigned long long)n, parts, KeyHashOwner{keys, (uint32_t)seed}, reinterpret_cast<unsigned long long*>(d_out), host,
                                           bl_ctx_stream(ctx));
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    for (uint32_t b = 0; b < parts; ++b) counts[b] = host[b];
    return BL_OK;
}

// sorted keys (duplicates kept) -> distinct keys + their multiplicities; returns the number of distinct keys
int bl_count_sorted_u64(bl_ctx* ctx, const uint64_t* d_sorted, uint64_t n, uint64_t* d_unique, uint32_t* d_counts, uint64_t* n_unique)
{
    if (!ctx || !n_unique || (n && (!d_sorted || !d_unique || !d_counts))) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *n_unique = 0;
    if (n == 0) return BL_OK;
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long* d_runs = nullptr;
    void* tmp = nullptr;
    size_t bytes = 0;
    d_runs = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 6, 64));
    if (!d_runs) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
    hipError_t e = rocprim::run_length_encode(nullptr, bytes, reinterpret_cast<const unsigned long long*>(d_sorted), n,
                                              reinterpret_cast<unsigned long long*>(d_unique), d_counts, d_runs, s);
    if (e == hipSuccess) {
        tmp = bl_ctx_scratch(ctx, 5, bytes ? bytes : 16);
        if (!tmp) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess)
        e = rocprim::run_length_encode(tmp, bytes, reinterpret_cast<const unsigned long long*>(d_sorted), n, reinterpret_cast<unsigned long long*>(d_unique),
                                       d_counts, d_runs, s);
    unsigned long long runs = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&runs, d_runs, sizeof(runs), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    *n_unique = runs;
    return BL_OK;
}

int bl_sort_u64(bl_ctx* ctx, uint64_t* d_keys, uint64_t n)
{
    if (!ctx || (n && !d_keys)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (n == 0) return BL_OK;
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(d_keys);
    unsigned long long* tmp = nullptr;
    void* scratch = nullptr;
    size_t bytes = 0;
    tmp = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 4, n * sizeof(unsigned long long)));
    if (!tmp) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
    hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, keys, tmp, n, 0, 64, s);
    if (e == hipSuccess) {
        scratch = bl_ctx_scratch(ctx, 5, bytes ? bytes : 16);
        if (!scratch) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess) e = rocprim::radix_sort_keys(scratch, bytes, keys, tmp, n, 0, 64, s);
    if (e == hipSuccess) e = hipMemcpyAsync(keys, tmp, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    return BL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Measured HBM peaks (SURVEY.md §8d "confirm on the box ... report the measured peak next to the spec"): a read-only
// streaming kernel (16 B per lane, grid-stride, XOR-folded so nothing is elided) and a device-to-device copy.
namespace {

struct Quad {
    unsigned int x, y, z, w;
};

__global__ __launch_bounds__(256) void stream_read_kernel(const Quad* __restrict__ src, unsigned long long n16, unsigned int* sink)
{
    unsigned int acc = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {  // four independent loads in flight per lane
        const Quad a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const Quad a = src[i];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;  // practically never: keeps the loads alive
}

}  // namespace

extern "C" int bl_probe_hbm(bl_ctx* ctx, uint64_t n_bytes, int iters, double* read_gbps, double* copy_gbps)
{
    if (!ctx || !read_gbps || !copy_gbps || iters < 1 || n_bytes < (1u << 20)) return bl_set_error(BL_ERR_INVALID, "bad argument (>= 1 MiB, >= 1 iteration)");
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    n_bytes &= ~(uint64_t)15;
    void *a = nullptr, *b = nullptr;
    unsigned int* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, n_bytes);
    if (e == hipSuccess) e = hipMalloc(&b, n_bytes);
    if (e == hipSuccess) e = hipMalloc(&sink, 4);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0x5a, n_bytes, s);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms_read = 0, ms_copy = 0;
    if (e == hipSuccess) {
        const unsigned blocks = 256 * 16;  // 16 workgroups per CU
        hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const Quad*>(a), n_bytes / 16, sink);  // warm-up
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const Quad*>(a), n_bytes / 16, sink);
        (void)hipEventRecord(e1, s);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_read, e0, e1);
    }
    if (e == hipSuccess) {
        e = hipMemcpyAsync(b, a, n_bytes, hipMemcpyDeviceToDevice, s);  // warm-up
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && e == hipSuccess; ++i) e = hipMemcpyAsync(b, a, n_bytes, hipMemcpyDeviceToDevice, s);
        (void)hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_copy, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    *read_gbps = (double)n_bytes * iters / (ms_read * 1e-3) / 1e9;
    *copy_gbps = 2.0 * (double)n_bytes * iters / (ms_copy * 1e-3) / 1e9;  // read + write
    return BL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Shader clock held by the chip WHILE other kernels run (MI355X_MICROARCH.md, DVFS item 6): one wave on its own stream
// sleeps for duration_ms and stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at both ends.  bench.py starts
// it at the head of the timed region, so the VALU ceiling is priced at the clock of the run, not at the nominal 2.4 GHz.
namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    // ends when the host raises out[2] (bl_clock_probe_finish) or, at the latest, after `ticks` x 10 ns: a wall-clock bound that
    // holds whatever else happens
    while (t - t0 < ticks && __hip_atomic_load(&out[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        __builtin_amdgcn_s_sleep(127);
        t = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = t - t0;
    }
}

}  // namespace

struct bl_clock_probe {
    int device;
    hipStream_t stream;
    unsigned long long* pinned;  // [0] cycles, [1] 10-ns ticks, [2] stop request (host -> device)
};

extern "C" int bl_clock_probe_start(bl_ctx* ctx, uint32_t duration_ms, bl_clock_probe** out)
{
    if (!ctx || !out || duration_ms == 0 || duration_ms > 5000) return bl_set_error(BL_ERR_INVALID, "clock probe: need a context and 1..5000 ms");
    *out = nullptr;
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    bl_clock_probe* p = new (std::nothrow) bl_clock_probe{bl_ctx_device(ctx), nullptr, nullptr};
    if (!p) return bl_set_error(BL_ERR_OOM, "host allocation failed");
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->pinned), 4 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) {
        p->pinned[0] = p->pinned[1] = p->pinned[2] = 0;
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, p->stream, p->pinned, (unsigned long long)duration_ms * 100000ull);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (p->pinned) (void)hipHostFree(p->pinned);
        if (p->stream) (void)hipStreamDestroy(p->stream);
        delete p;
        return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    }
    *out = p;
    return BL_OK;
}

extern "C" int bl_clock_probe_finish(bl_clock_probe* p, double* shader_ghz)
{
    if (!p) return bl_set_error(BL_ERR_INVALID, "probe is NULL");
    (void)hipSetDevice(p->device);
    __atomic_store_n(&p->pinned[2], 1ull, __ATOMIC_RELEASE);  // stop now: the probe must never outlive what it measures (a device-wide
                                                              // synchronise would otherwise wait for it)
    hipError_t e = hipStreamSynchronize(p->stream);
    const double cycles = (double)p->pinned[0], ticks = (double)p->pinned[1];
    (void)hipHostFree(p->pinned);
    (void)hipStreamDestroy(p->stream);
    delete p;
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    if (shader_ghz) *shader_ghz = ticks > 0 ? cycles / ticks * 0.1 : 0.0;
    return BL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Read side of the spill format: run files / stored vectors biolib wrote -> device arrays, and the
// k-way merge the reference's external_memory_vector iterator performs (external_memory_vector.hpp:265-347) as a tree of
// pairwise device merges (rocprim::merge: library plumbing) over the runs loaded back to back.
#include <rocprim/device/device_merge.hpp>

extern "C" int bl_read_file_u64(bl_ctx* ctx, const char* path, int with_count, uint64_t* d_out, uint64_t capacity, uint64_t* n)
{
    if (!ctx || !path) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    uint64_t cnt = 0;
    int rc = bl_file_count_u64(path, with_count, &cnt);
    if (rc != BL_OK) return rc;
    if (n) *n = cnt;
    if (cnt > capacity) return bl_set_error(BL_ERR_CAPACITY, "device array too small for the file");
    if (cnt == 0) return BL_OK;
    if (!d_out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    std::vector<uint64_t> host(cnt);
    rc = bl_read_file_u64_host(path, with_count, host.data(), cnt, nullptr);
    if (rc != BL_OK) return rc;
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    SET_HIP(hipMemcpyAsync(d_out, host.data(), cnt * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    SET_HIP(hipStreamSynchronize(s));  // `host` goes away
    return BL_OK;
}

extern "C" int bl_merge_runs_u64(bl_ctx* ctx, const char* const* paths, uint32_t n_paths, uint64_t* d_out, uint64_t capacity, uint64_t* n_total)
{
    if (!ctx || (n_paths && !paths) || !n_total) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    std::vector<uint64_t> len(n_paths), off(n_paths + 1, 0);
    for (uint32_t i = 0; i < n_paths; ++i) {
        int rc = bl_file_count_u64(paths[i], 0, &len[i]);
        if (rc != BL_OK) return rc;
        off[i + 1] = off[i] + len[i];
    }
    const uint64_t total = off[n_paths];
    *n_total = total;
    if (total > capacity) return bl_set_error(BL_ERR_CAPACITY, "device array too small for the merged runs");
    if (total == 0) return BL_OK;
    if (!d_out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    for (uint32_t i = 0; i < n_paths; ++i) {
        int rc = bl_read_file_u64(ctx, paths[i], 0, d_out + off[i], len[i], nullptr);
        if (rc != BL_OK) return rc;
    }
    SET_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    // runs = [off[i], off[i+1]); merge neighbours pairwise, ping-ponging between d_out and a scratch array
    unsigned long long* a = reinterpret_cast<unsigned long long*>(d_out);
    unsigned long long* b = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    std::vector<uint64_t> cur(off);
    bool in_a = true;
    while (cur.size() > 2 && e == hipSuccess) {
        if (!b) e = hipMalloc(reinterpret_cast<void**>(&b), total * sizeof(unsigned long long));
        unsigned long long* src = in_a ? a : b;
        unsigned long long* dst = in_a ? b : a;
        std::vector<uint64_t> next(1, 0);
        for (size_t i = 0; i + 1 < cur.size() && e == hipSuccess; i += 2) {
            const uint64_t lo = cur[i], mid = cur[i + 1], hi = i + 2 < cur.size() ? cur[i + 2] : cur[i + 1];
            if (hi == mid) {  // odd run out: copied through
                if (mid > lo) e = hipMemcpyAsync(dst + lo, src + lo, (mid - lo) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
            } else {
                size_t need = 0;
                e = rocprim::merge(nullptr, need, src + lo, src + mid, dst + lo, mid - lo, hi - mid, rocprim::less<unsigned long long>(), s);
                if (e == hipSuccess && need > tmp_bytes) {
                    if (tmp) { (void)hipStreamSynchronize(s); (void)hipFree(tmp); tmp = nullptr; }
                    tmp_bytes = need;
                    e = hipMalloc(&tmp, tmp_bytes);
                }
                if (e == hipSuccess) e = rocprim::merge(tmp, need, src + lo, src + mid, dst + lo, mid - lo, hi - mid, rocprim::less<unsigned long long>(), s);
            }
            next.push_back(hi);
        }
        cur.swap(next);
        in_a = !in_a;
    }
    if (e == hipSuccess && !in_a) e = hipMemcpyAsync(a, b, total * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
    const hipError_t se = hipStreamSynchronize(s);
    if (e == hipSuccess) e = se;
    if (b) (void)hipFree(b);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    return BL_OK;
}
