// bl_multi.cpp — the one collective of the path: the final count reduction across the GPUs of a node,
// on RCCL's C API (ncclAllReduce over xGMI), for single-process hosts that drive one context per device from C or C++
// (the Python binding uses torch.distributed for the same reduction; biolib_amd/shard.py).  librccl.so.1 is opened on
// first use, so a one-GPU deployment does not need it to be installed.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip
struct bl_ctx;
hipStream_t bl_ctx_stream(bl_ctx* ctx);
int bl_ctx_device(bl_ctx* ctx);

namespace {

// the slice of rccl.h this file needs (ABI of NCCL 2.x / RCCL)
typedef struct ncclComm* ncclComm_t;
enum { NCCL_SUCCESS = 0, NCCL_UINT64 = 5, NCCL_SUM = 0 };
struct Rccl {
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) { r.why = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.AllReduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
        if (!r.ok) r.why = "librccl.so.1 lacks the NCCL 2 entry points";
    });
    return r;
}

std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;  // one communicator clique per device list, kept for the process' life

}  // namespace

extern "C" int bl_count_allreduce(bl_ctx* const* ctxs, int n_gpu, uint64_t* counters, int n)
{
    if (!ctxs || !counters || n_gpu < 1 || n < 1) return bl_set_error(BL_ERR_INVALID, "bl_count_allreduce: need contexts, counters, n_gpu >= 1, n >= 1");
    std::vector<int> devs(n_gpu);
    std::set<int> seen;
    for (int g = 0; g < n_gpu; ++g) {
        if (!ctxs[g]) return bl_set_error(BL_ERR_INVALID, "bl_count_allreduce: NULL context");
        devs[g] = bl_ctx_device(ctxs[g]);
        if (!seen.insert(devs[g]).second) return bl_set_error(BL_ERR_INVALID, "bl_count_allreduce: one context per DISTINCT device (RCCL takes one rank per GPU)");
    }
    Rccl& r = rccl();
    if (!r.ok) return bl_set_error(BL_ERR_HIP, r.why.c_str());
    std::vector<ncclComm_t>* comms = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_comm_mutex);
        auto it = g_comms.find(devs);
        if (it == g_comms.end()) {
            std::vector<ncclComm_t> fresh(n_gpu);
            const int rc = r.CommInitAll(fresh.data(), n_gpu, devs.data());
            if (rc != NCCL_SUCCESS) return bl_set_error(BL_ERR_HIP, (std::string("ncclCommInitAll: ") + r.GetErrorString(rc)).c_str());
            it = g_comms.emplace(devs, std::move(fresh)).first;
        }
        comms = &it->second;
    }
    std::vector<unsigned long long*> d(n_gpu, nullptr);
    hipError_t e = hipSuccess;
    int nrc = NCCL_SUCCESS;
    for (int g = 0; g < n_gpu && e == hipSuccess; ++g) {
        e = hipSetDevice(devs[g]);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d[g]), (size_t)n * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d[g], counters + (size_t)g * n, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, bl_ctx_stream(ctxs[g]));
    }
    if (e == hipSuccess) {
        r.GroupStart();
        for (int g = 0; g < n_gpu && nrc == NCCL_SUCCESS; ++g) {
            (void)hipSetDevice(devs[g]);
            nrc = r.AllReduce(d[g], d[g], (size_t)n, NCCL_UINT64, NCCL_SUM, (*comms)[g], bl_ctx_stream(ctxs[g]));
        }
        const int end = r.GroupEnd();
        if (nrc == NCCL_SUCCESS) nrc = end;
    }
    for (int g = 0; g < n_gpu && e == hipSuccess && nrc == NCCL_SUCCESS; ++g) {
        e = hipSetDevice(devs[g]);
        if (e == hipSuccess) e = hipMemcpyAsync(counters + (size_t)g * n, d[g], (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, bl_ctx_stream(ctxs[g]));
    }
    for (int g = 0; g < n_gpu; ++g) {
        (void)hipSetDevice(devs[g]);
        const hipError_t s = hipStreamSynchronize(bl_ctx_stream(ctxs[g]));
        if (e == hipSuccess) e = s;
        if (d[g]) (void)hipFree(d[g]);
    }
    if (nrc != NCCL_SUCCESS) return bl_set_error(BL_ERR_HIP, (std::string("ncclAllReduce: ") + r.GetErrorString(nrc)).c_str());
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    return BL_OK;
}
