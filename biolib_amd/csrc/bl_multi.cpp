// bl_multi.cpp — the one collective of the path: the final count reduction across the GPUs of a node,
// on RCCL's C API (ncclAllReduce over xGMI), for single-process hosts that drive one context per device from C or C++
// (the Python binding uses torch.distributed for the same reduction; biolib_amd/shard.py).  Compiled against the installed
// <rccl/rccl.h>; librccl.so.1 itself is opened on first use — a one-GPU deployment does not need it at run time — and must
// report the header's major version.
//
// Build-time requirement: the RCCL development header.  A ROCm install without it still builds the library — bl_count_allreduce then
// returns BL_ERR_HIP with a message that says so (every other entry point is unaffected: one-GPU deployments need neither the
// header nor the library).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>) && !defined(BL_NO_RCCL_HEADER)
#define BL_HAVE_RCCL_HEADER 1
#endif
#endif
#ifdef BL_HAVE_RCCL_HEADER
#include <rccl/rccl.h>  // types, enumerators and prototypes of the INSTALLED RCCL; the library itself is opened on first use
#endif

#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip

#ifndef BL_HAVE_RCCL_HEADER
extern "C" int bl_count_allreduce(bl_ctx* const*, int, uint64_t*, int)
{
    return bl_set_error(BL_ERR_HIP, "bl_count_allreduce: this library was built without <rccl/rccl.h> (RCCL development header absent at build time)");
}
#else
struct bl_ctx;
hipStream_t bl_ctx_stream(bl_ctx* ctx);
int bl_ctx_device(bl_ctx* ctx);

namespace {

// entry points, typed by the header's own prototypes (a mismatch between this file and rccl.h is a compile error)
struct Rccl {
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int version = 0;
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
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!(r.GetVersion && r.CommInitAll && r.AllReduce && r.GroupStart && r.GroupEnd && r.GetErrorString)) {
            r.why = "librccl.so.1 lacks the NCCL 2 entry points";
            return;
        }
        // the library that was found must speak the ABI of the header this file was compiled against: same major version
        if (r.GetVersion(&r.version) != ncclSuccess || r.version / 10000 != NCCL_MAJOR) {
            r.why = "librccl.so.1 reports version " + std::to_string(r.version) + ", this library was built against NCCL " + std::to_string(NCCL_MAJOR) + ".x headers";
            return;
        }
        r.ok = true;
    });
    return r;
}

// One communicator clique per device list, kept for the process' life, with the small device buffers the reduction runs in
// (grown on demand, never freed: a hipMalloc / hipFree pair per call would synchronise every device twice) and a mutex:
// group calls on one set of communicators must not interleave between threads.
struct Clique {
    std::vector<ncclComm_t> comms;
    std::vector<unsigned long long*> buf;
    size_t cap = 0;  // counters each buffer holds
    std::mutex m;
};
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::unique_ptr<Clique>> g_cliques;

struct DeviceGuard {  // the caller's current device, put back on every way out
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

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
    DeviceGuard keep_device;
    Clique* q = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_comm_mutex);
        auto it = g_cliques.find(devs);
        if (it == g_cliques.end()) {
            auto fresh = std::make_unique<Clique>();
            fresh->comms.resize(n_gpu);
            fresh->buf.assign(n_gpu, nullptr);
            const ncclResult_t rc = r.CommInitAll(fresh->comms.data(), n_gpu, devs.data());
            if (rc != ncclSuccess) return bl_set_error(BL_ERR_HIP, (std::string("ncclCommInitAll: ") + r.GetErrorString(rc)).c_str());
            it = g_cliques.emplace(devs, std::move(fresh)).first;
        }
        q = it->second.get();
    }
    std::lock_guard<std::mutex> one_at_a_time(q->m);
    hipError_t e = hipSuccess;
    if ((size_t)n > q->cap) {  // grow the per-device buffers (rare: the reduction is a handful of counters)
        const size_t cap = (size_t)n < 64 ? 64 : (size_t)n;
        for (int g = 0; g < n_gpu && e == hipSuccess; ++g) {
            e = hipSetDevice(devs[g]);
            if (e == hipSuccess && q->buf[g]) e = hipFree(q->buf[g]);
            q->buf[g] = nullptr;
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&q->buf[g]), cap * sizeof(uint64_t));
        }
        q->cap = e == hipSuccess ? cap : 0;
        if (e != hipSuccess) return bl_set_error(BL_ERR_OOM, (std::string("bl_count_allreduce buffers: ") + hipGetErrorString(e)).c_str());
    }
    ncclResult_t nrc = ncclSuccess;
    for (int g = 0; g < n_gpu && e == hipSuccess; ++g) {
        e = hipSetDevice(devs[g]);
        if (e == hipSuccess) e = hipMemcpyAsync(q->buf[g], counters + (size_t)g * n, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, bl_ctx_stream(ctxs[g]));
    }
    if (e == hipSuccess) {
        r.GroupStart();
        for (int g = 0; g < n_gpu && nrc == ncclSuccess; ++g) {
            (void)hipSetDevice(devs[g]);
            nrc = r.AllReduce(q->buf[g], q->buf[g], (size_t)n, ncclUint64, ncclSum, q->comms[g], bl_ctx_stream(ctxs[g]));
        }
        const ncclResult_t end = r.GroupEnd();
        if (nrc == ncclSuccess) nrc = end;
    }
    for (int g = 0; g < n_gpu && e == hipSuccess && nrc == ncclSuccess; ++g) {
        e = hipSetDevice(devs[g]);
        if (e == hipSuccess) e = hipMemcpyAsync(counters + (size_t)g * n, q->buf[g], (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, bl_ctx_stream(ctxs[g]));
    }
    for (int g = 0; g < n_gpu; ++g) {
        (void)hipSetDevice(devs[g]);
        const hipError_t s = hipStreamSynchronize(bl_ctx_stream(ctxs[g]));
        if (e == hipSuccess) e = s;
    }
    if (nrc != ncclSuccess) return bl_set_error(BL_ERR_HIP, (std::string("ncclAllReduce: ") + r.GetErrorString(nrc)).c_str());
    if (e != hipSuccess) return bl_set_error(BL_ERR_HIP, hipGetErrorString(e));
    return BL_OK;
}

#endif  // BL_HAVE_RCCL_HEADER
