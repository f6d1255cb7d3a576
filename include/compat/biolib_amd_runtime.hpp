// biolib_amd_runtime.hpp — glue shared by the drop-in headers: one lazily created GPU context per
// process and thread, RAII for batches / device arrays, C status -> std::runtime_error (the
// reference signals errors with exceptions).  There is no CPU fallback: without a
// gfx950 device the first view that needs the GPU throws.
#ifndef BIOLIB_AMD_COMPAT_RUNTIME_HPP
#define BIOLIB_AMD_COMPAT_RUNTIME_HPP

#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../biolib_amd.h"

namespace biolib_amd {

inline void check(int rc, const char* what)
{
    if (rc != BL_OK) throw std::runtime_error(std::string("[biolib_amd] ") + what + ": " + bl_last_error());
}

class context
{
    public:
        static bl_ctx* get()
        {
            thread_local context instance;  // bl_ctx is per (thread, device)
            return instance.handle;
        }
    private:
        bl_ctx* handle = nullptr;
        context()
        {
            int device = 0;
            if (const char* e = std::getenv("BIOLIB_AMD_DEVICE")) device = std::atoi(e);
            check(bl_ctx_create(device, &handle), "bl_ctx_create");
        }
        ~context() { bl_ctx_destroy(handle); }
        context(context const&) = delete;
};

struct batch_handle {
    bl_batch* b = nullptr;
    batch_handle(const char* s, std::size_t n) { check(bl_batch_upload(context::get(), s, n, nullptr, 0, &b), "bl_batch_upload"); }
    ~batch_handle() { bl_batch_destroy(b); }
    batch_handle(batch_handle const&) = delete;
};

template <typename T>
struct device_array {
    T* d = nullptr;
    std::size_t n = 0;
    explicit device_array(std::size_t count) : n(count)
    {
        void* p = nullptr;
        check(bl_device_alloc(context::get(), count * sizeof(T), &p), "bl_device_alloc");
        d = static_cast<T*>(p);
    }
    ~device_array() { bl_device_free(context::get(), d); }
    device_array(device_array const&) = delete;
    std::vector<T> to_host(std::size_t count) const
    {
        std::vector<T> h(count);
        check(bl_copy_to_host(context::get(), h.data(), d, count * sizeof(T)), "bl_copy_to_host");
        return h;
    }
};

}  // namespace biolib_amd

#endif
