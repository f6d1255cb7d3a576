// TEST INFRASTRUCTURE ONLY — extern "C" shim over the reference's external-memory vector and visitor-free POD
// serialisation (include/external_memory_vector.hpp:186-262, include/io.hpp:27-122), compiled from where they
// lie.  Used to generate the spill-format goldens (tests/golden/spill/).
#include <cassert>
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "external_memory_vector.hpp"
#include "io.hpp"

extern "C" {

// push n keys into a sorted external_memory_vector<uint64_t> with the given RAM budget, flush, and report how
// many run files it wrote; they stay on disk until ref_emv_destroy.
void* ref_emv_create(const uint64_t* keys, uint64_t n, uint64_t ram_bytes, const char* tmp_dir, const char* name, uint64_t* n_runs)
{
    auto* v = new emem::external_memory_vector<uint64_t>(ram_bytes, std::string(tmp_dir), std::string(name));
    for (uint64_t i = 0; i < n; ++i) v->push_back(keys[i]);
    v->minimize();
    uint64_t runs = 0;
    for (;; ++runs) {
        std::string fn = std::string(tmp_dir) + "/tmp.run" + (std::string(name).empty() ? "" : "_" + std::string(name)) + "_" + std::to_string(runs) + ".bin";
        std::ifstream f(fn, std::ifstream::binary);
        if (!f.good()) break;
    }
    *n_runs = runs;
    return v;
}

// merged, sorted contents through the reference's own iterator
uint64_t ref_emv_read(void* h, uint64_t* out, uint64_t cap)
{
    auto* v = static_cast<emem::external_memory_vector<uint64_t>*>(h);
    uint64_t i = 0;
    for (auto it = v->cbegin(); it != v->cend(); ++it, ++i)
        if (i < cap) out[i] = *it;
    return i;
}

void ref_emv_destroy(void* h) { delete static_cast<emem::external_memory_vector<uint64_t>*>(h); }

void ref_store_vector_u64(const char* path, const uint64_t* keys, uint64_t n)
{
    std::vector<uint64_t> v(keys, keys + n);
    std::ofstream out(path, std::ofstream::binary);
    io::basic_store(v, out);
}

uint64_t ref_load_vector_u64(const char* path, uint64_t* out, uint64_t cap)
{
    std::vector<uint64_t> v;
    std::ifstream in(path, std::ifstream::binary);
    io::basic_load(in, v);
    for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}

}  // extern "C"
