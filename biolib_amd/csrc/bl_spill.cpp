// bl_spill.cpp — spill / wire formats of the step right AFTER the scan: what biolib's consumers
// write and read, both directions.
//   run file   emem::external_memory_vector<uint64_t>::sort_and_flush (reference external_memory_vector.hpp:243-262):
//              the sorted elements one after the other through io::basic_store = raw little-endian 8-byte values, no
//              header; file name <dir>/tmp.run[_<name>]_<id>.bin (:253-262)
//   vector     io::basic_store(std::vector<uint64_t>) / io::basic_load (io.hpp:104-122): size_t element count, then the elements
//   merge      external_memory_vector::const_iterator (:265-347): a k-way merge over the run files that yields the
//              elements in sorted order — here the runs are read to the device and merged there (bl_merge_runs_u64 in
//              bl_setops.hip does the device part)
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip

extern "C" {

int bl_run_file_name(const char* dir, const char* name, uint64_t id, char* out, uint64_t out_len)
{
    if (!dir || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    std::string fn = std::string(dir) + "/tmp.run";
    if (name && *name) fn += std::string("_") + name;
    fn += "_" + std::to_string(id) + ".bin";
    if (fn.size() + 1 > out_len) return bl_set_error(BL_ERR_INVALID, "file name buffer too small");
    std::memcpy(out, fn.c_str(), fn.size() + 1);
    return BL_OK;
}

static int write_u64_file(bl_ctx* ctx, const uint64_t* d_keys, uint64_t n, const char* path, bool with_count)
{
    if (!ctx || !path || (n && !d_keys)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    std::vector<uint64_t> host(n);
    int rc = bl_copy_to_host(ctx, host.data(), d_keys, n * sizeof(uint64_t));
    if (rc != BL_OK) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return bl_set_error(BL_ERR_INVALID, (std::string("cannot create ") + path).c_str());
    bool ok = true;
    if (with_count) {
        const size_t cnt = (size_t)n;  // io.hpp stores std::size_t
        ok = std::fwrite(&cnt, sizeof(cnt), 1, f) == 1;
    }
    if (ok && n) ok = std::fwrite(host.data(), sizeof(uint64_t), n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? BL_OK : bl_set_error(BL_ERR_INVALID, (std::string("short write to ") + path).c_str());
}

int bl_write_run_u64(bl_ctx* ctx, const uint64_t* d_sorted_keys, uint64_t n, const char* path) { return write_u64_file(ctx, d_sorted_keys, n, path, false); }
int bl_write_vector_u64(bl_ctx* ctx, const uint64_t* d_keys, uint64_t n, const char* path) { return write_u64_file(ctx, d_keys, n, path, true); }

// number of 8-byte elements a file holds: a run file (with_count = 0: size / 8) or a basic_store'd vector (its count word)
int bl_file_count_u64(const char* path, int with_count, uint64_t* n)
{
    if (!path || !n) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *n = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return bl_set_error(BL_ERR_INVALID, (std::string("cannot open ") + path).c_str());
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    int rc = BL_OK;
    if (size < 0) {
        rc = bl_set_error(BL_ERR_INVALID, "cannot size the file");
    } else if (with_count) {
        size_t cnt = 0;
        std::rewind(f);
        if ((size_t)size < sizeof(cnt) || std::fread(&cnt, sizeof(cnt), 1, f) != 1 || (uint64_t)size != sizeof(cnt) + (uint64_t)cnt * 8)
            rc = bl_set_error(BL_ERR_INVALID, "not an io::basic_store'd vector of 8-byte elements (count word and file size disagree)");
        else
            *n = cnt;
    } else if (size % 8) {
        rc = bl_set_error(BL_ERR_INVALID, "a run file of uint64_t must be a multiple of 8 bytes long");
    } else {
        *n = (uint64_t)size / 8;
    }
    std::fclose(f);
    return rc;
}

// read the elements of a run file / vector file into host memory (capacity elements)
int bl_read_file_u64_host(const char* path, int with_count, uint64_t* out, uint64_t capacity, uint64_t* n)
{
    uint64_t cnt = 0;
    int rc = bl_file_count_u64(path, with_count, &cnt);
    if (rc != BL_OK) return rc;
    if (n) *n = cnt;
    if (cnt > capacity) return bl_set_error(BL_ERR_CAPACITY, "output array too small for the file");
    if (cnt && !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return bl_set_error(BL_ERR_INVALID, (std::string("cannot open ") + path).c_str());
    if (with_count) std::fseek(f, (long)sizeof(size_t), SEEK_SET);
    const bool ok = cnt == 0 || std::fread(out, sizeof(uint64_t), cnt, f) == cnt;
    std::fclose(f);
    return ok ? BL_OK : bl_set_error(BL_ERR_INVALID, (std::string("short read from ") + path).c_str());
}

}  // extern "C"
