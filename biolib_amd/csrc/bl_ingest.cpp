// bl_ingest.cpp — host-side FASTA / FASTQ (plain or gzip) reader feeding device batches: the step
// immediately before the scan.  Record semantics follow the reader the
// reference's own tools use (tests/kseq.h:185-234 in the reference tree, studied, not copied):
//   * a record starts at the next '>' or '@'; name = up to the first whitespace, comment = rest of the line
//   * sequence lines are concatenated until a line starts with '>', '@' or '+'; empty lines are skipped;
//     a trailing '\r' of a line is dropped (when the accumulated sequence is longer than one char)
//   * after '+': the rest of that line is skipped and quality lines are consumed until they are at least
//     as long as the sequence; a different total length is a malformed record (BL_ERR_INVALID)
// Bases are passed through untouched (the scan's own table decides what is a break).
#include <zlib.h>

#include <cctype>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip

struct bl_reader {
    gzFile f = nullptr;
    std::vector<unsigned char> buf;
    int begin = 0, end = 0;
    bool eof = false, err = false;
    int last_char = 0;
    // current record
    std::string name, comment, seq, qual;
    // last batch
    std::string bases;
    std::vector<uint64_t> offsets;
    std::vector<std::string> names;
    bool have_pending = false;  // a record was parsed but did not fit the previous batch

    int getc()
    {
        if (err) return -3;
        if (begin >= end) {
            if (eof) return -1;
            begin = 0;
            end = gzread(f, buf.data(), (unsigned)buf.size());
            if (end == 0) { eof = true; return -1; }
            if (end < 0) { eof = true; err = true; end = 0; return -3; }
        }
        return buf[begin++];
    }

    // append up to (not including) the delimiter; line mode: delimiter '\n', else any isspace()
    // returns length so far, or -1 if nothing could be read at EOF, -3 on stream error; *dret = delimiter seen (0 at EOF)
    long get_until(bool line, std::string& str, int* dret, bool append)
    {
        bool gotany = false;
        if (dret) *dret = 0;
        if (!append) str.clear();
        for (;;) {
            if (err) return -3;
            if (begin >= end) {
                if (eof) break;
                begin = 0;
                end = gzread(f, buf.data(), (unsigned)buf.size());
                if (end == 0) { eof = true; break; }
                if (end < 0) { eof = true; err = true; end = 0; return -3; }
            }
            int i = begin;
            if (line) { while (i < end && buf[i] != '\n') ++i; }
            else { while (i < end && !std::isspace(buf[i])) ++i; }
            gotany = true;
            str.append(reinterpret_cast<const char*>(buf.data()) + begin, (size_t)(i - begin));
            begin = i + 1;
            if (i < end) {
                if (dret) *dret = buf[i];
                break;
            }
        }
        if (!gotany && eof && begin >= end) return -1;
        if (line && str.size() > 1 && str.back() == '\r') str.pop_back();
        return (long)str.size();
    }

    // >= 0 sequence length, -1 EOF, -2 truncated / mismatched quality, -3 stream error
    long next()
    {
        int c;
        if (last_char == 0) {
            while ((c = getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) return c;
            last_char = c;
        }
        comment.clear(); seq.clear(); qual.clear();
        long r = get_until(false, name, &c, false);
        if (r < 0) return r;
        if (c != '\n') get_until(true, comment, nullptr, false);
        while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            get_until(true, seq, nullptr, true);
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') return (long)seq.size();  // FASTA (or end of file)
        while ((c = getc()) >= 0 && c != '\n') {}
        if (c == -1) return -2;
        long q;
        while ((q = get_until(true, qual, nullptr, true)) >= 0 && qual.size() < seq.size()) {}
        if (q == -3) return -3;
        last_char = 0;
        if (seq.size() != qual.size()) return -2;
        return (long)seq.size();
    }
};

extern "C" {

int bl_reader_open(const char* path, bl_reader** out)
{
    if (!path || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    bl_reader* r = new (std::nothrow) bl_reader();
    if (!r) return bl_set_error(BL_ERR_OOM, "host allocation failed");
    r->f = gzopen(path, "rb");
    if (!r->f) {
        delete r;
        return bl_set_error(BL_ERR_INVALID, (std::string("cannot open ") + path).c_str());
    }
    gzbuffer(r->f, 1 << 20);
    r->buf.resize(1 << 18);
    *out = r;
    return BL_OK;
}

int bl_reader_close(bl_reader* r)
{
    if (!r) return BL_OK;
    if (r->f) gzclose(r->f);
    delete r;
    return BL_OK;
}

int bl_reader_next_record(bl_reader* r, const char** name, const char** seq, uint64_t* seq_len)
{
    if (!r || !seq_len) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    const long n = r->next();
    if (n == -1) { *seq_len = 0; if (name) *name = nullptr; if (seq) *seq = nullptr; return 1; }  // end of file
    if (n == -2) return bl_set_error(BL_ERR_INVALID, "truncated or mismatched FASTQ quality string");
    if (n < 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
    if (name) *name = r->name.c_str();
    if (seq) *seq = r->seq.data();
    *seq_len = (uint64_t)n;
    return BL_OK;
}

int bl_reader_next_batch(bl_ctx* ctx, bl_reader* r, uint64_t max_bases, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !r || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    r->bases.clear();
    r->offsets.assign(1, 0);
    r->names.clear();
    for (;;) {
        if (!r->have_pending) {
            const long n = r->next();
            if (n == -1) break;
            if (n == -2) return bl_set_error(BL_ERR_INVALID, "truncated or mismatched FASTQ quality string");
            if (n < 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
        }
        r->have_pending = false;
        if (!r->names.empty() && max_bases && r->bases.size() + r->seq.size() > max_bases) {
            r->have_pending = true;  // keep the parsed record for the next batch
            break;
        }
        r->bases += r->seq;
        r->offsets.push_back(r->bases.size());
        r->names.push_back(r->name);
    }
    if (n_seqs) *n_seqs = r->names.size();
    if (n_bases) *n_bases = r->bases.size();
    if (r->names.empty()) return BL_OK;  // end of file: *out stays NULL
    return bl_batch_upload(ctx, r->bases.data(), r->bases.size(), r->offsets.data(), r->names.size(), out);
}

int bl_reader_last_batch(bl_reader* r, const char** bases, const uint64_t** offsets, uint64_t* n_seqs)
{
    if (!r) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (bases) *bases = r->bases.data();
    if (offsets) *offsets = r->offsets.data();
    if (n_seqs) *n_seqs = r->names.size();
    return BL_OK;
}

const char* bl_reader_last_name(bl_reader* r, uint64_t i) { return (r && i < r->names.size()) ? r->names[i].c_str() : nullptr; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Spill / wire formats of the step right AFTER the scan: what biolib's consumers read.
//   run file   emem::external_memory_vector<uint64_t>::sort_and_flush (external_memory_vector.hpp:243-262):
//              the sorted elements one after the other through io::basic_store = raw little-endian 8-byte
//              values, no header; file name <dir>/tmp.run[_<name>]_<id>.bin (:253-262)
//   vector     io::basic_store(std::vector<uint64_t>) (io.hpp:104-112): size_t element count, then the elements
#include <cstdio>

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

}  // extern "C"
