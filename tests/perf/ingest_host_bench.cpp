// Host-only throughput of the reader (no GPU): decompressed text spans and the record loop, per input file.
//   g++ -O2 -I include tests/perf/ingest_host_bench.cpp -L biolib_amd/lib -lbiolib_amd -o ingest_host_bench
//   ingest_host_bench FILE [threads] [span MiB]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "biolib_amd.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 0;
    const uint64_t span = (argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 256) << 20;
    bl_reader* r = nullptr;
    double t0 = 0, t_text = 0, t_first = 0;
    uint64_t text_bytes = 0, spans = 0, n = 0;
    const char* p = nullptr;
    int rc = 0;
    for (int pass = 0; pass < 2; ++pass) {  // the first pass pays for the first touch of every buffer in the process
        if (pass) bl_reader_close(r);
        if (bl_reader_open_threads(argv[1], threads, &r) != BL_OK) { std::fprintf(stderr, "%s\n", bl_last_error()); return 1; }
        t0 = now();
        text_bytes = spans = 0;
        while ((rc = bl_reader_next_text(r, span, &p, &n)) == BL_OK) { text_bytes += n; ++spans; }
        t_text = now() - t0;
        if (!pass) t_first = t_text;
    }
    const char* kind = bl_reader_kind(r);
    std::printf("{\"file\": \"%s\", \"kind\": \"%s\", \"text_GBps_first_pass\": %.3f, \"text_GBps\": %.3f, \"spans\": %llu, \"text_bytes\": %llu", argv[1], kind, text_bytes / t_first / 1e9, text_bytes / t_text / 1e9,
                (unsigned long long)spans, (unsigned long long)text_bytes);
    bl_reader_close(r);
    if (rc != 1) { std::printf(", \"error\": \"%s\"}\n", bl_last_error()); return 1; }
    if (bl_reader_open_threads(argv[1], threads, &r) != BL_OK) return 1;
    t0 = now();
    uint64_t recs = 0, bases = 0, len = 0;
    const char *name = nullptr, *seq = nullptr;
    while ((rc = bl_reader_next_record(r, &name, &seq, &len)) == BL_OK) { ++recs; bases += len; }
    const double t_rec = now() - t0;
    bl_reader_close(r);
    std::printf(", \"records\": %llu, \"bases\": %llu, \"records_Gbp_s\": %.3f, \"records_text_GBps\": %.3f, \"text_Gbp_s\": %.3f}\n", (unsigned long long)recs,
                (unsigned long long)bases, bases / t_rec / 1e9, text_bytes / t_rec / 1e9, bases / t_text / 1e9);
    return rc == 1 ? 0 : 1;
}
