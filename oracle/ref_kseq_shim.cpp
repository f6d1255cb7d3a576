// TEST INFRASTRUCTURE ONLY — extern "C" shim over the reference's own FASTA/FASTQ reader
// (tests/kseq.h in the reference tree, MIT, used by tests/test_kmer_view.cpp:30-42), compiled from where
// it lies.  Used to generate / check the ingest goldens (tests/golden/ingest/).
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <optional>
#include <cassert>
#include <iterator>
#include "kmer_view.hpp"

extern "C" {
#include "kseq.h"
}
KSEQ_INIT(gzFile, gzread)

extern "C" {

// Reads the whole file.  bases: concatenated sequences (capacity cap_bases), offsets: n+1 entries
// (capacity cap_seqs+1), names: '\n'-joined (capacity cap_names).  Returns the number of records, or the
// negative kseq error code of the first failing record (-2 truncated quality, -3 stream error), or -100 on overflow.
long ref_kseq_read_all(const char* path, char* bases, uint64_t cap_bases, uint64_t* offsets, uint64_t cap_seqs, char* names, uint64_t cap_names)
{
    gzFile fp = gzopen(path, "r");
    if (!fp) return -99;
    kseq_t* seq = kseq_init(fp);
    long n = 0, rc;
    uint64_t nb = 0, nn = 0;
    offsets[0] = 0;
    while ((rc = kseq_read(seq)) >= 0) {
        if ((uint64_t)n >= cap_seqs || nb + seq->seq.l > cap_bases || nn + seq->name.l + 1 > cap_names) { n = -100; break; }
        std::memcpy(bases + nb, seq->seq.s, seq->seq.l);
        nb += seq->seq.l;
        std::memcpy(names + nn, seq->name.s, seq->name.l);
        nn += seq->name.l;
        names[nn++] = '\n';
        offsets[++n] = nb;
    }
    if (n >= 0 && rc < -1) n = rc;
    kseq_destroy(seq);
    gzclose(fp);
    return n;
}

// The reference driver's own loop (tests/test_kmer_view.cpp:30-42): its reader, one kmer_view per record, every k-mer used.
// Timed by tests/perf/view_loop_bench.py as the CPU figure beside the drop-in's pooled loop.  Returns the XOR of the values.
uint64_t ref_read_loop_kmers_xor(const char* path, uint8_t k, int canonical, uint64_t* n_reads, uint64_t* n_kmers)
{
    gzFile fp = gzopen(path, "r");
    if (!fp) return 0;
    kseq_t* seq = kseq_init(fp);
    uint64_t x = 0, reads = 0, kmers = 0;
    while (kseq_read(seq) >= 0) {
        auto view = wrapper::kmer_view_from_cstr<uint64_t>(seq->seq.s, seq->seq.l, k, canonical != 0);
        for (auto itr = view.cbegin(); itr != view.cend(); ++itr) {
            if ((*itr).value) { x ^= *((*itr).value); ++kmers; }
        }
        ++reads;
    }
    kseq_destroy(seq);
    gzclose(fp);
    if (n_reads) *n_reads = reads;
    if (n_kmers) *n_kmers = kmers;
    return x;
}

}  // extern "C"
