/* biolib_amd.h — C ABI of the MI355X-native k-mer / minimizer streaming scan.
 *
 * This is the drop-in boundary.  The reference (yhhshb/biolib) has no FFI layer:
 * its boundary is the header-only C++ template surface, so each entry point below names the
 * reference template whose bulk result it produces, and the headers under include/compat/ re-expose those
 * templates (same names, signatures and iteration protocol) on top of these calls.
 *
 *   bl_scan_kmers        wrapper::kmer_view<uint64_t,It>                  include/kmer_view.hpp:25-83 and 162-234
 *                        + hash::hash64::hash(value, seed)                include/hash.hpp:50-59
 *   bl_scan_minimizers   wrapper::minimizer_view<K,M,hash64,It>           include/minimizer_view.hpp:14-98 (intended semantics)
 *                        sampler::minimizer_sampler<It,Hash>              include/minimizer_sampler.hpp:12-70
 *   bl_scan_super_kmers  wrapper::super_kmer_view<K,M,hash64>             include/super_kmer_view.hpp:11-58, 121-135
 *   bl_scan_super_kmer_records  the same groups as self-contained records  include/super_kmer_view.hpp:20-24 (the record), §8f rank 4
 *   bl_scan_syncmers     sampler::syncmer_sampler<It,minimizer_position_extractor>
 *                                                                         include/syncmer_sampler.hpp:9-137, include/kmer_view.hpp:250-283
 *   bl_hash64_u64        hash::hash64::hash<uint64_t>                     include/hash.hpp:55-59 (host-side convenience, bit-exact)
 *
 * Conventions
 *   - Plain C: opaque handles, plain pointers and sizes.  Never throws; every call returns a status
 *     (0 = BL_OK, negative = error) and bl_last_error() gives the message of the calling thread's
 *     last failure.
 *   - A batch is a set of sequences concatenated without separators; offsets[0..n_seqs] delimits
 *     them.  All positions reported are 0-based indices into that concatenation ("global positions");
 *     position - offsets[seq] is the reference's per-view position.
 *   - Output arrays are DEVICE pointers supplied by the caller (any may be NULL = not wanted) and
 *     hold `capacity` records; the scan always returns the full count, so a short buffer is
 *     detected (BL_ERR_CAPACITY) and can be re-run.  Records are in increasing position order.
 *   - All work is enqueued on the context's HIP stream.  Scans are asynchronous unless
 *     BL_FLAG_SYNC is given; results (`bl_result`) are valid after bl_ctx_sync().
 *   - One context per (host thread, device).  A context is not thread-safe; distinct contexts are
 *     independent.
 */
#ifndef BIOLIB_AMD_H
#define BIOLIB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BL_VERSION 100 /* 0.1.0 */

enum {
    BL_OK = 0,
    BL_ERR_INVALID = -1,     /* bad argument (k > 32, w > 64, NULL handle, misaligned pointer, ...) */
    BL_ERR_HIP = -2,         /* a HIP runtime call failed */
    BL_ERR_OOM = -3,         /* device or pinned-host allocation failed */
    BL_ERR_CAPACITY = -4,    /* output arrays too small: result.count says how many records exist */
    BL_ERR_NO_DEVICE = -5,   /* no gfx950 device visible */
    BL_ERR_INTERNAL = -6     /* an internal consistency check failed (never expected; results invalid) */
};

enum {
    BL_FLAG_CANONICAL = 1u << 0, /* numeric min(forward, reverse complement), kmer_view.hpp:196 */
    BL_FLAG_DROP_LAST = 1u << 1, /* reproduce the `it != cend()` idiom: the k-mer that ends a sequence is skipped (quirk Q1) */
    BL_FLAG_SYNC = 1u << 2       /* wait for completion and fill `result` before returning */
};

typedef struct bl_ctx bl_ctx;
typedef struct bl_batch bl_batch;

/* Filled asynchronously; read after bl_ctx_sync() (or immediately with BL_FLAG_SYNC).  The struct passed to a scan
 * must stay alive until then (the library writes it during the sync); pass NULL if the digest is not wanted. */
typedef struct bl_result {
    uint64_t count;      /* records found (k-mers / minimizer occurrences / super-k-mers / syncmers) */
    uint64_t xor_value;  /* XOR of the 2-bit packed values of all records */
    uint64_t xor_hash;   /* XOR of their 64-bit hashes */
    uint64_t xor_pos;    /* XOR of their global positions (k-mer scan: wrapping SUM of hashes instead) */
    uint64_t aux;        /* super-k-mers: number of group ends seen (== count when consistent) */
    int32_t status;      /* BL_OK or BL_ERR_CAPACITY */
    int32_t redone;      /* diagnostic: tiles whose pass 1 could not decide a window on what it looks at (an approximation of the hash's
                          * high dword, or high dwords alone) and were counted a second time on the hashes themselves; the records are the
                          * same either way */
} bl_result;

const char* bl_last_error(void);
int bl_version(void);
int bl_device_count(int* n);

/* ---- contexts ---------------------------------------------------------------------------------- */
int bl_ctx_create(int device, bl_ctx** out);
/* Destroying a context also destroys the batches created on it that are still alive (their handles
 * become invalid). */
int bl_ctx_destroy(bl_ctx* ctx);
/* Borrow the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream): every call of this context is then
 * enqueued on it, in order with the caller's own work.  NULL is a stream like any other here — the legacy default
 * stream, which is what torch's default stream is — so a NULL handle is honoured, not read as "unset".
 * bl_ctx_use_own_streams returns to the context's own (non-blocking) streams, which are NOT ordered with any other
 * stream: device buffers handed to the library must then be complete (synchronise the producing stream first). */
int bl_ctx_set_stream(bl_ctx* ctx, void* hip_stream);
int bl_ctx_use_own_streams(bl_ctx* ctx);
int bl_ctx_sync(bl_ctx* ctx);
/* Execution lanes of a context that runs on its own streams: with n = 2 consecutive asynchronous scans alternate between two
 * streams, staggered so that the record pass of one scan runs beside the hashing pass of the next (+12 % on MI355X).  Scans in
 * flight together must not share output arrays.  n = 1 (default) keeps scans strictly ordered.  Ignored on a borrowed stream. */
/* (Measured on MI355X: a stream of minimizer scans with records runs fastest in ranges of 0.5-1 Gbp — 525-535 Gbp/s with two lanes, 2-2.5 %
 * above ranges of 1.5 Gbp; below 0.3 Gbp the per-scan launches show.  Consecutive scans must not share output arrays: double-buffer them.) */
int bl_ctx_set_lanes(bl_ctx* ctx, int n);

/* ---- batches (device-resident sequences) -------------------------------------------------------- */
/* Copy host sequences to the device.  offsets has n_seqs+1 entries, offsets[0] = 0,
 * offsets[n_seqs] = n_bases; NULL offsets = one sequence. */
int bl_batch_upload(bl_ctx* ctx, const char* bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, bl_batch** out);
/* The same for reads of ONE length laid end to end (a shorter last read is kept): no offsets array to build or to check — a
 * batch of 150-bp reads has 7 million of them per gigabase.  Such batches are scanned by the read-tiled kernels. */
int bl_batch_upload_reads(bl_ctx* ctx, const char* bases, uint64_t n_bases, uint64_t read_len, bl_batch** out);
/* Wrap bases already in device memory (16-byte aligned, not copied, must outlive the batch).
 * Sequences are either given by host `offsets` (as above) or, if offsets is NULL and read_len > 0,
 * consecutive slices of read_len bases (a shorter last read is kept); both NULL/0 = one sequence. */
int bl_batch_from_device(bl_ctx* ctx, const void* d_bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs,
                         uint64_t read_len, bl_batch** out);
/* Generate SURVEY.md §8d synthetic DNA on the device:
 *   base[i] = "ACGT"[(splitmix64(seed + (i>>5)) >> (2*(i&31))) & 3], reads = consecutive read_len slices
 * (read_len = 0: one sequence). */
int bl_batch_synth(bl_ctx* ctx, uint64_t seed, uint64_t n_bases, uint64_t read_len, bl_batch** out);
int bl_batch_destroy(bl_batch* batch);
uint64_t bl_batch_n_bases(const bl_batch* batch);
uint64_t bl_batch_n_seqs(const bl_batch* batch);
const void* bl_batch_device_bases(const bl_batch* batch);
/* A batch that is a PIECE of a longer concatenation (one GPU's part of a contig that was cut across GPUs, SURVEY.md §8e;
 * the reference streams a contig of any length through one view, kmer_view.hpp:46-54): `origin` is the position its base 0 has
 * in the whole.  Every position a scan of this batch reports (d_positions, d_first_pos) and folds into xor_pos is then
 * origin + the position inside the batch, so the records of the pieces concatenate to the records of the whole; the
 * ranges [first, first+n) of the scan calls stay relative to the batch.  Default 0. */
int bl_batch_set_origin(bl_batch* batch, uint64_t origin);
uint64_t bl_batch_origin(const bl_batch* batch);
/* Copy bases [first, first+n) back to the host (synchronous). */
int bl_batch_download(bl_batch* batch, uint64_t first, uint64_t n, char* out);

/* ---- scans ---------------------------------------------------------------------------------------
 * Every scan works on the windows / k-mers whose FIRST base lies in [first, first+n) of the batch
 * (n = 0 means "to the end"); bases beyond the range are read as needed, so the union of the results
 * of consecutive ranges equals the result of one scan over their union.  A range may hold at most
 * 2^31 positions. */

/* k-mers (C2): for every position p in the range, dense arrays indexed by p - first:
 *   d_values[p-first]  2-bit packed (canonical) k-mer, first base in the most significant pair
 *   d_hashes[p-first]  hash64(value, seed) = low 64 bits of MurmurHash3_x64_128 of its 8 raw bytes
 *   d_valid[p-first]   1 if a k-mer starts at p (no break inside, not crossing a sequence end), else 0
 * result: count, xor_value, xor_hash, xor_pos := wrapping sum of hashes.  1 <= k <= 32. */
int bl_scan_kmers(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t k, uint64_t seed, uint32_t flags,
                  uint64_t* d_values, uint64_t* d_hashes, uint8_t* d_valid, bl_result* result);

/* minimizers (C3): hashed unit = (canonical) `unit`-mer, window = w consecutive units inside one
 * sequence and one break-free run, leftmost minimum hash; one record each time the minimizer
 * occurrence changes (or a run begins):
 *   d_values[r] unit value, d_positions[r] global start position of the unit, d_hashes[r] its hash.
 * In biolib's naming this is minimizer_view(k = unit + w - 1, m = unit).  1 <= unit <= 32, 1 <= w <= 64. */
int bl_scan_minimizers(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t unit, uint32_t w, uint64_t seed,
                       uint32_t flags, uint64_t* d_values, uint64_t* d_positions, uint64_t* d_hashes, uint64_t capacity,
                       bl_result* result);

/* hash_sampler over kmer_view (reference hash_sampler.hpp:73-78,136-141): the k-mers whose
 * hash64(value, seed) is BELOW `threshold` (the reference computes threshold = rate * 2^64-1 in double), position-ordered,
 * same arrays as bl_scan_minimizers.  threshold = UINT64_MAX with BL_FLAG_DROP_LAST gives exactly the k-mers the
 * reference idiom visits.  (bl_scan_minimizers with w = 1 is the same list without a threshold.) */
int bl_scan_hash_sample(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t k, uint64_t seed, uint64_t threshold,
                        uint32_t flags, uint64_t* d_values, uint64_t* d_positions, uint64_t* d_hashes, uint64_t capacity,
                        bl_result* result);

/* super-k-mers (C4): maximal groups of consecutive k-mers sharing one minimizer occurrence
 * (m-mer, w = k - m + 1):
 *   d_minimizers[r] m-mer value, d_first_pos[r] global position of the group's first k-mer,
 *   d_mm_pos[r] minimizer offset inside that k-mer, d_sizes[r] number of k-mers (<= w),
 *   d_hashes[r] hash of the minimizer.  Any of the arrays may be NULL. */
int bl_scan_super_kmers(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t k, uint32_t m, uint64_t seed,
                        uint32_t flags, uint64_t* d_minimizers, uint64_t* d_first_pos, uint8_t* d_mm_pos, uint8_t* d_sizes,
                        uint64_t* d_hashes, uint64_t capacity, bl_result* result);

/* syncmers (C5): k-mers whose minimum-hash s-mer (leftmost in the canonical k-mer, hash seed `seed`,
 * 0 in the reference) sits at offset start_offset or end_offset:
 *   d_positions[r] global position of the k-mer (may be NULL: count only).
 * result: count, xor_pos. */
int bl_scan_syncmers(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t k, uint32_t s, uint32_t start_offset,
                     uint32_t end_offset, uint64_t seed, uint32_t flags, uint64_t* d_positions, uint64_t capacity,
                     bl_result* result);

/* on != 0: every later scan on this context decides its windows on the 64-bit hashes themselves — no pass 1 on the approximate high
 * dword, no closed-syncmer form (§5.4).  Same records, 2-4 % slower; for checks and A/B measurements. */
int bl_ctx_set_exact_windows(bl_ctx* ctx, int on);
/* Tuning and test switches of a context, by name (BL_ERR_INVALID for a name or value it does not know).  None of them changes a result.
 *   "exact_windows"   0 / 1       = bl_ctx_set_exact_windows
 *   "lanes"           1 / 2       = bl_ctx_set_lanes
 *   "position_tiled"  0 / 1       1: batches of fixed-length reads are scanned by the position-tiled kernels too (default 0: read-tiled
 *                                 where that layout applies)
 *   "emit_lds_bytes"  0 or bytes  two-lane contexts: LDS footprint the record pass's workgroups are padded to, which caps how many of
 *                                 them a CU holds beside the next scan's hashing pass (0: the built-in default per scan kind)
 * The library reads no environment variable on the scan path. */
int bl_ctx_set_option(bl_ctx* ctx, const char* name, int64_t value);

/* Elapsed GPU time of the most recent scan call on this context, from HIP events recorded on the
 * context's stream around its kernels (milliseconds).  Synchronises. */
int bl_ctx_last_scan_ms(bl_ctx* ctx, float* ms);
/* Per-launch timing of the main scan kernel alone: after bl_ctx_kernel_timing(ctx, 1) every scan
 * brackets its tile kernel with a HIP event pair on the context's stream; bl_ctx_kernel_time
 * synchronises and returns the summed kernel time and the number of launches since timing was
 * switched on (switching it on or off resets both). */
int bl_ctx_kernel_timing(bl_ctx* ctx, int enable);
int bl_ctx_kernel_time(bl_ctx* ctx, double* total_ms, uint64_t* launches);
/* Markers on the device's timeline that do not stop it.  bl_ctx_mark enqueues one behind everything issued so far on the
 * context's stream(s); bl_ctx_mark_times synchronises, returns for each marker the milliseconds after marker 0 was reached at
 * which the work in front of it had finished (with two lanes: on both), and forgets the markers.  A benchmark times its steps
 * with them: a bl_ctx_sync between steps would cost the overlap of one step's last record pass with the next step's first scan.
 * At most 4,096 markers may be outstanding (BL_ERR_CAPACITY from bl_ctx_mark beyond that).  bl_ctx_mark_times with a capacity
 * below the number of markers returns that number in *n_marks and keeps them: BL_OK for capacity 0 (a size query), BL_ERR_CAPACITY otherwise. */
int bl_ctx_mark(bl_ctx* ctx);
int bl_ctx_mark_times(bl_ctx* ctx, double* ms, uint32_t capacity, uint32_t* n_marks);

/* ---- ingest: FASTA / FASTQ, plain or gzip -> batches -----------------------------
 * Record semantics are those of the reader biolib's own tools use (reference tests/kseq.h:185-234): a record
 * starts at '>' or '@'; the name ends at the first whitespace; sequence lines are concatenated, empty lines and a
 * line-final '\r' dropped; FASTQ quality must match the sequence length.  Bases are not altered: the scans treat
 * everything but ACGTUacgtu as a break, exactly as the reference table does (constants.hpp:12-21). */
typedef struct bl_reader bl_reader;
/* The file is decompressed by background threads that run ahead of the parser: BGZF (bgzip) members are inflated in parallel
 * by `threads` workers (0 = one per core, at most 16); any other gzip file by the same number of workers that each decode a
 * part of the ONE deflate stream from a block start they find by themselves, without the 32 KiB of text before it, which is
 * filled in when the part before is done (pipes and files of a few MiB: one zlib thread); plain files are read ahead. */
int bl_reader_open(const char* path, bl_reader** out);
int bl_reader_open_threads(const char* path, int threads, bl_reader** out);
/* One part of a file, for several readers (GPUs, ranks) that take one file between them.  A PLAIN text file: part `rank` is the
 * byte range from the first record start that can be recognised in the text behind byte size / world * rank (a line that opens a
 * record, a newline in front of it) to the next part's; every call of the reader works on such a part.  A BGZF file: part `rank` holds the
 * members from the first member boundary at or behind byte size / world * rank to the next part's, and delivers the records that
 * BEGIN in its text — from the first record start that can be recognised there (a line that opens a record, a newline in front of
 * it inside the part's text) to the place where the next part's reader finds its own, found by inflating into the next part.
 * The parts' records, in rank order, are the file's records; the readers do not talk to each other.
 * RESTRICTION: a FASTQ record start is recognised as a line that opens with '@' and whose second-next line opens with '+', i.e.
 * FOUR-LINE FASTQ (one sequence line, one quality line per record: what sequencers and every current tool write).  Multi-line
 * FASTQ — which the whole-file readers accept, as kseq does — must be read with world = 1: in parts, its boundaries may not be
 * found (records then fall to an earlier part) or, rarely, a quality line may be taken for a header.  FASTA has no such limit.
 * Device batches only
 * (bl_reader_next_batch_device) from a BGZF part.  BL_ERR_INVALID for a gzip file that is not BGZF (one stream, no entry points).  This is how north_star's "shard by read" reaches the
 * file: the reference's drivers read one file per process (tests/test_kmer_view.cpp:23-42). */
int bl_reader_open_shard(const char* path, uint32_t rank, uint32_t world, bl_reader** out);
/* the bytes of the file whose members are the reader's own: [first_byte, end_byte), end_byte = UINT64_MAX for the last part
 * (0 / UINT64_MAX for a reader of the whole file) */
int bl_reader_shard_range(bl_reader* reader, uint64_t* first_byte, uint64_t* end_byte);
int bl_reader_close(bl_reader* reader);
/* "plain", "gzip" or "bgzf": how the file is being read */
const char* bl_reader_kind(bl_reader* reader);
/* Host only: next record.  Returns BL_OK, 1 at end of file, or an error.  Pointers stay valid until the next call. */
int bl_reader_next_record(bl_reader* reader, const char** name, const char** seq, uint64_t* seq_len);
/* Next batch of whole records holding at most max_bases bases (0 = the rest of the file; always at least one
 * record), uploaded to the device.  At end of file *out is NULL and *n_seqs is 0.  A thread of the reader parses the following
 * batch meanwhile; max_bases is fixed by the first call (BL_ERR_INVALID if a later call differs), and single records
 * (bl_reader_next_record) cannot be mixed with batches on one reader. */
int bl_reader_next_batch(bl_ctx* ctx, bl_reader* reader, uint64_t max_bases, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases);
/* The high-throughput path for regular files: the decompressed TEXT, cut at record boundaries (4-line FASTQ: in front of a header
 * line, recognised by the '+' line two lines on; FASTA: before a line-initial '>'), goes to the device-side parser
 * (bl_batch_from_text) — parallel inflate, one H2D copy, parsing on the GPU.  A thread of the reader assembles the next spans
 * while the caller works on the current one, in buffers that live with the reader (page-locked for bl_reader_next_batch_device).
 * bl_reader_next_text hands out the next span of at most max_bytes (0 = 64 MiB; a longer single record is not split), valid
 * until the next call, 1 at end of file; max_bytes is fixed by the first call (BL_ERR_INVALID if a later call differs).
 * bl_reader_next_batch_device parses the span on the device (names are not kept; *out NULL at end of file; BL_ERR_INVALID for
 * layouts the device parser refuses — reopen and use the record calls).  Records and spans cannot be mixed on one reader. */
int bl_reader_next_text(bl_reader* reader, uint64_t max_bytes, const char** text, uint64_t* n_bytes);
int bl_reader_next_batch_device(bl_ctx* ctx, bl_reader* reader, uint64_t max_text_bytes, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases);
/* Host copy of the batch produced last: concatenated bases, offsets[n_seqs+1], names. */
int bl_reader_last_batch(bl_reader* reader, const char** bases, const uint64_t** offsets, uint64_t* n_seqs);
const char* bl_reader_last_name(bl_reader* reader, uint64_t i);

/* ---- set operations on k-mer lists (the consumer in reference tests/test_jaccard.cpp:55-130) ---
 * bl_sort_unique_u64: sorts d_keys[0..n) in place and removes duplicates (ordered_unique_sampler.hpp:115-130 over a
 * sorted vector); *n_unique = number of distinct keys now at the front.  Synchronous.
 * bl_jaccard_sorted_u64: |A n B| and |A u B| of two sorted duplicate-free device arrays (jaccard.hpp:8-37). */
int bl_sort_unique_u64(bl_ctx* ctx, uint64_t* d_keys, uint64_t n, uint64_t* n_unique);
int bl_jaccard_sorted_u64(bl_ctx* ctx, const uint64_t* d_a, uint64_t na, const uint64_t* d_b, uint64_t nb, uint64_t* intersection,
                          uint64_t* union_size);

/* ---- multi-GPU k-mer counting pieces ------------------------------------------------------
 * bl_partition_u64: reorder keys into `parts` (<= 64) buckets by hash64(key, seed) % parts — the owner rank of a
 *   k-mer in a partitioned count; counts[b] (host) = size of bucket b, buckets are contiguous in d_out in bucket order.
 * bl_sort_u64: in-place ascending sort (duplicates kept).  bl_count_sorted_u64: run-length count of a sorted list.
 * The exchange between the two (all-to-all over RCCL / xGMI) is biolib_amd/shard.py:exchange_and_count. */
int bl_partition_u64(bl_ctx* ctx, const uint64_t* d_keys, uint64_t n, uint32_t parts, uint64_t seed, uint64_t* d_out, uint64_t* counts);
int bl_sort_u64(bl_ctx* ctx, uint64_t* d_keys, uint64_t n);
int bl_count_sorted_u64(bl_ctx* ctx, const uint64_t* d_sorted, uint64_t n, uint64_t* d_unique, uint32_t* d_counts, uint64_t* n_unique);

/* Count reduction across the GPUs of one node: ctxs[g] is the context of device g (all distinct devices),
 * counters holds n_gpu rows of n 64-bit counters — row g = GPU g's local counts in, the column sums out (in every row).  One
 * ncclAllReduce(sum, uint64) per GPU over RCCL / xGMI, called on RCCL's C API directly (librccl.so.1 is loaded on first use);
 * communicators are created once per device set and cached.  XOR digests have no RCCL reduction: fold them on the host. */
int bl_count_allreduce(bl_ctx* const* ctxs, int n_gpu, uint64_t* counters, int n);

/* Super-k-mer bucket exchange (record of reference super_kmer_view.hpp:20-24 made self-contained).
 * bl_pack_super_kmers: one 16-byte record per group of bl_scan_super_kmers — d_records[2g] = bases 0..31 of the group's
 *   size + k - 1 bases (2 bits each, first base most significant), d_records[2g+1] = bases 32.. in bits 63..10, mm_pos in
 *   bits 9..5 (d_mm_pos is REQUIRED: bl_count_super_kmers finds a record's minimizer through it), size - 1 in bits 4..0.
 *   Needs 2k - m <= 59.  d_first_pos holds what the scan of THIS batch reported: positions in the caller's whole when the batch
 *   has an origin (bl_batch_set_origin); a position in front of the origin or beyond the batch packs an empty record.
 * bl_partition_records: reorder 16-byte records into `parts` (<= 64) contiguous buckets by d_hashes[g] % parts (the
 *   minimizer hash the scan returned = the owner rank); counts[b] (host) = records in bucket b.
 * bl_expand_super_kmers: records -> their k-mers (canonical with BL_FLAG_CANONICAL), group after group;
 *   BL_ERR_CAPACITY with *n_kmers = need when d_kmers is too small. */
int bl_pack_super_kmers(bl_ctx* ctx, const bl_batch* batch, const uint64_t* d_first_pos, const uint8_t* d_sizes, const uint8_t* d_mm_pos, uint64_t n_groups,
                        uint32_t k, uint32_t m, uint64_t* d_records);
/* bl_scan_super_kmer_records: bl_scan_super_kmers + bl_pack_super_kmers in one scan — the groups leave the scan as packed records
 *   (d_records[2r], d_records[2r+1] as above, with mm_pos) beside the hashes of their minimizers (d_hashes[r]: the owner of the
 *   record), built from the 2-bit codes the scan holds anyway: no position / size arrays written and read back, no second pass
 *   over the bases.  Needs 2k - m <= 59.  result as bl_scan_super_kmers. */
int bl_scan_super_kmer_records(bl_ctx* ctx, const bl_batch* batch, uint64_t first, uint64_t n, uint32_t k, uint32_t m, uint64_t seed, uint32_t flags,
                               uint64_t* d_records, uint64_t* d_hashes, uint64_t capacity, bl_result* result);
/* bl_count_super_kmers: the exact multiplicity of every (canonical) k-mer of the packed records WITHOUT a global sort: records are
 *   grouped by the hash of their minimizer (found again through mm_pos; `m`, `seed` and the canonical flag as given to the scan)
 *   into buckets of a few thousand k-mers, and every bucket is expanded and counted in an LDS hash table by one workgroup (buckets
 *   that do not fit take a sort + run-length path).  All occurrences of a canonical k-mer share their minimizer, so they meet in
 *   one bucket and the count is exact.  d_kmers / d_counts receive the distinct k-mers and their multiplicities in NO particular
 *   order; BL_ERR_CAPACITY with *n_distinct = need when they are too small.  A bucket's table is filled in rounds, so what bounds a
 *   bucket is its DISTINCT k-mers, not their occurrences: reads of high coverage (every minimizer many times over) stay in the tables. */
int bl_count_super_kmers(bl_ctx* ctx, const uint64_t* d_records, uint64_t n_groups, uint32_t k, uint32_t m, uint64_t seed, uint32_t flags, uint64_t* d_kmers,
                         uint32_t* d_counts, uint64_t capacity, uint64_t* n_distinct);
int bl_partition_records(bl_ctx* ctx, const uint64_t* d_hashes, const uint64_t* d_records, uint64_t n, uint32_t parts, uint64_t* d_out, uint64_t* counts);
int bl_expand_super_kmers(bl_ctx* ctx, const uint64_t* d_records, uint64_t n_groups, uint32_t k, uint32_t flags, uint64_t* d_kmers, uint64_t capacity,
                          uint64_t* n_kmers);

/* ---- BGZF on the device (ingest) -----------------------------------------------------------------------
 * A bgzip'ed file is a chain of gzip members of at most 64 KiB of text each, every one an independent deflate stream: the
 * device inflates a span's members side by side (one wavefront each) and checks their CRC-32, so that the file crosses PCIe
 * compressed.  bl_reader_next_batch_device does this for BGZF input by itself; the two calls below are its building blocks.
 * The reference reads .gz through zlib's gzread (tests/kseq.h:41-42 KSEQ_INIT(gzFile, gzread), tests/test_kmer_view.cpp:23-27):
 * same text, produced here. */
typedef struct {
    uint64_t src_off;  /* the member's deflate data inside the packed buffer */
    uint64_t dst_off;  /* where its text goes inside the text buffer */
    uint32_t src_len;  /* bytes of deflate data */
    uint32_t isize;    /* bytes of text (gzip ISIZE), <= 65536 */
    uint32_t crc;      /* CRC-32 of the text (gzip trailer) */
    uint32_t reserved;
} bl_bgzf_member;
/* Host: walk the member headers in bytes[0 .. n_bytes) and fill members[] (at most `capacity`) with offsets src_base + ... /
 * dst_base + ...; stops in front of a member that is not completely inside the bytes.  *consumed = bytes walked, *text_bytes =
 * the sum of the members' text sizes.  BL_ERR_INVALID for anything that is not a BGZF member header. */
int bl_bgzf_walk(const void* bytes, uint64_t n_bytes, uint64_t src_base, uint64_t dst_base, bl_bgzf_member* members, uint64_t capacity, uint64_t* n_members,
                 uint64_t* consumed, uint64_t* text_bytes);
/* Device, asynchronous on the context's stream: inflate the members described by d_members[0 .. n_members) from d_packed
 * (4-byte aligned, packed_bytes long, its allocation a whole number of dwords) into d_text (text_bytes long).  d_status[i] becomes 0 for a sound member, otherwise a
 * non-zero code (damaged deflate data, text size or CRC-32 that differ from the trailer); a table entry that points outside the
 * two buffers is refused, not followed, and nothing outside a member's own [dst_off, dst_off + isize) is ever written. */
int bl_bgzf_inflate(bl_ctx* ctx, const void* d_packed, uint64_t packed_bytes, const bl_bgzf_member* d_members, uint64_t n_members, void* d_text,
                    uint64_t text_bytes, uint32_t* d_status);

/* Measurement helper: sustained HBM rates of THIS device — a read-only streaming kernel over n_bytes and a
 * device-to-device copy (read + write bytes counted), `iters` repetitions each; bench.py reports them next to the 8 TB/s spec. */
int bl_probe_hbm(bl_ctx* ctx, uint64_t n_bytes, int iters, double* read_gbps, double* copy_gbps);

/* Shader clock held while other work runs: bl_clock_probe_start launches one sleeping wave on its own stream that stamps the
 * shader-cycle and the 100 MHz real-time counters until bl_clock_probe_finish stops it (at the latest duration_ms later, 1..5000);
 * finish returns cycles / time in GHz and releases the probe.  Finish it BEFORE any device-wide synchronise.  Used by bench.py to price the VALU ceiling at the clock of the run. */
typedef struct bl_clock_probe bl_clock_probe;
int bl_clock_probe_start(bl_ctx* ctx, uint32_t duration_ms, bl_clock_probe** out);
int bl_clock_probe_finish(bl_clock_probe* probe, double* shader_ghz);

/* Device-side parser: copy raw FASTA / FASTQ TEXT (already in host memory, e.g. a read()/mmap of the file) to the GPU
 * and build the batch there: newline index, line classification, prefix sums, gather of the sequence lines.  Same
 * sequences as bl_reader_* for the regular layouts it accepts — FASTQ with exactly 4 lines per record, FASTA with any
 * line wrapping, LF or CRLF — and BL_ERR_INVALID for anything else (never a silent mis-parse).  Names are not kept. */
int bl_batch_from_text(bl_ctx* ctx, const char* text, uint64_t n_bytes, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases);

/* ---- spill / wire formats read by biolib's consumers -------------------------------------
 * bl_write_run_u64: a run file of emem::external_memory_vector<uint64_t> (external_memory_vector.hpp:243-262): the
 *   SORTED keys as raw little-endian 8-byte values, no header.  bl_run_file_name builds the reference's file name
 *   <dir>/tmp.run[_<name>]_<id>.bin (:253-262).
 * bl_write_vector_u64: io::basic_store(std::vector<uint64_t>) (io.hpp:104-112): size_t count, then the elements.
 * Both copy the device array to the host and write synchronously. */
int bl_run_file_name(const char* dir, const char* name, uint64_t id, char* out, uint64_t out_len);
int bl_write_run_u64(bl_ctx* ctx, const uint64_t* d_sorted_keys, uint64_t n, const char* path);
int bl_write_vector_u64(bl_ctx* ctx, const uint64_t* d_keys, uint64_t n, const char* path);
/* The read side (reference io::basic_load io.hpp:114-122; external_memory_vector::const_iterator :265-347, a k-way merge over
 * the run files): files biolib wrote feed the device set operations.
 * bl_file_count_u64: elements in a run file (with_count = 0) or a basic_store'd vector (with_count = 1, checked against the size).
 * bl_read_file_u64_host / bl_read_file_u64: its elements into a host / device array of `capacity` elements.
 * bl_merge_runs_u64: the sorted union (duplicates kept) of n_paths run files in one device array — what iterating the
 *   reference's external_memory_vector yields; BL_ERR_CAPACITY with *n_total = need when d_out is too small. */
int bl_file_count_u64(const char* path, int with_count, uint64_t* n);
int bl_read_file_u64_host(const char* path, int with_count, uint64_t* out, uint64_t capacity, uint64_t* n);
int bl_read_file_u64(bl_ctx* ctx, const char* path, int with_count, uint64_t* d_out, uint64_t capacity, uint64_t* n);
int bl_merge_runs_u64(bl_ctx* ctx, const char* const* paths, uint32_t n_paths, uint64_t* d_out, uint64_t capacity, uint64_t* n_total);

/* ---- device memory helpers (for callers without their own allocator) ----------------------------- */
int bl_device_alloc(bl_ctx* ctx, uint64_t bytes, void** d_ptr);
int bl_device_free(bl_ctx* ctx, void* d_ptr);
/* Page-locked host memory (copies to and from it are single DMA transfers; its pages are mapped once, at allocation): what a
 * caller that downloads results batch after batch should copy into (include/compat/read_pool.hpp does). */
int bl_host_alloc(bl_ctx* ctx, uint64_t bytes, void** ptr);
int bl_host_free(bl_ctx* ctx, void* ptr);
int bl_copy_to_host(bl_ctx* ctx, void* dst, const void* d_src, uint64_t bytes); /* synchronous */
int bl_copy_to_device(bl_ctx* ctx, void* d_dst, const void* src, uint64_t bytes); /* synchronous */

/* ---- host-side scalar helper (bit-exact with the device hash) ------------------------------------- */
uint64_t bl_hash64_u64(uint64_t value, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* BIOLIB_AMD_H */
