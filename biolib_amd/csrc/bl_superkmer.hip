// Packed super-k-mer records for the multi-GPU bucket exchange.
//
// A super-k-mer (reference record: include/super_kmer_view.hpp:20-24 — minimizer, mm_pos, size) stands for `size`
// consecutive k-mers = size + k - 1 bases.  What travels between GPUs is its sequence, 2 bits per base, in a fixed
// 16-byte record, routed by the hash of its minimizer — every occurrence of a canonical k-mer has the same minimizer
// value, hence the same owner, so the owner can count k-mers exactly with no further exchange:
//
//   rec[0]  bases 0..31, first base in the most significant pair (kmer_view.hpp:194 packing)
//   rec[1]  bases 32..58 in bits 63..10 (same order), bits 9..5 = mm_pos (offset of the minimizer in the first k-mer,
//           super_kmer_view.hpp:132), bits 4..0 = size - 1 (number of k-mers, 1 .. k-m+1)
//
// so size + k - 1 <= 59 bases (k = 31, m = 15: at most 47).  8 B/k-mer become ~1.8 B/base on the links.  mm_pos makes a record
// self-describing: its owner can find the minimizer again (one m-mer extraction + one hash) and bucket the record by it,
// which is what bl_count_super_kmers does to count k-mers in LDS-sized buckets instead of sorting them all.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <vector>

#include "../../include/biolib_amd.h"
#include "bl_partition.hpp"
#include "bl_scan_core.hpp"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);
extern int bl_ctx_device(bl_ctx* ctx);
extern void* bl_ctx_scratch(bl_ctx* ctx, int slot, size_t bytes);  // bl_capi.hip: device scratch that lives with the context

namespace {

constexpr int MAX_PARTS = 64;
constexpr int MAX_BASES = 59;

#define SK_HIP(call)                                                                                                             \
    do {                                                                                                                         \
        hipError_t e_ = (call);                                                                                                  \
        if (e_ != hipSuccess) return bl_set_error(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e_));    \
    } while (0)

__device__ __forceinline__ unsigned code_of(unsigned char c) { return ((c >> 1) ^ (c >> 2)) & 3u; }  // A0 C1 G2 T/U3 (constants.hpp:12-21)

__device__ __forceinline__ unsigned base_at(unsigned long long hi, unsigned long long lo, int i)
{
    return i < 32 ? (unsigned)(hi >> (62 - 2 * i)) & 3u : (unsigned)(lo >> (62 - 2 * (i - 32))) & 3u;
}

// 8 bases at `at` as 16 bits of 2-bit codes, the first base in the two most significant bits: one (unaligned) 8-byte load and
// a dozen word operations instead of 8 byte loads (the records of neighbouring groups overlap by k - 1 bases: the loads hit in L1)
struct __attribute__((packed)) Unaligned8 {
    unsigned long long v;
};
__device__ __forceinline__ unsigned long long codes8(const unsigned char* __restrict__ bases, unsigned long long at, unsigned long long n_bases)
{
    unsigned long long w = 0;
    if (at + 8 <= n_bases) {
        w = reinterpret_cast<const Unaligned8*>(bases + at)->v;
    } else {  // the last bytes of the batch (a borrowed buffer has nothing behind them that may be read)
        for (int i = 0; i < 8; ++i)
            if (at + i < n_bases) w |= (unsigned long long)bases[at + i] << (8 * i);
    }
    w = __builtin_bswap64(w);                                            // first base in the most significant byte
    unsigned long long x = ((w >> 1) ^ (w >> 2)) & 0x0303030303030303ULL;  // code_of() of all eight bytes
    x = (x | (x >> 6)) & 0x000f000f000f000fULL;
    x = (x | (x >> 12)) & 0x000000ff000000ffULL;
    x = (x | (x >> 24)) & 0xffffULL;
    return x;
}

__global__ __launch_bounds__(256) void pack_kernel(const unsigned char* __restrict__ bases, unsigned long long n_bases,
                                                   const unsigned long long* __restrict__ first_pos, const unsigned char* __restrict__ sizes,
                                                   const unsigned char* __restrict__ mm_pos, unsigned long long n, int k, ulonglong2* __restrict__ out,
                                                   unsigned long long origin)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    // the scan reported origin + the position inside the batch (bl_batch_set_origin); a position in front of the origin wraps to a
    // huge value and takes the "beyond the batch" exit below
    const unsigned long long p = first_pos[g] - origin;
    const int size = sizes[g];
    int nb = size + k - 1;
    if (p + (unsigned long long)nb > n_bases) nb = p < n_bases ? (int)(n_bases - p) : 0;  // never read past the batch (a caller error; the record is then short)
    unsigned long long hi = 0, lo = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (8 * j < nb) hi |= codes8(bases, p + 8 * j, n_bases) << (48 - 16 * j);
#pragma unroll
    for (int j = 4; j < 8; ++j)
        if (8 * j < nb) lo |= codes8(bases, p + 8 * j, n_bases) << (48 - 16 * (j - 4));
    // bases beyond the record's own are not part of it
    if (nb < 32) hi &= nb ? ~0ULL << (64 - 2 * nb) : 0ULL;
    if (nb > 32 && nb < 64) lo &= ~0ULL << (64 - 2 * (nb - 32));
    if (nb <= 32) lo = 0;
    const unsigned long long mp = mm_pos ? (unsigned long long)(mm_pos[g] & 31u) : 0ULL;
    out[g] = make_ulonglong2(hi, (lo & ~0x3ffULL) | (mp << 5) | (unsigned long long)((size - 1) & 31));
}

struct HashArrayOwner {
    const unsigned long long* hashes;
    __device__ uint32_t operator()(unsigned long long i, uint32_t parts) const { return blpart::bucket_of(hashes[i], parts); }
};

__global__ void sizes_kernel(const ulonglong2* recs, unsigned long long n, unsigned long long* sizes)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) sizes[g] = (recs[g].y & 31ULL) + 1;
}

__global__ __launch_bounds__(256) void expand_kernel(const ulonglong2* __restrict__ recs, const unsigned long long* __restrict__ offsets,
                                                     unsigned long long n, int k, int canonical, unsigned long long* __restrict__ out)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const unsigned long long hi = recs[g].x, lo = recs[g].y;
    const int size = (int)(lo & 31ULL) + 1;
    const unsigned long long mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int shift = 2 * (k - 1);
    unsigned long long fwd = 0, rc = 0;
    for (int i = 0; i < k - 1; ++i) {  // kmer_view.hpp:190-199, started from the packed codes
        const unsigned long long c = base_at(hi, lo, i);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3ULL ^ c) << shift);
    }
    unsigned long long* dst = out + offsets[g];
    for (int j = 0; j < size; ++j) {
        const unsigned long long c = base_at(hi, lo, k - 1 + j);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3ULL ^ c) << shift);
        dst[j] = canonical ? (fwd < rc ? fwd : rc) : fwd;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Counting without a global sort of the k-mers (bl_count_super_kmers).
//   1. bucket_id_kernel   per record: the minimizer m-mer (at mm_pos of the first k-mer), canonical as the scan took it, hashed as
//                         the scan hashed it; bucket = the top 32 bits of a multiplicative remix of that hash (independent of
//                         the low bits used to pick the owner rank) scaled to the number of buckets, which is whatever
//                         makes a bucket ~36 records: the per-bucket costs are spread over as many keys for every input size
//   2. rocprim::radix_sort_pairs (bucket id -> the 16-byte record itself): library plumbing, <= 26 key bits; the records of
//                         a bucket end up contiguous
//   3. bucket_starts_kernel  where each bucket begins in the sorted order (one thread per bucket, independent binary searches)
//   4. count_buckets_kernel<false>  one WAVE per bucket (grid-stride over waves, 13 KB of LDS per wave, no workgroup barrier):
//                         the bucket's k-mers go into an LDS hash table (1024 slots, linear probing, 64-bit compare-and-swap on
//                         the key, 32-bit add on the count), in rounds of up to 64 records / CT_CAP k-mers; only the number of
//                         distinct k-mers is kept.  A bucket that may come to hold more than CT_FULL DISTINCT k-mers (or more
//                         than CT_MAXREC records) is listed for the fallback instead.
//   5. exclusive scan of those numbers = where every bucket writes
//   6. count_buckets_kernel<true>   the tables are built again and written out at the buckets' offsets.  Two table passes
//                         instead of one returning atomic per bucket on a single output cursor (~88 M/s on this chip: 48 ms for
//                         4 M buckets); the output order is deterministic as a bonus.
//   7. fallback           the records of listed buckets: gather -> expand -> radix sort -> run-length encode -> appended
constexpr int CT_SLOTS = 1024;  // table slots of ONE WAVE
constexpr int CT_CAP = 700;     // k-mers one ROUND of a bucket inserts (the work list's size)
constexpr int CT_FULL = 820;    // DISTINCT k-mers a bucket's table may come to hold (load 0.8): beyond it the fallback counts the bucket
constexpr int CT_RECS = 64;     // records a round stages (one per lane)
constexpr int CT_MAXREC = 2040; // records a bucket may hold at all: 2040 x 32 k-mers < 2^16, the width of a count
constexpr int CT_WIDE = 2;      // k-mers a lane inserts per round
constexpr int CT_WAVES = 2;     // waves per workgroup (12.4 KB of LDS per wave: 6 workgroups = 12 waves per CU)
static_assert(CT_MAXREC * 32 < 65536, "counts are kept in 16 bits");
constexpr unsigned long long CT_EMPTY = ~0ULL;  // never a k-mer for k < 32, nor a canonical 32-mer (its reverse complement is 0)

__device__ __forceinline__ unsigned long long mmer_at(unsigned long long hi, unsigned long long lo, int pos, int m)
{
    // m bases from base `pos` (< 32) of the record: the top 2m bits of (hi:lo) << 2 pos
    const unsigned long long lob = lo & ~0x3ffULL;
    const unsigned long long top = pos == 0 ? hi : (hi << (2 * pos)) | (lob >> (64 - 2 * pos));
    return top >> (64 - 2 * m);
}

__global__ __launch_bounds__(256) void bucket_id_kernel(const ulonglong2* __restrict__ recs, unsigned long long n, int m, int canonical, uint32_t seed,
                                                        uint32_t n_buckets, uint32_t* __restrict__ ids)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const ulonglong2 r = recs[g];
    const int mp = (int)((r.y >> 5) & 31ULL);
    unsigned long long v = mmer_at(r.x, r.y, mp, m);
    if (canonical) {
        unsigned long long rc = bl::pairrev64(v) >> (64 - 2 * m);
        rc ^= m == 32 ? ~0ULL : ((1ULL << (2 * m)) - 1);
        v = rc < v ? rc : v;
    }
    const unsigned long long h = bl::murmur64(v, seed);
    // 32 well-mixed bits of the hash, scaled to [0, n_buckets): any number of buckets, not only powers of two
    ids[g] = (uint32_t)((((h * 0x9E3779B97F4A7C15ULL) >> 32) * (unsigned long long)n_buckets) >> 32);
}

// starts[b] = first position of the sorted ids holding a value >= b, for b = 0 .. n_buckets
__global__ __launch_bounds__(256) void bucket_starts_kernel(const uint32_t* __restrict__ ids, uint32_t n, uint32_t n_buckets, uint32_t* __restrict__ starts)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < b) lo = mid + 1;
        else hi = mid;
    }
    starts[b] = lo;
}

// diagnostic build (-DBL_COUNT_STAMPS, tools/count_stamps.py): shader cycles per phase, accumulated per wave in registers
#ifdef BL_COUNT_STAMPS
__device__ unsigned long long bl_count_stamps[8];
#define STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[i] += now_ - stamp_t_; stamp_t_ = now_; } while (0)
#define STAMP_ARGS , unsigned long long* acc, unsigned long long& stamp_t_
#define STAMP_PASS , acc, stamp_t_
#else
#define STAMP(i) do {} while (0)
#define STAMP_ARGS
#define STAMP_PASS
#endif

struct WaveTable {
    unsigned long long keys[CT_SLOTS];
    unsigned int cnt[CT_SLOTS / 2];  // 16 bits per slot (a bucket holds at most CT_CAP k-mers): slot h is half h & 1 of word h >> 1
    ulonglong2 recs[CT_RECS];
    unsigned short work[CT_CAP + 4];  // k-mer j of the bucket = k-mer (entry & 31) of record (entry >> 5)
};

// LDS traffic of one wave is served in issue order, so lanes of a wave see each other's LDS writes as soon as the compiler
// keeps the program order: a wavefront-scope fence does that (no s_barrier, no other wave involved)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t table_slot(unsigned long long key) { return (uint32_t)((key * 0xD6E8FEB86659FD93ULL) >> (64 - 10)); }

// one probe: what slot h held (CT_EMPTY: the key sits there now; the key itself: it sat there already)
__device__ __forceinline__ unsigned long long table_probe(WaveTable& t, uint32_t h, unsigned long long key)
{
    return atomicCAS(&t.keys[h], CT_EMPTY, key);
}

// W keys at once (live[i] = false: key i is not there): all first probes are issued before any result is looked at, so their
// LDS round trips overlap; the few that collide (load <= 0.68) walk on one by one.  The table never fills.
// Returns how many of the lane's keys were NEW to the table.
template <int W>
__device__ __forceinline__ unsigned int table_insert_many(WaveTable& t, const unsigned long long (&key)[W], const bool (&live)[W])
{
    uint32_t h[W];
    unsigned long long old[W];
#pragma unroll
    for (int i = 0; i < W; ++i) h[i] = table_slot(key[i]);
#pragma unroll
    for (int i = 0; i < W; ++i) old[i] = live[i] ? atomicCAS(&t.keys[h[i]], CT_EMPTY, key[i]) : key[i];
    unsigned int fresh = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        while (old[i] != CT_EMPTY && old[i] != key[i]) {
            h[i] = (h[i] + 1) & (CT_SLOTS - 1);
            old[i] = table_probe(t, h[i], key[i]);
        }
        fresh += (live[i] && old[i] == CT_EMPTY) ? 1u : 0u;
    }
#pragma unroll
    for (int i = 0; i < W; ++i)
        if (live[i]) atomicAdd(&t.cnt[h[i] >> 1], 1u << (16 * (h[i] & 1u)));
    return fresh;
}

__device__ __forceinline__ unsigned long long kmer_of(ulonglong2 r, int q, int k, int canonical, unsigned long long kmask)
{
    // bases [q, q + k) of the record (q <= 31): the top 2k bits of (hi:lo) << 2q
    const unsigned long long lob = r.y & ~0x3ffULL;
    const unsigned long long top = q == 0 ? r.x : (r.x << (2 * q)) | (lob >> (64 - 2 * q));
    const unsigned long long fwd = top >> (64 - 2 * k);
    if (!canonical) return fwd;
    const unsigned long long rc = (bl::pairrev64(fwd) >> (64 - 2 * k)) ^ kmask;
    return rc < fwd ? rc : fwd;  // kmer_view.hpp:196
}

__device__ __forceinline__ unsigned int wave_incl_scan(unsigned int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ void list_overflow(int lane, uint32_t lo, uint32_t hi, unsigned long long* cursor, uint2* overflow, uint32_t max_overflow)
{
    if (lane == 0) {
        const unsigned long long slot = atomicAdd(&cursor[1], 1ULL);
        if (slot < max_overflow) overflow[slot] = make_uint2(lo, hi);
    }
}

// Where the counted k-mers go when they are written in the same pass that counts them (no second pass over the buckets to learn
// the offsets first): every wave fills chunks of CT_CHUNK output slots that it takes from one global cursor, a bucket's k-mers
// going to the rest of the wave's current chunk and on into a fresh one, so that only each wave's LAST chunk is left partly
// empty.  Those holes (one per wave, noted at the end of the kernel) are filled afterwards from the end of the array
// (fill_holes_kernel).  Positions at or beyond the caller's capacity land in a side buffer of one chunk per wave: that is how far
// the holes can push the end of the array beyond the number of distinct k-mers.
constexpr unsigned int CT_CHUNK = 4096;
struct CountedOut {
    unsigned long long* keys;
    unsigned int* counts;
    unsigned long long capacity;
    unsigned long long* ext_keys;
    unsigned int* ext_counts;
    unsigned long long ext_n;
    unsigned long long* cursor;  // [0] next free chunk, [1] buckets listed for the fallback, [2] distinct k-mers counted
    ulonglong2* holes;           // per wave: [first unused slot, end) of its last chunk
};
__device__ __forceinline__ void put_counted(const CountedOut& o, unsigned long long pos, unsigned long long key, unsigned int c)
{
    if (pos < o.capacity) {
        o.keys[pos] = key;
        o.counts[pos] = c;
    } else if (pos - o.capacity < o.ext_n) {
        o.ext_keys[pos - o.capacity] = key;
        o.ext_counts[pos - o.capacity] = c;
    }
}
struct WaveChunk {
    unsigned long long at = 0, end = 0;  // wave-uniform
    unsigned long long distinct = 0;
};

// one bucket, one wave: records [lo, hi) of the sorted order; `rec` = this lane's record of it (lane l holds record lo + l; lanes
// beyond the bucket hold something else and do not use it).
// A bucket is counted in ROUNDS: a round stages up to 64 records — as many as have <= CT_CAP k-mers between them —, expands them
// through the work list and inserts their k-mers; the table and its counts live on from round to round.  What bounds a bucket
// is therefore the number of DISTINCT k-mers it holds (CT_FULL), not the number of occurrences: on reads of high coverage every
// minimizer comes 30 times over with the same k-mers behind it, a bucket of the usual ~36 records holds a handful of distinct
// minimizers, and its hundred records are three rounds into a table that stays nearly empty (before: straight to the sort).
template <bool WRITE, typename RecPtr>
__device__ __forceinline__ void count_one_bucket(WaveTable& t, int lane, uint32_t bucket, uint32_t lo, uint32_t hi, ulonglong2 rec, RecPtr grecs, uint32_t last_r, int k,
                                                 int canonical, unsigned long long kmask, unsigned int* __restrict__ distinct, const CountedOut& out, WaveChunk& chunk,
                                                 unsigned long long* cursor, uint2* __restrict__ overflow, uint32_t max_overflow STAMP_ARGS)
{
    STAMP(0);  // everything between two buckets: loop control, the waits for this bucket's records and range
    const uint32_t n_rec = hi - lo;  // wave-uniform
    if (n_rec == 0) {
        if (!WRITE && lane == 0) distinct[bucket] = 0;
        return;
    }
    if (n_rec > CT_MAXREC) {  // more occurrences than a 16-bit count holds: the fallback counts this bucket
        list_overflow(lane, lo, hi, cursor, overflow, max_overflow);
        if (!WRITE && lane == 0) distinct[bucket] = 0;
        return;
    }
    wave_lds_sync();  // the previous bucket's table has been read out
#pragma unroll
    for (int i = 0; i < CT_SLOTS / 64; ++i) {
        t.keys[i * 64 + lane] = CT_EMPTY;
        if (i < CT_SLOTS / 128) t.cnt[i * 64 + lane] = 0;
    }
    STAMP(1);  // table cleared
    unsigned int held = 0;  // distinct k-mers in the table (wave-uniform)
    for (uint32_t at = lo; at < hi;) {
        if (at != lo) {  // later rounds fetch their records themselves (the first round's came with the software pipeline)
            const uint32_t c = at + lane < last_r ? at + lane : last_r;
            rec.x = grecs[2ull * c];
            rec.y = grecs[2ull * c + 1];
        }
        const unsigned int size = at + (uint32_t)lane < hi ? (unsigned int)(rec.y & 31ULL) + 1 : 0u;
        const unsigned int incl = wave_incl_scan(size, lane);
        // the records of this round: the longest prefix whose k-mers fit the work list (sizes are <= 32: at least 21 records)
        const unsigned long long fits = __ballot(size != 0 && incl <= (unsigned int)CT_CAP);
        const int n_take = (int)__popcll(fits);  // wave-uniform; a prefix of the lanes, since incl only grows
        const unsigned int total = __shfl(incl, n_take - 1, 64);
        STAMP(2);  // size prefix
        if (held + total > (unsigned int)CT_FULL) {  // the table might fill up (every k-mer of the round may be new): the fallback's
            list_overflow(lane, lo, hi, cursor, overflow, max_overflow);
            if (!WRITE && lane == 0) distinct[bucket] = 0;
            return;
        }
        if (at != lo) wave_lds_sync();  // the round before has read its records and its work list
        // work list: k-mer j of the round -> (record, index inside it), written by the record's own lane at its prefix (fire and
        // forget LDS stores, no search later)
        if (lane < n_take) {
            t.recs[lane] = rec;
            const unsigned int first = incl - size;
            for (unsigned int q = 0; q < size; ++q) t.work[first + q] = (unsigned short)((lane << 5) | q);
        }
        wave_lds_sync();
        // CT_WIDE k-mers per lane and round, their table probes in flight together: the rounds are chains of LDS round trips (work
        // list -> record -> probe -> count) with little to overlap them with, so fewer, wider rounds it is
        unsigned int fresh = 0;
        for (unsigned int j = lane; j < total; j += 64 * CT_WIDE) {
            unsigned long long key[CT_WIDE];
            bool live[CT_WIDE];
            unsigned int w[CT_WIDE];
#pragma unroll
            for (int i = 0; i < CT_WIDE; ++i) {
                live[i] = j + 64 * i < total;
                w[i] = t.work[live[i] ? j + 64 * i : j];
            }
#pragma unroll
            for (int i = 0; i < CT_WIDE; ++i) key[i] = kmer_of(t.recs[w[i] >> 5], (int)(w[i] & 31u), k, canonical, kmask);
            fresh += table_insert_many<CT_WIDE>(t, key, live);
        }
        at += (uint32_t)n_take;
        if (at < hi) held += __shfl(wave_incl_scan(fresh, lane), 63, 64);  // (only buckets of several rounds pay for this sum)
    }
    wave_lds_sync();
    STAMP(3);  // k-mers inserted
    // occupied slots, 64 consecutive ones per round (lane l looks at slot 64 i + l: no bank conflict), ranked by ballot: the
    // bucket's k-mers leave in slot order and the lanes of a round write NEIGHBOURING output positions
    unsigned int cnt[CT_SLOTS / 64];
    unsigned long long key[CT_SLOTS / 64];
    unsigned long long occupied[CT_SLOTS / 64];
    unsigned int d = 0;
#pragma unroll
    for (int i = 0; i < CT_SLOTS / 64; ++i) {
        cnt[i] = (t.cnt[(i * 64 + lane) >> 1] >> (16 * (lane & 1))) & 0xffffu;
        if (WRITE) key[i] = t.keys[i * 64 + lane];  // (all of them, now: the reads overlap, and the loop below is stores only)
    }
#pragma unroll
    for (int i = 0; i < CT_SLOTS / 64; ++i) {
        occupied[i] = __ballot(cnt[i] != 0);
        d += (unsigned int)__popcll(occupied[i]);
    }
    STAMP(4);  // occupied slots counted
    if (!WRITE) {
        if (lane == 0) distinct[bucket] = d;
        return;
    }
    // the bucket's distinct k-mers go to the rest of this wave's chunk and, when they do not fit, on into a fresh one
    const unsigned long long room = chunk.end - chunk.at;
    unsigned long long fresh = 0;
    if (d > room) {
        unsigned long long got = 0;
        if (lane == 0) got = atomicAdd(&out.cursor[0], (unsigned long long)CT_CHUNK);
        fresh = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
    }
    unsigned int before = 0;  // occupied slots of the rounds done (wave-uniform)
    if (d <= room && chunk.at + d <= out.capacity) {  // the usual case: one base address for the whole bucket, 32-bit offsets
        unsigned long long* __restrict__ kp = out.keys + chunk.at;
        unsigned int* __restrict__ cp = out.counts + chunk.at;
#pragma unroll
        for (int i = 0; i < CT_SLOTS / 64; ++i) {
            if (cnt[i]) {
                const unsigned int idx = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(occupied[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)occupied[i], 0u));
                kp[idx] = key[i];
                cp[idx] = cnt[i];
            }
            before += (unsigned int)__popcll(occupied[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CT_SLOTS / 64; ++i) {
            if (cnt[i]) {
                const unsigned int idx = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(occupied[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)occupied[i], 0u));
                put_counted(out, idx < room ? chunk.at + idx : fresh + (idx - room), key[i], cnt[i]);
            }
            before += (unsigned int)__popcll(occupied[i]);
        }
    }
    if (d > room) {
        chunk.at = fresh + (d - room);
        chunk.end = fresh + CT_CHUNK;
    } else {
        chunk.at += d;
    }
    chunk.distinct += d;
    STAMP(5);  // written out
}

// Every WAVE walks its own buckets with its own LDS: 12 buckets in flight per CU.  Software pipeline on top: while a wave
// counts bucket b it already holds the records of its next bucket in registers and has the range of the one after that on
// its way.  Three things keep the LDS waits of the bucket being counted from also sitting out that prefetch's DRAM latency:
//   * the ranges are NOT fetched with scalar loads (scalar memory shares the wait counter lgkmcnt with LDS): a zero the
//     compiler cannot see through puts them on the vector memory path, readfirstlane makes them scalar when consumed;
//   * the records are fetched through a global-address-space pointer (a flat load counts on lgkmcnt as well);
//   * every load of the pipeline is unconditional, indices clamped instead of branched on: with loads inside branches the
//     compiler cannot count what is in flight and waits for everything (vmcnt(0)) at the first use of anything.
template <bool WRITE>
__global__ __launch_bounds__(64 * CT_WAVES) void count_buckets_kernel(const ulonglong2* __restrict__ recs, uint32_t n, const uint32_t* __restrict__ starts,
                                                                      uint32_t n_buckets, int k, int canonical, unsigned int* __restrict__ distinct,
                                                                      CountedOut out, unsigned long long* cursor, uint2* __restrict__ overflow, uint32_t max_overflow)
{
    __shared__ WaveTable tables[CT_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    WaveTable& t = tables[wv];
    const unsigned long long kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    typedef const __attribute__((address_space(1))) unsigned long long* gptr_t;
    typedef const __attribute__((address_space(1))) uint32_t* gptr32_t;
    gptr_t grecs = (gptr_t) reinterpret_cast<const unsigned long long*>(recs);
    gptr32_t gstarts = (gptr32_t)starts;
    uint32_t vz = 0;
    asm volatile("" : "+v"(vz));
    const uint32_t stride = gridDim.x * CT_WAVES, last_b = n_buckets - 1, last_r = n - 1;
    auto range_of = [&](uint32_t bucket, uint32_t& vlo, uint32_t& vhi) {  // vector loads, always issued
        const uint32_t c = (bucket < n_buckets ? bucket : last_b) + vz;
        vlo = gstarts[c];
        vhi = gstarts[c + 1];
    };
    auto record_of = [&](uint32_t pos) {  // always issued; lanes beyond the bucket get a record they will not use
        const uint32_t c = pos < n ? pos : last_r;
        ulonglong2 r;
        r.x = grecs[2ull * c];
        r.y = grecs[2ull * c + 1];
        return r;
    };
    uint32_t b = blockIdx.x * CT_WAVES + wv, bn = b + stride;
    WaveChunk chunk;
    if (b >= n_buckets) {
        if (WRITE && lane == 0) out.holes[blockIdx.x * CT_WAVES + wv] = make_ulonglong2(0, 0);
        return;
    }
#ifdef BL_COUNT_STAMPS
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime();
#endif
    uint32_t vlo, vhi, vnlo, vnhi;
    range_of(b, vlo, vhi);
    range_of(bn, vnlo, vnhi);
    uint32_t lo = __builtin_amdgcn_readfirstlane(vlo), hi = __builtin_amdgcn_readfirstlane(vhi);
    ulonglong2 rec = record_of(lo + lane);
    while (b < n_buckets) {
        const uint32_t nlo = __builtin_amdgcn_readfirstlane(vnlo), nhi = __builtin_amdgcn_readfirstlane(vnhi);
        const ulonglong2 nrec = record_of(nlo + lane);  // the next bucket's records
        const uint32_t b2 = bn + stride;
        uint32_t v2lo, v2hi;
        range_of(b2, v2lo, v2hi);                       // the range of the one after
        count_one_bucket<WRITE>(t, lane, b, lo, hi, rec, grecs, last_r, k, canonical, kmask, distinct, out, chunk, cursor, overflow, max_overflow STAMP_PASS);
#ifdef BL_COUNT_STAMPS
        acc[7] += 1;
#endif
        b = bn; bn = b2;
        lo = nlo; hi = nhi; rec = nrec;
        vnlo = v2lo; vnhi = v2hi;
    }
    if (WRITE && lane == 0) {
        out.holes[blockIdx.x * CT_WAVES + wv] = make_ulonglong2(chunk.at, chunk.end);
        atomicAdd(&out.cursor[2], chunk.distinct);
    }
#ifdef BL_COUNT_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&bl_count_stamps[i], acc[i]);
#endif
}

// Fill the holes below `n_valid` (the number of counted k-mers) with the entries that sit at or beyond it: thread t moves the
// t-th such entry into the t-th hole slot.  below_*: the holes' parts below n_valid (start, exclusive prefix of their lengths),
// above_*: their parts at or beyond it (start, exclusive prefix), both in increasing position; the host prepares the four arrays
// from the few thousand holes.
__global__ __launch_bounds__(256) void fill_holes_kernel(CountedOut o, unsigned long long n_valid, unsigned long long n_moves, const unsigned long long* below_start,
                                                         const unsigned long long* below_prefix, uint32_t n_below, const unsigned long long* above_start,
                                                         const unsigned long long* above_prefix, uint32_t n_above)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_moves) return;
    uint32_t lo = 0, hi = n_below;  // last hole part with prefix <= t
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (below_prefix[mid] <= t) lo = mid;
        else hi = mid;
    }
    const unsigned long long dst = below_start[lo] + (t - below_prefix[lo]);
    // the t-th entry at or beyond n_valid that is not in a hole: n_valid + t + (hole slots in front of it); j = number of hole
    // parts in front of it = the first j for which that position lies in front of part j
    uint32_t a = 0, b = n_above;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (n_valid + t + above_prefix[mid] < above_start[mid]) b = mid;
        else a = mid + 1;
    }
    const unsigned long long src = n_valid + t + above_prefix[a];  // (above_prefix has n_above + 1 entries)
    unsigned long long key;
    unsigned int c;
    if (src < o.capacity) {
        key = o.keys[src];
        c = o.counts[src];
    } else {
        key = o.ext_keys[src - o.capacity];
        c = o.ext_counts[src - o.capacity];
    }
    o.keys[dst] = key;
    o.counts[dst] = c;
}

// records of the listed ranges, one after the other: the fallback's input (one workgroup per range)
__global__ __launch_bounds__(256) void gather_ranges_kernel(const ulonglong2* __restrict__ recs, const uint2* __restrict__ ranges,
                                                            const unsigned long long* __restrict__ dst_off, ulonglong2* __restrict__ dst)
{
    const uint2 r = ranges[blockIdx.x];
    ulonglong2* out = dst + dst_off[blockIdx.x];
    for (uint32_t i = r.x + threadIdx.x; i < r.y; i += blockDim.x) out[i - r.x] = recs[i];
}

__global__ void append_counted_kernel(const unsigned long long* keys, const unsigned int* counts, unsigned long long n_runs, unsigned long long* out_keys,
                                      unsigned int* out_counts, unsigned long long capacity, unsigned long long base)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_runs && base + i < capacity) {
        out_keys[base + i] = keys[i];
        out_counts[base + i] = counts[i];
    }
}

}  // namespace

#ifdef BL_COUNT_STAMPS
extern "C" void bl_dbg_count_stamps(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(bl_count_stamps), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(bl_count_stamps), z, sizeof(z));
}
#endif

extern "C" {

int bl_pack_super_kmers(bl_ctx* ctx, const bl_batch* batch, const uint64_t* d_first_pos, const uint8_t* d_sizes, const uint8_t* d_mm_pos, uint64_t n_groups,
                        uint32_t k, uint32_t m, uint64_t* d_records)
{
    if (!ctx || !batch || (n_groups && (!d_first_pos || !d_sizes || !d_records))) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    // a record without its minimizer's offset would be bucketed by its first m-mer by bl_count_super_kmers: occurrences of one k-mer
    // could then meet in different buckets and come out as several partial counts, with no error anywhere
    if (n_groups && !d_mm_pos) return bl_set_error(BL_ERR_INVALID, "d_mm_pos is required: the record's owner finds the minimizer through it");
    if (k < 1 || k > 32 || m < 1 || m > k || 2 * k - m > MAX_BASES) return bl_set_error(BL_ERR_INVALID, "need 1 <= m <= k <= 32 and 2k - m <= 59 (bases per packed record)");
    if (n_groups == 0) return BL_OK;
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned char*>(bl_batch_device_bases(batch)),
                       (unsigned long long)bl_batch_n_bases(batch), reinterpret_cast<const unsigned long long*>(d_first_pos), d_sizes, d_mm_pos,
                       (unsigned long long)n_groups, (int)k, reinterpret_cast<ulonglong2*>(d_records), (unsigned long long)bl_batch_origin(batch));
    SK_HIP(hipGetLastError());
    return BL_OK;
}

int bl_partition_records(bl_ctx* ctx, const uint64_t* d_hashes, const uint64_t* d_records, uint64_t n, uint32_t parts, uint64_t* d_out, uint64_t* counts)
{
    if (!ctx || !counts || parts == 0 || parts > MAX_PARTS || (n && (!d_hashes || !d_records || !d_out)))
        return bl_set_error(BL_ERR_INVALID, "bad argument (1 <= parts <= 64)");
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    unsigned long long host[MAX_PARTS];
    const hipError_t e = blpart::partition(reinterpret_cast<const ulonglong2*>(d_records), (unsigned long long)n, parts,
                                           HashArrayOwner{reinterpret_cast<const unsigned long long*>(d_hashes)}, reinterpret_cast<ulonglong2*>(d_out), host,
                                           bl_ctx_stream(ctx));
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    for (uint32_t b = 0; b < parts; ++b) counts[b] = host[b];
    return BL_OK;
}

int bl_expand_super_kmers(bl_ctx* ctx, const uint64_t* d_records, uint64_t n_groups, uint32_t k, uint32_t flags, uint64_t* d_kmers, uint64_t capacity,
                          uint64_t* n_kmers)
{
    if (!ctx || !n_kmers || (n_groups && !d_records)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (k < 1 || k > 32) return bl_set_error(BL_ERR_INVALID, "need 1 <= k <= 32");
    *n_kmers = 0;
    if (n_groups == 0) return BL_OK;
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    unsigned long long *sizes = nullptr, *offsets = nullptr;
    void* tmp = nullptr;
    size_t bytes = 0;
    const unsigned blocks = (unsigned)((n_groups + 255) / 256);
    sizes = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 4, 2 * n_groups * sizeof(unsigned long long)));  // (slots 4-6: the set operations')
    if (!sizes) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
    hipError_t e = hipSuccess;
    offsets = sizes + n_groups;
    hipLaunchKernelGGL(sizes_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const ulonglong2*>(d_records), (unsigned long long)n_groups, sizes);
    e = rocprim::exclusive_scan(nullptr, bytes, sizes, offsets, 0ull, n_groups, rocprim::plus<unsigned long long>(), s);
    if (e == hipSuccess) {
        tmp = bl_ctx_scratch(ctx, 5, bytes ? bytes : 16);
        if (!tmp) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, bytes, sizes, offsets, 0ull, n_groups, rocprim::plus<unsigned long long>(), s);
    unsigned long long last[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&last[0], offsets + n_groups - 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], sizes + n_groups - 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int rc = BL_OK;
    if (e == hipSuccess) {
        *n_kmers = last[0] + last[1];
        if (*n_kmers > capacity || (!d_kmers && *n_kmers)) {
            rc = bl_set_error(BL_ERR_CAPACITY, "expanded k-mers exceed the capacity of d_kmers (n_kmers holds the need)");
        } else {
            hipLaunchKernelGGL(expand_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const ulonglong2*>(d_records), offsets, (unsigned long long)n_groups,
                               (int)k, (flags & BL_FLAG_CANONICAL) ? 1 : 0, reinterpret_cast<unsigned long long*>(d_kmers));
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    return rc;
}

int bl_count_super_kmers(bl_ctx* ctx, const uint64_t* d_records, uint64_t n_groups, uint32_t k, uint32_t m, uint64_t seed, uint32_t flags, uint64_t* d_kmers,
                         uint32_t* d_counts, uint64_t capacity, uint64_t* n_distinct)
{
    if (!ctx || !n_distinct || (n_groups && !d_records)) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (k < 1 || k > 32 || m < 1 || m > k || 2 * k - m > MAX_BASES) return bl_set_error(BL_ERR_INVALID, "need 1 <= m <= k <= 32 and 2k - m <= 59 (bases per packed record)");
    if (n_groups >= (1ull << 32)) return bl_set_error(BL_ERR_INVALID, "at most 2^32 - 1 records per call");
    const int canonical = (flags & BL_FLAG_CANONICAL) ? 1 : 0;
    if (k == 32 && !canonical) return bl_set_error(BL_ERR_INVALID, "k = 32 needs the canonical flag here (the all-T 32-mer is the table's empty mark); use bl_expand_super_kmers + bl_sort_u64");
    *n_distinct = 0;
    if (n_groups == 0) return BL_OK;
    SK_HIP(hipSetDevice(bl_ctx_device(ctx)));
    hipStream_t s = bl_ctx_stream(ctx);
    const uint32_t n = (uint32_t)n_groups;
    // buckets of ~36 records (about 300 k-mers at k = 31, m = 15), one wave each; at most 2^26 of them
    unsigned long long want_buckets = (n_groups + 35) / 36;
    if (want_buckets > (1ull << 26)) want_buckets = 1ull << 26;
    const uint32_t n_buckets = (uint32_t)(want_buckets ? want_buckets : 1);
    int bits = 0;  // of a bucket number (what the radix sort looks at)
    while ((1ull << bits) < n_buckets) ++bits;
    const uint32_t max_overflow = n_buckets;  // every bucket may be oversized (reads of high coverage: few minimizers, many occurrences each)
    // scratch kept by the context between calls (hipMalloc / hipFree of gigabytes cost more than the kernels)
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t ids_bytes = up16(2ull * n * sizeof(uint32_t)), recs_bytes = (size_t)n * sizeof(ulonglong2), starts_bytes = up16(((size_t)n_buckets + 2) * sizeof(uint32_t));
    const size_t over_bytes = up16((size_t)max_overflow * sizeof(uint2)), offs_bytes = up16(((size_t)n_buckets + 1) * sizeof(unsigned long long));
    const size_t dist_bytes = up16(((size_t)n_buckets + 1) * sizeof(unsigned int));
    unsigned char* arena = static_cast<unsigned char*>(bl_ctx_scratch(ctx, 0, ids_bytes + recs_bytes + starts_bytes + over_bytes + offs_bytes + dist_bytes + 64));
    if (!arena) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
    uint32_t* ids = reinterpret_cast<uint32_t*>(arena);
    uint32_t* ids_sorted = ids + n;
    ulonglong2* recs_sorted = reinterpret_cast<ulonglong2*>(arena + ids_bytes);
    uint32_t* starts = reinterpret_cast<uint32_t*>(arena + ids_bytes + recs_bytes);
    uint2* overflow = reinterpret_cast<uint2*>(arena + ids_bytes + recs_bytes + starts_bytes);
    unsigned long long* offsets = reinterpret_cast<unsigned long long*>(arena + ids_bytes + recs_bytes + starts_bytes + over_bytes);
    unsigned int* distinct = reinterpret_cast<unsigned int*>(arena + ids_bytes + recs_bytes + starts_bytes + over_bytes + offs_bytes);
    unsigned long long* cursor = reinterpret_cast<unsigned long long*>(arena + ids_bytes + recs_bytes + starts_bytes + over_bytes + offs_bytes + dist_bytes);
    const ulonglong2* recs = reinterpret_cast<const ulonglong2*>(d_records);
    hipError_t e = hipMemsetAsync(cursor, 0, 4 * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bucket_id_kernel, dim3((n + 255) / 256), dim3(256), 0, s, recs, (unsigned long long)n, (int)m, canonical, (uint32_t)seed, n_buckets, ids);
        e = hipGetLastError();
    }
    if (e == hipSuccess && bits > 0) {
        size_t tmp_bytes = 0;
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, ids, ids_sorted, recs, recs_sorted, (size_t)n, 0, (unsigned)bits, s);
        void* tmp = nullptr;
        if (e == hipSuccess) {
            tmp = bl_ctx_scratch(ctx, 1, tmp_bytes ? tmp_bytes : 16);
            if (!tmp) e = hipErrorOutOfMemory;
        }
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, ids, ids_sorted, recs, recs_sorted, (size_t)n, 0, (unsigned)bits, s);
    } else if (e == hipSuccess) {
        ids_sorted = ids;
        recs_sorted = const_cast<ulonglong2*>(recs);
    }
    const uint32_t want = (n_buckets + CT_WAVES - 1) / CT_WAVES;
    const uint32_t grid = want < 256u * 6u ? want : 256u * 6u;  // 6 workgroups of 2 waves per CU by LDS (12.4 KB per wave): all resident, grid-stride over the buckets
    const uint32_t n_waves = grid * CT_WAVES;
    const bool write = d_kmers && d_counts;
    unsigned long long cur[3] = {0, 0, 0};  // [0] distinct k-mers of the table buckets, [1] buckets left to the fallback, [2] end of the chunks handed out
    CountedOut out{};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bucket_starts_kernel, dim3(n_buckets / 256 + 1), dim3(256), 0, s, ids_sorted, n, n_buckets, starts);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !write) {
        // the caller only asks how many: distinct k-mers per bucket, summed
        hipLaunchKernelGGL((count_buckets_kernel<false>), dim3(grid), dim3(64 * CT_WAVES), 0, s, recs_sorted, n, starts, n_buckets, (int)k, canonical, distinct, out,
                           cursor, overflow, max_overflow);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemsetAsync(distinct + n_buckets, 0, sizeof(unsigned int), s);
        size_t scan_bytes = 0;
        if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, scan_bytes, distinct, offsets, 0ull, (size_t)n_buckets + 1, rocprim::plus<unsigned long long>(), s);
        void* scan_tmp = nullptr;
        if (e == hipSuccess) {
            scan_tmp = bl_ctx_scratch(ctx, 2, scan_bytes ? scan_bytes : 16);
            if (!scan_tmp) e = hipErrorOutOfMemory;
        }
        if (e == hipSuccess) e = rocprim::exclusive_scan(scan_tmp, scan_bytes, distinct, offsets, 0ull, (size_t)n_buckets + 1, rocprim::plus<unsigned long long>(), s);
        if (e == hipSuccess) e = hipMemcpyAsync(&cur[0], offsets + n_buckets, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&cur[1], cursor + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    int rc = BL_OK;
    if (e == hipSuccess && write) {
        // ONE pass: count and write (see CountedOut).  Side buffer and hole list come from the context's scratch.
        const unsigned long long ext_n = (unsigned long long)n_waves * CT_CHUNK;
        const size_t ext_keys_bytes = up16(ext_n * sizeof(unsigned long long)), ext_counts_bytes = up16(ext_n * sizeof(unsigned int));
        const size_t holes_bytes = up16((size_t)n_waves * sizeof(ulonglong2)), lists_bytes = up16(((size_t)n_waves + 1) * sizeof(unsigned long long));
        unsigned char* side = static_cast<unsigned char*>(bl_ctx_scratch(ctx, 3, ext_keys_bytes + ext_counts_bytes + holes_bytes + 4 * lists_bytes + 64));
        if (!side) return bl_set_error(BL_ERR_OOM, "scratch allocation failed");
        out.keys = reinterpret_cast<unsigned long long*>(d_kmers);
        out.counts = d_counts;
        out.capacity = capacity;
        out.ext_keys = reinterpret_cast<unsigned long long*>(side);
        out.ext_counts = reinterpret_cast<unsigned int*>(side + ext_keys_bytes);
        out.ext_n = ext_n;
        out.cursor = cursor;
        out.holes = reinterpret_cast<ulonglong2*>(side + ext_keys_bytes + ext_counts_bytes);
        unsigned long long* lists = reinterpret_cast<unsigned long long*>(side + ext_keys_bytes + ext_counts_bytes + holes_bytes);
        hipLaunchKernelGGL((count_buckets_kernel<true>), dim3(grid), dim3(64 * CT_WAVES), 0, s, recs_sorted, n, starts, n_buckets, (int)k, canonical, distinct, out,
                           cursor, overflow, max_overflow);
        e = hipGetLastError();
        std::vector<ulonglong2> holes(n_waves);
        unsigned long long c3[3] = {0, 0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(c3, cursor, sizeof(c3), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(holes.data(), out.holes, (size_t)n_waves * sizeof(ulonglong2), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            cur[0] = c3[2];
            cur[1] = c3[1];
            cur[2] = c3[0];
            // the holes, in increasing position; those below the number of k-mers are filled from the entries beyond it
            std::vector<std::pair<unsigned long long, unsigned long long>> hs;
            unsigned long long hole_slots = 0;
            for (const ulonglong2& h : holes)
                if (h.y > h.x) {
                    hs.emplace_back(h.x, h.y);
                    hole_slots += h.y - h.x;
                }
            std::sort(hs.begin(), hs.end());
            const unsigned long long n_valid = cur[0];
            if (cur[2] - hole_slots != n_valid) rc = bl_set_error(BL_ERR_INTERNAL, "chunk accounting of the k-mer counter does not add up");
            if (rc == BL_OK && n_valid <= capacity && n_valid > 0) {
                std::vector<unsigned long long> below_start, below_prefix, above_start, above_prefix;
                unsigned long long n_moves = 0, above_total = 0;
                for (const auto& h : hs) {
                    if (h.first < n_valid) {
                        const unsigned long long end = h.second < n_valid ? h.second : n_valid;
                        below_start.push_back(h.first);
                        below_prefix.push_back(n_moves);
                        n_moves += end - h.first;
                    }
                    if (h.second > n_valid) {
                        const unsigned long long start = h.first > n_valid ? h.first : n_valid;
                        above_start.push_back(start);
                        above_prefix.push_back(above_total);
                        above_total += h.second - start;
                    }
                }
                above_prefix.push_back(above_total);
                above_start.push_back(~0ULL);
                if (n_moves) {
                    const size_t nb = below_start.size(), na = above_start.size() - 1;
                    unsigned long long *d_bs = lists, *d_bp = lists + (n_waves + 1), *d_as = lists + 2 * (size_t)(n_waves + 1), *d_ap = lists + 3 * (size_t)(n_waves + 1);
                    e = hipMemcpyAsync(d_bs, below_start.data(), nb * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_bp, below_prefix.data(), nb * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_as, above_start.data(), (na + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_ap, above_prefix.data(), (na + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) {
                        hipLaunchKernelGGL(fill_holes_kernel, dim3((unsigned)((n_moves + 255) / 256)), dim3(256), 0, s, out, n_valid, n_moves, d_bs, d_bp, (uint32_t)nb, d_as,
                                           d_ap, (uint32_t)na);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the host vectors go out of scope
                }
            }
        }
    }
    unsigned long long total = cur[0];
    if (e == hipSuccess && rc == BL_OK && cur[1] > max_overflow) rc = bl_set_error(BL_ERR_INTERNAL, "more oversized buckets than the fallback list holds");
    if (e == hipSuccess && rc == BL_OK && cur[1] > 0) {
        // fallback for the listed buckets: gather their records, expand, sort, run-length encode, append
        const uint32_t n_over = (uint32_t)cur[1];
        std::vector<uint2> ranges(n_over);
        e = hipMemcpy(ranges.data(), overflow, n_over * sizeof(uint2), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> off(n_over + 1, 0);
        for (uint32_t i = 0; i < n_over; ++i) off[i + 1] = off[i] + (ranges[i].y - ranges[i].x);
        const unsigned long long n_over_recs = off[n_over];
        ulonglong2* gathered = nullptr;
        unsigned long long *d_off = nullptr, *kmers = nullptr, *uniq = nullptr;
        unsigned int* cnts = nullptr;
        uint64_t n_kmers = 0;
        // (scratch slots 1-3 are free again here: the sort's workspace, the side buffer and the hole lists have done their work)
        if (e == hipSuccess) {
            const size_t gathered_bytes = up16(n_over_recs * sizeof(ulonglong2));
            unsigned char* a1 = static_cast<unsigned char*>(bl_ctx_scratch(ctx, 1, gathered_bytes + (n_over + 1) * sizeof(unsigned long long)));
            if (!a1) e = hipErrorOutOfMemory;
            gathered = reinterpret_cast<ulonglong2*>(a1);
            d_off = reinterpret_cast<unsigned long long*>(a1 + gathered_bytes);
        }
        if (e == hipSuccess) e = hipMemcpy(d_off, off.data(), (n_over + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_ranges_kernel, dim3(n_over), dim3(256), 0, s, recs_sorted, overflow, d_off, gathered);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            int r2 = bl_expand_super_kmers(ctx, reinterpret_cast<const uint64_t*>(gathered), n_over_recs, k, flags, nullptr, 0, &n_kmers);
            if (r2 != BL_OK && r2 != BL_ERR_CAPACITY) rc = r2;
        }
        if (e == hipSuccess && rc == BL_OK) {
            kmers = static_cast<unsigned long long*>(bl_ctx_scratch(ctx, 2, 2 * n_kmers * sizeof(unsigned long long) + 8));
            cnts = static_cast<unsigned int*>(bl_ctx_scratch(ctx, 3, n_kmers * sizeof(unsigned int) + 4));
            if (!kmers || !cnts) e = hipErrorOutOfMemory;
        }
        if (e == hipSuccess && rc == BL_OK) {
            uniq = kmers + n_kmers;
            rc = bl_expand_super_kmers(ctx, reinterpret_cast<const uint64_t*>(gathered), n_over_recs, k, flags, reinterpret_cast<uint64_t*>(kmers), n_kmers, &n_kmers);
            if (rc == BL_OK) rc = bl_sort_u64(ctx, reinterpret_cast<uint64_t*>(kmers), n_kmers);
            uint64_t runs = 0;
            if (rc == BL_OK) rc = bl_count_sorted_u64(ctx, reinterpret_cast<const uint64_t*>(kmers), n_kmers, reinterpret_cast<uint64_t*>(uniq), cnts, &runs);
            if (rc == BL_OK && runs) {
                if (d_kmers && d_counts) {
                    hipLaunchKernelGGL(append_counted_kernel, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, s, uniq, cnts, (unsigned long long)runs,
                                       reinterpret_cast<unsigned long long*>(d_kmers), d_counts, (unsigned long long)capacity, total);
                    e = hipGetLastError();
                    if (e == hipSuccess) e = hipStreamSynchronize(s);
                }
                total += runs;
            }
        }
    }
    if (e != hipSuccess) return bl_set_error(e == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e));
    if (rc != BL_OK) return rc;
    *n_distinct = total;
    if (total > capacity || ((!d_kmers || !d_counts) && total)) return bl_set_error(BL_ERR_CAPACITY, "distinct k-mers exceed the capacity of the output arrays (n_distinct holds the need)");
    return BL_OK;
}

}  // extern "C"
