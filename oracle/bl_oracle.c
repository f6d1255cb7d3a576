/* TEST INFRASTRUCTURE ONLY — see bl_oracle.h.  Plain C11 restatement of the reference's
 * algorithms for the k-mer / minimizer streaming path, each function citing the reference
 * file:line it follows.  Written for clarity, not speed (the streaming variants keep the
 * reference's operation counts because they double as the CPU baseline in bench.py). */
#include "bl_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- synthetic input */

uint64_t blo_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

void blo_synth(uint64_t seed, uint64_t first, uint64_t n, char* out)
{
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t j = 0; j < n; ++j) {
        uint64_t i = first + j;
        uint64_t word = blo_splitmix64(seed + (i >> 5));
        out[j] = acgt[(word >> (2 * (i & 31))) & 3];
    }
}

/* ---------------------------------------------------------------- nucleotide table (constants.hpp:12-21) */

uint8_t blo_nt4(uint8_t c)
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

/* ---------------------------------------------------------------- MurmurHash3_x64_128 (bundled/MurmurHash3.cpp) */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); } /* :39-42 */

static inline uint64_t fmix64(uint64_t k) /* :81-90 */
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t* p)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

void blo_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t out[2]) /* :263-341 */
{
    const uint8_t* data = (const uint8_t*)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;

    for (int i = 0; i < nblocks; ++i) { /* body :280-292 */
        uint64_t k1 = load_le64(data + 16 * i), k2 = load_le64(data + 16 * i + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }

    const uint8_t* tail = data + 16 * nblocks; /* tail :297-323 */
    const int rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for (int i = rem - 1; i >= 8; --i) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 ^= (uint64_t)tail[i] << (8 * i);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }

    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len; /* finalisation :328-340 */
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

uint64_t blo_hash64_bytes(const uint8_t* key, uint32_t len, uint32_t seed) /* hash.hpp:50-53 */
{
    uint64_t h[2];
    blo_murmur3_x64_128(key, (int)len, seed, h);
    return h[0];
}

/* hash.hpp:55-59: the raw bytes of v (little-endian object representation), seed -> uint32_t.  MurmurHash3_x64_128 on a key of
 * exactly 8 bytes runs no block (nblocks = 0), the tail's case 8..1 builds k1 = the key itself, k2 stays 0 (MurmurHash3.cpp:297-323):
 * written out, so that the compiler sees what the reference's compiler sees after inlining (the byte-wise general form above cost
 * the timed CPU baseline a third of its rate against the reference).  blo_hash64_u64_general keeps the general path: tests pin
 * the two against each other and against the reference's own hash values. */
uint64_t blo_hash64_u64_general(uint64_t v, uint64_t seed)
{
    uint8_t raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = (uint8_t)(v >> (8 * i));
    return blo_hash64_bytes(raw, 8, (uint32_t)seed);
}

uint64_t blo_hash64_u64(uint64_t v, uint64_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = (uint32_t)seed, h2 = (uint32_t)seed;
    uint64_t k1 = v;
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; /* tail :312-323 with len & 15 == 8 */
    h1 ^= 8; h2 ^= 8;                                  /* finalisation :328-340 */
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

uint64_t blo_remix(uint64_t z) /* hash.hpp:81-85 */
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* ---------------------------------------------------------------- rolling 2-bit unit (kmer_view.hpp:190-199, 218-227) */

typedef struct {
    uint64_t fwd, rc, mask;
    unsigned k, shift, run, strand;
    int canonical;
} roller;

static void roller_init(roller* r, unsigned k, int canonical)
{
    r->fwd = r->rc = 0; /* zero-initialised buffers: the contract of SURVEY.md §8a-a2 */
    r->k = k;
    r->mask = (2 * k != 64) ? ((1ULL << (2 * k)) - 1) : ~0ULL; /* :166-167 */
    r->shift = 2 * (k - 1);                                   /* :168 */
    r->run = 0;
    r->strand = 0;
    r->canonical = canonical;
}

/* consume one character; returns 1 if it was a base, 0 if it was a break */
static inline int roller_push(roller* r, uint8_t ch)
{
    uint8_t c = blo_nt4(ch);
    if (c < 4) {
        r->fwd = ((r->fwd << 2) | c) & r->mask;                 /* :194 */
        r->rc = (r->rc >> 2) | ((3ULL ^ c) << r->shift);        /* :195 */
        if (r->canonical && r->fwd != r->rc) r->strand = r->fwd < r->rc ? 0 : 1; /* :196 */
        ++r->run;
        return 1;
    }
    r->run = 0;
    return 0;
}

static inline uint64_t roller_value(const roller* r) { return r->strand ? r->rc : r->fwd; }

/* ---------------------------------------------------------------- kmer_view iterator protocol */

size_t blo_kmer_items(const char* s, size_t n, unsigned k, int canonical, int complete,
                      uint64_t* values, uint8_t* is_null, uint64_t* positions, uint64_t* ids, size_t cap)
{
    roller r;
    roller_init(&r, k, canonical);
    size_t pos = 0, id = 0, cnt = 0;
    int dead = 0; /* reference would run out of bounds here (Q2): we stop instead */

#define FIND_FIRST_GOOD()                                                          \
    do { /* kmer_view.hpp:213-234 */                                               \
        while (pos != n && r.run < k) roller_push(&r, (uint8_t)s[pos++]);          \
        if (r.run < k) { r.run = 0; dead = 1; }                                    \
    } while (0)
#define PUT(null_)                                                                 \
    do {                                                                           \
        if (cnt < cap) {                                                           \
            values[cnt] = (null_) ? 0 : roller_value(&r);                          \
            is_null[cnt] = (uint8_t)(null_);                                       \
            positions[cnt] = pos - k;                                              \
            ids[cnt] = id;                                                         \
        }                                                                          \
        ++cnt;                                                                     \
    } while (0)

    if (k == 0 || k > 32) return 0;
    FIND_FIRST_GOOD(); /* cbegin(), :162-170 */
    while (!dead && pos != n) { /* it != cend() compares the char iterator only, :57 */
        PUT(r.run == 0);       /* operator*, :172-179 */
        ++id;                  /* operator++, :181-202 */
        if (r.run == 0) FIND_FIRST_GOOD();
        else roller_push(&r, (uint8_t)s[pos++]);
    }
    if (complete && !dead && r.run >= k) PUT(0); /* the item the idiom leaves behind (Q1) */
    return cnt;
#undef FIND_FIRST_GOOD
#undef PUT
}

/* ---------------------------------------------------------------- per-position unit table over a batch */

void blo_units(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned k, int canonical,
               uint64_t* values, uint8_t* valid)
{
    for (size_t q = 0; q < n_seqs; ++q) {
        uint64_t b = offsets[q], e = offsets[q + 1];
        roller r;
        roller_init(&r, k, canonical);
        for (uint64_t p = b; p < e; ++p) { values[p] = 0; valid[p] = 0; }
        for (uint64_t p = b; p < e; ++p) {
            roller_push(&r, (uint8_t)s[p]);
            if (r.run >= k) {
                values[p + 1 - k] = roller_value(&r);
                valid[p + 1 - k] = 1;
            }
        }
    }
}

void blo_kmer_digest(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned k, int canonical,
                     uint64_t seed, int drop_last, int threads, uint64_t digest[4])
{
    uint64_t cnt = 0, xv = 0, xh = 0, sh = 0;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1) reduction(+ : cnt, sh) reduction(^ : xv, xh)
    for (size_t q = 0; q < n_seqs; ++q) {
        uint64_t b = offsets[q], e = offsets[q + 1];
        roller r;
        roller_init(&r, k, canonical);
        for (uint64_t p = b; p < e; ++p) {
            roller_push(&r, (uint8_t)s[p]);
            if (r.run >= k && !(drop_last && p + 1 == e)) {
                uint64_t v = roller_value(&r);
                uint64_t h = blo_hash64_u64(v, seed);
                ++cnt; xv ^= v; xh ^= h; sh += h;
            }
        }
    }
    digest[0] = cnt; digest[1] = xv; digest[2] = xh; digest[3] = sh;
}

/* ---------------------------------------------------------------- minimizers */

typedef struct {
    uint64_t* value; uint64_t* pos; uint64_t* hash; size_t cap; size_t cnt;
    uint64_t xv, xh, xp;
} mm_sink;

static inline void mm_emit(mm_sink* o, uint64_t v, uint64_t p, uint64_t h)
{
    if (o->cnt < o->cap) {
        if (o->value) o->value[o->cnt] = v;
        if (o->pos) o->pos[o->cnt] = p;
        if (o->hash) o->hash[o->cnt] = h;
    }
    ++o->cnt; o->xv ^= v; o->xh ^= h; o->xp ^= p;
}

/* brute-force checker: materialise the units of one sequence, then leftmost argmin per window */
static void mm_seq_brute(const char* s, uint64_t b, uint64_t e, unsigned unit, unsigned w, uint64_t seed,
                         int canonical, mm_sink* o)
{
    uint64_t n = e - b;
    if (n < (uint64_t)unit + w - 1) return;
    uint64_t nu = n - unit + 1;
    uint64_t* val = (uint64_t*)malloc(nu * sizeof(uint64_t));
    uint64_t* hsh = (uint64_t*)malloc(nu * sizeof(uint64_t));
    uint8_t* ok = (uint8_t*)calloc(nu, 1);
    roller r;
    roller_init(&r, unit, canonical);
    for (uint64_t p = 0; p < n; ++p) {
        roller_push(&r, (uint8_t)s[b + p]);
        if (r.run >= unit) {
            uint64_t u = p + 1 - unit;
            val[u] = roller_value(&r);
            hsh[u] = blo_hash64_u64(val[u], seed); /* minimizer_view.hpp:202,260 */
            ok[u] = 1;
        }
    }
    int have_prev = 0;
    uint64_t prev_arg = 0;
    for (uint64_t j = 0; j + w <= nu; ++j) {
        int valid = 1;
        uint64_t arg = j;
        for (unsigned t = 0; t < w; ++t) {
            if (!ok[j + t]) { valid = 0; break; }
            if (hsh[j + t] < hsh[arg]) arg = j + t; /* strict '<': leftmost wins, :283,374 */
        }
        if (!valid) { have_prev = 0; continue; } /* a break clears the window, :241-242 */
        if (!have_prev || arg != prev_arg) mm_emit(o, val[arg], b + arg, hsh[arg]);
        have_prev = 1;
        prev_arg = arg;
    }
    free(val); free(hsh); free(ok);
}

/* streaming variant: per-base roll, one Murmur per unit, ring of w hashes, push-compare with the
 * running minimum (:283) and a rescan when the minimum leaves the window (:367-376) */
#define BLO_MAX_W 256
static void mm_seq_stream(const char* s, uint64_t b, uint64_t e, unsigned unit, unsigned w, uint64_t seed,
                          int canonical, mm_sink* o)
{
    uint64_t rv[BLO_MAX_W], rh[BLO_MAX_W];
    roller r;
    roller_init(&r, unit, canonical);
    uint64_t filled = 0;      /* units in the current valid run */
    uint64_t min_u = 0;       /* unit index (sequence-relative) of the running leftmost minimum */
    uint64_t last_emitted = ~0ULL;
    for (uint64_t p = 0; p < e - b; ++p) {
        if (!roller_push(&r, (uint8_t)s[b + p])) { filled = 0; last_emitted = ~0ULL; continue; }
        if (r.run < unit) continue;
        uint64_t u = p + 1 - unit;
        uint64_t v = roller_value(&r), h = blo_hash64_u64(v, seed);
        rv[u % w] = v; rh[u % w] = h;
        if (filled == 0) min_u = u;
        else if (min_u + w <= u) { /* the minimum fell out: rescan the w live slots left to right */
            min_u = u + 1 - w;
            for (uint64_t t = u + 2 - w; t <= u; ++t)
                if (rh[t % w] < rh[min_u % w]) min_u = t;
        } else if (h < rh[min_u % w]) min_u = u;
        ++filled;
        if (filled >= w && min_u != last_emitted) {
            mm_emit(o, rv[min_u % w], b + min_u, rh[min_u % w]);
            last_emitted = min_u;
        }
    }
}

size_t blo_minimizers(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned unit, unsigned w,
                      uint64_t seed, int canonical, int brute,
                      uint64_t* out_value, uint64_t* out_pos, uint64_t* out_hash, size_t cap)
{
    mm_sink o = {out_value, out_pos, out_hash, cap, 0, 0, 0, 0};
    if (unit == 0 || unit > 32 || w == 0 || w > BLO_MAX_W) return 0;
    for (size_t q = 0; q < n_seqs; ++q) {
        if (brute) mm_seq_brute(s, offsets[q], offsets[q + 1], unit, w, seed, canonical, &o);
        else mm_seq_stream(s, offsets[q], offsets[q + 1], unit, w, seed, canonical, &o);
    }
    return o.cnt;
}

void blo_minimizer_digest(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned unit, unsigned w,
                          uint64_t seed, int canonical, int threads, uint64_t digest[4])
{
    uint64_t cnt = 0, xv = 0, xh = 0, xp = 0;
    digest[0] = digest[1] = digest[2] = digest[3] = 0;
    if (unit == 0 || unit > 32 || w == 0 || w > BLO_MAX_W) return;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1) reduction(+ : cnt) reduction(^ : xv, xh, xp)
    for (size_t q = 0; q < n_seqs; ++q) {
        mm_sink o = {NULL, NULL, NULL, 0, 0, 0, 0, 0};
        mm_seq_stream(s, offsets[q], offsets[q + 1], unit, w, seed, canonical, &o);
        cnt += o.cnt; xv ^= o.xv; xh ^= o.xh; xp ^= o.xp;
    }
    digest[0] = cnt; digest[1] = xv; digest[2] = xh; digest[3] = xp;
}

/* ---------------------------------------------------------------- super-k-mers (super_kmer_view.hpp:121-135 intent) */

size_t blo_super_kmers(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned k, unsigned m,
                       uint64_t seed, int canonical,
                       uint64_t* out_minimizer, uint64_t* out_first_pos, uint8_t* out_mm_pos, uint8_t* out_size,
                       uint64_t* out_hash, size_t cap)
{
    size_t cnt = 0;
    if (m == 0 || m > 32 || k < m || k - m + 1 > BLO_MAX_W) return 0;
    const unsigned w = k - m + 1; /* minimizer_view.hpp:168 */
    for (size_t q = 0; q < n_seqs; ++q) {
        uint64_t b = offsets[q], e = offsets[q + 1], n = e - b;
        if (n < k) continue;
        uint64_t nu = n - m + 1;
        uint64_t* val = (uint64_t*)malloc(nu * sizeof(uint64_t));
        uint64_t* hsh = (uint64_t*)malloc(nu * sizeof(uint64_t));
        uint8_t* ok = (uint8_t*)calloc(nu, 1);
        roller r;
        roller_init(&r, m, canonical);
        for (uint64_t p = 0; p < n; ++p) {
            roller_push(&r, (uint8_t)s[b + p]);
            if (r.run >= m) {
                uint64_t u = p + 1 - m;
                val[u] = roller_value(&r);
                hsh[u] = blo_hash64_u64(val[u], seed);
                ok[u] = 1;
            }
        }
        int open = 0;
        uint64_t g_arg = 0, g_first = 0, g_size = 0;
        for (uint64_t j = 0;; ++j) { /* j = k-mer start (sequence-relative); one step past the end closes the last group */
            int valid = (j + w <= nu);
            uint64_t arg = j;
            if (valid)
                for (unsigned t = 0; t < w; ++t) {
                    if (!ok[j + t]) { valid = 0; break; }
                    if (hsh[j + t] < hsh[arg]) arg = j + t;
                }
            if (open && (!valid || arg != g_arg)) { /* close the running group */
                if (cnt < cap) {
                    if (out_minimizer) out_minimizer[cnt] = val[g_arg];
                    if (out_first_pos) out_first_pos[cnt] = b + g_first;
                    if (out_mm_pos) out_mm_pos[cnt] = (uint8_t)(g_arg - g_first); /* :132 */
                    if (out_size) out_size[cnt] = (uint8_t)g_size;                 /* :133 */
                    if (out_hash) out_hash[cnt] = hsh[g_arg];
                }
                ++cnt;
                open = 0;
            }
            if (j + w > nu) break;
            if (valid) {
                if (!open) { open = 1; g_arg = arg; g_first = j; g_size = 0; }
                ++g_size;
            }
        }
        free(val); free(hsh); free(ok);
    }
    return cnt;
}

/* ---------------------------------------------------------------- syncmers */

unsigned blo_minimizer_position(uint64_t km, unsigned k, unsigned m) /* kmer_view.hpp:266-283 */
{
    const uint64_t mask = (2 * m != 64) ? ((1ULL << (2 * m)) - 1) : ~0ULL; /* src/kmer_view.cpp:13-14 */
    uint64_t mval = blo_hash64_u64(km & mask, 0); /* :272 — the "+1" hash of the reference */
    unsigned minpos = 0;
    for (unsigned i = 0; i < (uint8_t)(k - m + 1); ++i) {
        uint64_t val = blo_hash64_u64(km & mask, 0); /* :275, seed hard-wired 0 */
        if (mval >= val) { mval = val; minpos = i; }  /* '>=': the later (= more leftward) m-mer wins ties */
        km >>= 2;
    }
    return k - m - minpos; /* :282 */
}

size_t blo_syncmers(const char* s, const uint64_t* offsets, size_t n_seqs, unsigned k, unsigned m,
                    unsigned soff, unsigned eoff, int canonical, int drop_last, int threads,
                    uint64_t* out_pos, size_t cap)
{
    uint64_t cnt = 0;
    if (k == 0 || k > 32 || m == 0 || m > k) return 0;
    if (out_pos) threads = 1; /* ordered output: single thread */
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1) reduction(+ : cnt)
    for (size_t q = 0; q < n_seqs; ++q) {
        uint64_t b = offsets[q], e = offsets[q + 1];
        roller r;
        roller_init(&r, k, canonical);
        for (uint64_t p = b; p < e; ++p) {
            roller_push(&r, (uint8_t)s[p]);
            if (r.run >= k && !(drop_last && p + 1 == e)) {
                unsigned off = blo_minimizer_position(roller_value(&r), k, m);
                if (off == soff || off == eoff) { /* syncmer_sampler.hpp:130-137 */
                    if (out_pos && cnt < cap) out_pos[cnt] = p + 1 - k;
                    ++cnt;
                }
            }
        }
    }
    return (size_t)cnt;
}
