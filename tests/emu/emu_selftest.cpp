// TEST INFRASTRUCTURE — sanitizer self-test: the emulated HIP tile pipeline (emu_scan.cpp, built
// with -fsanitize=address,undefined) against the C oracle (oracle/bl_oracle.c) on randomised
// batches: breaks, ragged / empty / short sequences, fixed-length reads, sub-ranges, tie-heavy
// low-complexity DNA, sizes straddling tile boundaries.  Exit code 0 = all equal.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../oracle/bl_oracle.h"

struct EmuBatch;
extern "C" {
EmuBatch* emu_batch(const uint8_t* bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len);
void emu_batch_free(EmuBatch* b);
void emu_minimizers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned unit, unsigned w, uint64_t seed, unsigned flags,
                    uint64_t* out_value, uint64_t* out_pos, uint64_t* out_hash, uint64_t capacity, unsigned long long* result);
void emu_super_kmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, unsigned m, uint64_t seed, unsigned flags,
                     uint64_t* out_min, uint64_t* out_first, uint8_t* out_mmpos, uint8_t* out_size, uint64_t* out_hash, uint64_t capacity,
                     unsigned long long* result);
void emu_syncmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, unsigned s, unsigned soff, unsigned eoff, uint64_t seed,
                  unsigned flags, uint64_t* out_pos, uint64_t capacity, unsigned long long* result);
void emu_kmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, uint64_t seed, unsigned flags, uint64_t* out_value,
               uint64_t* out_hash, uint8_t* out_valid, unsigned long long* result);
void emu_hash_sample(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, uint64_t seed, uint64_t threshold, unsigned flags,
                     uint64_t* out_value, uint64_t* out_pos, uint64_t* out_hash, uint64_t capacity, unsigned long long* result);
uint64_t emu_hash64(uint64_t v, uint64_t seed);
int emu_frl_scans();
int emu_frl_redone();
int emu_closed_redone();
int emu_sy2_redone();
int emu_pos_redone();
int emu_frl_tiles();
}

static int g_fail = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            if (g_fail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } \
            ++g_fail;                                      \
        }                                                  \
    } while (0)

struct Case {
    std::vector<uint8_t> seq;
    std::vector<uint64_t> offsets;  // always materialised for the oracle
    uint64_t read_len;              // != 0: emu gets read_len instead of offsets
    bool single;                    // emu gets neither (one sequence)
    std::string name;
};

static std::vector<uint64_t> fixed_offsets(uint64_t n, uint64_t L)
{
    std::vector<uint64_t> o;
    for (uint64_t p = 0; p < n; p += L) o.push_back(p);
    o.push_back(n);
    if (n == 0) o = {0, 0};
    return o;
}

static void check_case(const Case& c, std::mt19937_64& rng)
{
    const uint64_t n = c.seq.size();
    const size_t n_seqs = c.offsets.size() - 1;
    EmuBatch* b = c.single ? emu_batch(c.seq.data(), n, nullptr, 0, 0)
                  : c.read_len ? emu_batch(c.seq.data(), n, nullptr, 0, c.read_len)
                               : emu_batch(c.seq.data(), n, c.offsets.data(), n_seqs, 0);
    const char* s = reinterpret_cast<const char*>(c.seq.data());
    const uint64_t cap = n + 2;
    std::vector<uint64_t> ov(cap), op(cap), oh(cap), ev(cap), ep(cap), eh(cap), of(cap), ef(cap);
    std::vector<uint8_t> om(cap), os(cap), em(cap), es(cap);
    unsigned long long res[8];

    struct MM { unsigned unit, w; uint64_t seed; int canon; };
    const MM mms[] = {{31, 11, 42, 1}, {15, 17, 42, 1}, {15, 10, 7, 1}, {19, 19, 8, 1}, {21, 5, 9, 0}, {11, 21, 0, 0}, {5, 4, 1, 1}, {32, 2, 9, 1}, {8, 1, 3, 0}, {1, 1, 0, 1},
                      {32, 64, 5, 1}, {21, 33, 6, 0}, {3, 16, 7, 1}, {16, 17, 8, 0}};
    std::vector<MM> mm_cases(mms, mms + sizeof(mms) / sizeof(mms[0]));
    for (int extra = 0; extra < 3; ++extra)  // every width from 2 to 32 has a kernel of its own: three of them, at random, per case
        mm_cases.push_back(MM{(unsigned)(1 + rng() % 32), (unsigned)(2 + rng() % 31), rng() % 100, (int)(rng() % 2)});
    for (const MM& m : mm_cases) {
        // whole batch
        size_t cnt = blo_minimizers(s, c.offsets.data(), n_seqs, m.unit, m.w, m.seed, m.canon, 1, ov.data(), op.data(), oh.data(), cap);
        emu_minimizers(b, 0, 0, m.unit, m.w, m.seed, m.canon ? 1 : 0, ev.data(), ep.data(), eh.data(), cap, res);
        CHECK(res[0] == cnt, "%s mm(%u,%u) count %llu vs %zu", c.name.c_str(), m.unit, m.w, res[0], cnt);
        if (res[0] == cnt) {
            uint64_t xv = 0, xh = 0, xp = 0;
            for (size_t i = 0; i < cnt; ++i) {
                CHECK(ev[i] == ov[i] && ep[i] == op[i] && eh[i] == oh[i], "%s mm(%u,%u) record %zu: pos %llu vs %llu", c.name.c_str(), m.unit,
                      m.w, i, (unsigned long long)ep[i], (unsigned long long)op[i]);
                xv ^= ov[i]; xh ^= oh[i]; xp ^= op[i];
            }
            CHECK(res[1] == xv && res[2] == xh && res[3] == xp, "%s mm(%u,%u) digest", c.name.c_str(), m.unit, m.w);
        }
        // a random sub-range: the union rule says its records are exactly the whole-batch records whose
        // WINDOW starts in the range; check via two adjacent ranges that together give the whole
        if (n > 40) {
            const uint64_t cut = 1 + rng() % (n - 1);
            unsigned long long r1[8], r2[8];
            emu_minimizers(b, 0, cut, m.unit, m.w, m.seed, m.canon ? 1 : 0, ev.data(), ep.data(), eh.data(), cap, r1);
            const uint64_t c1 = r1[0];
            emu_minimizers(b, cut, 0, m.unit, m.w, m.seed, m.canon ? 1 : 0, ev.data() + c1, ep.data() + c1, eh.data() + c1, cap - c1, r2);
            CHECK(c1 + r2[0] == cnt, "%s mm(%u,%u) split at %llu: %llu + %llu vs %zu", c.name.c_str(), m.unit, m.w, (unsigned long long)cut,
                  (unsigned long long)c1, r2[0], cnt);
            if (c1 + r2[0] == cnt)
                for (size_t i = 0; i < cnt; ++i)
                    CHECK(ev[i] == ov[i] && ep[i] == op[i] && eh[i] == oh[i], "%s mm(%u,%u) split record %zu", c.name.c_str(), m.unit, m.w, i);
        }
    }

    struct SK { unsigned k, m; uint64_t seed; int canon; };
    const SK sks[] = {{31, 15, 42, 1}, {21, 8, 0, 0}, {24, 15, 6, 1}, {27, 9, 7, 1}, {19, 15, 8, 0}, {31, 31, 5, 1}, {31, 21, 1, 1}, {21, 11, 2, 0}, {32, 1, 3, 1}, {40, 9, 4, 1}};
    std::vector<SK> sk_cases(sks, sks + sizeof(sks) / sizeof(sks[0]));
    for (int extra = 0; extra < 2; ++extra) {  // every width from 2 to 32 has a kernel of its own in this mode too: two at random per case
        const unsigned m = (unsigned)(1 + rng() % 32), w = (unsigned)(2 + rng() % 31);
        sk_cases.push_back(SK{m + w - 1, m, rng() % 100, (int)(rng() % 2)});
    }
    for (const SK& k : sk_cases) {
        size_t cnt = blo_super_kmers(s, c.offsets.data(), n_seqs, k.k, k.m, k.seed, k.canon, ov.data(), of.data(), om.data(), os.data(),
                                     oh.data(), cap);
        emu_super_kmers(b, 0, 0, k.k, k.m, k.seed, k.canon ? 1 : 0, ev.data(), ef.data(), em.data(), es.data(), eh.data(), cap, res);
        CHECK(res[0] == cnt && res[4] == cnt, "%s sk(%u,%u) count %llu/%llu vs %zu", c.name.c_str(), k.k, k.m, res[0], res[4], cnt);
        if (res[0] == cnt)
            for (size_t i = 0; i < cnt; ++i)
                CHECK(ev[i] == ov[i] && ef[i] == of[i] && em[i] == om[i] && es[i] == os[i] && eh[i] == oh[i],
                      "%s sk(%u,%u) group %zu: first %llu vs %llu size %u vs %u", c.name.c_str(), k.k, k.m, i, (unsigned long long)ef[i],
                      (unsigned long long)of[i], es[i], os[i]);
    }

    struct SY { unsigned k, s, a, b; int canon; };
    const SY sys[] = {{31, 11, 0, 20, 1}, {31, 11, 0, 20, 0}, {21, 8, 0, 13, 1}, {7, 4, 0, 3, 1}, {15, 15, 0, 0, 1}, {31, 15, 0, 16, 1},
                      {32, 12, 3, 9, 1}, {9, 1, 0, 8, 0}, {28, 12, 5, 5, 1},
                      // k = 31, s = 11 with offsets other than {0, 20}: count_tile's SY = 2 form (exact argmins deferred to a second run)
                      {31, 11, 3, 9, 1}, {31, 11, 0, 0, 1}, {31, 11, 20, 20, 1}, {31, 11, 5, 20, 1},
                      // closed syncmers of other shapes: the run-time width kernels (phase_sync_closed_rt), every doubling width and both groups
                      {31, 8, 0, 23, 1}, {31, 8, 23, 0, 0}, {32, 1, 0, 31, 1}, {20, 16, 0, 4, 1}, {25, 12, 0, 13, 1}, {21, 11, 0, 10, 1}, {15, 5, 0, 10, 0},
                      {31, 15, 16, 0, 1}, {12, 10, 0, 2, 1}, {9, 8, 0, 1, 1}, {30, 13, 0, 17, 1}, {32, 14, 0, 18, 1}, {16, 13, 0, 3, 1}, {14, 9, 0, 5, 0}};
    for (const SY& y : sys) {
        for (int drop = 0; drop < 2; ++drop) {
            size_t cnt = blo_syncmers(s, c.offsets.data(), n_seqs, y.k, y.s, y.a, y.b, y.canon, drop, 1, op.data(), cap);
            emu_syncmers(b, 0, 0, y.k, y.s, y.a, y.b, 0, (y.canon ? 1u : 0u) | (drop ? 2u : 0u), ep.data(), cap, res);
            CHECK(res[0] == cnt, "%s sync(%u,%u,%u,%u,c%d,d%d) count %llu vs %zu", c.name.c_str(), y.k, y.s, y.a, y.b, y.canon, drop, res[0], cnt);
            if (res[0] == cnt)
                for (size_t i = 0; i < cnt; ++i)
                    CHECK(ep[i] == op[i], "%s sync(%u,%u) pos %zu: %llu vs %llu", c.name.c_str(), y.k, y.s, i, (unsigned long long)ep[i],
                          (unsigned long long)op[i]);
        }
    }

    std::vector<uint64_t> uv(n + 1), kv(n + 1), kh(n + 1);
    std::vector<uint8_t> uok(n + 1), kok(n + 1);
    for (unsigned k : {1u, 2u, 5u, 15u, 16u, 17u, 21u, 31u, 32u}) {
        for (int canon = 0; canon < 2; ++canon) {
            blo_units(s, c.offsets.data(), n_seqs, k, canon, uv.data(), uok.data());
            emu_kmers(b, 0, 0, k, 77, canon ? 1u : 0u, kv.data(), kh.data(), kok.data(), res);
            uint64_t cnt = 0;
            for (uint64_t p = 0; p < n; ++p) {
                CHECK(kok[p] == uok[p], "%s kmers k%u c%d valid @%llu: %u vs %u", c.name.c_str(), k, canon, (unsigned long long)p, kok[p], uok[p]);
                if (uok[p] && kok[p]) {
                    CHECK(kv[p] == uv[p], "%s kmers k%u c%d value @%llu", c.name.c_str(), k, canon, (unsigned long long)p);
                    CHECK(kh[p] == blo_hash64_u64(uv[p], 77), "%s kmers k%u c%d hash @%llu", c.name.c_str(), k, canon, (unsigned long long)p);
                }
                cnt += uok[p];
            }
            CHECK(res[0] == cnt, "%s kmers k%u count", c.name.c_str(), k);
            uint64_t dg[4];
            blo_kmer_digest(s, c.offsets.data(), n_seqs, k, canon, 77, 1, 1, dg);
            emu_kmers(b, 0, 0, k, 77, (canon ? 1u : 0u) | 2u, nullptr, nullptr, nullptr, res);
            CHECK(res[0] == dg[0] && res[1] == dg[1] && res[2] == dg[2] && res[3] == dg[3], "%s kmers k%u c%d drop_last digest %llu vs %llu",
                  c.name.c_str(), k, canon, res[0], (unsigned long long)dg[0]);
        }
    }
    // hash_sampler over kmer_view (hash_sampler.hpp:136-141): units with hash < threshold, optionally minus the
    // k-mer the `it != cend()` idiom never reaches
    for (unsigned k : {5u, 21u, 31u}) {
        for (int canon = 0; canon < 2; ++canon) {
            blo_units(s, c.offsets.data(), n_seqs, k, canon, uv.data(), uok.data());
            for (uint64_t thr : {~0ULL, 1ULL << 62, 0ULL}) {
                for (int drop = 0; drop < 2; ++drop) {
                    std::vector<uint64_t> xp, xv;
                    for (size_t q = 0; q < n_seqs; ++q)
                        for (uint64_t pp = c.offsets[q]; pp + k <= c.offsets[q + 1]; ++pp)
                            if (uok[pp] && blo_hash64_u64(uv[pp], 9) < thr && !(drop && pp + k == c.offsets[q + 1])) { xp.push_back(pp); xv.push_back(uv[pp]); }
                    emu_hash_sample(b, 0, 0, k, 9, thr, (canon ? 1u : 0u) | (drop ? 2u : 0u), ev.data(), ep.data(), eh.data(), cap, res);
                    CHECK(res[0] == xp.size(), "%s hash_sample k%u c%d thr %llx drop %d: %llu vs %zu", c.name.c_str(), k, canon, (unsigned long long)thr, drop, res[0], xp.size());
                    if (res[0] == xp.size())
                        for (size_t i = 0; i < xp.size(); ++i) CHECK(ep[i] == xp[i] && ev[i] == xv[i], "%s hash_sample record %zu", c.name.c_str(), i);
                }
            }
        }
    }
    emu_batch_free(b);
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    std::mt19937_64 rng(12345);
    // hash spot check
    for (int i = 0; i < 1000; ++i) {
        uint64_t v = rng(), sd = rng();
        CHECK(emu_hash64(v, sd) == blo_hash64_u64(v, sd), "hash64");
    }
    const char alphabet_breaks[] = "NnRYK-*X\x80\xff@`bBfF";
    for (int round = 0; round < rounds; ++round) {
        const uint64_t sizes[] = {0, 1, 15, 16, 17, 31, 40, 41, 42, 150, 1000, 1007, 1008, 1009, 1024, 2016, 4079, 4080, 4081, 4096, 8063, 8064, 8065, 8160, 9000, 12345, 16129, 17000};
        for (uint64_t n : sizes) {
            for (int flavour = 0; flavour < 6; ++flavour) {
                Case c;
                c.seq.resize(n);
                c.read_len = 0;
                c.single = false;
                if (flavour == 4) {  // tie-heavy low-complexity DNA
                    const char* motifs[] = {"A", "AC", "ACG", "AAAT", "ACGTT", "GATTACA"};
                    const std::string mo = motifs[rng() % 6];
                    for (uint64_t i = 0; i < n; ++i) c.seq[i] = (uint8_t)mo[i % mo.size()];
                    for (uint64_t i = 0; i + 1 < n; i += 97 + rng() % 300) c.seq[i] = "ACGT"[rng() % 4];
                } else {
                    blo_synth(1000 * round + n + flavour, 0, n, reinterpret_cast<char*>(c.seq.data()));
                    if (flavour == 5)
                        for (uint64_t i = 0; i < n; ++i)
                            if (rng() % 3 == 0) c.seq[i] = (uint8_t)"acgtuU"[rng() % 6];  // case / U handling
                }
                if ((flavour == 1 || flavour == 3) && n)  // breaks, any position incl. first/last
                    for (uint64_t j = 0, nb = 1 + n / 200; j < nb; ++j) c.seq[rng() % n] = (uint8_t)alphabet_breaks[rng() % (sizeof(alphabet_breaks) - 1)];
                if (flavour == 0 || flavour == 1 || flavour == 4 || flavour == 5) {
                    if (rng() % 2) { c.offsets = {0, n}; c.single = true; c.name = "single"; }
                    else {
                        const uint64_t L = (const uint64_t[]){150, 41, 64, 10000, 1, 33}[rng() % 6];
                        c.offsets = fixed_offsets(n, L);
                        c.read_len = L;
                        c.name = "reads" + std::to_string(L);
                        if (L >= n) { c.single = true; c.read_len = 0; c.offsets = {0, n}; }
                    }
                } else {  // ragged, with empty and tiny sequences
                    c.offsets.push_back(0);
                    uint64_t p = 0;
                    while (p < n) {
                        const uint64_t kind = rng() % 6;
                        const uint64_t len = kind == 0 ? 0 : kind == 1 ? 1 + rng() % 40 : kind == 2 ? 41 + rng() % 200 : 1 + rng() % 3000;
                        p = p + len > n ? n : p + len;
                        c.offsets.push_back(p);
                    }
                    if (c.offsets.size() == 1) c.offsets.push_back(0);
                    c.name = "ragged";
                }
                c.name += "/n" + std::to_string(n) + "/f" + std::to_string(flavour) + "/r" + std::to_string(round);
                check_case(c, rng);
            }
        }
    }
    // batches of whole fixed-length reads: the read-tiled layout (bl_scan_frl.hpp) — clean, with breaks, tie-heavy,
    // read counts that fill tiles exactly / leave partial waves, sub-ranges cut at read boundaries (check_case cuts anywhere:
    // cuts inside a read take the position-tiled path, which must agree)
    for (int round = 0; round < rounds; ++round) {
        const uint64_t lens[] = {150, 150, 100, 250, 64, 76, 41, 151, 33, 300, 1000};
        for (uint64_t L : lens) {
            const uint64_t read_counts[] = {1, 7, 8, 31, 32, 33, 64, 65, 200};
            for (uint64_t nr : read_counts) {
                for (int flavour = 0; flavour < 3; ++flavour) {
                    if (L * nr > 40000) continue;
                    Case c;
                    const uint64_t n = L * nr;
                    c.seq.resize(n);
                    c.single = false;
                    if (flavour == 2) {
                        const char* motifs[] = {"A", "AC", "ACG", "AAAT", "ACGTT", "GATTACA"};
                        const std::string mo = motifs[rng() % 6];
                        for (uint64_t i = 0; i < n; ++i) c.seq[i] = (uint8_t)mo[i % mo.size()];
                        for (uint64_t i = 0; i + 1 < n; i += 97 + rng() % 300) c.seq[i] = "ACGT"[rng() % 4];
                    } else {
                        blo_synth(777 * round + n + flavour, 0, n, reinterpret_cast<char*>(c.seq.data()));
                    }
                    if (flavour == 1)
                        for (uint64_t q = 0, nb = 1 + n / 400; q < nb; ++q) c.seq[rng() % n] = (uint8_t)alphabet_breaks[rng() % (sizeof(alphabet_breaks) - 1)];
                    c.offsets = fixed_offsets(n, L);
                    c.read_len = L;
                    if (nr == 1) { c.single = true; c.read_len = 0; c.offsets = {0, n}; }
                    c.name = "frl" + std::to_string(L) + "x" + std::to_string(nr) + "/f" + std::to_string(flavour) + "/r" + std::to_string(round);
                    check_case(c, rng);
                    // cuts at read boundaries keep both halves read-tiled
                    if (nr >= 2) {
                        EmuBatch* b = emu_batch(c.seq.data(), n, nullptr, 0, L);
                        const uint64_t cut = L * (1 + rng() % (nr - 1)), cap = n + 2;
                        std::vector<uint64_t> ov(cap), op(cap), oh(cap), ev(cap), ep(cap), eh(cap);
                        unsigned long long r1[8], r2[8];
                        const size_t cnt = blo_minimizers(reinterpret_cast<const char*>(c.seq.data()), c.offsets.data(), c.offsets.size() - 1, 31, 11, 42, 1, 1,
                                                          ov.data(), op.data(), oh.data(), cap);
                        emu_minimizers(b, 0, cut, 31, 11, 42, 1, ev.data(), ep.data(), eh.data(), cap, r1);
                        emu_minimizers(b, cut, 0, 31, 11, 42, 1, ev.data() + r1[0], ep.data() + r1[0], eh.data() + r1[0], cap - r1[0], r2);
                        CHECK(r1[0] + r2[0] == cnt, "%s read-aligned split at %llu: %llu + %llu vs %zu", c.name.c_str(), (unsigned long long)cut, r1[0], r2[0], cnt);
                        if (r1[0] + r2[0] == cnt)
                            for (size_t i = 0; i < cnt; ++i)
                                CHECK(ev[i] == ov[i] && ep[i] == op[i] && eh[i] == oh[i], "%s read-aligned split record %zu", c.name.c_str(), i);
                        emu_batch_free(b);
                    }
                }
            }
        }
    }
    CHECK(emu_frl_scans() > 100, "the read-tiled path was hardly exercised: %d scans", emu_frl_scans());
    std::printf("read-tiled scans run: %d (tiles %d, of which decided again on the hashes: %d; tiles of closed-syncmer scans decided again: %d, of argmin syncmer scans with the exact form deferred: %d)\n", emu_frl_scans(), emu_frl_tiles(), emu_frl_redone(), emu_closed_redone(), emu_sy2_redone());
    CHECK(emu_sy2_redone() > 0, "no tile of a deferred-argmin syncmer scan was decided again: the tie path did not run");
    std::printf("tiles of the position-tiled minimizer scan on murmur64_top decided again: %d\n", emu_pos_redone());
    CHECK(emu_pos_redone() > 0, "no tile of the position-tiled approximate minimizer scan was decided again: the tie path did not run");

    if (g_fail) {
        std::printf("emu_selftest: %d mismatches\n", g_fail);
        return 1;
    }
    std::printf("emu_selftest: OK\n");
    return 0;
}
