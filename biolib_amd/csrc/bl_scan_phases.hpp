// bl_scan_phases.hpp — the tile pipeline of the fused scan, one function per phase.  The HIP kernels
// (bl_kernels.hip) call these with barriers in between; the CPU emulation harness (tests/emu/)
// calls the very same functions thread by thread.
//
// Tile geometry (all positions are indices into the batch's base buffer):
//   A workgroup tile is NWAVE independent WAVE TILES.  Wave `wv` of tile t hashes the WH = 1024 unit
//   start positions [wq0, wq0+1024), wq0 = origin + t*stride + wv*wstride (16-aligned); lane `l`
//   owns the S = 16 positions 16*l .. 16*l+15 of it — exactly the bases of one coalesced 16-byte load.
//   Only the first wstride = 1024 - 16*ceil(w/16) positions of a wave tile are OWNED (their records
//   are reported by this wave); the rest is halo that the next wave tile hashes again (1.6 % extra
//   work at w = 11), which makes a wave self-sufficient: hashes never leave registers, the halo a
//   lane needs comes from lanes l+1, l+2 by DPP wave shifts, and no barrier is needed until the
//   per-wave record counts are combined.
//   minimizer / super-k-mer modes: the lane that owns position i decides, from the argmins of
//   windows i and i+1, whether window i+1 starts a new minimizer occurrence and whether window i
//   ends one.
#pragma once
#include <type_traits>

#include "bl_scan_core.hpp"

namespace bl {

constexpr int NWAVE = TPB / 64;
constexpr int WH = 64 * S;      // positions hashed per wave tile
constexpr int WCHUNK = 64 + 8;  // 16-base chunks staged per wave tile (halo of up to 128 bases)

template <int MODE, int W>
struct TileShared {
    // staged once per workgroup tile, flat over the tile: wave wv's chunk c is entry wv*(wstride/16) + c
    // (consecutive wave tiles overlap by their halo, so the halo of a wave is the next wave's data)
    uint32_t codes[NCHUNK];         // 2-bit codes, 16 bases per dword, first base most significant
    uint32_t flags[NCHUNK];         // [15:0] good-base bits, [31:16] sequence-start bits (bit b = base b)
    // runtime-width kernels (W < 0), exact branch only.  Minimizer / super-k-mer scans: one dword of hash[s][tid] at a time
    // (high halves, then low halves: 16 KB, which lifts the LDS limit from 3 to 6 workgroups per CU).  The syncmer scan, whose
    // registers hold it to 2 waves per SIMD anyway and whose small s-mers tie often, keeps whole hashes (one pass).
    // The largest size group (W = -32) is register-bound at 3 waves per SIMD and keeps whole hashes too.
    typename std::conditional<(MODE == MODE_SYNCMER || W == -32), uint64_t, uint32_t>::type half[W < 0 ? S : 1][W < 0 ? TPB : 1];
    alignas(4) uint16_t list_a[H];  // compacted records: (wave << 12) | wave-relative argmin position
                                    // (syncmer mode: the k-mer's own position)
    alignas(4) uint16_t list_j[MODE == MODE_SUPERKMER ? H : 2];  // compacted: first window of the occurrence
    alignas(4) uint16_t list_e[MODE == MODE_SUPERKMER ? H : 2];  // compacted: last window of the occurrence
    uint32_t wave_tot[NWAVE];
    uint32_t redo;                  // closed-syncmer scans: some wave of the tile could not decide (equal high dwords)
    uint32_t wave_bad[NWAVE];       // read-tiled scans: wave wv staged at least one base that is not ACGTUacgtu
    unsigned long long dig[4];
};

struct ThreadState {
    uint64_t h[S];    // hashes of the owned positions (forward strand for syncmers)
    uint64_t h2[S];   // syncmer: reverse-strand s-mer hashes
    uint32_t emit;    // bit s: a record starts at window s+1 (minimizer modes) / window s is a syncmer
    uint32_t endm;    // bit s: an occurrence ends at window s (super-k-mer mode)
    uint32_t strand;  // syncmer: bit s set <=> reverse strand is canonical for the k-mer at s
    uint64_t apk0, apk1;  // minimizer modes: argmin (lane-relative element index) of window s+1 in byte s (8 per word)
    // read-tiled scans (bl_scan_frl.hpp): byte s of apk = argmin of window s itself
    int32_t lane_base;    // tile-relative base (from the tile's first staged chunk) of the lane's first unit
    int32_t jlane;        // index of the lane inside its read, -1: the lane has no read
    uint32_t a_first, a_last;  // argmin index of the lane's first / last window
    uint32_t vmask;       // bit s: window s exists and is valid
    uint32_t cw[3], rw[3];  // closed-syncmer scans: the lane's 48 bases as codes, and their reverse complement (first base of each in the top pair of word 0)
    uint32_t hmax;        // scans on murmur64_top: the largest of the lane's own values
    uint32_t hlow;        // closed-syncmer scans on murmur64_top<true>: the smallest of the lane's own forward values
    uint32_t occ;         // element-centric minimizer scans: bit e = element e (own or, from NS on, the next lane's) is the argmin of a valid window
};

// ------------------------------------------------------------------------------------------------
// Phase 1: coalesced 16-byte loads -> 2-bit codes + validity/start flags in LDS.
struct alignas(16) Vec16 {
    uint32_t x, y, z, w;
};

BL_DEV void stage_chunk(const ScanParams& p, uint32_t* codes, uint32_t* flags, int c, int64_t q0)
{
    const int64_t g = q0 + 16 * (int64_t)c;
    uint32_t d[4] = {0, 0, 0, 0};
    if (g >= 0 && g + 16 <= p.n_bases) {
        const Vec16 v = *reinterpret_cast<const Vec16*>(p.bases + g);  // one global_load_dwordx4 (a non-temporal load measured no faster)
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if (BL_COLD(g + 16 > 0 && g < p.n_bases)) {  // ragged edge: byte-wise, zeros (= breaks) outside
        for (int b = 0; b < 16; ++b) {
            const int64_t q = g + b;
            if (q >= 0 && q < p.n_bases) d[b >> 2] |= (uint32_t)p.bases[q] << (8 * (b & 3));
        }
    }
    uint32_t code, bad;
    encode16(d, code, bad);
    uint32_t start = 0;
    if (p.start_bits) {
        if (g >= 0 && g < p.n_bases) start = (p.start_bits[g >> 5] >> (g & 31)) & 0xffffu;
    } else if (g == 0) {
        start = 1;
    }
    codes[c] = code;
    flags[c] = (~bad & 0xffffu) | (start << 16);
}

// Pass 2 only needs the 2-bit codes (to rebuild unit values): no validity, no sequence starts.
BL_DEV Vec16 load_chunk(const ScanParams& p, int c, int64_t q0)
{
    const int64_t g = q0 + 16 * (int64_t)c;
    Vec16 v{0, 0, 0, 0};
    if (g >= 0 && g + 16 <= p.n_bases) {
        v = *reinterpret_cast<const Vec16*>(p.bases + g);
    } else if (g + 16 > 0 && g < p.n_bases) {
        uint32_t d[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; ++b) {
            const int64_t q = g + b;
            if (q >= 0 && q < p.n_bases) d[b >> 2] |= (uint32_t)p.bases[q] << (8 * (b & 3));
        }
        v = Vec16{d[0], d[1], d[2], d[3]};
    }
    return v;
}

BL_DEV uint32_t codes_of(const Vec16& v)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t code = 0;
    BL_UNROLL
    for (int i = 0; i < 4; ++i) {
        uint32_t x = (d[i] >> 1) & 0x03030303u;
        x ^= (x >> 1) & 0x01010101u;
        code = (code << 8) | ((x * 0x40100401u) >> 24);
    }
    return code;
}

// index of the thread's wave inside the workgroup, as a value the compiler knows to be wave-uniform (scalar registers
// and scalar arithmetic for everything derived from it: chunk bases, wave origins, list tags)
BL_DEV int wave_index(int tid)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    return __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    return tid >> 6;
#endif
}

// global position of the first hashed position of wave `wv` of the tile that starts at q0
BL_DEV int64_t wave_origin(const ScanParams& p, int64_t q0, int wv) { return q0 + (int64_t)wv * (p.stride / NWAVE); }

// first staged chunk of wave `wv` inside the flat per-tile arrays
BL_DEV int wave_chunk0(const ScanParams& p, int wv) { return wv * (p.stride / NWAVE / 16); }

// Chunks of the flat per-tile arrays that hold data some OWNED window depends on: the last wave's lanes read codes up to
// chunk lane+2 <= 65 and good/start bits up to position S + unit + w of their last owning lane.  For the three tuned
// configurations this is <= TPB chunks, so one staging pass per thread suffices (the generic bound WCHUNK needs a
// second pass that only 8 lanes of wave 0 use but the whole wave executes).  Entries beyond it keep stale LDS
// contents; only lanes that own nothing look at them.
BL_DEV int staged_chunks(const ScanParams& p)
{
    if (p.frl) return p.slot_chunks;  // read-tiled: every chunk of the tile's reads (bl_scan_frl.hpp)
    const int wchunks = p.stride / NWAVE / 16;
    int per_wave = (wchunks - 1) + (S + p.unit + p.w) / 16 + 1;
    per_wave = per_wave < 66 ? 66 : (per_wave > WCHUNK ? WCHUNK : per_wave);
    return wave_chunk0(p, NWAVE - 1) + per_wave;  // <= NCHUNK
}

template <int MODE, int W>
BL_DEV void phase_load(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0)
{
    const int needed = staged_chunks(p);
    if (tid < needed) stage_chunk(p, sh.codes, sh.flags, tid, q0);
    if (TPB + tid < needed) stage_chunk(p, sh.codes, sh.flags, TPB + tid, q0);  // a few lanes of wave 0 only
}

// good / start bit-vectors for the lane's positions i0 .. i0+127 (bit i = wave position 16*lane + i)
BL_DEV void gather_flags(const uint32_t* flags, int lane, Bits128& good, Bits128& start)
{
    uint64_t g[2] = {0, 0}, s[2] = {0, 0};
    BL_UNROLL
    for (int c = 0; c < 8; ++c) {
        const uint32_t f = flags[lane + c];  // lane + 7 <= 70 < WCHUNK
        g[c >> 2] |= (uint64_t)(f & 0xffffu) << (16 * (c & 3));
        s[c >> 2] |= (uint64_t)(f >> 16) << (16 * (c & 3));
    }
    good = Bits128{g[0], g[1]};
    start = Bits128{s[0], s[1]};
}

// n bases (n <= 16) from base `o` of the 48 bases a0:a1:a2 (16 per word, first base in the top pair), as an n-mer in the low 2n bits.
// o and n are compile-time constants where this is used: one v_bfe_u32 when the bases lie in one word, v_alignbit_b32 + shift
// otherwise — against the six instructions a rolling update of a forward and a reverse-complement register costs per base.
BL_DEV uint32_t bases_at(uint32_t a0, uint32_t a1, uint32_t a2, int o, int n)
{
    const int q = o >> 4, r = o & 15;
    const uint32_t hi = q == 0 ? a0 : (q == 1 ? a1 : a2), lo = q == 0 ? a1 : (q == 1 ? a2 : 0u);
    if (r + n <= 16) return n == 16 ? hi : (hi >> (32 - 2 * (r + n))) & ((1u << (2 * n)) - 1u);
    const uint32_t w16 = funnel_shr(hi, lo, 32 - 2 * r);  // the 16 bases from o on (r != 0 here)
    return n == 16 ? w16 : w16 >> (32 - 2 * n);
}

// reverse complement of 16 bases (one word of codes)
BL_DEV uint32_t revcomp16(uint32_t c)
{
    const uint32_t r = __builtin_bitreverse32(c);
    return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
}

// ------------------------------------------------------------------------------------------------
// Phase 2: roll the owned 16 units in registers and hash them.
// U: the unit length as a compile-time constant when it is one and fits a word (1..16; the BASELINE C4 kernel: 15-mers) — the
// units then come straight from the codes and from their reverse complement (bases_at: one or two instructions each, the
// canonical one by a single v_min_u32) instead of two rolling registers, a 64-bit compare and two selects.  0: rolling registers.
// GENERIC: the kernel takes the unit length at run time (64-bit rolling registers with run-time shifts).  Those kernels hash with the
// compiler's own multiply: the split one (mul64c) holds more register pairs alive, and compiled for five waves per SIMD they
// spilled into the hashing loop with it (unit 15, w 10 on 150-bp reads: 381 -> 218 Gbp/s until this was noticed).
// APPROX (minimizer scans, rolling registers): st.h[s] holds murmur64_top in its high dword (low dword 0) and st.hmax the largest of them, as in
// phase_hash_frl: the window phase works on those and reports what it cannot decide
template <int MODE, int W, int U = 0, bool GENERIC = false, bool APPROX = false>
BL_DEV void phase_hash(const ScanParams& p, TileShared<MODE, W>& sh, int tid, ThreadState& st)
{
    const int wv = wave_index(tid), lane = tid & 63;
    const uint32_t* wcodes = sh.codes + wave_chunk0(p, wv);
    const uint32_t c0 = wcodes[lane], c1 = wcodes[lane + 1], c2 = wcodes[lane + 2];
    Roller r;
    roller_start(r, c0, c1, c2, p.unit);
    if (MODE == MODE_SYNCMER) {
        // s-mers are substrings of the canonical K-MER, not canonical themselves (kmer_view.hpp:274-281):
        // hash the forward s-mer and its reverse complement; the k-mer's strand picks one later.
        Roller rk;
        roller_start(rk, c0, c1, c2, p.unit + p.w - 1);
        uint32_t strand = 0;
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            roller_step(r, s);
            roller_step(rk, s);
            st.h[s] = murmur64_plain(roller_fwd(r), p.seed);  // (the compiler's multiply here: see murmur64_plain)
            st.h2[s] = murmur64_plain(roller_rc(r), p.seed);
            if (p.canonical && roller_rc(rk) < roller_fwd(rk)) strand |= 1u << s;  // kmer_view.hpp:196
        }
        st.strand = strand;
    } else if (U >= 1 && U <= 16) {
        const uint32_t r0 = revcomp16(c2), r1 = revcomp16(c1), r2 = revcomp16(c0);  // base b of the lane = base 47 - b of r0:r1:r2
        uint32_t hmax = 0;
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            const uint32_t fw = bases_at(c0, c1, c2, s, U >= 1 && U <= 16 ? U : 1);
            uint32_t v = fw;
            if (p.canonical) {
                const uint32_t rv = bases_at(r0, r1, r2, 48 - (U >= 1 && U <= 16 ? U : 1) - s, U >= 1 && U <= 16 ? U : 1);
                v = rv < fw ? rv : fw;  // numeric minimum, minimizer_view.hpp:236-238
            }
            if (APPROX) {
                const uint32_t top = murmur64_top(v, p.seed);
                hmax = top > hmax ? top : hmax;
                st.h[s] = (uint64_t)top << 32;
            } else {
                st.h[s] = murmur64(v, p.seed);
            }
        }
        st.hmax = hmax;
    } else {
        uint32_t hmax = 0;
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            roller_step(r, s);
            const uint64_t fw = roller_fwd(r), rv = roller_rc(r);
            const uint64_t v = (p.canonical && rv < fw) ? rv : fw;  // minimizer_view.hpp:236-238
            if (APPROX) {
                const uint32_t top = murmur64_top(v, p.seed);
                hmax = top > hmax ? top : hmax;
                st.h[s] = (uint64_t)top << 32;
            } else {
                st.h[s] = GENERIC ? murmur64_plain(v, p.seed) : murmur64(v, p.seed);
            }
        }
        st.hmax = hmax;
    }
    (void)sh;  // the runtime-width kernels write hashes to LDS only in their exact branch (lane_window_argmin_generic)
}

// ------------------------------------------------------------------------------------------------
// Halo exchange: elements S .. S+NE-1 of a lane's window input are the first hashes of the lanes
// that follow it in the wave.  On the GPU this is a DPP wave shift (v_mov_b32_dpp wave_shl:1 —
// lane l reads lane l+1, lane 63 keeps its own value); the emulation reads the neighbour's state.
// The last ceil(NE/16) lanes of a wave receive garbage: they own no window (stride <= 1024 - w).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
BL_DEV uint64_t lane_next(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
#endif

template <int NE, bool SECOND>
BL_DEV void gather_halo(const ThreadState* all, int tid, const ThreadState& st, uint64_t* e)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    (void)tid;
    uint64_t n1[S];
    BL_UNROLL
    for (int x = 0; x < S; ++x) {
        if (x < NE || x < NE - 16 || x < NE - 32 || x < NE - 48) n1[x] = lane_next(SECOND ? st.h2[x] : st.h[x]);
        if (x < NE) e[S + x] = n1[x];
    }
    if (NE > 16) {
        uint64_t n2[S];
        BL_UNROLL
        for (int x = 0; x < S; ++x) {
            if (x < NE - 16 || x < NE - 32 || x < NE - 48) n2[x] = lane_next(n1[x]);
            if (x < NE - 16) e[2 * S + x] = n2[x];
        }
        if (NE > 32) {
            uint64_t n3[S];
            BL_UNROLL
            for (int x = 0; x < S; ++x) {
                if (x < NE - 32 || x < NE - 48) n3[x] = lane_next(n2[x]);
                if (x < NE - 32) e[3 * S + x] = n3[x];
            }
            if (NE > 48) {
                BL_UNROLL
                for (int x = 0; x < S; ++x)
                    if (x < NE - 48) e[4 * S + x] = lane_next(n3[x]);
            }
        }
    }
#else
    const int lane = tid & 63;
    for (int x = 0; x < NE; ++x) {
        const int nb = lane + 1 + (x >> 4);
        e[S + x] = nb < 64 ? (SECOND ? all[tid + 1 + (x >> 4)].h2[x & 15] : all[tid + 1 + (x >> 4)].h[x & 15]) : 0xDEADBEEFDEADBEEFull;
    }
    (void)st;
#endif
}

// Packed keys of the lanes that follow: a lane's own keys carry the tags 0..15 (63..48 when the rightmost wins), the
// element x of the lane `hop` lanes further on is this lane's element 16*hop + x, so its tag moves by 16 per hop —
// one DPP move and one add per element (the compiler may fuse them into v_add_u32_dpp).
template <int NE, bool LEFT>
BL_DEV void gather_halo_keys(uint32_t* key)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    uint32_t cur[S];
    BL_UNROLL
    for (int x = 0; x < S; ++x) cur[x] = key[x];
    BL_UNROLL
    for (int hop = 0; hop < (NE + S - 1) / S; ++hop) {
        BL_UNROLL
        for (int x = 0; x < S; ++x) {
            if (hop * S + x < NE || (hop + 1) * S + x < NE || (hop + 2) * S + x < NE || (hop + 3) * S + x < NE) {
                const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur[x], 0x130, 0xf, 0xf, true);
                cur[x] = LEFT ? nb + 16u : nb - 16u;
            }
            if (hop * S + x < NE) key[(hop + 1) * S + x] = cur[x];
        }
    }
#else
    (void)key;  // the emulation builds the keys from the neighbours' states (lane_window_argmin)
#endif
}

#ifdef BL_EXPERIMENT_COUNT_FALLBACK
__device__ unsigned long long bl_dbg_fallbacks;
#endif
// Window argmins of one lane: packed 32-bit keys first, the exact 64-bit form when a prefix tie was seen anywhere
// in the wave among lanes that own windows.
// DEFER: no exact form in here — a prefix tie in a lane that owns windows is reported through *tie, and the caller has the whole
// tile decided again by a kernel that carries the exact form (the syncmer scan: without it, and without the hashes' low dwords it
// would keep alive, the kernel fits three waves per SIMD instead of two).
// APPROX (with DEFER): the keys come from murmur64_top — two of them are told apart when their prefixes are two or more apart, and a lane that
// holds a key whose prefix could wrap reports a tie whether it owns windows or not (its keys are the halo of lanes that do)
template <int NW, int W, bool LEFT, bool SECOND, bool RAW = false, bool DEFER = false, bool APPROX = false>
BL_DEV void lane_window_argmin(const ThreadState* all, int tid, const ThreadState& st, bool owns, uint32_t* a, bool* tie = nullptr)
{
    uint32_t key[S + W];
    BL_UNROLL
    for (int s = 0; s < S; ++s) key[s] = packed_key((uint32_t)((SECOND ? st.h2[s] : st.h[s]) >> 32), s, LEFT);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    gather_halo_keys<W, LEFT>(key);
#else
    {
        const int lane = tid & 63;
        for (int x = 0; x < W; ++x) {
            const int nb = lane + 1 + (x >> 4);
            const uint32_t hi = nb < 64 ? (uint32_t)((SECOND ? all[tid + 1 + (x >> 4)].h2[x & 15] : all[tid + 1 + (x >> 4)].h[x & 15]) >> 32) : 0xDEADBEEFu;
            key[S + x] = packed_key(hi, S + x, LEFT);
        }
    }
#endif
    static_assert(!APPROX || DEFER, "keys from murmur64_top: no exact form in here");
    const uint32_t dmin = window_argmin_packed<NW, W, LEFT, RAW && LEFT, APPROX>(key, a);
    if (DEFER) {
        if ((owns && dmin < (APPROX ? 128u : 64u)) || (APPROX && st.hmax >= 0xffffffc0u)) *tie = true;
        return;
    }
    if (BL_COLD(wave_any(owns && dmin < 64u))) {
#ifdef BL_EXPERIMENT_COUNT_FALLBACK
        if ((tid & 63) == 0) atomicAdd(&bl_dbg_fallbacks, 1ull);
#endif
        uint64_t e[S + W];
        BL_UNROLL
        for (int s = 0; s < S; ++s) e[s] = SECOND ? st.h2[s] : st.h[s];
        gather_halo<W, SECOND>(all, tid, st, e);
        window_argmin<NW, W, LEFT>(e, a);
    }
}

// bit s set <=> lo <= s < hi, for s in 0..S
BL_DEV uint32_t range_mask(int64_t lo, int64_t hi)
{
    const int l = lo < 0 ? 0 : (lo > S + 1 ? S + 1 : (int)lo);
    const int h = hi < 0 ? 0 : (hi > S + 1 ? S + 1 : (int)hi);
    return h > l ? (((1u << h) - 1) & ~((1u << l) - 1)) : 0u;
}

// bit s: lane owns wave position 16*lane + s
BL_DEV uint32_t owned_mask(const ScanParams& p, int lane)
{
    const int own = p.stride / NWAVE - 16 * lane;
    return own >= S ? 0xffffu : (own > 0 ? (1u << own) - 1 : 0u);
}

// Exact argmins of the lane's NW windows of run-time size w from the wave's hashes, through LDS, one 32-bit half at a
// time so that the buffer is 4 KB per wave: pass 1 publishes the high dwords and finds, per window, the minimum high
// dword and the set of positions that attain it (a w-bit mask); pass 2 publishes the low dwords into the SAME buffer and
// picks, among those positions, the minimum low dword — leftmost or rightmost on a full tie.  Wave-local: the lanes of a
// wave run in lockstep, so every read of pass 1 precedes the writes of pass 2 (the emulation passes all[]).
// The window count is a template parameter: a[] is indexed statically and stays in the registers the fast form left it in.
template <int MODE, int W, bool LEFT, bool SECOND, int NW>
BL_DEV void window_argmin_lds_exact(TileShared<MODE, W>& sh, const ThreadState* all, int tid, const ThreadState& st, int w, uint32_t* a)
{
    const int wbase = tid & ~63, lane = tid & 63;
    if (MODE == MODE_SYNCMER || W == -32) {  // whole hashes in LDS: one pass
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        BL_UNROLL
        for (int s = 0; s < S; ++s) sh.half[s][W < 0 ? tid : 0] = SECOND ? st.h2[s] : st.h[s];
#else
        for (int t = wbase; t < wbase + 64; ++t)
            for (int s = 0; s < S; ++s) sh.half[s][W < 0 ? t : 0] = SECOND ? all[t].h2[s] : all[t].h[s];
#endif
        BL_UNROLL
        for (int i = 0; i < NW; ++i) {
            uint64_t best = 0;
            int arg = 0;
            for (int x = 0; x < w; ++x) {
                int pos = 16 * lane + i + x;
                pos = pos < WH ? pos : WH - 1;
                const uint64_t v = sh.half[pos & 15][W < 0 ? wbase + (pos >> 4) : 0];
                if (x == 0 || (LEFT ? v < best : v <= best)) { best = v; arg = i + x; }
            }
            a[i] = (uint32_t)arg;
        }
        return;
    }
    using Mask = typename std::conditional<(W == -8 || W == -16), uint32_t, unsigned long long>::type;  // w <= 32 fits a dword
    Mask cand[NW];
    for (int half = 0; half < 2; ++half) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        (void)all;
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            const uint64_t h = SECOND ? st.h2[s] : st.h[s];
            sh.half[s][W < 0 ? tid : 0] = (uint32_t)(half == 0 ? h >> 32 : h);
        }
#else
        (void)st;
        for (int t = wbase; t < wbase + 64; ++t)
            for (int s = 0; s < S; ++s) {
                const uint64_t h = SECOND ? all[t].h2[s] : all[t].h[s];
                sh.half[s][W < 0 ? t : 0] = (uint32_t)(half == 0 ? h >> 32 : h);
            }
#endif
        BL_UNROLL
        for (int i = 0; i < NW; ++i) {
            uint32_t best = 0;
            Mask mask = 0;
            int arg = 0;
            for (int x = 0; x < w; ++x) {
                int pos = 16 * lane + i + x;
                pos = pos < WH ? pos : WH - 1;  // beyond the wave tile: never part of an owned window
                const uint32_t v = (uint32_t)sh.half[pos & 15][W < 0 ? wbase + (pos >> 4) : 0];
                if (half == 0) {
                    if (x == 0 || v < best) { best = v; mask = (Mask)1 << x; }
                    else if (v == best) mask |= (Mask)1 << x;
                } else if ((cand[i] >> x) & 1) {
                    const bool first = ((cand[i] & (((Mask)1 << x) - 1)) == 0);
                    if (first || (LEFT ? v < best : v <= best)) { best = v; arg = i + x; }
                }
            }
            if (half == 0) cand[i] = mask;
            else a[i] = (uint32_t)arg;
        }
    }
}

// Runtime window size (kernels with W < 0): packed keys of the lane's own
// 16 hashes and of the 2P that follow (DPP hops, as in the templated form; 7-bit tags) go through
// window_argmin_doubling<P> in registers.  A prefix tie in an owning lane sends the wave through the exact scan: only
// then are the wave's hashes written to LDS (wave-local region, no workgroup barrier) for window_argmin_lds_exact (4 KB per wave, one hash half at a time).
// LEFT = false (rightmost wins, tags 127 - x) and SECOND (st.h2) serve the reverse-strand pass of the syncmer scan, whose
// ThreadState keeps both hash arrays alive anyway; the minimizer scans (REDO) drop st.h after packing and recompute it in
// the rare exact branch.  a[] = argmin positions (plain indices) for the syncmer callers, raw keys for RAW.
template <int MODE, int W, int NW, int P, bool LEFT, bool SECOND, bool RAW>
BL_DEV void lane_window_argmin_generic(const ScanParams& p, TileShared<MODE, W>& sh, const ThreadState* all, int tid, ThreadState& st, int w, bool owns,
                                       uint32_t* a)
{
    constexpr int NE = NW + 2 * P - 1, NH = NE - S;  // NH halo elements
    uint32_t key[NE];
    BL_UNROLL
    for (int s = 0; s < S; ++s) key[s] = packed_key7((uint32_t)((SECOND ? st.h2[s] : st.h[s]) >> 32), LEFT ? s : 127 - s);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    {
        uint32_t cur[S];
        BL_UNROLL
        for (int x = 0; x < S; ++x) cur[x] = key[x];
        BL_UNROLL
        for (int hop = 0; hop * S < NH; ++hop) {
            BL_UNROLL
            for (int x = 0; x < S; ++x) {
                if (hop * S + x < NH || (hop + 1) * S + x < NH || (hop + 2) * S + x < NH || (hop + 3) * S + x < NH || (hop + 4) * S + x < NH) {
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur[x], 0x130, 0xf, 0xf, true);
                    cur[x] = LEFT ? nb + 16u : nb - 16u;
                }
                if (hop * S + x < NH) key[(hop + 1) * S + x] = cur[x];
            }
        }
    }
#else
    {
        const int lane = tid & 63;
        for (int x = 0; x < NH; ++x) {
            const int nb = lane + 1 + (x >> 4);
            const uint64_t hv = nb < 64 ? (SECOND ? all[tid + 1 + (x >> 4)].h2[x & 15] : all[tid + 1 + (x >> 4)].h[x & 15]) : 0xDEADBEEFDEADBEEFull;
            key[S + x] = packed_key7((uint32_t)(hv >> 32), LEFT ? S + x : 127 - (S + x));
        }
    }
#endif
    BL_UNROLL
    for (int x = NW + P; x < NE; ++x)  // beyond the last wanted element (runtime: w): pads that never win and never tie
        if (x >= NW + w - 1) key[x] = pad_key7(x);
    const uint32_t dmin = window_argmin_doubling<NW, P>(key, w, a);
    if (!RAW) {
        BL_UNROLL
        for (int i = 0; i < NW; ++i) a[i] = LEFT ? (a[i] & 127u) : 127u - (a[i] & 127u);
    }
    if (BL_COLD(wave_any(owns && dmin < 128u))) {
#ifdef BL_EXPERIMENT_COUNT_FALLBACK
        if ((tid & 63) == 0) atomicAdd(&bl_dbg_fallbacks, 1ull);
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        // minimizer scans dropped the full hashes after packing (they would pin 32 registers through the fast path)
        if (MODE != MODE_SYNCMER) phase_hash<MODE, W, 0, true>(p, sh, tid, st);
#endif
        window_argmin_lds_exact<MODE, W, LEFT, SECOND, NW>(sh, all, tid, st, w, a);
    }
}

// picks the doubling width for a run-time window size inside the size group of the kernel (W = -8 / -16 / -32)
template <int MODE, int W, int NW, bool LEFT, bool SECOND, bool RAW>
BL_DEV void window_argmin_runtime(const ScanParams& p, TileShared<MODE, W>& sh, const ThreadState* all, int tid, ThreadState& st, int w, uint32_t* a)
{
    const bool owns = owned_mask(p, tid & 63) != 0;
    if (W == -8) {
        if (w <= 2) lane_window_argmin_generic<MODE, W, NW, 1, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
        else if (w <= 4) lane_window_argmin_generic<MODE, W, NW, 2, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
        else if (w <= 8) lane_window_argmin_generic<MODE, W, NW, 4, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
        else lane_window_argmin_generic<MODE, W, NW, 8, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
    } else if (W == -16) {
        lane_window_argmin_generic<MODE, W, NW, 16, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
    } else {
        lane_window_argmin_generic<MODE, W, NW, 32, LEFT, SECOND, RAW>(p, sh, all, tid, st, w, owns, a);
    }
}


// position-tiled minimizer scans with a window width in registers and an element mask that fits a dword
template <int MODE, int W>
constexpr bool pos_occ_form() { return MODE == MODE_MINIMIZER && W >= 2 && W <= 16; }

// ------------------------------------------------------------------------------------------------
// Phase 3 (minimizer / super-k-mer): window argmins, validity, start/end decisions.
// Returns the packed per-thread counts: starts | ends << 16.
// APPROX: the hashes are murmur64_top values (phase_hash<..., APPROX>); *tie is set where a lane could not tell two keys apart, and the
// caller has the tile decided again on the hashes themselves
template <int MODE, int W, bool APPROX = false>
BL_DEV uint32_t phase_window(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0, ThreadState& st,
                             const ThreadState* all, bool* tie = nullptr)
{
    const int wv = wave_index(tid), lane = tid & 63;
    const int w = W > 0 ? W : p.w;
    uint32_t a[S + 1];
    uint32_t below = 0x1ffffu;  // w = 1 with a hash threshold (hash_sampler): bit s = hash of unit s is below it
    if (W > 1) {
        lane_window_argmin<S + 1, (W > 1 ? W : 2), true, false, true, APPROX, APPROX>(all, tid, st, owned_mask(p, tid & 63) != 0, a, tie);
    } else if (W == 1) {
        uint64_t e[S + 1];
        BL_UNROLL
        for (int s = 0; s < S; ++s) e[s] = st.h[s];
        gather_halo<1, false>(all, tid, st, e);
        BL_UNROLL
        for (int s = 0; s <= S; ++s) a[s] = (uint32_t)s;
        if (p.use_threshold) {
            below = 0;
            BL_UNROLL
            for (int s = 0; s <= S; ++s)
                if (e[s] < p.hash_below) below |= 1u << s;
        }
    } else if (w >= 2 && W < 0) {
        // runtime w: W = -8 : w <= 16, W = -16 : 17..32, W = -32 : 33..64 (launch_count_mode picks the kernel)
        window_argmin_runtime<MODE, W, S + 1, true, false, true>(p, sh, all, tid, st, w, a);
    }
    Bits128 good, start;
    gather_flags(sh.flags + wave_chunk0(p, wv), lane, good, start);
    uint32_t valid = window_valid_mask(good, start, p.unit + w - 1);
    // windows outside the requested range: never reported; in super-k-mer mode they also cut groups
    const int64_t j0 = wave_origin(p, q0, wv) + 16 * (int64_t)lane;  // global position of the lane's window 0
    if (W == 1) {  // a plain list of units (k-mers): optional hash threshold and the reference idiom's dropped last k-mer
        valid &= below;
        if (p.drop_last) {
            uint32_t last = (uint32_t)b128_shr(start, p.unit).lo & 0x1ffffu;  // a sequence starts right after unit s
            const int64_t s_end = p.n_bases - p.unit - j0;                    // ... or the batch ends there
            if (s_end >= 0 && s_end <= S) last |= 1u << s_end;
            valid &= ~last;
        }
    }
    // all but the first and last tiles of a range lie inside it: decide that once per wave, with scalar arithmetic
    const int64_t wj0 = wave_origin(p, q0, wv);
    const bool inside = wj0 >= p.win_first && wj0 + WH + 1 <= p.win_end;
    uint32_t inrange = 0x1ffffu;
    if (!inside) inrange = range_mask(p.win_first - j0, p.win_end - j0);
    if (MODE == MODE_SUPERKMER) valid &= inrange;
    if (pos_occ_form<MODE, W>()) {
        // ELEMENT-CENTRIC decisions (see bl_scan_frl.hpp: a minimizer occurrence is an element that some valid window chooses; the
        // windows that choose it are consecutive and argmins never move left).  The lane reports the windows 1..16 it owns: one bit
        // per window into a mask indexed by element (element = low bits of the packed minimum; at most 16 + W - 1 <= 31).  An
        // element whose run of windows began BEFORE the first window this wave reports — the window in front of lane 0's, or one in
        // front of the scanned range — is a continuation, not a new occurrence: the bit of that window's argmin is cleared.
        const uint32_t owned = owned_mask(p, lane);
        const uint32_t cnt = (valid >> 1) & owned & (inrange >> 1) & 0xffffu;  // bit s: window s + 1 is this lane's to report
        uint32_t occ = 0;
        if (BL_COLD(wave_any(cnt != 0xffffu && cnt != 0u))) {
            BL_UNROLL
            for (int s = 0; s < S; ++s) occ |= ((cnt >> s) & 1u) << (a[s + 1] & 31u);
        } else {  // every lane reports all of its windows or none (the wave's last lane owns none)
            BL_UNROLL
            for (int s = 0; s < S; ++s) occ |= 1u << (a[s + 1] & 31u);
            occ = cnt ? occ : 0u;
        }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        const uint32_t prev_cnt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x138, 0xf, 0xf, true);  // wave_shr:1, lane 0 gets 0
#else
        const uint32_t prev_cnt = lane > 0 ? all[tid - 1].vmask : 0u;
#endif
        const uint32_t pc = (cnt << 1) | ((prev_cnt >> 15) & 1u);  // bit s: window s is reported by this wave (window 0: by the lane before)
        const uint32_t edge = valid & ~pc & (pc >> 1);              // bit s: window s is valid, not reported here, and window s + 1 is
        uint32_t kill = (edge & 1u) << (a[0] & 31u);
        if (BL_COLD(wave_any((edge >> 1) != 0u))) {                 // (only at the front of a range that does not start its batch)
            BL_UNROLL
            for (int s = 1; s < S; ++s) kill |= ((edge >> s) & 1u) << (a[s] & 31u);
        }
        // (the windows that choose an element sit in its own lane and in the one before: a cleared element is cleared in both)
        st.vmask = cnt;
        st.occ = occ;
        st.a_first = kill;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        const uint32_t prev_occ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)occ, 0x138, 0xf, 0xf, true);
        const uint32_t prev_kill = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kill, 0x138, 0xf, 0xf, true);
#else
        const uint32_t prev_occ = lane > 0 ? all[tid - 1].occ : 0u, prev_kill = lane > 0 ? all[tid - 1].a_first : 0u;
#endif
        st.emit = ((occ & 0xffffu) | (prev_occ >> 16)) & ~((kill & 0xffffu) | (prev_kill >> 16));  // bit s: the lane's own position s is a minimizer occurrence
        st.endm = 0;
        return (uint32_t)__builtin_popcount(st.emit);
    }
    // a[s] holds the argmin of window s in its low 6 bits (the packed form leaves hash bits above them).  Four at a
    // time: apk[j] = bytes a[4j+1..4j+4] (what the list phase reads), prv[j] = bytes a[4j..4j+3]; differ bit s = the two
    // bytes at s disagree = argmin of window s+1 is a different occurrence than that of window s.
    constexpr uint32_t IDX = W > 1 ? 0x3fu : 0x7fu;  // templated windows: 6-bit tags under hash bits; runtime w: plain indices < 16 + 64
    uint32_t apk[4], differ = 0;
    BL_UNROLL
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo2 = byte_perm(a[4 * j + 2], a[4 * j + 1], 0x0c0c0400u);  // byte 0 <- a[4j+1], byte 1 <- a[4j+2]
        const uint32_t hi2 = byte_perm(a[4 * j + 4], a[4 * j + 3], 0x04000c0cu);  // byte 2 <- a[4j+3], byte 3 <- a[4j+4]
        apk[j] = (lo2 | hi2) & (IDX * 0x01010101u);
    }
    BL_UNROLL
    for (int j = 0; j < 4; ++j) {
        const uint32_t prv = j ? funnel_shr(apk[j], apk[j - 1], 24) : ((apk[0] << 8) | (a[0] & IDX));
        const uint32_t x = apk[j] ^ prv;                                    // bytes < 128
        const uint32_t nz = ((x + 0x7f7f7f7fu) >> 7) & 0x01010101u;         // 1 in every non-zero byte
        differ |= ((nz * 0x01020408u) >> 24) << (4 * j);                    // byte b -> bit b
    }
    st.apk0 = ((uint64_t)apk[1] << 32) | apk[0];  // register pairs: no data movement
    st.apk1 = ((uint64_t)apk[3] << 32) | apk[2];
    const uint32_t v0 = valid & 0xffffu, v1 = (valid >> 1) & 0xffffu;
    const uint32_t owned = owned_mask(p, lane);
    st.emit = v1 & (~v0 | differ) & owned & ((inrange >> 1) & 0xffffu);  // window s+1 starts an occurrence
    st.endm = MODE == MODE_SUPERKMER ? (v0 & (~v1 | differ) & owned) : 0;  // window s ends one
    return (uint32_t)__builtin_popcount(st.emit) | ((uint32_t)__builtin_popcount(st.endm) << 16);
}

// Phase 3 (syncmer): leftmost minimum over the forward s-mer hashes, rightmost over the reverse ones.
template <int MODE, int W, bool DEFER = false>
BL_DEV void phase_sync_fwd(const ScanParams& p, TileShared<MODE, W>& sh, int tid, ThreadState& st, const ThreadState* all,
                           uint32_t* af, bool* tie = nullptr)
{
    if (W < 0 && p.w >= 2) {
        window_argmin_runtime<MODE, W, S, true, false, false>(p, sh, all, tid, st, p.w, af);
    } else if (W > 1) {
        lane_window_argmin<S, (W > 1 ? W : 2), true, false, false, DEFER>(all, tid, st, owned_mask(p, tid & 63) != 0, af, tie);
    } else if (W == 1) {
        BL_UNROLL
        for (int s = 0; s < S; ++s) af[s] = (uint32_t)s;
    }
}

template <int MODE, int W, bool DEFER = false>
BL_DEV uint32_t phase_sync_rev(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0, ThreadState& st,
                               const ThreadState* all, const uint32_t* af, bool* tie = nullptr)
{
    const int wv = wave_index(tid), lane = tid & 63;
    const int w = W > 0 ? W : p.w;
    const int k = p.unit + w - 1;
    uint32_t ar[S + 1];
    if (p.canonical) {
        if (W < 0 && w >= 2) {
            window_argmin_runtime<MODE, W, S, false, true, false>(p, sh, all, tid, st, w, ar);
        } else if (W > 1) {
            lane_window_argmin<S, (W > 1 ? W : 2), false, true, false, DEFER>(all, tid, st, owned_mask(p, tid & 63) != 0, ar, tie);
        } else if (W == 1) {
            BL_UNROLL
            for (int s = 0; s < S; ++s) ar[s] = (uint32_t)s;
        }
    }
    Bits128 good, start;
    gather_flags(sh.flags + wave_chunk0(p, wv), lane, good, start);
    const uint32_t valid = window_valid_mask(good, start, k) & 0xffffu;
    const int64_t j0 = wave_origin(p, q0, wv) + 16 * (int64_t)lane;
    const int64_t wj0 = wave_origin(p, q0, wv);
    uint32_t inrange = 0x1ffffu;
    if (!(wj0 >= p.win_first && wj0 + WH + 1 <= p.win_end)) inrange = range_mask(p.win_first - j0, p.win_end - j0);
    const uint32_t keepable = valid & owned_mask(p, lane) & inrange;
    uint32_t emit = 0;
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        // canonical k-mer on the reverse strand: its j-th m-mer from the left is the reverse complement
        // of the forward m-mer at W-1-j, and "leftmost" becomes "rightmost" 
        int off = af[s] - s;
        if (p.canonical && ((st.strand >> s) & 1)) off = (w - 1) - (ar[s] - s);
        const bool hit = off == p.soff || off == p.eoff;  // syncmer_sampler.hpp:130-137
        bool keep = hit && ((keepable >> s) & 1);
        if (keep && p.drop_last) {  // the k-mer that ends its sequence is never examined by the idiom (Q1)
            const int nxt = s + k;  // < 128
            const bool seq_end = (nxt < 64 ? (start.lo >> nxt) : (start.hi >> (nxt - 64))) & 1;
            keep = !(seq_end || j0 + s + k >= p.n_bases);
        }
        if (keep) emit |= 1u << s;
    }
    st.emit = emit;
    st.endm = 0;
    return (uint32_t)__builtin_popcount(emit);
}

// ------------------------------------------------------------------------------------------------
// CLOSED syncmers — offsets {0, W - 1}, the BASELINE C5 configuration (k = 31, s = 11, offsets 0 and 20).  Whether the
// minimum s-mer sits at the k-mer's first or last position needs no argmin, only the minimum VALUE of the other W - 1 s-mers:
//   forward strand canonical (leftmost minimum wins, kmer_view.hpp:274-281):
//       offset 0      <=>  Hf[p] <= min Hf[p+1 .. p+W-1]          offset W-1  <=>  Hf[p+W-1] <  min Hf[p .. p+W-2]
//   reverse strand canonical (positions mirror, the leftmost becomes the rightmost):
//       offset 0      <=>  Hr[p+W-1] <= min Hr[p .. p+W-2]        offset W-1  <=>  Hr[p] <  min Hr[p+1 .. p+W-1]
// Either pair of conditions is ONE comparison (closed_hits): the smaller of the two END s-mers against the minimum of the W - 2
// between.  So one sliding minimum of width W - 2 per strand over the hashes' HIGH DWORDS (no position tags, one v_min_u32 per
// step, no tie bookkeeping inside the windows) and two comparisons per k-mer decide.  A comparison whose two dwords are EQUAL
// (or, on approximate dwords, less than 2 apart) is undecided; the tile is then decided again in the exact argmin form
// (scan_redo_kernel: phase_sync_fwd / phase_sync_rev).

// Phase 2 of the closed-syncmer scan (compile-time s-mer length U <= 16): both s-mers of every position straight from the codes
// and from their reverse complement — no rolling registers, and no k-mer register at all: which strand of the K-MER is canonical
// is decided later from its first 16 bases on either strand (phase_sync_closed).
// BOTH (the argmin form with its exact part deferred, count_tile SY = 2): both strands' hashes whole, and the k-mers' strands here,
// from the 16 leading bases of either strand; *tie is set where those are equal (the tile is then decided again, exactly).
// AP (closed syncmers): murmur64_top<true> in place of the hashes' high dwords — the true dword is that value or one below it, so a
// comparison of two of them stands when they are two or more apart (phase_sync_closed); st.hlow keeps the smallest, because a
// value of 0 may stand for a true dword of 0xffffffff.
template <int MODE, int W, int U, bool BOTH = false, bool AP = false>
BL_DEV void phase_hash_closed(const ScanParams& p, TileShared<MODE, W>& sh, int tid, ThreadState& st, bool* tie = nullptr)
{
    static_assert(!(AP && BOTH), "the argmin form works on whole hashes");
    uint32_t hlow = ~0u;
    static_assert(U >= 1 && U <= 16, "s-mers of the closed-syncmer kernel fit one word");
    const int wv = wave_index(tid), lane = tid & 63;
    const uint32_t* wcodes = sh.codes + wave_chunk0(p, wv);
    const uint32_t c0 = wcodes[lane], c1 = wcodes[lane + 1], c2 = wcodes[lane + 2];
    const uint32_t r0 = revcomp16(c2), r1 = revcomp16(c1), r2 = revcomp16(c0);  // base b of the lane = base 47 - b of r0:r1:r2
    st.cw[0] = c0; st.cw[1] = c1; st.cw[2] = c2;
    st.rw[0] = r0; st.rw[1] = r1; st.rw[2] = r2;
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        const uint32_t fw = bases_at(c0, c1, c2, s, U);            // bases s .. s+U-1
        const uint32_t rv = bases_at(r0, r1, r2, 48 - U - s, U);   // their reverse complement
        if (AP) {
            const uint32_t top = murmur64_top<true>(fw, p.seed);
            hlow = top < hlow ? top : hlow;
            st.h[s] = (uint64_t)top << 32;
        } else {
            st.h[s] = murmur64(fw, p.seed);
        }
        if ((s & 3) == 3) BL_SCHED_FENCE();
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        if (BOTH) st.h2[s] = murmur64(rv, p.seed);
        else (void)rv;  // the reverse-strand s-mers are hashed where their minima are taken (phase_sync_closed): one strand's dwords alive at a time
#else
        st.h2[s] = AP ? (uint64_t)murmur64_top<true>(rv, p.seed) << 32 : murmur64(rv, p.seed);
#endif
    }
    st.hlow = hlow;
    uint32_t strand = 0;
    if (BOTH && p.canonical) {  // reverse strand canonical <=> rc < fwd (kmer_view.hpp:196), read off the 16 leading bases of each
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            const uint32_t f16 = bases_at(c0, c1, c2, s, 16);
            const uint32_t r16 = bases_at(r0, r1, r2, 48 - (U + W - 1) - s, 16);
            if (r16 < f16) strand |= 1u << s;
            if (r16 == f16) *tie = true;
        }
    }
    st.strand = strand;
}

// m[i] = min(key[i .. i+WW-1]) for i < NW, over key[0 .. NW+WW-2] (van Herk / Gil-Werman on values)
template <int NW, int WW>
BL_DEV void window_min(const uint32_t* key, uint32_t* m)
{
    BL_UNROLL
    for (int base = 0; base < NW; base += WW) {
        uint32_t sv[WW];
        sv[WW - 1] = key[base + WW - 1];
        BL_UNROLL
        for (int i = WW - 2; i >= 0; --i) sv[i] = key[base + i] < sv[i + 1] ? key[base + i] : sv[i + 1];
        m[base] = sv[0];
        uint32_t pv = 0;
        BL_UNROLL
        for (int i = 1; i < WW; ++i) {
            if (base + i >= NW) break;
            const uint32_t x = key[base + WW - 1 + i];
            pv = (i == 1 || x < pv) ? x : pv;
            m[base + i] = pv < sv[i] ? pv : sv[i];
        }
    }
}

// One strand's verdicts on the lane's 16 k-mers from its dwords key[0 .. S+W-2]: bit s = the smaller of k-mer s's two END s-mers
// lies below every s-mer between them.  That is the whole closed-syncmer test on either strand: with e = min(first, last) and
// mid = min(the W - 2 between), e < mid makes one end the minimum — the first if it is not above the last (leftmost wins, offset
// 0), else the last, strictly (offset W - 1); mirrored on the reverse strand — and e > mid makes neither.  e against mid is also the
// ONLY comparison that can be too close to call (two ends that tie below mid are a hit whichever way the tie goes): its
// distance folds into `closest` (xor: 0 = equal dwords; AP: absolute difference, < 2 = undecided).  Per k-mer and strand: one
// v_min_u32, one compare whose bit shifts in through an add-with-carry, one distance, half a v_min3_u32 — the two-comparison
// form (first <= min of the rest || last < min of the rest) cost 7.
template <int W, bool AP>
BL_DEV uint32_t closed_hits(const uint32_t* key, uint32_t& closest)
{
    if (W == 2) return 0xffffu;  // two s-mers: one of them is the minimum
    constexpr int WM = W > 2 ? W - 2 : 1;
    uint32_t mid[S];
    window_min<S, WM>(key + 1, mid);
    uint32_t hit = 0, d_prev = 0;
    BL_UNROLL
    for (int s = S - 1; s >= 0; --s) {  // last k-mer first: every bit enters at the bottom and the earlier ones move up
        const uint32_t e = key[s] < key[s + W - 1] ? key[s] : key[s + W - 1];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        // hit = 2 * hit + (e < mid): the compare's carry straight into an add-with-carry (the compiler writes v_cndmask + v_lshl_or)
        asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(hit) : "v"(e), "v"(mid[s]) : "vcc");
#else
        hit = hit + hit + (e < mid[s] ? 1u : 0u);
#endif
        const uint32_t d = AP ? abs_diff(e, mid[s]) : e ^ mid[s];
        if (s & 1) d_prev = d;
        else fold_min3(closest, d, d_prev);  // (opaque to the optimizer, which would turn the chain into a tree with every leaf alive)
    }
    return hit;
}

// key[S .. S+NE-1] = the first NE values of the lanes that follow (16 per hop); the last lanes of a wave get values that no
// owned k-mer looks at
template <int NE, bool SECOND>
BL_DEV void gather_halo_hi(const ThreadState* all, int tid, uint32_t* key)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    (void)tid;
    uint32_t cur[S];
    BL_UNROLL
    for (int x = 0; x < S; ++x) cur[x] = key[x];
    BL_UNROLL
    for (int hop = 0; hop * S < NE; ++hop) {
        BL_UNROLL
        for (int x = 0; x < S; ++x) {
            if (hop * S + x < NE || (hop + 1) * S + x < NE || (hop + 2) * S + x < NE || (hop + 3) * S + x < NE)
                cur[x] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur[x], 0x130, 0xf, 0xf, true);  // wave_shl:1
            if (hop * S + x < NE) key[(hop + 1) * S + x] = cur[x];
        }
    }
#else
    const int lane = tid & 63;
    for (int x = 0; x < NE; ++x) {
        const int nb = lane + 1 + (x >> 4);
        key[S + x] = nb < 64 ? (uint32_t)((SECOND ? all[tid + 1 + (x >> 4)].h2[x & 15] : all[tid + 1 + (x >> 4)].h[x & 15]) >> 32) : 0xDEADBEEFu;
    }
#endif
}

// Returns the number of syncmers the lane reports; `undecided` = true when one of the lane's comparisons met equal high dwords.
// DIRECT: the hashes came from phase_hash_closed<U>; the k-mer's strand is then decided here, from the k-mer's first 16 bases on
// the forward strand against its first 16 on the reverse strand (k = U + W - 1 >= 16; equal words: undecided, the exact form runs).
// AP: the dwords are murmur64_top<true> values (phase_hash_closed); a comparison is undecided when its two sides are less than 2
// apart, and so is the tile when any lane, owning k-mers or not, holds a value below 2.
template <int MODE, int W, int U = 0, bool AP = false>
BL_DEV uint32_t phase_sync_closed(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0, ThreadState& st, const ThreadState* all, bool& undecided)
{
    static_assert(!AP || U != 0, "approximate dwords: the direct form only");
    static_assert(W >= 2, "closed syncmers need at least two s-mers per k-mer");
    constexpr bool DIRECT = U != 0;
    static_assert(!DIRECT || (U + W - 1 >= 16 && U + W - 1 <= 32), "the strand test reads 16 bases from either end of the k-mer");
    const int wv = wave_index(tid), lane = tid & 63;
    constexpr int NE = W - 1;        // halo elements
    // One strand at a time — its 16 + W - 1 high dwords, their sliding minima, then ONE BIT per k-mer: "hit if this strand is the
    // canonical one" — so that only one strand's arrays are alive at any moment (both at once do not fit the registers of four
    // waves per SIMD); the k-mers' strands then pick between the two 16-bit masks with three word operations.
    uint32_t closest = ~0u;  // smallest xor distance between the two sides of any comparison: 0 = some comparison met equal dwords
                             // (AP: smallest absolute difference)
    uint32_t same = ~0u;     // AP: the k-mers' strand tests, which stay exact, fold here
    uint32_t low = AP ? st.hlow : ~0u;
    uint32_t hit_f = 0, hit_r = 0;
    {
        uint32_t key[S + NE];
        BL_UNROLL
        for (int s = 0; s < S; ++s) key[s] = (uint32_t)(st.h[s] >> 32);
        gather_halo_hi<NE, false>(all, tid, key);
        hit_f = closed_hits<W, AP>(key, closest);  // forward strand canonical: minimum at offset 0 (leftmost wins a tie) or strictly at W - 1
    }
    uint32_t hit = hit_f;
    if (p.canonical) {
        uint32_t key[S + NE];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        if (DIRECT) {
            BL_UNROLL
            for (int s = 0; s < S; ++s) {
                const uint32_t rv = bases_at(st.rw[0], st.rw[1], st.rw[2], 48 - (DIRECT ? U : 1) - s, DIRECT ? U : 1);
                key[s] = AP ? murmur64_top<true>(rv, p.seed) : (uint32_t)(murmur64(rv, p.seed) >> 32);
                if ((s & 3) == 3) BL_SCHED_FENCE();
            }
        } else
#endif
        {
            BL_UNROLL
            for (int s = 0; s < S; ++s) key[s] = (uint32_t)(st.h2[s] >> 32);
        }
        if (AP) {
            BL_UNROLL
            for (int s = 0; s + 1 < S; s += 2) fold_min3(low, key[s], key[s + 1]);
        }
        gather_halo_hi<NE, true>(all, tid, key);
        hit_r = closed_hits<W, AP>(key, closest);  // reverse strand canonical: positions mirror, the rightmost wins a tie
        uint32_t rev = st.strand;
        if (DIRECT) {  // reverse strand canonical <=> rc < fwd (kmer_view.hpp:196), read off the 16 leading bases of each
            rev = 0;
            uint32_t x_prev = 0;
            BL_UNROLL
            for (int s = S - 1; s >= 0; --s) {  // (last k-mer first, bits shifted in: see closed_hits)
                const uint32_t f16 = bases_at(st.cw[0], st.cw[1], st.cw[2], s, 16);
                const uint32_t r16 = bases_at(st.rw[0], st.rw[1], st.rw[2], 48 - (U + W - 1) - s, 16);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
                asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(rev) : "v"(r16), "v"(f16) : "vcc");
#else
                rev = rev + rev + (r16 < f16 ? 1u : 0u);
#endif
                if (s & 1) x_prev = f16 ^ r16;
                else fold_min3(AP ? same : closest, f16 ^ r16, x_prev);
            }
        }
        hit = (rev & hit_r) | (~rev & hit_f);
    }
    Bits128 good, start;
    gather_flags(sh.flags + wave_chunk0(p, wv), lane, good, start);
    const int k = p.unit + W - 1;
    const uint32_t valid = window_valid_mask(good, start, k) & 0xffffu;
    const int64_t j0 = wave_origin(p, q0, wv) + 16 * (int64_t)lane;
    const int64_t wj0 = wave_origin(p, q0, wv);
    uint32_t inrange = 0x1ffffu;
    if (!(wj0 >= p.win_first && wj0 + WH + 1 <= p.win_end)) inrange = range_mask(p.win_first - j0, p.win_end - j0);
    uint32_t emit = hit & valid & owned_mask(p, lane) & inrange;
    if (p.drop_last) {  // the k-mer that ends its sequence is never examined by the idiom (Q1)
        uint32_t last = (uint32_t)b128_shr(start, k).lo & 0xffffu;  // a sequence starts right behind the k-mer at s
        last |= range_mask(p.n_bases - k - j0, S + 1);              // ... or the batch ends there
        emit &= ~last;
    }
    st.emit = emit;
    st.endm = 0;
    undecided = AP ? ((closest < 2u || same == 0u) && owned_mask(p, lane) != 0) || low < 2u : closest == 0 && owned_mask(p, lane) != 0;
    return (uint32_t)__builtin_popcount(emit);
}

// ------------------------------------------------------------------------------------------------
// Closed syncmers for ANY (k, s) — the window count w = k - s + 1 is a run-time value, 1 <= w <= 32.  Same test as closed_hits,
// on the hashes' own high dwords: per strand the sliding minimum `mid` of the w - 2 s-mers between the k-mer's two ends comes from a
// sparse table inside the lane (doubling levels up to P, the largest power of two <= w - 2, then min(t[i], t[i + (w - 2) - P]):
// window_argmin_doubling on plain values), and both the second table entry and the k-mer's last s-mer sit at a run-time offset
// from the first — the same offset, (w - 2) - P, applied bit by bit with uniform branches over register moves (shift_stage).
// Short s-mers repeat inside a window as a matter of course (s = 8: the smaller end equals the minimum between in one k-mer of
// 3,000; s = 5: in most), so a comparison that meets equal dwords is not handed to another kernel, tile and all: the K-MER is decided
// again where it stands, on the 64-bit hashes, which the wave publishes in LDS when — and only when — one of its lanes asks.
// hashes: phase_hash (rolling registers, both strands).

// dst[i] = src[i + sh] for i < S, 0 <= sh <= MAXSH (MAXSH + 1 a power of two or MAXSH = 0); src holds S + MAXSH entries
template <int MAXSH>
BL_DEV void shifted_slice(const uint32_t* src, int sh, uint32_t* dst)
{
    uint32_t t[S + (MAXSH > 0 ? MAXSH : 1)];
    BL_UNROLL
    for (int i = 0; i < S + MAXSH; ++i) t[i] = src[i];
    // after the stage of bit B the shifts still to come sum to at most B - 1: t[0 .. S + B - 1) stays valid
    if (MAXSH >= 16) shift_stage<S + 15, 16>(t, sh);
    if (MAXSH >= 8) shift_stage<S + 7, 8>(t, sh);
    if (MAXSH >= 4) shift_stage<S + 3, 4>(t, sh);
    if (MAXSH >= 2) shift_stage<S + 1, 2>(t, sh);
    if (MAXSH >= 1) shift_stage<S, 1>(t, sh);
    BL_UNROLL
    for (int i = 0; i < S; ++i) dst[i] = t[i];
}

// one strand's verdicts (bit s: k-mer s has its minimum s-mer at one of its two ends, if this strand is the canonical one) from the
// lane's dwords key[0 .. S + 2P): P <= w - 2 < 2P.  tie: bit s set where the smaller end and the minimum between share their dword.
template <int P>
BL_DEV uint32_t closed_hits_rt(const uint32_t* key, int w, uint32_t& tie)
{
    constexpr int NT = S + 2 * P - 1;  // entries of the table: t[j] covers key[j + 1 .. j + P]
    uint32_t t[NT];
    BL_UNROLL
    for (int j = 0; j < NT; ++j) t[j] = key[j + 1];
    uint32_t unused = ~0u;
    if (P > 1) doubling_level<NT, 1>(t, unused);
    if (P > 2) doubling_level<NT, 2>(t, unused);
    if (P > 4) doubling_level<NT, 4>(t, unused);
    if (P > 8) doubling_level<NT, 8>(t, unused);
    const int sh = (w - 2) - P;  // 0 .. P - 1
    uint32_t t2[S], last[S];
    shifted_slice<P - 1>(t, sh, t2);              // t2[i] = t[i + sh]: covers key[i + 1 + sh .. i + sh + P] = up to key[i + w - 2]
    shifted_slice<P - 1>(key + P + 1, sh, last);  // last[i] = key[i + P + 1 + sh] = key[i + w - 1]
    uint32_t hit = 0, eq = 0;
    BL_UNROLL
    for (int i = S - 1; i >= 0; --i) {
        const uint32_t mid = t[i] < t2[i] ? t[i] : t2[i];
        const uint32_t e = key[i] < last[i] ? key[i] : last[i];
        hit = hit + hit + (e < mid ? 1u : 0u);
        eq = eq + eq + (e == mid ? 1u : 0u);
    }
    tie = eq;
    return hit;
}

// w by run time, in the size group of the kernel: GROUP_B false: w <= 17 (one halo hop), true: 18 <= w <= 32 (two hops)
template <bool GROUP_B>
BL_DEV uint32_t closed_hits_any(const uint32_t* key, int w, uint32_t& tie)
{
    tie = 0;
    if (w <= 2) return 0xffffu;  // one or two s-mers: the minimum sits at an end
    const int wm = w - 2;
    if (GROUP_B) return closed_hits_rt<16>(key, w, tie);
    if (wm >= 8) return closed_hits_rt<8>(key, w, tie);
    if (wm >= 4) return closed_hits_rt<4>(key, w, tie);
    if (wm >= 2) return closed_hits_rt<2>(key, w, tie);
    return closed_hits_rt<1>(key, w, tie);
}

// The k-mers of `todo` (bit s = the lane's k-mer s), decided on the wave's 64-bit hashes through LDS: the position of the minimum
// among the k-mer's w s-mers — the leftmost of equals on the forward strand, the rightmost on the reverse one (SECOND) — is its first or
// its last.  The wave's hashes are published first (wave-local region of sh.half, as window_argmin_lds_exact does).
template <int MODE, int W, bool SECOND>
BL_DEV uint32_t closed_exact_kmers(TileShared<MODE, W>& sh, const ThreadState* all, int tid, const ThreadState& st, int w, uint32_t todo)
{
    const int wbase = tid & ~63, lane = tid & 63;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    BL_UNROLL
    for (int s = 0; s < S; ++s) sh.half[s][W < 0 ? tid : 0] = SECOND ? st.h2[s] : st.h[s];
#else
    (void)st;
    for (int t = wbase; t < wbase + 64; ++t)
        for (int s = 0; s < S; ++s) sh.half[s][W < 0 ? t : 0] = SECOND ? all[t].h2[s] : all[t].h[s];
#endif
    uint32_t hit = 0;
    while (todo) {
        const int i = __builtin_ctz(todo);
        todo &= todo - 1;
        uint64_t best = 0;
        int arg = 0;
        for (int x = 0; x < w; ++x) {
            int pos = 16 * lane + i + x;
            pos = pos < WH ? pos : WH - 1;  // beyond the wave tile: never part of an owned k-mer
            const uint64_t v = sh.half[pos & 15][W < 0 ? wbase + (pos >> 4) : 0];
            if (x == 0 || (SECOND ? v <= best : v < best)) { best = v; arg = x; }
        }
        if (arg == 0 || arg == w - 1) hit |= 1u << i;
    }
    return hit;
}

// W: the run-time width group of the kernel (-8: w <= 17, -16: 18 <= w <= 32; the TileShared of these groups carries the LDS
// region for the wave's hashes)
template <int MODE, int W>
BL_DEV uint32_t phase_sync_closed_rt(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0, ThreadState& st, const ThreadState* all)
{
    static_assert(W == -8 || W == -16, "run-time width groups");
    constexpr bool GROUP_B = W == -16;
    const int wv = wave_index(tid), lane = tid & 63;
    const int w = p.w;
    constexpr int NE = GROUP_B ? 2 * S : S;  // halo elements: keys up to index w + 14
    uint32_t hit_f, tie_f, hit_r = 0, tie_r = 0;
    {
        uint32_t key[S + NE];
        BL_UNROLL
        for (int s = 0; s < S; ++s) key[s] = (uint32_t)(st.h[s] >> 32);
        gather_halo_hi<NE, false>(all, tid, key);
        hit_f = closed_hits_any<GROUP_B>(key, w, tie_f);
    }
    if (p.canonical) {
        uint32_t key[S + NE];
        BL_UNROLL
        for (int s = 0; s < S; ++s) key[s] = (uint32_t)(st.h2[s] >> 32);
        gather_halo_hi<NE, true>(all, tid, key);
        hit_r = closed_hits_any<GROUP_B>(key, w, tie_r);
    }
    const uint32_t rev = p.canonical ? st.strand : 0u;  // bit s: the reverse strand is the canonical one for k-mer s
    Bits128 good, start;
    gather_flags(sh.flags + wave_chunk0(p, wv), lane, good, start);
    const int k = p.unit + w - 1;
    const uint32_t valid = window_valid_mask(good, start, k) & 0xffffu;
    const int64_t j0 = wave_origin(p, q0, wv) + 16 * (int64_t)lane;
    const int64_t wj0 = wave_origin(p, q0, wv);
    uint32_t inrange = 0x1ffffu;
    if (!(wj0 >= p.win_first && wj0 + WH + 1 <= p.win_end)) inrange = range_mask(p.win_first - j0, p.win_end - j0);
    uint32_t keepable = valid & owned_mask(p, lane) & inrange;
    if (p.drop_last) {  // the k-mer that ends its sequence is never examined by the idiom (Q1)
        uint32_t last = (uint32_t)b128_shr(start, k).lo & 0xffffu;
        last |= range_mask(p.n_bases - k - j0, S + 1);
        keepable &= ~last;
    }
    // equal dwords on the strand that counts, in a k-mer that could be reported: the 64-bit hashes decide
    const uint32_t todo_f = tie_f & ~rev & keepable, todo_r = tie_r & rev & keepable;
    if (BL_COLD(wave_any((todo_f | todo_r) != 0u))) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        // the hashes are computed a second time here: kept from phase_hash, their low dwords would sit in 32 registers all through the
        // sliding minima above (the kernel then spills at three waves per SIMD)
        ThreadState full;
        phase_hash<MODE, W>(p, sh, tid, full);
#else
        const ThreadState& full = st;
#endif
        const uint32_t xf = closed_exact_kmers<MODE, W, false>(sh, all, tid, full, w, todo_f);
        hit_f = (hit_f & ~todo_f) | xf;
        if (p.canonical) {
            const uint32_t xr = closed_exact_kmers<MODE, W, true>(sh, all, tid, full, w, todo_r);
            hit_r = (hit_r & ~todo_r) | xr;
        }
    }
    const uint32_t emit = ((rev & hit_r) | (~rev & hit_f)) & keepable;
    st.emit = emit;
    st.endm = 0;
    return (uint32_t)__builtin_popcount(emit);
}

// ------------------------------------------------------------------------------------------------
// Phase 4: tile-local compaction into LDS lists (position-ordered: rank = exclusive prefix + local index)
template <int MODE, int W>
BL_DEV void phase_list(TileShared<MODE, W>& sh, int tid, const ThreadState& st, uint32_t excl_s, uint32_t excl_e)
{
    const int wv = wave_index(tid), lane = tid & 63;
    const uint32_t tag = (uint32_t)wv << 12;  // which wave's staged codes the record refers to
    uint32_t m = st.emit;
    uint32_t r = excl_s;
    while (m) {
        const int s = __builtin_ctz(m);
        m &= m - 1;
        if (MODE == MODE_SYNCMER || pos_occ_form<MODE, W>()) {  // the bit's own position is the record's
            sh.list_a[r] = (uint16_t)(tag | (uint32_t)(16 * lane + s));
        } else {
            const uint32_t arel = (uint32_t)((s < 8 ? st.apk0 : st.apk1) >> (8 * (s & 7))) & 0xffu;
            sh.list_a[r] = (uint16_t)(tag | (16 * lane + arel));
            if (MODE == MODE_SUPERKMER) sh.list_j[r] = (uint16_t)(tag | (uint32_t)(16 * lane + s + 1));
        }
        ++r;
    }
    if (MODE == MODE_SUPERKMER) {
        m = st.endm;
        r = excl_e;
        while (m) {
            const int s = __builtin_ctz(m);
            m &= m - 1;
            sh.list_e[r] = (uint16_t)(tag | (uint32_t)(16 * lane + s));
            ++r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 5: materialise the records densely (thread r handles record r, r+TPB, ...) with coalesced
// stores: the unit is re-extracted from the staged codes and re-hashed (0.13 records per base — cheaper
// than keeping 8 bytes of hash per position in LDS), and folded into the thread's digest accumulators.
struct Digest {
    unsigned long long xv, xh, xp;
};

// One record, ready to be stored once the tile's global offset is known.
struct Record {
    uint64_t v, h, pos, first;
    uint32_t mmpos;
};

// The record lists of a tile as pass 2 sees them: plain pointers — into the tile's global slots on the GPU (every entry is
// read exactly once, by the thread that builds its record: staging them in LDS would only cost residency), into the
// emulated LDS lists in the harness.
struct TileLists {
    const uint32_t* codes;   // the tile's 2-bit codes (LDS: units are extracted at arbitrary offsets)
    const uint16_t* list_a;  // [n_s]
    const uint16_t* list_j;  // [n_s]  super-k-mer mode
    const uint16_t* list_e;  // [n_e]  super-k-mer mode
    const uint16_t* next_e;  // super-k-mer mode: list_e of the NEXT tile (its first entry closes a group this tile leaves open)
};

// Where a group ends.  Starts and ends pair up in global order, and at any point of the scan at most one group is open
// (a group is <= w windows long, far shorter than a tile), so the r-th start of a tile pairs with local end r + d, where
// d = (starts before this tile) - (ends before this tile) is 0 or 1; the one start a tile may leave without an end finds it
// as the first end of the next tile.  The group's size follows without a scratch array or a second kernel.
BL_DEV int64_t end_position(const ScanParams& p, const TileLists& L, uint32_t tile, int64_t q0, uint32_t e, uint32_t n_e)
{
    const bool here = e < n_e;
    const uint32_t ent = here ? L.list_e[e] : L.next_e[0];
    const int64_t tq0 = here ? q0 : (p.frl ? ((p.origin + (int64_t)(tile + 1) * p.stride) & ~15LL) : p.origin + (int64_t)(tile + 1) * p.stride);
    return p.frl ? tq0 + ent : wave_origin(p, tq0, ent >> 12) + (ent & 0xfff);
}

// 5a: one record from its list entries.  Entries: (wave << 12) | wave-relative position, or — read-tiled scans — a flat
// tile-relative position.
template <int MODE, bool FENCED = false>
BL_DEV Record emit_prepare(const ScanParams& p, const uint32_t* codes, int64_t q0, uint32_t ent, uint32_t ent_j, Digest& dg)
{
    Record rec{0, 0, 0, 0, 0};
    const int wv = p.frl ? 0 : ent >> 12, ap = p.frl ? ent : ent & 0xfff;
    const int64_t wq0 = p.frl ? q0 : wave_origin(p, q0, wv);
    rec.pos = (uint64_t)(wq0 + ap);  // inside the batch; p.pos_base is added where positions leave the kernel (digest, stores)
    dg.xp ^= rec.pos + (uint64_t)p.pos_base;
    if (MODE != MODE_SYNCMER) {
        rec.v = extract_unit(p.frl ? codes : codes + wave_chunk0(p, wv), ap, p.unit, p.canonical);
        if (FENCED) BL_SCHED_FENCE();  // (the looped form of pass 2 lives on few registers: see scan_emit_kernel)
        rec.h = murmur64(rec.v, p.seed);
        if (FENCED) BL_SCHED_FENCE();
        dg.xv ^= rec.v;
        dg.xh ^= rec.h;
        if (MODE == MODE_SUPERKMER) {
            const int j = p.frl ? ent_j : ent_j & 0xfff;
            rec.first = (uint64_t)(wq0 + j);
            rec.mmpos = (uint32_t)(ap - j);  // super_kmer_view.hpp:132
        }
    }
    return rec;
}

// 5b: coalesced stores (thread r writes record r).  Records are written once and not read again by
// the scan: non-temporal stores keep them from displacing the tile data in L2.
template <typename T>
BL_DEV void stream_store(T* dst, T v)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}

// CHECK = false: the caller has established that the whole tile fits below the capacity
template <int MODE, bool CHECK = true>
BL_DEV void emit_store(const ScanParams& p, const Record& rec, uint64_t g)
{
    if (CHECK && g >= p.capacity) return;
    if (MODE == MODE_SYNCMER) {
        if (p.out_pos) stream_store(&p.out_pos[g], rec.pos + (uint64_t)p.pos_base);
        return;
    }
    if (p.out_value) stream_store(&p.out_value[g], rec.v);
    if (p.out_hash) stream_store(&p.out_hash[g], rec.h);
    if (p.out_pos) stream_store(&p.out_pos[g], rec.pos + (uint64_t)p.pos_base);
    if (MODE == MODE_SUPERKMER) {
        if (p.out_first) stream_store(&p.out_first[g], rec.first + (uint64_t)p.pos_base);
        if (p.out_mmpos) p.out_mmpos[g] = (uint8_t)rec.mmpos;
    }
}

// super-k-mer scans that hand out packed records: the group's bases come from the tile's staged codes (a group begins in the
// tile and is at most 2k - m <= 59 bases long: inside what the tile staged for its own windows)
BL_DEV void emit_record(const ScanParams& p, const uint32_t* codes, int last_chunk, int64_t q0, const Record& rec, int size, uint64_t g)
{
    uint64_t x, y;
    pack_group(codes, last_chunk, (int)((int64_t)rec.first - q0), size + p.unit + p.w - 2, rec.mmpos, size, x, y);
    stream_store(&p.out_records[2 * g], x);
    stream_store(&p.out_records[2 * g + 1], y);
}

// whole phase for one thread: records tid, tid + TPB, ...
template <int MODE>
BL_DEV void phase_emit(const ScanParams& p, const TileLists& L, uint32_t tile, int tid, int64_t q0, uint32_t n_s, uint32_t n_e, uint64_t base_s, uint64_t base_e,
                       Digest& dg)
{
    const bool fits = !BL_COLD(base_s + n_s > p.capacity);  // the usual case, uniform for the workgroup: no per-record capacity test
    const uint32_t d = (uint32_t)(base_s - base_e);        // 0 or 1: a group left open by the tiles before this one
    for (uint32_t r = tid; r < n_s; r += TPB) {
        const Record rec = emit_prepare<MODE>(p, L.codes, q0, L.list_a[r], MODE == MODE_SUPERKMER ? L.list_j[r] : 0u, dg);
        if (fits) emit_store<MODE, false>(p, rec, base_s + r);
        else emit_store<MODE, true>(p, rec, base_s + r);
        if (MODE == MODE_SUPERKMER && (p.out_size || p.out_records) && base_s + r < p.capacity) {
            const int size = (int)(end_position(p, L, tile, q0, r + d, n_e) - (int64_t)rec.first + 1);
            if (p.out_size) p.out_size[base_s + r] = (uint8_t)size;
            if (p.out_records) emit_record(p, L.codes, staged_chunks(p) - 1, q0, rec, size, base_s + r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dense k-mer scan (config C2): every position's unit value / hash / validity, no windows.
struct KmerParams {
    const uint8_t* bases;
    int64_t n_bases;
    const uint32_t* start_bits;
    int64_t first, end;          // positions [first, end) are reported
    int64_t origin;              // 16-aligned, <= first
    int32_t n_tiles;
    int32_t unit;
    uint32_t seed;
    int32_t canonical;
    int32_t drop_last;
    uint64_t* out_value;         // dense, indexed by position - first (nullable)
    uint64_t* out_hash;
    uint8_t* out_valid;
    unsigned long long* shards;  // [NSHARD][8]: count, xor value, xor hash, sum hash
};

struct KmerAcc {
    unsigned long long cnt, xv, xh, sh;
};

BL_DEV void kmer_thread(const KmerParams& p, const uint32_t* codes, const uint32_t* flags, int tid, int64_t q0, KmerAcc& acc)
{
    Roller r;
    roller_start(r, codes[tid], codes[tid + 1], codes[tid + 2], p.unit);
    Bits128 good, start;
    gather_flags(flags, tid, good, start);
    const int64_t j0 = q0 + 16 * (int64_t)tid;
    // which of the lane's 16 positions count, as ONE mask (no 64-bit compare per position): a k-mer starts there, the position
    // lies in the requested range, and — reference idiom — it is not the k-mer that ends its sequence (quirk Q1)
    const uint32_t inrange = range_mask(p.first - j0, p.end - j0) & 0xffffu;
    uint32_t ok = window_valid_mask(good, start, p.unit) & inrange;
    if (p.drop_last) {
        uint32_t last = (uint32_t)b128_shr(start, p.unit).lo & 0xffffu;  // a sequence starts right after the k-mer at s
        const int64_t s_end = p.n_bases - p.unit - j0;                   // ... or the batch ends there
        if (s_end >= 0 && s_end < S) last |= 1u << s_end;
        ok &= ~last;
    }
    const bool any_out = p.out_value || p.out_hash || p.out_valid;  // uniform: the digest-only scan (C2) stores nothing
    // every lane of the wave counts all 16 of its positions (the inside of a long sequence): nothing to mask.  The masked form spends
    // 11 instructions per position on `take ? x : 0` and the three folds, the plain one 5.  (One loop with a scalar branch per
    // position, not two loops: the compiler hoists what two loops share — all 16 canonical k-mers — above the branch, 151 registers.)
    const bool plain = !any_out && !wave_any(ok != 0xffffu);
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        roller_step(r, s);
        const uint64_t fw = roller_fwd(r), rv = roller_rc(r);
        const uint64_t v = (p.canonical && rv < fw) ? rv : fw;
        const uint64_t h = murmur64(v, p.seed);
        if (plain) {
            acc.xv ^= v;
            acc.xh ^= h;
            acc.sh += h;
            continue;
        }
        const bool take = (ok >> s) & 1;
        const uint32_t m32 = 0u - ((ok >> s) & 1u);  // all ones / zero: and-masks (and, folded into the xors, v_bitop3) instead of selects
        const uint64_t m = ((uint64_t)m32 << 32) | m32;
        const uint64_t vm = v & m, hm = h & m;
        acc.xv ^= vm;
        acc.xh ^= hm;
        acc.sh += hm;
        if (any_out && ((inrange >> s) & 1)) {
            const int64_t o = j0 + s - p.first;
            if (p.out_value) p.out_value[o] = vm;
            if (p.out_hash) p.out_hash[o] = hm;
            if (p.out_valid) p.out_valid[o] = take ? 1 : 0;
        }
    }
    acc.cnt += (unsigned)__builtin_popcount(ok);
}

}  // namespace bl
