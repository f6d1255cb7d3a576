// bl_scan_phases.hpp — the tile pipeline of the fused scan, one function per barrier-delimited
// phase.  The HIP kernels (bl_kernels.hip) call these with __syncthreads() in between; the CPU
// emulation harness (tests/emu/) calls the very same functions thread by thread.
//
// Tile geometry (all positions are indices into the batch's base buffer):
//   tile t hashes the H = 4096 unit start positions [q0, q0+H), q0 = origin + t*stride, 16-aligned.
//   thread `tid` owns the S = 16 positions i0 .. i0+15, i0 = 16*tid (tile-relative) — exactly the
//   bases of one coalesced 16-byte load.
//   minimizer / super-k-mer modes: the thread that owns position i decides, from the argmins of
//   windows i and i+1, whether window i+1 starts a new minimizer occurrence and whether window i
//   ends one.  Only the first `stride` positions of a tile are owned (stride <= H - w), so every
//   decision sees all the hashes it needs without a second pass over the neighbouring tile.
#pragma once
#include "bl_scan_core.hpp"

namespace bl {

template <int MODE>
struct TileShared {
    uint32_t codes[NCHUNK];        // 2-bit codes, 16 bases per dword, first base most significant
    uint32_t flags[NCHUNK];        // [15:0] good-base bits, [31:16] sequence-start bits (bit b = base b)
    uint64_t hash[S][TPB];         // hash[s][t] = hash of the unit at tile position 16*t + s (bank-conflict-free)
    uint16_t list_a[H];            // compacted: argmin position (minimizer modes) / window position (syncmer)
    uint16_t list_j[MODE == MODE_SUPERKMER ? H : 1];  // compacted: first window of the occurrence
    uint16_t list_e[MODE == MODE_SUPERKMER ? H : 1];  // compacted: last window of the occurrence
    uint32_t wave_tot[TPB / 64];
    unsigned long long dig[4];
    uint32_t tile;
    uint32_t base_s, base_e;       // global record offsets of this tile (from the look-back)
};

struct ThreadState {
    uint64_t h[S];    // hashes of the owned positions (forward strand for syncmers)
    uint64_t h2[S];   // syncmer: reverse-strand s-mer hashes
    uint32_t valid;   // bit s: window starting at owned position s is valid (bits 0..S)
    uint32_t emit;    // bit s: a record starts at window s+1 (minimizer modes) / window s is a syncmer
    uint32_t endm;    // bit s: an occurrence ends at window s (super-k-mer mode)
    uint32_t strand;  // syncmer: bit s set <=> reverse strand is canonical for the k-mer at s
    uint32_t lastk;   // syncmer: bit s set <=> the k-mer at s ends its sequence
    uint64_t apk[2];  // minimizer modes: argmin (thread-relative element index) of window s+1 in byte s (8 per word)
    uint8_t a[S];     // syncmer: offset of the forward-strand minimum inside k-mer s (compile-time indexed only)
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
        const Vec16 v = *reinterpret_cast<const Vec16*>(p.bases + g);  // one global_load_dwordx4
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if (g + 16 > 0 && g < p.n_bases) {  // ragged edge: byte-wise, zeros (= breaks) outside
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

template <int MODE>
BL_DEV void phase_load(const ScanParams& p, TileShared<MODE>& sh, int tid, int64_t q0)
{
    stage_chunk(p, sh.codes, sh.flags, tid, q0);
    if (tid < NCHUNK - TPB) stage_chunk(p, sh.codes, sh.flags, TPB + tid, q0);
}

// good / start bit-vectors for the thread's positions i0 .. i0+127
BL_DEV void gather_flags(const uint32_t* flags, int tid, Bits128& good, Bits128& start)
{
    uint64_t g[2] = {0, 0}, s[2] = {0, 0};
    BL_UNROLL
    for (int c = 0; c < 8; ++c) {
        const uint32_t f = flags[tid + c];  // tid + 7 <= 262 < NCHUNK
        g[c >> 2] |= (uint64_t)(f & 0xffffu) << (16 * (c & 3));
        s[c >> 2] |= (uint64_t)(f >> 16) << (16 * (c & 3));
    }
    good = Bits128{g[0], g[1]};
    start = Bits128{s[0], s[1]};
}

// ------------------------------------------------------------------------------------------------
// Phase 2: roll the owned 16 units in registers, hash them, publish the hashes to LDS.
template <int MODE>
BL_DEV void phase_hash(const ScanParams& p, TileShared<MODE>& sh, int tid, ThreadState& st)
{
    const uint32_t c0 = sh.codes[tid], c1 = sh.codes[tid + 1], c2 = sh.codes[tid + 2];
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
            st.h[s] = murmur64(r.fwd, p.seed);
            st.h2[s] = murmur64(r.rc, p.seed);
            if (p.canonical && rk.rc < rk.fwd) strand |= 1u << s;  // kmer_view.hpp:196
        }
        st.strand = strand;
    } else {
        BL_UNROLL
        for (int s = 0; s < S; ++s) {
            roller_step(r, s);
            const uint64_t v = (p.canonical && r.rc < r.fwd) ? r.rc : r.fwd;  // minimizer_view.hpp:236-238
            st.h[s] = murmur64(v, p.seed);
        }
    }
    BL_UNROLL
    for (int s = 0; s < S; ++s) sh.hash[s][tid] = st.h[s];
}

// syncmer mode, second pass over the same LDS array
template <int MODE>
BL_DEV void phase_publish_h2(TileShared<MODE>& sh, int tid, const ThreadState& st)
{
    BL_UNROLL
    for (int s = 0; s < S; ++s) sh.hash[s][tid] = st.h2[s];
}

// elements S .. S+NE-1 of the thread's window input come from the following threads' hashes
template <int MODE, int NE>
BL_DEV void gather_halo(const TileShared<MODE>& sh, int tid, uint64_t* e)
{
    BL_UNROLL
    for (int x = 0; x < NE; ++x) {
        int t = tid + 1 + (x >> 4);
        t = t < TPB ? t : TPB - 1;  // beyond the tile: never owned, any value will do
        e[S + x] = sh.hash[x & 15][t];
    }
}

// bit s set <=> lo <= s < hi, for s in 0..S
BL_DEV uint32_t range_mask(int64_t lo, int64_t hi)
{
    const int l = lo < 0 ? 0 : (lo > S + 1 ? S + 1 : (int)lo);
    const int h = hi < 0 ? 0 : (hi > S + 1 ? S + 1 : (int)hi);
    return h > l ? (((1u << h) - 1) & ~((1u << l) - 1)) : 0u;
}

// runtime-w fallback: argmin by direct scan of the LDS hashes (slow path for unlisted window sizes)
template <int MODE, bool LEFT>
BL_DEV void window_argmin_lds(const TileShared<MODE>& sh, int tid, int w, int nw, uint8_t* a)
{
    for (int i = 0; i < nw; ++i) {
        uint64_t best = 0;
        int arg = 0;
        for (int x = 0; x < w; ++x) {
            int pos = 16 * tid + i + x;
            pos = pos < H ? pos : H - 1;
            const uint64_t v = sh.hash[pos & 15][pos >> 4];
            const bool take = x == 0 || (LEFT ? v < best : v <= best);
            if (take) { best = v; arg = i + x; }
        }
        a[i] = (uint8_t)arg;
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 3 (minimizer / super-k-mer): window argmins, validity, start/end decisions.
// Returns the packed per-thread counts: starts | ends << 16.
template <int MODE, int W>
BL_DEV uint32_t phase_window(const ScanParams& p, const TileShared<MODE>& sh, int tid, int64_t q0, ThreadState& st)
{
    const int w = W ? W : p.w;
    uint8_t a[S + 1];
    if (W) {
        uint64_t e[S + (W ? W : 1)];
        BL_UNROLL
        for (int s = 0; s < S; ++s) e[s] = st.h[s];
        gather_halo<MODE, (W ? W : 1)>(sh, tid, e);
        window_argmin<S + 1, (W ? W : 1), true>(e, a);
    } else {
        window_argmin_lds<MODE, true>(sh, tid, w, S + 1, a);
    }
    Bits128 good, start;
    gather_flags(sh.flags, tid, good, start);
    uint32_t valid = window_valid_mask(good, start, p.unit + w - 1);
    // windows outside the requested range: never reported; in super-k-mer mode they also cut groups
    const int64_t j0 = q0 + 16 * (int64_t)tid;  // global position of the thread's window 0
    const uint32_t inrange = range_mask(p.win_first - j0, p.win_end - j0);
    if (MODE == MODE_SUPERKMER) valid &= inrange;
    uint32_t differ = 0;  // bit s: argmin of window s+1 is a different occurrence than window s
    uint64_t apk0 = 0, apk1 = 0;
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        if (a[s + 1] != a[s]) differ |= 1u << s;
        if (s < 8) apk0 |= (uint64_t)a[s + 1] << (8 * s);
        else apk1 |= (uint64_t)a[s + 1] << (8 * (s - 8));
    }
    st.apk[0] = apk0;
    st.apk[1] = apk1;
    const uint32_t v0 = valid & 0xffffu, v1 = (valid >> 1) & 0xffffu;
    uint32_t owned = 0;
    const int own = p.stride - 16 * tid;  // owned positions of this thread: s < own
    if (own >= S) owned = 0xffffu;
    else if (own > 0) owned = (1u << own) - 1;
    st.valid = valid;
    st.emit = v1 & (~v0 | differ) & owned & ((inrange >> 1) & 0xffffu);  // window s+1 starts an occurrence
    st.endm = MODE == MODE_SUPERKMER ? (v0 & (~v1 | differ) & owned) : 0;  // window s ends one
    return (uint32_t)__builtin_popcount(st.emit) | ((uint32_t)__builtin_popcount(st.endm) << 16);
}

// Phase 3 (syncmer), two sub-phases around the republish of the reverse-strand hashes.
template <int MODE, int W>
BL_DEV void phase_sync_fwd(const ScanParams& p, const TileShared<MODE>& sh, int tid, ThreadState& st)
{
    if (W) {
        uint64_t e[S + (W ? W : 1)];
        BL_UNROLL
        for (int s = 0; s < S; ++s) e[s] = st.h[s];
        gather_halo<MODE, (W ? W : 1)>(sh, tid, e);
        uint8_t a[S + 1];
        window_argmin<S, (W ? W : 1), true>(e, a);
        BL_UNROLL
        for (int s = 0; s < S; ++s) st.a[s] = (uint8_t)(a[s] - s);  // offset of the leftmost forward minimum
    } else {
        uint8_t a[S + 1];
        window_argmin_lds<MODE, true>(sh, tid, p.w, S, a);
        for (int s = 0; s < S; ++s) st.a[s] = (uint8_t)(a[s] - s);
    }
}

template <int MODE, int W>
BL_DEV uint32_t phase_sync_rev(const ScanParams& p, const TileShared<MODE>& sh, int tid, int64_t q0, ThreadState& st)
{
    const int w = W ? W : p.w;
    const int k = p.unit + w - 1;
    uint8_t ar[S + 1];
    if (p.canonical) {
        if (W) {
            uint64_t e[S + (W ? W : 1)];
            BL_UNROLL
            for (int s = 0; s < S; ++s) e[s] = st.h2[s];
            gather_halo<MODE, (W ? W : 1)>(sh, tid, e);
            window_argmin<S, (W ? W : 1), false>(e, ar);
        } else {
            window_argmin_lds<MODE, false>(sh, tid, w, S, ar);
        }
    }
    Bits128 good, start;
    gather_flags(sh.flags, tid, good, start);
    const uint32_t valid = window_valid_mask(good, start, k) & 0xffffu;
    const int64_t j0 = q0 + 16 * (int64_t)tid;
    uint32_t emit = 0;
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        // canonical k-mer on the reverse strand: its j-th m-mer from the left is the reverse complement
        // of the forward m-mer at W-1-j, and "leftmost" becomes "rightmost" 
        int off = st.a[s];
        if (p.canonical && ((st.strand >> s) & 1)) off = (w - 1) - (ar[s] - s);
        const bool hit = off == p.soff || off == p.eoff;  // syncmer_sampler.hpp:130-137
        const int64_t j = j0 + s;
        bool keep = hit && ((valid >> s) & 1) && 16 * tid + s < p.stride && j >= p.win_first && j < p.win_end;
        if (keep && p.drop_last) {  // the k-mer that ends its sequence is never examined by the idiom (Q1)
            const int nxt = s + k;  // < 128
            const bool seq_end = (nxt < 64 ? (start.lo >> nxt) : (start.hi >> (nxt - 64))) & 1;
            keep = !(seq_end || j + k >= p.n_bases);
        }
        if (keep) emit |= 1u << s;
    }
    st.valid = valid;
    st.emit = emit;
    st.endm = 0;
    return (uint32_t)__builtin_popcount(emit);
}

// ------------------------------------------------------------------------------------------------
// Phase 4: tile-local compaction into LDS lists (position-ordered: rank = exclusive prefix + local index)
template <int MODE>
BL_DEV void phase_list(TileShared<MODE>& sh, int tid, const ThreadState& st, uint32_t excl_s, uint32_t excl_e)
{
    uint32_t m = st.emit;
    uint32_t r = excl_s;
    while (m) {
        const int s = __builtin_ctz(m);
        m &= m - 1;
        if (MODE == MODE_SYNCMER) {
            sh.list_a[r] = (uint16_t)(16 * tid + s);
        } else {
            const uint32_t arel = (uint32_t)((s < 8 ? st.apk[0] : st.apk[1]) >> (8 * (s & 7))) & 0xffu;
            sh.list_a[r] = (uint16_t)(16 * tid + arel);
            if (MODE == MODE_SUPERKMER) sh.list_j[r] = (uint16_t)(16 * tid + s + 1);
        }
        ++r;
    }
    if (MODE == MODE_SUPERKMER) {
        m = st.endm;
        r = excl_e;
        while (m) {
            const int s = __builtin_ctz(m);
            m &= m - 1;
            sh.list_e[r] = (uint16_t)(16 * tid + s);
            ++r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 5: materialise the records densely (thread r handles record r, r+TPB, ...) with coalesced
// stores, and fold them into the thread's digest accumulators.
struct Digest {
    unsigned long long xv, xh, xp;
};

template <int MODE>
BL_DEV void phase_emit(const ScanParams& p, const TileShared<MODE>& sh, int tid, int64_t q0, uint32_t n_s, uint32_t n_e,
                       uint64_t base_s, uint64_t base_e, Digest& dg)
{
    for (uint32_t r = tid; r < n_s; r += TPB) {
        const int ap = sh.list_a[r];
        const uint64_t g = base_s + r;
        const uint64_t pos = (uint64_t)(q0 + ap);
        if (MODE == MODE_SYNCMER) {
            dg.xp ^= pos;
            if (p.out_pos && g < p.capacity) p.out_pos[g] = pos;
        } else {
            const uint64_t h = sh.hash[ap & 15][ap >> 4];
            const uint64_t v = extract_unit(sh.codes, ap, p.unit, p.canonical);
            dg.xv ^= v; dg.xh ^= h; dg.xp ^= pos;
            if (g < p.capacity) {
                if (p.out_value) p.out_value[g] = v;
                if (p.out_hash) p.out_hash[g] = h;
                if (MODE == MODE_SUPERKMER) {
                    const int j = sh.list_j[r];
                    if (p.out_first) p.out_first[g] = (uint64_t)(q0 + j);
                    if (p.out_mmpos) p.out_mmpos[g] = (uint8_t)(ap - j);  // super_kmer_view.hpp:132
                    if (p.out_pos) p.out_pos[g] = pos;
                } else if (p.out_pos) {
                    p.out_pos[g] = pos;
                }
            }
        }
    }
    if (MODE == MODE_SUPERKMER) {
        for (uint32_t r = tid; r < n_e; r += TPB) {
            const uint64_t g = base_e + r;
            if (p.out_last && g < p.capacity) p.out_last[g] = (uint64_t)(q0 + sh.list_e[r]);
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
    uint32_t valid = window_valid_mask(good, start, p.unit) & 0xffffu;
    const int64_t j0 = q0 + 16 * (int64_t)tid;
    BL_UNROLL
    for (int s = 0; s < S; ++s) {
        roller_step(r, s);
        const uint64_t v = (p.canonical && r.rc < r.fwd) ? r.rc : r.fwd;
        const uint64_t h = murmur64(v, p.seed);
        const int64_t j = j0 + s;
        bool ok = ((valid >> s) & 1) && j >= p.first && j < p.end;
        if (ok && p.drop_last) {
            const int nxt = s + p.unit;
            const bool seq_end = (start.lo >> nxt) & 1;  // nxt <= 47
            ok = !(seq_end || j + p.unit >= p.n_bases);
        }
        if (ok) { acc.cnt += 1; acc.xv ^= v; acc.xh ^= h; acc.sh += h; }
        if (j >= p.first && j < p.end) {
            const int64_t o = j - p.first;
            if (p.out_value) p.out_value[o] = ok ? v : 0;
            if (p.out_hash) p.out_hash[o] = ok ? h : 0;
            if (p.out_valid) p.out_valid[o] = ok ? 1 : 0;
        }
    }
}

}  // namespace bl
