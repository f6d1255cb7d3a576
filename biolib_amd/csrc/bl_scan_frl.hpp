// bl_scan_frl.hpp — the READ-TILED layout of the minimizer / super-k-mer scan for batches of fixed-length short reads
// (plan_scan_frl in bl_scan_core.hpp decides when it applies; 150-bp reads with k = 31, w = 11 is the headline case).
//
// The position-tiled layout (bl_scan_phases.hpp) gives every lane 16 consecutive POSITIONS; on 150-bp reads 30 of every
// 150 positions cannot start a 31-mer, yet they are rolled, hashed and thrown away, and sequence starts are read from a
// 1-bit-per-base vector.  Here a wave takes rpw whole reads, lpr lanes per read, and a lane takes NS consecutive UNIT
// STARTS of its read (150 bp, k = 31: 8 reads per wave, 8 lanes per read, 15 units per lane = all 120 units of a read):
//   * the unit-1 tail positions of a read are never hashed (1.27 x fewer hashes per useful window at C3),
//   * reads do not share windows, so a wave needs no halo from the next wave tile,
//   * where a read starts and which windows exist is arithmetic on (lane, slot): no start_bits, no flag gathering,
//   * bases that are not ACGTUacgtu are the only thing left to look up, and only in tiles that hold one.
// Semantics are those of the position-tiled scan (reference minimizer_view.hpp:229-245: the window is cleared at a break
// and at a sequence end; leftmost minimum wins, :283,374) and the record order is the same, so both layouts produce
// identical arrays; tests/emu runs these phases thread by thread against the oracle under ASan.
#pragma once
#include "bl_scan_phases.hpp"

namespace bl {

// global position the tile's staged chunk 0 starts at (both layouts)
BL_DEV int64_t tile_q0(const ScanParams& p, uint32_t tile)
{
    const int64_t g = p.origin + (int64_t)tile * p.stride;
    return p.frl ? (g & ~15LL) : g;  // read-tiled: origin >= 0, tiles start at a read, not at a 16-byte boundary
}

// ------------------------------------------------------------------------------------------------
// Phase 1: coalesced 16-byte loads -> 2-bit codes + good-base bits; true if the chunk holds a break
BL_DEV bool stage_chunk_frl(const ScanParams& p, uint32_t* codes, uint32_t* flags, int c, int64_t q0)
{
    const int64_t g = q0 + 16 * (int64_t)c;  // >= 0
    uint32_t d[4] = {0, 0, 0, 0};
    if (g + 16 <= p.n_bases) {
        const Vec16 v = *reinterpret_cast<const Vec16*>(p.bases + g);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if (BL_COLD(g < p.n_bases)) {  // ragged end of the batch: byte-wise, zeros (= breaks) outside
        for (int b = 0; b < 16; ++b) {
            const int64_t q = g + b;
            if (q < p.n_bases) d[b >> 2] |= (uint32_t)p.bases[q] << (8 * (b & 3));
        }
    }
    uint32_t code, bad;
    encode16(d, code, bad);
    codes[c] = code;
    flags[c] = ~bad & 0xffffu;
    return bad != 0;
}

template <int MODE, int W>
BL_DEV void phase_load_frl(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0)
{
    bool bad = false;
    for (int c = tid; c < p.slot_chunks; c += TPB) bad |= stage_chunk_frl(p, sh.codes, sh.flags, c, q0);  // 2 rounds at most; the 2nd is a few lanes of wave 0
    const int wv = wave_index(tid);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    const bool any = wave_any(bad);
    if ((tid & 63) == 0) sh.wave_bad[wv] = any ? 1u : 0u;
#else
    if ((tid & 63) == 0) sh.wave_bad[wv] = 0;  // threads of a wave are emulated in order
    if (bad) sh.wave_bad[wv] = 1;
#endif
}

// ------------------------------------------------------------------------------------------------
// Phase 2: which read and which units this lane owns; roll and hash them.
// GENERIC: run-time unit length (see phase_hash: those kernels hash with the compiler's own multiply)
// APPROX: st.h[s] holds murmur64_top in its high dword (low dword 0) and st.hmax the largest of them: the window phase works on
// those and reports what it cannot decide (lane_window_argmin_frl)
template <int MODE, int W, int NS, bool GENERIC = false, bool APPROX = false>
BL_DEV void phase_hash_frl(const ScanParams& p, TileShared<MODE, W>& sh, int tid, int64_t q0, uint32_t tile, ThreadState& st)
{
    const int wv = wave_index(tid), lane = tid & 63;
    const int r = (int)(((uint32_t)lane * p.lpr_inv) >> 16);  // read of this wave the lane works on
    const int j = lane - r * p.lpr;                           // lane inside that read
    const int64_t read_idx = ((int64_t)tile * NWAVE + wv) * p.rpw + r;
    const bool active = r < p.rpw && read_idx < p.n_reads;
    const int off0 = (int)(p.origin + (int64_t)tile * p.stride - q0);  // 0..15
    const int base = active ? off0 + (wv * p.rpw + r) * p.read_len + j * NS : off0;
    st.lane_base = base;
    st.jlane = active ? j : -1;
    // the lane's 48 bases from `base` on: four staged dwords funnel-shifted to the lane's own alignment
    const uint32_t* cp = sh.codes + (base >> 4);
    const uint32_t c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
    const int sh2 = 2 * (base & 15);
    const uint32_t a0 = (uint32_t)(((((uint64_t)c0 << 32) | c1) << sh2) >> 32);
    const uint32_t a1 = (uint32_t)(((((uint64_t)c1 << 32) | c2) << sh2) >> 32);
    const uint32_t a2 = (uint32_t)(((((uint64_t)c2 << 32) | c3) << sh2) >> 32);
    Roller rr;
    roller_start(rr, a0, a1, a2, p.unit);
    uint32_t hmax = 0;
    BL_UNROLL
    for (int s = 0; s < NS; ++s) {
        roller_step(rr, s);
        const uint64_t fw = roller_fwd(rr), rv = roller_rc(rr);
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

// ------------------------------------------------------------------------------------------------
// Neighbouring lanes: values of lane - 1 / lane + 1 (DPP wave shifts on the GPU, lane 0 / 63 get their own value back,
// which no owned window ever uses; the emulation reads the neighbour's state).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
BL_DEV uint32_t dpp_next32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }  // wave_shl:1
// the same as a rotation: lane 63 reads lane 0 (wave_rol:1)
BL_DEV uint32_t dpp_next32_rot(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xf, 0xf, true); }
BL_DEV uint32_t dpp_prev32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }  // wave_shr:1
#endif

// Window argmins of a lane that owns NS elements: the W - 1 halo elements come from the following lanes, NS per hop.
// a[i] = raw packed key (element index in its low 6 bits) or, after the exact branch, the plain element index.
// APPROX (keys from murmur64_top): no exact form in here; *tie is set when a lane that owns windows met two keys less than two
// prefixes apart, or holds a key whose prefix could wrap, and the caller has the whole tile decided again (scan_redo_frl_kernel)
template <int NS, int W, bool APPROX = false>
BL_DEV void lane_window_argmin_frl(const ThreadState* all, int tid, const ThreadState& st, bool owns, uint32_t* a, bool* tie = nullptr)
{
    constexpr int NE = W - 1, NK = NS + NE;  // NK <= 64: 6-bit tags
    uint32_t key[NK];
    BL_UNROLL
    for (int s = 0; s < NS; ++s) key[s] = packed_key((uint32_t)(st.h[s] >> 32), s, true);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    (void)tid;
    {
        uint32_t cur[NS];
        BL_UNROLL
        for (int x = 0; x < NS; ++x) cur[x] = key[x];
        BL_UNROLL
        for (int hop = 0; hop * NS < NE; ++hop) {
            BL_UNROLL
            for (int x = 0; x < NS; ++x) {
                // lane 63 owns windows too (64 = rpw * lpr lanes at work) but has no lane to take a halo from: the shift is a ROTATION
                // and it gets lane 0's keys — some other read's, as good as random — so that its (non-existent) halo windows look
                // like a hash tie no more often than real ones; what they choose is cleared (phase_window_frl_a).  (A rotation reads
                // a lane everywhere: no `old` operand to initialise, one v_mov per element less than a shift with pad keys.)
                if (hop * NS + x < NE || (hop + 1) * NS + x < NE || (hop + 2) * NS + x < NE)
                    cur[x] = dpp_next32_rot(cur[x]) + (uint32_t)NS;
                if (hop * NS + x < NE) key[(hop + 1) * NS + x] = cur[x];
            }
        }
    }
#else
    {
        const int lane = tid & 63;
        for (int x = 0; x < NE; ++x) {
            const int nb = lane + 1 + x / NS;
            const uint32_t hi = nb < 64 ? (uint32_t)(all[tid + 1 + x / NS].h[x % NS] >> 32) : (0x03fffff0u - 2u * (uint32_t)x) << 6;  // pads, two prefixes apart
            key[NS + x] = packed_key(hi, NS + x, true);
        }
    }
#endif
    const uint32_t dmin = window_argmin_packed<NS, W, true, true, APPROX>(key, a);
    if (APPROX) {
        // (a lane's keys: its own NS elements and the W - 1 that follow; the lanes those belong to test their own hmax)
        if (owns && (dmin < 128u || st.hmax >= 0xffffffc0u)) *tie = true;
        return;
    }
    if (BL_COLD(wave_any(owns && dmin < 64u))) {  // a prefix tie somewhere in the wave: the exact 64-bit form
        uint64_t e[NK];
        BL_UNROLL
        for (int s = 0; s < NS; ++s) e[s] = st.h[s];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        uint64_t cur[NS];
        BL_UNROLL
        for (int x = 0; x < NS; ++x) cur[x] = st.h[x];
        BL_UNROLL
        for (int hop = 0; hop * NS < NE; ++hop) {
            BL_UNROLL
            for (int x = 0; x < NS; ++x) {
                if (hop * NS + x < NE || (hop + 1) * NS + x < NE || (hop + 2) * NS + x < NE) cur[x] = lane_next(cur[x]);
                if (hop * NS + x < NE) e[(hop + 1) * NS + x] = cur[x];
            }
        }
#else
        const int lane = tid & 63;
        for (int x = 0; x < NE; ++x) {
            const int nb = lane + 1 + x / NS;
            e[NS + x] = nb < 64 ? all[tid + 1 + x / NS].h[x % NS] : 0xDEADBEEFDEADBEEFull;
        }
#endif
        window_argmin<NS, W, true>(e, a);
    }
}

// bit s: every base of the window that starts at staged position base + s, span bases long, is one of ACGTUacgtu
// (only run for tiles that hold a break)
template <int NS>
BL_DEV uint32_t frl_good_mask(const ScanParams& p, const uint32_t* flags, int base, int span)
{
    const int ch = base >> 4, off = base & 15;
    uint64_t g[3] = {0, 0, 0};
    for (int c = 0; c < 9; ++c) {  // 144 staged bits cover off + 15 + 95 positions
        int idx = ch + c;
        idx = idx < p.slot_chunks ? idx : p.slot_chunks - 1;  // beyond the staged tile: only windows that do not exist look there
        g[c >> 2] |= (uint64_t)(flags[idx] & 0xffffu) << (16 * (c & 3));
    }
    Bits128 good;
    good.lo = off ? (g[0] >> off) | (g[1] << (64 - off)) : g[0];
    good.hi = off ? (g[1] >> off) | (g[2] << (64 - off)) : g[1];
    return (uint32_t)and_run(good, span).lo & ((1u << NS) - 1);
}

// ------------------------------------------------------------------------------------------------
// ELEMENT-CENTRIC decisions (minimizer scans).  A minimizer occurrence is one ELEMENT (unit start) that is the argmin of at
// least one valid window: the valid windows that choose an element are consecutive (leftmost minimum: if windows i < j
// choose e, every window between them lies inside their union and chooses e too, and it is valid because its bases are
// theirs), and argmins never move left as the window slides, so the records of a scan — "one each time the occurrence
// changes or a run starts", minimizer_view.hpp:193-208 — are exactly the chosen elements in position order.  A lane
// therefore ORs one bit per window into a mask indexed by element (v_lshl_or_b32: the element index is the low bits of the
// packed minimum), hands the bits of elements >= NS to the lane that owns them (one DPP move: valid windows never leave
// the read, W - 1 <= NS) and lists its own set bits.  No per-window comparison with the previous window, no byte-packed
// argmins, no start/end masks.  Applies when the mask fits a dword.
template <int MODE, int W, int NS>
constexpr bool frl_occ_form() { return MODE == MODE_MINIMIZER && W > 1 && W - 1 <= NS && NS + W - 1 <= 32; }

// Phase 3a: window argmins of the lane's NS windows and which of them exist.
// LIM_LAST != 0 (the compile-time geometry of scan_count_frl_kernel): every lane of a read owns NS windows but the last,
// which owns LIM_LAST (1..NS); 0: any geometry.
template <int MODE, int W, int NS, int LIM_LAST = 0, bool APPROX = false>
BL_DEV void phase_window_frl_a(const ScanParams& p, TileShared<MODE, W>& sh, int tid, ThreadState& st, const ThreadState* all, bool* tie = nullptr)
{
    uint32_t a[S];
    lane_window_argmin_frl<NS, (W > 1 ? W : 2), APPROX>(all, tid, st, st.jlane >= 0, a, tie);
    // windows of the read: window index j * NS + s must be below nwin; breaks only where the tile holds one
    uint32_t vmask = 0;
    if (st.jlane >= 0) {
        int lim = p.nwin - st.jlane * NS;
        lim = lim < 0 ? 0 : (lim > NS ? NS : lim);
        vmask = (1u << lim) - 1u;
    }
    const uint32_t tile_bad = sh.wave_bad[0] | sh.wave_bad[1] | sh.wave_bad[2] | sh.wave_bad[3];
    if (BL_COLD(tile_bad)) vmask &= frl_good_mask<NS>(p, sh.flags, st.lane_base, p.unit + (W > 0 ? W : p.w) - 1);
    st.vmask = vmask;
    if (frl_occ_form<MODE, W, NS>()) {
        uint32_t occ = 0;
        if (LIM_LAST == 0 || BL_COLD(tile_bad)) {
            BL_UNROLL
            for (int s = 0; s < NS; ++s) occ |= ((vmask >> s) & 1u) << (a[s] & 31u);
        } else {
            BL_UNROLL
            for (int s = 0; s < NS; ++s) occ |= 1u << (a[s] & 31u);
            // windows that do not exist (the read's last lane) set bits too, none below the argmin of the last window
            // that does (argmins never move left): clear everything above it
            const uint32_t a_sel = st.jlane == p.lpr - 1 ? a[LIM_LAST > 0 ? LIM_LAST - 1 : 0] : a[NS - 1];
            occ &= (2u << (a_sel & 31u)) - 1u;
            if (st.jlane < 0) occ = 0;
        }
        st.occ = occ;
        return;
    }
    BL_UNROLL
    for (int s = NS; s < S; ++s) a[s] = 0;
    uint32_t apk[4];
    BL_UNROLL
    for (int q = 0; q < 4; ++q) {  // byte s of apk = element index of the argmin of window s
        const uint32_t lo2 = byte_perm(a[4 * q + 1], a[4 * q], 0x0c0c0400u);
        const uint32_t hi2 = byte_perm(a[4 * q + 3], a[4 * q + 2], 0x04000c0cu);
        apk[q] = (lo2 | hi2) & 0x3f3f3f3fu;
    }
    st.apk0 = ((uint64_t)apk[1] << 32) | apk[0];
    st.apk1 = ((uint64_t)apk[3] << 32) | apk[2];
    st.a_first = a[0] & 63u;
    st.a_last = a[NS - 1] & 63u;
}

// Phase 3b: start / end decisions.  Returns starts | ends << 16.
template <int MODE, int W, int NS>
BL_DEV uint32_t phase_window_frl_b(const ScanParams& p, int tid, ThreadState& st, const ThreadState* all)
{
    if (frl_occ_form<MODE, W, NS>()) {  // take over the bits the previous lane found for this lane's elements
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
        (void)all;
        (void)tid;
        const uint32_t prev = dpp_prev32(st.occ);
#else
        const uint32_t prev = (tid & 63) > 0 ? all[tid - 1].occ : 0u;
#endif
        st.endm = 0;
        st.emit = (st.occ | (prev >> NS)) & ((1u << NS) - 1u);
        return (uint32_t)__builtin_popcount(st.emit);
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BL_CPU_EMU)
    (void)all;
    (void)tid;
    const uint32_t prev_last = dpp_prev32(st.a_last), prev_vm = dpp_prev32(st.vmask);
    const uint32_t next_first = dpp_next32(st.a_first), next_vm = dpp_next32(st.vmask);
#else
    const int lane = tid & 63;
    const uint32_t prev_last = lane > 0 ? all[tid - 1].a_last : 0xAAu, prev_vm = lane > 0 ? all[tid - 1].vmask : 0xAAAAu;
    const uint32_t next_first = lane < 63 ? all[tid + 1].a_first : 0x55u, next_vm = lane < 63 ? all[tid + 1].vmask : 0x5555u;
#endif
    const int j = st.jlane;
    const uint32_t apk[4] = {(uint32_t)st.apk0, (uint32_t)(st.apk0 >> 32), (uint32_t)st.apk1, (uint32_t)(st.apk1 >> 32)};
    // differ bit s: the argmin of window s is another occurrence than that of window s - 1 (the previous lane's last
    // window for s = 0: its element index moves by NS into this lane's frame; an index below NS there wraps to >= 112)
    const uint32_t p0 = (prev_last - (uint32_t)NS) & 0x7fu;
    uint32_t differ = 0;
    BL_UNROLL
    for (int q = 0; q < 4; ++q) {
        const uint32_t prv = q ? funnel_shr(apk[q], apk[q - 1], 24) : ((apk[0] << 8) | p0);
        const uint32_t x = apk[q] ^ prv;                             // bytes < 128
        const uint32_t nz = ((x + 0x7f7f7f7fu) >> 7) & 0x01010101u;  // 1 in every non-zero byte
        differ |= ((nz * 0x01020408u) >> 24) << (4 * q);             // byte b -> bit b
    }
    const uint32_t vm = st.vmask;
    const uint32_t pv = (j > 0) ? ((prev_vm >> (NS - 1)) & 1u) : 0u;  // the window before this lane's first, same read
    const uint32_t vprev = (vm << 1) | pv;
    st.emit = vm & (~vprev | differ);  // window s starts an occurrence
    st.endm = 0;
    if (MODE == MODE_SUPERKMER) {
        const uint32_t nv = (j >= 0 && j < p.lpr - 1) ? (next_vm & 1u) : 0u;  // the window after this lane's last, same read
        const uint32_t dn = ((next_first + (uint32_t)NS) != st.a_last) ? 1u : 0u;
        const uint32_t vnext = (vm >> 1) | (nv << (NS - 1));
        const uint32_t dnext = (differ >> 1) | (dn << (NS - 1));
        st.endm = vm & (~vnext | dnext);  // window s ends one
    }
    return (uint32_t)__builtin_popcount(st.emit) | ((uint32_t)__builtin_popcount(st.endm) << 16);
}

// ------------------------------------------------------------------------------------------------
// Phase 4: tile-local compaction.  List entries are FLAT positions relative to the tile's first staged chunk
// (< 16 * NCHUNK), not (wave, wave-relative position) pairs: pass 2 tells the two apart by p.frl.
template <int MODE, int W, int NS = 0>
BL_DEV void phase_list_frl(TileShared<MODE, W>& sh, const ThreadState& st, uint32_t excl_s, uint32_t excl_e)
{
    uint32_t m = st.emit;
    uint32_t r = excl_s;
    if (NS != 0 && frl_occ_form<MODE, W, (NS ? NS : 1)>()) {  // bit s: the lane's own element s is a minimizer occurrence
        while (m) {
            const int s = __builtin_ctz(m);
            m &= m - 1;
            sh.list_a[r] = (uint16_t)(st.lane_base + s);
            ++r;
        }
        return;
    }
    while (m) {
        const int s = __builtin_ctz(m);
        m &= m - 1;
        const uint32_t arel = (uint32_t)((s < 8 ? st.apk0 : st.apk1) >> (8 * (s & 7))) & 0xffu;
        sh.list_a[r] = (uint16_t)((uint32_t)st.lane_base + arel);
        if (MODE == MODE_SUPERKMER) sh.list_j[r] = (uint16_t)(st.lane_base + s);
        ++r;
    }
    if (MODE == MODE_SUPERKMER) {
        m = st.endm;
        r = excl_e;
        while (m) {
            const int s = __builtin_ctz(m);
            m &= m - 1;
            sh.list_e[r] = (uint16_t)(st.lane_base + s);
            ++r;
        }
    }
}

}  // namespace bl
