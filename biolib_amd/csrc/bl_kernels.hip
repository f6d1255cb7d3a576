// bl_kernels.hip — gfx950 kernels of the fused k-mer / minimizer scan and their launchers.
// Written for CDNA4 only (wave64, 256-thread workgroups, LDS-staged tiles); no portability layer.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "bl_scan_frl.hpp"
#include "bl_launch.hpp"

#ifndef BL_SY0_WAVES
#define BL_SY0_WAVES 2  // waves per SIMD the argmin syncmer kernels with the exact form inline are compiled for
#endif
#ifndef BL_SY2_WAVES
#define BL_SY2_WAVES 3  // ... and the argmin syncmer kernel whose exact form is deferred to scan_redo_kernel
#endif
#ifndef BL_CS_WAVES
#define BL_CS_WAVES 3  // waves per SIMD the closed-syncmer kernel is compiled for
#endif
#ifndef BL_POSAX_WAVES
#define BL_POSAX_WAVES 4  // ... and the position-tiled minimizer kernel that decides on murmur64_top (111 registers; compiled for 5 it spills 30 accesses and runs at the exact kernel's rate, 4 gives +5.5 %)
#endif
#ifndef BL_SKAX_WAVES
#define BL_SKAX_WAVES 4  // ... and the BASELINE super-k-mer kernel that decides on murmur64_top
#endif
#ifndef BL_C4_WAVES
#define BL_C4_WAVES 5  // ... and the BASELINE super-k-mer kernel (96 registers and two scratch accesses; at 4 — 97 registers — 1-2 % slower, A/B on one box)
#endif
#ifndef BL_CSRT_WAVES
#define BL_CSRT_WAVES 3  // ... and the closed-syncmer kernels for a window count given at run time
#endif

namespace bl {

// ------------------------------------------------------------------------------------------------
// wave / workgroup primitives

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
    // six DPP steps (the sequence LLVM's atomic optimizer uses on gfx9): inside each row of 16 lanes by row_shr 1, 2,
    // 4, 8, then lane 15 of every row into the following row (row_bcast15, rows 1 and 3), then lane 31 into the upper
    // half (row_bcast31, rows 2 and 3).  Lanes without a source add the `old` operand, 0.  (The __shfl_up form costs
    // ~40 VALU per lane around its ds_bpermute's, 2.5 per base.)
    (void)lane;
#ifndef BL_SCAN_BUILTIN_DPP
    // one v_add_u32_dpp per step: v = v + dpp(v), lanes without a source add 0 (bound_ctrl), rows outside the mask keep v.  Through
    // __builtin_amdgcn_update_dpp + add the compiler writes v_mov (the `old` operand) + v_mov_b32_dpp + v_add and a wait state per step.
    // (a DPP operand written by the instruction before needs two wait states; the assembler does not look inside the asm)
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return v;
#else
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
    return v;
#endif
}

// XOR of v over the wave, valid in LANE 63 ONLY (the inclusive-scan DPP sequence of wave_incl_scan_u32 with xor): the
// per-tile digest fold of pass 2 runs this three times per wave; the butterfly of shuffles cost ~100 VALU there.
__device__ __forceinline__ uint32_t wave_xor_to_last_u32(uint32_t v)
{
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ unsigned long long wave_xor_to_last_u64(unsigned long long v)
{
    return ((unsigned long long)wave_xor_to_last_u32((uint32_t)(v >> 32)) << 32) | wave_xor_to_last_u32((uint32_t)v);
}

__device__ __forceinline__ unsigned long long wave_xor_u64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// a value every lane of the wave holds alike, moved to scalar registers
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// exclusive scan over the workgroup of a value that packs two 16-bit counters (sums stay < 65536)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_tot, int tid, uint32_t& total)
{
    const int lane = tid & 63, wv = wave_index(tid);
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) {
        const uint32_t t = wave_tot[i];
        if (i < wv) before += t;
        all += t;
    }
    total = all;
    return before + incl - v;
}

// copy n u16 entries between LDS and the tile's global slot as dwords (both 4-byte aligned; the odd
// tail entry travels with a don't-care partner)
__device__ __forceinline__ void spill_list(uint16_t* dst, const uint16_t* src, uint32_t n, int tid)
{
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    const uint32_t nd = (n + 1) / 2;  // ~260 for a tile of short reads: one store per thread, a second for a few
    if ((uint32_t)tid < nd) d32[tid] = s32[tid];
#pragma unroll 1
    for (uint32_t i = TPB + tid; i < nd; i += TPB) d32[i] = s32[i];
}

__device__ __forceinline__ void fill_list(uint16_t* dst, const uint16_t* src, uint32_t n, int tid)
{
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
#pragma unroll 1
    for (uint32_t i = tid; i < (n + 1) / 2; i += TPB) d32[i] = s32[i];
}

// ------------------------------------------------------------------------------------------------
// Pass 1 of one tile: hash, window minimum, start/end decisions, tile-local compaction; leaves the
// tile's record counts and u16 lists in global scratch.  Tiles are independent: no ticket, no
// inter-workgroup wait, any dispatch order.
// SY (syncmer scans): 0 = tagged argmins with the exact form inline (also what scan_redo_kernel runs), 1 = closed syncmers
// (sliding minima), 2 = tagged argmins WITHOUT the exact form: a tile that met a prefix tie is listed for scan_redo_kernel;
// W = -8 / -16 with SY = 1: closed syncmers for a window count given at run time (phase_sync_closed_rt), two size groups
template <int MODE, int W, int SY = 0, int U = 0>
__device__ __forceinline__ void count_tile(const ScanParams& p, TileShared<MODE, W>& sh, uint32_t tile, int tid)
{
    const int64_t q0 = p.origin + (int64_t)tile * p.stride;
    phase_load<MODE, W>(p, sh, tid, q0);
    constexpr bool CS = SY == 1;
    if (SY != 0 && tid == 0) sh.redo = 0;
    if (MODE != MODE_SYNCMER) {  // hand the packed codes to pass 2 (0.26 B/base instead of re-reading and re-encoding 1 B/base there);
                                 // a syncmer record is a position: its pass 2 rebuilds nothing and reads no codes
        const int needed = staged_chunks(p);
        uint32_t* sc = p.slots_c + (size_t)tile * p.slot_chunks;
        if (tid < needed) sc[tid] = sh.codes[tid];  // own LDS entries: no barrier needed
        if (TPB + tid < needed) sc[TPB + tid] = sh.codes[TPB + tid];
    }
    __syncthreads();

    ThreadState st;
    constexpr bool DIRECT = MODE == MODE_SYNCMER && SY != 0 && U >= 1 && U <= 16 && U + W - 1 >= 16 && U + W - 1 <= 32;  // phase_hash_closed applies
    bool tie = false;
#ifdef BL_CLOSED_ON_HASHES  // A/B builds: the closed form on the hashes' own high dwords
    constexpr bool AP = false;
#else
    constexpr bool AP = DIRECT && CS;  // ... on murmur64_top (phase_hash_closed)
#endif
    constexpr bool MAX = MODE != MODE_SYNCMER && SY == 2;  // minimizer / super-k-mer scans: windows decided on murmur64_top, the exact form in scan_redo_kernel
    if (DIRECT) phase_hash_closed<MODE, W, (DIRECT ? U : 1), SY == 2, AP>(p, sh, tid, st, &tie);
    else phase_hash<MODE, W, (MODE != MODE_SYNCMER && U >= 1 && U <= 16 ? U : 0), U == 0, MAX>(p, sh, tid, st);

    uint32_t packed;
    if constexpr (MODE == MODE_SYNCMER && CS && W < 0) {  // closed syncmers for any (k, s): w by run time (W = -8: w <= 17, W = -16: 18 <= w <= 32)
        packed = phase_sync_closed_rt<MODE, (W == -16 ? -16 : -8)>(p, reinterpret_cast<TileShared<MODE, (W == -16 ? -16 : -8)>&>(sh), tid, q0, st, nullptr);
    } else if (MODE == MODE_SYNCMER && CS) {  // closed syncmers: sliding minima of the high dwords, no argmin
        bool undecided;
        packed = phase_sync_closed<MODE, (W > 1 ? W : 2), (DIRECT ? U : 0), AP>(p, reinterpret_cast<TileShared<MODE, (W > 1 ? W : 2)>&>(sh), tid, q0, st, nullptr, undecided);
        // equal high dwords somewhere in the wave: the tile is listed and counted again, in the exact form, by scan_redo_kernel — a
        // kernel of its own, because that form needs twice the registers and, inlined here, pushes spills into this kernel's hot path
        if (BL_COLD(wave_any(undecided)) && (tid & 63) == 0) sh.redo = 1;
    } else if (MODE == MODE_SYNCMER && SY == 2 && W > 1) {
        uint32_t af[S + 1];
        phase_sync_fwd<MODE, W, true>(p, sh, tid, st, nullptr, af, &tie);
        packed = phase_sync_rev<MODE, W, true>(p, sh, tid, q0, st, nullptr, af, &tie);
        if (BL_COLD(wave_any(tie)) && (tid & 63) == 0) sh.redo = 1;
    } else if (MODE == MODE_SYNCMER) {
        uint32_t af[S + 1];
        phase_sync_fwd<MODE, W>(p, sh, tid, st, nullptr, af);
        packed = phase_sync_rev<MODE, W>(p, sh, tid, q0, st, nullptr, af);
    } else if (MAX) {
        packed = phase_window<MODE, W, MAX>(p, sh, tid, q0, st, nullptr, &tie);
        if (BL_COLD(wave_any(tie)) && (tid & 63) == 0) sh.redo = 1;
    } else {
        packed = phase_window<MODE, W>(p, sh, tid, q0, st, nullptr);
    }

    uint32_t total;
    const uint32_t excl = block_excl_scan(packed, sh.wave_tot, tid, total);
    const uint32_t n_s = total & 0xffffu, n_e = total >> 16;  // a tile owns fewer than H positions: 16 bits suffice
    phase_list<MODE, W>(sh, tid, st, excl & 0xffffu, excl >> 16);
    if (tid == 0) p.tile_counts[tile] = (unsigned long long)n_s | ((unsigned long long)n_e << 32);
    __syncthreads();  // lists complete
    if (SY != 0 && tid == 0 && BL_COLD(sh.redo != 0)) p.redo_list[atomicAdd(p.redo_count, 1ull)] = tile;

    // spill the compacted lists: two u16 entries per 32-bit store (sub-dword global stores are not
    // write-combined on gfx950: 2-byte stores cost a 32-byte memory write each, measured 6.4 GB of
    // WRITE_SIZE per 1.5 Gbp launch for 0.4 GB of payload)
    const size_t slot = (size_t)tile * p.stride;  // multiple of 16 entries: dword aligned
    spill_list(p.slots_a + slot, sh.list_a, n_s, tid);
    if (MODE == MODE_SUPERKMER) {
        spill_list(p.slots_j + slot, sh.list_j, n_s, tid);
        spill_list(p.slots_e + slot, sh.list_e, n_e, tid);
    }
}

// Pass 1 of one tile in the read-tiled layout (fixed-length short reads, bl_scan_frl.hpp): same outputs as count_tile.
// APPROX: the windows are decided on murmur64_top (bl_scan_core.hpp), 7 instructions per hash cheaper than the hash; a tile in which
// some lane could not tell two keys apart is listed for scan_redo_frl_kernel, which runs this function without the flag.
template <int MODE, int W, int NS, int LIM_LAST, bool GENERIC, bool APPROX = false>
__device__ __forceinline__ void count_tile_frl(const ScanParams& p, TileShared<MODE, W>& sh, uint32_t tile, int tid)
{
    const int64_t q0 = tile_q0(p, tile);
    phase_load_frl<MODE, W>(p, sh, tid, q0);
    if (APPROX && tid == 0) sh.redo = 0;
    {   // hand the packed codes to pass 2
        uint32_t* sc = p.slots_c + (size_t)tile * p.slot_chunks;
        for (int c = tid; c < p.slot_chunks; c += TPB) sc[c] = sh.codes[c];  // own LDS entries: no barrier needed
    }
    __syncthreads();

    ThreadState st;
    bool tie = false;
    phase_hash_frl<MODE, W, NS, GENERIC, APPROX>(p, sh, tid, q0, tile, st);
    phase_window_frl_a<MODE, W, NS, LIM_LAST, APPROX>(p, sh, tid, st, nullptr, &tie);
    if (APPROX && BL_COLD(wave_any(tie)) && (tid & 63) == 0) sh.redo = 1;
    const uint32_t packed = phase_window_frl_b<MODE, W, NS>(p, tid, st, nullptr);

    uint32_t total;
    const uint32_t excl = block_excl_scan(packed, sh.wave_tot, tid, total);
    const uint32_t n_s = total & 0xffffu, n_e = total >> 16;
    phase_list_frl<MODE, W, NS>(sh, st, excl & 0xffffu, excl >> 16);
    if (tid == 0) p.tile_counts[tile] = (unsigned long long)n_s | ((unsigned long long)n_e << 32);
    __syncthreads();  // lists complete
    if (APPROX && tid == 0 && BL_COLD(sh.redo != 0)) p.redo_list[atomicAdd(p.redo_count, 1ull)] = tile;
    const size_t slot = (size_t)tile * p.stride;  // stride is a multiple of 4 entries: dword aligned
    spill_list(p.slots_a + slot, sh.list_a, n_s, tid);
    if (MODE == MODE_SUPERKMER) {
        spill_list(p.slots_j + slot, sh.list_j, n_s, tid);
        spill_list(p.slots_e + slot, sh.list_e, n_e, tid);
    }
}

// Pass 2 of one tile: reload the tile's 2-bit codes (spilled by pass 1) into LDS, rebuild each record from its u16 list entry
// (unit value, hash, position) and store it at the tile's global offset with coalesced stores.  The list entries are read
// straight from the tile's global slots — each exactly once, by the thread that builds the record — so the kernel's LDS
// footprint is the 2 KB of codes and its residency is set by launch_scan_emit's padding alone (it used to stage up to
// three 8 KB lists per tile, which, beside a hashing kernel that needs 12 KB per workgroup, decided who got the CU).
// LOOPED: called from scan_emit_kernel's tile loop (one barrier per tile whatever the tile holds; scheduling fences that hold the registers down)
template <int MODE, bool LOOPED = false>
__device__ __forceinline__ void emit_tile(const ScanParams& p, uint32_t* codes, uint32_t tile, int tid, Digest& dg)
{
    // one memory round trip for the common case: every load of the tile — counts, offsets, codes, the first 2 * TPB list entries
    // (speculatively: a tile of 150-bp reads holds ~600) — is issued before the first one is consumed
    const int64_t q0 = tile_q0(p, tile);
    const size_t slot = (size_t)tile * p.stride;
    // (uniform64: inside scan_emit_kernel's tile loop the compiler reads these through vector loads and keeps counts, offsets and every
    // comparison with them in vector registers — 34-40 registers instead of 23)
    const unsigned long long cnt = uniform64(p.tile_counts[tile]);
    const unsigned long long base = uniform64(p.tile_base[tile] + p.block_base[tile / SCAN_BLK]);
    const int needed = staged_chunks(p);
    const uint32_t* sc = p.slots_c + (size_t)tile * p.slot_chunks;
    const uint32_t c0 = (MODE != MODE_SYNCMER && tid < needed) ? sc[tid] : 0;
    const uint32_t c1 = (MODE != MODE_SYNCMER && TPB + tid < needed) ? sc[TPB + tid] : 0;
    const uint16_t* la = p.slots_a + slot;
    const uint16_t* lj = MODE == MODE_SUPERKMER ? p.slots_j + slot : nullptr;
    const bool in0 = tid < p.stride, in1 = TPB + tid < p.stride;  // inside the slot (its tail past n_s holds stale entries: never used)
    const uint32_t a0 = in0 ? la[tid] : 0, a1 = in1 ? la[TPB + tid] : 0;
    uint32_t j0 = 0, j1 = 0;
    if (MODE == MODE_SUPERKMER) {
        j0 = in0 ? lj[tid] : 0;
        j1 = in1 ? lj[TPB + tid] : 0;
    }
    const uint32_t n_s = (uint32_t)cnt, n_e = (uint32_t)(cnt >> 32);
    if (n_s == 0 && n_e == 0) {  // uniform for the workgroup
        if (LOOPED && MODE != MODE_SYNCMER) __syncthreads();  // (the caller's tile loop counts on one barrier per tile: scan_emit_kernel)
        return;
    }
    const uint64_t base_s = base & 0xffffffffull, base_e = base >> 32;
    if (MODE != MODE_SYNCMER) {  // (syncmers: no codes, no barrier — a record is its list entry's position)
        if (tid < needed) codes[tid] = c0;
        if (TPB + tid < needed) codes[TPB + tid] = c1;
        __syncthreads();
    }
    const bool fits = !BL_COLD(base_s + n_s > p.capacity);
    TileLists L{codes, la, lj, MODE == MODE_SUPERKMER ? p.slots_e + slot : nullptr, MODE == MODE_SUPERKMER ? p.slots_e + slot + p.stride : nullptr};
    const uint32_t d = (uint32_t)(base_s - base_e);  // 0 or 1 (see end_position)
    auto one = [&](uint32_t r, uint32_t ent, uint32_t ent_j) {
        const Record rec = emit_prepare<MODE, LOOPED>(p, codes, q0, ent, ent_j, dg);
        if (fits) emit_store<MODE, false>(p, rec, base_s + r);
        else emit_store<MODE, true>(p, rec, base_s + r);
        if (MODE == MODE_SUPERKMER && (p.out_size || p.out_records) && (fits || base_s + r < p.capacity)) {
            const int size = (int)(end_position(p, L, tile, q0, r + d, n_e) - (int64_t)rec.first + 1);
            if (p.out_size) p.out_size[base_s + r] = (uint8_t)size;
            if (p.out_records) emit_record(p, codes, needed - 1, q0, rec, size, base_s + r);
        }
    };
    // (fences: left alone the scheduler interleaves the two records for instruction-level parallelism — 30 registers instead of 22, and
    // this kernel's registers decide how many of its waves fit a SIMD beside the hashing pass: 96 x 4 or 5 waves of the 512 there are)
    if ((uint32_t)tid < n_s) one(tid, a0, j0);
    if (LOOPED) BL_SCHED_FENCE();
    if ((uint32_t)(TPB + tid) < n_s) one(TPB + tid, a1, j1);
    if (LOOPED) BL_SCHED_FENCE();
#pragma unroll 1
    for (uint32_t r = 2 * TPB + tid; r < n_s; r += TPB) one(r, la[r], MODE == MODE_SUPERKMER ? lj[r] : 0u);  // rarely any
}

// ------------------------------------------------------------------------------------------------
// The two kernels.  One workgroup per tile; a scan is  count -> tile prefix scan -> emit  on one stream.
// (Fusing pass 1 of one tile group with pass 2 of the previous one into a single launch — even
// workgroups counting, odd ones emitting — was measured SLOWER, 200 vs 290 Gbp/s: the emit
// workgroups inherit pass 1's 96-VGPR footprint and take residency away from the ALU-bound pass.)
// U (unit length) and C (canonical flag) specialise the BASELINE configurations at compile time: the
// parameter block is copied and the fields overwritten with constants, which the inlined phases fold
// (constant shifts and masks in the roller, no strand selects).  U = 0 / C = -1: taken from the arguments.
// SY: the syncmer form (count_tile): 1 = closed syncmers (offsets {0, W - 1}; phase_sync_closed), 2 = argmins with the exact form deferred
template <int MODE, int W, int U, int C, int SY = 0>
__global__ __launch_bounds__(TPB, (MODE == MODE_MINIMIZER && SY == 2 ? BL_POSAX_WAVES : MODE == MODE_SUPERKMER && SY == 2 ? BL_SKAX_WAVES : SY == 1 && W < 0 ? BL_CSRT_WAVES : SY == 1 ? BL_CS_WAVES : SY == 2 ? BL_SY2_WAVES : (MODE == MODE_SYNCMER && W > 0 ? BL_SY0_WAVES : MODE == MODE_SYNCMER || (MODE == MODE_SUPERKMER && W == -32) ? 2 : (W == -32 || (W < 0 && MODE == MODE_SUPERKMER) ? 3 : (W == -16 ? 5 : (W < 0 ? 4 : (W <= 11 ? 5 : (MODE == MODE_SUPERKMER && W == 17 && U == 15 ? BL_C4_WAVES : 4)))))))) void scan_count_kernel(const ScanParams pin, GroupRange g)
{
    __shared__ TileShared<MODE, W> sh;
    ScanParams p = pin;
    if (U != 0) p.unit = U;
    if (W > 0) {
        p.w = W;
        p.stride = NWAVE * (64 * S - 16 * ((W + 15) / 16));  // plan_scan's value, as a constant
    }
    if (C >= 0) p.canonical = C;
    if (blockIdx.x < g.count) count_tile<MODE, W, SY, U>(p, sh, g.first + blockIdx.x, threadIdx.x);
}

// The tiles a closed-syncmer pass 1 could not decide (p.redo_list), counted again in the exact argmin form: same outputs, written
// over what pass 1 left for them.  Runs between pass 1 and the prefix scan; clean data lists nothing and every workgroup leaves at once.
template <int MODE, int W, int U, int C>
__global__ __launch_bounds__(TPB, 2) void scan_redo_kernel(const ScanParams pin)
{
    __shared__ TileShared<MODE, W> sh;
    ScanParams p = pin;
    if (U != 0) p.unit = U;
    if (W > 0) {  // (W < 0: one of the run-time width groups, w and stride as given)
        p.w = W;
        p.stride = NWAVE * (64 * S - 16 * ((W + 15) / 16));
    }
    if (C >= 0) p.canonical = C;
    const unsigned long long n = *p.redo_count;
    for (unsigned long long i = blockIdx.x; i < n; i += gridDim.x) {
        count_tile<MODE, W>(p, sh, p.redo_list[i], threadIdx.x);
        __syncthreads();  // the lists in LDS have been spilled before the next tile overwrites them
    }
}

// Read-tiled pass 1.  L (read length) and U, C specialise the headline configuration as in scan_count_kernel; L fixes
// the whole lane -> (read, unit) map at compile time (lanes per read, reads per wave, windows per read).
template <int MODE, int W, int NS, int U, int L, int C>
__device__ __forceinline__ ScanParams frl_params(const ScanParams& pin)
{
    ScanParams p = pin;
    p.w = W;
    p.ns = NS;
    if (U != 0) p.unit = U;
    if (C >= 0) p.canonical = C;
    if (L != 0) {  // plan_scan_frl's values for (L, U, W, NS), as constants
        constexpr int nu = L - U + 1, lpr = (nu + S - 1) / S, rpw = 64 / (lpr > 0 ? lpr : 1);
        p.read_len = L;
        p.lpr = lpr;
        p.rpw = rpw;
        p.nwin = nu - W + 1;
        p.lpr_inv = (65536u + lpr - 1) / lpr;
        p.stride = NWAVE * rpw * L;
        p.slot_chunks = (15 + NWAVE * rpw * L + 15) / 16 + 3;
    }
    return p;
}
// compile-time geometry: every lane of a read owns NS windows but the last one (the launcher checks what this assumes)
template <int W, int NS, int U, int L>
constexpr int frl_lim_last() { return L != 0 ? (L - U + 1 - W + 1) - ((L - U + 1 + S - 1) / S - 1) * NS : 0; }

// APPROX: see count_tile_frl; the launcher follows such a launch with scan_redo_frl_kernel
#ifndef BL_FRL_WAVES
#define BL_FRL_WAVES 5  // waves per SIMD the read-tiled pass 1 with a window of at most 11 is compiled for (6: 80 registers, 60 bytes of scratch, 427 against 486 Gbp/s; 4: the same code as 5)
#endif
template <int MODE, int W, int NS, int U, int L, int C, bool APPROX = false>
__global__ __launch_bounds__(TPB, (W <= 11 ? BL_FRL_WAVES : 4)) void scan_count_frl_kernel(const ScanParams pin, GroupRange g)
{
    __shared__ TileShared<MODE, W> sh;
    const ScanParams p = frl_params<MODE, W, NS, U, L, C>(pin);
    constexpr int LIM_LAST = frl_lim_last<W, NS, U, L>();
    static_assert(L == 0 || (LIM_LAST >= 1 && LIM_LAST <= NS), "read-tiled geometry: the last lane of a read must own 1..NS windows");
    if (blockIdx.x < g.count) count_tile_frl<MODE, W, NS, LIM_LAST, U == 0, APPROX>(p, sh, g.first + blockIdx.x, threadIdx.x);
}

// The tiles an APPROX pass 1 listed, counted again on the hashes themselves (same outputs, written over what pass 1 left for them);
// between pass 1 and the prefix scan, like scan_redo_kernel.  Random reads list a tile in a few hundred.
template <int MODE, int W, int NS, int U, int L, int C>
__global__ __launch_bounds__(TPB, (W <= 11 ? 5 : 4)) void scan_redo_frl_kernel(const ScanParams pin)
{
    __shared__ TileShared<MODE, W> sh;
    const ScanParams p = frl_params<MODE, W, NS, U, L, C>(pin);
    const unsigned long long n = *p.redo_count;
    for (unsigned long long i = blockIdx.x; i < n; i += gridDim.x) {
        count_tile_frl<MODE, W, NS, frl_lim_last<W, NS, U, L>(), U == 0>(p, sh, p.redo_list[i], threadIdx.x);
        __syncthreads();  // the lists in LDS have been spilled before the next tile overwrites them
    }
}

// Minimizer scans: a workgroup takes BL_EMIT_TILES consecutive tiles, so that the digest's wave reductions and atomics, a third of this
// kernel's instructions when paid per tile, are paid once per workgroup.  No loads of a later tile are in flight while a tile is built
// (that form, tried in round 3, needs 52 registers for 22 and crowds the hashing pass); the codes ping-pong between two LDS buffers so
// that the loop needs no barrier of its own (tile k + 2 overwrites what tile k read only after every thread has passed tile k + 1's
// barrier), and the thread's digest words live in LDS between tiles (in registers they cost 34 instead of 28, i.e. 40 allocated for 32).
// Measured on C3 (two lanes, 20 steps, A/B on one box): the same 500 Gbp/s as one tile per workgroup — the scan is held by the card's
// power limit — with 2.3 fewer lane-instructions per base.  Super-k-mer and syncmer scans keep one tile per workgroup:
// their record code is longer, the loop took it from 32 to 45 registers and C4 from 391 to 384 Gbp/s.
// U, C, FRL: unit length, canonical flag and layout as compile-time constants for the BASELINE C3 shape (0 / -1 / -1: from the arguments).
#ifndef BL_EMIT_TILES
#define BL_EMIT_TILES 4
#endif
template <int MODE>
constexpr int emit_tiles() { return MODE == MODE_MINIMIZER ? BL_EMIT_TILES : 1; }

template <int MODE, int U = 0, int C = -1, int FRL = -1>
__global__ __launch_bounds__(TPB) void scan_emit_kernel(const ScanParams pin, GroupRange g)
{
    constexpr int K = emit_tiles<MODE>();
    __shared__ uint32_t codes[K > 1 ? 2 : 1][NCHUNK];
    const int tid = threadIdx.x;
    ScanParams p = pin;
    if (U != 0) p.unit = U;
    if (C >= 0) p.canonical = C;
    if (FRL >= 0) p.frl = FRL;
    Digest dg{0, 0, 0};
    if constexpr (K > 1) {
        __shared__ uint32_t acc[6][TPB];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][tid] = 0;
        const uint32_t t0 = blockIdx.x * K;
#pragma unroll 1
        for (uint32_t k = 0; k < (uint32_t)K && t0 + k < g.count; ++k) {
            Digest d1{0, 0, 0};
            emit_tile<MODE, true>(p, codes[k & 1], g.first + t0 + k, tid, d1);
            acc[0][tid] ^= (uint32_t)d1.xv; acc[1][tid] ^= (uint32_t)(d1.xv >> 32);
            acc[2][tid] ^= (uint32_t)d1.xh; acc[3][tid] ^= (uint32_t)(d1.xh >> 32);
            acc[4][tid] ^= (uint32_t)d1.xp; acc[5][tid] ^= (uint32_t)(d1.xp >> 32);
        }
        dg.xv = ((unsigned long long)acc[1][tid] << 32) | acc[0][tid];
        dg.xh = ((unsigned long long)acc[3][tid] << 32) | acc[2][tid];
        dg.xp = ((unsigned long long)acc[5][tid] << 32) | acc[4][tid];
    } else {
        if (blockIdx.x >= g.count) return;
        emit_tile<MODE, false>(p, codes[0], g.first + blockIdx.x, tid, dg);
    }

    // digest: wave reduce (DPP xor-scan), then one set of atomics per WAVE into a shard line.  Measured alternatives: an LDS stage
    // with two more barriers per tile (no gain); folding the 256 threads' words with LDS atomics on three addresses (-30 % on the
    // whole scan: same-address LDS atomics serialise)
    // (a syncmer record is a position: no value, no hash to fold — two of the three 64-bit wave reductions less, a third of that kernel)
    const unsigned long long xv = MODE == MODE_SYNCMER ? 0ull : wave_xor_to_last_u64(dg.xv), xh = MODE == MODE_SYNCMER ? 0ull : wave_xor_to_last_u64(dg.xh);
    const unsigned long long xp = wave_xor_to_last_u64(dg.xp);
    if ((tid & 63) == 63 && (xv | xh | xp)) {
        unsigned long long* shard = p.shards + 8 * ((blockIdx.x * NWAVE + (tid >> 6)) % NSHARD);
        if (MODE != MODE_SYNCMER) {
            atomicXor(&shard[1], xv);
            atomicXor(&shard[2], xh);
        }
        atomicXor(&shard[3], xp);
    }
}

// ------------------------------------------------------------------------------------------------
// Prefix scan over the tile counts of one group (two 32-bit counters packed in 64 bits; totals stay
// below 2^31).  Block b of the group covers tiles first + b*SCAN_BLK ...
__global__ __launch_bounds__(512) void tile_scan_local_kernel(const unsigned long long* counts, unsigned long long* tile_base,
                                                              unsigned long long* block_tot, uint32_t first_tile, uint32_t n_group)
{
    __shared__ unsigned long long wtot[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t rel = blockIdx.x * SCAN_BLK + tid * 4;  // tile index inside the group
    unsigned long long c[4], run = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i] = rel + i < n_group ? counts[first_tile + rel + i] : 0;
        run += c[i];
    }
    unsigned long long incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < wv) before += wtot[i];
        all += wtot[i];
    }
    unsigned long long ex = before + incl - run;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (rel + i < n_group) tile_base[first_tile + rel + i] = ex;
        ex += c[i];
    }
    if (tid == 0) block_tot[first_tile / SCAN_BLK + blockIdx.x] = all;
}

// one thread: exclusive scan of the group's block totals, continuing from the running total of the
// previous groups (*carry); the grand totals go to the digest
__global__ void tile_scan_top_kernel(const unsigned long long* block_tot, unsigned long long* block_base, uint32_t first_block,
                                     uint32_t n_blocks, unsigned long long* carry, unsigned long long* shards)
{
    // one wave: lane l owns the blocks [l*per, (l+1)*per); local sums -> wave scan -> local exclusive prefixes
    // (the serial form, ~180 dependent global loads, took 30 us per scan)
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const uint32_t lane = threadIdx.x, per = (n_blocks + 63) / 64;
    const uint32_t lo = lane * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += block_tot[first_block + i];
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    const unsigned long long before = *carry;
    unsigned long long run = before + incl - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        block_base[first_block + i] = run;
        run += block_tot[first_block + i];
    }
    const unsigned long long total = __shfl(incl, 63, 64);
    if (lane == 0) {
        *carry = before + total;
        shards[0] += total & 0xffffffffull;  // records (starts)
        shards[4] += total >> 32;            // group ends (super-k-mer mode)
    }
}

// ------------------------------------------------------------------------------------------------
// Dense k-mer scan kernel (C2): no windows, no compaction.
struct KmerShared {
    uint32_t codes[NCHUNK_POS];
    uint32_t flags[NCHUNK_POS];
    unsigned long long dig[4];
};

__global__ __launch_bounds__(TPB) void kmer_kernel(const KmerParams p)
{
    __shared__ KmerShared sh;
    const int tid = threadIdx.x;
    KmerAcc acc{0, 0, 0, 0};
    if (tid < 4) sh.dig[tid] = 0;
    ScanParams lp{};  // the staging code only looks at these four fields
    lp.bases = p.bases;
    lp.n_bases = p.n_bases;
    lp.start_bits = p.start_bits;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int64_t q0 = p.origin + (int64_t)tile * H;
        __syncthreads();
        stage_chunk(lp, sh.codes, sh.flags, tid, q0);
        if (tid < NCHUNK_POS - TPB) stage_chunk(lp, sh.codes, sh.flags, TPB + tid, q0);
        __syncthreads();
        kmer_thread(p, sh.codes, sh.flags, tid, q0, acc);
    }
    __syncthreads();
    const unsigned long long c = wave_sum_u64(acc.cnt), xv = wave_xor_u64(acc.xv), xh = wave_xor_u64(acc.xh), s = wave_sum_u64(acc.sh);
    if ((tid & 63) == 0) {
        atomicAdd(&sh.dig[0], c);
        atomicXor(&sh.dig[1], xv);
        atomicXor(&sh.dig[2], xh);
        atomicAdd(&sh.dig[3], s);
    }
    __syncthreads();
    if (tid < 4) {
        unsigned long long* shard = p.shards + 8 * (blockIdx.x % NSHARD);
        if (tid == 0 || tid == 3) atomicAdd(&shard[tid], sh.dig[tid]);
        else atomicXor(&shard[tid], sh.dig[tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// small helper kernels

// fold the NSHARD digest lines into result[0..7]; slots listed in add_mask are sums, the others XORs
__global__ void reduce_shards_kernel(const unsigned long long* shards, unsigned long long* result, uint32_t add_mask, const unsigned long long* redone)
{
    const int slot = threadIdx.x;
    if (slot == 8) result[8] = 0;
    if (slot == 9) result[5] = *redone;  // tiles pass 1 listed for a second run (no digest uses slot 5; written after the fold below by another lane: disjoint words)
    if (slot >= 8) return;
    if (slot == 5) return;
    unsigned long long acc = 0;
    for (int i = 0; i < NSHARD; ++i) {
        const unsigned long long v = shards[8 * i + slot];
        if ((add_mask >> slot) & 1) acc += v;
        else acc ^= v;
    }
    result[slot] = acc;
}

// SURVEY.md §8d generator: base[i] = "ACGT"[(splitmix64(seed + (i>>5)) >> (2*(i&31))) & 3]
// one thread = one splitmix word = 32 bases = two 16-byte stores
__global__ void synth_kernel(uint8_t* bases, uint64_t first, uint64_t n, uint64_t seed)
{
    const uint64_t wordi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // word index relative to `first` >> 5
    const uint64_t i0 = wordi * 32;                                           // first is a multiple of 32
    if (i0 >= n) return;
    uint64_t x = seed + ((first + i0) >> 5);
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    const uint32_t lut = 0x54474341u;  // 'A','C','G','T'
    uint32_t out[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t c4 = (uint32_t)(x >> (8 * q)) & 0xffu;  // 4 bases
        const uint32_t sel = (c4 & 3u) | (((c4 >> 2) & 3u) << 8) | (((c4 >> 4) & 3u) << 16) | (((c4 >> 6) & 3u) << 24);
        out[q] = __builtin_amdgcn_perm(lut, lut, sel);
    }
    if (i0 + 32 <= n) {
        uint4* dst = reinterpret_cast<uint4*>(bases + i0);
        dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
        dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    } else {
        for (uint64_t b = 0; i0 + b < n; ++b) bases[i0 + b] = (uint8_t)(out[b >> 2] >> (8 * (b & 3)));
    }
}

// sequence-start bit vector for fixed-length reads: bit p set <=> p % read_len == 0
__global__ void start_bits_fixed_kernel(uint32_t* bits, uint64_t n_words, uint64_t n_bases, uint64_t read_len)
{
    const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const uint64_t p0 = wi * 32;
    uint64_t r = p0 % read_len;
    uint64_t nxt = r == 0 ? p0 : p0 + (read_len - r);  // first multiple of read_len >= p0
    uint32_t w = 0;
    while (nxt < p0 + 32 && nxt < n_bases) {
        w |= 1u << (nxt - p0);
        nxt += read_len;
    }
    bits[wi] = w;
}

// sequence-start bit vector from an offsets array (bits pre-zeroed)
__global__ void start_bits_offsets_kernel(uint32_t* bits, const uint64_t* offsets, uint64_t n_seqs, uint64_t n_bases)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_seqs) return;
    const uint64_t p = offsets[q];
    if (p < n_bases && offsets[q + 1] > p) atomicOr(&bits[p >> 5], 1u << (p & 31));
}

// ------------------------------------------------------------------------------------------------
// launchers

static hipError_t launch_count_frl(int mode, const ScanParams& p, GroupRange g, hipStream_t stream)
{
    const dim3 grid(g.count), block(TPB);
    if (mode == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical && p.read_len == 150 && p.ns == 15 && p.rpw == 8) {
        if (p.redo_list && g.first == 0 && !p.exact_windows) {  // BASELINE C3 (exact_windows, bl_ctx_set_exact_windows: windows decided on the hashes themselves)
            hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 11, 15, 31, 150, 1, true>), grid, block, 0, stream, p, g);
            hipLaunchKernelGGL((scan_redo_frl_kernel<MODE_MINIMIZER, 11, 15, 31, 150, 1>), dim3(g.count < 512u ? g.count : 512u), block, 0, stream, p);
        } else {
            hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 11, 15, 31, 150, 1>), grid, block, 0, stream, p, g);
        }
        return hipGetLastError();
    }
    // the BASELINE shape (canonical 31-mers, window 11) on reads of ANY fixed length the planner gives 14, 15 or 16 units per lane for
    // (every length from 100 to 300 bp, 151 among them): the same kernel with the read geometry taken from the arguments
    if (mode == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical && p.ns >= 14 && p.ns <= 16) {
        const bool approx = p.redo_list && g.first == 0 && !p.exact_windows;
        const dim3 redo_grid(g.count < 512u ? g.count : 512u);
#define BL_FRL_SHAPE(NSV)                                                                                                                \
    if (approx) {                                                                                                                        \
        hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 11, NSV, 31, 0, 1, true>), grid, block, 0, stream, p, g);               \
        hipLaunchKernelGGL((scan_redo_frl_kernel<MODE_MINIMIZER, 11, NSV, 31, 0, 1>), redo_grid, block, 0, stream, p);                    \
    } else {                                                                                                                             \
        hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 11, NSV, 31, 0, 1>), grid, block, 0, stream, p, g);                     \
    }
        if (p.ns == 14) { BL_FRL_SHAPE(14) } else if (p.ns == 15) { BL_FRL_SHAPE(15) } else { BL_FRL_SHAPE(16) }
#undef BL_FRL_SHAPE
        return hipGetLastError();
    }
    if (p.ns != S) return hipErrorInvalidValue;  // the general kernels give every lane S unit starts
    if (mode == MODE_MINIMIZER) {
        switch (p.w) {
            case 5: hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 5, S, 0, 0, -1>), grid, block, 0, stream, p, g); break;
            case 10: hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 10, S, 0, 0, -1>), grid, block, 0, stream, p, g); break;
            case 11: hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 11, S, 0, 0, -1>), grid, block, 0, stream, p, g); break;
            case 19: hipLaunchKernelGGL((scan_count_frl_kernel<MODE_MINIMIZER, 19, S, 0, 0, -1>), grid, block, 0, stream, p, g); break;
            default: return hipErrorInvalidValue;
        }
    } else if (mode == MODE_SUPERKMER && p.w == 17) {
        hipLaunchKernelGGL((scan_count_frl_kernel<MODE_SUPERKMER, 17, S, 0, 0, -1>), grid, block, 0, stream, p, g);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_count_mode(const ScanParams& p, GroupRange g, hipStream_t stream)
{
    const dim3 grid(g.count), block(TPB);
    // the BASELINE.json configurations, fully specialised
    if (MODE == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical) {
#ifndef BL_NO_POSAX
        if (p.redo_list && g.first == 0 && !p.exact_windows) {  // long reads, contigs, reads of mixed lengths: windows decided on murmur64_top here too
            hipLaunchKernelGGL((scan_count_kernel<MODE_MINIMIZER, 11, 31, 1, 2>), grid, block, 0, stream, p, g);
            hipLaunchKernelGGL((scan_redo_kernel<MODE_MINIMIZER, 11, 31, 1>), dim3(g.count < 512u ? g.count : 512u), block, 0, stream, p);
            return hipGetLastError();
        }
#endif
        hipLaunchKernelGGL((scan_count_kernel<MODE, 11, 31, 1>), grid, block, 0, stream, p, g);
        return hipGetLastError();
    }
    if (MODE == MODE_SUPERKMER && p.w == 17 && p.unit == 15 && p.canonical) {
#ifdef BL_SKAX  // measured twice (rounds 3 and 4; this form: 112 registers, no spills, 17 % fewer static instructions): 398.5 / 399.9 against 399.0 / 401.6
               // Gbp/s for the exact kernel, A/B on one box — the super-k-mer scan is not limited by the hash's four instructions.  Not built by default.
        if (p.redo_list && g.first == 0 && !p.exact_windows) {  // BASELINE C4: windows decided on murmur64_top, the listed tiles again on the hashes
            hipLaunchKernelGGL((scan_count_kernel<MODE_SUPERKMER, 17, 15, 1, 2>), grid, block, 0, stream, p, g);
            hipLaunchKernelGGL((scan_redo_kernel<MODE_SUPERKMER, 17, 15, 1>), dim3(g.count < 512u ? g.count : 512u), block, 0, stream, p);
            return hipGetLastError();
        }
#endif
        hipLaunchKernelGGL((scan_count_kernel<MODE, 17, 15, 1>), grid, block, 0, stream, p, g);
        return hipGetLastError();
    }
    if (MODE == MODE_SYNCMER && p.w == 21 && p.unit == 11 && p.canonical) {
        const bool closed = (p.soff == 0 && p.eoff == 20) || (p.soff == 20 && p.eoff == 0);
        const unsigned redo_grid = g.count < 512u ? g.count : 512u;
        if (closed && !p.exact_windows && p.redo_list && g.first == 0) {  // BASELINE C5 (exact_windows: the argmin form)
            hipLaunchKernelGGL((scan_count_kernel<MODE_SYNCMER, 21, 11, 1, 1>), grid, block, 0, stream, p, g);
            hipLaunchKernelGGL((scan_redo_kernel<MODE_SYNCMER, 21, 11, 1>), dim3(redo_grid), block, 0, stream, p);
#ifndef BL_NO_SY2
        } else if (p.redo_list && g.first == 0) {  // any other pair of offsets: argmins, the exact form in the redo kernel
            hipLaunchKernelGGL((scan_count_kernel<MODE_SYNCMER, 21, 11, 1, 2>), grid, block, 0, stream, p, g);
            hipLaunchKernelGGL((scan_redo_kernel<MODE_SYNCMER, 21, 11, 1>), dim3(redo_grid), block, 0, stream, p);
#endif
        } else {
            hipLaunchKernelGGL((scan_count_kernel<MODE, 21, 11, 1>), grid, block, 0, stream, p, g);
        }
        return hipGetLastError();
    }
    if (MODE == MODE_SYNCMER && p.w <= 32 && !p.exact_windows && ((p.soff == 0 && p.eoff == p.w - 1) || (p.soff == p.w - 1 && p.eoff == 0))) {
        // closed syncmers of any (k, s): sliding minima over the hashes' high dwords, w by run time; a k-mer whose comparison meets equal
        // dwords is decided on the 64-bit hashes inside the kernel (short s-mers repeat within a window: no second kernel, no listed tiles)
        if (p.w <= 17) hipLaunchKernelGGL((scan_count_kernel<MODE_SYNCMER, -8, 0, -1, 1>), grid, block, 0, stream, p, g);
        else hipLaunchKernelGGL((scan_count_kernel<MODE_SYNCMER, -16, 0, -1, 1>), grid, block, 0, stream, p, g);
        return hipGetLastError();
    }
    if (MODE != MODE_SYNCMER && p.w >= 2 && p.w <= 32) {
        // minimizer and super-k-mer scans: every window width up to 32 has its own kernel — van Herk / Gil-Werman on packed keys in
        // registers, for minimizers the element-centric decisions up to 16.  (The sparse-table kernels below, which take the width at run
        // time, ran these widths at 220-340 Gbp/s where a width of its own gives 340-420; 2 x 27 more kernels cost the build half a minute.)
        constexpr int MM = MODE == MODE_SYNCMER ? MODE_MINIMIZER : MODE;  // never instantiated for syncmers
        switch (p.w) {
#define BL_W(WV) case WV: hipLaunchKernelGGL((scan_count_kernel<MM, WV, 0, -1>), grid, block, 0, stream, p, g); return hipGetLastError();
            BL_W(2) BL_W(3) BL_W(4) BL_W(5) BL_W(6) BL_W(7) BL_W(8) BL_W(9) BL_W(10) BL_W(11) BL_W(12) BL_W(13) BL_W(14) BL_W(15) BL_W(16)
            BL_W(17) BL_W(18) BL_W(19) BL_W(20) BL_W(21) BL_W(22) BL_W(23) BL_W(24) BL_W(25) BL_W(26) BL_W(27) BL_W(28) BL_W(29) BL_W(30)
            BL_W(31) BL_W(32)
#undef BL_W
        }
    }
    switch (p.w) {
        case 1: hipLaunchKernelGGL((scan_count_kernel<MODE, 1, 0, -1>), grid, block, 0, stream, p, g); break;
        case 11: hipLaunchKernelGGL((scan_count_kernel<MODE, 11, 0, -1>), grid, block, 0, stream, p, g); break;
        case 17: hipLaunchKernelGGL((scan_count_kernel<MODE, 17, 0, -1>), grid, block, 0, stream, p, g); break;
        case 21: hipLaunchKernelGGL((scan_count_kernel<MODE, 21, 0, -1>), grid, block, 0, stream, p, g); break;
        default:
            // runtime window size: sparse-table argmin in registers, one kernel per size group (minimizer and super-k-mer scans come
            // here with widths beyond 32 only)
            if constexpr (MODE == MODE_SYNCMER) {
                if (p.w <= 16) { hipLaunchKernelGGL((scan_count_kernel<MODE, -8, 0, -1>), grid, block, 0, stream, p, g); break; }
                if (p.w <= 32) { hipLaunchKernelGGL((scan_count_kernel<MODE, -16, 0, -1>), grid, block, 0, stream, p, g); break; }
            }
            hipLaunchKernelGGL((scan_count_kernel<MODE, -32, 0, -1>), grid, block, 0, stream, p, g);
            break;
    }
    return hipGetLastError();
}

// pass 1 over the tiles of group g
hipError_t launch_scan_count(int mode, const ScanParams& p, GroupRange g, hipStream_t stream)
{
    if (g.count == 0) return hipSuccess;
    if (p.frl) return launch_count_frl(mode, p, g, stream);
    switch (mode) {
        case MODE_MINIMIZER: return launch_count_mode<MODE_MINIMIZER>(p, g, stream);
        case MODE_SUPERKMER: return launch_count_mode<MODE_SUPERKMER>(p, g, stream);
        case MODE_SYNCMER: return launch_count_mode<MODE_SYNCMER>(p, g, stream);
    }
    return hipErrorInvalidValue;
}

// pass 2 over the tiles of group g (its prefix scan must have run)
// lds_per_wg: when non-zero, pad the workgroup's LDS footprint up to this many bytes with (unused) dynamic LDS.  Its only
// purpose is to cap how many emit workgroups a CU holds when the pass runs beside the next scan's pass 1 (two-lane
// contexts): uncapped, the memory-bound emit waves crowd the ALU-bound pass out of the SIMDs (349 vs 364 Gbp/s measured).
template <int MODE>
static void launch_emit_mode(const ScanParams& p, GroupRange g, hipStream_t stream, uint32_t lds_per_wg)
{
    constexpr int K = emit_tiles<MODE>();
    const uint32_t have = (uint32_t)(((K > 1 ? 2 : 1) * NCHUNK + (K > 1 ? 6 * TPB : 0)) * sizeof(uint32_t));  // the kernel's static LDS: code buffers, digest words
    const uint32_t pad = lds_per_wg > have ? lds_per_wg - have : 0;
    const dim3 grid((g.count + K - 1) / K), block(TPB);
    if (MODE == MODE_MINIMIZER && p.frl && p.unit == 31 && p.canonical) {  // BASELINE C3
        hipLaunchKernelGGL((scan_emit_kernel<MODE_MINIMIZER, 31, 1, 1>), grid, block, pad, stream, p, g);
        return;
    }
    hipLaunchKernelGGL((scan_emit_kernel<MODE>), grid, block, pad, stream, p, g);
}

hipError_t launch_scan_emit(int mode, const ScanParams& p, GroupRange g, hipStream_t stream, uint32_t lds_per_wg)
{
    if (g.count == 0) return hipSuccess;
    switch (mode) {
        case MODE_MINIMIZER: launch_emit_mode<MODE_MINIMIZER>(p, g, stream, lds_per_wg); break;
        case MODE_SUPERKMER: launch_emit_mode<MODE_SUPERKMER>(p, g, stream, lds_per_wg); break;
        case MODE_SYNCMER: launch_emit_mode<MODE_SYNCMER>(p, g, stream, lds_per_wg); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// prefix scan of the tile counts of group g (first tile must be a multiple of SCAN_BLK)
hipError_t launch_tile_scan(const ScanParams& p, GroupRange g, unsigned long long* block_tot, unsigned long long* carry, hipStream_t stream)
{
    if (g.count == 0) return hipSuccess;
    const uint32_t n_blocks = (g.count + SCAN_BLK - 1) / SCAN_BLK;
    hipLaunchKernelGGL(tile_scan_local_kernel, dim3(n_blocks), dim3(512), 0, stream, p.tile_counts, p.tile_base, block_tot, g.first, g.count);
    hipLaunchKernelGGL(tile_scan_top_kernel, dim3(1), dim3(64), 0, stream, block_tot, p.block_base, g.first / SCAN_BLK, n_blocks, carry,
                       p.shards);
    return hipGetLastError();
}

hipError_t launch_kmers(const KmerParams& p, int n_blocks, hipStream_t stream)
{
    if (p.n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kmer_kernel, dim3(n_blocks), dim3(TPB), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_reduce_shards(const unsigned long long* shards, unsigned long long* result, uint32_t add_mask, const unsigned long long* redone, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_shards_kernel, dim3(1), dim3(64), 0, stream, shards, result, add_mask, redone);
    return hipGetLastError();
}

hipError_t launch_synth(uint8_t* bases, uint64_t first, uint64_t n, uint64_t seed, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t words = (n + 31) / 32;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, bases, first, n, seed);
    return hipGetLastError();
}

hipError_t launch_start_bits_fixed(uint32_t* bits, uint64_t n_words, uint64_t n_bases, uint64_t read_len, hipStream_t stream)
{
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(start_bits_fixed_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, bits, n_words, n_bases, read_len);
    return hipGetLastError();
}

hipError_t launch_start_bits_offsets(uint32_t* bits, const uint64_t* offsets, uint64_t n_seqs, uint64_t n_bases, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(start_bits_offsets_kernel, dim3((unsigned)((n_seqs + 255) / 256)), dim3(256), 0, stream, bits, offsets, n_seqs, n_bases);
    return hipGetLastError();
}

}  // namespace bl

#ifdef BL_EXPERIMENT_COUNT_FALLBACK
extern "C" unsigned long long bl_dbg_read_fallbacks()
{
    unsigned long long v = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(bl::bl_dbg_fallbacks), sizeof(v));
    return v;
}
#endif
