// TEST INFRASTRUCTURE — CPU emulation of the HIP tile pipeline.
//
// Compiles biolib_amd/csrc/bl_scan_phases.hpp (the very functions the gfx950 kernels are made of)
// for the host with BL_CPU_EMU and runs them thread by thread, tile by tile, with the barriers of
// bl_kernels.hip replaced by loop boundaries.  Built with -fsanitize=address,undefined by
// tests/emu/Makefile so that indexing mistakes in the kernel logic are caught here, on the CPU,
// and not as a GPU fault.  This is NOT a product path and not a CPU fallback: nothing in
// biolib_amd/ loads it; only tests/test_emu_*.py does.
#define BL_CPU_EMU 1
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../biolib_amd/csrc/bl_scan_frl.hpp"

using namespace bl;

namespace {

std::vector<uint32_t> make_start_bits(uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len)
{
    std::vector<uint32_t> bits((n_bases + 31) / 32 + 4, 0u);
    if (offsets) {
        for (uint64_t q = 0; q < n_seqs; ++q) {
            const uint64_t p = offsets[q];
            if (p < n_bases && offsets[q + 1] > p) bits[p >> 5] |= 1u << (p & 31);
        }
    } else if (read_len) {
        for (uint64_t p = 0; p < n_bases; p += read_len) bits[p >> 5] |= 1u << (p & 31);
    } else if (n_bases) {
        bits[0] |= 1u;
    }
    return bits;
}

// pass 1 (scan_count_kernel) for every tile, the tile-count prefix scan, pass 2 (scan_emit_kernel)
int g_closed_redone = 0, g_sy2_redone = 0, g_pos_redone = 0;

// SY: the syncmer form of count_tile (bl_kernels.hip): 0 = argmins with the exact form inline, 1 = closed syncmers on murmur64_top,
// 2 = argmins from phase_hash_closed<BOTH> with the exact form deferred to a second run of the tile (scan_redo_kernel)
template <int MODE, int W, int SY = 0, int U = 0>
void run_tiles(ScanParams p, unsigned long long* result)
{
    constexpr bool CS = SY == 1;
    const size_t nt = (size_t)p.n_tiles;
    std::vector<unsigned long long> counts(nt), base(nt);
    std::vector<uint16_t> sa(nt * p.stride), sj(MODE == MODE_SUPERKMER ? nt * p.stride : 1), se(MODE == MODE_SUPERKMER ? nt * p.stride : 1);
    std::vector<uint32_t> sc(nt * p.slot_chunks);
    {
        auto* sh = new TileShared<MODE, W>();
        std::vector<ThreadState> st(TPB);
        std::vector<uint32_t> packed(TPB), excl(TPB);
        std::vector<uint32_t> af(TPB * (S + 1));
        for (size_t tile = 0; tile < nt; ++tile) {
            std::memset(sh, 0xA5, sizeof(*sh));  // poison: nothing may depend on stale LDS contents
            std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
            const int64_t q0 = p.origin + (int64_t)tile * p.stride;
            for (int tid = 0; tid < TPB; ++tid) phase_load<MODE, W>(p, *sh, tid, q0);
            for (int c = 0; c < staged_chunks(p); ++c) sc[tile * p.slot_chunks + c] = sh->codes[c];  // codes spill
            constexpr int CSU = (MODE == MODE_SYNCMER && SY != 0 && W > 0) ? 11 : 0;  // the closed-syncmer kernels are instantiated for s = 11 (launch_count_mode)
            bool sy2_tie = false;
            constexpr bool RT = MODE == MODE_SYNCMER && W < 0 && SY == 1;  // closed syncmers, window count by run time (count_tile)
            if (RT) {
                for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W>(p, *sh, tid, st[tid]);
            } else if (CSU && CS) {
                for (int tid = 0; tid < TPB; ++tid) phase_hash_closed<MODE, W, (CSU ? CSU : 1), false, true>(p, *sh, tid, st[tid]);
            } else if (CSU) {
                for (int tid = 0; tid < TPB; ++tid) phase_hash_closed<MODE, W, (CSU ? CSU : 1), true, false>(p, *sh, tid, st[tid], &sy2_tie);
            } else if (MODE != MODE_SYNCMER && SY == 2) {  // count_tile's MAX form: windows decided on murmur64_top
                for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W, (MODE != MODE_SYNCMER ? U : 0), false, true>(p, *sh, tid, st[tid]);
            } else {
                for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W, (MODE != MODE_SYNCMER ? U : 0)>(p, *sh, tid, st[tid]);
            }
            if constexpr (RT) {
                for (int tid = 0; tid < TPB; ++tid) packed[tid] = phase_sync_closed_rt<MODE, (W == -16 ? -16 : -8)>(p, reinterpret_cast<TileShared<MODE, (W == -16 ? -16 : -8)>&>(*sh), tid, q0, st[tid], st.data());
            } else if (MODE == MODE_SYNCMER && CS) {  // closed syncmers on murmur64_top; a tile with an undecided lane again in the argmin form (scan_redo_kernel)
                bool any = false;
                for (int tid = 0; tid < TPB; ++tid) {
                    bool undecided;
                    packed[tid] = phase_sync_closed<MODE, (W > 1 ? W : 2), CSU, CSU != 0>(p, reinterpret_cast<TileShared<MODE, (W > 1 ? W : 2)>&>(*sh), tid, q0, st[tid], st.data(), undecided);
                    any |= undecided;
                }
                if (any) {
                    ++g_closed_redone;
                    std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
                    for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W>(p, *sh, tid, st[tid]);
                    for (int tid = 0; tid < TPB; ++tid) phase_sync_fwd<MODE, W>(p, *sh, tid, st[tid], st.data(), &af[tid * (S + 1)]);
                    for (int tid = 0; tid < TPB; ++tid)
                        packed[tid] = phase_sync_rev<MODE, W>(p, *sh, tid, q0, st[tid], st.data(), &af[tid * (S + 1)]);
                }
            } else if (MODE == MODE_SYNCMER && SY == 2) {  // count_tile's SY = 2 branch: ties reported, the whole tile decided again exactly
                for (int tid = 0; tid < TPB; ++tid) phase_sync_fwd<MODE, W, true>(p, *sh, tid, st[tid], st.data(), &af[tid * (S + 1)], &sy2_tie);
                for (int tid = 0; tid < TPB; ++tid)
                    packed[tid] = phase_sync_rev<MODE, W, true>(p, *sh, tid, q0, st[tid], st.data(), &af[tid * (S + 1)], &sy2_tie);
                if (sy2_tie) {
                    ++g_sy2_redone;
                    std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
                    for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W>(p, *sh, tid, st[tid]);
                    for (int tid = 0; tid < TPB; ++tid) phase_sync_fwd<MODE, W>(p, *sh, tid, st[tid], st.data(), &af[tid * (S + 1)]);
                    for (int tid = 0; tid < TPB; ++tid)
                        packed[tid] = phase_sync_rev<MODE, W>(p, *sh, tid, q0, st[tid], st.data(), &af[tid * (S + 1)]);
                }
            } else if (MODE == MODE_SYNCMER) {
                for (int tid = 0; tid < TPB; ++tid) phase_sync_fwd<MODE, W>(p, *sh, tid, st[tid], st.data(), &af[tid * (S + 1)]);
                for (int tid = 0; tid < TPB; ++tid)
                    packed[tid] = phase_sync_rev<MODE, W>(p, *sh, tid, q0, st[tid], st.data(), &af[tid * (S + 1)]);
            } else if (MODE != MODE_SYNCMER && SY == 2) {
                bool tie = false;
                for (int tid = 0; tid < TPB; ++tid) packed[tid] = phase_window<MODE, W, MODE != MODE_SYNCMER && SY == 2>(p, *sh, tid, q0, st[tid], st.data(), &tie);
                if (tie) {  // scan_redo_kernel<MINIMIZER, W, U, C>: the tile again on the hashes themselves
                    ++g_pos_redone;
                    std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
                    for (int tid = 0; tid < TPB; ++tid) phase_hash<MODE, W, (MODE != MODE_SYNCMER ? U : 0)>(p, *sh, tid, st[tid]);
                    for (int tid = 0; tid < TPB; ++tid) packed[tid] = phase_window<MODE, W>(p, *sh, tid, q0, st[tid], st.data());
                }
            } else {
                for (int tid = 0; tid < TPB; ++tid) packed[tid] = phase_window<MODE, W>(p, *sh, tid, q0, st[tid], st.data());
            }
            uint32_t run = 0;
            for (int tid = 0; tid < TPB; ++tid) {
                excl[tid] = run;
                run += packed[tid];
            }
            const uint32_t n_s = run & 0xffffu, n_e = run >> 16;
            for (int tid = 0; tid < TPB; ++tid) phase_list<MODE, W>(*sh, tid, st[tid], excl[tid] & 0xffffu, excl[tid] >> 16);
            counts[tile] = (unsigned long long)n_s | ((unsigned long long)n_e << 32);
            for (uint32_t r = 0; r < n_s; ++r) {
                sa[tile * p.stride + r] = sh->list_a[r];
                if (MODE == MODE_SUPERKMER) sj[tile * p.stride + r] = sh->list_j[r];
            }
            if (MODE == MODE_SUPERKMER)
                for (uint32_t r = 0; r < n_e; ++r) se[tile * p.stride + r] = sh->list_e[r];
        }
        delete sh;
    }
    unsigned long long run = 0;
    for (size_t tile = 0; tile < nt; ++tile) {
        base[tile] = run;
        run += counts[tile];
    }
    Digest dg{0, 0, 0};
    {
        auto* sh = new TileShared<MODE, 1>();
        for (size_t tile = 0; tile < nt; ++tile) {
            const uint32_t n_s = (uint32_t)counts[tile], n_e = (uint32_t)(counts[tile] >> 32);
            if (n_s == 0 && n_e == 0) continue;
            std::memset(sh, 0xA5, sizeof(*sh));
            const int64_t q0 = p.origin + (int64_t)tile * p.stride;
            for (int c = 0; c < staged_chunks(p); ++c) sh->codes[c] = sc[tile * p.slot_chunks + c];  // pass 2 reloads the codes
            for (uint32_t r = 0; r < n_s; ++r) {
                sh->list_a[r] = sa[tile * p.stride + r];
                if (MODE == MODE_SUPERKMER) sh->list_j[r] = sj[tile * p.stride + r];
            }
            if (MODE == MODE_SUPERKMER)
                for (uint32_t r = 0; r < n_e; ++r) sh->list_e[r] = se[tile * p.stride + r];
            const TileLists lists{sh->codes, sh->list_a, sh->list_j, sh->list_e, MODE == MODE_SUPERKMER && tile + 1 < nt ? &se[(tile + 1) * p.stride] : sh->list_e};
            for (int tid = 0; tid < TPB; ++tid)
                phase_emit<MODE>(p, lists, (uint32_t)tile, tid, q0, n_s, n_e, base[tile] & 0xffffffffull, base[tile] >> 32, dg);
        }
        delete sh;
    }
    result[0] = run & 0xffffffffull;
    result[1] = dg.xv;
    result[2] = dg.xh;
    result[3] = dg.xp;
    result[4] = run >> 32;
}

// the read-tiled pass 1 (scan_count_frl_kernel), then the common prefix scan and pass 2
// APPROX: pass 1 on murmur64_top, and a tile in which some lane reports a tie decided again on the hashes (scan_redo_frl_kernel)
int g_frl_redone = 0, g_frl_tiles = 0;
template <int MODE, int W, int NS, int LIM_LAST = 0, bool APPROX = false>
void run_tiles_frl(ScanParams p, unsigned long long* result)
{
    const size_t nt = (size_t)p.n_tiles;
    std::vector<unsigned long long> counts(nt), base(nt);
    std::vector<uint16_t> sa(nt * p.stride), sj(MODE == MODE_SUPERKMER ? nt * p.stride : 1), se(MODE == MODE_SUPERKMER ? nt * p.stride : 1);
    std::vector<uint32_t> sc(nt * p.slot_chunks);
    {
        auto* sh = new TileShared<MODE, W>();
        std::vector<ThreadState> st(TPB);
        std::vector<uint32_t> packed(TPB), excl(TPB);
        for (size_t tile = 0; tile < nt; ++tile) {
            std::memset(sh, 0xA5, sizeof(*sh));
            std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
            const int64_t q0 = tile_q0(p, (uint32_t)tile);
            for (int tid = 0; tid < TPB; ++tid) phase_load_frl<MODE, W>(p, *sh, tid, q0);
            for (int c = 0; c < p.slot_chunks; ++c) sc[tile * p.slot_chunks + c] = sh->codes[c];
            bool tie = false;
            for (int tid = 0; tid < TPB; ++tid) phase_hash_frl<MODE, W, NS, false, APPROX>(p, *sh, tid, q0, (uint32_t)tile, st[tid]);
            for (int tid = 0; tid < TPB; ++tid) phase_window_frl_a<MODE, W, NS, LIM_LAST, APPROX>(p, *sh, tid, st[tid], st.data(), &tie);
            ++g_frl_tiles;
            if (APPROX && tie) {
                ++g_frl_redone;
                std::memset(st.data(), 0x5A, st.size() * sizeof(ThreadState));
                for (int tid = 0; tid < TPB; ++tid) phase_hash_frl<MODE, W, NS>(p, *sh, tid, q0, (uint32_t)tile, st[tid]);
                for (int tid = 0; tid < TPB; ++tid) phase_window_frl_a<MODE, W, NS, LIM_LAST>(p, *sh, tid, st[tid], st.data());
            }
            for (int tid = 0; tid < TPB; ++tid) packed[tid] = phase_window_frl_b<MODE, W, NS>(p, tid, st[tid], st.data());
            uint32_t run = 0;
            for (int tid = 0; tid < TPB; ++tid) {
                excl[tid] = run;
                run += packed[tid];
            }
            const uint32_t n_s = run & 0xffffu, n_e = run >> 16;
            for (int tid = 0; tid < TPB; ++tid) phase_list_frl<MODE, W, NS>(*sh, st[tid], excl[tid] & 0xffffu, excl[tid] >> 16);
            counts[tile] = (unsigned long long)n_s | ((unsigned long long)n_e << 32);
            for (uint32_t r = 0; r < n_s; ++r) {
                sa[tile * p.stride + r] = sh->list_a[r];
                if (MODE == MODE_SUPERKMER) sj[tile * p.stride + r] = sh->list_j[r];
            }
            if (MODE == MODE_SUPERKMER)
                for (uint32_t r = 0; r < n_e; ++r) se[tile * p.stride + r] = sh->list_e[r];
        }
        delete sh;
    }
    unsigned long long run = 0;
    for (size_t tile = 0; tile < nt; ++tile) {
        base[tile] = run;
        run += counts[tile];
    }
    Digest dg{0, 0, 0};
    {
        auto* sh = new TileShared<MODE, 1>();
        for (size_t tile = 0; tile < nt; ++tile) {
            const uint32_t n_s = (uint32_t)counts[tile], n_e = (uint32_t)(counts[tile] >> 32);
            if (n_s == 0 && n_e == 0) continue;
            std::memset(sh, 0xA5, sizeof(*sh));
            const int64_t q0 = tile_q0(p, (uint32_t)tile);
            for (int c = 0; c < p.slot_chunks; ++c) sh->codes[c] = sc[tile * p.slot_chunks + c];
            for (uint32_t r = 0; r < n_s; ++r) {
                sh->list_a[r] = sa[tile * p.stride + r];
                if (MODE == MODE_SUPERKMER) sh->list_j[r] = sj[tile * p.stride + r];
            }
            if (MODE == MODE_SUPERKMER)
                for (uint32_t r = 0; r < n_e; ++r) sh->list_e[r] = se[tile * p.stride + r];
            const TileLists lists{sh->codes, sh->list_a, sh->list_j, sh->list_e, MODE == MODE_SUPERKMER && tile + 1 < nt ? &se[(tile + 1) * p.stride] : sh->list_e};
            for (int tid = 0; tid < TPB; ++tid)
                phase_emit<MODE>(p, lists, (uint32_t)tile, tid, q0, n_s, n_e, base[tile] & 0xffffffffull, base[tile] >> 32, dg);
        }
        delete sh;
    }
    result[0] = run & 0xffffffffull;
    result[1] = dg.xv;
    result[2] = dg.xh;
    result[3] = dg.xp;
    result[4] = run >> 32;
}

// the instantiations launch_count_frl (bl_kernels.hip) makes
template <int MODE>
void run_mode_frl(const ScanParams& p, unsigned long long* result)
{
    if (MODE == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical && p.read_len == 150 && p.ns == 15 && p.rpw == 8) {
        run_tiles_frl<MODE_MINIMIZER, 11, 15, 5, true>(p, result);  // the BASELINE C3 kernel
        return;
    }
    if (MODE == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical && p.ns >= 14 && p.ns <= 16) {  // the same shape on other read lengths
        if (p.ns == 14) run_tiles_frl<MODE_MINIMIZER, 11, 14, 0, true>(p, result);
        else if (p.ns == 15) run_tiles_frl<MODE_MINIMIZER, 11, 15, 0, true>(p, result);
        else run_tiles_frl<MODE_MINIMIZER, 11, 16, 0, true>(p, result);
        return;
    }
    if (MODE == MODE_MINIMIZER) {
        switch (p.w) {
            case 5: run_tiles_frl<MODE_MINIMIZER, 5, S>(p, result); return;
            case 10: run_tiles_frl<MODE_MINIMIZER, 10, S>(p, result); return;
            case 11: run_tiles_frl<MODE_MINIMIZER, 11, S>(p, result); return;
            case 19: run_tiles_frl<MODE_MINIMIZER, 19, S>(p, result); return;
        }
    }
    if (MODE == MODE_SUPERKMER && p.w == 17) { run_tiles_frl<MODE_SUPERKMER, 17, S>(p, result); return; }
    std::abort();
}

template <int MODE>
void run_mode(const ScanParams& p, unsigned long long* result)
{
    if (p.frl) { run_mode_frl<MODE == MODE_SYNCMER ? MODE_MINIMIZER : MODE>(p, result); return; }
    if (MODE == MODE_SYNCMER && p.w <= 32 && ((p.soff == 0 && p.eoff == p.w - 1) || (p.soff == p.w - 1 && p.eoff == 0)) &&
        !(p.w == 21 && p.unit == 11 && p.canonical)) {  // closed syncmers of any (k, s) (launch_count_mode; the (31, 11) shape has its own kernel)
        if (p.w <= 17) run_tiles<MODE_SYNCMER, -8, 1>(p, result);
        else run_tiles<MODE_SYNCMER, -16, 1>(p, result);
        return;
    }
    if (MODE == MODE_MINIMIZER && p.w == 11 && p.unit == 31 && p.canonical) {  // the BASELINE shape, position-tiled: decided on murmur64_top (launch_count_mode)
        run_tiles<MODE_MINIMIZER, 11, 2>(p, result);
        return;
    }
    if (MODE == MODE_SUPERKMER && p.w == 17 && p.unit == 15 && p.canonical) { run_tiles<MODE_SUPERKMER, 17, 0, 15>(p, result); return; }  // the BASELINE C4 kernel
    if (MODE != MODE_SYNCMER && p.w >= 2 && p.w <= 32) {  // a kernel per width (launch_count_mode)
        constexpr int MM = MODE == MODE_SYNCMER ? MODE_MINIMIZER : MODE;
        switch (p.w) {
#define BL_W(WV) case WV: run_tiles<MM, WV>(p, result); return;
            BL_W(2) BL_W(3) BL_W(4) BL_W(5) BL_W(6) BL_W(7) BL_W(8) BL_W(9) BL_W(10) BL_W(11) BL_W(12) BL_W(13) BL_W(14) BL_W(15) BL_W(16)
            BL_W(17) BL_W(18) BL_W(19) BL_W(20) BL_W(21) BL_W(22) BL_W(23) BL_W(24) BL_W(25) BL_W(26) BL_W(27) BL_W(28) BL_W(29) BL_W(30)
            BL_W(31) BL_W(32)
#undef BL_W
        }
    }
    switch (p.w) {
        case 1: run_tiles<MODE, 1>(p, result); break;
        case 5: if (MODE != MODE_SYNCMER) { run_tiles<MODE, 5>(p, result); break; } run_tiles<MODE, -8>(p, result); break;
        case 10: if (MODE != MODE_SYNCMER) { run_tiles<MODE, 10>(p, result); break; } run_tiles<MODE, -8>(p, result); break;
        case 19: if (MODE != MODE_SYNCMER) { run_tiles<MODE, 19>(p, result); break; } run_tiles<MODE, -16>(p, result); break;
        case 11: run_tiles<MODE, 11>(p, result); break;
        case 17:
            if (MODE == MODE_SUPERKMER && p.unit == 15 && p.canonical) { run_tiles<MODE_SUPERKMER, 17, 0, 15>(p, result); break; }  // the BASELINE C4 kernel
            run_tiles<MODE, 17>(p, result);
            break;
        case 21:
            if (MODE == MODE_SYNCMER && p.unit == 11 && p.canonical && ((p.soff == 0 && p.eoff == 20) || (p.soff == 20 && p.eoff == 0))) {
                run_tiles<MODE_SYNCMER, 21, 1>(p, result);  // the BASELINE C5 kernel (launch_count_mode)
                break;
            }
            if (MODE == MODE_SYNCMER && p.unit == 11 && p.canonical) {
                run_tiles<MODE_SYNCMER, 21, 2>(p, result);  // any other pair of offsets: argmins, the exact form deferred (launch_count_mode)
                break;
            }
            run_tiles<MODE, 21>(p, result);
            break;
        default:
            if (p.w <= 16) run_tiles<MODE, -8>(p, result);
            else if (p.w <= 32) run_tiles<MODE, -16>(p, result);
            else run_tiles<MODE, -32>(p, result);
            break;
    }
}

int g_frl_scans = 0;  // how many scans took the read-tiled path (the selftest checks that it is exercised at all)

void fill_common(ScanParams& p, const uint8_t* bases, uint64_t n_bases, const uint32_t* bits, int mode, uint64_t first, uint64_t n,
                 unsigned unit, unsigned w, uint64_t seed, unsigned flags, uint64_t read_len = 0)
{
    uint64_t end = n == 0 ? n_bases : first + n;
    if (end > n_bases) end = n_bases;
    p.bases = bases;
    p.n_bases = (int64_t)n_bases;
    p.start_bits = bits;
    plan_scan(mode, (int64_t)first, (int64_t)end, (int)w, p);
    if (read_len && !p.use_threshold && frl_width_built(mode, (int)w)) {  // the decision of bl_capi.hip: scan_windows
        if (plan_scan_frl_for(mode, (int64_t)first, (int64_t)end, (int64_t)n_bases, (int64_t)read_len, (int)unit, (int)w, (flags & 1) != 0, p)) ++g_frl_scans;
    }
    p.unit = (int)unit;
    p.w = (int)w;
    p.seed = (uint32_t)seed;
    p.canonical = (flags & 1) ? 1 : 0;
    p.drop_last = (flags & 2) ? 1 : 0;
}

}  // namespace

extern "C" {

// The `bases` buffer must be readable for n_bases bytes only — the harness copies it into a
// 16-byte aligned buffer of exactly the size the product allocates (n + 64) so ASan sees the
// same bounds the device has.
struct EmuBatch {
    uint8_t* bases;  // exactly n_bases bytes, 16-byte aligned: ASan sees the bounds the kernel must respect
    uint64_t n_bases;
    std::vector<uint32_t> bits;
    bool single;
    uint64_t read_len;  // != 0: fixed-length reads (the read-tiled layout may apply)
};

EmuBatch* emu_batch(const uint8_t* bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len)
{
    auto* b = new EmuBatch();
    void* mem = nullptr;
    if (posix_memalign(&mem, 16, n_bases ? n_bases : 1) != 0) return nullptr;
    b->bases = static_cast<uint8_t*>(mem);
    if (n_bases) std::memcpy(b->bases, bases, n_bases);
    b->n_bases = n_bases;
    b->single = !offsets && (read_len == 0 || read_len >= n_bases);
    b->read_len = (!offsets && !b->single) ? read_len : 0;
    if (!b->single) b->bits = make_start_bits(n_bases, offsets, n_seqs, read_len);
    return b;
}

void emu_batch_free(EmuBatch* b)
{
    if (b) free(b->bases);
    delete b;
}

void emu_minimizers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned unit, unsigned w, uint64_t seed, unsigned flags,
                    uint64_t* out_value, uint64_t* out_pos, uint64_t* out_hash, uint64_t capacity, unsigned long long* result)
{
    ScanParams p{};
    fill_common(p, b->bases, b->n_bases, b->single ? nullptr : b->bits.data(), MODE_MINIMIZER, first, n, unit, w, seed, flags, b->read_len);
    p.out_value = out_value;
    p.out_pos = out_pos;
    p.out_hash = out_hash;
    p.capacity = capacity;
    std::memset(result, 0, 8 * sizeof(unsigned long long));
    run_mode<MODE_MINIMIZER>(p, result);
}

// murmur64_top against murmur64 for n keys: how many tops are neither T nor T - 1 (T = murmur64 >> 32; must be 0), how many are T - 1
void emu_top_check(const uint64_t* keys, uint64_t n, uint32_t seed, uint64_t* bad, uint64_t* below)
{
    *bad = *below = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t t = (uint32_t)(murmur64(keys[i], seed) >> 32), s = murmur64_top(keys[i], seed);
        if (t - s > 1u) ++*bad;
        if (t - s == 1u) ++*below;
    }
}
int emu_frl_scans() { return g_frl_scans; }
int emu_frl_redone() { return g_frl_redone; }
int emu_closed_redone() { return g_closed_redone; }
int emu_sy2_redone() { return g_sy2_redone; }
int emu_pos_redone() { return g_pos_redone; }
int emu_frl_tiles() { return g_frl_tiles; }

void emu_hash_sample(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, uint64_t seed, uint64_t threshold, unsigned flags,
                     uint64_t* out_value, uint64_t* out_pos, uint64_t* out_hash, uint64_t capacity, unsigned long long* result)
{
    ScanParams p{};
    fill_common(p, b->bases, b->n_bases, b->single ? nullptr : b->bits.data(), MODE_MINIMIZER, first, n, k, 1, seed, flags);
    p.use_threshold = 1;
    p.hash_below = threshold;
    p.out_value = out_value;
    p.out_pos = out_pos;
    p.out_hash = out_hash;
    p.capacity = capacity;
    std::memset(result, 0, 8 * sizeof(unsigned long long));
    run_mode<MODE_MINIMIZER>(p, result);
}

void emu_super_kmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, unsigned m, uint64_t seed, unsigned flags,
                     uint64_t* out_min, uint64_t* out_first, uint8_t* out_mmpos, uint8_t* out_size, uint64_t* out_hash, uint64_t capacity,
                     unsigned long long* result)
{
    ScanParams p{};
    fill_common(p, b->bases, b->n_bases, b->single ? nullptr : b->bits.data(), MODE_SUPERKMER, first, n, m, k - m + 1, seed, flags, b->read_len);
    p.out_value = out_min;
    p.out_first = out_first;
    p.out_mmpos = out_mmpos;
    p.out_hash = out_hash;
    p.out_size = out_size;
    p.capacity = capacity;
    std::memset(result, 0, 8 * sizeof(unsigned long long));
    run_mode<MODE_SUPERKMER>(p, result);
}

void emu_syncmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, unsigned s, unsigned soff, unsigned eoff, uint64_t seed,
                  unsigned flags, uint64_t* out_pos, uint64_t capacity, unsigned long long* result)
{
    ScanParams p{};
    fill_common(p, b->bases, b->n_bases, b->single ? nullptr : b->bits.data(), MODE_SYNCMER, first, n, s, k - s + 1, seed, flags);
    p.soff = (int)soff;
    p.eoff = (int)eoff;
    p.out_pos = out_pos;
    p.capacity = capacity;
    std::memset(result, 0, 8 * sizeof(unsigned long long));
    run_mode<MODE_SYNCMER>(p, result);
}

void emu_kmers(const EmuBatch* b, uint64_t first, uint64_t n, unsigned k, uint64_t seed, unsigned flags, uint64_t* out_value,
               uint64_t* out_hash, uint8_t* out_valid, unsigned long long* result)
{
    uint64_t end = n == 0 ? b->n_bases : first + n;
    if (end > b->n_bases) end = b->n_bases;
    KmerParams p{};
    p.bases = b->bases;
    p.n_bases = (int64_t)b->n_bases;
    p.start_bits = b->single ? nullptr : b->bits.data();
    p.first = (int64_t)first;
    p.end = (int64_t)end;
    p.origin = align_down16((int64_t)first);
    p.n_tiles = end > first ? (int32_t)(((int64_t)end - 1 - p.origin) / H + 1) : 0;
    p.unit = (int)k;
    p.seed = (uint32_t)seed;
    p.canonical = (flags & 1) ? 1 : 0;
    p.drop_last = (flags & 2) ? 1 : 0;
    p.out_value = out_value;
    p.out_hash = out_hash;
    p.out_valid = out_valid;
    ScanParams lp{};
    lp.bases = p.bases;
    lp.n_bases = p.n_bases;
    lp.start_bits = p.start_bits;
    KmerAcc acc{0, 0, 0, 0};
    std::vector<uint32_t> codes(NCHUNK_POS), flg(NCHUNK_POS);
    for (int tile = 0; tile < p.n_tiles; ++tile) {
        const int64_t q0 = p.origin + (int64_t)tile * H;
        for (int c = 0; c < NCHUNK_POS; ++c) stage_chunk(lp, codes.data(), flg.data(), c, q0);
        for (int tid = 0; tid < TPB; ++tid) kmer_thread(p, codes.data(), flg.data(), tid, q0, acc);
    }
    result[0] = acc.cnt;
    result[1] = acc.xv;
    result[2] = acc.xh;
    result[3] = acc.sh;
}

uint64_t emu_hash64(uint64_t v, uint64_t seed) { return murmur64(v, (uint32_t)seed); }

}  // extern "C"
