// bl_kernels.hip — gfx950 kernels of the fused k-mer / minimizer scan and their launchers.
// Written for CDNA4 only (wave64, 256-thread workgroups, LDS-staged tiles); no portability layer.
#include <hip/hip_runtime.h>
#include "bl_scan_phases.hpp"
#include "bl_launch.hpp"

namespace bl {

// ------------------------------------------------------------------------------------------------
// wave / workgroup primitives

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
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

// exclusive scan over the workgroup of a value that packs two 16-bit counters (sums stay < 65536)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_tot, int tid, uint32_t& total)
{
    const int lane = tid & 63, wv = tid >> 6;
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

// ------------------------------------------------------------------------------------------------
// Ordered compaction across tiles: decoupled look-back over one 8-byte status word per tile.
// The word carries flag + both counters, written by ONE agent-scope store and polled with
// agent-scope relaxed loads (sc1, L1-bypassing): the {data, tag} granule of MI355X_MICROARCH.md
// "R2" — no separate payload, hence no release/acquire pair is needed.
// Tile ids come from an atomic ticket, so every predecessor of a running tile is itself running
// or finished: the waits below cannot deadlock whatever the dispatch order.  Every spin is bounded.
constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, FLAG_MASK = 3ull << 62;
constexpr unsigned long long CNT_MASK = (1ull << 31) - 1;

__device__ __forceinline__ unsigned long long pack_status(unsigned long long flag, uint32_t s, uint32_t e)
{
    return flag | ((unsigned long long)e << 31) | s;
}

// called by the whole first wave; returns the exclusive prefix (starts, ends) of tile `tile`.
// One hop inspects LB_DEPTH x 64 predecessors (LB_DEPTH independent loads per lane in flight), so the
// number of tiles whose prefix is still unknown when a tile arrives — tile rate x visibility latency —
// is covered in one or two hops even at several hundred tiles per microsecond.
constexpr int LB_DEPTH = 4;

__device__ __forceinline__ void lookback(const ScanParams& p, uint32_t tile, uint32_t agg_s, uint32_t agg_e, int lane,
                                         uint32_t& excl_s, uint32_t& excl_e)
{
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&p.status[0], pack_status(FLAG_INC, agg_s, agg_e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        excl_s = excl_e = 0;
        return;
    }
    if (lane == 0) __hip_atomic_store(&p.status[tile], pack_status(FLAG_AGG, agg_s, agg_e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long run_s = 0, run_e = 0;
    int64_t pos = (int64_t)tile - 1;
    bool done = false;
    for (int hop = 0; hop < (1 << 20) && !done; ++hop) {
        unsigned long long w[LB_DEPTH];
#pragma unroll
        for (int j = 0; j < LB_DEPTH; ++j) {  // issue all loads first
            const int64_t idx = pos - lane - 64 * j;
            w[j] = idx >= 0 ? __hip_atomic_load(&p.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLAG_INC;  // before tile 0: prefix 0
        }
#pragma unroll
        for (int j = 0; j < LB_DEPTH; ++j) {  // nearest 64 predecessors first
            if (done) break;
            const int64_t idx = pos - lane - 64 * j;
            unsigned spins = 0;
            while ((w[j] & FLAG_MASK) == 0) {
                if (++spins > (1u << 22)) { atomicOr(p.error, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
                w[j] = __hip_atomic_load(&p.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long inc_mask = __ballot((w[j] & FLAG_MASK) == FLAG_INC);
            // lanes closer than the first inclusive prefix contribute their aggregate, that lane its prefix
            const int first_inc = inc_mask ? __builtin_ctzll(inc_mask) : 64;
            const bool use = lane <= first_inc && (w[j] & FLAG_MASK) != 0;
            run_s += wave_sum_u64(use ? (w[j] & CNT_MASK) : 0);
            run_e += wave_sum_u64(use ? ((w[j] >> 31) & CNT_MASK) : 0);
            if (inc_mask) done = true;
            if (__ballot((w[j] & FLAG_MASK) == 0)) done = true;  // timed out: error flag is set, leave
        }
        pos -= 64 * LB_DEPTH;
    }
    excl_s = (uint32_t)run_s;
    excl_e = (uint32_t)run_e;
    if (lane == 0)
        __hip_atomic_store(&p.status[tile], pack_status(FLAG_INC, excl_s + agg_s, excl_e + agg_e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// The fused scan kernel: one workgroup per tile.
template <int MODE, int W>
__global__ __launch_bounds__(TPB, (MODE == MODE_SYNCMER ? 2 : (W != 0 && W <= 11 ? 6 : 4))) void scan_kernel(const ScanParams p)
{
    __shared__ TileShared<MODE, W> sh;
    const int tid = threadIdx.x;
    if (tid == 0) sh.tile = (p.ablate & 1) ? blockIdx.x : atomicAdd(p.ticket, 1u);
    if (tid < 4) sh.dig[tid] = 0;
    __syncthreads();
    const uint32_t tile = sh.tile;
    const int64_t q0 = p.origin + (int64_t)tile * p.stride;

    phase_load<MODE, W>(p, sh, tid, q0);
    __syncthreads();

    ThreadState st;
    phase_hash<MODE, W>(p, sh, tid, st);
    if (W == 0) __syncthreads();  // runtime-w fallback exchanges hashes through LDS
    if (p.ablate & 32) return;

    uint32_t packed;
    if (MODE == MODE_SYNCMER) {
        uint8_t af[S + 1];
        phase_sync_fwd<MODE, W>(p, sh, tid, st, nullptr, af);
        if (W == 0 && p.canonical) {
            __syncthreads();
            phase_publish_h2<MODE, W>(sh, tid, st);
            __syncthreads();
        }
        packed = phase_sync_rev<MODE, W>(p, sh, tid, q0, st, nullptr, af);
    } else {
        packed = phase_window<MODE, W>(p, sh, tid, q0, st, nullptr);
    }

    if (p.ablate & 16) return;
    uint32_t total;
    const uint32_t excl = block_excl_scan(packed, sh.wave_tot, tid, total);
    const uint32_t n_s = total & 0xffffu, n_e = total >> 16;
    // note: a tile owns fewer than H positions, so the counters fit 16 bits

    phase_list<MODE, W>(sh, tid, st, excl & 0xffffu, excl >> 16);
    __syncthreads();  // lists complete

    // wave 0 resolves the tile's global offset while the other waves already rebuild their records
    if (tid < 64) {
        uint32_t bs, be;
        if (p.ablate & 2) { bs = 0; be = 0; }
        else lookback(p, tile, n_s, n_e, tid, bs, be);
        if (tid == 0) { sh.base_s = bs; sh.base_e = be; }
    }
    Digest dg{0, 0, 0};
    constexpr int PRE = 3;  // records prepared in registers per thread before the offset is known
    Record rec[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const uint32_t r = tid + k * TPB;
        if (r < n_s && !(p.ablate & 4)) rec[k] = emit_prepare<MODE, W>(p, sh, q0, r, dg);
    }
    __syncthreads();  // offset known
    const uint64_t base_s = sh.base_s, base_e = sh.base_e;
    if (!(p.ablate & 4)) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const uint32_t r = tid + k * TPB;
            if (r < n_s) emit_store<MODE>(p, rec[k], base_s + r);
        }
        for (uint32_t r = tid + PRE * TPB; r < n_s; r += TPB) {  // denser than 3 records per thread: rare
            const Record x = emit_prepare<MODE, W>(p, sh, q0, r, dg);
            emit_store<MODE>(p, x, base_s + r);
        }
        emit_ends<MODE, W>(p, sh, tid, q0, n_e, base_e);
    }
    if (p.ablate & 8) return;

    // digest: wave reduce -> LDS -> one set of atomics per tile into a shard line
    const unsigned long long xv = wave_xor_u64(dg.xv), xh = wave_xor_u64(dg.xh), xp = wave_xor_u64(dg.xp);
    if ((tid & 63) == 0) {
        atomicXor(&sh.dig[1], xv);
        atomicXor(&sh.dig[2], xh);
        atomicXor(&sh.dig[3], xp);
    }
    __syncthreads();
    if (tid < 4) {
        unsigned long long* shard = p.shards + 8 * (tile % NSHARD);
        if (tid == 0) atomicAdd(&shard[0], (unsigned long long)n_s);
        else atomicXor(&shard[tid], sh.dig[tid]);
    }
    if (MODE == MODE_SUPERKMER && tid == 4) atomicAdd(&p.shards[8 * (tile % NSHARD) + 4], (unsigned long long)n_e);
}

// ------------------------------------------------------------------------------------------------
// Dense k-mer scan kernel (C2): no windows, no compaction.
struct KmerShared {
    uint32_t codes[NCHUNK];
    uint32_t flags[NCHUNK];
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
        if (tid < NCHUNK - TPB) stage_chunk(lp, sh.codes, sh.flags, TPB + tid, q0);
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

// fold the NSHARD digest lines into result[0..7]; slots listed in add_mask are sums, the others XORs;
// result[8] = the protocol error word
__global__ void reduce_shards_kernel(const unsigned long long* shards, unsigned long long* result, uint32_t add_mask,
                                     const unsigned int* error)
{
    const int slot = threadIdx.x;
    if (slot == 8) result[8] = *error;
    if (slot >= 8) return;
    unsigned long long acc = 0;
    for (int i = 0; i < NSHARD; ++i) {
        const unsigned long long v = shards[8 * i + slot];
        if ((add_mask >> slot) & 1) acc += v;
        else acc ^= v;
    }
    result[slot] = acc;
}

// super-k-mer size = last k-mer - first k-mer + 1 (super_kmer_view.hpp:133); n = min(*count, capacity)
__global__ void superkmer_size_kernel(const uint64_t* first, const uint64_t* last, uint8_t* size,
                                      const unsigned long long* count, uint64_t capacity)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = *count < capacity ? *count : capacity;
    if (i < n) size[i] = (uint8_t)(last[i] - first[i] + 1);
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

template <int MODE>
static hipError_t launch_mode(const ScanParams& p, hipStream_t stream)
{
    const dim3 grid(p.n_tiles), block(TPB);
    switch (p.w) {
        case 11: hipLaunchKernelGGL((scan_kernel<MODE, 11>), grid, block, 0, stream, p); break;
        case 17: hipLaunchKernelGGL((scan_kernel<MODE, 17>), grid, block, 0, stream, p); break;
        case 21: hipLaunchKernelGGL((scan_kernel<MODE, 21>), grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL((scan_kernel<MODE, 0>), grid, block, 0, stream, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_scan(int mode, const ScanParams& p, hipStream_t stream)
{
    if (p.n_tiles <= 0) return hipSuccess;
    switch (mode) {
        case MODE_MINIMIZER: return launch_mode<MODE_MINIMIZER>(p, stream);
        case MODE_SUPERKMER: return launch_mode<MODE_SUPERKMER>(p, stream);
        case MODE_SYNCMER: return launch_mode<MODE_SYNCMER>(p, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_kmers(const KmerParams& p, int n_blocks, hipStream_t stream)
{
    if (p.n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kmer_kernel, dim3(n_blocks), dim3(TPB), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_reduce_shards(const unsigned long long* shards, unsigned long long* result, uint32_t add_mask,
                                const unsigned int* error, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_shards_kernel, dim3(1), dim3(64), 0, stream, shards, result, add_mask, error);
    return hipGetLastError();
}

hipError_t launch_superkmer_size(const uint64_t* first, const uint64_t* last, uint8_t* size, const unsigned long long* count,
                                 uint64_t capacity, hipStream_t stream)
{
    if (capacity == 0) return hipSuccess;
    hipLaunchKernelGGL(superkmer_size_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, stream, first, last, size, count,
                       capacity);
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
