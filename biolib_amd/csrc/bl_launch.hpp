// bl_launch.hpp — host-callable launchers of the gfx950 kernels in bl_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "bl_scan_phases.hpp"

namespace bl {
hipError_t launch_scan_count(int mode, const ScanParams& p, GroupRange g, hipStream_t stream);
hipError_t launch_scan_emit(int mode, const ScanParams& p, GroupRange g, hipStream_t stream, uint32_t lds_per_wg = 0);
hipError_t launch_tile_scan(const ScanParams& p, GroupRange g, unsigned long long* block_tot, unsigned long long* carry, hipStream_t stream);
hipError_t launch_kmers(const KmerParams& p, int n_blocks, hipStream_t stream);
hipError_t launch_reduce_shards(const unsigned long long* shards, unsigned long long* result, uint32_t add_mask, const unsigned long long* redone, hipStream_t stream);
hipError_t launch_synth(uint8_t* bases, uint64_t first, uint64_t n, uint64_t seed, hipStream_t stream);
hipError_t launch_start_bits_fixed(uint32_t* bits, uint64_t n_words, uint64_t n_bases, uint64_t read_len, hipStream_t stream);
hipError_t launch_start_bits_offsets(uint32_t* bits, const uint64_t* offsets, uint64_t n_seqs, uint64_t n_bases, hipStream_t stream);
}  // namespace bl
