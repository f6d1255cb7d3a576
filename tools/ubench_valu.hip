// tools/ubench_valu.hip — measured issue rates of the integer VALU instructions the hash path is
// made of, on gfx950.  Each kernel runs 16 independent chains of ONE instruction at full occupancy
// (8 waves per SIMD); the table gives time per wave-instruction relative to v_add_u32 and the
// absolute rate per SIMD.  Used to derive the "valu ceiling" quoted in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu && gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

// OPS: one asm body applied to a register r (32-bit) or pair
#define KERNEL32(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)            \
    {                                                                                    \
        uint32_t r[UNROLL];                                                              \
        for (int i = 0; i < UNROLL; ++i) r[i] = seed + threadIdx.x * 31u + i;            \
        uint32_t k = seed | 0x9e3779b1u;                                                 \
        for (int it = 0; it < ITERS; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(k)); \
        }                                                                                \
        uint32_t acc = 0;                                                                \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                    \
        if (acc == 0x12345678u) out[0] = acc;                                            \
    }

#define KERNEL64(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)            \
    {                                                                                    \
        uint64_t r[UNROLL];                                                              \
        for (int i = 0; i < UNROLL; ++i) r[i] = ((uint64_t)seed << 32) + threadIdx.x * 31u + i; \
        uint32_t k = seed | 0x9e3779b1u;                                                 \
        for (int it = 0; it < ITERS; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(k)); \
        }                                                                                \
        uint64_t acc = 0;                                                                \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                    \
        if (acc == 0x12345678u) out[0] = (uint32_t)acc;                                  \
    }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %0")
KERNEL32(k_fmac_f32, "v_fmac_f32 %0, %1, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_add_u32_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL32(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_bfrev, "v_bfrev_b32 %0, %0")
KERNEL32(k_mov_dpp_wave_shl, "v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_mov_dpp_row_shl, "v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_lshlrev_b64_var, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %0")

// compiler-generated compound operations (what the hash is really made of)
#include "../biolib_amd/csrc/bl_scan_core.hpp"
#define KERNELC(NAME, EXPR)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)            \
    {                                                                                    \
        uint64_t r[UNROLL];                                                              \
        for (int i = 0; i < UNROLL; ++i) r[i] = ((uint64_t)seed << 32) + threadIdx.x * 31u + i; \
        const uint64_t k = ((uint64_t)seed << 33) | 0x9e3779b1u;                         \
        for (int it = 0; it < ITERS; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { uint64_t x = r[i]; x = (EXPR); asm volatile("" : "+v"(x)); r[i] = x; } \
        }                                                                                \
        uint64_t acc = 0;                                                                \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                    \
        if (acc == 0x12345678u) out[0] = (uint32_t)acc;                                  \
    }
KERNELC(k_add64, x + k)
KERNELC(k_mul64c, x * 0x87c37b91114253d5ULL)
KERNELC(k_xorshift33, x ^ (x >> 33))
KERNELC(k_rotl31, (x << 31) | (x >> 33))
KERNELC(k_fmix64, bl::fmix64(x))
KERNELC(k_murmur64, bl::murmur64(x, seed))
KERNELC(k_min64, x < k ? x : k)

struct K { const char* name; void (*fn)(uint32_t*, uint32_t); int instrs_per_body; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    uint32_t* d;
    CHECK(hipMalloc(&d, 256));
    K ks[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_fmac_f32, 1},
        {"v_add_f32", k_add_f32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_min_u32", k_min_u32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_mov_b32", k_mov, 1},
        {"v_add_u32_e64", k_add_u32_e64, 1}, {"v_cmp_lt_u32", k_cmp_lt_u32, 1}, {"v_bitop3_b32 (xor3)", k_bitop3, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
        {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
        {"v_mad_u32_u16", k_mad_u32_u16, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_perm_b32", k_perm, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_xad_u32", k_xad, 1}, {"v_add3_u32", k_add3, 1},
        {"v_lshl_add_u32", k_lshl_add, 1}, {"v_bfrev_b32", k_bfrev, 1}, {"v_mov_dpp wave_shl", k_mov_dpp_wave_shl, 1},
        {"v_mov_dpp row_shl", k_mov_dpp_row_shl, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshlrev_b64 imm", k_lshlrev_b64, 1}, {"v_lshrrev_b64 imm", k_lshrrev_b64, 1},
        {"v_lshlrev_b64 var", k_lshlrev_b64_var, 1}, {"v_cmp_lt_u64", k_cmp_lt_u64, 1},
        {"C: x + k (64b add)", k_add64, 1}, {"C: x * const (64b mul)", k_mul64c, 1}, {"C: x ^ (x>>33)", k_xorshift33, 1},
        {"C: rotl64(x,31)", k_rotl31, 1}, {"C: fmix64", k_fmix64, 1}, {"C: murmur64 (whole hash)", k_murmur64, 1},
        {"C: min64(x,k)", k_min64, 1},
    };
    const int blocks = prop.multiProcessorCount * 8;  // 8 blocks x 4 waves = 32 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double base = 0;
    for (auto& k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);  // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_instrs = (double)blocks * 4 * ITERS * UNROLL;          // bodies executed chip-wide (per wave)
        const double per_simd = wave_instrs / (prop.multiProcessorCount * 4);   // bodies per SIMD
        const double ns_per_body = best * 1e6 / per_simd;
        if (base == 0) base = ns_per_body;
        printf("%-36s %8.3f ms  %7.3f ns/body/SIMD  x%.2f vs v_add_u32  (%.2f cycles @2.4GHz per instr)\n", k.name, best, ns_per_body,
               ns_per_body / base, ns_per_body * 2.4 / k.instrs_per_body);
    }
    return 0;
}
