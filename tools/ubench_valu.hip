// tools/ubench_valu.hip — issue cost, in SHADER CYCLES, of the integer VALU instructions the scan is made of (gfx950).
//
// Every kernel runs 16 independent chains of ONE instruction, ITERS times.  Each wave stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around its loop, so the table needs no assumption about the clock:
//   cycles per instruction on a SIMD = median over waves of (delta s_memtime / instructions of the wave) / waves per SIMD
//   clock held by the chip under that loop = delta s_memtime / delta s_memrealtime x 100 MHz   (MI355X_MICROARCH.md, DVFS item 6)
// Grids of 256 x {1, 2, 4, 8} workgroups of 256 threads = 1, 2, 4, 8 waves per SIMD, every CU busy.
// Output: a table for people and, with --json FILE, the same numbers for tools/valu_model.py (the VALU ceiling of bench.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_valu.hip -o tools/_build/ubench_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../biolib_amd/csrc/bl_scan_core.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 8192;
constexpr int UNROLL = 16;

struct Stamp {
    unsigned long long cycles, ticks;
};

__device__ __forceinline__ void stamp_begin(unsigned long long& c, unsigned long long& t)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    c = __builtin_amdgcn_s_memtime();
    t = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void stamp_end(Stamp* out, unsigned long long c0, unsigned long long t0)
{
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};
}

#define KERNEL32(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(Stamp* out, uint32_t* sink, uint32_t seed)                   \
    {                                                                                                        \
        uint32_t r[UNROLL];                                                                                  \
        for (int i = 0; i < UNROLL; ++i) r[i] = seed + threadIdx.x * 31u + i;                                \
        uint32_t k = seed | 0x9e3779b1u;                                                                     \
        unsigned long long c0, t0;                                                                           \
        stamp_begin(c0, t0);                                                                                 \
        for (int it = 0; it < ITERS; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(k));      \
        }                                                                                                    \
        stamp_end(out, c0, t0);                                                                              \
        uint32_t acc = 0;                                                                                    \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                                        \
        if (acc == 0x12345678u) sink[0] = acc;                                                               \
    }

#define KERNEL64(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(Stamp* out, uint32_t* sink, uint32_t seed)                   \
    {                                                                                                        \
        uint64_t r[UNROLL];                                                                                  \
        for (int i = 0; i < UNROLL; ++i) r[i] = ((uint64_t)seed << 32) + threadIdx.x * 31u + i;              \
        uint32_t k = seed | 0x9e3779b1u;                                                                     \
        unsigned long long c0, t0;                                                                           \
        stamp_begin(c0, t0);                                                                                 \
        for (int it = 0; it < ITERS; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(k));      \
        }                                                                                                    \
        stamp_end(out, c0, t0);                                                                              \
        uint64_t acc = 0;                                                                                    \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                                        \
        if (acc == 0x12345678u) sink[0] = (uint32_t)acc;                                                     \
    }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_or_b32, "v_or_b32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %0")
KERNEL32(k_add_u32_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL32(k_xor_b32_e64, "v_xor_b32_e64 %0, %0, %1")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_min3, "v_min3_u32 %0, %0, %1, %1")
KERNEL32(k_bfrev, "v_bfrev_b32 %0, %0")
KERNEL32(k_not, "v_not_b32 %0, %0")
KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %1")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %1")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_ashrrev, "v_ashrrev_i32 %0, 3, %0")
KERNEL32(k_mov_dpp_wave_shl, "v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_mov_dpp_row_shr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_wave_shl, "v_add_u32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %0")

// compiler-generated compound operations (what the hash path is really made of)
#define KERNELC(NAME, EXPR)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(Stamp* out, uint32_t* sink, uint32_t seed)                   \
    {                                                                                                        \
        uint64_t r[UNROLL];                                                                                  \
        for (int i = 0; i < UNROLL; ++i) r[i] = ((uint64_t)seed << 32) + threadIdx.x * 31u + i;              \
        const uint64_t k = ((uint64_t)seed << 33) | 0x9e3779b1u;                                             \
        unsigned long long c0, t0;                                                                           \
        stamp_begin(c0, t0);                                                                                 \
        for (int it = 0; it < ITERS; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { uint64_t x = r[i]; x = (EXPR); asm volatile("" : "+v"(x)); r[i] = x; } \
        }                                                                                                    \
        stamp_end(out, c0, t0);                                                                              \
        uint64_t acc = 0;                                                                                    \
        for (int i = 0; i < UNROLL; ++i) acc ^= r[i];                                                        \
        if (acc == 0x12345678u) sink[0] = (uint32_t)acc;                                                     \
    }
KERNELC(k_add64, x + k)
KERNELC(k_mul64c, x * 0x87c37b91114253d5ULL)
KERNELC(k_mul64split, bl::mul64c(x, 0x87c37b91114253d5ULL))
KERNELC(k_xorshift33, x ^ (x >> 33))
KERNELC(k_rotl31, bl::rotl64_31(x))
KERNELC(k_fmix64, bl::fmix64(x))
KERNELC(k_murmur64, bl::murmur64(x, seed))
KERNELC(k_min64, x < k ? x : k)

struct K {
    const char* name;
    void (*fn)(Stamp*, uint32_t*, uint32_t);
    int compound;  // 1: a compiler-generated body (cycles are per body, not per instruction)
};

static double median(std::vector<double>& v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char** argv)
{
    const char* json_path = nullptr;
    for (int i = 1; i + 1 < argc; ++i)
        if (!std::strcmp(argv[i], "--json")) json_path = argv[i + 1];
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %d MHz\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    printf("cycles = shader cycles (s_memtime) a SIMD spends per wave64 instruction; clock = s_memtime / s_memrealtime x 100 MHz\n");
    Stamp* d_out;
    uint32_t* d_sink;
    const int max_waves = n_cu * 8 * 4;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * max_waves));
    CHECK(hipMalloc(&d_sink, 256));
    K ks[] = {
        {"v_add_u32", k_add_u32, 0}, {"v_sub_u32", k_sub_u32, 0}, {"v_xor_b32", k_xor_b32, 0}, {"v_and_b32", k_and_b32, 0}, {"v_or_b32", k_or_b32, 0},
        {"v_mov_b32", k_mov, 0}, {"v_lshlrev_b32", k_lshlrev_b32, 0}, {"v_lshrrev_b32", k_lshrrev_b32, 0}, {"v_min_u32", k_min_u32, 0},
        {"v_cndmask_b32", k_cndmask, 0}, {"v_cmp_lt_u32", k_cmp_lt_u32, 0}, {"v_cndmask_b32_e64 sgpr", k_cndmask_e64, 0}, {"v_cmp+v_cndmask (pair)", k_cmp_cndmask, 1}, {"v_add_f32", k_add_f32, 0}, {"v_fma_f32", k_fma_f32, 0},
        {"v_add_u32_e64", k_add_u32_e64, 0}, {"v_xor_b32_e64", k_xor_b32_e64, 0}, {"v_bitop3_b32", k_bitop3, 0},
        {"v_mul_lo_u32", k_mul_lo_u32, 0}, {"v_mul_hi_u32", k_mul_hi_u32, 0}, {"v_mul_u32_u24", k_mul_u32_u24, 0}, {"v_mad_u32_u24", k_mad_u32_u24, 0},
        {"v_alignbit_b32", k_alignbit, 0}, {"v_perm_b32", k_perm, 0}, {"v_bfe_u32", k_bfe, 0}, {"v_lshl_or_b32", k_lshl_or, 0},
        {"v_and_or_b32", k_and_or, 0}, {"v_xad_u32", k_xad, 0}, {"v_add3_u32", k_add3, 0}, {"v_lshl_add_u32", k_lshl_add, 0}, {"v_min3_u32", k_min3, 0},
        {"v_bfrev_b32", k_bfrev, 0}, {"v_not_b32", k_not, 0}, {"v_or3_b32", k_or3, 0}, {"v_bfi_b32", k_bfi, 0}, {"v_bcnt_u32_b32", k_bcnt, 0},
        {"v_ashrrev_i32", k_ashrrev, 0}, {"v_mov_b32_dpp wave_shl", k_mov_dpp_wave_shl, 0}, {"v_mov_b32_dpp row_shr", k_mov_dpp_row_shr, 0},
        {"v_add_u32_dpp wave_shl", k_add_dpp_wave_shl, 0},
        {"v_mad_u64_u32", k_mad_u64_u32, 0}, {"v_lshlrev_b64", k_lshlrev_b64, 0}, {"v_lshrrev_b64", k_lshrrev_b64, 0}, {"v_cmp_lt_u64", k_cmp_lt_u64, 0},
        {"v_lshl_add_u64", k_lshl_add_u64, 0}, {"v_mov_b64", k_mov_b64, 0},
        {"C: x + k (u64)", k_add64, 1}, {"C: x * const (u64)", k_mul64c, 1}, {"C: mul64c(x, const): mul_lo + 2 mad + mov", k_mul64split, 1}, {"C: x ^ (x >> 33)", k_xorshift33, 1}, {"C: rotl64(x, 31)", k_rotl31, 1},
        {"C: fmix64", k_fmix64, 1}, {"C: murmur64 (whole hash)", k_murmur64, 1}, {"C: min(x, k) (u64)", k_min64, 1},
    };
    const int wps_list[4] = {1, 2, 4, 8};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    FILE* js = json_path ? fopen(json_path, "w") : nullptr;
    if (js) fprintf(js, "{\"device\": \"%s\", \"n_cu\": %d, \"iters\": %d, \"unroll\": %d, \"ops\": {\n", prop.gcnArchName, n_cu, ITERS, UNROLL);
    printf("%-28s", "instruction");
    for (int w : wps_list) printf("  %dw: cyc  wall   GHz", w);
    printf("\n");
    bool first = true;
    for (auto& k : ks) {
        printf("%-28s", k.name);
        if (js) fprintf(js, "%s  \"%s\": {\"compound\": %d", first ? "" : ",\n", k.name, k.compound);
        first = false;
        for (int wps : wps_list) {
            const int blocks = n_cu * wps;  // 4 waves per workgroup = one per SIMD
            // the chip settles its clock under load over tens of milliseconds: run the kernel back to back for ~0.2 s first
            for (int rep = 0; rep < (k.compound ? 3 : 10); ++rep) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 1u);
            CHECK(hipDeviceSynchronize());
            float wall_ms = 0.f;
            for (int t = 0; t < 3; ++t) {  // the best of three launches: one launch in a few hundred is held up by something else on the box
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 1u);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                wall_ms = (t == 0 || ms < wall_ms) ? ms : wall_ms;
            }
            std::vector<Stamp> h((size_t)blocks * 4);
            CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc, ghz;
            for (const Stamp& s : h) {
                cyc.push_back((double)s.cycles / ((double)ITERS * UNROLL) / wps);
                if (s.ticks) ghz.push_back((double)s.cycles / (double)s.ticks * 0.1);
            }
            const double c = median(cyc), g = median(ghz);
            // cross-check from the host: kernel wall time x that clock / instructions per SIMD (includes launch, ramp and tail)
            const double wall_c = (double)wall_ms * 1e6 * g / ((double)ITERS * UNROLL * wps);
            printf("  %6.2f %6.2f %5.2f", c, wall_c, g);
            if (js) fprintf(js, ", \"w%d\": {\"cycles\": %.4f, \"wall_cycles\": %.4f, \"ghz\": %.4f}", wps, c, wall_c, g);
        }
        printf("\n");
        if (js) fprintf(js, "}");
    }
    if (js) {
        fprintf(js, "\n}}\n");
        fclose(js);
    }
    return 0;
}
