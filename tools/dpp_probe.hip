// probe: which lane does v_mov_b32_dpp wave_shl:1 / wave_shr:1 / row_shl:1 read from on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    int lane = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, false);  // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xf, 0xf, false);  // row_shl:1
    int d = __builtin_amdgcn_update_dpp(-1, lane, 0x134, 0xf, 0xf, false);  // wave_rol:1
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c; out[192 + lane] = d;
}
int main()
{
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"wave_shl:1", "wave_shr:1", "row_shl:1", "wave_rol:1"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; ++i) printf(" %d", h[64 * t + i]); printf("\n"); }
    return 0;
}
