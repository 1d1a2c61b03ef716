// probe: semantics of v_permlane32_swap / v_permlane16_swap / DPP row ops on gfx950 (results checked on the host)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
template <int CTRL, int RM>
__device__ __forceinline__ int dpp(int x) { return __builtin_amdgcn_update_dpp(-1, x, CTRL, RM, 0xf, true); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    swap32(a, b);
    out[l] = a; out[64 + l] = b;
    a = l; b = 100 + l;
    swap16(a, b);
    out[128 + l] = a; out[192 + l] = b;
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
    out[256 + l] = r[0]; out[320 + l] = r[1];
    out[384 + l] = dpp<0x108, 0xf>(l);   // row_shl:8
    out[448 + l] = dpp<0x104, 0xf>(l);   // row_shl:4
    out[512 + l] = dpp<0x128, 0xf>(l);   // row_ror:8
    out[576 + l] = dpp<0x118, 0xf>(l);   // row_shr:8
}
int main() {
    int* d; hipMalloc(&d, 640 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"swap32 vdst", "swap32 src", "swap16 vdst", "swap16 src", "builtin32 r0", "builtin32 r1", "row_shl8", "row_shl4", "row_ror8", "row_shr8"};
    for (int t = 0; t < 10; ++t) { printf("%-13s:", names[t]); for (int l = 0; l < 64; ++l) printf(" %d", h[t * 64 + l]); printf("\n"); }
    return 0;
}
