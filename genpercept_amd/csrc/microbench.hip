// MFMA peak micro-benchmark: what the matrix cores of THIS chip sustain (clock under load included), printed by bench.py next to the
// nominal 2.5 PFLOP/s the roofline fractions are quoted against.  Register-resident operands, no memory
// traffic: every wave issues independent v_mfma_f32_32x32x16 (four accumulators) back to back.
#include "common.h"
#include "kernels.h"

// the same with v_mfma_f32_16x16x32 (the shape of the conv / GEMM kernels: K = 32 per instruction, a quarter of the accumulator registers
// per flop): sixteen independent accumulators of four registers
__global__ __launch_bounds__(256) void mfma_peak16_kernel(float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    h16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)f_to_h16(0.25f + 0.001f * (float)((lane * 8 + e) % 97));
        b[e] = (short)f_to_h16(-0.5f + 0.002f * (float)((lane * 5 + e * 3) % 89));
    }
    f32x4_t acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = mfma_16x16x32(a, b, acc[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_peak_kernel(float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    // non-trivial operand bits: all-zero operands clock (and therefore measure) ~20 % higher than real data (MI355X guide, DVFS)
    h16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)f_to_h16(0.25f + 0.001f * (float)((lane * 8 + e) % 97));
        b[e] = (short)f_to_h16(-0.5f + 0.002f * (float)((lane * 5 + e * 3) % 89));
    }
    f32x16_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma_32x32x16(a, b, acc[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

// returns measured TFLOP/s (< 0 on error); ~ms_target milliseconds of MFMA work on every CU, two waves per SIMD
double mfma_peak_tflops(int ms_target, hipStream_t s, int shape) {  // shape 0: 32x32x16, 1: 16x16x32
    int dev = 0, ncu = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount;
    const int blocks = ncu * 2;
    float* out = nullptr;
    if (hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double flop_per_iter = shape == 1 ? (double)blocks * 4 /*waves*/ * 64 /*mfma*/ * 2.0 * 16 * 16 * 32
                                            : (double)blocks * 4 /*waves*/ * 16 /*mfma*/ * 2.0 * 32 * 32 * 16;
    int iters = 2000;
    double best = -1.0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s);
        if (shape == 1) hipLaunchKernelGGL(mfma_peak16_kernel, dim3(blocks), dim3(256), 0, s, out, iters);
        else hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, out, iters);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms <= 0.f) break;
        const double tf = flop_per_iter * iters / (ms * 1e-3) / 1e12;
        if (rep > 0 && tf > best) best = tf;   // first launch: warm-up and sizing
        if (rep == 0) {
            iters = (int)(iters * (double)ms_target / ms);
            if (iters < 100) iters = 100;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return best;
}
