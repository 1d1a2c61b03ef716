// probe (r4, for the next round's design decision): can ONE wave per SIMD keep the matrix pipe busy on a 128 x 128 register tile?
// Every wave runs iterations of 16 independent v_mfma_f32_32x32x16_bf16 (a 4 x 4 outer product of 32-row fragments = 128 x 128 outputs, 256
// accumulator registers) with NR conflict-free ds_read_b128 that refill the OTHER fragment set (4 + 4 fragments of one K = 16 slice = 32
// registers per set) -- the inner loop a 512-pixel x 128-channel-per-wave conv / GEMM tile would have: 0.5 reads per 32x32x16 MFMA =
// 0.25 per 16x16x32-equivalent, half of conv3x3_halo3_kernel's fragment traffic per flop.  Variants: reads per iteration 0 / 4 / 8 / 16, one wave
// per SIMD (256 accumulators in AGPRs + 64 fragment registers).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/wave1_tile_probe.cpp -o tools/probe/wave1_tile_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short h16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16_t mfma32(h16x8_t a, h16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <int V> struct IC { static constexpr int value = V; };

template <int NR, int WPS>
__global__ __launch_bounds__(256 * WPS) void probe(float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const volatile __attribute__((address_space(3))) h16x8_t* vfrag_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        h16x8_t v;
        for (int e = 0; e < 8; ++e) v[e] = (short)(0x3c00 + ((lane * 8 + e) % 97));  // bf16 bit patterns around 0.0078 .. 0.0117, not constant
        for (int f = 0; f < 8; ++f) *(h16x8_t*)(smem + wave * 8192 + f * 1024 + lane * 16) = v;
    }
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)smem + (unsigned)(wave * 8192 + lane * 16);
    h16x8_t fa[2][4], fb[2][4];
    for (int s = 0; s < 2; ++s)
        for (int f = 0; f < 4; ++f) { fa[s][f] = *(vfrag_ptr)(base + f * 1024); fb[s][f] = *(vfrag_ptr)(base + (4 + f) * 1024); }
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto step = [&](auto curc) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (NR >= 8 || f < NR / 2) fa[NXT][f] = *(vfrag_ptr)(base + f * 1024);
            if (NR >= 8 || f < NR / 2) fb[NXT][f] = *(vfrag_ptr)(base + (4 + f) * 1024);
            if (NR > 8) {
                const h16x8_t t0 = *(vfrag_ptr)(base + f * 1024), t1 = *(vfrag_ptr)(base + (4 + f) * 1024);
                asm volatile("" ::"v"(t0), "v"(t1));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(fa[CUR][i], fb[CUR][j], acc[i][j]);
        if (NR > 0) {
            constexpr int PER = NR >= 16 ? 1 : NR >= 8 ? 2 : 4;
#pragma unroll
            for (int q = 0; q < 16 / PER; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < iters; ++i) {
        step(IC<0>{});
        step(IC<1>{});
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NR, int WPS>
static void run(int ncu) {
    const int threads = 256 * WPS, lds = 128 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<NR, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out;
    hipMalloc(&out, (size_t)ncu * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 1000;
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<NR, WPS>), dim3(ncu), dim3(threads), lds, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)ncu * 4 * WPS * 32 * 2.0 * 32 * 32 * 16 * iters / (ms * 1e-3) / 1e12;
        if (rep && tf > best) best = tf;
        if (!rep) { iters = (int)(iters * 10.0 / ms); if (iters < 100) iters = 100; }
    }
    printf("{\"reads_per_16_mfma_32x32x16\": %d, \"waves_per_simd\": %d, \"tflops\": %.1f}\n", NR, WPS, best);
    hipFree(out);
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    run<0, 1>(ncu); run<4, 1>(ncu); run<8, 1>(ncu); run<16, 1>(ncu);  // (two waves per SIMD cannot hold 256 accumulators each)
    return 0;
}
