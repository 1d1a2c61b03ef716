// probe (r4, for the next round's design decision): can ONE wave per SIMD keep the matrix pipe busy on a 128 x 128 register tile?
// Every wave runs iterations of 16 independent v_mfma_f32_32x32x16_bf16 (a 4 x 4 outer product of 32-row fragments = 128 x 128 outputs, 256
// accumulator registers) with NR conflict-free ds_read_b128 that refill the OTHER fragment set (4 + 4 fragments of one K = 16 slice = 32
// registers per set) -- the inner loop a 512-pixel x 128-channel-per-wave conv / GEMM tile would have: 0.5 reads per 32x32x16 MFMA =
// 0.25 per 16x16x32-equivalent, half of conv3x3_halo3_kernel's fragment traffic per flop.  Variants: reads per iteration 0 / 4 / 8 / 16, one wave
// per SIMD (256 accumulators in AGPRs + 64 fragment registers).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/wave1_tile_probe.cpp -o tools/probe/wave1_tile_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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

// conv3x3_halo3_kernel's inner loop (the library's gp_mfma_lds_probe, mode 0): TWO waves per SIMD, per iteration 16 independent v_mfma_f32_16x16x32 on a
// 64 x 64 register tile (64 accumulators) and NR ds_read_b128 refilling the other fragment set (NR = 8: 0.5 reads per MFMA)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NR, int RND = 0>
__global__ __launch_bounds__(512) void probe64(float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const volatile __attribute__((address_space(3))) h16x8_t* vfrag_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        for (int f = 0; f < 8; ++f) {
            h16x8_t v;
            for (int e = 0; e < 8; ++e) {
                if (RND) {  // every fragment different: random sign and mantissa, exponent in [2^-4, 2^0) -- what activations / weights look like to the multipliers
                    unsigned x = (unsigned)(((wave * 8 + f) * 64 + lane) * 8 + e) * 2654435761u;
                    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
                    v[e] = (short)((x & 0x807f) | ((0x7b + ((x >> 16) & 3)) << 7));
                } else {
                    v[e] = (short)(0x3c00 + ((lane * 8 + e) % 97));
                }
            }
            *(h16x8_t*)(smem + wave * 8192 + f * 1024 + lane * 16) = v;
        }
    }
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)smem + (unsigned)(wave * 8192 + lane * 16);
    h16x8_t fa[2][4], fb[2][4];
    for (int s = 0; s < 2; ++s)
        for (int f = 0; f < 4; ++f) { fa[s][f] = *(vfrag_ptr)(base + f * 1024); fb[s][f] = *(vfrag_ptr)(base + (4 + f) * 1024); }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto curc) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < NR / 2) fa[NXT][f] = *(vfrag_ptr)(base + f * 1024);
            if (f < NR / 2) fb[NXT][f] = *(vfrag_ptr)(base + (4 + f) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[CUR][i]), __builtin_bit_cast(bf16x8_t, fb[CUR][j]), acc[i][j], 0, 0, 0);
        if (NR > 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
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
        for (int j = 0; j < 4; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// sustained mode: launch the kernel back to back for `seconds`, print TFLOP/s per ~0.5 s window with wall-clock stamps (to line up with rocm-smi samples)
#include <chrono>
template <typename L>
static void sustain(const char* name, double flop_per_iter, int seconds, L launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 1000;
    { hipEventRecord(e0, 0); launch(iters); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); iters = (int)(iters * 100.0 / ms); }
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 5; ++k) launch(iters);   // ~0.5 s
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        printf("{\"config\": \"%s\", \"unix_time\": %.2f, \"elapsed_s\": %.2f, \"tflops\": %.1f}\n", name, now, el, flop_per_iter * iters * 5 / (ms * 1e-3) / 1e12);
        fflush(stdout);
        if (el > seconds) break;
    }
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
int main(int argc, char** argv) {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    if (argc > 1) {  // sustained runs: wave1_tile_probe <seconds>
        const int sec = atoi(argv[1]);
        float* out;
        hipMalloc(&out, (size_t)ncu * 512 * 4);
        const int lds = 128 * 1024;
        (void)hipFuncSetAttribute((const void*)probe<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)probe64<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)probe64<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        sustain("tile64x64_2waves_reads8", (double)ncu * 8 * 32 * 2.0 * 16 * 16 * 32, sec, [&](int it) { hipLaunchKernelGGL((probe64<8>), dim3(ncu), dim3(512), lds, 0, out, it); });
        sustain("tile128x128_1wave_reads8", (double)ncu * 4 * 32 * 2.0 * 32 * 32 * 16, sec, [&](int it) { hipLaunchKernelGGL((probe<8, 1>), dim3(ncu), dim3(256), lds, 0, out, it); });
        sustain("tile64x64_2waves_mfma_only", (double)ncu * 8 * 32 * 2.0 * 16 * 16 * 32, sec, [&](int it) { hipLaunchKernelGGL((probe64<0>), dim3(ncu), dim3(512), lds, 0, out, it); });
        // the same two loops with every fragment holding different random bits (the runs above feed identical fragments: no operand toggling at all)
        (void)hipFuncSetAttribute((const void*)probe64<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)probe64<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        sustain("random_bits_mfma_only", (double)ncu * 8 * 32 * 2.0 * 16 * 16 * 32, sec, [&](int it) { hipLaunchKernelGGL((probe64<0, 1>), dim3(ncu), dim3(512), lds, 0, out, it); });
        sustain("random_bits_reads8", (double)ncu * 8 * 32 * 2.0 * 16 * 16 * 32, sec, [&](int it) { hipLaunchKernelGGL((probe64<8, 1>), dim3(ncu), dim3(512), lds, 0, out, it); });
        return 0;
    }
    run<0, 1>(ncu); run<4, 1>(ncu); run<8, 1>(ncu); run<16, 1>(ncu);  // (two waves per SIMD cannot hold 256 accumulators each)
    return 0;
}
