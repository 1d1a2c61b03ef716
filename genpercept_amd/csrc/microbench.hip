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

// r4 probe: the inner loop of the conv / GEMM kernels in isolation, and then with their other ingredients added one at a time.  Every wave runs
// "steps" of 32 independent-accumulator v_mfma_f32_16x16x32 (two 4 x 4 outer products of fragments) with NR conflict-free ds_read_b128 per 16
// MFMAs that refill the OTHER fragment set (software pipeline of depth one, 2 MFMAs : 1 read like conv3x3_halo3_kernel).
// NR = 0 is the bare MFMA rate; 8 is the convs' ratio (0.5 reads per MFMA); 16 reads every fragment twice.  WPS = waves per SIMD (1, 2 or 4).
// MODE bits add what the real kernel does around that loop, per step:
//   1  one raw s_barrier (after s_waitcnt lgkmcnt(0)), i.e. the workgroup's waves in lockstep
//   2  the weight stream: two 1-KiB LDS-DMA pieces per wave (16 KiB per workgroup of 8 waves) from an L2-resident 2.25 MiB buffer every CU walks in
//      the same order, into a 3-deep ring, with the counted s_waitcnt vmcnt(2) that certifies the tile issued one step earlier
//   4  the weight fragments are read from the ring slot certified at the end of the previous step (needs 1 | 2), not from a private region
//   8  the DMA as buffer_load ... lds (MUBUF) instead of global_load_lds
//   16 the halo stream: every ninth step six more pieces per wave (48 KiB per workgroup) from a per-workgroup window of a 256 MiB buffer
template <int NR, int WPS, int MODE>
__global__ __launch_bounds__(256 * WPS) void mfma_lds_probe_kernel(float* __restrict__ out, int iters, const h16_t* __restrict__ wbuf,
                                                                  const h16_t* __restrict__ hbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const volatile __attribute__((address_space(3))) h16x8_t* vfrag_ptr;
    constexpr int NWAVE = 4 * WPS;
    constexpr int XREG = 0, HREG = NWAVE * 8192, RING = HREG + ((MODE & 16) ? NWAVE * 6 * 1024 : 0), SLOT = NWAVE * 2048;
    constexpr unsigned WBYTES = 2304u * 1024u, HBYTES = 256u * 1024u * 1024u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // 8 KiB per wave: eight 1-KiB fragments, lane l at byte 16 l (conflict-free for ds_read_b128 whatever the lane grouping)
        h16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (short)f_to_h16(0.25f + 0.001f * (float)((lane * 8 + e) % 97));
        for (int f = 0; f < 8; ++f) *(h16x8_t*)(smem + XREG + wave * 8192 + f * 1024 + lane * 16) = v;
        if (MODE & 2)
            for (int f = 0; f < 6; ++f) *(h16x8_t*)(smem + RING + (f >> 1) * SLOT + (wave * 2 + (f & 1)) * 1024 + lane * 16) = v;
    }
    __syncthreads();
    const unsigned xbase = (unsigned)(unsigned long long)smem + (unsigned)(XREG + wave * 8192 + lane * 16);
    // weight fragments from the ring: the wave's channel half = pieces (wave & 1) * NWAVE .. + 3 of a slot (any four conflict-free KiB do)
    const unsigned rbase = (unsigned)(unsigned long long)smem + (unsigned)(RING + (wave & 1) * (SLOT / 2) + lane * 16);
    h16x8_t fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < 4; ++f) { fa[s][f] = *(vfrag_ptr)(xbase + f * 1024); fb[s][f] = *(vfrag_ptr)(xbase + (4 + f) * 1024); }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned woff = (unsigned)(wave * 2048 + lane * 16);  // byte offset of this lane's first piece inside the weight buffer
    const buf_rsrc_t wrs = make_rsrc(wbuf, WBYTES);
    auto half = [&](auto curc, unsigned wsrc_base) __attribute__((always_inline)) {  // 16 MFMAs on set CUR, NR reads into the other set
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (NR <= 8) {
                if (f < (NR + 1) / 2) fa[NXT][f] = *(vfrag_ptr)(wsrc_base + f * 1024);
                if (f < NR / 2) fb[NXT][f] = *(vfrag_ptr)(xbase + (4 + f) * 1024);
            } else {
                fa[NXT][f] = *(vfrag_ptr)(wsrc_base + f * 1024);
                fb[NXT][f] = *(vfrag_ptr)(xbase + (4 + f) * 1024);
                const h16x8_t t0 = *(vfrag_ptr)(wsrc_base + f * 1024), t1 = *(vfrag_ptr)(xbase + (4 + f) * 1024);
                asm volatile("" ::"v"(t0), "v"(t1));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32(fa[CUR][i], fb[CUR][j], acc[i][j]);
        if (NR > 0) {
            constexpr int PER = NR >= 16 ? 1 : NR >= 8 ? 2 : NR >= 4 ? 4 : 8;  // MFMAs per read
#pragma unroll
            for (int q = 0; q < 16 / PER; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto slotc, int nine) __attribute__((always_inline)) {  // slotc: ring slot being READ this step; the DMA fills slot + 2
        constexpr int S = decltype(slotc)::value;
        if (MODE & 2) {
            char* dst = smem + RING + ((S + 2) % 3) * SLOT + wave * 2048;
            if (MODE & 8) { blds16(wrs, woff, 0, dst); blds16(wrs, woff + 1024u, 0, dst + 1024); }
            else { glds16((const char*)wbuf + woff, dst); glds16((const char*)wbuf + woff + 1024u, dst + 1024); }
            woff += (unsigned)(NWAVE * 2048);
            if (woff >= WBYTES) woff -= WBYTES;
        }
        if ((MODE & 16) && nine == 0) {
            const unsigned hwin = (unsigned)(((blockIdx.x * 977u + (unsigned)(woff >> 11)) * (unsigned)(NWAVE * 6144)) & (HBYTES - 1u)) & ~(unsigned)(NWAVE * 6144 - 1);
#pragma unroll
            for (int q = 0; q < 6; ++q) glds16((const char*)hbuf + hwin + (unsigned)((wave * 6 + q) * 1024 + lane * 16), smem + HREG + (wave * 6 + q) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned wsrc = (MODE & 4) ? rbase + (unsigned)(((S + 1) % 3) * SLOT) : xbase;  // (prefetch: the NEXT step's fragments)
        half(IC<0>{}, (MODE & 4) ? rbase + (unsigned)(S * SLOT) : xbase);  // second half of this step's tile
        half(IC<1>{}, wsrc);
        if (MODE & 2) {
            if (MODE & 16) { if (nine == 0) wait_vm<8>(); else wait_vm<2>(); }  // (the halo burst may stay in flight for one step, like the real kernel's)
            else wait_vm<2>();
        }
        if (MODE & 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    int nine = 0;
    for (int i = 0; i < iters; ++i) {
        step(IC<0>{}, nine); nine = nine == 8 ? 0 : nine + 1;
        step(IC<1>{}, nine); nine = nine == 8 ? 0 : nine + 1;
        step(IC<2>{}, nine); nine = nine == 8 ? 0 : nine + 1;
    }
    if (MODE & 2) wait_vm<0>();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NR, int WPS, int MODE>
static double run_lds_probe(int ncu, hipStream_t s) {
    const int threads = 256 * WPS, lds = 128 * 1024;  // (128 KiB: one workgroup per CU)
    (void)hipFuncSetAttribute((const void*)mfma_lds_probe_kernel<NR, WPS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out = nullptr;
    h16_t *wbuf = nullptr, *hbuf = nullptr;
    if (hipMalloc((void**)&out, (size_t)ncu * threads * sizeof(float)) != hipSuccess) return -1.0;
    if (MODE & 2) {
        if (hipMalloc((void**)&wbuf, 2304u * 1024u + 4096u) != hipSuccess) return -1.0;
        (void)hipMemsetAsync(wbuf, 0x3c, 2304u * 1024u + 4096u, s);  // 0x3c3c: 0.0115 (bf16) / 1.06 (fp16) -- not all-zero bits
    }
    if (MODE & 16) {
        if (hipMalloc((void**)&hbuf, 256u * 1024u * 1024u + 65536u) != hipSuccess) return -1.0;
        (void)hipMemsetAsync(hbuf, 0x3c, 256u * 1024u * 1024u + 65536u, s);
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double flop_per_iter = (double)ncu * 4 * WPS * 96 /*mfma per three steps*/ * 2.0 * 16 * 16 * 32;
    int iters = 1500;
    double best = -1.0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL((mfma_lds_probe_kernel<NR, WPS, MODE>), dim3(ncu), dim3(threads), lds, s, out, iters, (const h16_t*)wbuf, (const h16_t*)hbuf);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms <= 0.f) break;
        const double tf = flop_per_iter * iters / (ms * 1e-3) / 1e12;
        if (rep > 0 && tf > best) best = tf;
        if (rep == 0) {
            iters = (int)(iters * 10.0 / ms);
            if (iters < 100) iters = 100;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (wbuf) (void)hipFree(wbuf);
    if (hbuf) (void)hipFree(hbuf);
    return best;
}
// TFLOP/s of the probe with `reads` ds_read_b128 per 16 MFMAs (0, 2, 4, 8, 16) at `wps` waves per SIMD (1, 2, 4) and the MODE bits above (modes
// other than 0 exist for 8 reads at two waves per SIMD: the conv kernels' operating point); < 0 on error / unsupported arguments
double mfma_lds_probe_tflops(int reads, int wps, int mode, hipStream_t s) {
    int dev = 0, ncu = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount;
#define GP_PROBE(R, W, M) if (reads == R && wps == W && mode == M) return run_lds_probe<R, W, M>(ncu, s);
    GP_PROBE(0, 1, 0) GP_PROBE(2, 1, 0) GP_PROBE(4, 1, 0) GP_PROBE(8, 1, 0) GP_PROBE(16, 1, 0)
    GP_PROBE(0, 2, 0) GP_PROBE(2, 2, 0) GP_PROBE(4, 2, 0) GP_PROBE(8, 2, 0) GP_PROBE(16, 2, 0)
    GP_PROBE(0, 4, 0) GP_PROBE(2, 4, 0) GP_PROBE(4, 4, 0) GP_PROBE(8, 4, 0) GP_PROBE(16, 4, 0)
    GP_PROBE(8, 2, 1) GP_PROBE(8, 2, 2) GP_PROBE(8, 2, 3) GP_PROBE(8, 2, 7) GP_PROBE(8, 2, 10) GP_PROBE(8, 2, 11) GP_PROBE(8, 2, 15)
    GP_PROBE(8, 2, 18) GP_PROBE(8, 2, 19) GP_PROBE(8, 2, 23) GP_PROBE(0, 2, 1) GP_PROBE(0, 2, 3) GP_PROBE(0, 2, 19)
#undef GP_PROBE
    return -1.0;
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
