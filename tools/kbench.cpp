// kbench: torch-free micro-benchmark / checker of the per-kernel C-ABI entry points (include/genpercept_hip.h), bf16 library.
// A fresh gpurun box spends 1-2 minutes on its first `import torch`; this binary starts in a second, so kernel iterations cost
// GPU-seconds instead of GPU-minutes.  Build: tools/build_kbench.sh (hipcc host program linked against the in-tree .so).
//
//   kbench [iters=N] [cold=0|1] [check=0|1] spec...
//     gemm:M,N,K[,act[,res[,stats(unused)[,hint...]]]]   act: 0 none 3 geglu; res: 0/1; hints: tile_hint list (default 0)
//     conv:B,H,W,Cin,Cout[,ups[,hint...]]                 3x3 stride 1 pad 1
//     convg:B,H,W,Cin,Cout[,silu[,ups]]                   conv3x3 of the GroupNorm(+SiLU)-normalised input, apply fused into the conv
//     attn:B,T,heads                                      flash_attn64
//     qkv:M,C[,hint]                                      fused q|k|v^T projection (gp_gemm_qkv), when the library exports it
//   cold=1: the weight operand rotates through a 1 GiB arena (every launch streams its weights from HBM like the pipeline does),
//           the activation operand stays (it was just written by the previous kernel in the pipeline: MALL-warm).
// Output: one line per (spec, hint): us per launch (hot / cold), TFLOP/s, max|err| / max|ref| against a naive fp32 device reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/genpercept_hip.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
    } while (0)

typedef unsigned short h16;
static inline h16 f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (h16)(u >> 16);
}
__device__ __host__ static inline float bf2f(h16 h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

__global__ void fill_kernel(h16* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        const float f = ((float)(x & 0xffffff) / 8388608.f - 1.f) * scale;
        uint32_t u = __float_as_uint(f);
        u += 0x7fffu + ((u >> 16) & 1u);
        p[i] = (h16)(u >> 16);
    }
}
__global__ void fill_f32_kernel(float* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((float)(x & 0xffffff) / 8388608.f - 1.f) * scale;
    }
}
// naive reference: out[m][n] = sum_k a[m][k] w[n][k] + bias[n] (+ res[m][n]); one thread per output
__global__ void ref_gemm_kernel(const h16* a, int lda, const h16* w, int ldw, const float* bias, const h16* res, int ldres, float* out, int M, int N,
                                int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += bf2f(a[(size_t)m * lda + k]) * bf2f(w[(size_t)n * ldw + k]);
    if (bias) acc += bias[n];
    if (res) acc += bf2f(res[(size_t)m * ldres + n]);
    out[(size_t)m * N + n] = acc;
}
// max |out - ref| and max |ref| (out: bf16 [M][ldo], optional transposed layout outT[n][ldt] for the V^T check)
__global__ void cmp_kernel(const h16* out, int ldo, const float* ref, int M, int N, int transposed, float* res2) {
    float e = 0.f, r = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * N; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float o = transposed ? bf2f(out[(size_t)n * ldo + m]) : bf2f(out[(size_t)m * ldo + n]);
        const float f = ref[i];
        e = fmaxf(e, fabsf(o - f));
        r = fmaxf(r, fabsf(f));
    }
    atomicMax((unsigned*)&res2[0], __float_as_uint(e));
    atomicMax((unsigned*)&res2[1], __float_as_uint(r));
}

// GEGLU check: ref holds the pre-activation [M][N] in PACKED row order (value j at 32 (j / 16) + 8 ((j % 16) / 4) + j % 4, its gate 4 rows on);
// out [M][N/2] bf16 must be value * gelu_erf(gate)
__global__ void cmp_geglu_kernel(const h16* out, int ldo, const float* ref, int M, int N, float* res2) {
    const int nh = N / 2;
    float e = 0.f, r = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * nh; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nh), c = (int)(i % nh);
        const int vr = (c / 16) * 32 + ((c % 16) / 4) * 8 + (c % 4);
        const float val = ref[(size_t)m * N + vr], g = ref[(size_t)m * N + vr + 4];
        const float f = val * 0.5f * g * (1.f + erff(g * 0.70710678f));
        const float o = bf2f(out[(size_t)m * ldo + c]);
        e = fmaxf(e, fabsf(o - f));
        r = fmaxf(r, fabsf(f));
    }
    atomicMax((unsigned*)&res2[0], __float_as_uint(e));
    atomicMax((unsigned*)&res2[1], __float_as_uint(r));
}
__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static void fill(h16* p, size_t n, uint32_t seed, float scale = 1.f) { hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, seed, scale); }
static void fillf(float* p, size_t n, uint32_t seed, float scale = 1.f) { hipLaunchKernelGGL(fill_f32_kernel, dim3(256), dim3(256), 0, 0, p, n, seed, scale); }

static int g_iters = 20, g_cold = 1, g_check = 1;
static h16* g_arena = nullptr;  // 1 GiB of random weights
static const size_t ARENA = (size_t)1 << 30;

template <typename F>
static float time_us(F&& launch, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms * 1e3f / iters;
}

static std::vector<long long> parse_nums(const char* s) {
    std::vector<long long> v;
    while (*s) {
        char* e;
        v.push_back(strtoll(s, &e, 10));
        s = (*e == ',') ? e + 1 : e;
        if (e == s && *e) break;
    }
    return v;
}

static void bench_gemm(const std::vector<long long>& a) {
    const int M = (int)a[0], N = (int)a[1], K = (int)a[2];
    const int act = a.size() > 3 ? (int)a[3] : 0, use_res = a.size() > 4 ? (int)a[4] : 0;
    std::vector<int> hints;
    for (size_t i = 6; i < a.size(); ++i) hints.push_back((int)a[i]);
    if (hints.empty()) hints.push_back(0);
    const int nout = act == 3 ? N / 2 : N;
    const int nrows = gp_packed_rows(N);
    h16 *A, *O, *R = nullptr;
    float *bias, *ref = nullptr, *res2;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&O, (size_t)M * nout * 2));
    CK(hipMalloc(&bias, (size_t)nrows * 4));
    CK(hipMalloc(&res2, 8));
    fill(A, (size_t)M * K, 11);
    fillf(bias, nrows, 5, 0.5f);
    if (use_res) { CK(hipMalloc(&R, (size_t)M * nout * 2)); fill(R, (size_t)M * nout, 13); }
    const size_t wel = (size_t)nrows * K;
    const double flops = 2.0 * M * (double)N * K;
    const float wscale = 1.f / sqrtf((float)K) * 1.7f;
    (void)wscale;
    if (g_check) {
        CK(hipMalloc(&ref, (size_t)M * N * 4));
        hipLaunchKernelGGL(ref_gemm_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, A, K, g_arena, K, bias, R, nout, ref, M, N, K);
    }
    for (int hint : hints) {
        CK(hipMemset(O, 0xff, (size_t)M * nout * 2));
        auto go = [&](const h16* w) {
            gp_status st = gp_gemm(A, K, w, K, bias, 1, R, nout, O, nout, M, N, K, nrows, nout, act, 0, 1, 0, 0, 0, hint, nullptr);
            if (st != GP_OK) { fprintf(stderr, "gp_gemm failed (%d) M=%d N=%d K=%d hint=%d\n", (int)st, M, N, K, hint); exit(3); }
        };
        float err = -1.f, rmax = 0.f;
        go(g_arena);
        CK(hipDeviceSynchronize());
        if (ref) {
            CK(hipMemset(res2, 0, 8));
            if (act == 3) hipLaunchKernelGGL(cmp_geglu_kernel, dim3(512), dim3(256), 0, 0, O, nout, ref, M, N, res2);
            else hipLaunchKernelGGL(cmp_kernel, dim3(512), dim3(256), 0, 0, O, nout, ref, M, N, 0, res2);
            float h[2];
            CK(hipMemcpy(h, res2, 8, hipMemcpyDeviceToHost));
            err = h[0]; rmax = h[1];
        }
        const float hot = time_us([&](int) { go(g_arena); }, g_iters);
        float cold = -1.f;
        if (g_cold) {
            const size_t span = ARENA / 2 - wel - 4096;
            cold = time_us([&](int i) { go(g_arena + (((size_t)(i + 1) * 37 * wel) % span & ~(size_t)63)); }, g_iters);
        }
        float prod = -1.f;
        if (g_cold >= 2) {  // the activation operand freshly written by another kernel before every launch (what the pipeline looks like)
            static h16* Asrc = nullptr;
            static size_t Asrc_n = 0;
            const size_t an = (size_t)M * K;
            if (Asrc_n < an) { if (Asrc) CK(hipFree(Asrc)); CK(hipMalloc(&Asrc, an * 2)); Asrc_n = an; fill(Asrc, an, 11); }
            const size_t span = ARENA / 2 - wel - 4096;
            auto cp = [&]() { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)Asrc, (uint4*)A, an / 8); };
            const float both = time_us([&](int i) { cp(); go(g_arena + (((size_t)(i + 1) * 37 * wel) % span & ~(size_t)63)); }, g_iters);
            const float alone = time_us([&](int) { cp(); }, g_iters);
            prod = both - alone;
        }
        printf("gemm M=%-6d N=%-6d K=%-5d act=%d res=%d hint=%d  hot %8.2f us %7.1f TF/s   cold %8.2f us %7.1f TF/s   produced %8.2f us   relerr %.2e\n", M, N, K, act, use_res,
               hint, hot, flops / hot * 1e-6, cold, cold > 0 ? flops / cold * 1e-6 : 0.0, prod, rmax > 0 ? err / rmax : -1.0);
        fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(O)); CK(hipFree(bias)); CK(hipFree(res2));
    if (R) CK(hipFree(R));
    if (ref) CK(hipFree(ref));
}

static void bench_conv(const std::vector<long long>& a) {
    const int B = (int)a[0], H = (int)a[1], W = (int)a[2], Cin = (int)a[3], Cout = (int)a[4];
    const int ups = a.size() > 5 ? (int)a[5] : 0;
    std::vector<int> hints;
    for (size_t i = 6; i < a.size(); ++i) hints.push_back((int)a[i]);
    if (hints.empty()) hints.push_back(0);
    const int Ho = ups ? 2 * H : H, Wo = ups ? 2 * W : W;
    const int nrows = gp_packed_rows(Cout);
    h16 *X, *O;
    float* bias;
    CK(hipMalloc(&X, (size_t)B * H * W * Cin * 2));
    CK(hipMalloc(&O, (size_t)B * Ho * Wo * Cout * 2));
    CK(hipMalloc(&bias, (size_t)nrows * 4));
    fill(X, (size_t)B * H * W * Cin, 21);
    fillf(bias, nrows, 7, 0.5f);
    const size_t wel = (size_t)nrows * 9 * Cin;
    const double flops = 2.0 * B * Ho * Wo * (double)Cout * Cin * 9;
    for (int hint : hints) {
        auto go = [&](const h16* w) {
            gp_status st = gp_conv2d(X, w, bias, nullptr, O, B, H, W, Cin, Cout, 3, 1, 1, 1, Ho, Wo, ups ? Ho : 0, ups ? Wo : 0, 0, Cout, 0, hint, nullptr);
            if (st != GP_OK) { fprintf(stderr, "gp_conv2d failed (%d)\n", (int)st); exit(3); }
        };
        const float hot = time_us([&](int) { go(g_arena); }, g_iters);
        float cold = -1.f;
        if (g_cold) {
            const size_t span = ARENA / 2 - wel - 4096;
            cold = time_us([&](int i) { go(g_arena + (((size_t)(i + 1) * 5 * wel) % span & ~(size_t)63)); }, g_iters);
        }
        printf("conv B=%d %dx%d Cin=%-5d Cout=%-5d ups=%d hint=%d  hot %8.2f us %7.1f TF/s   cold %8.2f us %7.1f TF/s\n", B, H, W, Cin, Cout, ups, hint, hot,
               flops / hot * 1e-6, cold, cold > 0 ? flops / cold * 1e-6 : 0.0);
        fflush(stdout);
    }
    CK(hipFree(X)); CK(hipFree(O)); CK(hipFree(bias));
}

__global__ void checksum_kernel(const h16* p, size_t n, float* out) {  // per-block weighted sums (order-independent enough to compare two builds bit for bit)
    __shared__ float red[256];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += bf2f(p[i]) * (float)(1 + (i % 7));
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
static double checksum(const h16* p, size_t n) {
    float* d;
    CK(hipMalloc(&d, 1024 * 4));
    hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, p, n, d);
    std::vector<float> h(1024);
    CK(hipMemcpy(h.data(), d, 1024 * 4, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    double t = 0;
    for (float v : h) t += v;
    return t;
}
static void bench_convg(const std::vector<long long>& a) {  // convg:B,H,W,Cin,Cout[,silu[,ups]]  conv3x3(act(GroupNorm(in))) with the apply fused (gp_conv2d_gn: stats pass + conv)
    const int B = (int)a[0], H = (int)a[1], W = (int)a[2], Cin = (int)a[3], Cout = (int)a[4];
    const int silu = a.size() > 5 ? (int)a[5] : 1, ups = a.size() > 6 ? (int)a[6] : 0;
    const int Ho = ups ? 2 * H : H, Wo = ups ? 2 * W : W;
    const int nrows = gp_packed_rows(Cout);
    h16 *X, *O;
    float *bias, *gamma, *beta;
    CK(hipMalloc(&X, (size_t)B * H * W * Cin * 2));
    CK(hipMalloc(&O, (size_t)B * Ho * Wo * Cout * 2));
    CK(hipMalloc(&bias, (size_t)nrows * 4));
    CK(hipMalloc(&gamma, (size_t)Cin * 4));
    CK(hipMalloc(&beta, (size_t)Cin * 4));
    fill(X, (size_t)B * H * W * Cin, 21);
    fillf(bias, nrows, 7, 0.5f);
    fillf(gamma, Cin, 9, 1.0f);
    fillf(beta, Cin, 11, 0.3f);
    const double flops = 2.0 * B * Ho * Wo * (double)Cout * Cin * 9;
    auto go = [&](const h16* w) {
        gp_status st = gp_conv2d_gn(X, w, bias, nullptr, O, B, H, W, Cin, Cout, ups, 0, gamma, beta, 32, 1e-6f, silu, nullptr);
        if (st != GP_OK) { fprintf(stderr, "gp_conv2d_gn failed (%d)\n", (int)st); exit(3); }
    };
    const float hot = time_us([&](int) { go(g_arena); }, g_iters);
    go(g_arena);
    CK(hipDeviceSynchronize());
    printf("convg B=%d %dx%d Cin=%-5d Cout=%-5d silu=%d ups=%d  %8.2f us %7.1f TF/s (statistics pass + fused conv)  checksum %.9e\n", B, H, W, Cin, Cout, silu, ups, hot,
           flops / hot * 1e-6, checksum(O, (size_t)B * Ho * Wo * Cout));
    fflush(stdout);
    CK(hipFree(X)); CK(hipFree(O)); CK(hipFree(bias)); CK(hipFree(gamma)); CK(hipFree(beta));
}

static void bench_convs(const std::vector<long long>& a) {  // convs:B,H,W,Cin,Cout  conv3x3 whose epilogue leaves GroupNorm statistics (+ finalize launch)
    const int B = (int)a[0], H = (int)a[1], W = (int)a[2], Cin = (int)a[3], Cout = (int)a[4];
    const int nrows = gp_packed_rows(Cout);
    h16 *X, *O;
    float *bias, *g, *bt, *sc, *sh;
    CK(hipMalloc(&X, (size_t)B * H * W * Cin * 2)); CK(hipMalloc(&O, (size_t)B * H * W * Cout * 2));
    CK(hipMalloc(&bias, nrows * 4)); CK(hipMalloc(&g, Cout * 4)); CK(hipMalloc(&bt, Cout * 4)); CK(hipMalloc(&sc, B * Cout * 4)); CK(hipMalloc(&sh, B * Cout * 4));
    fill(X, (size_t)B * H * W * Cin, 21); fillf(bias, nrows, 7, 0.5f); fillf(g, Cout, 8); fillf(bt, Cout, 9);
    const double flops = 2.0 * B * H * W * (double)Cout * Cin * 9;
    auto go = [&]() {
        if (gp_conv2d_stats(X, g_arena, bias, nullptr, O, B, H, W, Cin, Cout, 3, 0, 0, g, bt, 32, 1e-6f, sc, sh, nullptr) != GP_OK) { fprintf(stderr, "conv2d_stats failed\n"); exit(3); }
    };
    const float t = time_us([&](int) { go(); }, g_iters);
    std::vector<float> hs((size_t)B * Cout), hh((size_t)B * Cout);
    CK(hipMemcpy(hs.data(), sc, hs.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hh.data(), sh, hh.size() * 4, hipMemcpyDeviceToHost));
    double c1 = 0, c2 = 0;
    for (size_t i = 0; i < hs.size(); ++i) { c1 += hs[i] * (double)((i % 7) + 1); c2 += hh[i] * (double)((i % 5) + 1); }
    printf("convs B=%d %dx%d Cin=%-5d Cout=%-5d  %8.2f us %7.1f TF/s (conv + stats epilogue + finalize)  scale/shift checksums %.8e %.8e\n", B, H, W, Cin, Cout, t,
           flops / t * 1e-6, c1, c2);
    fflush(stdout);
    CK(hipFree(X)); CK(hipFree(O)); CK(hipFree(bias)); CK(hipFree(g)); CK(hipFree(bt)); CK(hipFree(sc)); CK(hipFree(sh));
}

static void bench_attn(const std::vector<long long>& a) {
    const int B = (int)a[0], T = (int)a[1], heads = (int)a[2];
    const int C = heads * 64, Tpad = (T + 63) / 64 * 64;
    h16 *QK, *VT, *O;
    CK(hipMalloc(&QK, (size_t)B * T * 2 * C * 2));
    CK(hipMalloc(&VT, (size_t)B * C * Tpad * 2));
    CK(hipMalloc(&O, (size_t)B * T * C * 2));
    fill(QK, (size_t)B * T * 2 * C, 31, 1.5f);
    fill(VT, (size_t)B * C * Tpad, 33);
    const double flops = 4.0 * B * heads * (double)T * T * 64;
    const float hot = time_us([&](int) {
        if (gp_flash_attention(QK, QK + C, VT, O, B, T, heads, 2 * C, 2 * C, Tpad, C, nullptr) != GP_OK) { fprintf(stderr, "attn failed\n"); exit(3); }
    }, g_iters);
    printf("attn B=%d T=%-5d heads=%-2d  %8.2f us %7.1f TF/s\n", B, T, heads, hot, flops / hot * 1e-6);
    fflush(stdout);
    CK(hipFree(QK)); CK(hipFree(VT)); CK(hipFree(O));
}

static void bench_gn(const std::vector<long long>& a) {  // gn:B,HW,C,silu  (statistics + finalize + apply: three launches)
    const int B = (int)a[0], HW = (int)a[1], C = (int)a[2], silu = a.size() > 3 ? (int)a[3] : 1;
    h16 *X, *Y;
    float *g, *bt;
    const size_t n = (size_t)B * HW * C;
    CK(hipMalloc(&X, n * 2)); CK(hipMalloc(&Y, n * 2)); CK(hipMalloc(&g, C * 4)); CK(hipMalloc(&bt, C * 4));
    fill(X, n, 41); fillf(g, C, 3); fillf(bt, C, 4);
    const float t = time_us([&](int) {
        if (gp_groupnorm(X, Y, g, bt, B, HW, C, 32, 1e-6f, silu, nullptr) != GP_OK) { fprintf(stderr, "gn failed\n"); exit(3); }
    }, g_iters);
    // checksum of the output (A/B runs of two builds / env switches must print the same value)
    std::vector<h16> hy(n > 4000000 ? 4000000 : n);
    CK(hipMemcpy(hy.data(), Y, hy.size() * 2, hipMemcpyDeviceToHost));
    double cs = 0;
    for (size_t i = 0; i < hy.size(); ++i) cs += bf2f(hy[i]) * (double)((i % 7) + 1);
    printf("gn   B=%d HW=%-7d C=%-5d silu=%d  %8.2f us  (3 passes over %.0f MB: %.2f TB/s)  checksum %.6e\n", B, HW, C, silu, t, n * 2 / 1e6, 3.0 * n * 2 / t * 1e-6, cs);
    fflush(stdout);
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(g)); CK(hipFree(bt));
}

static void bench_xfold(const std::vector<long long>& a) {  // xfold:rows,C,heads
    const int rows = (int)a[0], C = (int)a[1], heads = (int)a[2];
    h16 *Yb, *Yo, *N3;
    float *U, *u0, *G, *c0, *g3, *b3;
    CK(hipMalloc(&Yb, (size_t)rows * C * 2)); CK(hipMalloc(&Yo, (size_t)rows * C * 2)); CK(hipMalloc(&N3, (size_t)rows * C * 2));
    CK(hipMalloc(&U, (size_t)heads * C * 4)); CK(hipMalloc(&G, (size_t)heads * C * 4)); CK(hipMalloc(&u0, heads * 4));
    CK(hipMalloc(&c0, C * 4)); CK(hipMalloc(&g3, C * 4)); CK(hipMalloc(&b3, C * 4));
    fill(Yb, (size_t)rows * C, 51); fillf(U, (size_t)heads * C, 1, 0.05f); fillf(G, (size_t)heads * C, 2, 0.05f); fillf(u0, heads, 3); fillf(c0, C, 4);
    fillf(g3, C, 5); fillf(b3, C, 6);
    const float t = time_us([&](int) {
        if (gp_cross_attention_fold(Yb, Yo, N3, U, u0, G, c0, g3, b3, rows, C, heads, 1e-5f, nullptr) != GP_OK) { fprintf(stderr, "xfold failed\n"); exit(3); }
    }, g_iters);
    std::vector<h16> hy((size_t)rows * C), hn((size_t)rows * C);
    CK(hipMemcpy(hy.data(), Yo, hy.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hn.data(), N3, hn.size() * 2, hipMemcpyDeviceToHost));
    double c1 = 0, c2 = 0;
    for (size_t i = 0; i < hy.size(); ++i) { c1 += bf2f(hy[i]) * (double)((i % 7) + 1); c2 += bf2f(hn[i]) * (double)((i % 5) + 1); }
    printf("xfold rows=%-6d C=%-5d heads=%-2d  %8.2f us  (%.0f MB: %.2f TB/s)  checksums %.8e %.8e\n", rows, C, heads, t, 3.0 * rows * C * 2 / 1e6,
           3.0 * rows * C * 2 / t * 1e-6, c1, c2);
    fflush(stdout);
}

#ifdef KBENCH_HAVE_QKV
static void bench_qkv(const std::vector<long long>& a) {
    const int Bimg = (int)a[0], T = (int)a[1], C = (int)a[2];
    const int hint = a.size() > 3 ? (int)a[3] : 0;
    const int M = Bimg * T, Tpad = (T + 63) / 64 * 64, N = 3 * C;
    const int nrows = gp_packed_rows(N);
    h16 *A, *QK, *VT;
    float *ref, *res2;
    CK(hipMalloc(&A, (size_t)M * C * 2));
    CK(hipMalloc(&QK, (size_t)M * 2 * C * 2));
    CK(hipMalloc(&VT, (size_t)Bimg * C * Tpad * 2));
    CK(hipMalloc(&ref, (size_t)M * N * 4));
    CK(hipMalloc(&res2, 8));
    fill(A, (size_t)M * C, 11);
    const size_t wel = (size_t)nrows * C;
    hipLaunchKernelGGL(ref_gemm_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, A, C, g_arena, C, nullptr, nullptr, 0, ref, M, N, C);
    auto go = [&](const h16* w) {
        gp_status st = gp_gemm_qkv(A, C, w, C, nrows, C, QK, VT, Bimg, T, C, Tpad, nullptr);
        if (st != GP_OK) { fprintf(stderr, "gp_gemm_qkv failed (%d)\n", (int)st); exit(3); }
    };
    CK(hipMemset(QK, 0xff, (size_t)M * 2 * C * 2));
    CK(hipMemset(VT, 0xff, (size_t)Bimg * C * Tpad * 2));
    go(g_arena);
    CK(hipDeviceSynchronize());
    // check q|k against columns [0, 2C) and v^T against columns [2C, 3C) image by image
    std::vector<h16> hqk((size_t)M * 2 * C), hvt((size_t)Bimg * C * Tpad);
    std::vector<float> href((size_t)M * N);
    CK(hipMemcpy(hqk.data(), QK, hqk.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hvt.data(), VT, hvt.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, r = 0, pad = 0;
    for (int m = 0; m < M; ++m) {
        const int b = m / T, t = m % T;
        for (int n = 0; n < 2 * C; ++n) { const double d = fabs(bf2f(hqk[(size_t)m * 2 * C + n]) - href[(size_t)m * N + n]); e1 = d > e1 ? d : e1; }
        for (int c = 0; c < C; ++c) {
            const double f = href[(size_t)m * N + 2 * C + c];
            const double d = fabs(bf2f(hvt[((size_t)b * C + c) * Tpad + t]) - f);
            e2 = d > e2 ? d : e2;
            r = fabs(f) > r ? fabs(f) : r;
        }
    }
    for (int b = 0; b < Bimg; ++b)
        for (int c = 0; c < C; ++c)
            for (int t = T; t < Tpad; ++t) pad += fabs(bf2f(hvt[((size_t)b * C + c) * Tpad + t]));
    const double flops = 2.0 * M * (double)N * C;
    const float hot = time_us([&](int) { go(g_arena); }, g_iters);
    const size_t span = ARENA / 2 - wel - 4096;
    const float cold = time_us([&](int i) { go(g_arena + (((size_t)(i + 1) * 37 * wel) % span & ~(size_t)63)); }, g_iters);
    printf("qkv  B=%d T=%-5d C=%-5d hint=%d  hot %8.2f us %7.1f TF/s   cold %8.2f us %7.1f TF/s   relerr qk %.2e vT %.2e pad %.1e\n", Bimg, T, C, hint, hot,
           flops / hot * 1e-6, cold, flops / cold * 1e-6, e1 / r, e2 / r, pad);
    fflush(stdout);
    CK(hipFree(A)); CK(hipFree(QK)); CK(hipFree(VT)); CK(hipFree(ref)); CK(hipFree(res2));
}
#endif

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs, %s\n", pr.name, pr.multiProcessorCount, gp_version());
    CK(hipMalloc(&g_arena, ARENA));
    fill(g_arena, ARENA / 2, 3, 0.05f);
    CK(hipDeviceSynchronize());
    printf("# mfma peak %.0f TF/s\n", gp_mfma_peak_tflops(0, nullptr));
    for (int i = 1; i < argc; ++i) {
        const char* s = argv[i];
        if (!strncmp(s, "iters=", 6)) g_iters = atoi(s + 6);
        else if (!strncmp(s, "cold=", 5)) g_cold = atoi(s + 5);
        else if (!strncmp(s, "check=", 6)) g_check = atoi(s + 6);
        else if (!strncmp(s, "gemm:", 5)) bench_gemm(parse_nums(s + 5));
        else if (!strncmp(s, "conv:", 5)) bench_conv(parse_nums(s + 5));
        else if (!strncmp(s, "attn:", 5)) bench_attn(parse_nums(s + 5));
        else if (!strncmp(s, "convs:", 6)) bench_convs(parse_nums(s + 6));
        else if (!strncmp(s, "convg:", 6)) bench_convg(parse_nums(s + 6));
        else if (!strncmp(s, "gn:", 3)) bench_gn(parse_nums(s + 3));
        else if (!strncmp(s, "xfold:", 6)) bench_xfold(parse_nums(s + 6));
#ifdef KBENCH_HAVE_QKV
        else if (!strncmp(s, "qkv:", 4)) bench_qkv(parse_nums(s + 4));
#endif
        else { fprintf(stderr, "unknown spec %s\n", s); return 1; }
    }
    return 0;
}
