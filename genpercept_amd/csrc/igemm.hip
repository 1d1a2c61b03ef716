// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// One kernel family covers K1/K2/K3/K10 of SURVEY.md §2.3: conv3x3 (stride 1/2, symmetric or right/bottom-only
// padding, optional fused nearest-upsample of the input), conv1x1 / Linear, batched "A x B^T" (VAE attention).
//
//   out[m][n] = act( sum_{tap, c} in[pixel(m, tap)][c] * wt[n][tap][c] + bias ) (+ res[m][n])
//
// Data layout (HBM): activations NHWC bf16, weights [Cout][tap][Cin] bf16 (k-contiguous), fp32 bias, fp32 accumulate.
// Tiling: BM pixels x BN channels per 256-thread workgroup (4 waves), BK = 64 channels of one tap per K-step.
// Both operand tiles are brought HBM->LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane) into 128-byte rows with a
// 16-byte-slot XOR swizzle applied on the SOURCE address (the DMA destination is lane-linear), double-buffered: the
// DMA of K-step t+1 is in flight while the MFMAs of K-step t run.  Padding / out-of-range rows read a zero page.
// The MFMA is issued with the WEIGHTS as the A operand and the PIXELS as the B operand, so each lane ends up holding
// 4 consecutive output channels of one pixel: bias is a float4, the bf16 store is 8 bytes per lane.
#include "common.h"
#include "kernels.h"

template <int N>
GP_DEV void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int WM, int WN, int KS, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IGemmParams p) {
    constexpr int NW = WM * WN;                     // waves per workgroup (4 or 8)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW);  // 8-row DMA groups per wave
    constexpr int LPS = A_IT + B_IT;                // LDS-DMA instructions per wave per stage
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    static_assert((NW == 4 || NW == 8) && TM % 16 == 0 && TN % 16 == 0 && (FN % 2) == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ncols = p.N > p.n_store ? p.N : p.n_store;  // n_store > N: zero-filled padding columns
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int sid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (sid / tiles_n) * BM, n0 = (sid % tiles_n) * BN;
    const int z = blockIdx.y;

    const bf16_t* in = p.in + (long long)z * p.in_bs;
    const bf16_t* wt = p.wt + (long long)z * p.wt_bs;
    const int Cin = p.Cin;
    const int cpt = Cin >> 6;                       // 64-channel chunks per tap
    const int nk = (KS == 3 ? 9 : 1) * cpt;
    const int taps = (KS == 3 ? 9 : 1);

    // this lane's 16-byte source chunk inside a 128-byte row (swizzled; identical for every DMA group of the wave)
    const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const bf16_t* zsrc = p.zero + chunk * 8;

    // ---- per-lane row descriptors -------------------------------------------------------------------------
    const bf16_t* a_base[A_IT];   // KS==1: row pointer (+chunk); KS==3: image base pointer (+chunk)
    int a_y0[A_IT], a_x0[A_IT];
    bool a_ok[A_IT];
    const bf16_t* a_tap[A_IT];    // KS==3: pointer for the current tap
    bool a_tok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + (lane >> 3);
        a_ok[i] = m < p.M;
        if (KS == 1) {
            a_base[i] = in + (long long)m * p.lda + chunk * 8;
            a_y0[i] = a_x0[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_y0[i] = oy * p.stride - p.pad_t;
            a_x0[i] = ox * p.stride - p.pad_l;
            a_base[i] = in + (long long)b * p.Hi * p.Wi * Cin + chunk * 8;
        }
        a_tap[i] = a_base[i];
        a_tok[i] = a_ok[i];
    }
    const bf16_t* w_base[B_IT];
    bool w_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + (wave + NW * i) * 8 + (lane >> 3);
        w_ok[i] = n < p.n_rows;
        w_base[i] = wt + (long long)n * p.ldw + chunk * 8;
    }
    const float ups_sy = p.ups ? (float)p.Hi / (float)p.Hu : 1.f;
    const float ups_sx = p.ups ? (float)p.Wi / (float)p.Wu : 1.f;

    int st_tap = 0, st_cc = 0, st_ky = 0, st_kx = 0;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE;
        if (KS == 3 && st_cc == 0) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int iy = a_y0[i] + st_ky, ix = a_x0[i] + st_kx;
                bool ok;
                if (p.ups) {
                    ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hu && (unsigned)ix < (unsigned)p.Wu;
                    iy = min((int)floorf((float)iy * ups_sy), p.Hi - 1);
                    ix = min((int)floorf((float)ix * ups_sx), p.Wi - 1);
                } else {
                    ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                }
                a_tok[i] = ok;
                a_tap[i] = a_base[i] + ((long long)iy * p.Wi + ix) * Cin;
            }
        }
        const int koff = st_cc << 6;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bf16_t* src = a_tok[i] ? a_tap[i] + koff : zsrc;
            glds16(src, sb + (wave + NW * i) * 1024);
        }
        const int woff = st_tap * Cin + koff;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bf16_t* src = w_ok[i] ? w_base[i] + woff : zsrc;
            glds16(src, sb + A_BYTES + (wave + NW * i) * 1024);
        }
        if (++st_cc == cpt) {
            st_cc = 0;
            ++st_tap;
            if (++st_kx == 3) { st_kx = 0; ++st_ky; }
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int xr = (lane >> 1) & 7;  // swizzle term of the fragment rows (row & 15 == lane & 15)
    const int a_row_off = (wm * TM + (lane & 15)) * 128;
    const int b_row_off = A_BYTES + (wn * TN + (lane & 15)) * 128;
    auto compute = [&](int buf) {
        const char* sb = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int so = ((kk * 4 + (lane >> 4)) ^ xr) << 4;
            bf16x8_t wf[FN], xf[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[i] = *(const bf16x8_t*)(sb + b_row_off + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[j] = *(const bf16x8_t*)(sb + a_row_off + j * 2048 + so);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: an NSTAGE-deep LDS ring, NSTAGE-1 K-steps of DMA in flight under the MFMAs -----------------------
    // Per step: counted vmcnt (this wave's DMAs of step kt have landed) -> raw s_barrier (everybody's have, and everybody
    // is done reading the slot about to be refilled) -> issue the DMA of step kt+NSTAGE-1 -> MFMAs of step kt.
    // __syncthreads() would drain vmcnt to 0 and serialise the ring (cdna guide, "Pipelining across barriers").
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk) stage(s0);
    int cur = 0, nxt = NSTAGE - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSTAGE - 2, nk - 1 - kt);  // stages issued after step kt's
        if (ahead >= 2) wait_vm_n<2 * LPS>();
        else if (ahead == 1) wait_vm_n<LPS>();
        else wait_vm_n<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < nk) stage(nxt);
        compute(cur);
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const bool geglu = p.act == GP_ACT_GEGLU;
    const int n_out = geglu ? (p.N >> 1) : p.N;
    const float* bias = p.bias ? p.bias + (long long)z * p.bias_bs : nullptr;
    const bf16_t* res = p.res ? p.res + (long long)z * p.res_bs : nullptr;
    const int q4 = 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * TM + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        float rb = 0.f;
        if (p.bias_mode == GP_BIAS_ROW) rb = bias[m];
#pragma unroll
        for (int i = 0; i < FN; i += 1) {
            const int nb = n0 + wn * TN + i * 16;
            float v[4];
            int col;
            if (geglu) {
                if (i & 1) continue;
                col = (nb >> 1) + q4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[i][j][r], g = acc[i + 1][j][r];
                    if (p.bias_mode == GP_BIAS_COL && nb + q4 + r < p.N) { a += bias[nb + q4 + r]; g += bias[nb + 16 + q4 + r]; }
                    v[r] = a * gelu_erf_f(g);
                }
            } else {
                col = nb + q4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[i][j][r] + rb;
                    if (p.bias_mode == GP_BIAS_COL && col + r < p.N) a += bias[col + r];
                    v[r] = a;
                }
            }
            if (col >= p.n_store) continue;
            if (res) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < n_out) v[r] += bf2f(res[(long long)m * p.ldres + col + r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (p.act == GP_ACT_SILU) v[r] = silu_f(v[r]);
                else if (p.act == GP_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
                if (col + r >= n_out) v[r] = 0.f;
            }
            if (p.out_fp32) {
                float* o = (float*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 3 < p.n_store && (p.ldo & 3) == 0) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.n_store) o[r] = v[r];
                }
            } else {
                bf16_t* o = (bf16_t*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 3 < p.n_store && (p.ldo & 3) == 0) {
                    *(uint2*)o = pack_bf16x4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.n_store) o[r] = f2bf(v[r]);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
static void launch_cfg(const IGemmParams& p, hipStream_t s) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles = ((p.M + BM - 1) / BM) * ((ncols + BN - 1) / BN);
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    dim3 grid(tiles, p.batch > 0 ? p.batch : 1), block(64 * WM * WN);
    if (p.ks == 3) {
        static bool attr3 = false;
        if (!attr3) { (void)hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, WM, WN, 3, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr3 = true; }
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 3, NSTAGE>), grid, block, lds, s, p);
    } else {
        static bool attr1 = false;
        if (!attr1) { (void)hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, WM, WN, 1, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr1 = true; }
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, NSTAGE>), grid, block, lds, s, p);
    }
}

// tile_hint: 0 auto, 1 = 128x128 (4 waves, 2-deep), 2 = 64x64 (4 waves, 3-deep), 3 = 256x32 (4 waves, 2-deep),
//            4 = 256x128 (8 waves, 3-deep ring: the large-problem configuration)
void launch_igemm(const IGemmParams& p, int tile_hint, hipStream_t s) {
    int cfg = tile_hint;
    if (cfg == 0) {
        const int ncols = p.N > p.n_store ? p.N : p.n_store;
        const long long nb = p.batch > 0 ? p.batch : 1;
        const long long t128 = (long long)((p.M + 127) / 128) * ((ncols + 127) / 128) * nb;
        const long long t256 = (long long)((p.M + 255) / 256) * ((ncols + 127) / 128) * nb;
        if (ncols <= 32) cfg = 3;
        else if (ncols <= 64 || t128 < 192) cfg = 2;
        else if (t256 >= 256) cfg = 4;
        else cfg = 1;
    }
    if (cfg == 1) launch_cfg<128, 128, 2, 2, 2>(p, s);
    else if (cfg == 2) launch_cfg<64, 64, 2, 2, 3>(p, s);
    else if (cfg == 3) launch_cfg<256, 32, 4, 1, 2>(p, s);
    else launch_cfg<256, 128, 4, 2, 3>(p, s);
}
