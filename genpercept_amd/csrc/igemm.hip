// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// One kernel family covers K1/K2/K3/K10 of SURVEY.md §2.3: conv3x3 (stride 1/2, symmetric or right/bottom-only
// padding, optional fused nearest-upsample of the input), conv1x1 / Linear, batched "A x B^T" (VAE attention).
//
//   out[m][n] = act( sum_{tap, c} in[pixel(m, tap)][c] * wt[n][tap][c] + bias ) (+ res[m][n])
//
// Data layout (HBM): activations NHWC bf16, weights [Cout][tap][Cin] bf16 (k-contiguous), fp32 bias, fp32 accumulate.
// Tiling: BM pixels x BN channels per workgroup of 4 or 8 waves, BK = 64 channels of one tap per K-step.
//  * Both operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane) into 128-byte rows with a 16-byte
//    slot XOR swizzle applied on the SOURCE address (the DMA destination is lane-linear); padding / out-of-range rows
//    read a zero page.  Pixel tile and weight tile use different swizzles, each conflict-free for its ds_read_b128 pattern.
//  * NSTAGE-deep LDS ring: counted s_waitcnt vmcnt + raw s_barrier, NSTAGE-1 K-steps of DMA in flight under the MFMAs.
//  * Role split: the first half of the waves runs MFMAs then issues DMA, the second half the other way round, so each
//    SIMD always has one wave feeding the matrix pipe while its partner feeds the memory pipe (measured: lock-step issue
//    made DMA and MFMA time add up instead of overlap).
//  * The MFMA takes the WEIGHTS as A operand and the PIXELS as B operand, and the weight-fragment rows are chosen so that
//    a lane ends up with 8 consecutive output channels of one pixel: bias is two float4 loaded before the K loop, the
//    bf16 store is 16 bytes per lane, the residual one 16-byte load.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "epilogue.h"
#include "kernels.h"

#include <atomic>

std::mutex& gp_attr_mutex() {
    static std::mutex m;
    return m;
}

static GpSwitches g_switches = [] {
    GpSwitches s{};
    s.gn_fuse_max_slices = s.gn_fuse_below_px = s.vt_tile = s.xfold_lds = -1;
    s.qkv_fuse_max_rows = 1 << 30;
    return s;
}();
const GpSwitches& gp_sw() { return g_switches; }
void gp_switches_reload() {
    auto flag = [](const char* n) { return getenv(n) != nullptr ? 1 : 0; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    GpSwitches s{};
    s.flash_ring3 = flag("GENPERCEPT_FLASH_RING3");
    s.no_flash512 = flag("GENPERCEPT_NO_FLASH512");
    s.f5_dbg = num("GENPERCEPT_F5_DBG", 0);              // timing ablations only: 1 no K DMA, 2 no V DMA
    s.no_conv_few = flag("GENPERCEPT_NO_CONV_FEW");
    s.no_conv_img = flag("GENPERCEPT_NO_CONV_IMG");
    s.conv_img_s = num("GENPERCEPT_CONV_IMG_S", 0);
    s.no_cross_fold = flag("GENPERCEPT_NO_CROSS_FOLD");
    s.no_gn_fusion = flag("GENPERCEPT_NO_GN_FUSION");
    s.gn_fuse_max_slices = num("GENPERCEPT_GN_FUSE_MAX_SLICES", -1);
    s.gn_fuse_below_px = num("GENPERCEPT_GN_FUSE_BELOW_PX", -1);
    s.no_stats_fusion = flag("GENPERCEPT_NO_STATS_FUSION");
    s.vt_tile = num("GENPERCEPT_VT_TILE", -1);
    s.no_gn_small = flag("GENPERCEPT_NO_GN_SMALL");
    s.fp32_scores = flag("GENPERCEPT_FP32_SCORES");
    s.no_qkv_fuse = flag("GENPERCEPT_NO_QKV_FUSE");
    s.qkv_fuse_max_rows = num("GENPERCEPT_QKV_FUSE_MAX_ROWS", 1 << 30);
    s.no_rgb_conv = flag("GENPERCEPT_NO_RGB_CONV");
    s.igemm_dbg = num("GENPERCEPT_IGEMM_DBG", 0);        // profiling ablations (tools/conv_bench.py, tools/kbench)
    s.no_splitk = flag("GENPERCEPT_NO_SPLITK");
    s.no_halo = flag("GENPERCEPT_NO_HALO");              // generic implicit GEMM everywhere
    s.no_pgemm = flag("GENPERCEPT_NO_PGEMM");
    s.gn_apply_old = flag("GENPERCEPT_GN_APPLY_OLD");
    s.no_up_phases = flag("GENPERCEPT_NO_UP_PHASES");      // A/B: the x2-upsample convs through the nine-tap upsample kernel instead of four phase convolutions (r5)
    s.gn_small_old = flag("GENPERCEPT_GN_SMALL_OLD");      // A/B: gn_small_kernel (three passes over L2) instead of gn_small_reg_kernel (r5)
    s.xfold_lds = num("GENPERCEPT_XFOLD_LDS", -1);       // 0 = the r2 cross-attention fold kernel
    s.no_fin_fuse = flag("GENPERCEPT_NO_FIN_FUSE");
    s.c_no_flash = flag("GENPERCEPT_C_NO_FLASH");         // A/B: contract precision's head_dim-64 attention unfused (logits in HBM) instead of flash_attn64_split_kernel
    s.pgemm_ring3 = flag("GENPERCEPT_PGEMM_RING3");       // A/B: the 128-row persistent GEMM with the 3-deep ring of r2 / r3 (default since r4: 4-deep)
    // Engines on other host threads read g_switches on their launch paths: write it only when the environment really changed (tests / A/B
    // scripts, between calls), under a lock, so that concurrent engine creation with an unchanged environment never stores to it.
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (memcmp(&g_switches, &s, sizeof(s)) != 0) g_switches = s;
}

constexpr bool PRIO = true;   // s_setprio around the MFMA cluster made hipcc wait lgkmcnt(0) before the first MFMA
template <int N>
GP_DEV void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KS: 1 = 1x1 / plain rows, 3 = 3x3, 4 = 3x3 on a nearest-upsampled input
template <int BM, int BN, int WM, int WN, int KS, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IGemmParams p) {
    constexpr int NW = WM * WN;                     // waves per workgroup (4 or 8)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16, FP = FN / 2;  // FP: pairs of weight fragments (32 output channels)
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW);  // 8-row DMA groups per wave
    constexpr int LPS = A_IT + B_IT;                // LDS-DMA instructions per wave per stage
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr bool CONV = KS != 1, UPS = KS == 4;
    static_assert((NW == 4 || NW == 8) && TM % 16 == 0 && TN % 32 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool second_half = wave >= NW / 2;

    const int ncols = p.N > p.n_store ? p.N : p.n_store;  // n_store > N: zero-filled padding columns
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int sid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (sid / tiles_n) * BM, n0 = (sid % tiles_n) * BN;
    const int z = blockIdx.y;

    const h16_t* in = p.in + (long long)z * p.in_bs;
    const h16_t* wt = p.wt + (long long)z * p.wt_bs;
    const int Cin = p.Cin;
    const int cpt = Cin >> 6;                       // 64-channel chunks per tap
    const int nk_all = (CONV ? 9 : 1) * cpt;
    // split-K: slice blockIdx.z of p.ksplit handles K-steps [k_begin, k_begin + nk)
    const int zs = p.ksplit > 1 ? (int)blockIdx.z : 0;
    const int k_begin = p.ksplit > 1 ? (int)((long long)nk_all * zs / p.ksplit) : 0;
    const int nk = p.ksplit > 1 ? (int)((long long)nk_all * (zs + 1) / p.ksplit) - k_begin : nk_all;

    // this lane's 16-byte source chunk inside a 128-byte row (swizzled; identical for every DMA group of the wave)
    //   pixel tile:  slot ^ (row & 7)                              rows read 16-consecutive (conflict-free for any start row)
    //   weight tile: slot ^ (b1 | b3 << 1 | b4 << 2) of the row    rows read as {8q + 4h + r}
    const int chunk_a = (lane & 7) ^ (lane >> 3);
    const int chunk_w = (lane & 7) ^ (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2));
    const h16_t* zsrc_a = p.zero + chunk_a * 8;
    const h16_t* zsrc_w = p.zero + chunk_w * 8;

    // ---- per-lane row descriptors ---------------------------------------------------------------------------------
    const h16_t* a_base[A_IT];   // plain: row pointer; conv: pointer of the (virtual) top-left tap pixel; ups: image base
    unsigned a_mask[A_IT];        // bit t: tap t of this row is inside the image (plain rows: bit 0)
    int a_y0[UPS ? A_IT : 1], a_x0[UPS ? A_IT : 1];
    const h16_t* a_tap[UPS ? A_IT : 1];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + (lane >> 3);
        const bool ok = m < p.M;
        if (!CONV) {
            a_base[i] = in + (long long)m * p.lda + chunk_a * 8;
            a_mask[i] = ok ? 1u : 0u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
            const h16_t* img = in + (long long)b * p.Hi * p.Wi * Cin + chunk_a * 8;
            unsigned mk = 0;
            const int hlim = UPS ? p.Hu : p.Hi, wlim = UPS ? p.Wu : p.Wi;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (ok && (unsigned)(y0 + t / 3) < (unsigned)hlim && (unsigned)(x0 + t % 3) < (unsigned)wlim) mk |= 1u << t;
            a_mask[i] = mk;
            if constexpr (UPS) {
                a_base[i] = img;
                a_y0[i] = y0;
                a_x0[i] = x0;
                a_tap[i] = img;
            } else {
                a_base[i] = img + ((long long)y0 * p.Wi + x0) * Cin;  // may point outside the image; only dereferenced under the mask
            }
        }
    }
    const h16_t* w_base[B_IT];
    bool w_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + (wave + NW * i) * 8 + (lane >> 3);
        w_ok[i] = n < p.n_rows;
        w_base[i] = wt + (long long)n * p.ldw + chunk_w * 8;
    }
    const float ups_sy = UPS ? (float)p.Hi / (float)p.Hu : 1.f;
    const float ups_sx = UPS ? (float)p.Wi / (float)p.Wu : 1.f;

    int st_tap = CONV ? k_begin / cpt : 0, st_cc = CONV ? k_begin - st_tap * cpt : k_begin;
    int st_ky = st_tap / 3, st_kx = st_tap - 3 * st_ky;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE;
        if constexpr (UPS) {
          if (st_cc == 0) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = min((int)floorf((float)(a_y0[i] + st_ky) * ups_sy), p.Hi - 1);
                const int ix = min((int)floorf((float)(a_x0[i] + st_kx) * ups_sx), p.Wi - 1);
                a_tap[i] = a_base[i] + ((long long)iy * p.Wi + ix) * Cin;
            }
          }
        }
        const int koff = st_cc << 6;
        const long long aoff = CONV && !UPS ? (long long)(st_ky * p.Wi + st_kx) * Cin + koff : koff;  // wave-uniform
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = (a_mask[i] >> st_tap) & 1u;
            const h16_t* ab;
            if constexpr (UPS) ab = a_tap[i]; else ab = a_base[i];
            const h16_t* src = ok ? ab + aoff : zsrc_a;
            glds16(src, sb + (wave + NW * i) * 1024);
        }
        const int woff = st_tap * Cin + koff;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const h16_t* src = w_ok[i] ? w_base[i] + woff : zsrc_w;
            glds16(src, sb + A_BYTES + (wave + NW * i) * 1024);
        }
        if (++st_cc == cpt) {
            st_cc = 0;
            if (CONV) {
                ++st_tap;
                if (++st_kx == 3) { st_kx = 0; ++st_ky; }
            }
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: pixel fragment j = 16 consecutive tile rows; weight fragment i of pair ip = i/2 takes tile rows
    // 32*ip + 8*(a>>2) + 4*(i&1) + (a&3) for MFMA row a = lane&15, i.e. accumulator (q = lane>>4, r) of fragments
    // (2ip, 2ip+1) are output channels 32*ip + 8q + r and 32*ip + 8q + 4 + r: 8 consecutive channels per lane.
    const int a15 = lane & 15;
    const int xr_a = lane & 7;
    const int xr_w = ((a15 >> 1) & 1) | (((a15 >> 2) & 1) << 1) | (((a15 >> 3) & 1) << 2);
    const int a_row_off = (wm * TM + a15) * 128;
    const int w_row_off = A_BYTES + (wn * TN + 8 * (a15 >> 2) + (a15 & 3)) * 128;
    auto compute = [&](int buf) {
        const char* sb = smem + buf * STAGE;
        // all 2*(FN+FM) fragment reads are issued up front so the LDS latency of the second k-half hides under the first
        // half's MFMAs (hipcc otherwise emits read-batch / lgkmcnt(0) / MFMA-batch with the latency exposed each time)
        h16x8_t wf[2][FN], xf[2][FM];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 4 + (lane >> 4);
            const int so_a = (sl ^ xr_a) << 4, so_w = (sl ^ xr_w) << 4;
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[kk][i] = *(const h16x8_t*)(sb + w_row_off + (i >> 1) * 4096 + (i & 1) * 512 + so_w);
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[kk][j] = *(const h16x8_t*)(sb + a_row_off + j * 2048 + so_a);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(wf[kk][i], xf[kk][j], acc[i][j]);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: this lane's bias values, then fill NSTAGE-1 ring slots ---------------------------------------------------
    // (bias first: an ordinary load issued while LDS-DMA is in flight makes hipcc drain vmcnt to 0 at its first use)
    float bcol[FP][8];
    load_bias_cols<FP>(p, z, n0 + wn * TN, 8 * (lane >> 4), bcol);
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk && !(p.dbg & 32)) stage(s0);

    // ---- main loop -------------------------------------------------------------------------------------------------------
    // Per step: counted vmcnt (this wave's DMAs of step kt have landed) -> raw s_barrier (everybody's have, and everybody
    // is done reading the slot about to be refilled) -> {MFMAs of step kt, DMA of step kt+NSTAGE-1} in role-dependent order.
    // __syncthreads() would drain vmcnt to 0 and serialise the ring (cdna guide, "Pipelining across barriers").
    int cur = 0, nxt = NSTAGE - 1;
    for (int kt = 0; kt < ((p.dbg & 8) ? 0 : nk); ++kt) {
        const int ahead = min(NSTAGE - 2, nk - 1 - kt);  // stages issued after step kt's
        if (!(p.dbg & 4)) {
            if (ahead >= 2) wait_vm_n<2 * LPS>();
            else if (ahead == 1) wait_vm_n<LPS>();
            else wait_vm_n<0>();
            __builtin_amdgcn_s_barrier();
        }
        const bool do_stage = kt + NSTAGE - 1 < nk && !(p.dbg & 1);
        const bool do_comp = !(p.dbg & 2);
        if (second_half && do_stage) stage(nxt);   // one compute() site: two would make the compiler shuffle all accumulators
        if (do_comp) compute(cur);
        if (!second_half && do_stage) stage(nxt);
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    }

    // ---- epilogue (epilogue.h): LDS-staged, fully coalesced bf16 stores ---------------------------------------------------
    if ((p.dbg & 16) && m0 >= 0) return;  // ablation: no epilogue
    if (p.ksplit > 1) {  // fp32 partial sums of this K slice (launch_igemm set out_fp32, no bias / residual / activation)
        IGemmParams pe = p;
        pe.out = (float*)p.out + (long long)zs * p.split_bs;
        conv_epilogue<BM, BN, WM, WN, 64 * NW>(pe, acc, bcol, n0, z, wave, lane, smem, [&](int pr) {
            const int m = m0 + pr;
            return m < p.M ? m : -1;
        }, sid / tiles_n);
        return;
    }
    conv_epilogue<BM, BN, WM, WN, 64 * NW>(p, acc, bcol, n0, z, wave, lane, smem, [&](int pr) {
        const int m = m0 + pr;
        return m < p.M ? m : -1;
    }, sid / tiles_n);
}

// ---- split-K: out[m][n] = act( sum_z part[z][m][n] + bias[n] ) (+ res[m][n]); columns in [n_out, n_store) are written as 0 ----------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, long long slice, int M, int N, int n_store,
                                                             const float* __restrict__ bias, const h16_t* __restrict__ res, int ldres, int act,
                                                             h16_t* __restrict__ out, int ldo, int f32) {  // f32 (contract precision): fp32 rows out, fp32 residual
    const int nv = n_store >> 2;  // 4 columns per thread (n_store % 4 == 0)
    const long long total = (long long)M * nv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / nv;
        const int c = (int)(i - m * nv) * 4;
        float4 a = *(const float4*)(part + m * n_store + c);
        for (int z = 1; z < S; ++z) {
            const float4 b = *(const float4*)(part + z * slice + m * n_store + c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e < N) {
                if (bias) v[e] += bias[c + e];
                if (res) v[e] += f32 ? ((const float*)res)[m * ldres + c + e] : h16_to_f(res[m * ldres + c + e]);
                if (act == GP_ACT_SILU) v[e] = silu_f(v[e]);
                else if (act == GP_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            } else {
                v[e] = 0.f;
            }
        }
        if (f32) *(float4*)((float*)out + m * ldo + c) = make_float4(v[0], v[1], v[2], v[3]);
        else *(uint2*)(out + m * ldo + c) = pack_h16x4(v[0], v[1], v[2], v[3]);
    }
}

// number of K slices launch_igemm will use (1 = none): 3x3 convs on tiny maps (M = 576 for 12x12 x 4 images, K = 11520 .. 23040) leave
// 180 workgroups of 4 waves on 256 CUs with 180+ dependent K-steps each -- 200 TFLOP/s; slicing K fills the machine.
int igemm_ksplit(const IGemmParams& p, int tile_hint) {
    const bool no_split = gp_sw().no_splitk;
    if (tile_hint == 0 && !no_split && conv_img_applicable(p)) return conv_img_ksplit(p);  // whole-image tiles (conv_img.hip)
    if (tile_hint != 0 && tile_hint != 2) return 1;
    if (no_split || p.ks != 3 || p.ups || p.batch > 1 || p.out_fp32 > 1 || p.act == GP_ACT_GEGLU || p.bias_mode == GP_BIAS_ROW || p.in_scale) return 1;
    if ((p.out_fp32 == 1) != (p.res_f32 != 0) && p.res) return 1;  // (the reduce kernel reads the residual in the output's element type)
    if ((p.n_store & 3) || (p.ldo & 3) || p.n_store != p.ldo) return 1;
    if (conv_uses_halo(p, tile_hint)) return 1;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const long long t64 = (long long)((p.M + 63) / 64) * ((ncols + 63) / 64);
    const int nk = 9 * (p.Cin >> 6);
    if (t64 >= 384 || nk < 32) return 1;
    int S = (int)(768 / t64);           // three 64x64 workgroups fit a CU
    if (S > 8) S = 8;
    while (S > 1 && nk / S < 12) --S;
    return S < 2 ? 1 : S;
}


template <int BM, int BN, int WM, int WN, int KS, int NSTAGE>
static void launch_one(const IGemmParams& p, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    static unsigned long long attr_mask = 0;  // per device: every GPU of the process needs its own attribute
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, WM, WN, KS, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, KS, NSTAGE>), grid, dim3(64 * WM * WN), lds, s, p);
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
static void launch_cfg(const IGemmParams& p, hipStream_t s) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles = ((p.M + BM - 1) / BM) * ((ncols + BN - 1) / BN);
    dim3 grid(tiles, p.batch > 0 ? p.batch : 1, p.ksplit > 1 ? p.ksplit : 1);
    if (p.ks == 3 && p.ups) launch_one<BM, BN, WM, WN, 4, NSTAGE>(p, grid, s);
    else if (p.ks == 3) launch_one<BM, BN, WM, WN, 3, NSTAGE>(p, grid, s);
    else launch_one<BM, BN, WM, WN, 1, NSTAGE>(p, grid, s);
}

// tile_hint: 0 auto, 1 = 128x128 (4 waves, 2-deep), 2 = 64x64 (4 waves, 3-deep), 3 = 256x32 (4 waves, 2-deep),
//            4 = 256x128 (8 waves, 3-deep ring: the large-problem configuration), 5 = conv_halo.hip, 6 = 128x64 (4 waves, 2-deep)
// true when launch_igemm(p, hint) hands the problem to conv_halo.hip (the only kernel that fuses IGemmParams::in_scale)
bool conv_uses_halo(const IGemmParams& p, int tile_hint) {
    const bool no_halo = gp_sw().no_halo;  // A/B switch: generic implicit GEMM everywhere
    if (no_halo && tile_hint != 5) return false;
    if (tile_hint == 0 && !gp_sw().no_splitk && conv_img_applicable(p)) return false;  // 24x24 / 12x12 maps: conv_img.hip
    if (!(tile_hint == 5 || tile_hint == 0) || !conv_halo_applicable(p)) return false;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const long long tiles = (long long)((p.Wo + 15) / 16) * ((p.Ho + 15) / 16) * p.B * ((ncols + 127) / 128);
    return tile_hint == 5 || (tiles >= 160 && ncols > 32);
}

bool igemm_uses_pgemm(const IGemmParams& p, int tile_hint) {
    const bool no_pgemm = gp_sw().no_pgemm;  // A/B switch
    if (tile_hint == 7) return pgemm_applicable(p);
    // fewer than 1024 rows (the UNet mid block at 768x768: 4 x 144 tokens): five 128-row tiles per column slice leave the persistent kernel's
    // workgroups one tile each, 64x64 tiles on three workgroups per CU are 20-25 % faster (tools/kbench: 11.6 vs 15.3 us at N = K = 1280,
    // 36.7 vs 46.8 us at K = 5120); the GEGLU and the fused q | k | V^T forms only exist in the persistent kernel
    if (tile_hint == 0 && p.M < 1024 && p.act != GP_ACT_GEGLU && !p.vt_out) return false;
    return tile_hint == 0 && !no_pgemm && pgemm_applicable(p);
}

static int select_cfg(const IGemmParams& p, int tile_hint) {
    int cfg = (tile_hint == 5 || tile_hint == 7) ? 0 : tile_hint;
    if (cfg == 0) {
        const int ncols = p.N > p.n_store ? p.N : p.n_store;
        const long long nb = p.batch > 0 ? p.batch : 1;
        const long long t128 = (long long)((p.M + 127) / 128) * ((ncols + 127) / 128) * nb;
        const long long t256 = (long long)((p.M + 255) / 256) * ((ncols + 127) / 128) * nb;
        const int nk = (p.ks == 3 ? 9 : 1) * (p.Cin >> 6);
        if (ncols <= 32) cfg = 3;
        else if (ncols <= 64 || t128 < 192) cfg = 2;
        else if (p.ks == 1 && nk <= 20 && nb == 1) cfg = 6;  // short-K GEMMs are bound by per-tile fixed cost: 3 small workgroups per CU overlap it
        else if (t256 >= 256) cfg = 4;
        else cfg = 1;
    }
    return cfg;
}

int igemm_tile_info(const IGemmParams& p, int tile_hint, int* mode, int* bm) {
    // the staged epilogue is the only one that accumulates statistics (epilogue.h)
    if (p.out_fp32 > 1 || p.act == GP_ACT_GEGLU || (p.ldo & 7) || p.batch > 1 || p.N != p.n_store) return 0;  // (fp32 rows, out_fp32 == 1, are staged too: contract precision)
    if (igemm_ksplit(p, tile_hint) > 1) return 0;  // partial sums: no epilogue statistics
    if (conv_uses_halo(p, tile_hint)) {
        const int R = conv_halo_stat_rows(p);
        if (R > 0) {  // persistent kernel: one partial row per workgroup
            *mode = 2;
            *bm = R;
            return R * p.B;
        }
        *mode = 1;
        *bm = 256;
        return ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16) * p.B;
    }
    const int cfg = select_cfg(p, tile_hint);
    *mode = 0;
    *bm = igemm_uses_pgemm(p, tile_hint) ? pgemm_bm(p) : (cfg == 1 || cfg == 6) ? 128 : cfg == 2 ? 64 : 256;
    const int hw = p.M / (p.B > 0 ? p.B : 1);
    if (p.B < 1 || hw * p.B != p.M || hw % *bm) return 0;  // a tile must not straddle two images
    return p.M / *bm;
}

void launch_igemm(const IGemmParams& p, int tile_hint, hipStream_t s) {
    if (conv_uses_halo(p, tile_hint)) {
        launch_conv_halo(p, s);
        return;
    }
    if (igemm_uses_pgemm(p, tile_hint)) {
        launch_pgemm(p, s);
        return;
    }
    const int S = igemm_ksplit(p, tile_hint);
    if (S > 1) {
        // the partial sums live in a workspace the CALLER owns (an engine takes it from its pool): no process-global state here
        const size_t slice = (size_t)p.M * p.n_store, need = slice * S;
        float* ws = (p.splitk_ws && (size_t)p.splitk_ws_floats >= need) ? p.splitk_ws : nullptr;
        if (ws && tile_hint == 0 && conv_img_applicable(p)) {
            launch_conv_img(p, ws, S, s);
            const long long total = (long long)p.M * (p.n_store >> 2);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, S, (long long)slice, p.M, p.N, p.n_store,
                               p.bias_mode == GP_BIAS_COL ? p.bias : nullptr, p.res, p.ldres, p.act, (h16_t*)p.out, p.ldo, p.out_fp32 == 1 ? 1 : 0);
            return;
        }
        if (ws) {
            IGemmParams q = p;
            q.out = ws; q.out_fp32 = 1; q.ldo = p.n_store; q.bias = nullptr; q.bias_mode = GP_BIAS_NONE; q.res = nullptr; q.act = GP_ACT_NONE;
            q.stats_out = nullptr; q.ksplit = S; q.split_bs = (long long)slice;
            launch_cfg<64, 64, 2, 2, 3>(q, s);
            const long long total = (long long)p.M * (p.n_store >> 2);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, S, (long long)slice, p.M, p.N, p.n_store,
                               p.bias_mode == GP_BIAS_COL ? p.bias : nullptr, p.res, p.ldres, p.act, (h16_t*)p.out, p.ldo, p.out_fp32 == 1 ? 1 : 0);
            return;
        }
    }
    const int cfg = select_cfg(p, tile_hint);
    if (cfg == 1) launch_cfg<128, 128, 2, 2, 2>(p, s);
    else if (cfg == 2) launch_cfg<64, 64, 2, 2, 3>(p, s);
    else if (cfg == 3) launch_cfg<256, 32, 4, 1, 2>(p, s);
    else if (cfg == 6) launch_cfg<128, 64, 2, 2, 2>(p, s);   // 48 KiB of LDS -> 3 workgroups per CU
    else launch_cfg<256, 128, 4, 2, 3>(p, s);
}

GP_SAT_TU(igemm)  // fp16 build: address of this translation unit's saturation flag (common.h)
