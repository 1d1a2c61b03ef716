// conv_img: conv3x3 (stride 1, pad 1) on the UNet's SMALL maps -- 24x24 and 12x12 at 768x768 (custom_unet.py:341-352 mid block, down block 3,
// up blocks 0/1) -- where a workgroup's output tile is a WHOLE image: 576 output pixels = one 24x24 map or four 12x12 maps.
//
// Why: the 16x16-pixel halo kernel (conv_halo.hip) covers a 24x24 map with four tiles of which 44 % is padding and leaves 96 of 256 CUs idle
// (580 TFLOP/s); the 12x12 maps went through 64x64 implicit-GEMM tiles with split-K, which move 0.5 GB through the vector caches for a
// 30 MB weight tensor (340 TFLOP/s).  Here
//   * M tile = 576 pixels exactly (no padding rows), N tile = 64 output channels, 12 waves x (48 pixels x 64 channels);
//   * the K loop is split over S workgroups per tile (units x Cout / 64 x S ~ 240 workgroups), fp32 partial sums go to the caller's split-K
//     workspace and splitk_reduce_kernel (igemm.hip) adds bias / residual and packs -- the same reduction the 64x64 path used;
//   * per 32-channel chunk the zero-padded input of the whole unit (<= 784 pixels x 64 bytes) is staged ONCE by LDS-DMA and read by all nine
//     taps; weights stream as [3 taps][64 rows][64 bytes] tiles (one DMA instruction per thread and step) through a 4-deep ring, three
//     steps ahead: every weight byte is read from HBM exactly once per (unit, K slice);
//   * one barrier per step = (chunk, kernel row): 36 MFMAs per wave between barriers.
// LDS rows are 64 bytes (4 slots of 16); slot ^ (((row >> 2) & 1) << 1) is conflict-free for the weight fragments (16 aligned rows) and
// two-way at most for the pixel fragments, whose 16 rows start anywhere in the padded image (exhaustive check over all alignments).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {
constexpr int CI_PIX = 576, CI_BN = 64, CI_WAVES = 12, CI_THREADS = 64 * CI_WAVES;
constexpr int CI_HROWS_MAX = 784;                       // 4 x 14 x 14 (also 26 x 26 = 676)
constexpr int CI_HBUF = CI_HROWS_MAX * 64;              // one halo buffer (32 channels)
constexpr int CI_HIT = (CI_HROWS_MAX * 4 + CI_THREADS - 1) / CI_THREADS;  // 5 DMA instructions per thread and chunk
constexpr int CI_WSTEP = 3 * CI_BN * 64;                // weight tile of one step: three taps x 64 rows x 64 bytes = 768 x 16 bytes
constexpr int CI_NW = 4;                                // weight ring slots
constexpr int CI_W_OFF = 2 * CI_HBUF, CI_DUMP_OFF = CI_W_OFF + CI_NW * CI_WSTEP, CI_LDS = CI_DUMP_OFF + 1024;
static_assert(CI_LDS <= 160 * 1024, "LDS budget");

GP_DEV int ci_swz(int row) { return ((row >> 2) & 1) << 1; }

template <int N>
GP_DEV void ci_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
GP_DEV void ci_wait_dyn(int n) {  // n in 0..7 (wave-uniform)
    switch (n) {
        case 0: ci_wait<0>(); break;
        case 1: ci_wait<1>(); break;
        case 2: ci_wait<2>(); break;
        case 3: ci_wait<3>(); break;
        case 4: ci_wait<4>(); break;
        case 5: ci_wait<5>(); break;
        case 6: ci_wait<6>(); break;
        default: ci_wait<7>(); break;
    }
}
}  // namespace

struct ConvImgParams {
    const h16_t* in;     // NHWC [B][H][W][Cin]
    const h16_t* wt;     // packed [rows][9][Cin]
    const h16_t* zero;
    float* part;         // fp32 [S][B*H*W][Cout]
    int B, H, W, Cin, Cout, NB, S, wrows;   // wrows: rows of `wt` that exist (>= Cout)
};

__global__ __launch_bounds__(CI_THREADS) void conv_img_kernel(const ConvImgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a15 = lane & 15, q = lane >> 4;
    const unsigned base = (unsigned)(unsigned long long)smem;

    const int Wp = p.W + 2, PP = (p.H + 2) * Wp, HROWS = p.NB * PP, HWi = p.H * p.W;
    const int tiles_n = p.Cout / CI_BN;
    int id = blockIdx.x;
    const int ks = id % p.S; id /= p.S;
    const int nt = id % tiles_n, unit = id / tiles_n;
    const int n0 = nt * CI_BN, b0 = unit * p.NB;
    const int nc_tot = p.Cin >> 5;
    const int c_begin = (int)((long long)nc_tot * ks / p.S), c_end = (int)((long long)nc_tot * (ks + 1) / p.S);
    const int NC = c_end - c_begin, nsteps = 3 * NC;

    // ---- DMA sources ----------------------------------------------------------------------------------------------------------------------------
    // halo: piece = i * 768 + tid -> padded row piece >> 2, physical slot piece & 3; source = that pixel's 8 channels (logical slot), or zeros
    // (LDS-DMA through buffer resources, common.h: blds16 -- with the FLAT-encoded global_load_lds in flight hipcc turns every LDS wait into
    // lgkmcnt(0), which serialised each ds_read with the three MFMAs it feeds; and a lane offset outside the resource simply reads zeros: the
    // zero padding of the maps needs no zero page)
    const buf_rsrc_t in_rs = make_rsrc(p.in, (unsigned)((long long)p.B * HWi * p.Cin * 2));
    const buf_rsrc_t wt_rs = make_rsrc(p.wt, (unsigned)((long long)p.wrows * 9 * p.Cin * 2));
    unsigned h_off[CI_HIT];   // byte offsets
#pragma unroll
    for (int i = 0; i < CI_HIT; ++i) {
        const int piece = i * CI_THREADS + tid, row = piece >> 2, ps = piece & 3;
        const int img = row / PP, rem = row - img * PP;
        const int py = rem / Wp, px = rem - py * Wp;
        const bool ok = row < HROWS && py >= 1 && py <= p.H && px >= 1 && px <= p.W;
        h_off[i] = ok ? (unsigned)((((((b0 + img) * p.H + (py - 1)) * p.W + (px - 1)) * p.Cin) + ((ps ^ ci_swz(row)) << 3)) * 2) : 0xfffff000u;
    }
    // weights: piece tid -> tap_in = tid >> 8, row = (tid & 255) >> 2, physical slot tid & 3
    unsigned w_off;
    {
        const int tap_in = tid >> 8, row = (tid & 255) >> 2, ps = tid & 3;
        w_off = (unsigned)((((n0 + row) * 9 + tap_in) * p.Cin + ((ps ^ ci_swz(row)) << 3)) * 2);
    }
    auto stage_w = [&](int s) __attribute__((always_inline)) {  // step s = (chunk c_begin + s / 3, kernel row s % 3)
        const int c = c_begin + s / 3, ky = s % 3;
        blds16(wt_rs, w_off, (unsigned)((ky * 3 * p.Cin + c * 32) * 2), smem + CI_W_OFF + (s & (CI_NW - 1)) * CI_WSTEP + wave * 1024);
    };
    auto stage_h = [&](int cl) __attribute__((always_inline)) {  // local chunk cl
        const int c = c_begin + cl;
        char* dst = smem + (cl & 1) * CI_HBUF;
#pragma unroll
        for (int i = 0; i < CI_HIT; ++i) {
            const int first = i * CI_THREADS + wave * 64;   // first piece of this wave's instruction
            blds16(in_rs, h_off[i], (unsigned)(c * 64), first < CI_HROWS_MAX * 4 ? dst + first * 16 : smem + CI_DUMP_OFF);
        }
    };

    // ---- fragment addresses ----------------------------------------------------------------------------------------------------------------------
    unsigned xa[3][9];   // pixel fragment j, tap: byte offset inside a halo buffer
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pl = 48 * wave + 16 * j + a15;           // pixel of the unit
        const int img = pl / HWi, rem = pl - img * HWi;
        const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int row = img * PP + (y + t / 3) * Wp + (x + t % 3);
            xa[j][t] = (unsigned)(row * 64 + ((q ^ ci_swz(row)) << 4));
        }
    }
    unsigned wa[4];      // weight fragment i: byte offset inside a step's tile (tap 0); taps 1, 2 are + 4096, + 8192
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * i + a15;
        wa[i] = (unsigned)(row * 64 + ((q ^ ci_swz(row)) << 4));
    }

    f32x4_t acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo of the first chunk, three weight tiles -----------------------------------------------------------------------------------
    if (NC > 0) stage_h(0);
    if (nsteps > 0) stage_w(0);
    if (nsteps > 1) stage_w(1);
    if (nsteps > 2) stage_w(2);

    // one step = (chunk cl, kernel row KY); KY is a compile-time constant so that the 27 pixel-fragment offsets stay in registers (as a
    // run-time select hipcc moved the table to scratch memory: a scratch_load + s_waitcnt vmcnt(0) in front of every pixel-fragment read)
    auto step = [&](auto kyc, int cl) __attribute__((always_inline)) {
        constexpr int ky = decltype(kyc)::value;
        const int s = 3 * cl + ky;
        // my DMA of W(s) (and, in a chunk's first step, of its halo) has landed; younger ones may stay in flight: W(s+1), W(s+2) and, after
        // a chunk's first step, the five halo pieces of the next chunk (issue order: W(s+3), then halo(cl+1), right after the barrier below)
        {
            int nw = nsteps - 1 - s;
            if (nw > 2) nw = 2;
            ci_wait_dyn(nw + ((ky != 0 && cl + 1 < NC) ? CI_HIT : 0));
        }
        __builtin_amdgcn_s_barrier();
        if (s + 3 < nsteps) stage_w(s + 3);               // slot of W(s-1): everybody is past step s-1
        if (ky == 0 && cl + 1 < NC) stage_h(cl + 1);      // buffer of chunk cl-1: last read in step s-1
        const unsigned hb = base + (cl & 1) * CI_HBUF;
        const unsigned wb = base + CI_W_OFF + (s & (CI_NW - 1)) * CI_WSTEP;
        // three taps; the seven fragments of tap kx+1 are requested before tap kx's twelve MFMAs issue (two register sets).  The LDS reads are
        // inline asm with hand-counted lgkmcnt waits: hipcc orders every ds_read it can see behind the buffer-form LDS-DMA still in flight
        // (s_waitcnt vmcnt(2) / (1) / (0) inside this loop: the weight ring drained every step), and with the FLAT-form DMA it waits
        // lgkmcnt(0) before every MFMA batch instead.  What orders these reads against the DMA is the protocol above (counted vmcnt + barrier).
        struct Frag { h16x8_t w[4], x[3]; };
        unsigned hb_v = hb, wb_v = wb;   // (non-const copies: the asm operands of a generic lambda may not name the const locals)
        auto load_frag = [&](Frag& f, auto kxc) __attribute__((always_inline)) {
            constexpr int KX = decltype(kxc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned aw = wb_v + wa[i];
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.w[i]) : "v"(aw), "n"(KX * 4096));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned ax = hb_v + xa[j][3 * ky + KX];
                asm volatile("ds_read_b128 %0, %1" : "=v"(f.x[j]) : "v"(ax));
            }
        };
        auto mfma12 = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[i][j]);
        };
        Frag fa, fb;
        load_frag(fa, IC<0>{});
        load_frag(fb, IC<1>{});
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fa.w[0]), "+v"(fa.w[1]), "+v"(fa.w[2]), "+v"(fa.w[3]), "+v"(fa.x[0]), "+v"(fa.x[1]), "+v"(fa.x[2]));
        __builtin_amdgcn_sched_barrier(0);
        mfma12(fa);
        __builtin_amdgcn_sched_barrier(0);
        load_frag(fa, IC<2>{});
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fb.w[0]), "+v"(fb.w[1]), "+v"(fb.w[2]), "+v"(fb.w[3]), "+v"(fb.x[0]), "+v"(fb.x[1]), "+v"(fb.x[2]));
        __builtin_amdgcn_sched_barrier(0);
        mfma12(fb);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa.w[0]), "+v"(fa.w[1]), "+v"(fa.w[2]), "+v"(fa.w[3]), "+v"(fa.x[0]), "+v"(fa.x[1]), "+v"(fa.x[2]));
        __builtin_amdgcn_sched_barrier(0);
        mfma12(fa);
    };
    for (int cl = 0; cl < NC; ++cl) {
        step(IC<0>{}, cl);
        step(IC<1>{}, cl);
        step(IC<2>{}, cl);
    }

    // ---- fp32 partial sums of this K slice: lane (a15, q) holds channels 16 i + 4 q .. + 3 of pixel 16 j + a15 ---------------------------------------
    const long long M = (long long)p.B * HWi;
    float* dst = p.part + ((long long)ks * M + (long long)unit * CI_PIX) * p.Cout + n0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pl = 48 * wave + 16 * j + a15;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4_t*)(dst + (long long)pl * p.Cout + 16 * i + 4 * q) = acc[i][j];
    }
}

int conv_img_ksplit(const IGemmParams& p);
// images per 576-pixel unit, or 0 when the map does not tile that way
static int conv_img_nb(int B, int H, int W) {
    if (H < 1 || W < 1 || (CI_PIX % (H * W))) return 0;
    const int nb = CI_PIX / (H * W);
    if (B % nb) return 0;
    if (nb * (H + 2) * (W + 2) > CI_HROWS_MAX) return 0;
    return nb;
}

bool conv_img_applicable(const IGemmParams& p) {
    const bool off = gp_sw().no_conv_img;  // A/B switch
    if (off || p.ks != 3 || p.stride != 1 || p.ups || p.pad_t != 1 || p.pad_l != 1 || p.batch > 1 || p.in_scale || p.out_fp32 > 1) return false;
    if ((p.out_fp32 == 1) != (p.res_f32 != 0) && p.res) return false;  // (splitk_reduce_kernel: residual in the output's element type)
    if (p.act == GP_ACT_GEGLU || p.bias_mode == GP_BIAS_ROW || p.Ho != p.Hi || p.Wo != p.Wi) return false;
    if ((p.Cin & 31) || (p.N & 63) || p.N != p.n_store || p.n_store != p.ldo || (p.ldo & 3) || p.Cin < 64) return false;
    return conv_img_nb(p.B, p.Hi, p.Wi) > 0 && conv_img_ksplit(p) >= 2;  // (the partial sums always go through the split-K workspace)
}

int conv_img_ksplit(const IGemmParams& p) {
    const int nb = conv_img_nb(p.B, p.Hi, p.Wi);
    const int tiles = (p.B / nb) * (p.N / CI_BN);
    int S = 256 / tiles;                  // ~ one workgroup per CU
    if (S < 2 && tiles <= 384) S = 2;     // (batch 8 at 24x24: 160 tiles; the partial sums always go through the workspace, so S >= 2)
    const int nc = p.Cin >> 5;
    if (S > nc / 2) S = nc / 2;           // at least two 32-channel chunks (six steps) per slice
    if (S > 16) S = 16;
    const int s_env = gp_sw().conv_img_s;  // tuning switch
    if (s_env > 0 && s_env <= nc / 2 && s_env <= 16) S = s_env;
    return S < 1 ? 1 : S;
}

void launch_conv_img(const IGemmParams& p, float* part, int S, hipStream_t s) {
    const int nb = conv_img_nb(p.B, p.Hi, p.Wi);
    ConvImgParams q{p.in, p.wt, p.zero, part, p.B, p.Hi, p.Wi, p.Cin, p.N, nb, S, p.n_rows};
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] { (void)hipFuncSetAttribute((const void*)conv_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CI_LDS); });
    const int grid = (p.B / nb) * (p.N / CI_BN) * S;
    hipLaunchKernelGGL(conv_img_kernel, dim3(grid), dim3(CI_THREADS), CI_LDS, s, q);
}
