// conv3x3 (stride 1, pad 1; optionally on a nearest-x2-upsampled input) with halo reuse — the kernels for the large VAE /
// UNet feature maps (57 % of the 768x768 pass), where the generic implicit GEMM (igemm.hip) is bound by operand loads: it
// re-fetches every input pixel once per tap.  Here a workgroup owns a 16x16 OUTPUT-PIXEL tile x 128 output channels and, per
// 64-channel chunk, stages the 18x18 input halo (10x10 source pixels in the x2-upsample case) in LDS ONCE; the nine taps then
// read shifted rows of that halo.  Only the weight tiles ([128 cout][64 cin] per tap) stream per K-step through an LDS-DMA ring.
//
//   Waves: 8 = 4 groups of 4 pixel rows x 2 channel halves, each 64 pixels x 64 channels = 16 accumulator tiles of
//        v_mfma_f32_16x16x32_bf16; 1 workgroup per CU.
//   K-step = (chunk, tap): 2 weight DMAs per wave (+ 6 halo DMAs once per chunk), 16 ds_read_b128, 32 MFMAs.
//   Software pipeline: the barrier at the end of step s-1 certifies the operands of step s+1, so the fragments of step s+1 are
//        read from LDS between step s's MFMAs (2 MFMA : 1 ds_read, pinned with sched_group_barrier); LDS latency never sits
//        between a barrier and an MFMA.
//   Sync: counted s_waitcnt vmcnt + ONE raw s_barrier per K-step; waves 4-7 issue their DMA before their MFMAs, waves 0-3
//        after (role split).
//   Optional fused input transform x -> act(x * scale[b][c] + shift[b][c]) (GroupNorm apply + SiLU) on the staged halo of the
//        NEXT chunk; padding stays exactly zero.
//   Two kernels share this tiling: conv3x3_halo3_kernel (persistent: a workgroup walks many tiles, per-wave epilogue; the product
//        path) and conv3x3_halo2_kernel (one tile per workgroup, shared epilogue.h; fallback for ragged channel slots and the
//        A/B partner for measurements).
#include "common.h"
#include "epilogue.h"
#include "kernels.h"

template <int N>
GP_DEV void halo_wait_vm() { wait_vm<N>(); }

constexpr int GN_MAXC = 2560;      // fused input transform: per-channel scale/shift of one image live in 20 KiB of LDS (widest UNet up-block input)
constexpr int HALO_NB = 3;         // weight ring depth
// ABL bit of conv3x3_halo3_kernel that is NOT an ablation: fp32 rows out and an fp32 residual in (IGemmParams::out_fp32 == 1 / res_f32), the
// epilogue of the contract precision (contract.hip: split-bf16 operands over a tripled K; the K loop is the ordinary one)
constexpr int HALO_F32O = 1 << 20;

// TR: output-pixel rows per wave row group (4 row groups per workgroup): 4 = the 16 x 16 tile; 3 = a 12-row x 16-column tile (r5, plain stride-1
// convs only: conv3x3_halo3_kernel<false, 0, 0, 3>, chosen where 16-row tiles quantise badly over the persistent grid, see halo_plan)
template <bool UPS, int TR = 4>
struct HaloGeom {
    static constexpr int HW_ = UPS ? 10 : 18;              // halo width (source pixels)
    static constexpr int HH_ = UPS ? 10 : 4 * TR + 2;      // halo height
    static constexpr int HROWS = HH_ * HW_;                // 100 / 324 (252 for 12-row tiles) halo pixels = LDS rows of 128 B
    static constexpr int GROUPS = (HROWS + 7) / 8;         // 13 / 41 (32) DMA groups of 8 rows
    static constexpr int A_IT = (GROUPS + 7) / 8;          // 2 / 6 (4) DMA instructions per wave per halo (extra ones hit the dump)
    static constexpr int A_BUF = GROUPS * 1024;            // 13 / 41 (32) KiB
    static constexpr int B_OFF = 2 * A_BUF;
    static constexpr int DUMP_OFF = B_OFF + HALO_NB * 16384;
    static constexpr int GN_OFF = DUMP_OFF + 1024;
    static constexpr int LDS_MIN = GN_OFF + 2 * GN_MAXC * 4;
    static constexpr int LDS = LDS_MIN > 131072 ? LDS_MIN : 131072;  // the epilogue stages 128 KiB
};

// ---------------------------------------------------------------------------------------------------------------------------
// One tile per workgroup (conv3x3_halo2_kernel): the fallback for shapes the persistent kernel below does not take (ragged channel
// slots), and the A/B partner for measurements (IGemmParams::dbg bit 256).  The nine taps are unrolled at compile time:
//   * the swizzle key of a halo row depends on its COLUMN hx only, so a lane needs six pixel-fragment base addresses
//     (3 kx x 2 k-halves) and two weight-fragment bases; tap row, pixel row, halo buffer and ring slot (3-deep ring: slot =
//     tap % 3) are ds_read immediates.  (A first version recomputed every ds_read_b128 address per step from run-time tap / chunk
//     / slot: ~80 VALU + ~90 SALU per 32 MFMAs, executed by both waves of a SIMD at once with the matrix pipe idle -- PMC: 2.6 VALU
//     per MFMA, MFMA busy 38 %; removing them was worth 12-30 %.)
//   * the weight-tile source pointers advance by a scalar stride per step (no per-step multiply), and
//   * each MFMA batch is interleaved with the LDS reads of a register set that is dead during that batch.
// swizzle key of halo column hx.  18-wide halo: hx & 7 (any 16 consecutive columns are conflict-free); 10-wide halo of the
// x2-upsample case (lane pairs share a column): table found by exhaustive search, one nibble per column.
template <bool UPS>
GP_DEV int halo_key(int hx) { return UPS ? (int)((0x4016642254ull >> (4 * hx)) & 7) : (hx & 7); }

template <bool UPS>
__global__ __launch_bounds__(512) void conv3x3_halo2_kernel(const IGemmParams p) {
    using G = HaloGeom<UPS>;
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, NW = 8, TN = 64, FM = 4, FN = 4, FP = 2;
    constexpr int HW_ = G::HW_, HROWS = G::HROWS, A_IT = G::A_IT, A_BUF = G::A_BUF;
    constexpr int B_STAGE = BN * 128, B_IT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + G::B_OFF;
    char* const dump = smem + G::DUMP_OFF;
    float* const s_gn = (float*)(smem + G::GN_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool second_half = wave >= NW / 2;
    const int a15 = lane & 15;

    const int Ho = p.Ho, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi, Cin = p.Cin;
    const int tiles_x = (Wo + 15) >> 4, tiles_y = (Ho + 15) >> 4;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + BN - 1) / BN;
    const int sid = xcd_remap(blockIdx.x, tiles_x * tiles_y * p.B * tiles_n);
    const int nt = sid % tiles_n;
    int sp = sid / tiles_n;
    const int tx = sp % tiles_x;
    sp /= tiles_x;
    const int ty = sp % tiles_y, b = sp / tiles_y;
    const int n0 = nt * BN;
    const int cpt = Cin >> 6;

    // ---- DMA sources -----------------------------------------------------------------------------------------------------------
    const int sy0 = UPS ? ty * 8 - 1 : ty * 16 - 1, sx0 = UPS ? tx * 8 - 1 : tx * 16 - 1;
    const h16_t* h_ptr[A_IT];
    unsigned h_ok = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = (wave + NW * i) * 8 + (lane >> 3);
        const int hy = r / HW_, hx = r - hy * HW_;
        const int iy = sy0 + hy, ix = sx0 + hx;
        const bool ok = r < HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        const int chunk = (lane & 7) ^ halo_key<UPS>(hx < HW_ ? hx : 0);
        h_ptr[i] = p.in + (((long long)b * Hi + iy) * Wi + ix) * Cin + chunk * 8;
        if (ok) h_ok |= 1u << i;
    }
    const h16_t* zsrc_a = p.zero;
    const int chunk_w = (lane & 7) ^ (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2));
    // weight rows n0 .. n0+127 always exist (launch_conv_halo checks n_rows); wq[i] walks the (chunk, tap) tiles in issue order
    const h16_t* wq[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) wq[i] = p.wt + (long long)(n0 + (wave + NW * i) * 8 + (lane >> 3)) * p.ldw + chunk_w * 8;
    const int w_step = Cin, w_wrap = 64 - 8 * Cin;  // elements: next tap / first tap of the next chunk

    // ---- fused input transform (GroupNorm apply + SiLU), see the first kernel ------------------------------------------------------
    constexpr int T_IT = (HROWS * 8 + 511) / 512;
    const bool fused = p.in_scale != nullptr;
    unsigned t_ok = 0;
    if (fused) {
        for (int c = tid; c < Cin; c += 512) {
            s_gn[c] = p.in_scale[(long long)b * Cin + c];
            s_gn[GN_MAXC + c] = p.in_shift[(long long)b * Cin + c];
        }
#pragma unroll
        for (int k = 0; k < T_IT; ++k) {
            const int r = (tid + 512 * k) >> 3;
            const int hy = r / HW_, hx = r - hy * HW_;
            if (r < HROWS && (unsigned)(sy0 + hy) < (unsigned)Hi && (unsigned)(sx0 + hx) < (unsigned)Wi) t_ok |= 1u << k;
        }
        __syncthreads();
    }
    const unsigned a_base = (unsigned)(unsigned long long)a_lds, b_base = (unsigned)(unsigned long long)b_lds;
    const unsigned gn_base = (unsigned)(unsigned long long)s_gn;
    auto transform_part = [&](int buf, int cc, int k) {  // part k = 16-byte item tid + 512*k of halo buffer `buf`, channels of chunk cc
        if (!((t_ok >> k) & 1u)) return;
        const int item = tid + 512 * k, r = item >> 3;
        const int ls = ((item & 7) ^ halo_key<UPS>(r % HW_)) << 3;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const unsigned a_item = a_base + buf * A_BUF + item * 16;
        const unsigned a_sc = gn_base + ((cc << 6) + ls) * 4, a_sh = a_sc + GN_MAXC * 4;
        u32x4_t raw;
        f4_t s0, s1, h0, h1;
        asm volatile(
            "ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\tds_read_b128 %3, %7\n\t"
            "ds_read_b128 %4, %7 offset:16\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(raw), "=&v"(s0), "=&v"(s1), "=&v"(h0), "=&v"(h1)
            : "v"(a_item), "v"(a_sc), "v"(a_sh)
            : "memory");
        float v[8] = {h16_lo(raw.x) * s0.x + h0.x, h16_hi(raw.x) * s0.y + h0.y, h16_lo(raw.y) * s0.z + h0.z, h16_hi(raw.y) * s0.w + h0.w,
                      h16_lo(raw.z) * s1.x + h1.x, h16_hi(raw.z) * s1.y + h1.y, h16_lo(raw.w) * s1.z + h1.z, h16_hi(raw.w) * s1.w + h1.w};
        if (p.in_silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        const u32x4_t ov = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
        asm volatile("ds_write_b128 %0, %1" ::"v"(a_item), "v"(ov) : "memory");
    };

    auto stage_halo = [&](int buf, int cc) {
        char* dst = a_lds + buf * A_BUF;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int g = wave + NW * i;
            const h16_t* src = ((h_ok >> i) & 1u) ? h_ptr[i] + (cc << 6) : zsrc_a;
            glds16(src, g < G::GROUPS ? dst + g * 1024 : dump);
        }
    };
    auto stage_w = [&](int slot, bool wrap) {  // next tile in (chunk, tap) order
        char* dst = b_lds + slot * B_STAGE;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            glds16(wq[i], dst + (wave + NW * i) * 1024);
            wq[i] += wrap ? w_wrap : w_step;
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- fragment base addresses (LDS byte offsets) ----------------------------------------------------------------------------------
    struct Half { h16x8_t w[FN], x[FM]; };  // fragments of one k-half (32 channels) of one step
    unsigned xb[3][2], wb[2];
    {
        const int xr_w = ((a15 >> 1) & 1) | (((a15 >> 2) & 1) << 1) | (((a15 >> 3) & 1) << 2);
        const int w_row_off = (wn * TN + 8 * (a15 >> 2) + (a15 & 3)) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 4 + (lane >> 4);
            wb[kk] = b_base + w_row_off + ((sl ^ xr_w) << 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int hx = UPS ? ((a15 + kx - 1) >> 1) + 1 : a15 + kx;
                const int hy0 = UPS ? 2 * wm : 4 * wm;
                xb[kx][kk] = a_base + (hy0 * HW_ + hx) * 128 + ((sl ^ halo_key<UPS>(hx)) << 4);
            }
        }
    }
    // k-half KK of step (TAP, halo buffer PAR): every offset below is an immediate
    auto load_half = [&](Half& f, auto tapc, auto parc, auto kkc) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, KK = decltype(kkc)::value;
        constexpr int KY = TAP / 3, KX = TAP % 3, SLOT = TAP % 3;
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = lds_frag(wb[KK], SLOT * B_STAGE + (i >> 1) * 4096 + (i & 1) * 512);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int hy = UPS ? ((j + KY - 1) >> 1) + 1 : j + KY;  // relative to the wave's first halo row
            f.x[j] = lds_frag(xb[KX][KK], PAR * A_BUF + hy * HW_ * 128);
        }
    };
    auto mfma16 = [&](const Half& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[i][j]);
    };
    // 16 MFMAs interleaved with 8 LDS reads (2 : 1), pinned
    auto interleave = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------------------
    float bcol[FP][8];
    load_bias_cols<FP>(p, 0, n0 + wn * TN, 8 * (lane >> 4), bcol);
    stage_halo(0, 0);
    stage_w(0, false);
    stage_w(1, false);
    stage_w(2, false);
    Half f0, f1a, f1b;  // f0: k-half 0 of the current step; f1a / f1b ping-pong: k-half 1 of the current / next step
    halo_wait_vm<B_IT>();  // halo 0 and the tiles of taps 0, 1 have landed
    __builtin_amdgcn_s_barrier();
    if (fused) {
#pragma unroll
        for (int k = 0; k < T_IT; ++k) transform_part(0, 0, k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    load_half(f0, IC<0>{}, IC<0>{}, IC<0>{});
    load_half(f1a, IC<0>{}, IC<0>{}, IC<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody holds its step-0 fragments: ring slot 0 may be refilled (3-deep ring)

    // ---- main loop -------------------------------------------------------------------------------------------------------------------
    // Invariant at the top of step s = (cc, TAP): the barrier that certified the operands of step s+1 has been passed, f0 / cur1
    // hold both k-halves of step s, weight tiles up to step s+2 are issued.  The step issues tile s+3 into slot TAP % 3 (its
    // previous content, tile s, was read during step s-1), at tap 0 the halo of chunk cc+1, and reads the fragments of step s+1.
    int cc = 0;
    bool last = cpt == 1;
    auto kstep = [&](auto tapc, auto parc, Half& cur1, Half& nxt1) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value;
        constexpr int TAP1 = (TAP + 1) % 9, PAR1 = TAP == 8 ? PAR ^ 1 : PAR;
        const bool issue_w = !(last && TAP >= 6), issue_h = TAP == 0 && !last;
        if (second_half) {
            if (issue_w) stage_w(TAP % 3, (TAP + 3) % 9 == 8);
            if (issue_h) stage_halo(PAR ^ 1, cc + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAP < 8) {
            load_half(nxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
            mfma16(f0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            load_half(f0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
            mfma16(cur1);
            interleave();
        } else {  // chunk boundary: the next step exists only if another chunk follows (single MFMA site either way)
            mfma16(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (!last) {
                load_half(f0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
                load_half(nxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma16(cur1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fused && !last) {  // the halo of chunk cc+1 is complete for everybody from the barrier before tap 3 on; first read in tap 8
            if (T_IT == 6) {
                if (TAP == 3) { transform_part(PAR ^ 1, cc + 1, 0); transform_part(PAR ^ 1, cc + 1, 1); }
                else if (TAP >= 4 && TAP <= 7) transform_part(PAR ^ 1, cc + 1, TAP - 2);
            } else {
                if (TAP == 3) transform_part(PAR ^ 1, cc + 1, 0);
                else if (TAP == 4) transform_part(PAR ^ 1, cc + 1, 1);
            }
        }
        if (!second_half) {
            if (issue_w) stage_w(TAP % 3, (TAP + 3) % 9 == 8);
            if (issue_h) stage_halo(PAR ^ 1, cc + 1);
        }
        // barrier(s+1): tile s+2 (and every halo issued before it) must have landed; tile s+3 and, while it was issued in
        // tap 0 of this chunk (i.e. after tile s+2 or s+3 ...), the halo of chunk cc+1 may stay in flight.
        if (TAP == 8 && last) return;
        if (TAP <= 1) { if (!last) halo_wait_vm<A_IT + B_IT>(); else halo_wait_vm<B_IT>(); }
        else if (TAP < 6) halo_wait_vm<B_IT>();
        else { if (last) halo_wait_vm<0>(); else halo_wait_vm<B_IT>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // next step's fragments (and my transform writes) are done
        __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](auto parc, Half& fa, Half& fb) {
        kstep(IC<0>{}, parc, fa, fb); kstep(IC<1>{}, parc, fb, fa); kstep(IC<2>{}, parc, fa, fb);
        kstep(IC<3>{}, parc, fb, fa); kstep(IC<4>{}, parc, fa, fb); kstep(IC<5>{}, parc, fb, fa);
        kstep(IC<6>{}, parc, fa, fb); kstep(IC<7>{}, parc, fb, fa); kstep(IC<8>{}, parc, fa, fb);
        ++cc;
        last = cc + 1 == cpt;
    };
    while (true) {
        chunk(IC<0>{}, f1a, f1b);
        if (cc == cpt) break;
        chunk(IC<1>{}, f1b, f1a);
        if (cc == cpt) break;
    }

    conv_epilogue<BM, BN, WM, WN, 512>(p, acc, bcol, n0, 0, wave, lane, smem, [&](int pr) {
        const int oy = ty * 16 + (pr >> 4), ox = tx * 16 + (pr & 15);
        return (oy < Ho && ox < Wo) ? (b * Ho + oy) * Wo + ox : -1;
    }, (b * tiles_y + ty) * tiles_x + tx);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent variant: one workgroup per CU walks a list of spatial tiles of ONE image and ONE 128-channel slice, so the (tile,
// chunk, tap) step stream never stops: the first halo and the first three weight tiles of the next tile are fetched during the
// last chunk of the current one (they are simply "chunk cc+1"), and the output stores of a tile drain while the next tile
// computes.  In the one-tile-per-workgroup kernels the prologue (launch, address setup, first HBM round trip) and the epilogue
// (LDS staging, store drain) were serial with the K loop and with each other: 57 % of the time of a 128->128 layer at 768^2
// (K = 1152; measured by ablation), which at 604 MB written + 756 MB read per call is also HBM-relevant, not only MFMA work.
//   * bias, GroupNorm scale/shift and the weight slice are per-workgroup constants (grid = B x J, J a multiple of tiles_n);
//   * the epilogue is per WAVE: four passes of [16 px][64 ch] fp32 through a private 4 KiB LDS window (the halo buffer the
//     finished chunk just released), no workgroup barrier; residual rows are prefetched before the first pass;
//   * GroupNorm partial statistics: per-wave sums -> 4 KiB LDS -> combined after the step barrier that follows.
// PH (r5): the x2-nearest-upsample conv as FOUR PHASE convolutions on the source map.  Output pixel (2y + a, 2x + b) of conv3x3(upsample2(s)) reads the
// source pixels {y - 1 + a, y + a} x {x - 1 + b, x + b} only -- the rows 2y + a - 1 ... 2y + a + 1 of the upsampled map fall onto two source rows -- with the
// kernel rows that land on the same source row SUMMED: phase a = 0: {w[0]}, {w[1] + w[2]}; a = 1: {w[0] + w[1]}, {w[2]} (same along x).  That is a 2 x 2-tap
// convolution per phase: 4 MFMA steps per 64-channel chunk instead of 9 for the same outputs (4/9 of the flops; the zero padding of the UPSAMPLED map
// is the zero padding of the source map).  In the 18 x 18 source halo of a 16 x 16 block of source positions phase (a, b) reads taps (a + ty, b + tx)
// of the plain kernel's 3 x 3 tap grid, so the variant is the plain kernel with: units = (tile, phase), four taps, fragment bases shifted by
// (a, b), a 4-deep weight ring (slot = tap) over [Cout][phase][2 x 2][Cin] weights summed once at load (engine: pack_phases), and an epilogue that
// writes output pixel (2y + a, 2x + b).
template <bool UPS, int TR = 4, bool PH = false>
struct Halo3Geom {
    static constexpr int A_BUF = HaloGeom<UPS, TR>::A_BUF;
    static constexpr int B_OFF = 2 * A_BUF;
    static constexpr int NBR = PH ? 4 : 3;                   // weight ring depth
    static constexpr int DUMP_OFF = B_OFF + NBR * 16384;
    static constexpr int GN_OFF = DUMP_OFF + 1024;
    static constexpr int ST_OFF = GN_OFF + (PH ? 0 : 2 * GN_MAXC * 4);  // [8 waves][64 ch][sum, sumsq] (PH: never fused, no scale / shift table)
    static constexpr int BIAS_OFF = ST_OFF + 4096;           // [128] bias of the workgroup's channel slice
    static constexpr int EP_OFF = BIAS_OFF + 512;            // x2-upsample geometry: its halo buffers are smaller than 32 KiB
    static constexpr int LDS = UPS ? EP_OFF + 32768 : EP_OFF;
};

// FUSED: 0 plain input, 1 input transform x * scale[b][c] + shift[b][c] (GroupNorm apply), 2 the same followed by SiLU.
// ABL: compile-time ablations for profiling (2 no MFMA, 4 no output stores, 8 no halo DMA, 16 no weight DMA, 32 no waits for the DMA,
// 64 / 128 every wave issues its DMA before / after its MFMAs, 256 every other step barrier, 512 no step barrier, 1024 no epilogue; >= 2 except 64 / 128: garbage results)
template <bool UPS, int FUSED, int ABL = 0, int TR = 4, bool PH = false>
__global__ __launch_bounds__(512) void conv3x3_halo3_kernel(const IGemmParams p) {
    using G = HaloGeom<UPS, TR>;
    using G3 = Halo3Geom<UPS, TR, PH>;
    static_assert(TR == 4 || (TR == 3 && !UPS && FUSED == 0), "12-row tiles: plain stride-1 convs only");
    static_assert(!PH || (!UPS && FUSED == 0 && TR == 4 && (ABL & ~HALO_F32O) == 0), "phase mode: its own geometry");
    constexpr bool F32O = (ABL & HALO_F32O) != 0;
    static_assert(!F32O || FUSED == 0, "fp32 rows out: plain input only");
    constexpr int NT = PH ? 4 : 9;        // taps (K-steps) per 64-channel chunk
    constexpr int NBR = G3::NBR;          // weight ring depth
    constexpr int BN = 128, NW = 8, TN = 64, FM = TR, FN = 4, FP = 2;
    constexpr int TH = 4 * TR;  // output rows of a tile (16 columns always)
    constexpr int HW_ = G::HW_, HROWS = G::HROWS, A_IT = G::A_IT, A_BUF = G::A_BUF;
    constexpr int B_STAGE = BN * 128, B_IT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + G3::B_OFF;
    char* const dump = smem + G3::DUMP_OFF;
    float* const s_gn = (float*)(smem + G3::GN_OFF);
    float* const s_st = (float*)(smem + G3::ST_OFF);
    float* const s_bias = (float*)(smem + G3::BIAS_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool second_half = wave >= NW / 2;
    const int a15 = lane & 15;

    const int Ho = p.Ho, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi, Cin = p.Cin;
    // PH: tiles are 16 x 16 blocks of SOURCE positions, a unit of work is (tile, phase): tiles_sp counts units
    const int tiles_x = PH ? (Wi + 15) >> 4 : (Wo + 15) >> 4, tiles_y = PH ? (Hi + 15) >> 4 : (Ho + TH - 1) / TH;
    const int tiles_sp = tiles_x * tiles_y * (PH ? 4 : 1);
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + BN - 1) / BN;
    const int J = gridDim.x / p.B;                  // workgroups per image, a multiple of tiles_n
    const int b = blockIdx.x / J;
    int jw = blockIdx.x - b * J;
    if ((J & 7) == 0) jw = (jw & 7) * (J >> 3) + (jw >> 3);  // workgroups of one XCD (id % 8) take neighbouring tiles
    const int nt = jw % tiles_n, sp_stride = J / tiles_n;
    int sp_cur = jw / tiles_n;                       // spatial tile being computed
    const int n0 = nt * BN;
    const int cpt = Cin >> 6;
    const h16_t* const in_b = p.in + (long long)b * Hi * Wi * Cin;

    // ---- fetch state: the tile whose halo is being staged / normalised (one chunk ahead of the compute) ----------------------------
    constexpr int T_IT = (HROWS * 8 + 511) / 512;
    constexpr bool fused = FUSED != 0;
    // r4: LDS-DMA through buffer resources (common.h blds16: buffer_load_dwordx4 ... offen lds) -- per DMA instruction a 32-bit lane offset and a
    // uniform offset instead of a 64-bit pointer per lane (no v_lshl_add_u64 / select per piece), and zero padding comes from the range check
    // instead of a pointer to a zero page.  In the isolated loop (tools/mfma_lds_probe.py, modes 3 / 11 and 7 / 15) the MUBUF form is 1.4-2.8 %
    // faster than global_load_lds.  ABL & 65536 (GP_HALO_ABLATIONS builds): the FLAT form of r2 / r3.
    constexpr bool MUBUF = !(ABL & 65536);
    const buf_rsrc_t in_rs = make_rsrc(in_b, (unsigned)Hi * (unsigned)Wi * (unsigned)Cin * 2u);
    const buf_rsrc_t w_rs = make_rsrc(p.wt, (unsigned)p.n_rows * (unsigned)p.ldw * 2u);
    int h_off[A_IT];
    unsigned h_ok = 0, t_ok = 0;
    auto setup_fetch = [&](int sp) __attribute__((always_inline)) {
        if (PH) sp >>= 2;
        const int fty = sp / tiles_x, ftx = sp - fty * tiles_x;
        const int sy0 = UPS ? fty * 8 - 1 : fty * TH - 1, sx0 = UPS ? ftx * 8 - 1 : ftx * 16 - 1;
        h_ok = 0;
        t_ok = 0;
        int lane_o = lane, tid_o = tid;
        asm volatile("" : "+v"(lane_o), "+v"(tid_o));  // opaque (see epilogue)
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = (wave + NW * i) * 8 + (lane_o >> 3);
            const int hy = r / HW_, hx = r - hy * HW_;
            const int iy = sy0 + hy, ix = sx0 + hx;
            const bool ok = r < HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            h_off[i] = (iy * Wi + ix) * Cin + (((lane_o & 7) ^ halo_key<UPS>(hx)) << 3);
            if (MUBUF) h_off[i] = ok ? (int)((unsigned)h_off[i] * 2u) : (int)0xfffffff0u;  // byte offset; past the image: the buffer load returns zeros (the padding)
            if (ok) h_ok |= 1u << i;
        }
        if (fused) {
#pragma unroll
            for (int k = 0; k < T_IT; ++k) {
                const int r = (tid_o + 512 * k) >> 3;
                const int hy = r / HW_, hx = r - hy * HW_;
                if (r < HROWS && (unsigned)(sy0 + hy) < (unsigned)Hi && (unsigned)(sx0 + hx) < (unsigned)Wi) t_ok |= 1u << k;
            }
        }
    };
    const h16_t* zsrc_a = p.zero;
    const int chunk_w = (lane & 7) ^ (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2));
    const h16_t* wq[B_IT];
    unsigned w_lane[B_IT], w_uni = 0;  // MUBUF: byte offset of this lane's (row, 16-byte slot) and the running (tile, chunk, tap) offset, wave-uniform
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        wq[i] = p.wt + (long long)(n0 + (wave + NW * i) * 8 + (lane >> 3)) * p.ldw + chunk_w * 8;
        w_lane[i] = (unsigned)(((n0 + (wave + NW * i) * 8 + (lane >> 3)) * p.ldw + chunk_w * 8) * 2);
    }
    const int w_step = Cin, w_wrap = 64 - (NT - 1) * Cin, w_tile_wrap = -(NT - 1) * Cin - (cpt - 1) * 64;  // next tap / next chunk / first tile again

    if (fused) {
        for (int c = tid; c < Cin; c += 512) {
            s_gn[c] = p.in_scale[(long long)b * Cin + c];
            s_gn[GN_MAXC + c] = p.in_shift[(long long)b * Cin + c];
        }
        __syncthreads();
    }
    const unsigned a_base = (unsigned)(unsigned long long)a_lds, b_base = (unsigned)(unsigned long long)b_lds;
    const unsigned gn_base = (unsigned)(unsigned long long)s_gn;
    // Input transform of one 16-byte item (8 channels of one halo pixel) per thread and part: part k = item tid + 512 * k of halo buffer
    // `buf`, channels of chunk cc.  Split in a load half and a finish half so that the K-step can put the LDS reads among the MFMAs of its
    // first batch and the arithmetic (8 FMA + 8 SiLU with two quarter-rate transcendentals each + packing) among those of the second: run
    // as a block after the MFMAs -- both waves of a SIMD at the same time -- it left the matrix pipe idle and cost ~20 % per conv.
    // No branches: items beyond the halo go to the dump KiB, items outside the image (zero padding of the NORMALISED tensor) keep
    // their zeros through a select.  LDS accesses by integer address (see common.h).
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_t* lds_u4_ptr;
    struct TPart { u32x4_t raw; f32x2_t v[4]; unsigned addr; bool ok; };
    const unsigned dump_base = (unsigned)(unsigned long long)dump;
    // A thread always handles the same LOGICAL channel slot c = tid & 7 of its halo rows (physical 16-byte slot c ^ key(column)), so scale / shift
    // of its eight channels are read once per chunk into registers (tbl_load) instead of four 16-byte LDS reads per item.
    f32x4_t gs0, gs1, gh0, gh1;
    auto tbl_load = [&](int cc) __attribute__((always_inline)) {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const unsigned a_sc = gn_base + ((cc << 6) + ((tid_o & 7) << 3)) * 4;
        gs0 = *(lds_f4_ptr)a_sc;
        gs1 = *(lds_f4_ptr)(a_sc + 16);
        gh0 = *(lds_f4_ptr)(a_sc + GN_MAXC * 4);
        gh1 = *(lds_f4_ptr)(a_sc + GN_MAXC * 4 + 16);
    };
    // the column of the thread's first halo row stays in a register: the column of row r0 + 64 k needs no division that way
    const unsigned tp_c0 = (unsigned)((tid >> 3) % HW_);
    auto tp_load = [&](TPart& t, int buf, int k) __attribute__((always_inline)) {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));  // opaque: recompute the address values here instead of carrying them through the loop
        const bool inr = k < (HROWS * 8) / 512 ? true : tid_o + 512 * k < HROWS * 8;  // (only the last, partial part can leave the halo)
        const unsigned tp_a0 = a_base + (unsigned)(tid_o >> 3) * 128u;
        unsigned col = tp_c0 + (unsigned)((64 * k) % HW_);
        col = min(col, col - (unsigned)HW_);                                         // mod HW_ (the wrapped difference is huge when col < HW_)
        const unsigned ps = (unsigned)(tid_o & 7) ^ (unsigned)halo_key<UPS>((int)col);
        t.ok = (t_ok >> k) & 1u;
        t.addr = inr ? tp_a0 + (unsigned)(buf * A_BUF + k * 8192) + (ps << 4) : dump_base + (unsigned)(tid_o & 63) * 16u;
        t.raw = *(lds_u4_ptr)t.addr;
    };
    // arithmetic in two halves so that it can sit under BOTH MFMA batches of the step (one VALU pipe per SIMD, two waves on it: with
    // all of it under the second batch that batch was VALU-bound while the first one had idle VALU slots).
    // r5: two channels per instruction -- v_pk_fma_f32 for x * scale + shift, v_pk_mul_f32 / v_pk_add_f32 around the two transcendentals of the
    // SiLU (x * rcp(1 + exp2(-x log2 e))): per 8-channel item 16 packed + 8 unpack + 8 pack / select instructions and 16 quarter-rate
    // transcendentals instead of 48 + 16 (the transform's VALU time, which the two waves of a SIMD spend with the matrix pipe half idle, -14 %)
    auto silu_pk = [&](f32x2_t v) __attribute__((always_inline)) {
        const f32x2_t nl2e = {-1.44269504088896340736f, -1.44269504088896340736f}, one = {1.f, 1.f};
        f32x2_t e = v * nl2e;
        e.x = __builtin_amdgcn_exp2f(e.x);
        e.y = __builtin_amdgcn_exp2f(e.y);
        e = e + one;
        e.x = __builtin_amdgcn_rcpf(e.x);
        e.y = __builtin_amdgcn_rcpf(e.y);
        return v * e;
    };
    auto tp_mid = [&](TPart& t) __attribute__((always_inline)) {
        const f32x2_t x0 = {h16_lo(t.raw.x), h16_hi(t.raw.x)}, x1 = {h16_lo(t.raw.y), h16_hi(t.raw.y)};
        const f32x2_t x2 = {h16_lo(t.raw.z), h16_hi(t.raw.z)}, x3 = {h16_lo(t.raw.w), h16_hi(t.raw.w)};
        t.v[0] = __builtin_elementwise_fma(x0, f32x2_t{gs0.x, gs0.y}, f32x2_t{gh0.x, gh0.y});
        t.v[1] = __builtin_elementwise_fma(x1, f32x2_t{gs0.z, gs0.w}, f32x2_t{gh0.z, gh0.w});
        t.v[2] = __builtin_elementwise_fma(x2, f32x2_t{gs1.x, gs1.y}, f32x2_t{gh1.x, gh1.y});
        t.v[3] = __builtin_elementwise_fma(x3, f32x2_t{gs1.z, gs1.w}, f32x2_t{gh1.z, gh1.w});
        if (FUSED == 2) {
            t.v[0] = silu_pk(t.v[0]);
            t.v[1] = silu_pk(t.v[1]);
        }
    };
    auto tp_finish = [&](TPart& t) __attribute__((always_inline)) {
        if (FUSED == 2) {
            t.v[2] = silu_pk(t.v[2]);
            t.v[3] = silu_pk(t.v[3]);
        }
        // (normalised activations: |x| stays within a few tens, no saturation needed in the fp16 build)
        u32x4_t ov = {pack_h16x2_ns(t.v[0].x, t.v[0].y), pack_h16x2_ns(t.v[1].x, t.v[1].y), pack_h16x2_ns(t.v[2].x, t.v[2].y), pack_h16x2_ns(t.v[3].x, t.v[3].y)};
        ov = t.ok ? ov : t.raw;
        *(lds_u4_ptr)t.addr = ov;
    };
    auto stage_halo = [&](int buf, int cc) __attribute__((always_inline)) {
        char* dst = a_lds + buf * A_BUF;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int g = wave + NW * i;
            if (MUBUF) {
                if (!(ABL & 8)) blds16(in_rs, (unsigned)h_off[i], (unsigned)(cc << 7), g < G::GROUPS ? dst + g * 1024 : dump);
            } else {
                const h16_t* src = ((h_ok >> i) & 1u) ? in_b + (h_off[i] + (cc << 6)) : zsrc_a;
                if (!(ABL & 8)) glds16(src, g < G::GROUPS ? dst + g * 1024 : dump);
            }
        }
    };
    auto stage_w = [&](int slot, int adv) __attribute__((always_inline)) {  // next weight tile in (tile, chunk, tap) order, then advance by `adv` elements
        char* dst = b_lds + slot * B_STAGE;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if (MUBUF) {
                if (!(ABL & 16)) blds16(w_rs, w_lane[i], w_uni, dst + (wave + NW * i) * 1024);
            } else {
                if (!(ABL & 16)) glds16(wq[i], dst + (wave + NW * i) * 1024);
                wq[i] += adv;
            }
        }
        if (MUBUF) w_uni += (unsigned)(adv * 2);
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    struct Half { h16x8_t w[FN], x[FM]; };
    unsigned xb[3][2], wb[2];
    {
        const int xr_w = ((a15 >> 1) & 1) | (((a15 >> 2) & 1) << 1) | (((a15 >> 3) & 1) << 2);
        const int w_row_off = (wn * TN + 8 * (a15 >> 2) + (a15 & 3)) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 4 + (lane >> 4);
            wb[kk] = b_base + w_row_off + ((sl ^ xr_w) << 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int hx = UPS ? ((a15 + kx - 1) >> 1) + 1 : a15 + kx;
                const int hy0 = UPS ? 2 * wm : TR * wm;
                xb[kx][kk] = a_base + (hy0 * HW_ + hx) * 128 + ((sl ^ halo_key<UPS>(hx)) << 4);
            }
        }
    }
    // PH: pixel-fragment bases of the unit whose fragments are being read: phase (a, b) shifts the plain kernel's bases by a halo rows and b columns
    // (the swizzle key follows the column, so the column shift is a choice between the precomputed bases of kx and kx + 1)
    unsigned xs[2][2];
    auto set_phase_bases = [&](int unit) __attribute__((always_inline)) {
        const int a = (unit >> 1) & 1, bb = unit & 1;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[tx][kk] = (bb ? xb[tx + 1][kk] : xb[tx][kk]) + (unsigned)(a * HW_ * 128);
    };
    auto load_half = [&](Half& f, auto tapc, auto parc, auto kkc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, KK = decltype(kkc)::value;
        constexpr int KY = PH ? TAP / 2 : TAP / 3, KX = PH ? TAP % 2 : TAP % 3, SLOT = PH ? TAP : TAP % 3;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            if (ABL & 8192) asm volatile("" : "+v"(f.w[i]));  // r4 diagnostic: no weight-fragment reads (registers keep whatever they hold: garbage results)
            else f.w[i] = lds_frag(wb[KK], SLOT * B_STAGE + (i >> 1) * 4096 + (i & 1) * 512);
            typedef const volatile __attribute__((address_space(3))) h16x8_t* lds_frag_vptr;
            if (ABL & 32768) {  // r4 diagnostic: every fragment read issued TWICE (the copy is discarded): is LDS array time additive to the MFMA time?
                const h16x8_t t = *(lds_frag_vptr)(wb[KK] + (unsigned)(SLOT * B_STAGE + (i >> 1) * 4096 + (i & 1) * 512));
                asm volatile("" ::"v"(t));
            }
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int hy = UPS ? ((j + KY - 1) >> 1) + 1 : j + KY;
            if (ABL & 16384) asm volatile("" : "+v"(f.x[j]));  // ... no pixel-fragment reads
            else f.x[j] = lds_frag(PH ? xs[KX][KK] : xb[KX][KK], PAR * A_BUF + hy * HW_ * 128);
            typedef const volatile __attribute__((address_space(3))) h16x8_t* lds_frag_vptr;
            if (ABL & 32768) {
                const h16x8_t t = *(lds_frag_vptr)(xb[KX][KK] + (unsigned)(PAR * A_BUF + hy * HW_ * 128));
                asm volatile("" ::"v"(t));
            }
        }
    };
    auto mfma16 = [&](const Half& f) __attribute__((always_inline)) {
        if (ABL & 2) {  // keep the fragment reads alive
            asm volatile("" ::"v"(f.w[0]), "v"(f.w[1]), "v"(f.w[2]), "v"(f.w[3]), "v"(f.x[0]), "v"(f.x[1]), "v"(f.x[2]), "v"(f.x[3]));
            return;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[i][j]);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
        if (ABL & 2048) {  // r4 experiment: the eight fragment reads under the FIRST eight MFMAs (the last one then has eight MFMAs of cover
                           // before the step's lgkmcnt(0) + barrier instead of none)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            return;
        }
        if (FM == 3) {  // 12 MFMAs, 7 fragment reads (4 weight + 3 pixel)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };

    // bias of this workgroup's 128 channels: LDS (512 B in the dump KiB's tail is not free -> own slot after the statistics)
    if (tid < BN) s_bias[tid] = (p.bias && p.bias_mode == GP_BIAS_COL && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
    __syncthreads();

    // ---- per-wave epilogue of the finished tile (sp_cur); `stg` = 4 KiB private LDS window ---------------------------------------------
    const int n_out = p.N;
    const bool want_stats = p.stats_out != nullptr;
    // All LDS traffic below goes through integer-addressed address_space(3) accesses (like lds_frag): accesses the compiler can
    // trace back to `smem` make it wait for vmcnt(0) first (possible alias with an LDS-DMA in flight), i.e. for the residual
    // loads just issued and for the previous pass's stores -- four exposed HBM round trips per tile when measured.
    // Launch-time guarantees (launch_conv_halo): n_store, ldo (and ldres) multiples of 8, so every 8-channel slot is stored whole.
    const unsigned st_base = (unsigned)(unsigned long long)s_st, bias_base = (unsigned)(unsigned long long)s_bias;
    // Specialised at compile time on (activation present, residual present, statistics wanted): with run-time checks per element
    // the epilogue was ~2000 VALU + 130 scalar branches per wave and tile, as much SIMD time as the 18 K-steps of a K = 1152 layer.
    // the accumulators of a tile start at the bias of their channels (rows of fragment i: channels wn * 64 + 32 * (i / 2) + 8 * q + 4 * (i & 1) ..+3)
    auto acc_init = [&]() __attribute__((always_inline)) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const unsigned ba = bias_base + (wn * TN + 8 * (lane_o >> 4)) * 4;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const f32x4_t bv = *(lds_f4_ptr)(ba + (32 * (i >> 1) + 4 * (i & 1)) * 4);
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = bv;
        }
    };
    auto epilogue_body = [&](unsigned stg, auto actc, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr bool ACT = decltype(actc)::value != 0, RES = decltype(resc)::value != 0, STATS = decltype(statc)::value != 0;
        const int tile_i = PH ? sp_cur >> 2 : sp_cur, ph_a = (sp_cur >> 1) & 1, ph_b = sp_cur & 1;  // (PH: unit = 4 * tile + 2 a + b)
        const int ty = tile_i / tiles_x, tx = tile_i - ty * tiles_x;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque copy: keeps the address arithmetic below inside the epilogue (hoisted out of the
                                          // tile loop it occupied ~40 VGPRs for the whole K loop and pushed the kernel into spills)
        const int q = lane_o >> 4, a15 = lane_o & 15;
        const int pl = lane_o >> 3, sl8 = lane_o & 7;      // read-back role: pixels pl and pl + 8 of a tile row, channel slot sl8
        const int col = n0 + wn * TN + 8 * sl8;
        const bool col_ok = col < p.n_store;
        // slot reaches into the zero-padded channels: masks for the packed words (4 VALU per stored slot instead of 16 selects)
        unsigned tmask[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) tmask[w] = (col + 2 * w < n_out ? 0xffffu : 0u) | (col + 2 * w + 1 < n_out ? 0xffff0000u : 0u);
        h16_t* outp = (h16_t*)p.out;
        float* const outf = (float*)p.out;             // F32O
        const float* const resf = (const float*)p.res;  // F32O: the residual is an fp32 tensor
        int m2[FM][2];
        uint4 rv[FM][2];
        f32x4_t rf[2][2];  // F32O: fp32 residual of the NEXT tile row only (prefetched one row ahead: all FM rows would be 64 VGPRs)
        auto res_f32_load = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rf[h][0] = rf[h][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (m2[j][h] >= 0 && p.res) {
                    const float* rp = resf + (long long)m2[j][h] * p.ldres + col;
                    rf[h][0] = *(const f32x4_t*)rp;
                    rf[h][1] = *(const f32x4_t*)(rp + 4);
                }
            }
        };
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // residual rows first: their latency hides under the LDS round trips
                int oy = ty * TH + TR * wm + j, ox = tx * 16 + pl + 8 * h;
                if (PH) { oy = 2 * oy + ph_a; ox = 2 * ox + ph_b; }  // source position (y, x) of phase (a, b) -> output pixel (2 y + a, 2 x + b); Ho = 2 Hi
                m2[j][h] = (oy < Ho && ox < Wo && col_ok) ? (b * Ho + oy) * Wo + ox : -1;
                if (RES && !F32O) {
                    rv[j][h] = make_uint4(0u, 0u, 0u, 0u);
                    if (m2[j][h] >= 0 && p.res) rv[j][h] = *(const uint4*)(p.res + (long long)m2[j][h] * p.ldres + col);
                }
            }
        if (RES && F32O) res_f32_load(0);
        float st_s[8], st_q[8];
        float satm = 0.f;  // fp16 build: max |value| this thread packs in this tile (common.h: sat_track / sat_report)
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
#pragma unroll
            for (int ip = 0; ip < FP; ++ip) {
                const unsigned d = stg + (a15 * 64 + (((4 * ip + q) ^ (a15 & 7)) << 3)) * 4;
                *(lds_f4_ptr)d = acc[2 * ip][j];
                *(lds_f4_ptr)(d + 16) = acc[2 * ip + 1][j];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pr = pl + 8 * h;
                const unsigned sa = stg + (pr * 64 + ((sl8 ^ (pr & 7)) << 3)) * 4;
                const f32x4_t x0 = *(lds_f4_ptr)sa, x1 = *(lds_f4_ptr)(sa + 16);
                const long long m = m2[j][h];
                if (F32O) {
                    if (m >= 0) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        if (RES) {
                            v[0] += rf[h][0].x; v[1] += rf[h][0].y; v[2] += rf[h][0].z; v[3] += rf[h][0].w;
                            v[4] += rf[h][1].x; v[5] += rf[h][1].y; v[6] += rf[h][1].z; v[7] += rf[h][1].w;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (ACT) {
                                if (p.act == GP_ACT_SILU) v[e] = silu_f(v[e]);
                                else if (p.act == GP_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                            }
                            if (col + e >= n_out) v[e] = 0.f;  // zero-padded channels of the last slot
                        }
                        float* o = outf + m * p.ldo + col;
                        *(f32x4_t*)o = f32x4_t{v[0], v[1], v[2], v[3]};
                        *(f32x4_t*)(o + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                        if (STATS) {  // the stored values are the fp32 values themselves
#pragma unroll
                            for (int e = 0; e < 8; ++e) { st_s[e] += v[e]; st_q[e] += v[e] * v[e]; }
                        }
                    }
                    if (RES && h == 1 && j + 1 < FM) res_f32_load(j + 1);  // (both pixels of row j have consumed rf)
                    continue;
                }
                if (m >= 0) {
                    float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if (RES) {
                        const uint4 r4 = rv[j][h];
                        v[0] += h16_lo(r4.x); v[1] += h16_hi(r4.x); v[2] += h16_lo(r4.y); v[3] += h16_hi(r4.y);
                        v[4] += h16_lo(r4.z); v[5] += h16_hi(r4.z); v[6] += h16_lo(r4.w); v[7] += h16_hi(r4.w);
                    }
                    if (ACT) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (p.act == GP_ACT_SILU) v[e] = silu_f(v[e]);
                            else if (p.act == GP_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                        }
                    }
                    uint4 pk;
                    pk.x = pack_h16x2_t(v[0], v[1], satm) & tmask[0]; pk.y = pack_h16x2_t(v[2], v[3], satm) & tmask[1];
                    pk.z = pack_h16x2_t(v[4], v[5], satm) & tmask[2]; pk.w = pack_h16x2_t(v[6], v[7], satm) & tmask[3];
                    if (!(ABL & 4)) *(uint4*)(outp + m * p.ldo + col) = pk;
                    else asm volatile("" ::"v"(pk.x), "v"(pk.y), "v"(pk.z), "v"(pk.w));
                    if (STATS) {
                        const float r[8] = {h16_lo(pk.x), h16_hi(pk.x), h16_lo(pk.y), h16_hi(pk.y), h16_lo(pk.z), h16_hi(pk.z), h16_lo(pk.w), h16_hi(pk.w)};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { st_s[e] += r[e]; st_q[e] += r[e] * r[e]; }
                    }
                }
            }
        }
        sat_report(satm);
        if (STATS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { st_s[e] = slot_sum<8>(st_s[e]); st_q[e] = slot_sum<8>(st_q[e]); }
            if (lane_o < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    *(lds_f_ptr)(st_base + ((wave * 64 + 8 * lane_o + e) * 2) * 4) = st_s[e];
                    *(lds_f_ptr)(st_base + ((wave * 64 + 8 * lane_o + e) * 2 + 1) * 4) = st_q[e];
                }
            }
        }
    };
    const int ep_variant = (p.act != GP_ACT_NONE ? 4 : 0) | (p.res ? 2 : 0) | (want_stats ? 1 : 0);
    auto epilogue = [&](unsigned stg) __attribute__((always_inline)) {
        // (a direct form -- 16-byte stores straight from the MFMA layout, no LDS round trip -- measured 15 % slower end to end)
        switch (ep_variant) {
            case 0: epilogue_body(stg, IC<0>{}, IC<0>{}, IC<0>{}); break;
            case 1: epilogue_body(stg, IC<0>{}, IC<0>{}, IC<1>{}); break;
            case 2: epilogue_body(stg, IC<0>{}, IC<1>{}, IC<0>{}); break;
            case 3: epilogue_body(stg, IC<0>{}, IC<1>{}, IC<1>{}); break;
            default: epilogue_body(stg, IC<1>{}, IC<1>{}, IC<1>{}); break;  // (rare: activation fused into a halo conv)
        }
        acc_init();
    };
    // Statistics are accumulated per WORKGROUP (all its tiles belong to one image and one channel slice) and written once at the end:
    // row (b, jw / tiles_n) of [B * R][N][2], R = J / tiles_n rows per image, followed by the pixel count of every row -- 36x fewer
    // partial rows for the finalize kernel than one per tile (2304 tiles per 768x768 image).
    float run_s = 0.f, run_q = 0.f;
    int run_px = 0;
    auto flush_stats = [&]() __attribute__((always_inline)) {  // after a workgroup barrier that follows epilogue(): waves (wm, wn) -> channel sums
        const int tile_i = PH ? sp_cur >> 2 : sp_cur;
        const int ty = tile_i / tiles_x, tx = tile_i - ty * tiles_x;
        run_px += PH ? min(16, Hi - 16 * ty) * min(16, Wi - 16 * tx) : min(TH, Ho - TH * ty) * min(16, Wo - 16 * tx);
        if (tid < BN) {
            // (inline asm: a compiler-visible LDS read here would make hipcc drain the DMA ring first, see tp_load)
            const unsigned a = st_base + (unsigned)tid * 8u;  // [(wm * 2 + wn) * 64 + ch][2] floats, tid = wn * 64 + ch
            f32x2_t v0, v1, v2, v3;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:2048\n\t"
                         "ds_read_b64 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
            run_s += ((v0.x + v1.x) + v2.x) + v3.x;
            run_q += ((v0.y + v1.y) + v2.y) + v3.y;
        }
    };
    auto store_stats = [&]() __attribute__((always_inline)) {
        const int R = J / tiles_n, row = b * R + jw / tiles_n;
        if (tid < BN && n0 + tid < n_out) {
            float* so = p.stats_out + ((long long)row * p.N + n0 + tid) * 2;
            so[0] = run_s;
            so[1] = run_q;
        }
        if (tid == 0 && nt == 0) p.stats_out[(long long)p.B * R * p.N * 2 + row] = (float)run_px;
    };

    // ---- prologue (first tile) ---------------------------------------------------------------------------------------------------------
    acc_init();
    setup_fetch(sp_cur);
    if (PH) {
        w_uni = (unsigned)((sp_cur & 3) * NT * Cin * 2);  // weights [n][phase][2 x 2 taps][Cin]: this unit's phase
        set_phase_bases(sp_cur);
    }
    stage_halo(0, 0);
    stage_w(0, w_step);
    stage_w(1, w_step);
    stage_w(2, w_step);
    Half f0, f1a, f1b;
    halo_wait_vm<B_IT>();
    __builtin_amdgcn_s_barrier();
    if (fused) {  // chunk 0 of the first tile has nothing to hide under
        tbl_load(0);
#pragma unroll
        for (int k = 0; k < T_IT; ++k) {
            TPart t;
            tp_load(t, 0, k);
            tp_mid(t);
            tp_finish(t);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    load_half(f0, IC<0>{}, IC<0>{}, IC<0>{});
    load_half(f1a, IC<0>{}, IC<0>{}, IC<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if ((ABL & 4096) && second_half) __builtin_amdgcn_s_setprio(1);  // r4 experiment: static priority for the later-dispatched half (guide T5)
    // ---- main loop over (tile, chunk), nine unrolled taps each --------------------------------------------------------------------------
    int cc = 0;
    bool tile_end = cpt == 1;                                    // this chunk is the last of its tile
    bool final_ = tile_end && sp_cur + sp_stride >= tiles_sp;    // ... and of the workgroup
    auto kstep = [&](auto tapc, auto parc, Half& cur1, Half& nxt1) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value;
        constexpr int TAP1 = (TAP + 1) % NT, PAR1 = TAP == NT - 1 ? PAR ^ 1 : PAR;
        constexpr int WSLOT = PH ? (TAP + 3) % 4 : TAP % 3;  // ring slot of tile s + 3 (3-deep ring: the slot of tile s)
        const bool issue_w = !(final_ && TAP >= NT - 3), issue_h = TAP == 0 && !final_;
        const int fcc = tile_end ? 0 : cc + 1;  // chunk (of the fetch tile) staged at tap 0 and normalised in taps 3..7
        // after tile s + 3: next tap / first tap of the next chunk / first tile of the next unit (PH: of ITS phase)
        int adv = (TAP + 3) % NT == NT - 1 ? (tile_end ? w_tile_wrap : w_wrap) : w_step;
        if (PH && (TAP + 3) % NT == NT - 1 && tile_end) adv += (((sp_cur + sp_stride) & 3) - (sp_cur & 3)) * NT * Cin;
        // role split: waves 4-7 issue their DMA before the MFMAs, waves 0-3 after -- except in a tile's last step, where a DMA
        // issued first would sit under the epilogue's vmcnt(0)
        const bool dma_first = ((ABL & 64) ? true : (ABL & 128) ? false : second_half) && !(TAP == NT - 1 && tile_end);
        if (dma_first) {
            if (issue_w) stage_w(WSLOT, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        __builtin_amdgcn_sched_barrier(0);
        // transform parts of this step (halo of chunk cc+1: complete for everybody from the barrier before tap 3, first read in tap 8)
        // 18x18 halo: 2592 items = 5 full parts (taps 3..7, one each) + 32 items left over, which only wave 0 handles (in tap 3, after the
        // MFMAs): as a sixth part for everybody they cost 1/6 of the transform's VALU time for 1 % of its work.  10x10 halo: two parts.
        constexpr int T_FULL = (HROWS * 8) / 512, T_TAIL_WAVES = ((HROWS * 8) % 512 + 63) / 64;  // 5, 1  /  1, 5
        constexpr int NP = !fused ? 0 : T_IT == 6 ? ((TAP >= 3 && TAP <= 7) ? 1 : 0) : ((TAP == 3 || TAP == 4) ? 1 : 0);
        constexpr int P0 = TAP - 3;
        if constexpr (TAP < NT - 1) {
            TPart tp[NP > 0 ? NP : 1];
            if constexpr (NP > 0) {
                if constexpr (TAP == 3) tbl_load(fcc);
#pragma unroll
                for (int u = 0; u < NP; ++u) tp_load(tp[u], PAR ^ 1, P0 + u);
            }
            load_half(nxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
            if constexpr (NP > 0) {
#pragma unroll
                for (int u = 0; u < NP; ++u) tp_mid(tp[u]);
            }
            mfma16(f0);
            if constexpr (NP == 0) {
                interleave();
            } else {  // 16 MFMAs; 8 fragment reads + the item (+ the 4 table reads of the chunk in tap 3) under the first five, then ~28 VALU + 8 transcendentals
#pragma unroll
                for (int q = 0; q < 4; ++q) { sgb<0x008, 1>(); sgb<0x100, 2>(); }
                sgb<0x008, 1>(); sgb<0x100, TAP == 3 ? 5 : 1>();
#pragma unroll
                for (int q = 5; q < 16; ++q) { sgb<0x008, 1>(); sgb<0x002, 3>(); sgb<0x400, 1>(); }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_half(f0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
            if constexpr (NP > 0) {
#pragma unroll
                for (int u = 0; u < NP; ++u) tp_finish(tp[u]);
            }
            mfma16(cur1);
            if constexpr (NP == 0) {
                interleave();
            } else {  // 16 MFMAs, 8 LDS reads, ~24 NP VALU + 8 NP transcendentals, NP LDS writes
#pragma unroll
                for (int q = 0; q < 8; ++q) { sgb<0x008, 2>(); sgb<0x100, 1>(); sgb<0x002, 3 * NP>(); sgb<0x400, NP>(); }
                sgb<0x200, NP>();
            }
            if constexpr (fused && T_IT == 6 && TAP == 3) {  // the 32 left-over items (uniform branch: wave 0 only)
                __builtin_amdgcn_sched_barrier(0);
                if (wave < T_TAIL_WAVES) {
                    TPart t;
                    tp_load(t, PAR ^ 1, T_FULL);
                    tp_mid(t);
                    tp_finish(t);
                }
            }
        } else {
            mfma16(f0);
            mfma16(cur1);
            __builtin_amdgcn_sched_barrier(0);
            if (tile_end) {
                if (!(ABL & 32)) halo_wait_vm<0>();  // everything this wave has in flight has landed: stores issued below cannot delay a certification
                if (ABL & 1024) {  // no epilogue, accumulators kept alive (the MFMAs stay)
#pragma unroll
                    for (int i = 0; i < FN; ++i)
#pragma unroll
                        for (int j = 0; j < FM; ++j) asm volatile("" ::"v"(acc[i][j]));
                    acc_init();
                } else if (!(ABL & 1)) epilogue((UPS ? a_base + G3::EP_OFF : a_base + PAR * A_BUF) + wave * 4096);
            }
            if (!final_) {
                if (PH && tile_end) set_phase_bases(sp_cur + sp_stride);  // the fragments read from here on belong to the next unit
                load_half(f0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
                load_half(nxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!dma_first) {
            if (issue_w) stage_w(WSLOT, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        if (TAP == NT - 1 && final_) return;
        if (ABL & 32) {}  // (no waits for the DMA: results are garbage)
        else if (PH) {  // four steps per chunk: the halo issued in tap 0 is read from tap 3 on; the workgroup's last three steps issue no tile
            if (TAP <= 1) { if (!final_) halo_wait_vm<A_IT + B_IT>(); else if (TAP == 0) halo_wait_vm<B_IT>(); else halo_wait_vm<0>(); }
            else if (TAP == 2) { if (final_) halo_wait_vm<0>(); else halo_wait_vm<B_IT>(); }
            else if (!tile_end) halo_wait_vm<B_IT>();
        }
        else if (TAP <= 1) { if (!final_) halo_wait_vm<A_IT + B_IT>(); else halo_wait_vm<B_IT>(); }
        else if (TAP < 6) halo_wait_vm<B_IT>();
        else if (TAP < 8) { if (final_) halo_wait_vm<0>(); else halo_wait_vm<B_IT>(); }
        else if (!tile_end) halo_wait_vm<B_IT>();  // (tile end: certified by the vmcnt(0) ahead of the epilogue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(((ABL & 256) && (TAP & 1)) || (ABL & 512))) __builtin_amdgcn_s_barrier();  // (256: every other barrier, 512: none -- garbage results)
    };
    auto chunk = [&](auto parc, Half& fa, Half& fb) __attribute__((always_inline)) {
        if (tile_end && !final_) setup_fetch(sp_cur + sp_stride);  // from here on halo staging / normalisation belong to the next tile
        kstep(IC<0>{}, parc, fa, fb); kstep(IC<1>{}, parc, fb, fa); kstep(IC<2>{}, parc, fa, fb);
        kstep(IC<3>{}, parc, fb, fa);
        if constexpr (!PH) {
            kstep(IC<4>{}, parc, fa, fb); kstep(IC<5>{}, parc, fb, fa);
            kstep(IC<6>{}, parc, fa, fb); kstep(IC<7>{}, parc, fb, fa); kstep(IC<8>{}, parc, fa, fb);
        }
        if (tile_end) {
            if (want_stats) {
                if (final_) __syncthreads();  // (nothing in flight any more)
                flush_stats();
                if (final_) store_stats();
            }
            sp_cur += sp_stride;
            cc = 0;
        } else {
            ++cc;
        }
        tile_end = cc == cpt - 1;
        final_ = tile_end && sp_cur + sp_stride >= tiles_sp;
    };
    while (true) {
        chunk(IC<0>{}, f1a, f1b);
        if (sp_cur >= tiles_sp) break;
        if constexpr (PH) chunk(IC<1>{}, f1a, f1b);  // (an even number of steps per chunk: the fragment ping-pong is back where it started)
        else chunk(IC<1>{}, f1b, f1a);
        if (sp_cur >= tiles_sp) break;
    }
}

bool conv_halo_applicable(const IGemmParams& p) {
    if (p.ks != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.batch > 1 || p.out_fp32 > 1 || p.act == GP_ACT_GEGLU) return false;
    if (p.bias_mode == GP_BIAS_ROW || (p.ldo & 7)) return false;
    if (p.out_fp32 == 1) {  // fp32 rows out (contract precision): the persistent kernel's F32O epilogue, plain input, whole slots, exact x2 only as phases
        if (p.in_scale || (p.n_store & 7) || (p.res && (!p.res_f32 || (p.ldres & 7) || p.ldres < p.n_store)) || (p.dbg & 256)) return false;
        if (p.ups && !conv_halo_uses_phases(p)) return false;
    } else if (p.res_f32) return false;
    if (p.in_scale && p.Cin > GN_MAXC) return false;
    if (p.ups) {
        if (p.Hu != 2 * p.Hi || p.Wu != 2 * p.Wi || p.Ho != p.Hu || p.Wo != p.Wu) return false;
    } else if (p.Ho != p.Hi || p.Wo != p.Wi) return false;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    if (((ncols + 127) / 128) * 128 > p.n_rows) return false;  // the kernels read whole 128-row weight tiles
    // 32-bit byte offsets into one image / the packed weight (buffer-resource DMA of conv3x3_halo3_kernel)
    if ((long long)p.Hi * p.Wi * p.Cin * 2 >= 0xfffffff0ll || (long long)p.n_rows * p.ldw * 2 >= 0xfffffff0ll) return false;
    return p.Ho >= 16 && p.Wo >= 16;
}

template <bool UPS, int FUSED, int ABL, int TR = 4, bool PH = false>
static void launch_halo3_one(const IGemmParams& p, int grid, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    constexpr int lds = Halo3Geom<UPS, TR, PH>::LDS;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo3_kernel<UPS, FUSED, ABL, TR, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    hipLaunchKernelGGL((conv3x3_halo3_kernel<UPS, FUSED, ABL, TR, PH>), dim3(grid), dim3(512), lds, s, p);
}

// x2-upsample conv as four phase convolutions (PH): p.wt_ph = [n_rows][4 phases][2 x 2 taps][Cin] weights (engine: pack_phases)
static void launch_halo3_ph(const IGemmParams& p, int grid, hipStream_t s) {
    IGemmParams q = p;
    q.wt = p.wt_ph;
    q.ldw = 16 * p.Cin;
    if (p.out_fp32 == 1) launch_halo3_one<false, 0, HALO_F32O, 4, true>(q, grid, s);
    else launch_halo3_one<false, 0, 0, 4, true>(q, grid, s);
}
// the persistent kernel (conv3x3_halo3_kernel) takes the problem: whole 8-channel slots everywhere (else the one-tile-per-workgroup fallback)
static bool halo_persistent(const IGemmParams& p) {
    const bool slots_ok = (p.n_store & 7) == 0 && (p.ldo & 7) == 0 && (!p.res || ((p.ldres & 7) == 0 && p.ldres >= p.n_store));
    if (p.out_fp32 == 1) return slots_ok;  // (no one-tile-per-workgroup fallback with fp32 rows: conv_halo_applicable refuses those shapes)
    return slots_ok && !(p.dbg & 256);
}
// ONE decision for the launcher, the engine's executed-flop accounting and the test entry point's guard: the phase kernel exists in the
// persistent form only
bool conv_halo_uses_phases(const IGemmParams& p) {
    return p.ups && p.wt_ph && !p.in_scale && !gp_sw().no_up_phases && p.Ho == 2 * p.Hi && p.Wo == 2 * p.Wi && halo_persistent(p) &&
           (long long)p.n_rows * 16 * p.Cin * 2 < 0xfffffff0ll;
}

static void launch_halo3(const IGemmParams& p, int grid, int tr, hipStream_t s) {
    const int abl = (p.dbg >> 9) & 2047;  // profiling ablations (GENPERCEPT_IGEMM_DBG = 512 * ABL), plain convs only
    const int fused = !p.in_scale ? 0 : p.in_silu ? 2 : 1;
    if (p.out_fp32 == 1) {  // contract precision (conv_halo_applicable: plain input, no nine-tap upsample)
        if (tr == 3) launch_halo3_one<false, 0, HALO_F32O, 3>(p, grid, s);
        else launch_halo3_one<false, 0, HALO_F32O, 4>(p, grid, s);
        return;
    }
    if (tr == 3) { launch_halo3_one<false, 0, 0, 3>(p, grid, s); return; }  // (halo_plan: plain stride-1 convs only)
    if (p.ups) {
        if (fused == 2) launch_halo3_one<true, 2, 0>(p, grid, s);
        else if (fused == 1) launch_halo3_one<true, 1, 0>(p, grid, s);
        else launch_halo3_one<true, 0, 0>(p, grid, s);
    } else if (fused == 2) launch_halo3_one<false, 2, 0>(p, grid, s);
    else if (fused == 1) launch_halo3_one<false, 1, 0>(p, grid, s);
    else if (abl == 2) launch_halo3_one<false, 0, 2>(p, grid, s);
    else if (abl == 24) launch_halo3_one<false, 0, 24>(p, grid, s);
#ifdef GP_HALO_ABLATIONS  // the other r3 / r4 experiments (profiles/HISTORY.md): hipcc ... -DGP_HALO_ABLATIONS=1
    else if (((p.dbg >> 24) & 3) == 1) launch_halo3_one<false, 0, 2048>(p, grid, s);   // r4 (IGemmParams::dbg bits 24-25): front-loaded fragment
    else if (((p.dbg >> 24) & 3) == 2) launch_halo3_one<false, 0, 4096>(p, grid, s);   // reads / static wave priority / both: all neutral
    else if (((p.dbg >> 24) & 3) == 3) launch_halo3_one<false, 0, 6144>(p, grid, s);   // (profiles/r04_halo3_schedule_ab.json)
    else if (((p.dbg >> 26) & 3) == 1) launch_halo3_one<false, 0, 8192>(p, grid, s);    // r4 (bits 26-27): how much of the K loop is the LDS fragment
    else if (((p.dbg >> 26) & 3) == 2) launch_halo3_one<false, 0, 16384>(p, grid, s);   // traffic -- no weight fragments / no pixel fragments / neither
    else if (((p.dbg >> 26) & 3) == 3) launch_halo3_one<false, 0, 24576>(p, grid, s);   // (MFMAs, DMA, waits, barriers, epilogue unchanged; garbage results)
    else if ((p.dbg >> 30) & 1) launch_halo3_one<false, 0, 32768>(p, grid, s);          // r4 (bit 30): every fragment read issued twice
    else if ((p.dbg >> 31) & 1) launch_halo3_one<false, 0, 65536>(p, grid, s);          // r4 (bit 31): LDS-DMA as global_load_lds (r2 / r3) instead of MUBUF
    else if (abl == 4) launch_halo3_one<false, 0, 4>(p, grid, s);
    else if (abl == 8) launch_halo3_one<false, 0, 8>(p, grid, s);
    else if (abl == 16) launch_halo3_one<false, 0, 16>(p, grid, s);
    else if (abl == 32) launch_halo3_one<false, 0, 32>(p, grid, s);
    else if (abl == 64) launch_halo3_one<false, 0, 64>(p, grid, s);
    else if (abl == 128) launch_halo3_one<false, 0, 128>(p, grid, s);
    else if (abl == 256) launch_halo3_one<false, 0, 256>(p, grid, s);
    else if (abl == 512) launch_halo3_one<false, 0, 512>(p, grid, s);
    else if (abl == 536) launch_halo3_one<false, 0, 536>(p, grid, s);
    else if (abl == 1024) launch_halo3_one<false, 0, 1024>(p, grid, s);
#endif
    else launch_halo3_one<false, 0, 0>(p, grid, s);
}

static int halo_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}
// persistent kernel: B x J workgroups, J = workgroups per image (a multiple of tiles_n, about #CU / B); th = output rows of a tile (16 or 12)
static int halo3_wgs_per_image(const IGemmParams& p, int ncu, int th = 16) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + 127) / 128, tiles_sp = ((p.Wo + 15) / 16) * ((p.Ho + th - 1) / th);
    int per_img = ncu / p.B;
    if (per_img < 1) per_img = 1;
    int J = (per_img / tiles_n) * tiles_n;
    if (J < tiles_n) J = tiles_n;
    if (J > tiles_sp * tiles_n) J = tiles_sp * tiles_n;
    return J;
}
// workgroups per image (a multiple of tiles_n) of the persistent kernel: one function for the launch AND for the statistics-row count the engine
// allocates.  (r4's alternative structures -- 32 x 16 tiles, two workgroups per CU, Winograd F(2,3) along x -- measured 0-10 % slower and live in
// tools/experiments/ with their tests and profiles since r5.)
//
// r5: 12-row x 16-column tiles (TR = 3) where the 16 x 16 tiling quantises badly over the persistent grid.  The makespan of a launch is
// ceil(tiles / slots) rounds of one tile, slots = workgroups per (image, channel slice).  At 96 x 96 with 512 output channels and batch 4 (the VAE
// encoder's last level / mid block and the decoder's first: 19 launches, 3.3 ms of a 45-ms pass) that is 36 tiles on 16 slots = 3 rounds for 2.25
// rounds of work (~1000 TFLOP/s against ~1400 for the well-quantised 192 x 192 shapes); 12-row tiles give 48 tiles = exactly 3 rounds of tiles 3/4
// the size.  A 12-row tile does 12 MFMAs per 7 fragment reads and per weight tile instead of 16 per 8, so it is priced at 0.80 of a 16-row tile,
// not 0.75, and taken only for a >= 5 % shorter makespan.  p.dbg bits 20-21 (tests / kbench): 1 forces TR = 3 where it applies, 2 forbids it.
static int halo_plan(const IGemmParams& p, int* tr) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + 127) / 128;
    const int J16 = halo3_wgs_per_image(p, halo_ncu(), 16);
    *tr = 4;
    const int force = (p.dbg >> 20) & 3;
    const bool can12 = !p.ups && !p.in_scale && force != 2 && !((p.dbg >> 9) & 2047) && p.Ho >= 12;
    if (!can12) return J16;
    const int J12 = halo3_wgs_per_image(p, halo_ncu(), 12);
    const int t16 = ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16), t12 = ((p.Wo + 15) / 16) * ((p.Ho + 11) / 12);
    const int s16 = J16 / tiles_n, s12 = J12 / tiles_n;
    const double c16 = (double)((t16 + s16 - 1) / s16), c12 = 0.80 * (double)((t12 + s12 - 1) / s12);
    if (force == 1 || c12 <= 0.95 * c16) {
        *tr = 3;
        return J12;
    }
    return J16;
}
// statistics rows per image the halo kernel will write for this problem (per-workgroup partials + counts, "mode 2"); 0 = one row per
// 16x16 tile ("mode 1", conv3x3_halo2_kernel)
int conv_halo_stat_rows(const IGemmParams& p) {
    if (!halo_persistent(p)) return 0;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    int tr;
    return halo_plan(p, &tr) / ((ncols + 127) / 128);
}

void launch_conv_halo(const IGemmParams& p, hipStream_t s) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + 127) / 128, tiles_sp = ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16);
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HaloGeom<false>::LDS);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HaloGeom<true>::LDS);
    });
    // (conv_halo_applicable guarantees whole 128-row weight tiles)
    if (halo_persistent(p)) {
        int tr;
        const int J = halo_plan(p, &tr);
        if (conv_halo_uses_phases(p)) launch_halo3_ph(p, p.B * J, s);
        else launch_halo3(p, p.B * J, tr, s);
        return;
    }
    const int tiles = tiles_sp * p.B * tiles_n;
    if (p.ups) hipLaunchKernelGGL((conv3x3_halo2_kernel<true>), dim3(tiles), dim3(512), HaloGeom<true>::LDS, s, p);
    else hipLaunchKernelGGL((conv3x3_halo2_kernel<false>), dim3(tiles), dim3(512), HaloGeom<false>::LDS, s, p);
}

GP_SAT_TU(conv_halo)  // fp16 build: address of this translation unit's saturation flag (common.h)
