// conv3x3 (stride 1, pad 1) with halo reuse on a 512-PIXEL workgroup tile -- the lower-bytes-per-flop successor of conv3x3_halo3_kernel
// (conv_halo.hip) for the plain 3x3 convs of the large VAE / UNet maps (r3: 60 launches, 15.5 of a 45.7 ms pass at 1140-1400 TFLOP/s).
// r3's ablations showed that kernel bound by the chip's power budget: every byte moved L2 -> LDS and every fragment read is paid in clock
// (1.92 GHz as shipped, 2.23 GHz without the LDS-DMA).  Same dataflow, twice the tile:
//
//   Workgroup tile: 32 x 16 OUTPUT PIXELS x 128 output channels (halo3: 16 x 16 x 128).  8 waves = 4 groups of 4 pixel rows x 2 channel
//        halves; a wave owns 4 rows x 32 pixels x 64 channels = 8 accumulator tiles of v_mfma_f32_32x32x16 (128 registers; halo3: 16
//        tiles of 16x16x32, 64 registers).  Two waves per SIMD, like halo3.
//   K chunk = 32 input channels (halo3: 64): the 34 x 18 halo of a chunk is 612 LDS rows of 64 bytes = 38.25 KiB, double-buffered;
//        K-step = (chunk, tap) = [128 cout][32 cin] weight tile of 8 KiB through a 3-deep LDS-DMA ring (slot = tap % 3).
//   Per K-step and workgroup: 8 KiB of weights + 4.3 KiB of halo by LDS-DMA for 8.4 MFLOP -- 41 % fewer bytes per flop than halo3
//        (16 + 4.6 KiB for 4.2 MFLOP) -- and 12 ds_read_b128 per 16 MFMAs of 32x32x16 (halo3: 16 reads per 32 MFMAs of 16x16x32 with half
//        the flops each): 0.75x the fragment bytes per flop, and the 32x32 shape reads its operands from the register file half as often.
//   Software pipeline, synchronisation, persistence, per-wave epilogue, statistics: halo3's (see conv_halo.hip), restated for the new
//        geometry: fragments of step s+1 are read between step s's MFMAs (1 MFMA : 1 ds_read), counted vmcnt + ONE raw s_barrier per
//        K-step, waves 4-7 issue their DMA before their MFMAs and waves 0-3 after, one workgroup per CU walks the tiles of one image and
//        one 128-channel slice, the epilogue stages [32 px][32 ch] fp32 blocks through the halo buffer the finished chunk released.
//   LDS images (tests/test_lds_layout.py checks all of them against the ds_read_b128 / ds_write_b128 bank model):
//        halo row R = hy * 34 + hx, 64 bytes = four 16-byte slots, logical slot s at physical slot s ^ ((hx >> 2) & 3): a 32x32x16
//        fragment read (lane = pixel x, k-half = lane >> 5) of 32 consecutive halo columns is conflict-free for every tap;
//        weight row r (64 bytes) the same with key (r >> 2) & 3; epilogue block [32 px][8 units of 16 B], unit u at u ^ f(px),
//        f(px) = ((px >> 1) & 3) | ((px & 1) << 2): conflict-free for the accumulator writes and for the row read-back.
// Launch policy (conv_halo.hip: launch_conv_halo): plain input (no fused GroupNorm transform, no x2 upsample), no activation, and a tile
// count that fills the persistent grid at least as well as the 16 x 16 tiling does.
#include "common.h"
#include "kernels.h"

constexpr int H4_HW = 34, H4_HROWS = 18 * 34;         // halo: 34 columns x 18 rows of source pixels
constexpr int H4_GROUPS = (H4_HROWS + 15) / 16;       // 39 DMA pieces of 16 rows (1 KiB)
constexpr int H4_A_IT = (H4_GROUPS + 7) / 8;          // 5 DMA instructions per wave and halo (the 40th piece hits the dump KiB)
constexpr int H4_A_BUF = H4_GROUPS * 1024;            // 39 KiB
constexpr int H4_B_STAGE = 128 * 64;                  // [128 cout][32 cin] 16-bit
constexpr int H4_B_OFF = 2 * H4_A_BUF;
constexpr int H4_DUMP_OFF = H4_B_OFF + 3 * H4_B_STAGE;
constexpr int H4_ST_OFF = H4_DUMP_OFF + 1024;         // [8 waves][64 ch][sum, sumsq]
constexpr int H4_BIAS_OFF = H4_ST_OFF + 4096;         // [128] bias of the workgroup's channel slice
constexpr int H4_RUN_OFF = H4_BIAS_OFF + 512;         // [128 ch][sum, sumsq] running statistics of the workgroup
constexpr int H4_FETCH_OFF = H4_RUN_OFF + 1024;       // [512 threads][8 ints]: the fetch tile's halo source offsets + validity bits (see setup_fetch)
constexpr int H4_LDS = H4_FETCH_OFF + 512 * 32;       // 126 464 bytes
static_assert(8 * 4096 <= H4_A_BUF, "the epilogue stages 4 KiB per wave in a released halo buffer");

// lane id from the hardware (v_mbcnt), as a volatile asm: recomputed at every use instead of occupying a register -- or a scratch slot,
// whose reload costs an s_waitcnt vmcnt(0) in the middle of the DMA ring -- through the whole persistent loop
GP_DEV int h4_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
GP_DEV int h4_key(int hx) { return (hx >> 2) & 3; }
GP_DEV int h4_stg_key(int px) { return ((px >> 1) & 3) | ((px & 1) << 2); }

template <bool X3>
__global__ __launch_bounds__(512) void conv3x3_halo4_kernel(const IGemmParams p) {
    constexpr int BN = 128, NW = 8, A_IT = H4_A_IT, A_BUF = H4_A_BUF, B_STAGE = H4_B_STAGE, HW_ = H4_HW, B_IT = 1;
    constexpr int FC = 2, FJ = 4;  // accumulator tiles per wave: 2 blocks of 32 channels x 4 pixel rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + H4_B_OFF;
    char* const dump = smem + H4_DUMP_OFF;
    float* const s_st = (float*)(smem + H4_ST_OFF);
    float* const s_bias = (float*)(smem + H4_BIAS_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool second_half = wave >= NW / 2;

    const int Ho = p.Ho, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi, Cin = p.Cin;
    const int tiles_x = (Wo + 31) >> 5, tiles_y = (Ho + 15) >> 4, tiles_sp = tiles_x * tiles_y;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + BN - 1) / BN;
    const int J = gridDim.x / p.B;                  // workgroups per image, a multiple of tiles_n
    const int b = blockIdx.x / J;
    int jw = blockIdx.x - b * J;
    if ((J & 7) == 0) jw = (jw & 7) * (J >> 3) + (jw >> 3);  // workgroups of one XCD (id % 8) take neighbouring tiles
    const int nt = jw % tiles_n, sp_stride = J / tiles_n;
    int sp_cur = jw / tiles_n;                       // spatial tile being computed
    const int n0 = nt * BN;
    const int cpt = Cin >> 5;                        // 32-channel chunks per tile
    const h16_t* const in_b = p.in + (long long)b * Hi * Wi * Cin;

    // ---- fetch state: the tile whose halo is being staged (one chunk ahead of the compute) ------------------------------------------
    // Each thread's five source offsets (elements, inside the image) and their validity bits live in a private 32-byte LDS record, not in
    // registers: with 128 accumulator + 56 fragment registers hipcc kept them in SCRATCH through the K loop, and every reload at a chunk's
    // tap 0 came with an s_waitcnt vmcnt(0) that drained the DMA ring (seen in the ISA).  An LDS read only costs lgkmcnt.
    typedef __attribute__((address_space(3))) int* lds_i_ptr;
    auto setup_fetch = [&](int sp) __attribute__((always_inline)) {
        const int fty = sp / tiles_x, ftx = sp - fty * tiles_x;
        const int sy0 = fty * 16 - 1, sx0 = ftx * 32 - 1;
        unsigned ok_bits = 0;
        const int lane_o = h4_lane_now();
        const unsigned rec = (unsigned)(unsigned long long)(smem + H4_FETCH_OFF) + (unsigned)(wave * 64 + lane_o) * 32u;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = (wave + NW * i) * 16 + (lane_o >> 2);
            const int hy = r / HW_, hx = r - hy * HW_;
            const int iy = sy0 + hy, ix = sx0 + hx;
            const bool ok = r < H4_HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            *(lds_i_ptr)(rec + 4 * i) = (iy * Wi + ix) * Cin + (((lane_o & 3) ^ h4_key(hx)) << 3);
            if (ok) ok_bits |= 1u << i;
        }
        *(lds_i_ptr)(rec + 4 * A_IT) = (int)ok_bits;
    };
    const h16_t* zsrc_a = p.zero;
    // weight rows n0 .. n0+127 always exist (conv_halo4_applicable checks n_rows); wq walks the (tile, chunk, tap) tiles in issue order
    const h16_t* wq;
    {
        const int row = wave * 16 + (lane >> 2);
        wq = p.wt + (long long)(n0 + row) * p.ldw + (((lane & 3) ^ h4_key(row)) << 3);
    }
    const int w_step = Cin, w_wrap = 32 - 8 * Cin, w_tile_wrap = -8 * Cin - (cpt - 1) * 32;  // next tap / next chunk / first tile again

    const unsigned a_base = (unsigned)(unsigned long long)a_lds, b_base = (unsigned)(unsigned long long)b_lds;
    auto stage_halo = [&](int buf, int cc) __attribute__((always_inline)) {
        char* dst = a_lds + buf * A_BUF;
        const unsigned fetch_rec = (unsigned)(unsigned long long)(smem + H4_FETCH_OFF) + (unsigned)(wave * 64 + h4_lane_now()) * 32u;
        int off[A_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) off[i] = *(lds_i_ptr)(fetch_rec + 4 * i);
        const unsigned ok_bits = (unsigned)*(lds_i_ptr)(fetch_rec + 4 * A_IT);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int g = wave + NW * i;
            const h16_t* src = ((ok_bits >> i) & 1u) ? in_b + (off[i] + (cc << 5)) : zsrc_a;
            glds16(src, g < H4_GROUPS ? dst + g * 1024 : dump);
        }
    };
    auto stage_w = [&](int slot, int adv) __attribute__((always_inline)) {  // next weight tile in (tile, chunk, tap) order, then advance
        glds16(wq, b_lds + slot * B_STAGE + wave * 1024);
        wq += adv;
    };

    f32x16_t acc[FC][FJ];

    // Fragments of one k-half (16 channels) of a step.  Weight fragments run a FULL step ahead in three register sets like halo3's (the slot
    // of tile s is refilled during step s, so its last read has to be over at the barrier before); pixel fragments only HALF a step ahead
    // in two sets (a halo buffer stays put for a whole chunk): 24 + 32 registers instead of 72 -- with 128 accumulator registers the third
    // pixel set was what pushed the loop's long-lived addresses into scratch (a vmcnt(0) per reload, seen in the ISA).
    struct WHalf { h16x8_t w[FC]; };
    struct XHalf { h16x8_t x[FJ]; };
    unsigned xb[3][2], wb[2];
    auto frag_bases = [&]() __attribute__((always_inline)) {  // (re)computed after every epilogue: values that live ACROSS it end up in scratch
        const int lane_o = h4_lane_now();
        const int xl = lane_o & 31, hl = lane_o >> 5;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 2 + hl;
            wb[kk] = b_base + (wn * 64 + xl) * 64 + ((sl ^ h4_key(xl)) << 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int hx = xl + kx;
                xb[kx][kk] = a_base + ((4 * wm) * HW_ + hx) * 64 + ((sl ^ h4_key(hx)) << 4);
            }
        }
    };
    frag_bases();
    auto load_w = [&](WHalf& f, auto tapc, auto kkc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, KK = decltype(kkc)::value, SLOT = TAP % 3;
#pragma unroll
        for (int c = 0; c < FC; ++c) f.w[c] = lds_frag(wb[KK], SLOT * B_STAGE + c * 32 * 64);
    };
    auto load_x = [&](XHalf& f, auto tapc, auto parc, auto kkc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, KK = decltype(kkc)::value;
        constexpr int KY = TAP / 3, KX = TAP % 3;
#pragma unroll
        for (int j = 0; j < FJ; ++j) f.x[j] = lds_frag(xb[KX][KK], PAR * A_BUF + (j + KY) * HW_ * 64);
    };
    auto mfma8 = [&](const WHalf& fw, const XHalf& fx) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < FC; ++c)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[c][j] = mfma_32x32x16(fw.w[c], fx.x[j], acc[c][j]);
    };
    auto interleave = [&](int nreads) __attribute__((always_inline)) {  // 8 MFMAs, the fragment reads of the other register sets between the first ones
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < nreads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };

    if (tid < BN) {
        s_bias[tid] = (p.bias && p.bias_mode == GP_BIAS_COL && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
        ((float*)(smem + H4_RUN_OFF))[2 * tid] = 0.f;
        ((float*)(smem + H4_RUN_OFF))[2 * tid + 1] = 0.f;
    }
    __syncthreads();

    // ---- per-wave epilogue of the finished tile (sp_cur); `stg` = 4 KiB private LDS window ---------------------------------------------
    // (all LDS traffic through integer-addressed address_space(3) accesses: see conv_halo.hip)
    const int n_out = p.N;
    const bool want_stats = p.stats_out != nullptr;
    const unsigned st_base = (unsigned)(unsigned long long)s_st, bias_base = (unsigned)(unsigned long long)s_bias;
    // the accumulators of a tile start at the bias of their channels: element r of tile c = channel wn * 64 + 32 c + 8 (r >> 2) + 4 hi + (r & 3)
    auto acc_init = [&]() __attribute__((always_inline)) {
        const int lane_o = h4_lane_now();
        const unsigned ba = bias_base + (wn * 64 + 4 * (lane_o >> 5)) * 4;
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            f32x16_t bv;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t q = *(lds_f4_ptr)(ba + (32 * c + 8 * g) * 4);
                bv[4 * g] = q.x; bv[4 * g + 1] = q.y; bv[4 * g + 2] = q.z; bv[4 * g + 3] = q.w;
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[c][j] = bv;
        }
    };
    auto epilogue_body = [&](unsigned stg, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr bool RES = decltype(resc)::value != 0, STATS = decltype(statc)::value != 0;
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        const int lane_o = h4_lane_now();  // (keeps the address arithmetic below inside the epilogue)
        const int xw = lane_o & 31, hw = lane_o >> 5;       // write role: pixel column, channel half of an 8-channel slot
        const int px0 = lane_o >> 2, sl8 = lane_o & 3;      // read-back role: pixels px0 and px0 + 16 of a tile row, 8-channel slot sl8 of the block
        h16_t* outp = (h16_t*)p.out;
        const int ox0 = tx * 32 + px0, oy0 = ty * 16 + 4 * wm;
        const int col0 = n0 + wn * 64 + 8 * sl8;
        auto row_index = [&](int jj, int h) __attribute__((always_inline)) {  // output row (pixel) index of pass jj, item h, or -1
            const int oy = oy0 + jj, ox = ox0 + 16 * h;
            return (oy < Ho && ox < Wo) ? (b * Ho + oy) * Wo + ox : -1;
        };
        // residual rows: eight 16-byte loads per lane for the first 32-channel block up front; each register set is refilled with the second
        // block's row as soon as the first block's pass has consumed it, i.e. four passes ahead of its use (all sixteen rows at once would
        // need 64 registers on top of the 128 accumulators: the K loop's fragment bases then live in scratch, seen in the ISA)
        uint4 rv[FJ][2];
        auto load_res = [&](int c, int jj, int h) __attribute__((always_inline)) {
            const int m = row_index(jj, h), col = col0 + 32 * c;
            rv[jj][h] = make_uint4(0u, 0u, 0u, 0u);
            if (m >= 0 && col < p.n_store) rv[jj][h] = *(const uint4*)(p.res + (long long)m * p.ldres + col);
        };
        if (RES) {
#pragma unroll
            for (int jj = 0; jj < FJ; ++jj) { load_res(0, jj, 0); load_res(0, jj, 1); }
        }
        float satm = 0.f;  // fp16 build: max |value| this thread packs in this tile (common.h: sat_track / sat_report)
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const int col = col0 + 32 * c;
            const bool col_ok = col < p.n_store;
            unsigned tmask[4];  // slot reaches into the zero-padded channels: masks for the packed words
#pragma unroll
            for (int w = 0; w < 4; ++w) tmask[w] = (col + 2 * w < n_out ? 0xffffu : 0u) | (col + 2 * w + 1 < n_out ? 0xffff0000u : 0u);
            float st_s[8], st_q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
            for (int jj = 0; jj < FJ; ++jj) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // 4 channels (8 g + 4 hw ..+3 of the block) of pixel xw: 16-byte unit 2 g + hw of the pixel's row
                    const unsigned d = stg + xw * 128 + (((2 * g + hw) ^ h4_stg_key(xw)) << 4);
                    *(lds_f4_ptr)d = f32x4_t{acc[c][jj][4 * g], acc[c][jj][4 * g + 1], acc[c][jj][4 * g + 2], acc[c][jj][4 * g + 3]};
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int px = px0 + 16 * h;
                    const unsigned sa = stg + px * 128;
                    const int k = h4_stg_key(px);
                    const f32x4_t x0 = *(lds_f4_ptr)(sa + (((2 * sl8) ^ k) << 4)), x1 = *(lds_f4_ptr)(sa + (((2 * sl8 + 1) ^ k) << 4));
                    const long long m = row_index(jj, h);
                    if (m >= 0 && col_ok) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        if (RES) {
                            const uint4 r4 = rv[jj][h];
                            v[0] += h16_lo(r4.x); v[1] += h16_hi(r4.x); v[2] += h16_lo(r4.y); v[3] += h16_hi(r4.y);
                            v[4] += h16_lo(r4.z); v[5] += h16_hi(r4.z); v[6] += h16_lo(r4.w); v[7] += h16_hi(r4.w);
                        }
                        uint4 pk;
                        pk.x = pack_h16x2_t(v[0], v[1], satm) & tmask[0]; pk.y = pack_h16x2_t(v[2], v[3], satm) & tmask[1];
                        pk.z = pack_h16x2_t(v[4], v[5], satm) & tmask[2]; pk.w = pack_h16x2_t(v[6], v[7], satm) & tmask[3];
                        *(uint4*)(outp + m * p.ldo + col) = pk;
                        if (STATS) {
                            const float r[8] = {h16_lo(pk.x), h16_hi(pk.x), h16_lo(pk.y), h16_hi(pk.y), h16_lo(pk.z), h16_hi(pk.z), h16_lo(pk.w), h16_hi(pk.w)};
#pragma unroll
                            for (int e = 0; e < 8; ++e) { st_s[e] += r[e]; st_q[e] += r[e] * r[e]; }
                        }
                    }
                    if (RES && c == 0) load_res(1, jj, h);  // (this register set is free again)
                }
            }
            if (STATS) {  // lanes sharing a slot (lane & 3) -> lanes 0..3; [(wave) * 64 + 32 c + 8 slot + e][sum, sumsq]
#pragma unroll
                for (int e = 0; e < 8; ++e) { st_s[e] = slot_sum<4>(st_s[e]); st_q[e] = slot_sum<4>(st_q[e]); }
                if (lane_o < 4) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        *(lds_f_ptr)(st_base + ((wave * 64 + 32 * c + 8 * lane_o + e) * 2) * 4) = st_s[e];
                        *(lds_f_ptr)(st_base + ((wave * 64 + 32 * c + 8 * lane_o + e) * 2 + 1) * 4) = st_q[e];
                    }
                }
            }
        }
        sat_report(satm);
    };
    const int ep_variant = (p.res ? 2 : 0) | (want_stats ? 1 : 0);
    auto epilogue = [&](unsigned stg) __attribute__((always_inline)) {
        switch (ep_variant) {
            case 0: epilogue_body(stg, IC<0>{}, IC<0>{}); break;
            case 1: epilogue_body(stg, IC<0>{}, IC<1>{}); break;
            case 2: epilogue_body(stg, IC<1>{}, IC<0>{}); break;
            default: epilogue_body(stg, IC<1>{}, IC<1>{}); break;
        }
        acc_init();
        frag_bases();
    };
    // statistics per WORKGROUP (all its tiles belong to one image and one channel slice), written once at the end: the layout of
    // conv3x3_halo3_kernel's "mode 2" (conv_halo.hip): row (b, jw / tiles_n) of [B * R][N][2], then the pixel count of every row
    // (the running sums live in LDS: as registers they sat in scratch through the K loop, reloaded -- s_waitcnt vmcnt(0) -- under the DMA ring)
    int run_px = 0;
    const unsigned run_base = (unsigned)(unsigned long long)(smem + H4_RUN_OFF);
    auto flush_stats = [&]() __attribute__((always_inline)) {  // after a workgroup barrier that follows epilogue(): waves (wm, wn) -> channel sums
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        run_px += min(16, Ho - 16 * ty) * min(32, Wo - 32 * tx);
        const int tid_o = wave * 64 + h4_lane_now();
        if (tid_o < BN) {
            const unsigned a = st_base + (unsigned)tid_o * 8u;  // [(wm * 2 + wn) * 64 + ch][2] floats, tid = wn * 64 + ch
            const unsigned r = run_base + (unsigned)tid_o * 8u;
            f32x2_t v0, v1, v2, v3, acc0;
            asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:1024\n\tds_read_b64 %2, %5 offset:2048\n\t"
                         "ds_read_b64 %3, %5 offset:3072\n\tds_read_b64 %4, %6\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(acc0) : "v"(a), "v"(r) : "memory");
            acc0.x += ((v0.x + v1.x) + v2.x) + v3.x;
            acc0.y += ((v0.y + v1.y) + v2.y) + v3.y;
            asm volatile("ds_write_b64 %0, %1" ::"v"(r), "v"(acc0) : "memory");
        }
    };
    auto store_stats = [&]() __attribute__((always_inline)) {
        const int R = J / tiles_n, row = b * R + jw / tiles_n;
        const int tid_o = wave * 64 + h4_lane_now();
        if (tid_o < BN && n0 + tid_o < n_out) {
            f32x2_t acc0;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(acc0) : "v"(run_base + (unsigned)tid_o * 8u) : "memory");
            float* so = p.stats_out + ((long long)row * p.N + n0 + tid_o) * 2;
            so[0] = acc0.x;
            so[1] = acc0.y;
        }
        if (tid_o == 0 && nt == 0) p.stats_out[(long long)p.B * R * p.N * 2 + row] = (float)run_px;
    };

    // ---- prologue (first tile) ---------------------------------------------------------------------------------------------------------
    acc_init();
    setup_fetch(sp_cur);
    stage_halo(0, 0);
    stage_w(0, w_step);
    stage_w(1, w_step);
    stage_w(2, w_step);
    WHalf w0, w1a, w1b;  // w0: k-half 0 of the current step; w1a / w1b ping-pong: k-half 1 of the current / next step
    XHalf x0, x1;        // pixel fragments of k-half 0 / 1 of the current step (X3 = false: x1 is read during the step's first MFMA batch)
    XHalf x1b;           // X3: x1 / x1b ping-pong like w1a / w1b (k-half 1 of the current / next step), a full step ahead
    wait_vm<B_IT>();     // halo 0 and the tiles of taps 0, 1 have landed
    __builtin_amdgcn_s_barrier();
    load_w(w0, IC<0>{}, IC<0>{});
    load_w(w1a, IC<0>{}, IC<1>{});
    load_x(x0, IC<0>{}, IC<0>{}, IC<0>{});
    if constexpr (X3) load_x(x1, IC<0>{}, IC<0>{}, IC<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody holds its step-0 fragments: ring slot 0 may be refilled (3-deep ring)

    // ---- main loop over (tile, chunk), nine unrolled taps each --------------------------------------------------------------------------
    // Invariant at the top of step s = (cc, TAP): the barrier that certified the operands of step s+1 has been passed, w0 / cur1 hold both
    // weight k-halves of step s, x0 the pixel fragments of its first k-half, weight tiles up to step s+2 are issued.  The step issues tile s+3 into slot TAP % 3 (its previous content, tile s,
    // was read during step s-1), at tap 0 the halo of the next chunk, and reads the fragments of step s+1.
    int cc = 0;
    bool tile_end = cpt == 1;                                    // this chunk is the last of its tile
    bool final_ = tile_end && sp_cur + sp_stride >= tiles_sp;    // ... and of the workgroup
    auto kstep = [&](auto tapc, auto parc, WHalf& cur1, WHalf& nxt1, XHalf& xcur1, XHalf& xnxt1) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value;
        constexpr int TAP1 = (TAP + 1) % 9, PAR1 = TAP == 8 ? PAR ^ 1 : PAR;
        const bool issue_w = !(final_ && TAP >= 6), issue_h = TAP == 0 && !final_;
        const int fcc = tile_end ? 0 : cc + 1;  // chunk (of the fetch tile) staged at tap 0
        const int adv = (TAP + 3) % 9 == 8 ? (tile_end ? w_tile_wrap : w_wrap) : w_step;
        // role split: waves 4-7 issue their DMA before the MFMAs, waves 0-3 after -- except in a tile's last step, where a DMA issued first
        // would sit under the epilogue's vmcnt(0)
        const bool dma_first = second_half && !(TAP == 8 && tile_end);
        if (dma_first) {
            if (issue_w) stage_w(TAP % 3, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAP < 8) {
            // first batch: only the four pixel fragments the SECOND batch needs, requested between its first MFMAs (four MFMAs of cover before
            // they are used); everything for the next step -- eight fragments -- goes under the second batch and is waited for at the barrier
            if constexpr (X3) {
                load_x(xnxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
                load_w(nxt1, IC<TAP1>{}, IC<1>{});
                mfma8(w0, x0);
                interleave(6);
            } else {
                load_x(xcur1, IC<TAP>{}, IC<PAR>{}, IC<1>{});
                mfma8(w0, x0);
                interleave(4);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_x(x0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
            load_w(w0, IC<TAP1>{}, IC<0>{});
            if constexpr (!X3) load_w(nxt1, IC<TAP1>{}, IC<1>{});
            mfma8(cur1, xcur1);
            if constexpr (X3) interleave(6);
            else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        } else {
            if constexpr (!X3) load_x(xcur1, IC<TAP>{}, IC<PAR>{}, IC<1>{});
            mfma8(w0, x0);
            if constexpr (!X3) interleave(4);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(cur1, xcur1);
            __builtin_amdgcn_sched_barrier(0);
            if (tile_end) {
                // X3 = false reads this step's second pixel-fragment set from the halo buffer DURING the step: a wave that is held up (issuing
                // its DMA) must have them before a faster wave turns that buffer into its staging window
                if constexpr (!X3) __builtin_amdgcn_s_barrier();
                wait_vm<0>();  // everything this wave has in flight has landed: stores issued below cannot delay a certification
                epilogue(a_base + PAR * A_BUF + wave * 4096);
            }
            if (!final_) {
                load_x(x0, IC<TAP1>{}, IC<PAR1>{}, IC<0>{});
                load_w(w0, IC<TAP1>{}, IC<0>{});
                load_w(nxt1, IC<TAP1>{}, IC<1>{});
                if constexpr (X3) load_x(xnxt1, IC<TAP1>{}, IC<PAR1>{}, IC<1>{});
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!dma_first) {
            if (issue_w) stage_w(TAP % 3, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        // barrier(s+1): tile s+2 (and every halo issued before it) must have landed; tile s+3 and, while it was issued in tap 0 of this
        // chunk, the halo of the next chunk may stay in flight
        if (TAP == 8 && final_) return;
        if (TAP <= 1) { if (!final_) wait_vm<A_IT + B_IT>(); else wait_vm<B_IT>(); }
        else if (TAP < 6) wait_vm<B_IT>();
        else if (TAP < 8) { if (final_) wait_vm<0>(); else wait_vm<B_IT>(); }
        else if (!tile_end) wait_vm<B_IT>();  // (tile end: certified by the vmcnt(0) ahead of the epilogue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // next step's fragments are in registers
        __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](auto parc, WHalf& fa, WHalf& fb, XHalf& xa, XHalf& xb2) __attribute__((always_inline)) {
        if (tile_end && !final_) setup_fetch(sp_cur + sp_stride);  // from here on halo staging belongs to the next tile
        // (X3 = false: both pixel k-half-1 references name the same register set)
        auto step = [&](auto tapc, WHalf& c1, WHalf& n1, XHalf& xc, XHalf& xn) __attribute__((always_inline)) {
            if constexpr (X3) kstep(tapc, parc, c1, n1, xc, xn);
            else kstep(tapc, parc, c1, n1, x1, x1);
        };
        step(IC<0>{}, fa, fb, xa, xb2); step(IC<1>{}, fb, fa, xb2, xa); step(IC<2>{}, fa, fb, xa, xb2);
        step(IC<3>{}, fb, fa, xb2, xa); step(IC<4>{}, fa, fb, xa, xb2); step(IC<5>{}, fb, fa, xb2, xa);
        step(IC<6>{}, fa, fb, xa, xb2); step(IC<7>{}, fb, fa, xb2, xa); step(IC<8>{}, fa, fb, xa, xb2);
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
        chunk(IC<0>{}, w1a, w1b, x1, x1b);
        if (sp_cur >= tiles_sp) break;
        chunk(IC<1>{}, w1b, w1a, x1b, x1);
        if (sp_cur >= tiles_sp) break;
    }
}

// The plain stride-1 3x3 convs halo3 takes, minus fused input transforms, x2 upsampling and fused activations.
bool conv_halo4_applicable(const IGemmParams& p) {
    if (gp_sw().no_halo4 || !conv_halo_applicable(p) || p.ups || p.in_scale || p.act != GP_ACT_NONE) return false;
    if ((p.n_store & 7) || (p.ldo & 7) || (p.res && ((p.ldres & 7) || p.ldres < p.n_store))) return false;  // whole 8-channel slots (halo_persistent)
    if ((p.Cin & 31) || p.Cin < 64) return false;
    return p.Ho >= 16 && p.Wo >= 32;
}

// Does the 32 x 16 tiling fill a persistent grid of J workgroups per image at least as well as the 16 x 16 tiling (within `slack`)?
// The per-tile speed advantage (~10 %, kbench) is lost when the last round of tiles is mostly empty (96 x 96 maps: 18 tiles on 16 slots).
bool conv_halo4_preferred(const IGemmParams& p, int J) {
    if (!gp_sw().halo4_auto) return false;
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + 127) / 128;
    const int slots = J / tiles_n;
    if (slots < 1) return false;
    const int t4 = ((p.Wo + 31) / 32) * ((p.Ho + 15) / 16), t3 = ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16);
    if (t4 < slots) return false;  // (the grid -- and with it the statistics rows -- must not depend on the kernel choice)
    const double e4 = (double)t4 / (double)(((t4 + slots - 1) / slots) * slots), e3 = (double)t3 / (double)(((t3 + slots - 1) / slots) * slots);
    return e4 * 1.08 >= e3;
}

void launch_conv_halo4(const IGemmParams& p, int grid, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, H4_LDS);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, H4_LDS);
    });
    if ((p.dbg >> 22) & 1) hipLaunchKernelGGL(conv3x3_halo4_kernel<true>, dim3(grid), dim3(512), H4_LDS, s, p);   // A/B: three pixel-fragment sets
    else hipLaunchKernelGGL(conv3x3_halo4_kernel<false>, dim3(grid), dim3(512), H4_LDS, s, p);
}

GP_SAT_TU(conv_halo4)  // fp16 build: address of this translation unit's saturation flag (common.h)
