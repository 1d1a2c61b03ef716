// conv3x3 (stride 1, pad 1) with halo reuse at FOUR waves per SIMD: two independent 512-thread workgroups per CU, 128 registers per wave,
// <= 80 KiB of LDS each -- the occupancy experiment of round 4.
//
// Why: conv3x3_halo3_kernel (conv_halo.hip) runs ONE workgroup per CU, two waves per SIMD that move in lockstep (one s_barrier per K-step):
// its epilogue (7-15 % of the kernel, kbench "no epilogue"), its per-step barrier / DMA waits and the latency tail of its LDS fragment reads
// (10-14 % for the weight fragments alone, "no_wfrag") all happen in BOTH waves of a SIMD at the same time, so nothing covers them -- the matrix
// pipe is busy 57-68 % of the cycles.  Re-ordering inside that structure was neutral five times (profiles/HISTORY.md, r04_halo3_schedule_ab.json),
// and a bigger tile (conv_halo4.hip) ran the same number of cycles.  The hardware's own answer to "nothing covers the stall" is more independent
// waves: here a second workgroup on the same CU works on a different tile in its own phase, so one workgroup's epilogue, barrier and fragment
// latency sit under the other's MFMAs.
//
//   Workgroup tile: 16 x 16 output pixels x 128 output channels (halo3's).  8 waves = 4 groups of 4 pixel rows x 2 channel halves; a wave owns
//        4 rows x 16 pixels x 64 channels = 16 accumulator tiles of v_mfma_f32_16x16x32 (64 registers).
//   K chunk = 32 input channels (one MFMA K): the 18 x 18 halo of a chunk is 324 LDS rows of 64 bytes (21 KiB), double-buffered; K-step = (chunk, tap) =
//        [128 cout][32 cin] weight tile of 8 KiB through a 3-deep LDS-DMA ring (slot = tap % 3).  78.5 KiB per workgroup.
//   No register prefetch: the 8 fragments of a step (32 registers, ONE set) are read at its start and the compiler's lgkmcnt waits release the
//        16 MFMAs as they arrive; with 128 registers there is no room for halo3's second fragment set -- the other workgroup's waves are the cover.
//   Ring: step s reads tile s (certified by the barrier that ended step s-1), issues tile s+2 into the slot tile s-1 left at that barrier, and
//        at tap 0 the halo of the next chunk; counted vmcnt + one raw s_barrier per step, waves 4-7 issue their DMA before their MFMAs and waves
//        0-3 after (halo3's role split).
//   Tile end: one extra workgroup barrier (every wave has read its last fragments) before the per-wave epilogue stages [16 px][32 ch] fp32 blocks
//        (2 KiB per wave) through the halo buffer the finished chunk released.
//   LDS images (tests/test_lds_layout.py): halo row R = hy * 18 + hx, 64 bytes = four 16-byte slots, logical slot s at physical slot
//        s ^ (3 * ((hx >> 2) & 1)); weight row r the same with key 3 * ((r >> 3) & 1) (fragment rows 8 (a >> 2) + (a & 3) + const);
//        epilogue block [16 px][8 units of 16 B], unit u at u ^ f(px), f(px) = ((px >> 1) & 1) | ((px & 1) << 1) | (px & 4).
// Launch policy (conv_halo.hip: halo_plan): the plain stride-1 convs conv3x3_halo4_kernel can take, 2 x #CU workgroups.
#include "common.h"
#include "kernels.h"

constexpr int H5_HW = 18, H5_HROWS = 18 * 18;          // halo: 18 x 18 source pixels
constexpr int H5_GROUPS = (H5_HROWS + 15) / 16;        // 21 DMA pieces of 16 rows (1 KiB)
constexpr int H5_A_IT = (H5_GROUPS + 7) / 8;           // 3 DMA instructions per wave and halo (pieces 21..23 hit the dump KiB)
constexpr int H5_A_BUF = H5_GROUPS * 1024;             // 21 KiB
constexpr int H5_B_STAGE = 128 * 64;                   // [128 cout][32 cin] 16-bit
constexpr int H5_B_OFF = 2 * H5_A_BUF;
constexpr int H5_DUMP_OFF = H5_B_OFF + 3 * H5_B_STAGE;
constexpr int H5_ST_OFF = H5_DUMP_OFF + 1024;          // [8 waves][64 ch][sum, sumsq]
constexpr int H5_BIAS_OFF = H5_ST_OFF + 4096;          // [128] bias of the workgroup's channel slice
constexpr int H5_RUN_OFF = H5_BIAS_OFF + 512;          // [128 ch][sum, sumsq] running statistics of the workgroup
constexpr int H5_FETCH_OFF = H5_RUN_OFF + 1024;        // [512 threads][3 ints]: the fetch tile's halo source offsets, bit 0 = inside the image
constexpr int H5_LDS = H5_FETCH_OFF + 512 * 12;        // 80 384 bytes: two workgroups per CU (160 KiB)
static_assert(H5_LDS <= 80 * 1024, "two workgroups per CU");
static_assert(8 * 2048 <= H5_A_BUF, "the epilogue stages 2 KiB per wave in a released halo buffer");

GP_DEV int h5_lane_now() {  // (conv_halo4.hip: the lane id re-read from the hardware instead of a register that lives through the persistent loop)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
GP_DEV int h5_key(int hx) { return 3 * ((hx >> 2) & 1); }
GP_DEV int h5_wkey(int row) { return 3 * ((row >> 3) & 1); }
GP_DEV int h5_stg_key(int px) { return ((px >> 1) & 1) | ((px & 1) << 1) | (px & 4); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3x3_halo5_kernel(const IGemmParams p) {
    constexpr int BN = 128, NW = 8, A_IT = H5_A_IT, A_BUF = H5_A_BUF, B_STAGE = H5_B_STAGE, HW_ = H5_HW, B_IT = 1;
    constexpr int FN = 4, FM = 4;  // accumulator tiles per wave: 4 x 16 channels (two 32-channel blocks) x 4 pixel rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + H5_B_OFF;
    char* const dump = smem + H5_DUMP_OFF;
    float* const s_st = (float*)(smem + H5_ST_OFF);
    float* const s_bias = (float*)(smem + H5_BIAS_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool second_half = wave >= NW / 2;

    const int Ho = p.Ho, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi, Cin = p.Cin;
    const int tiles_x = (Wo + 15) >> 4, tiles_y = (Ho + 15) >> 4, tiles_sp = tiles_x * tiles_y;
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

    // ---- fetch state: the tile whose halo is being staged (one chunk ahead of the compute); per-thread records in LDS (conv_halo4.hip) -----
    typedef __attribute__((address_space(3))) int* lds_i_ptr;
    auto setup_fetch = [&](int sp) __attribute__((always_inline)) {
        const int fty = sp / tiles_x, ftx = sp - fty * tiles_x;
        const int sy0 = fty * 16 - 1, sx0 = ftx * 16 - 1;
        const int lane_o = h5_lane_now();
        const unsigned rec = (unsigned)(unsigned long long)(smem + H5_FETCH_OFF) + (unsigned)(wave * 64 + lane_o) * 12u;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = (wave + NW * i) * 16 + (lane_o >> 2);
            const int hy = r / HW_, hx = r - hy * HW_;
            const int iy = sy0 + hy, ix = sx0 + hx;
            const bool ok = r < H5_HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            const int off = (iy * Wi + ix) * Cin + (((lane_o & 3) ^ h5_key(hx)) << 3);  // a multiple of 8: bit 0 carries `ok`
            *(lds_i_ptr)(rec + 4 * i) = ok ? (off | 1) : 0;
        }
    };
    const h16_t* zsrc_a = p.zero;
    // weight rows n0 .. n0+127 always exist (conv_halo5_applicable checks n_rows); wq walks the (tile, chunk, tap) tiles in issue order
    const h16_t* wq;
    {
        const int row = wave * 16 + (lane >> 2);
        wq = p.wt + (long long)(n0 + row) * p.ldw + (((lane & 3) ^ h5_wkey(row)) << 3);
    }
    const int w_step = Cin, w_wrap = 32 - 8 * Cin, w_tile_wrap = -8 * Cin - (cpt - 1) * 32;  // next tap / next chunk / first tile again

    const unsigned a_base = (unsigned)(unsigned long long)a_lds, b_base = (unsigned)(unsigned long long)b_lds;
    auto stage_halo = [&](int buf, int cc) __attribute__((always_inline)) {
        char* dst = a_lds + buf * A_BUF;
        const unsigned fetch_rec = (unsigned)(unsigned long long)(smem + H5_FETCH_OFF) + (unsigned)(wave * 64 + h5_lane_now()) * 12u;
        int off[A_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) off[i] = *(lds_i_ptr)(fetch_rec + 4 * i);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int g = wave + NW * i;
            const h16_t* src = (off[i] & 1) ? in_b + ((off[i] & ~1) + (cc << 5)) : zsrc_a;
            glds16(src, g < H5_GROUPS ? dst + g * 1024 : dump);
        }
    };
    auto stage_w = [&](int slot, int adv) __attribute__((always_inline)) {  // next weight tile in (tile, chunk, tap) order, then advance
        glds16(wq, b_lds + slot * B_STAGE + wave * 1024);
        wq += adv;
    };

    f32x4_t acc[FN][FM];

    struct Frags { h16x8_t w[FN], x[FM]; };
    unsigned xb[3], wb;
    auto frag_bases = [&]() __attribute__((always_inline)) {  // (re)computed after every epilogue: values that live ACROSS it end up in scratch
        const int lane_o = h5_lane_now();
        const int a15 = lane_o & 15, q = lane_o >> 4;
        wb = b_base + (wn * 64 + 8 * (a15 >> 2) + (a15 & 3)) * 64 + ((q ^ h5_wkey(8 * (a15 >> 2))) << 4);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = a15 + kx;
            xb[kx] = a_base + ((4 * wm) * HW_ + hx) * 64 + ((q ^ h5_key(hx)) << 4);
        }
    };
    frag_bases();
    auto load_frags = [&](Frags& f, auto tapc, auto parc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, SLOT = TAP % 3, KY = TAP / 3, KX = TAP % 3;
        // fragment i: MFMA row a -> weight row wn * 64 + 32 (i >> 1) + 4 (i & 1) + 8 (a >> 2) + (a & 3), so that accumulator element r of lane (q, a)
        // is channel 32 (i >> 1) + 8 q + 4 (i & 1) + r: fragments (2 c, 2 c + 1) give a lane 8 consecutive channels of block c (halo3's row order)
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = lds_frag(wb, SLOT * B_STAGE + (32 * (i >> 1) + 4 * (i & 1)) * 64);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.x[j] = lds_frag(xb[KX], PAR * A_BUF + (j + KY) * HW_ * 64);
    };
    auto mfma16 = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[i][j]);
    };

    if (tid < BN) {
        s_bias[tid] = (p.bias && p.bias_mode == GP_BIAS_COL && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
        ((float*)(smem + H5_RUN_OFF))[2 * tid] = 0.f;
        ((float*)(smem + H5_RUN_OFF))[2 * tid + 1] = 0.f;
    }
    __syncthreads();

    // ---- per-wave epilogue of the finished tile (sp_cur); `stg` = 2 KiB private LDS window ---------------------------------------------
    // (all LDS traffic through integer-addressed address_space(3) accesses: see conv_halo.hip)
    const int n_out = p.N;
    const bool want_stats = p.stats_out != nullptr;
    const unsigned st_base = (unsigned)(unsigned long long)s_st, bias_base = (unsigned)(unsigned long long)s_bias;
    auto acc_init = [&]() __attribute__((always_inline)) {  // the accumulators of a tile start at the bias of their channels
        const int lane_o = h5_lane_now();
        const unsigned ba = bias_base + (wn * 64 + 8 * (lane_o >> 4)) * 4;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const f32x4_t bv = *(lds_f4_ptr)(ba + (32 * (i >> 1) + 4 * (i & 1)) * 4);
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = bv;
        }
    };
    auto epilogue_body = [&](unsigned stg, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr bool RES = decltype(resc)::value != 0, STATS = decltype(statc)::value != 0;
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        const int lane_o = h5_lane_now();  // (keeps the address arithmetic below inside the epilogue)
        const int aw = lane_o & 15, qw = lane_o >> 4;       // write role: pixel column, 8-channel group of the block
        const int px0 = lane_o >> 2, sl8 = lane_o & 3;      // read-back role: pixel px0 of a tile row, 8-channel slot sl8 of the block
        h16_t* outp = (h16_t*)p.out;
        const int ox = tx * 16 + px0, oy0 = ty * 16 + 4 * wm;
        const int col0 = n0 + wn * 64 + 8 * sl8;
        auto row_index = [&](int jj) __attribute__((always_inline)) { return (oy0 + jj < Ho && ox < Wo) ? (b * Ho + oy0 + jj) * Wo + ox : -1; };
        // residual rows: the four 16-byte loads of the first 32-channel block up front; each register set is refilled with the second block's
        // row as soon as the first block's pass has consumed it
        uint4 rv[FM];
        auto load_res = [&](int c, int jj) __attribute__((always_inline)) {
            const int m = row_index(jj), col = col0 + 32 * c;
            rv[jj] = make_uint4(0u, 0u, 0u, 0u);
            if (m >= 0 && col < p.n_store) rv[jj] = *(const uint4*)(p.res + (long long)m * p.ldres + col);
        };
        if (RES) {
#pragma unroll
            for (int jj = 0; jj < FM; ++jj) load_res(0, jj);
        }
        const unsigned wr = stg + aw * 128, wk = (unsigned)h5_stg_key(aw);
        const unsigned rd = stg + px0 * 128, rk = (unsigned)h5_stg_key(px0);
        float satm = 0.f;  // fp16 build: max |value| this thread packs in this tile (common.h: sat_track / sat_report)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = col0 + 32 * c;
            const bool col_ok = col < p.n_store;
            unsigned tmask[4];  // slot reaches into the zero-padded channels: masks for the packed words
#pragma unroll
            for (int w = 0; w < 4; ++w) tmask[w] = (col + 2 * w < n_out ? 0xffffu : 0u) | (col + 2 * w + 1 < n_out ? 0xffff0000u : 0u);
            float st_s[8], st_q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
            for (int jj = 0; jj < FM; ++jj) {
                // 8 channels 8 qw .. 8 qw + 7 of the block for pixel aw: 16-byte units 2 qw and 2 qw + 1 of the pixel's 128-byte row
                *(lds_f4_ptr)(wr + (((2 * qw) ^ wk) << 4)) = acc[2 * c][jj];
                *(lds_f4_ptr)(wr + (((2 * qw + 1) ^ wk) << 4)) = acc[2 * c + 1][jj];
                const f32x4_t x0 = *(lds_f4_ptr)(rd + (((2 * sl8) ^ rk) << 4)), x1 = *(lds_f4_ptr)(rd + (((2 * sl8 + 1) ^ rk) << 4));
                const long long m = row_index(jj);
                if (m >= 0 && col_ok) {
                    float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if (RES) {
                        const uint4 r4 = rv[jj];
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
                if (RES && c == 0) load_res(1, jj);  // (this register set is free again)
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
    // statistics per WORKGROUP, written once at the end: conv3x3_halo3_kernel's "mode 2" layout (conv_halo.hip), running sums in LDS
    int run_px = 0;
    const unsigned run_base = (unsigned)(unsigned long long)(smem + H5_RUN_OFF);
    auto flush_stats = [&]() __attribute__((always_inline)) {  // after a workgroup barrier that follows epilogue(): waves (wm, wn) -> channel sums
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        run_px += min(16, Ho - 16 * ty) * min(16, Wo - 16 * tx);
        const int tid_o = wave * 64 + h5_lane_now();
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
        const int tid_o = wave * 64 + h5_lane_now();
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the thread reads back its own record)
    stage_halo(0, 0);
    stage_w(0, w_step);
    stage_w(1, w_step);
    wait_vm<B_IT>();     // halo 0 and the tile of tap 0 have landed
    __builtin_amdgcn_s_barrier();

    // ---- main loop over (tile, chunk), nine unrolled taps each --------------------------------------------------------------------------
    // Invariant at the top of step s = (cc, TAP): the barrier that certified tile s (and the halo of chunk cc) has been passed, tile s+1 is in
    // flight.  The step issues tile s+2 into slot (TAP + 2) % 3 (its previous content, tile s-1, was read before that barrier), at tap 0 the
    // halo of the next chunk, reads its eight fragments and runs its 16 MFMAs.
    int cc = 0;
    bool tile_end = cpt == 1;                                    // this chunk is the last of its tile
    bool final_ = tile_end && sp_cur + sp_stride >= tiles_sp;    // ... and of the workgroup
    Frags f;
    auto kstep = [&](auto tapc, auto parc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value;
        const bool issue_w = !(final_ && TAP >= 7), issue_h = TAP == 0 && !final_;
        const int fcc = tile_end ? 0 : cc + 1;  // chunk (of the fetch tile) staged at tap 0
        const int adv = (TAP + 2) % 9 == 8 ? (tile_end ? w_tile_wrap : w_wrap) : w_step;
        // role split: waves 4-7 issue their DMA before the MFMAs, waves 0-3 after -- except in a tile's last step, where a DMA issued first
        // would sit under the epilogue's vmcnt(0)
        const bool dma_first = second_half && !(TAP == 8 && tile_end);
        if (dma_first) {
            if (issue_w) stage_w((TAP + 2) % 3, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_frags(f, tapc, parc);
        mfma16(f);
        __builtin_amdgcn_sched_barrier(0);
        if (TAP == 8 && tile_end) {
            __builtin_amdgcn_s_barrier();  // every wave holds its last fragments: the finished chunk's halo buffer becomes the staging area
            wait_vm<0>();                  // everything this wave has in flight has landed: stores issued below cannot delay a certification
            epilogue(a_base + PAR * A_BUF + wave * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!dma_first) {
            if (issue_w) stage_w((TAP + 2) % 3, adv);
            if (issue_h) stage_halo(PAR ^ 1, fcc);
        }
        // barrier(s+1): tile s+1 (and every halo issued before it) must have landed; tile s+2 and, while it was issued in tap 0 of this chunk,
        // the halo of the next chunk may stay in flight
        if (TAP == 8 && final_) return;
        if (TAP <= 1) { if (!final_) wait_vm<A_IT + B_IT>(); else wait_vm<B_IT>(); }
        else if (TAP < 7) wait_vm<B_IT>();
        else if (TAP == 7) { if (final_) wait_vm<0>(); else wait_vm<B_IT>(); }
        else if (!tile_end) wait_vm<B_IT>();  // (tile end: certified by the vmcnt(0) ahead of the epilogue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](auto parc) __attribute__((always_inline)) {
        if (tile_end && !final_) setup_fetch(sp_cur + sp_stride);  // from here on halo staging belongs to the next tile
        kstep(IC<0>{}, parc); kstep(IC<1>{}, parc); kstep(IC<2>{}, parc);
        kstep(IC<3>{}, parc); kstep(IC<4>{}, parc); kstep(IC<5>{}, parc);
        kstep(IC<6>{}, parc); kstep(IC<7>{}, parc); kstep(IC<8>{}, parc);
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
        chunk(IC<0>{});
        if (sp_cur >= tiles_sp) break;
        chunk(IC<1>{});
        if (sp_cur >= tiles_sp) break;
    }
}

// The plain stride-1 3x3 convs halo3 takes, minus fused input transforms, x2 upsampling and fused activations (conv_halo4_applicable's set)
bool conv_halo5_applicable(const IGemmParams& p) {
    if (!conv_halo_applicable(p) || p.ups || p.in_scale || p.act != GP_ACT_NONE) return false;
    if ((p.n_store & 7) || (p.ldo & 7) || (p.res && ((p.ldres & 7) || p.ldres < p.n_store))) return false;  // whole 8-channel slots (halo_persistent)
    if ((p.Cin & 31) || p.Cin < 64) return false;
    return p.Ho >= 16 && p.Wo >= 16;
}

void launch_conv_halo5(const IGemmParams& p, int grid, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H5_LDS);
    });
    hipLaunchKernelGGL(conv3x3_halo5_kernel, dim3(grid), dim3(512), H5_LDS, s, p);
}

GP_SAT_TU(conv_halo5)  // fp16 build: address of this translation unit's saturation flag (common.h)
