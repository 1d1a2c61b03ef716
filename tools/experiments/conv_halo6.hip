// conv3x3 (stride 1, pad 1) as Winograd F(2, 3) ALONG X, direct taps along y -- 2/3 of the direct conv's MFMA work (r4 experiment; opt-in).
//
// Why this form: the pipeline is bound by energy per image, and fewer matrix flops is the one lever left.  Parity was measured
// first (tools/precision_ablation.py, rows *winograd1d*): with the TRANSFORMED operands rounded to 16 bits the final maps move by nothing
// (fp16 4.2e-4 vs 4.3e-4 mean |delta|).  The two-dimensional F(2x2, 3x3) needs 16 accumulator sets per (channel, patch) alive across the channel
// loop -- 256 registers per thread for a 16 x 16 x 128 tile; the one-dimensional form needs 4 = 128 registers, conv3x3_halo4_kernel's budget.
//
//   y[r][2q + {0,1}] = A^T sum_ky sum_c ( U[ky][p][c] . V[p][r + ky][q][c] ),  p = 0..3
//   V[p][hy][q] = B^T d[hy][2q .. 2q+3]:  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3        (the staged halo, transformed IN LDS per chunk)
//   U[ky][p]    = G g[ky][0..2]:          U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2   (once per weight: wino_weights_kernel)
//   A^T M:      y0 = M0 + M1 + M2, y1 = M1 - M2 - M3                                                     (in the epilogue, fp32)
//
//   Workgroup tile: 16 x 16 output pixels x 128 channels = 16 rows x 8 column pairs ("patches") x 4 positions.  8 waves = 4 groups of 4 rows x 2
//        channel halves; a wave owns 64 channels x 32 patches x 4 positions = 4 x 4 x 2 accumulator tiles of v_mfma_f32_16x16x32 (128 registers).
//   K chunk = 32 input channels; K-step = (ky, position pair) = two weight planes [128 cout][32 cin] of 8 KiB through a ring of SIX plane slots, six
//        steps per chunk, 16 MFMAs per wave and step; fragments of one plane per register set, the next plane always read under the current one's MFMAs (the direct conv: nine taps x 16 = 144 MFMAs per 32-channel chunk, here 96).
//   LDS: the raw 18 x 18 halo of a chunk (21 KiB, DMA target, single) -> transformed V[4][18][8] rows of 64 bytes (36 KiB, double-buffered): the
//        transform of chunk c+1 runs in steps 3 and 4 of chunk c (its halo was certified by the barrier that ended step 2).
//   Ring / waits / role split / persistence / statistics / epilogue staging: conv3x3_halo5_kernel's (conv_halo5.hip), restated for six steps.
//   LDS images (tests/test_lds_layout.py): V row R = (p * 18 + hy) * 8 + q, logical slot s at s ^ (3 * ((R >> 2) & 1)); weight row r the same with
//        key 3 * ((r >> 3) & 1); epilogue block [2 rows x 16 px][8 units of 16 B], unit u at u ^ ((px >> 1) & 7); the raw halo is not swizzled (its
//        four reads per item are 2-way conflicted: 144 LDS cycles per chunk).
// Weights: [n_rows][12 = 4 ky + p][Cin] 16-bit, produced from the packed direct weights by wino_weights_kernel (cached per weight pointer: test path).
#include "common.h"
#include "kernels.h"
#include <map>
#include <mutex>
#include <tuple>

constexpr int H6_HW = 18, H6_HROWS = 18 * 18;          // raw halo: 18 x 18 source pixels, 64-byte rows (32 channels)
constexpr int H6_GROUPS = (H6_HROWS + 15) / 16;        // 21 DMA pieces of 16 rows (1 KiB)
constexpr int H6_A_IT = (H6_GROUPS + 7) / 8;           // 3 DMA instructions per wave and halo (pieces 21..23 hit the dump KiB)
constexpr int H6_RAW = H6_GROUPS * 1024;               // 21 KiB
constexpr int H6_VROWS = 4 * 18 * 8;                   // transformed halo: [position][halo row][column pair] rows of 64 bytes
constexpr int H6_V_BUF = H6_VROWS * 64;                // 36 KiB
constexpr int H6_V_OFF = H6_RAW;
constexpr int H6_PLANE = 128 * 64;                     // one (ky, position) weight plane [128 cout][32 cin] 16-bit = 8 KiB; ring of six
constexpr int H6_B_OFF = H6_V_OFF + 2 * H6_V_BUF;
constexpr int H6_DUMP_OFF = H6_B_OFF + 6 * H6_PLANE;
constexpr int H6_ST_OFF = H6_DUMP_OFF + 1024;          // [8 waves][64 ch][sum, sumsq]
constexpr int H6_BIAS_OFF = H6_ST_OFF + 4096;          // [128] bias of the workgroup's channel slice
constexpr int H6_RUN_OFF = H6_BIAS_OFF + 512;          // [128 ch][sum, sumsq] running statistics of the workgroup
constexpr int H6_FETCH_OFF = H6_RUN_OFF + 1024;        // [512 threads][3 ints]: the fetch tile's halo source offsets (bytes; out-of-image: past the resource)
constexpr int H6_LDS = H6_FETCH_OFF + 512 * 12;        // 157 184 bytes
static_assert(H6_LDS <= 160 * 1024, "LDS");
static_assert(8 * 4096 <= H6_V_BUF, "the epilogue stages 4 KiB per wave in a released V buffer");

GP_DEV int h6_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
GP_DEV int h6_vkey(int row) { return 3 * ((row >> 2) & 1); }
GP_DEV int h6_wkey(int row) { return 3 * ((row >> 3) & 1); }
GP_DEV int h6_stg_key(int px) { return (px >> 1) & 7; }

// U[n][ky][p][c] = sum_kx G[p][kx] w[n][3 ky + kx][c] from the packed direct weight [n_rows][9][Cin] (fp32 arithmetic, one rounding)
__global__ __launch_bounds__(256) void wino_weights_kernel(const h16_t* __restrict__ w, h16_t* __restrict__ u, long long n_rows, int Cin, int ldw) {
    const long long n = (long long)n_rows * 3 * Cin;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % Cin);
        const long long r = idx / Cin;
        const int ky = (int)(r % 3);
        const long long row = r / 3;
        const h16_t* src = w + row * ldw + (long long)(3 * ky) * Cin + c;
        const float g0 = h16_to_f(src[0]), g1 = h16_to_f(src[Cin]), g2 = h16_to_f(src[2 * Cin]);
        h16_t* dst = u + row * (12LL * Cin) + (long long)(4 * ky) * Cin + c;
        dst[0] = f_to_h16(g0);
        dst[Cin] = f_to_h16(0.5f * ((g0 + g1) + g2));
        dst[2 * Cin] = f_to_h16(0.5f * ((g0 - g1) + g2));
        dst[3 * Cin] = f_to_h16(g2);
    }
}

__global__ __launch_bounds__(512) void conv3x3_halo6_kernel(const IGemmParams p, const h16_t* __restrict__ uw) {
    constexpr int BN = 128, NW = 8, A_IT = H6_A_IT, PLANE = H6_PLANE, HW_ = H6_HW, B_IT = 2, V_BUF = H6_V_BUF;
    constexpr int FN = 4, FJ = 2;  // per position: 4 x 16 channels (two 32-channel blocks) x 2 x 16 patches (two row pairs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const raw_lds = smem;
    char* const b_lds = smem + H6_B_OFF;
    char* const dump = smem + H6_DUMP_OFF;
    float* const s_st = (float*)(smem + H6_ST_OFF);
    float* const s_bias = (float*)(smem + H6_BIAS_OFF);

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
    const buf_rsrc_t in_rs = make_rsrc(in_b, (unsigned)Hi * (unsigned)Wi * (unsigned)Cin * 2u);
    const buf_rsrc_t w_rs = make_rsrc(uw, (unsigned)p.n_rows * (unsigned)(12 * Cin) * 2u);

    // ---- fetch state: the tile whose halo is being staged (one chunk ahead of the compute); per-thread records in LDS (conv_halo4.hip) -----
    typedef __attribute__((address_space(3))) int* lds_i_ptr;
    auto setup_fetch = [&](int sp) __attribute__((always_inline)) {
        const int fty = sp / tiles_x, ftx = sp - fty * tiles_x;
        const int sy0 = fty * 16 - 1, sx0 = ftx * 16 - 1;
        const int lane_o = h6_lane_now();
        const unsigned rec = (unsigned)(unsigned long long)(smem + H6_FETCH_OFF) + (unsigned)(wave * 64 + lane_o) * 12u;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = (wave + NW * i) * 16 + (lane_o >> 2);
            const int hy = r / HW_, hx = r - hy * HW_;
            const int iy = sy0 + hy, ix = sx0 + hx;
            const bool ok = r < H6_HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            const unsigned off = ((unsigned)((iy * Wi + ix) * Cin) + (unsigned)((lane_o & 3) << 3)) * 2u;  // byte offset of (pixel, 8-channel slot)
            *(lds_i_ptr)(rec + 4 * i) = ok ? (int)off : (int)0xfffffff0u;                                  // past the image: the range check returns zeros
        }
    };
    // weight rows n0 .. n0+127 always exist (conv_halo6_applicable checks n_rows); the 12 (ky, position) planes of a chunk are consecutive, two per step
    unsigned w_lane, w_uni = 0;
    {
        const int row = wave * 16 + (lane >> 2);
        w_lane = ((unsigned)(n0 + row) * (unsigned)(12 * Cin) + (unsigned)(((lane & 3) ^ h6_wkey(row)) << 3)) * 2u;
    }
    const int w_step = Cin, w_wrap = 32 - 11 * Cin, w_tile_wrap = -11 * Cin - (cpt - 1) * 32;  // next plane / next chunk / first chunk again (elements)

    const unsigned raw_base = (unsigned)(unsigned long long)raw_lds, v_base = (unsigned)(unsigned long long)(smem + H6_V_OFF);
    const unsigned b_base = (unsigned)(unsigned long long)b_lds;
    auto stage_halo = [&](int cc) __attribute__((always_inline)) {
        const unsigned fetch_rec = (unsigned)(unsigned long long)(smem + H6_FETCH_OFF) + (unsigned)(wave * 64 + h6_lane_now()) * 12u;
        int off[A_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) off[i] = *(lds_i_ptr)(fetch_rec + 4 * i);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int g = wave + NW * i;
            blds16(in_rs, (unsigned)off[i], (unsigned)(cc << 6), g < H6_GROUPS ? raw_lds + g * 1024 : dump);
        }
    };
    auto stage_plane = [&](int slot, int adv) __attribute__((always_inline)) {  // next weight plane in (tile, chunk, ky, position) order, then advance
        blds16(w_rs, w_lane, w_uni, b_lds + slot * PLANE + wave * 1024);       // rows wave * 16 .. + 15 of the plane
        w_uni += (unsigned)(adv * 2);
    };
    // ---- B^T d: the raw halo of the chunk just certified -> V buffer `vb`; item = (halo row hy, column pair q, 8-channel slot) --------------------
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_t* lds_u4_ptr;
    struct TItem { u32x4_t d0, d1, d2, d3; };
    auto transform_load = [&](TItem& t, int item) __attribute__((always_inline)) {
        const int hy = item >> 5, q = (item >> 2) & 7, slot = item & 3;
        const unsigned src = raw_base + (unsigned)((hy * HW_ + 2 * q) * 64 + slot * 16);
        t.d0 = *(lds_u4_ptr)src; t.d1 = *(lds_u4_ptr)(src + 64); t.d2 = *(lds_u4_ptr)(src + 128); t.d3 = *(lds_u4_ptr)(src + 192);
    };
    auto transform_finish = [&](const TItem& t, int vb, int item) __attribute__((always_inline)) {
        const int hy = item >> 5, q = (item >> 2) & 7, slot = item & 3;
        u32x4_t v0, v1, v2, v3;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a0 = h16_lo(t.d0[w]), b0 = h16_hi(t.d0[w]), a1 = h16_lo(t.d1[w]), b1 = h16_hi(t.d1[w]);
            const float a2 = h16_lo(t.d2[w]), b2 = h16_hi(t.d2[w]), a3 = h16_lo(t.d3[w]), b3 = h16_hi(t.d3[w]);
            v0[w] = pack_h16x2(a0 - a2, b0 - b2);  // (saturating in the fp16 build: |d0 - d2| may leave the fp16 range where d0, d2 do not)
            v1[w] = pack_h16x2(a1 + a2, b1 + b2);
            v2[w] = pack_h16x2(a2 - a1, b2 - b1);
            v3[w] = pack_h16x2(a1 - a3, b1 - b3);
        }
        const int r0 = hy * 8 + q;  // row of position 0; position p: + p * 144 rows (the key only depends on q)
        const unsigned dst = v_base + (unsigned)(vb * V_BUF + r0 * 64 + ((slot ^ h6_vkey(r0)) << 4));
        *(lds_u4_ptr)dst = v0;
        *(lds_u4_ptr)(dst + 144 * 64) = v1;
        *(lds_u4_ptr)(dst + 2 * 144 * 64) = v2;
        *(lds_u4_ptr)(dst + 3 * 144 * 64) = v3;
    };
    auto transform_item = [&](int vb, int item) __attribute__((always_inline)) {
        TItem t;
        transform_load(t, item);
        transform_finish(t, vb, item);
    };

    f32x4_t acc[4][FN][FJ];

    // Fragments of ONE position (plane): two register sets.  Step s computes position pair (2 t, 2 t + 1): `f0` (plane 2 s, read during the second
    // half of step s-1) first, `f1` (plane 2 s + 1, read at the start of step s) second; while the second half runs, f0 is refilled for step s+1.
    struct HalfFrags { h16x8_t w[FN], x[FJ]; };
    unsigned xb, wb;
    auto frag_bases = [&]() __attribute__((always_inline)) {  // (re)computed after every epilogue: values that live ACROSS it end up in scratch
        const int lane_o = h6_lane_now();
        const int a15 = lane_o & 15, qk = lane_o >> 4;
        wb = b_base + (wn * 64 + 8 * (a15 >> 2) + (a15 & 3)) * 64 + ((qk ^ h6_wkey(8 * (a15 >> 2))) << 4);
        // patch a15 of a fragment: row (a15 >> 3) of a row pair, column pair a15 & 7
        xb = v_base + ((4 * wm + (a15 >> 3)) * 8 + (a15 & 7)) * 64 + ((qk ^ h6_vkey(a15 & 7)) << 4);
    };
    frag_bases();
    // plane index PL within the chunk stream (0..11 this chunk, 12 = plane 0 of the next chunk, whose V is the other buffer): ky = (PL % 12) / 4,
    // position = PL % 4, ring slot PL % 6
    auto load_half = [&](HalfFrags& f, auto plc, auto parc) __attribute__((always_inline)) {
        constexpr int PL = decltype(plc)::value, PAR = decltype(parc)::value ^ (PL >= 12 ? 1 : 0), PI = PL % 12, KY = PI >> 2, POS = PI & 3, SLOT = PL % 6;
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = lds_frag(wb, SLOT * PLANE + (32 * (i >> 1) + 4 * (i & 1)) * 64);
#pragma unroll
        for (int j = 0; j < FJ; ++j) f.x[j] = lds_frag(xb, PAR * V_BUF + ((POS * 18 + 2 * j + KY) * 8) * 64);
    };
    auto mfma8 = [&](const HalfFrags& f, auto posc) __attribute__((always_inline)) {
        constexpr int POS = decltype(posc)::value;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[POS][i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[POS][i][j]);
    };

    if (tid < BN) {
        s_bias[tid] = (p.bias && p.bias_mode == GP_BIAS_COL && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
        ((float*)(smem + H6_RUN_OFF))[2 * tid] = 0.f;
        ((float*)(smem + H6_RUN_OFF))[2 * tid + 1] = 0.f;
    }
    __syncthreads();

    // ---- per-wave epilogue of the finished tile (sp_cur); `stg` = 4 KiB private LDS window ---------------------------------------------
    const int n_out = p.N;
    const bool want_stats = p.stats_out != nullptr;
    const unsigned st_base = (unsigned)(unsigned long long)s_st, bias_base = (unsigned)(unsigned long long)s_bias;
    auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pz = 0; pz < 4; ++pz)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[pz][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto epilogue_body = [&](unsigned stg, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr bool RES = decltype(resc)::value != 0, STATS = decltype(statc)::value != 0;
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        const int lane_o = h6_lane_now();
        const int aw = lane_o & 15, qw = lane_o >> 4;       // write role: patch (row aw >> 3 of the pair, column pair aw & 7), 8-channel group of the block
        const int px0 = lane_o >> 2, sl8 = lane_o & 3;      // read-back role: pixel px0 of a tile row, 8-channel slot sl8 of the block
        h16_t* outp = (h16_t*)p.out;
        const int ox = tx * 16 + px0, oy0 = ty * 16 + 4 * wm;
        const int col0 = n0 + wn * 64 + 8 * sl8;
        auto row_index = [&](int rr) __attribute__((always_inline)) { return (oy0 + rr < Ho && ox < Wo) ? (b * Ho + oy0 + rr) * Wo + ox : -1; };
        const unsigned wr = stg + (unsigned)(((aw >> 3) * 16 + 2 * (aw & 7)) * 128);  // staging row of pixel x = 2 q of this lane's patch row; x + 1: + 128
        const unsigned wk0 = (unsigned)h6_stg_key((aw >> 3) * 16 + 2 * (aw & 7)), wk1 = wk0;  // (px >> 1) & 7 is the same for x = 2 q and 2 q + 1
        float satm = 0.f;
        const unsigned ba = bias_base + (wn * 64 + 8 * qw) * 4;
        uint4 rv[4];  // residual rows of the block being stored (four output rows of this lane's pixel column), loaded a block ahead
        auto load_res = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = row_index(rr), col = col0 + 32 * c;
                rv[rr] = make_uint4(0u, 0u, 0u, 0u);
                if (m >= 0 && col < p.n_store) rv[rr] = *(const uint4*)(p.res + (long long)m * p.ldres + col);
            }
        };
        if (RES) load_res(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = col0 + 32 * c;
            const bool col_ok = col < p.n_store;
            unsigned tmask[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) tmask[w] = (col + 2 * w < n_out ? 0xffffu : 0u) | (col + 2 * w + 1 < n_out ? 0xffff0000u : 0u);
            const f32x4_t bv0 = *(lds_f4_ptr)(ba + (32 * c) * 4), bv1 = *(lds_f4_ptr)(ba + (32 * c + 4) * 4);  // bias of channels 8 qw .. + 7 of the block
            float st_s[8], st_q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
            for (int jj = 0; jj < FJ; ++jj) {
                // A^T M: y0 = M0 + M1 + M2, y1 = M1 - M2 - M3 (+ bias); channels 8 qw .. 8 qw + 7 of the block: fragments 2 c (low four) and 2 c + 1
                const f32x4_t y0a = ((acc[0][2 * c][jj] + acc[1][2 * c][jj]) + acc[2][2 * c][jj]) + bv0;
                const f32x4_t y0b = ((acc[0][2 * c + 1][jj] + acc[1][2 * c + 1][jj]) + acc[2][2 * c + 1][jj]) + bv1;
                const f32x4_t y1a = ((acc[1][2 * c][jj] - acc[2][2 * c][jj]) - acc[3][2 * c][jj]) + bv0;
                const f32x4_t y1b = ((acc[1][2 * c + 1][jj] - acc[2][2 * c + 1][jj]) - acc[3][2 * c + 1][jj]) + bv1;
                *(lds_f4_ptr)(wr + (((2 * qw) ^ wk0) << 4)) = y0a;
                *(lds_f4_ptr)(wr + (((2 * qw + 1) ^ wk0) << 4)) = y0b;
                *(lds_f4_ptr)(wr + 128 + (((2 * qw) ^ wk1) << 4)) = y1a;
                *(lds_f4_ptr)(wr + 128 + (((2 * qw + 1) ^ wk1) << 4)) = y1b;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int pxs = rb * 16 + px0;
                    const unsigned rd = stg + (unsigned)(pxs * 128), rk = (unsigned)h6_stg_key(pxs);
                    const f32x4_t x0 = *(lds_f4_ptr)(rd + (((2 * sl8) ^ rk) << 4)), x1 = *(lds_f4_ptr)(rd + (((2 * sl8 + 1) ^ rk) << 4));
                    const long long m = row_index(2 * jj + rb);
                    if (m >= 0 && col_ok) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        if (RES) {
                            const uint4 r4 = rv[2 * jj + rb];
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
                }
            }
            if (RES && c == 0) load_res(1);  // (the statistics reduction below is its cover)
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
    int run_px = 0;
    const unsigned run_base = (unsigned)(unsigned long long)(smem + H6_RUN_OFF);
    auto flush_stats = [&]() __attribute__((always_inline)) {
        const int ty = sp_cur / tiles_x, tx = sp_cur - ty * tiles_x;
        run_px += min(16, Ho - 16 * ty) * min(16, Wo - 16 * tx);
        const int tid_o = wave * 64 + h6_lane_now();
        if (tid_o < BN) {
            const unsigned a = st_base + (unsigned)tid_o * 8u;
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
        const int tid_o = wave * 64 + h6_lane_now();
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
    stage_halo(0);
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) stage_plane(pl, w_step);   // planes 0 .. 4 (cpt >= 2: all of the first chunk)
    wait_vm<B_IT>();     // the raw halo of chunk 0 and planes 0 .. 2 have landed
    __builtin_amdgcn_s_barrier();
    transform_item(0, tid);
    if (tid < 18 * 32 - 512) transform_item(0, tid + 512);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    HalfFrags f0, f1;
    load_half(f0, IC<0>{}, IC<0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave holds plane 0's fragments: step 0 refills that slot (plane 6)

    // ---- main loop over (tile, chunk), six unrolled steps each ----------------------------------------------------------------------------
    // Invariant at the top of step s = (cc, T): f0 holds the fragments of plane 2 s (read during step s-1), planes <= 2 s + 2 are certified, planes
    // 2 s + 3 and 2 s + 4 in flight.  The step issues planes 2 s + 5 and 2 s + 6 into the slots planes 2 s - 1 and 2 s left at the last barrier, at
    // T = 0 the raw halo of the next chunk (certified by the barrier that ends step 2, transformed in steps 3 and 4 into the other V buffer), reads
    // plane 2 s + 1 into f1 at its start (eight MFMAs of cover) and plane 2 s + 2 into f0 under its second half.
    int cc = 0;
    bool tile_end = cpt == 1;
    bool final_ = tile_end && sp_cur + sp_stride >= tiles_sp;
    auto kstep = [&](auto tc, auto parc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value, PAR = decltype(parc)::value;
        // planes 2 T + 5 (index 11 of this chunk at T = 3: the walk wraps behind it) and 2 T + 6; the workgroup's last chunk has no planes >= 12
        const bool issue_a = !(final_ && T >= 4), issue_b = !(final_ && T >= 3), issue_h = T == 0 && !final_;
        const int fcc = tile_end ? 0 : cc + 1;  // chunk (of the fetch tile) staged at step 0
        const int adv_a = T == 3 ? (tile_end ? w_tile_wrap : w_wrap) : w_step;
        const bool dma_first = second_half && !(T == 5 && tile_end);
        if (dma_first) {
            if (issue_a) stage_plane((2 * T + 5) % 6, adv_a);
            if (issue_b) stage_plane((2 * T + 6) % 6, w_step);
            if (issue_h) stage_halo(fcc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (T == 3 && !final_) {
            // the next chunk's B^T d: the item's four raw reads go out with plane 2 s + 1's fragment reads, its ~70 VALU instructions run next to the
            // first half's MFMAs
            TItem ti;
            const int item = wave * 64 + h6_lane_now();
            transform_load(ti, item);
            load_half(f1, IC<2 * T + 1>{}, parc);
            mfma8(f0, IC<2 * (T & 1)>{});
            transform_finish(ti, PAR ^ 1, item);
        } else {
            load_half(f1, IC<2 * T + 1>{}, parc);
            mfma8(f0, IC<2 * (T & 1)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool prefetch = !(T == 5 && tile_end);  // (tile end: after the epilogue; the workgroup's last step: nothing follows)
        if (prefetch) load_half(f0, IC<2 * T + 2>{}, parc);
        mfma8(f1, IC<2 * (T & 1) + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (T == 4 && !final_ && wave == 0) transform_item(PAR ^ 1, 512 + h6_lane_now());  // the 64 left-over items (uniform branch)
        if (T == 5 && tile_end) {
            __builtin_amdgcn_s_barrier();  // every wave holds its last fragments: the finished chunk's V buffer becomes the staging area
            wait_vm<0>();
            epilogue(v_base + PAR * V_BUF + wave * 4096);
            if (!final_) load_half(f0, IC<12>{}, parc);  // plane 0 of the next tile (landed: vmcnt(0) above), its V in the other buffer
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!dma_first) {
            if (issue_a) stage_plane((2 * T + 5) % 6, adv_a);
            if (issue_b) stage_plane((2 * T + 6) % 6, w_step);
            if (issue_h) stage_halo(fcc);
        }
        // barrier(s+1): planes 2 s + 3 and 2 s + 4 (and every halo issued before them) must have landed; the two planes of this step and, while it
        // was issued in step 0 of this chunk, the halo of the next chunk may stay in flight.  The workgroup's last chunk stops issuing: drain.
        if (T == 5 && final_) return;
        if (final_ && T >= 3) wait_vm<0>();
        else if (T <= 1) { if (!final_) wait_vm<A_IT + B_IT>(); else wait_vm<B_IT>(); }
        else if (!(T == 5 && tile_end)) wait_vm<B_IT>();  // (tile end: certified by the vmcnt(0) ahead of the epilogue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](auto parc) __attribute__((always_inline)) {
        if (tile_end && !final_) setup_fetch(sp_cur + sp_stride);
        kstep(IC<0>{}, parc); kstep(IC<1>{}, parc); kstep(IC<2>{}, parc);
        kstep(IC<3>{}, parc); kstep(IC<4>{}, parc); kstep(IC<5>{}, parc);
        if (tile_end) {
            if (want_stats) {
                if (final_) __syncthreads();
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

// conv3x3_halo5_kernel's set (plain stride-1 3x3 convs, no fused input transform / upsample / activation)
bool conv_halo6_applicable(const IGemmParams& p) {
    if (!conv_halo_applicable(p) || p.ups || p.in_scale || p.act != GP_ACT_NONE) return false;
    if ((p.n_store & 7) || (p.ldo & 7) || (p.res && ((p.ldres & 7) || p.ldres < p.n_store))) return false;
    if ((p.Cin & 31) || p.Cin < 64 || p.ldw != 9 * p.Cin) return false;
    if ((long long)p.n_rows * 12 * p.Cin * 2 >= 0xfffffff0ll) return false;
    return p.Ho >= 16 && p.Wo >= 16;
}

// Transformed weights of a packed direct weight, made on first use and kept for the life of the process (experiment / test path: the engine would
// pack U from the fp32 checkpoint once, at gp_finalize)
static const h16_t* wino_weights_for(const IGemmParams& p, hipStream_t s) {
    static std::mutex mu;
    static std::map<std::tuple<const void*, int, int>, h16_t*> cache;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple((const void*)p.wt, p.n_rows, p.Cin);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    h16_t* u = nullptr;
    if (hipMalloc((void**)&u, (size_t)p.n_rows * 12 * p.Cin * sizeof(h16_t)) != hipSuccess) return nullptr;
    const long long n = (long long)p.n_rows * 3 * p.Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, p.wt, u, (long long)p.n_rows, p.Cin, p.ldw);
    cache[key] = u;
    return u;
}

void launch_conv_halo6(const IGemmParams& p, int grid, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H6_LDS);
    });
    const h16_t* uw = wino_weights_for(p, s);
    if (!uw) return;
    hipLaunchKernelGGL(conv3x3_halo6_kernel, dim3(grid), dim3(512), H6_LDS, s, p, uw);
}

GP_SAT_TU(conv_halo6)  // fp16 build: address of this translation unit's saturation flag (common.h)
