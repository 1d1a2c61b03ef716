// Persistent GEMM  out[m][n] = sum_k A[m][k] * W[n][k] + bias (+ res)  for the short-K linear layers / 1x1 convolutions of the UNet
// transformer blocks (K = 320 ... 2560, i.e. 5 ... 40 K-steps per tile) and the VAE's 1x1 projections.
//
// With one tile per workgroup (igemm.hip) these launches were fixed cost, not matrix work: for M = 36864, N = K = 320 the kernel
// took 23.7 us, of which 10 us remained with neither K loop nor epilogue (launch of 1440 workgroups, address setup, the first HBM
// round trip) and 5-8 us were an epilogue that nothing overlapped.  Here (same recipe as conv3x3_halo3_kernel):
//   * one workgroup per CU walks the M tiles of ONE 128-column slice: bias and weight rows are per-workgroup constants;
//   * the (tile, k) step stream never stops: a 3-deep LDS-DMA ring (A tile BM x 64 + W tile 128 x 64 per stage) runs three steps ahead
//     ACROSS tile boundaries, so a tile's first operands arrive while the previous tile still computes;
//   * ring slot = step % 3 is a compile-time constant (loop unrolled by three): every ds_read_b128 uses an immediate offset;
//     fragments of step s+1 are read during step s's two MFMA batches (register sets dead in that batch);
//   * per-WAVE epilogue, no workgroup barrier: fp32 staging through the LDS pieces the wave itself will refill next (its own DMA
//     destinations of the slot just consumed), coalesced 16-byte stores, residual rows prefetched; stores drain under the next tile;
//   * optional GroupNorm partial statistics per (M tile, channel) as in epilogue.h.
// Not handled here (launch_igemm keeps them on igemm_kernel): fp32 output, batched problems, row bias, odd strides.
#include "common.h"
#include "kernels.h"

constexpr int PG_BN = 128;

template <int BM, int NB = 3>
struct PGemmGeom {
    static constexpr int WM = BM / 64, WN = 8 / WM;      // 8 waves: 4 x 2 (BM 256) or 2 x 4 (BM 128); wave tile 64 rows x TN columns
    static constexpr int TN = PG_BN / WN, FM = 4, FN = TN / 16, FP = FN / 2;
    static constexpr int A_IT = BM / 64, B_IT = 2, LPS = A_IT + B_IT;
    static constexpr int A_BYTES = BM * 128, STAGE = (BM + PG_BN) * 128;
    static constexpr int ST_OFF = NB * STAGE;             // [8 waves][TN][2] partial statistics (<= 4 KiB)
    static constexpr int BIAS_OFF = ST_OFF + 4096;        // [128] bias of the column slice
    static constexpr int LDS = BIAS_OFF + 512;
};

// NB: ring depth (stages in flight ahead of the compute).  3 everywhere in r2 / r3; r4: 4 for the 128-row tile (4 x 32 KiB), whose launches are the
// latency-bound ones -- M = 2304 / 9216 with one or two tiles per workgroup and 10-40 K-steps each, ~1.5 us per K-step against 0.2 us of MFMA work:
// with a prefetch distance of three steps a stage has 1.5x as long to arrive.
// ABL: compile-time ablations for profiling (1: LDS-DMA as global_load_lds, the r2 / r3 form; 2: no MFMA, 4: no output stores, 8: no DMA)
template <int BM, int ABL = 0, int NB = 3>
__global__ __launch_bounds__(512) void pgemm_kernel(const IGemmParams p) {
    using G = PGemmGeom<BM, NB>;
    constexpr int WN = G::WN, TN = G::TN, FM = G::FM, FN = G::FN, FP = G::FP;
    constexpr int A_IT = G::A_IT, B_IT = G::B_IT, LPS = G::LPS, A_BYTES = G::A_BYTES, STAGE = G::STAGE;
    constexpr int SLW = TN / 8;          // 8-channel slots per staged row
    constexpr int RPI = 64 / SLW;        // rows one read-back instruction covers (8 or 16)
    constexpr int NH = 16 / RPI;         // read-back instructions per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool second_half = wave >= 4;
    const int a15 = lane & 15;

    // ---- work assignment: column slice nt, M tiles g, g + GR, ... ------------------------------------------------------------------
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + PG_BN - 1) / PG_BN, tiles_m = (p.M + BM - 1) / BM;
    const int GR = gridDim.x / tiles_n;                   // M-tile groups
    const int G8 = GR & ~7;
    int g, nt;
    {
        const int id = blockIdx.x;
        if (id < G8 * tiles_n) { g = (id & 7) + 8 * ((id >> 3) / tiles_n); nt = (id >> 3) % tiles_n; }  // a group's slices share an XCD (id % 8)
        else { const int r = id - G8 * tiles_n; g = G8 + r / tiles_n; nt = r % tiles_n; }
    }
    const int n0 = nt * PG_BN;
    const int nk = p.Cin >> 6;
    const int my_tiles = (tiles_m - g + GR - 1) / GR;
    const int total = my_tiles * nk;                      // steps of this workgroup
    const int tile_adv = GR * BM;                         // rows between two of my tiles

    // ---- DMA sources: A rows advance 64 elements per step, by a_wrap at the end of a tile; W rows wrap back -----------------------
    const int chunk_a = (lane & 7) ^ (lane >> 3);
    const int chunk_w = (lane & 7) ^ (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2));
    const h16_t* aq[A_IT];
    const h16_t* wq[B_IT];
    unsigned a_ok = 0, w_ok = 0;
    int im0 = g * BM;                                     // first row of the tile being STAGED (runs ahead of the compute)
    int ikt = 0;
    auto set_a_ok = [&]() __attribute__((always_inline)) {
        a_ok = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (im0 + (wave + 8 * i) * 8 + (lane >> 3) < p.M) a_ok |= 1u << i;
    };
#pragma unroll
    for (int i = 0; i < A_IT; ++i) aq[i] = p.in + (long long)(im0 + (wave + 8 * i) * 8 + (lane >> 3)) * p.lda + chunk_a * 8;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + (wave + 8 * i) * 8 + (lane >> 3);
        if (n < p.n_rows) w_ok |= 1u << i;
        wq[i] = p.wt + (long long)n * p.ldw + chunk_w * 8;
    }
    set_a_ok();
    const long long a_wrap = (long long)tile_adv * p.lda - (nk - 1) * 64;
    const int w_wrap = -(nk - 1) * 64;
    const h16_t* zsrc = p.zero;
    // r4: LDS-DMA through buffer resources (common.h blds16; conv_halo.hip has the measurement: 2-4 % on every conv shape): a 32-bit lane offset
    // (row, 16-byte slot) + the k offset as the uniform operand, no 64-bit pointer arithmetic per piece; rows past M / past n_rows are past the
    // resource's size, so the hardware's range check writes the zeros the pointer-to-a-zero-page select used to fetch.  ABL & 1: the FLAT form.
    constexpr bool MUBUF = !(ABL & 1);
    const buf_rsrc_t a_rs = make_rsrc(p.in, (unsigned)(((long long)(p.M - 1) * p.lda + p.Cin) * 2));
    const buf_rsrc_t w_rs = make_rsrc(p.wt, (unsigned)((long long)p.n_rows * p.ldw * 2));
    unsigned a_lane[A_IT], w_lane[B_IT], k_uni = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_lane[i] = (unsigned)(im0 + (wave + 8 * i) * 8 + (lane >> 3)) * (unsigned)(p.lda * 2) + (unsigned)(chunk_a * 16);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_lane[i] = (unsigned)(n0 + (wave + 8 * i) * 8 + (lane >> 3)) * (unsigned)(p.ldw * 2) + (unsigned)(chunk_w * 16);
    const unsigned a_tile_adv = (unsigned)tile_adv * (unsigned)(p.lda * 2);
    auto stage = [&](int slot) __attribute__((always_inline)) {  // next stage in (tile, k) order into ring slot `slot`
        char* sb = smem + slot * STAGE;
        const bool wrap = ikt == nk - 1;
        if (MUBUF) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (!(ABL & 8)) blds16(a_rs, a_lane[i], k_uni, sb + (wave + 8 * i) * 1024);
                if (wrap) a_lane[i] += a_tile_adv;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                if (!(ABL & 8)) blds16(w_rs, w_lane[i], k_uni, sb + A_BYTES + (wave + 8 * i) * 1024);
            if (wrap) { ikt = 0; k_uni = 0; } else { ++ikt; k_uni += 128u; }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const h16_t* src = ((a_ok >> i) & 1u) ? aq[i] : zsrc;
            if (!(ABL & 8)) glds16(src, sb + (wave + 8 * i) * 1024);
            aq[i] += wrap ? a_wrap : 64;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const h16_t* src = ((w_ok >> i) & 1u) ? wq[i] : zsrc;
            if (!(ABL & 8)) glds16(src, sb + A_BYTES + (wave + 8 * i) * 1024);
            wq[i] += wrap ? w_wrap : 64;
        }
        if (wrap) { ikt = 0; im0 += tile_adv; set_a_ok(); } else ++ikt;
    };

    // ---- constants of the workgroup: bias slice in LDS --------------------------------------------------------------------------------
    float* const s_bias = (float*)(smem + G::BIAS_OFF);
    if (tid < PG_BN) s_bias[tid] = (p.bias && p.bias_mode == GP_BIAS_COL && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
    __syncthreads();
    const unsigned smem_base = (unsigned)(unsigned long long)smem;
    const unsigned st_base = smem_base + G::ST_OFF, bias_base = smem_base + G::BIAS_OFF;

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- fragment bases (LDS byte addresses); slot, fragment row block and k-half offsets are immediates ----------------------------
    struct Half { h16x8_t w[FN], x[FM]; };
    unsigned xb[2], wb[2];
    {
        const int xr_w = ((a15 >> 1) & 1) | (((a15 >> 2) & 1) << 1) | (((a15 >> 3) & 1) << 2);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 4 + (lane >> 4);
            xb[kk] = smem_base + (wm * 64 + a15) * 128 + ((sl ^ (a15 & 7)) << 4);
            wb[kk] = smem_base + A_BYTES + (wn * TN + 8 * (a15 >> 2) + (a15 & 3)) * 128 + ((sl ^ xr_w) << 4);
        }
    }
    auto load_half = [&](Half& f, auto slotc, auto kkc) __attribute__((always_inline)) {
        constexpr int S = decltype(slotc)::value, KK = decltype(kkc)::value;
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = lds_frag(wb[KK], S * STAGE + (i >> 1) * 4096 + (i & 1) * 512);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.x[j] = lds_frag(xb[KK], S * STAGE + j * 2048);
    };
    auto mfma_half = [&](const Half& f) __attribute__((always_inline)) {
        if (ABL & 2) {
            asm volatile("" ::"v"(f.w[0]), "v"(f.w[FN - 1]), "v"(f.x[0]), "v"(f.x[1]), "v"(f.x[2]), "v"(f.x[3]));
            return;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mfma_16x16x32(f.w[i], f.x[j], acc[i][j]);
    };
    auto interleave = [&]() __attribute__((always_inline)) {  // FN*FM MFMAs with FN+FM LDS reads between them
        if (FN == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };

    // ---- per-wave epilogue of the finished tile (rows cm0 ...), staged through the wave's own DMA pieces of ring slot S ---------------
    int cm0 = g * BM;                                     // first row of the tile being COMPUTED
    const int n_out = p.N;
    const bool want_stats = p.stats_out != nullptr;
    auto epilogue_body = [&](auto slotc, auto actc, auto resc, auto statc) __attribute__((always_inline)) {
        constexpr int S = decltype(slotc)::value;
        constexpr bool ACT = decltype(actc)::value != 0, RES = decltype(resc)::value != 0, STATS = decltype(statc)::value != 0;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque copy: keeps this address arithmetic out of the K loop's live registers (see conv_halo.hip)
        const int q = lane_o >> 4, a = lane_o & 15;
        const int pl = lane_o / SLW, sl8 = lane_o % SLW;  // read-back role: rows pl (+ RPI), channel slot sl8
        const int col = n0 + wn * TN + 8 * sl8;
        const bool col_ok = col < p.n_store;
        const bool tail = col + 7 >= n_out;
        h16_t* outp = (h16_t*)p.out;
        // window: staged row r (0..15) of TN fp32 lives in DMA piece r / RPP of this wave: slot base + (wave + 8 * piece) KiB
        constexpr int RPP = 1024 / (TN * 4);
        const unsigned win = smem_base + S * STAGE + wave * 1024;
        int m2[FM][NH];
        uint4 rv[FM][NH];
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int m = cm0 + wm * 64 + 16 * j + pl + RPI * h;
                m2[j][h] = (m < p.M && col_ok) ? m : -1;
                if (RES) {
                    rv[j][h] = make_uint4(0u, 0u, 0u, 0u);
                    if (m2[j][h] >= 0 && p.res) rv[j][h] = *(const uint4*)(p.res + (long long)m2[j][h] * p.ldres + col);
                }
            }
        f32x4_t bv[FP][2];
#pragma unroll
        for (int ip = 0; ip < FP; ++ip) {
            bv[ip][0] = *(lds_f4_ptr)(bias_base + (wn * TN + 32 * ip + 8 * q) * 4);
            bv[ip][1] = *(lds_f4_ptr)(bias_base + (wn * TN + 32 * ip + 8 * q) * 4 + 16);
        }
        float st_s[8], st_q[8];
        float satm = 0.f;  // fp16 build: max |value| this thread packs in this tile (common.h: sat_track / sat_report)
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
#pragma unroll
            for (int ip = 0; ip < FP; ++ip) {
                const unsigned d = win + (a / RPP) * 8192 + (a % RPP) * (TN * 4) + ((((4 * ip + q) ^ (a & (SLW - 1)))) << 5);
                *(lds_f4_ptr)d = acc[2 * ip][j] + bv[ip][0];
                *(lds_f4_ptr)(d + 16) = acc[2 * ip + 1][j] + bv[ip][1];
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int pr = pl + RPI * h;
                const unsigned sa = win + (pr / RPP) * 8192 + (pr % RPP) * (TN * 4) + ((sl8 ^ (pr & (SLW - 1))) << 5);
                const f32x4_t x0 = *(lds_f4_ptr)sa, x1 = *(lds_f4_ptr)(sa + 16);
                const long long m = m2[j][h];
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
                    if (tail) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (col + e >= n_out) v[e] = 0.f;
                    }
                    uint4 pk;
                    pk.x = pack_h16x2_t(v[0], v[1], satm); pk.y = pack_h16x2_t(v[2], v[3], satm); pk.z = pack_h16x2_t(v[4], v[5], satm); pk.w = pack_h16x2_t(v[6], v[7], satm);
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
            for (int e = 0; e < 8; ++e) { st_s[e] = slot_sum<SLW>(st_s[e]); st_q[e] = slot_sum<SLW>(st_q[e]); }
            if (lane_o < SLW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    *(lds_f_ptr)(st_base + ((wave * TN + 8 * lane_o + e) * 2) * 4) = st_s[e];
                    *(lds_f_ptr)(st_base + ((wave * TN + 8 * lane_o + e) * 2 + 1) * 4) = st_q[e];
                }
            }
        }
    };
    // GEGLU (the transformer's first feed-forward GEMM): packed weight rows interleave value and gate so that a lane's 8 columns of a
    // fragment pair are 4 values + their 4 gates (engine: geglu_row); out[m][c] = value * gelu(gate), 4 outputs = 8 bytes per lane,
    // written straight from the accumulators (no staging: the 32-byte row pieces are what igemm_kernel's direct path wrote as well).
    auto epilogue_geglu = [&](auto slotc) __attribute__((always_inline)) {
        // r3: the 4 outputs per (fragment pair, pixel fragment) used to go straight to HBM as 8-byte stores, sixteen per lane, each instruction
        // touching sixteen rows with 32 bytes: 45 of the 100 us of the 36864 x 2560 x 320 GEMM were those stores (ablation: 97 -> 54 us without
        // them, while the staged 16-byte stores of the plain epilogue cost 17 us for twice the bytes).  Now the packed outputs of the wave's
        // 64 x TN/2 block are staged through the wave's own DMA pieces of the slot just consumed (like epilogue_body) and leave as 16-byte
        // stores covering the wave's whole column span of a row (64 / 32 bytes) per row.
        constexpr int S = decltype(slotc)::value;
        constexpr int RB = TN;                 // bytes of one staged row: TN / 2 outputs x 2 bytes
        constexpr int RPP = 1024 / RB;         // staged rows per 1 KiB DMA piece
        constexpr int UPR = RB / 16;           // 16-byte units per row
        constexpr int NIT = 64 * UPR / 64;     // read-back iterations (64 rows x UPR units / 64 lanes)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int q = lane_o >> 4, a = lane_o & 15;
        const int n_half = p.N >> 1;
        h16_t* outp = (h16_t*)p.out;
        const unsigned win = smem_base + S * STAGE + wave * 1024;
        float satm = 0.f;
#pragma unroll
        for (int ip = 0; ip < FP; ++ip) {
            const f32x4_t bl = *(lds_f4_ptr)(bias_base + (wn * TN + 32 * ip + 8 * q) * 4);
            const f32x4_t bh = *(lds_f4_ptr)(bias_base + (wn * TN + 32 * ip + 8 * q) * 4 + 16);
            const int col = ((n0 + wn * TN + 32 * ip) >> 1) + 4 * q;   // first of this lane's 4 output columns
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const f32x4_t val = acc[2 * ip][j] + bl, gate = acc[2 * ip + 1][j] + bh;
                float v[4] = {val.x * gelu_erf_f(gate.x), val.y * gelu_erf_f(gate.y), val.z * gelu_erf_f(gate.z), val.w * gelu_erf_f(gate.w)};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r >= n_half) v[r] = 0.f;
                const uint2 pk = pack_h16x4_t(v[0], v[1], v[2], v[3], satm);
                const int r16 = 16 * j + a;    // staged row; 8-byte slot ip * 4 + q, XORed with an even key of the row (2-way on ds_write_b64 at most)
                const unsigned d = win + (r16 / RPP) * 8192 + (r16 % RPP) * RB + (((ip * 4 + q) ^ (((r16 >> 1) & (UPR - 1)) << 1)) << 3);
                asm volatile("ds_write_b64 %0, %1" ::"v"(d), "v"(pk) : "memory");
            }
        }
        sat_report(satm);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int x = t * 64 + lane_o, r16 = x / UPR, c16 = x % UPR;
            const unsigned sa = win + (r16 / RPP) * 8192 + (r16 % RPP) * RB + (((2 * c16) ^ (((r16 >> 1) & (UPR - 1)) << 1)) << 3);
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t o4;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o4) : "v"(sa) : "memory");
            const int m = cm0 + wm * 64 + r16;
            const int col = ((n0 + wn * TN) >> 1) + 8 * c16;
            if (m < p.M && col < p.n_store) {
                if (!(ABL & 4)) *(uint4*)(outp + (long long)m * p.ldo + col) = make_uint4(o4.x, o4.y, o4.z, o4.w);
                else asm volatile("" ::"v"(o4));
            }
        }
    };
    // V^T slices of the fused q | k | v projection (IGemmParams::vt_out): the accumulators are staged exactly like epilogue_body's, read back
    // COLUMN-wise -- a lane takes 8 consecutive rows (tokens) of one channel -- and stored as 16 bytes along t of the transposed tensor.
    auto epilogue_vt = [&](auto slotc) __attribute__((always_inline)) {
        constexpr int S = decltype(slotc)::value;
        constexpr int RPP = 1024 / (TN * 4);
        constexpr int NI = TN * 2 / 64;       // (channel, row octet) items per lane and pass: TN channels x 2 octets of the 16 staged rows
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int q = lane_o >> 4, a = lane_o & 15;
        const unsigned win = smem_base + S * STAGE + wave * 1024;
        const int nv = p.N - p.vt_col0;       // channels of V
        float satm = 0.f;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
#pragma unroll
            for (int ip = 0; ip < FP; ++ip) {
                const unsigned d = win + (a / RPP) * 8192 + (a % RPP) * (TN * 4) + ((((4 * ip + q) ^ (a & (SLW - 1)))) << 5);
                *(lds_f4_ptr)d = acc[2 * ip][j];
                *(lds_f4_ptr)(d + 16) = acc[2 * ip + 1][j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int it = t * 64 + lane_o, c = it % TN, o = it / TN;   // channel c of the wave's TN, rows 8 o .. 8 o + 7 of this pass
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = 8 * o + k;
                    v[k] = *(lds_f_ptr)(win + (r / RPP) * 8192 + (r % RPP) * (TN * 4) + ((((c >> 3) ^ (r & (SLW - 1)))) << 5) + (c & 7) * 4);
                }
                const int m = cm0 + wm * 64 + 16 * j + 8 * o;               // first of the 8 rows: all of one image (vt_T % 16 == 0)
                const int ch = n0 - p.vt_col0 + wn * TN + c;
                if (m < p.M && ch < nv) {
                    const int b = m / p.vt_T, tt = m - b * p.vt_T;
                    uint4 pk;
                    pk.x = pack_h16x2_t(v[0], v[1], satm); pk.y = pack_h16x2_t(v[2], v[3], satm); pk.z = pack_h16x2_t(v[4], v[5], satm); pk.w = pack_h16x2_t(v[6], v[7], satm);
                    if (!(ABL & 4)) *(uint4*)(p.vt_out + ((long long)b * nv + ch) * p.vt_Tpad + tt) = pk;
                    else asm volatile("" ::"v"(pk.x), "v"(pk.y), "v"(pk.z), "v"(pk.w));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next pass overwrites the window)
        }
        sat_report(satm);
    };
    const bool vt_slice = p.vt_out != nullptr && n0 >= p.vt_col0;  // workgroup-uniform
    const int ep_variant = vt_slice ? 16 : p.act == GP_ACT_GEGLU ? 8 : (p.act != GP_ACT_NONE ? 4 : 0) | (p.res ? 2 : 0) | (want_stats ? 1 : 0);
    auto epilogue = [&](auto slotc) __attribute__((always_inline)) {
        switch (ep_variant) {
            case 0: epilogue_body(slotc, IC<0>{}, IC<0>{}, IC<0>{}); break;
            case 1: epilogue_body(slotc, IC<0>{}, IC<0>{}, IC<1>{}); break;
            case 2: epilogue_body(slotc, IC<0>{}, IC<1>{}, IC<0>{}); break;
            case 3: epilogue_body(slotc, IC<0>{}, IC<1>{}, IC<1>{}); break;
            case 8: epilogue_geglu(slotc); break;
            case 16: epilogue_vt(slotc); break;
            default: epilogue_body(slotc, IC<1>{}, IC<1>{}, IC<1>{}); break;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto flush_stats = [&](int tile_row) __attribute__((always_inline)) {  // after a workgroup barrier that follows epilogue()
        if (tid < PG_BN && n0 + tid < n_out) {
            const int cwn = tid / TN, ch = tid % TN;
            const unsigned a = st_base + (unsigned)((cwn * TN + ch) * 8);  // wave (wm, wn) = wm * WN + wn: stride WN * TN * 8 bytes over wm
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int w = 0; w < G::WM; ++w) {
                f32x2_t v;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a + (unsigned)(w * WN * TN * 8)) : "memory");
                ss += v.x;
                qq += v.y;
            }
            float* so = p.stats_out + ((long long)tile_row * p.N + n0 + tid) * 2;
            so[0] = ss;
            so[1] = qq;
        }
    };

    // ---- prologue: three stages in flight, fragments of step 0 in registers ----------------------------------------------------------
    if (total > 0) stage(0);
    if (total > 1) stage(1);
    if (total > 2) stage(2);
    if (NB > 3 && total > 3) stage(3);
    Half f0, f1a, f1b;
    // stages 0 and 1 have landed: whatever was issued behind them may stay in flight
    if (NB > 3 && total > 3) wait_vm<2 * LPS>(); else if (total > 2) wait_vm<LPS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    load_half(f0, IC<0>{}, IC<0>{});
    load_half(f1a, IC<0>{}, IC<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody holds its step-0 fragments: slot 0 may be refilled

    // ---- main loop -------------------------------------------------------------------------------------------------------------------
    // Invariant at the top of step gs (slot S = gs % NB): the barrier certifying stage gs+1 has been passed, f0 / cur1 hold both k-halves
    // of step gs, stages up to gs+NB-1 are issued.  The step issues stage gs+NB into slot S (read during step gs-1) and reads the fragments of
    // step gs+1 (unconditionally: past the end they are stale LDS bytes nobody uses).
    int gs = 0, kt = 0;
    auto kstep = [&](auto slotc, Half& cur1, Half& nxt1) __attribute__((always_inline)) {
        constexpr int S = decltype(slotc)::value, S1 = (S + 1) % NB;
        const bool tile_end = kt == nk - 1;
        const bool issue = gs + NB < total, more = gs + 1 < total;
        const bool dma_first = second_half && !tile_end;  // role split; at a tile end the slot is the epilogue's window first
        if (dma_first && issue) stage(S);
        __builtin_amdgcn_sched_barrier(0);
        load_half(nxt1, IC<S1>{}, IC<1>{});
        mfma_half(f0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        load_half(f0, IC<S1>{}, IC<0>{});
        mfma_half(cur1);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        if (tile_end) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my DMA has landed (stores below cannot delay a certification); next
                                                                         // step's fragments are in registers before the window is overwritten
            epilogue(slotc);
        }
        if (!dma_first && issue) stage(S);
        const int done_row = cm0 / BM;
        if (tile_end) { cm0 += tile_adv; kt = 0; } else ++kt;
        ++gs;
        if (!more) return;
        // stage gs+2 (ring position of the NEW gs: +1) must have landed; the stages issued behind it may stay in flight
        if (!tile_end) {
            if (NB == 3) { if (issue) wait_vm<LPS>(); else wait_vm<0>(); }
            else {
                const int behind = total - 2 - gs;  // stages issued behind the one needed (gs already advanced): min(NB - 2, total - 1 - (gs + 1))
                if (behind >= 2) wait_vm<2 * LPS>(); else if (behind == 1) wait_vm<LPS>(); else wait_vm<0>();
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile_end && want_stats) flush_stats(done_row);
    };
    while (gs < total) {
        if constexpr (NB == 3) {
            kstep(IC<0>{}, f1a, f1b);
            if (gs >= total) break;
            kstep(IC<1>{}, f1b, f1a);
            if (gs >= total) break;
            kstep(IC<2>{}, f1a, f1b);
            if (gs >= total) break;
            kstep(IC<0>{}, f1b, f1a);
            if (gs >= total) break;
            kstep(IC<1>{}, f1a, f1b);
            if (gs >= total) break;
            kstep(IC<2>{}, f1b, f1a);
        } else {
            kstep(IC<0>{}, f1a, f1b);
            if (gs >= total) break;
            kstep(IC<1>{}, f1b, f1a);
            if (gs >= total) break;
            kstep(IC<2>{}, f1a, f1b);
            if (gs >= total) break;
            kstep(IC<3>{}, f1b, f1a);
        }
    }
    if (want_stats && total > 0) {  // the last tile's partials
        __syncthreads();
        flush_stats((cm0 - tile_adv) / BM);
    }
}

bool pgemm_applicable(const IGemmParams& p) {
    if (p.vt_out && ((p.vt_col0 & 127) || (p.vt_T & 15) || (p.vt_Tpad & 7) || p.vt_col0 != p.n_store || p.res || p.stats_out || p.act != GP_ACT_NONE ||
                     p.bias_mode != GP_BIAS_NONE || p.M % p.vt_T)) return false;
    if (p.ks != 1 || p.batch > 1 || p.out_fp32 || p.bias_mode == GP_BIAS_ROW || p.in_scale) return false;
    // 32-bit byte offsets into A (incl. the rows of a ragged last tile) and into the packed weight: the buffer-resource DMA
    if (((long long)p.M + 256) * p.lda * 2 >= 0xfffffff0ll || (long long)p.n_rows * p.ldw * 2 >= 0xfffffff0ll) return false;
    if (p.act == GP_ACT_GEGLU) return !p.res && !p.stats_out && (p.N & 63) == 0 && (p.Cin & 63) == 0 && (p.lda & 7) == 0 && (p.ldw & 7) == 0 &&
                                      (p.ldo & 7) == 0 && (p.n_store & 7) == 0 && p.M >= 256;  // (16-byte output stores)
    if ((p.Cin & 63) || (p.lda & 7) || (p.ldw & 7) || (p.ldo & 7) || (p.n_store & 7)) return false;
    if (p.res && ((p.ldres & 7) || p.ldres < p.n_store)) return false;
    return p.M >= 256;
}

int pgemm_bm(const IGemmParams& p) {
    // fewer than ~1.5 tiles of 256 rows per workgroup: the makespan is set by tile quantisation, take 128-row tiles
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + PG_BN - 1) / PG_BN;
    int groups = 256 / tiles_n;
    if (groups < 1) groups = 1;
    const int t256 = (p.M + 255) / 256;
    return t256 >= 3 * groups ? 256 : 128;
}

template <int BM, int ABL, int NB = 3>
static void launch_pgemm_one(const IGemmParams& p, int ncu, hipStream_t s) {
    using G = PGemmGeom<BM, NB>;
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] { (void)hipFuncSetAttribute((const void*)pgemm_kernel<BM, ABL, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS); });
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles_n = (ncols + PG_BN - 1) / PG_BN, tiles_m = (p.M + BM - 1) / BM;
    int groups = ncu / tiles_n;
    if (groups < 1) groups = 1;
    if (groups > tiles_m) groups = tiles_m;
    hipLaunchKernelGGL((pgemm_kernel<BM, ABL, NB>), dim3(groups * tiles_n), dim3(512), G::LDS, s, p);
}

void launch_pgemm(const IGemmParams& p, hipStream_t s) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int abl = (p.dbg >> 9) & 15;  // profiling ablations (GENPERCEPT_IGEMM_DBG = 512 * ABL)
    if (pgemm_bm(p) == 256) {
        if (abl == 1) launch_pgemm_one<256, 1>(p, ncu, s);   // A/B: LDS-DMA as global_load_lds (r2 / r3)
        else if (abl == 2) launch_pgemm_one<256, 2>(p, ncu, s);
        else if (abl == 4) launch_pgemm_one<256, 4>(p, ncu, s);
        else if (abl == 8) launch_pgemm_one<256, 8>(p, ncu, s);
        else launch_pgemm_one<256, 0>(p, ncu, s);
    } else {
        if (abl == 1) launch_pgemm_one<128, 1, 4>(p, ncu, s);
        else if (abl == 2) launch_pgemm_one<128, 2>(p, ncu, s);
        else if (abl == 4) launch_pgemm_one<128, 4>(p, ncu, s);
        else if (abl == 8) launch_pgemm_one<128, 8>(p, ncu, s);
        else if (abl == 10) launch_pgemm_one<128, 10>(p, ncu, s);
        else if (abl == 14) launch_pgemm_one<128, 14>(p, ncu, s);
        else if (((p.dbg >> 23) & 1) || gp_sw().pgemm_ring3) launch_pgemm_one<128, 0>(p, ncu, s);   // A/B: the 3-deep ring of r2 / r3
        else launch_pgemm_one<128, 0, 4>(p, ncu, s);
    }
}

GP_SAT_TU(pgemm)  // fp16 build: address of this translation unit's saturation flag (common.h)
