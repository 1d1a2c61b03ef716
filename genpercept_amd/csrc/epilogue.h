// Shared epilogue of the MFMA conv/GEMM kernels.
//
// Accumulator layout on entry (see igemm.hip): for pixel fragment j and weight-fragment pair ip, lane (q = lane>>4,
// a = lane&15) holds acc[2ip + (e>>2)][j][e&3], e = 0..7 = output channels (tile-local) 32*ip + 8q + e of tile row
// 16*j + a of this wave's (wm, wn) sub-tile.
//
// Fast path (bf16 output, no GEGLU, 16-byte-aligned rows): the tile is staged through LDS as fp32 (the K-loop ring is free
// by then), then written out cooperatively with every lane storing 16 contiguous bytes and a wave covering whole pixel rows:
// full-line HBM writes instead of 64 scattered 16-byte pieces per instruction (measured: the scattered form cost more than
// the K loop on K = 1152 layers).  The residual is read the same coalesced way and added in fp32 before the single rounding.
// Everything else (fp32 output, GEGLU, odd strides) takes the direct per-lane path.
#pragma once
#include "common.h"
#include "kernels.h"

template <int BM, int BN, int WM, int WN, int NTHREADS, typename RowMap>
GP_DEV void conv_epilogue(const IGemmParams& p, f32x4_t (&acc)[(BN / WN) / 16][(BM / WM) / 16], float (&bcol)[(BN / WN) / 32][8], int n0, int z,
                          int wave, int lane, char* smem, RowMap row_to_m, int tile_row = 0) {
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FP = TN / 32;
    constexpr int SL = BN / 8;  // 8-channel slots per tile row
    const int wm = wave / WN, wn = wave % WN;
    const int a15 = lane & 15, q8 = 8 * (lane >> 4);
    const bool geglu = p.act == GP_ACT_GEGLU;
    const int n_out = geglu ? (p.N >> 1) : p.N;
    const float* bias = p.bias ? p.bias + (long long)z * p.bias_bs : nullptr;
    const h16_t* res = p.res ? p.res + (long long)z * p.res_bs : nullptr;
    const bool f32o = p.out_fp32 == 1;  // contract precision: fp32 rows out, fp32 residual in (IGemmParams::res_f32)
    const bool staged = (!p.out_fp32 || f32o) && !geglu && (p.ldo & 7) == 0 && !(p.dbg & 64);

    if (staged) {
        float* stg = (float*)smem;  // [BM][BN] fp32, 32-byte slots XOR-swizzled with the row index
        __builtin_amdgcn_s_barrier();  // every wave has consumed its last fragments: the ring may be overwritten
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int pr = wm * TM + j * 16 + a15;
            const int mrow = row_to_m(pr);
            const float rb = (p.bias_mode == GP_BIAS_ROW && mrow >= 0) ? bias[mrow] : 0.f;
#pragma unroll
            for (int ip = 0; ip < FP; ++ip) {
                const int s = (wn * TN + 32 * ip + q8) >> 3;
                float* d = stg + (long long)pr * BN + ((s ^ (pr & (SL - 1))) << 3);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[2 * ip + (e >> 2)][j][e & 3] + bcol[ip][e] + rb;
                *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        __syncthreads();
        h16_t* outp = (h16_t*)p.out + (long long)z * p.out_bs;
        const bool res_vec = res && (p.ldres & 7) == 0;
        // GroupNorm statistics of the tensor being written (NTHREADS % SL == 0: a thread always handles the same 8 channels)
        static_assert(NTHREADS % SL == 0, "stats: a thread must keep its channel slot");
        const bool want_stats = p.stats_out != nullptr;
        float st_s[8], st_q[8];
        float satm = 0.f;  // fp16 build: max |value| packed by this thread (sat_report below)
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
        for (int idx = threadIdx.x; idx < BM * SL; idx += NTHREADS) {
            const int pr = idx / SL, s = idx - pr * SL;
            const int col = n0 + s * 8;
            const int m = row_to_m(pr);
            if (m < 0 || col >= p.n_store) continue;
            const float* sp = stg + (long long)pr * BN + ((s ^ (pr & (SL - 1))) << 3);
            const float4 x0 = *(const float4*)sp, x1 = *(const float4*)(sp + 4);
            float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            if (res && p.res_f32) {
                const float* rp = (const float*)p.res + (long long)z * p.res_bs + (long long)m * p.ldres + col;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < n_out) v[e] += rp[e];
            } else if (res) {
                const h16_t* rp = res + (long long)m * p.ldres + col;
                if (res_vec && col + 7 < n_out) {
                    const uint4 rv = *(const uint4*)rp;
                    v[0] += h16_lo(rv.x); v[1] += h16_hi(rv.x); v[2] += h16_lo(rv.y); v[3] += h16_hi(rv.y);
                    v[4] += h16_lo(rv.z); v[5] += h16_hi(rv.z); v[6] += h16_lo(rv.w); v[7] += h16_hi(rv.w);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (col + e < n_out) v[e] += h16_to_f(rp[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (p.act == GP_ACT_SILU) v[e] = silu_f(v[e]);
                else if (p.act == GP_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                if (col + e >= n_out) v[e] = 0.f;
            }
            if (f32o) {  // the values themselves are what is stored: statistics of v
                float* o = (float*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 7 < p.n_store) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (col + e < p.n_store) o[e] = v[e];
                }
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { st_s[e] += v[e]; st_q[e] += v[e] * v[e]; }
                }
                continue;
            }
            h16_t* o = outp + (long long)m * p.ldo + col;
            uint4 pk;
            pk.x = pack_h16x2_t(v[0], v[1], satm); pk.y = pack_h16x2_t(v[2], v[3], satm); pk.z = pack_h16x2_t(v[4], v[5], satm); pk.w = pack_h16x2_t(v[6], v[7], satm);
            if (col + 7 < p.n_store) {
                *(uint4*)o = pk;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < p.n_store) o[e] = f_to_h16(v[e]);
            }
            if (want_stats) {  // of the values as stored (bf16-rounded), exactly what a read pass would see
                const float r[8] = {h16_lo(pk.x), h16_hi(pk.x), h16_lo(pk.y), h16_hi(pk.y), h16_lo(pk.z), h16_hi(pk.z), h16_lo(pk.w), h16_hi(pk.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) { st_s[e] += r[e]; st_q[e] += r[e] * r[e]; }
            }
        }
        sat_report(satm);
        if (want_stats) {
            // deterministic tree: lanes of a wave sharing a slot (xor-shuffles), then the waves through LDS (the staging area is
            // free once everybody has left the store loop), then one thread per channel writes this tile's partial
#pragma unroll
            for (int e = 0; e < 8; ++e) { st_s[e] = slot_sum<SL>(st_s[e]); st_q[e] = slot_sum<SL>(st_q[e]); }
            __syncthreads();
            constexpr int NWV = NTHREADS / 64;
            float* red = (float*)smem;  // [NWV][SL][16]
            if (lane < SL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { red[(wave * SL + lane) * 16 + e] = st_s[e]; red[(wave * SL + lane) * 16 + 8 + e] = st_q[e]; }
            }
            __syncthreads();
            for (int c = threadIdx.x; c < BN; c += NTHREADS) {
                if (n0 + c >= n_out) continue;
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) { ss += red[(w * SL + (c >> 3)) * 16 + (c & 7)]; qq += red[(w * SL + (c >> 3)) * 16 + 8 + (c & 7)]; }
                float* so = p.stats_out + ((long long)tile_row * p.N + n0 + c) * 2;
                so[0] = ss;
                so[1] = qq;
            }
        }
        return;
    }

    // ---- direct path --------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = row_to_m(wm * TM + j * 16 + a15);
        if (m < 0) continue;
        const float rb = p.bias_mode == GP_BIAS_ROW ? bias[m] : 0.f;
#pragma unroll
        for (int ip = 0; ip < FP; ++ip) {
            const int cb = n0 + wn * TN + 32 * ip;  // first packed column of this fragment pair
            if (geglu) {
                // packed rows of a 32-block: 8q + r = value, 8q + 4 + r = gate of output column cb/2 + 4q + r
                const int col = (cb >> 1) + (q8 >> 1);
                if (col >= p.n_store) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[2 * ip][j][r] + bcol[ip][r], g = acc[2 * ip + 1][j][r] + bcol[ip][4 + r];
                    v[r] = a * gelu_erf_f(g);
                    if (col + r >= n_out) v[r] = 0.f;
                }
                if (f32o) {
                    float* o = (float*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.n_store) o[r] = v[r];
                    continue;
                }
                h16_t* o = (h16_t*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 3 < p.n_store && (p.ldo & 3) == 0) {
                    *(uint2*)o = pack_h16x4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.n_store) o[r] = f_to_h16(v[r]);
                }
                continue;
            }
            const int col = cb + q8;
            if (col >= p.n_store) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * ip + (e >> 2)][j][e & 3] + bcol[ip][e] + rb;
            if (res && p.res_f32) {
                const float* rp = (const float*)p.res + (long long)z * p.res_bs + (long long)m * p.ldres + col;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < n_out) v[e] += rp[e];
            } else if (res) {
                const h16_t* rp = res + (long long)m * p.ldres + col;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < n_out) v[e] += h16_to_f(rp[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (p.act == GP_ACT_SILU) v[e] = silu_f(v[e]);
                else if (p.act == GP_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                if (col + e >= n_out) v[e] = 0.f;
            }
            if (p.out_fp32 == 2) {  // fp16 output (attention logits: 11 significant bits), saturating at the fp16 range
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e], -65504.f), 65504.f);
                _Float16* o = (_Float16*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 7 < p.n_store && (p.ldo & 7) == 0) {
                    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
                    h8_t hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[e];
                    *(h8_t*)o = hv;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (col + e < p.n_store) o[e] = (_Float16)v[e];
                }
            } else if (p.out_fp32) {
                float* o = (float*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
                if (col + 7 < p.n_store && (p.ldo & 3) == 0) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (col + e < p.n_store) o[e] = v[e];
                }
            } else {
                h16_t* o = (h16_t*)p.out + (long long)z * p.out_bs + (long long)m * p.ldo + col;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < p.n_store) o[e] = f_to_h16(v[e]);
            }
        }
    }
}

// preload this lane's 8 bias values per fragment pair (column bias); zeros otherwise
template <int FP>
GP_DEV void load_bias_cols(const IGemmParams& p, int z, int col0_of_pair0, int q8, float (&bcol)[FP][8]) {
    const float* bias = p.bias ? p.bias + (long long)z * p.bias_bs : nullptr;
#pragma unroll
    for (int ip = 0; ip < FP; ++ip) {
        const int c0 = col0_of_pair0 + 32 * ip + q8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bcol[ip][e] = 0.f;
        if (p.bias_mode == GP_BIAS_COL) {
            if (c0 + 7 < p.N) {
                const float4 b0 = *(const float4*)(bias + c0), b1 = *(const float4*)(bias + c0 + 4);
                bcol[ip][0] = b0.x; bcol[ip][1] = b0.y; bcol[ip][2] = b0.z; bcol[ip][3] = b0.w;
                bcol[ip][4] = b1.x; bcol[ip][5] = b1.y; bcol[ip][6] = b1.z; bcol[ip][7] = b1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (c0 + e < p.N) bcol[ip][e] = bias[c0 + e];
            }
        }
    }
}
