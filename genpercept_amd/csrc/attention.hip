// Attention kernels (K4/K5/K6 of SURVEY.md §2.3).
//
// flash_attn64: UNet self-attention, head_dim 64, flash-style (scores never leave the CU) on v_mfma_f32_32x32x16_bf16.
//   Per 256-thread workgroup: 128 query rows of one (image, head); each wave owns 32 query rows.  K tiles [64 keys][64 d]
//   and V^T tiles [64 d][64 keys] are brought in by LDS-DMA (double-buffered, swizzled like the GEMM tiles).
//   The wave computes S^T = K Q^T (keys x queries), so every lane holds 32 scores of ONE query (q = lane & 31):
//   the softmax max/sum/rescale are lane-local (one cross-half exchange with lane ^ 32), and the probabilities are
//   already in B-operand order for O^T += V^T P^T -- no LDS round trip, no permutes.  The K rows are staged with bits 2 and 3 of
//   their index swapped, which makes the 8 keys of a lane's k-step consecutive: the V^T A-operand is then one 16-byte read of the
//   tile the projection GEMM produced transposed ([C][Tpad], zero-padded beyond T).
// cross_attn_small: cross-attention against the constant, tiny text context (L = 2 for GenPercept's empty prompt).
// softmax_rows: row softmax for the GEMM-based single-head VAE attention (head_dim 512).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

// 32-row MFMA fragments (rows = lane & 31, slots s / s+1 for the two half-waves) are conflict-free with slot ^ ((row >> 1) & 7);
// the conv kernels' slot ^ (row & 7) is not for this pattern (checked exhaustively), so the attention tiles keep their own swizzle.
GP_DEV int attn_off128(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

#ifndef GP_FLASH_DEFER
#define GP_FLASH_DEFER 0.f  // rescale threshold in log2 units, see tile(): 0 = the reference is the exact running maximum
#endif
// NKB: 32-key blocks per KV tile (2: 64 keys, 4: 128 keys -- one online-softmax update, one barrier and one DMA wait per 128 keys,
// longer independent MFMA runs)
// NST: depth of the K / V ring in LDS (16 KiB per stage at NKB = 2).  2 = one tile ahead, a full vmcnt(0) + barrier per tile; 3 = two tiles
// ahead with COUNTED waits and one raw s_barrier per tile -- the DMA of tile t+2 is issued before tile t computes and is only waited for at
// the top of tile t+2, so an L2 / MALL round trip no longer has to fit inside one tile's compute.
template <int NKB, int NST>
__global__ __launch_bounds__(256, 2) void flash_attn64_kernel(const h16_t* __restrict__ Q, const h16_t* __restrict__ K,
                                                            const h16_t* __restrict__ Vt, h16_t* __restrict__ O, int T, int heads, int ldq,
                                                            int ldk, int Tpad, int ldo) {
    constexpr int KEYS = 32 * NKB, KBYTES = KEYS * 128, NH = NKB / 2;  // NH 64-key halves, each with its own [64 d][64 keys] V^T tile
    constexpr int STAGE = 2 * KBYTES;  // K tile + V^T tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: workgroup id b runs on XCD b % 8 (observed), so give each XCD a contiguous run of (image, head, query block):
    // the ~96 workgroups resident on an XCD then share the K / V of one or two (image, head) pairs (2.4 MB each at T = 9216) in their
    // 4 MiB L2 instead of all twenty pairs streaming through every L2 (speed only, never correctness)
    const int nqb = (T + 127) >> 7;
    const int sid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = sid % nqb, bh = sid / nqb;
    const int h = bh % heads, b = bh / heads;
    const int q0 = qb * 128 + wave * 32;
    const int l31 = lane & 31, hh = lane >> 5;

    const h16_t* Qb = Q + (long long)b * T * ldq + h * 64;
    const h16_t* Kb = K + (long long)b * T * ldk + h * 64;
    const h16_t* Vb = Vt + ((long long)b * heads + h) * 64 * Tpad;

    // Q fragments (B operand of S^T = K Q^T): lane (q = l31, half hh) holds Q[q][16*ks + 8*hh .. +7]
    h16x8_t qf[4];
    {
        const int q = q0 + l31;
        const bool ok = q < T;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ok) qf[ks] = *(const h16x8_t*)(Qb + (long long)q * ldq + ks * 16 + hh * 8);
            else qf[ks] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const int nt = (T + KEYS - 1) / KEYS;
    // LDS-DMA through buffer resources (common.h: blds16): K rows of this (image, head) -- reads past row T - 1 return zeros -- and the 64
    // V^T rows of this head; per lane a byte offset that is the same for every tile, per instruction a uniform one
    const buf_rsrc_t k_rs = make_rsrc(Kb, (unsigned)(((long long)T - 1) * ldk * 2 + 128));
    const buf_rsrc_t v_rs = make_rsrc(Vb, (unsigned)(64ll * Tpad * 2));
    // (the K rows of a 16-key group are staged in the order pi(i) = i with bits 2 and 3 swapped: LDS rows 8 g + r hold keys
    // 16 (g >> 1) + 4 (g & 1) + (r & 3) + 8 (r >> 2), so that accumulator registers 8 j .. 8 j + 7 of a lane are the CONSECUTIVE keys
    // 16 j + 8 hh + 0..7 and the matching V^T operand is one 16-byte LDS read instead of two 8-byte ones)
    const unsigned k_lane = (unsigned)(((lane >> 3) & 3) + 8 * (lane >> 5)) * (unsigned)(ldk * 2) + chunk * 16;
    const unsigned v_lane = ((unsigned)(lane >> 3) * Tpad + chunk * 8) * 2;
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NKB; ++i) {  // K rows: 8 * (wave + 4 i) + lane / 8
            const int g = wave + 4 * i;
            blds16(k_rs, k_lane, (unsigned)(kt * KEYS + (g >> 1) * 16 + 4 * (g & 1)) * (unsigned)(ldk * 2), sb + g * 1024);
        }
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // V^T rows = head channels d; key blocks past Tpad: an offset outside the resource (zeros)
                const int k0 = kt * KEYS + hf * 64;
                const unsigned so = k0 < Tpad ? (unsigned)((wave + 4 * i) * 8 * Tpad + k0) * 2u : 0xfffff000u;  // (Tpad % 64 == 0)
                blds16(v_rs, v_lane, so, sb + KBYTES + hf * 8192 + (wave + 4 * i) * 1024);
            }
    };

    f32x16_t o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1e30f;   // running maximum in raw-score units
    float l_run = 0.f;
    const float sc = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    int cur = 0;
    const unsigned smem_base = (unsigned)(unsigned long long)smem;  // integer-addressed LDS reads: see common.h (no compiler vmcnt(0) before them)

    // One KV tile.  The softmax is the bound of this kernel (head_dim 64: 16 MFMAs of 32 cycles against ~32 scores per lane), so it is
    // kept to the minimum: raw v_exp_f32 (the libm exp2f wrapper added a compare, two selects and a v_ldexp per score), the 1/sqrt(d)
    // * log2(e) scale folded into one FMA per score (the running maximum is tracked on the RAW scores; sc > 0 keeps the order), key
    // masking only in the last tile (MASK), and the accumulator rescale skipped while no lane's maximum moves.
    // (Row sums through an extra all-ones MFMA block were tried: no gain, 16 more VGPRs.)
    auto tile = [&](int kt, auto maskc) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value != 0;
        const unsigned sb = smem_base + cur * STAGE;
        // ---- S^T = K Q^T: two 32-key blocks, 4 k-steps of 16 over d
        f32x16_t s_acc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int row = kb * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8_t kf = lds_frag(sb + attn_off128(row, ks * 2 + hh), 0);
                s_acc[kb] = mfma_32x32x16(kf, qf[ks], ks == 0 ? zero16 : s_acc[kb]);  // C = 0: inline constant
            }
        }
        // ---- online softmax over this lane's 32 keys (+ the other half's 32 via lane ^ 32)
        if (MASK) {
            const int kbase = kt * KEYS + 8 * hh;  // register r <-> LDS row (r & 3) + 4 hh + 8 (r >> 2) <-> key 8 hh + (r & 7) + 16 (r >> 3)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kb * 32 + (r & 7) + 16 * (r >> 3) >= T) s_acc[kb][r] = -1e30f;
        }
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
        mx = xor32_max(mx);                                // the other half-wave holds this query's other 32 keys (v_permlane32_swap, no LDS trip)
        // The reference m_run of the exponentials moves (and accumulators and row sum are rescaled) only in tiles where some query of the wave
        // sets a new maximum.  A threshold GP_FLASH_DEFER > 0 would defer that until a maximum outgrows its reference by 2^threshold (exact
        // in fp32, -2.4 % kernel time at 8) -- measured and NOT used: with the exact maximum as reference the dominant probabilities lie just
        // below 1.0, where the 16-bit rounding of the P operand is finest; against a stale reference their mantissas are arbitrary and the
        // error of the output grows by 20 % (bf16: 1.7e-3 -> 2.1e-3 of the mean magnitude).
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * sc > GP_FLASH_DEFER) != 0ull) {  // (uniform)
            const float m_new = fmaxf(m_run, mx);          // raw-score units
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        // (two scores per instruction where the ISA has a packed form: v_pk_fma_f32 for the scale / reference, v_pk_add_f32 for the row sum;
        // this loop is the kernel's bound -- 33 quarter-rate v_exp_f32 and ~100 other VALU per 16 MFMAs)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t sc2 = {sc, sc}, nm2 = {-m_run * sc, -m_run * sc};
        f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t t = {s_acc[kb][r], s_acc[kb][r + 1]};
                t = __builtin_elementwise_fma(t, sc2, nm2);
                t.x = __builtin_amdgcn_exp2f(t.x);
                t.y = __builtin_amdgcn_exp2f(t.y);
                s_acc[kb][r] = t.x;
                s_acc[kb][r + 1] = t.y;
                rs2 += t;
            }
        l_run += rs2.x + rs2.y;
        // ---- O^T += V^T P^T: k-steps (kb, j) of 16 keys; this lane's P for its own query is the B operand
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                union { h16x8_t v; unsigned u[4]; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_h16x2_ns(s_acc[kb][8 * j + 2 * e], s_acc[kb][8 * j + 2 * e + 1]);  // probabilities: in [0, 1]
                const int slot = (kb & 1) * 4 + 2 * j + hh;  // keys 8 slot .. 8 slot + 7 of the 64-key half
                // (requesting all eight V^T fragments ahead of the softmax -- 136 registers, three waves per SIMD instead of four -- measured equal)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const h16x8_t vf = lds_frag(sb + KBYTES + (kb >> 1) * 8192 + attn_off128(d * 32 + l31, slot), 0);
                    o_acc[d] = mfma_32x32x16(vf, pf.v, o_acc[d]);
                }
            }
    };

    constexpr int LPS = NKB + 2 * NH;  // LDS-DMA instructions per wave and stage
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nt) stage(s0, s0);
    int nxt = NST - 1;
    auto step = [&](int kt, auto maskc) __attribute__((always_inline)) {
        // my DMAs of tile kt have landed (NST - 2 younger stages may stay in flight) ...
        const int ahead = min(NST - 2, nt - 1 - kt);
        if (ahead >= 1) wait_vm<LPS>(); else wait_vm<0>();
        static_assert(NST == 2 || NST == 3, "ring depth");
        // ... and after the barrier everybody's have, and everybody is done reading the slot of tile kt - 1, which tile kt + NST - 1 refills
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nt) stage(nxt, kt + NST - 1);
        tile(kt, maskc);
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    };
    // Only the last tile can reach past key T - 1.  It runs after the loop: with both tile variants inside the loop body the accumulators
    // lived in different registers on the two paths and every iteration paid 16 v_mov_b64 of copies at the merge.
    const bool ragged = nt * KEYS > T;
    const int nfull = ragged ? nt - 1 : nt;
    for (int kt = 0; kt < nfull; ++kt) step(kt, IC<0>{});
    if (ragged) step(nt - 1, IC<1>{});
    // ---- normalise and store O[q][d] (this lane: q = l31, d = 32*blk + 8*(r>>2) + 4*hh + (r&3))
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.f / l_run;
    const int q = q0 + l31;
    if (q < T) {
        h16_t* ob = O + ((long long)b * T + q) * ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 pk = pack_h16x4(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv, o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv);
                *(uint2*)(ob + d * 32 + 8 * g + 4 * hh) = pk;
            }
    }
}

void launch_flash_attn64(const h16_t* q, const h16_t* k, const h16_t* vt, h16_t* out, int B, int T, int heads,
                         int ldq, int ldk, int Tpad, int ldo, hipStream_t s) {
    dim3 grid(((T + 127) / 128) * heads * B);
    // r3: 113 VGPRs, four waves per SIMD (LDS: 32 KiB per workgroup), 506 us at T = 9216 / batch 4 / 5 heads (860 TFLOP/s; r2: 183 VGPRs, two waves,
    // 580 us) -- the whole difference is the masked tile variant moved out of the loop (see the loop).  PMC of the r3 kernel: matrix pipe busy 40 % of
    // the cycles, VALU-active 61 %: the two do not overlap to speak of, the SIMD is busy issuing ~147 VALU (33 of them quarter-rate v_exp_f32) per 16 MFMAs;
    // packed fma / add forms (kept) and a deferred rescale (not kept, see tile()) did not change the time.
    // r2 notes (183-register kernel): forcing three waves, __launch_bounds__(256, 3), spilled 38 registers.
    // Measured alternatives, all slower at T = 9216 (610 us, kernel only): 128-key tiles -8 % (r1); a three-stage K / V ring with counted waits
    // -2 % (stays as a switch, GENPERCEPT_FLASH_RING3: K / V latency is not what the kernel waits for); two 32-query blocks per wave so that each
    // K / V^T fragment read feeds two MFMAs: 1100 us with 256 VGPRs + 42 spilled, 1050 us at one wave per SIMD; one online-softmax update per
    // 32 keys instead of 64 (165 VGPRs): 644 us.  r2 ablations of the shipped shape: no softmax arithmetic 494 us, no P.V MFMAs 587 us, neither
    // 442 us, additionally no K.Q^T MFMAs 249 us (the LDS fragment reads, DMA and loop alone), no waits / barriers 601 us.
    const bool ring2 = !gp_sw().flash_ring3;
    if (ring2) hipLaunchKernelGGL((flash_attn64_kernel<2, 2>), grid, dim3(256), 2 * 16384, s, q, k, vt, out, T, heads, ldq, ldk, Tpad, ldo);
    else hipLaunchKernelGGL((flash_attn64_kernel<2, 3>), grid, dim3(256), 3 * 16384, s, q, k, vt, out, T, heads, ldq, ldk, Tpad, ldo);
}

// ---- flash_attn64 with SPLIT operands (contract precision, contract.hip): q = q_hi + q_lo, k, v, p likewise (bf16 pieces of fp32 values);
//   S^T = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T,   O^T += V_hi^T P_hi^T + V_hi^T P_lo^T + V_lo^T P_hi^T   (fp32 accumulators, fp32 softmax).
// Same dataflow as flash_attn64_kernel (lane-local online softmax on S^T, probabilities already in B-operand order); every K / V^T fragment
// read from LDS feeds one (lo) or two (hi) MFMAs, so the three-fold matrix work costs 1.5x the LDS traffic.  Per stage: K_hi | K_lo | V_hi^T |
// V_lo^T tiles of 8 KiB each; two stages = 64 KiB per workgroup.  Inputs: planes QKhi / QKlo [B*T][ld] (q | k row-major, head h at columns
// h * 64 and C + h * 64), Vthi / Vtlo [B * heads][64][Tpad] (zero beyond T).  Output: the attention result as the A-order split operand
// [B*T][3 C] ([hi | lo | hi]) the output projection reads.
__global__ __launch_bounds__(256, 2) void flash_attn64_split_kernel(const h16_t* __restrict__ QKhi, const h16_t* __restrict__ QKlo,
                                                                  const h16_t* __restrict__ Vthi, const h16_t* __restrict__ Vtlo,
                                                                  h16_t* __restrict__ O, int T, int heads, int ld, int Tpad) {
    constexpr int KEYS = 64, KBYTES = 8192, STAGE = 4 * KBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (T + 127) >> 7;
    const int sid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = sid % nqb, bh = sid / nqb;
    const int h = bh % heads, b = bh / heads;
    const int C = heads * 64;
    const int q0 = qb * 128 + wave * 32;
    const int l31 = lane & 31, hh = lane >> 5;

    const long long qoff = (long long)b * T * ld + h * 64;
    const long long voff = ((long long)b * heads + h) * 64 * Tpad;
    h16x8_t qh[4], ql[4];
    {
        const int q = q0 + l31;
        const bool ok = q < T;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ok) {
                qh[ks] = *(const h16x8_t*)(QKhi + qoff + (long long)q * ld + ks * 16 + hh * 8);
                ql[ks] = *(const h16x8_t*)(QKlo + qoff + (long long)q * ld + ks * 16 + hh * 8);
            } else {
                qh[ks] = ql[ks] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const int nt = (T + KEYS - 1) / KEYS;
    const unsigned kbytes = (unsigned)(((long long)T - 1) * ld * 2 + 128);
    const buf_rsrc_t kh_rs = make_rsrc(QKhi + qoff + C, kbytes), kl_rs = make_rsrc(QKlo + qoff + C, kbytes);
    const buf_rsrc_t vh_rs = make_rsrc(Vthi + voff, (unsigned)(64ll * Tpad * 2)), vl_rs = make_rsrc(Vtlo + voff, (unsigned)(64ll * Tpad * 2));
    const unsigned k_lane = (unsigned)(((lane >> 3) & 3) + 8 * (lane >> 5)) * (unsigned)(ld * 2) + chunk * 16;  // (K rows staged with bits 2 and 3 swapped)
    const unsigned v_lane = ((unsigned)(lane >> 3) * Tpad + chunk * 8) * 2;
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = wave + 4 * i;
            const unsigned ko = (unsigned)(kt * KEYS + (g >> 1) * 16 + 4 * (g & 1)) * (unsigned)(ld * 2);
            blds16(kh_rs, k_lane, ko, sb + g * 1024);
            blds16(kl_rs, k_lane, ko, sb + KBYTES + g * 1024);
            const int k0 = kt * KEYS;
            const unsigned so = k0 < Tpad ? (unsigned)((wave + 4 * i) * 8 * Tpad + k0) * 2u : 0xfffff000u;
            blds16(vh_rs, v_lane, so, sb + 2 * KBYTES + g * 1024);
            blds16(vl_rs, v_lane, so, sb + 3 * KBYTES + g * 1024);
        }
    };
    f32x16_t o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sc = 0.125f * 1.44269504088896340736f;
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    int cur = 0;
    const unsigned smem_base = (unsigned)(unsigned long long)smem;

    auto tile = [&](int kt, auto maskc) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value != 0;
        const unsigned sb = smem_base + cur * STAGE;
        f32x16_t s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8_t kfh = lds_frag(sb + attn_off128(row, ks * 2 + hh), 0);
                const h16x8_t kfl = lds_frag(sb + KBYTES + attn_off128(row, ks * 2 + hh), 0);
                s_acc[kb] = mfma_32x32x16(kfl, qh[ks], ks == 0 ? zero16 : s_acc[kb]);  // small terms first
                s_acc[kb] = mfma_32x32x16(kfh, ql[ks], s_acc[kb]);
                s_acc[kb] = mfma_32x32x16(kfh, qh[ks], s_acc[kb]);
            }
        }
        if (MASK) {
            const int kbase = kt * KEYS + 8 * hh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kb * 32 + (r & 7) + 16 * (r >> 3) >= T) s_acc[kb][r] = -1e30f;
        }
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
        mx = xor32_max(mx);
        if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0ull) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        const float nm = -m_run * sc;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, nm));
                s_acc[kb][r] = t;
                rs += t;
            }
        l_run += rs;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                union { h16x8_t v; unsigned u[4]; } ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = s_acc[kb][8 * j + 2 * e], c = s_acc[kb][8 * j + 2 * e + 1];
                    ph.u[e] = pack_h16x2_ns(a, c);
                    pl.u[e] = pack_h16x2_ns(a - h16_lo(ph.u[e]), c - h16_hi(ph.u[e]));
                }
                const int slot = kb * 4 + 2 * j + hh;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const h16x8_t vfh = lds_frag(sb + 2 * KBYTES + attn_off128(d * 32 + l31, slot), 0);
                    const h16x8_t vfl = lds_frag(sb + 3 * KBYTES + attn_off128(d * 32 + l31, slot), 0);
                    o_acc[d] = mfma_32x32x16(vfl, ph.v, o_acc[d]);
                    o_acc[d] = mfma_32x32x16(vfh, pl.v, o_acc[d]);
                    o_acc[d] = mfma_32x32x16(vfh, ph.v, o_acc[d]);
                }
            }
    };
    stage(0, 0);
    auto step = [&](int kt, auto maskc) __attribute__((always_inline)) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
        tile(kt, maskc);
        cur ^= 1;
    };
    const bool ragged = nt * KEYS > T;
    const int nfull = ragged ? nt - 1 : nt;
    for (int kt = 0; kt < nfull; ++kt) step(kt, IC<0>{});
    if (ragged) step(nt - 1, IC<1>{});
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.f / l_run;
    const int q = q0 + l31;
    if (q < T) {  // this lane: query q, channels d = 32 blk + 8 g + 4 hh + 0..3; written as [hi | lo | hi] blocks of C
        h16_t* ob = O + ((long long)b * T + q) * 3 * C + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v0 = o_acc[d][4 * g] * inv, v1 = o_acc[d][4 * g + 1] * inv, v2 = o_acc[d][4 * g + 2] * inv, v3 = o_acc[d][4 * g + 3] * inv;
                const unsigned h0 = pack_h16x2(v0, v1), h1 = pack_h16x2(v2, v3);
                const unsigned l0 = pack_h16x2(v0 - h16_lo(h0), v1 - h16_hi(h0)), l1 = pack_h16x2(v2 - h16_lo(h1), v3 - h16_hi(h1));
                const int c = d * 32 + 8 * g + 4 * hh;
                *(uint2*)(ob + c) = make_uint2(h0, h1);
                *(uint2*)(ob + C + c) = make_uint2(l0, l1);
                *(uint2*)(ob + 2 * C + c) = make_uint2(h0, h1);
            }
    }
}
void launch_flash_attn64_split(const h16_t* qk_hi, const h16_t* qk_lo, const h16_t* vt_hi, const h16_t* vt_lo, h16_t* out, int B, int T, int heads, int ld,
                               int Tpad, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)flash_attn64_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
    });
    dim3 grid(((T + 127) / 128) * heads * B);
    hipLaunchKernelGGL(flash_attn64_split_kernel, grid, dim3(256), 2 * 32768, s, qk_hi, qk_lo, vt_hi, vt_lo, out, T, heads, ld, Tpad);
}

// ---- flash_attn512: the VAE mid-block attention (one head, head_dim 512; genpercept_pipeline.py:500-501,521-522) ----------------------
// Same dataflow as flash_attn64 (S^T = K Q^T on v_mfma_f32_32x32x16, lane-local online softmax, the probabilities already in
// B-operand order for O^T += V^T P^T), sized for d = 512:
//   * one wave = 32 queries: Q^T stays in 128 VGPRs, O^T (512 x 32 fp32) in 256 accumulator registers -> one wave per SIMD, one
//     256-thread workgroup (128 queries) per CU;
//   * 32-key tiles: K tile [32 keys][512 d] (1 KiB rows) + V^T tile [512 d][32 keys] (64-byte rows) = 64 KiB per stage, two stages;
//     64 MFMAs (2048 matrix-pipe cycles) per wave and tile against 64 KiB of ds_read_b128 per wave (1024 LDS cycles per CU);
//   * the K rows are staged in the order pi(i) = i with bits 2 and 3 swapped, so that the 8 keys a lane's accumulator registers
//     8j .. 8j+7 belong to are CONSECUTIVE (16 j + 8 hh + 0..7) and its V^T operand is one 16-byte read (flash_attn64 needs two 8-byte
//     reads; those only reach their rate with >= 4 waves per SIMD);
//   * LDS slots: K physical slot = slot ^ (row & 15) (64 slots per row, XOR on the low four bits), V^T physical slot =
//     slot ^ ((row >> 2) & 3) (4 slots per row): both conflict-free for the 32-row fragments (tests/test_lds_layout.py);
//   * work split: B * ceil(T / 128) query blocks over G = min(blocks, CUs) persistent workgroups.  Whole rounds of G blocks are
//     written directly; the L = blocks mod G left-over blocks are cut along the KEYS into S = G / L parts, one per workgroup
//     (unnormalised O^T, running maximum and sum to a workspace), and flash512_combine_kernel merges the parts.  At 768^2, batch 4:
//     288 blocks on 256 CUs = 1.125 rounds instead of 2.
constexpr int F5_KBYTES = 32 * 1024;                    // one K tile [32 keys][512 d] or one V^T tile [512 d][32 keys]
constexpr int F5_QV = 24;                               // Q k-steps (of 32) kept in VGPRs; the rest is parked in LDS
constexpr int F5_RD = 6;                                // fragment reads in flight ahead of their MFMA
constexpr int F5_LDS = 4 * F5_KBYTES + 4 * (32 - F5_QV) * 1024;
// max / sum over the two half-waves without going through the LDS crossbar (ds_bpermute): v_permlane32_swap exchanges the upper half of
// one register with the lower half of another
typedef __attribute__((address_space(3))) h16x8_t* lds_frag_wptr;
GP_DEV float other_half(float v) {
    typedef unsigned u32x2p_t __attribute__((ext_vector_type(2)));
    const u32x2p_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);   // r[0] = [lo, lo], r[1] = [hi, hi]
}
GP_DEV int k512_off(int row, int slot) { return row * 1024 + ((slot ^ (row & 15)) << 4); }
GP_DEV int v512_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }
GP_DEV int pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }  // swap bits 2 and 3

__global__ __launch_bounds__(256) void flash_attn512_kernel(const h16_t* __restrict__ Q, const h16_t* __restrict__ K,
                                                             const h16_t* __restrict__ Vt, h16_t* __restrict__ O,
                                                             float* __restrict__ part_o,
                                                             float* __restrict__ part_ml, int B, int T, int ldq, int ldk, int Tpad, int ldo,
                                                             float scale, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int nqb = (T + 127) >> 7, nblocks = B * nqb, G = gridDim.x;
    const int nt = (T + 31) >> 5;
    const int rounds = nblocks / G, L = nblocks - rounds * G, S = L ? G / L : 0;
    const int sid = xcd_remap(blockIdx.x, G);           // consecutive sid = consecutive query blocks of one image on one XCD
    const unsigned smem_base = (unsigned)(unsigned long long)smem;
    const float sc = scale * 1.44269504088896340736f;
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    const int items = rounds + ((L && sid / S < L) ? 1 : 0);
    for (int it = 0; it < items; ++it) {
        const bool whole = it < rounds;
        const int part = whole ? 0 : sid % S;
        const int t0 = whole ? 0 : (int)((long long)nt * part / S), t1 = whole ? nt : (int)((long long)nt * (part + 1) / S);
        // which blocks are cut along the keys must not depend on the image's position in the batch (results are compared bit for bit
        // across batch permutations): when L divides by B every image contributes its LAST L / B query blocks, otherwise the last L
        // blocks of the batch are taken
        const int Li = (L % B == 0) ? L / B : 0;
        int b, qb;
        if (whole) {
            const int w = it * G + sid, per = nqb - Li;
            b = Li ? w / per : w / nqb;
            qb = Li ? w - b * per : w - b * nqb;
        } else {
            const int lb = sid / S;
            b = Li ? lb / Li : (rounds * G + lb) / nqb;
            qb = Li ? nqb - Li + (lb - b * Li) : rounds * G + lb - b * nqb;
        }
        const int q0 = qb * 128 + wave * 32;
        const h16_t* Qb = Q + (long long)b * T * ldq;
        const h16_t* Kb = K + (long long)b * T * ldk;
        const h16_t* Vb = Vt + (long long)b * 512 * Tpad;

        // Q fragments (B operand of S^T = K Q^T): lane (q = l31, half hh) holds Q[q][16 ks + 8 hh .. +7].  k-steps 0 .. F5_QV-1 stay in
        // VGPRs, the rest (32 KiB for the workgroup: what is left of the LDS) is parked in LDS, lane-major, and read back per tile
        __syncthreads();                                  // the previous item's last tile and Q rows have been read by every wave
        h16x8_t qf[F5_QV];
        const unsigned q_lds = smem_base + 4 * F5_KBYTES + wave * ((32 - F5_QV) * 1024) + lane * 16;
        {
            const int q = q0 + l31;
            const bool ok = q < T;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                h16x8_t v = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *(const h16x8_t*)(Qb + (long long)q * ldq + ks * 16 + hh * 8);
                if ((ks & 3) != 3) qf[3 * (ks >> 2) + (ks & 3)] = v;   // every fourth k-step is parked in LDS
                else *(lds_frag_wptr)(q_lds + (ks >> 2) * 1024) = v;
            }
            // the loads have landed HERE (they return in order): otherwise hipcc waits for them, with vmcnt(0), at their first use inside
            // the tile loop, i.e. behind the LDS-DMA of the next tile that was just issued
            asm volatile("" ::"v"(qf[F5_QV - 1]));
        }
        // DMA sources: one buffer resource per operand (base of this image), a per-lane byte offset that is the same for every tile and a
        // uniform byte offset per instruction
        const buf_rsrc_t k_rs = make_rsrc(Kb, (unsigned)((long long)T * ldk * 2));
        const buf_rsrc_t v_rs = make_rsrc(Vb, (unsigned)(512ll * Tpad * 2));
        unsigned koff[8];  // K: my LDS slot `lane` of row i = 8 wave + n holds logical slot lane ^ (i & 15)
#pragma unroll
        for (int n = 0; n < 8; ++n) koff[n] = (unsigned)(lane ^ ((wave & 1) * 8 + n)) << 4;
        const unsigned v_lane_off = ((unsigned)(lane >> 2) * Tpad + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2;  // V^T: row 16 n + lane / 4, logical slot of my LDS slot
        // LDS: K slots at 0 / 32 KiB, V^T slots at 64 / 96 KiB (tile t in slot (t - t0) & 1 of each), parked Q fragments at 128 KiB
        auto stage_piece = [&](int slot, int kt, int pc) __attribute__((always_inline)) {   // piece 0..7: a K row, 8..15: 16 V^T rows
            char* sb = smem + slot * F5_KBYTES;
            if (pc < 8) {
                const int i = wave * 8 + pc;                    // LDS row i <- key kt * 32 + pi(i); one 1 KiB row per instruction
                const int key = min(kt * 32 + pi23(i), T - 1);  // rows past T repeat the last one: their scores are masked (MASK)
                if (!(dbg & 1)) blds16(k_rs, koff[pc & 7], (unsigned)key * (unsigned)(ldk * 2), sb + i * 1024);
            } else {
                const int m = pc - 8;                           // 16 rows (channels) of 64 bytes per instruction
                if (!(dbg & 2)) blds16(v_rs, v_lane_off, (unsigned)((128 * wave + 16 * m) * Tpad + kt * 32) * 2u, sb + 2 * F5_KBYTES + (wave * 8 + m) * 1024);
            }
        };
        auto stage = [&](int slot, int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) stage_piece(slot, kt, pc);
        };

        f32x16_t o_acc[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) o_acc[d] = zero16;
        float m_run = -1e30f, l_run = 0.f, alpha = 1.f;
        // fragment addresses inside a slot: K slot (2 ks + hh) ^ (row & 15) = (hh ^ (row & 15)) ^ 2 (ks & 7), plus 256 bytes per 8 k-steps
        const unsigned ka0 = l31 * 1024 + ((hh ^ (l31 & 15)) << 4);
        unsigned va[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) va[j] = 2 * F5_KBYTES + l31 * 64 + (((2 * j + hh) ^ ((l31 >> 2) & 3)) << 4);

        // Fragments are read F5_RD MFMAs ahead of their use (counted lgkmcnt waits: the DMA is the MUBUF form, common.h); the next tile's
        // sixteen DMA pieces are issued between the MFMAs of the S^T phase, one per two MFMAs.
        // The accumulators (256 registers: the whole AGPR half of the file) must only ever be MFMA operands in the hot loop, so the
        // online-softmax rescale is (a) lazy -- the reference maximum m_run only moves when some query's tile maximum exceeds it by more
        // than 8 in log2 units; until then probabilities may reach 2^8, exact in the quotient sum(p v) / sum(p) and harmless in fp32 /
        // 16-bit P -- and (b) done outside the hot loop: the loop breaks before the tile's P.V, the cold path rescales and finishes the tile.
        h16x8_t pf[2];
        // ---- S^T = K Q^T over d = 512: 32 k-steps alternating between two accumulators (a single chain would wait for its own result)
        auto s_phase = [&](unsigned sb, int kt, bool do_stage, int slot_next, auto maskc) __attribute__((always_inline)) -> bool {
            constexpr bool MASK = decltype(maskc)::value != 0;
            f32x16_t s0, s1;  // (first MFMA of each chain takes the inline constant 0 as C: no 32 register writes per tile)
            h16x8_t fr[F5_RD], ql[2];
            auto rd = [&](int ks) __attribute__((always_inline)) {
                fr[ks % F5_RD] = lds_frag(sb + (ka0 ^ ((ks & 7) << 5)), (ks >> 3) * 256);
                if ((ks & 3) == 3) ql[(ks >> 2) & 1] = lds_frag(q_lds, (ks >> 2) * 1024);
            };
#pragma unroll
            for (int ks = 0; ks < F5_RD; ++ks) rd(ks);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const h16x8_t qv = (ks & 3) != 3 ? qf[3 * (ks >> 2) + ((ks & 3) != 3 ? (ks & 3) : 0)] : ql[(ks >> 2) & 1];
                if (ks & 1) s1 = mfma_32x32x16(fr[ks % F5_RD], qv, ks == 1 ? zero16 : s1);
                else s0 = mfma_32x32x16(fr[ks % F5_RD], qv, ks == 0 ? zero16 : s0);
                if (ks + F5_RD < 32) rd(ks + F5_RD);
                if (do_stage && (ks & 1)) stage_piece(slot_next, kt + 1, ks >> 1);   // the next tile's 16 DMA pieces, one per two MFMAs
            }
            f32x16_t s_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[r] = s0[r] + s1[r];
            // register r <-> LDS row (r & 3) + 4 hh + 8 (r >> 2) <-> key kt * 32 + 8 hh + (r & 7) + 16 (r >> 3)
            if (MASK) {
                const int kbase = kt * 32 + 8 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 7) + 16 * (r >> 3) >= T) s_acc[r] = -1e30f;
            }
            float mx = s_acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_acc[r]);
            mx = fmaxf(mx, other_half(mx));
            const bool moved = __builtin_amdgcn_ballot_w64((mx - m_run) * sc > 8.f) != 0ull;  // uniform
            if (moved) {
                const float m_new = fmaxf(m_run, mx);
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
                m_run = m_new;
                l_run *= alpha;
            }
            const float nm = -m_run * sc;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s_acc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, nm));
                rs += s_acc[r];
            }
            l_run += rs;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                union { h16x8_t v; unsigned u[4]; } t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t.u[e] = pack_h16x2_ns(s_acc[8 * j + 2 * e], s_acc[8 * j + 2 * e + 1]);
                pf[j] = t.v;
            }
            return moved;
        };
        // ---- O^T += V^T P^T: two k-steps of 16 keys x 16 channel blocks of 32 (step = 16 j + block)
        auto pv_phase = [&](unsigned sb) __attribute__((always_inline)) {
            h16x8_t fr[F5_RD];
            auto rd = [&](int st) __attribute__((always_inline)) { fr[st % F5_RD] = lds_frag(sb + va[st >> 4], (st & 15) * 2048); };
#pragma unroll
            for (int st = 0; st < F5_RD; ++st) rd(st);
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                o_acc[st & 15] = mfma_32x32x16(fr[st % F5_RD], pf[st >> 4], o_acc[st & 15]);
                if (st + F5_RD < 32) rd(st + F5_RD);
            }
        };

        if (t0 < t1) stage(0, t0);
        __builtin_amdgcn_s_waitcnt(0x0070);              // (compiler-visible: see the end of the cold path)
        int kt = t0;
        for (;;) {
            bool pending = false;
            unsigned sb = 0;
            for (; kt < t1; ++kt) {                       // hot loop
                const int rel = (kt - t0) & 1;
                sb = smem_base + rel * F5_KBYTES;
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                const bool nx = kt + 1 < t1;
                const bool moved = (kt * 32 + 32 > T) ? s_phase(sb, kt, nx, rel ^ 1, IC<1>{}) : s_phase(sb, kt, nx, rel ^ 1, IC<0>{});
                if (moved) { pending = true; break; }
                pv_phase(sb);
            }
            if (!pending) break;
#pragma unroll
            for (int d = 0; d < 16; ++d) {                // one 16-register block at a time through the VGPRs (no reordering across blocks:
#pragma unroll                                            // all 256 at once would spill the Q fragments)
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
                __builtin_amdgcn_sched_barrier(0);
            }
            pv_phase(sb);
            ++kt;
            // a wait the COMPILER sees (its own bookkeeping ignores the asm waits): whatever this cold path reloaded from scratch is
            // complete here, so that the hot loop's first MFMA is not made to wait, with vmcnt(0), behind the DMA it has just issued
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
        l_run += other_half(l_run);
        const int q = q0 + l31;
        if (whole) {
            // ---- normalise and store O[q][d] (this lane: q = l31, d = 32 blk + 8 (r >> 2) + 4 hh + (r & 3))
            const float inv = 1.f / l_run;
            if (q < T) {
                h16_t* ob = O + ((long long)b * T + q) * ldo;
#pragma unroll
                for (int d = 0; d < 16; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 pk = pack_h16x4(o_acc[d][4 * g] * inv, o_acc[d][4 * g + 1] * inv, o_acc[d][4 * g + 2] * inv, o_acc[d][4 * g + 3] * inv);
                        *(uint2*)(ob + d * 32 + 8 * g + 4 * hh) = pk;
                    }
            }
        } else {
            // ---- one part of a left-over block: unnormalised accumulators + (maximum, sum) per query
            const long long slot = (long long)(sid / S) * S + part;
            float* po = part_o + (slot * 128 + wave * 32 + l31) * 512;
#pragma unroll
            for (int d = 0; d < 16; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(po + d * 32 + 8 * g + 4 * hh) = float4{o_acc[d][4 * g], o_acc[d][4 * g + 1], o_acc[d][4 * g + 2], o_acc[d][4 * g + 3]};
            if (hh == 0) {
                float* pm = part_ml + (slot * 128 + wave * 32 + l31) * 2;
                pm[0] = m_run;
                pm[1] = l_run;
            }
        }
    }
}

// merge the S key-parts of each left-over query block: out = sum_p w_p O_p / sum_p w_p l_p, w_p = 2^((m_p - max m) sc)
__global__ __launch_bounds__(256) void flash512_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                h16_t* __restrict__ O, int T, int nqb, int first_blk, int Li, int S, int ldo,
                                                                float scale) {
    const int lb = blockIdx.x >> 2, qq = (blockIdx.x & 3) * 32 + (threadIdx.x >> 3);   // 32 queries per workgroup, 8 threads per query
    const int b = Li ? lb / Li : (first_blk + lb) / nqb;                                // (the kernel's block numbering)
    const int q = (Li ? nqb - Li + (lb - b * Li) : first_blk + lb - b * nqb) * 128 + qq;
    if (q >= T) return;
    const float sc = scale * 1.44269504088896340736f;
    float m = -1e30f;
    for (int p = 0; p < S; ++p) m = fmaxf(m, part_ml[(((long long)lb * S + p) * 128 + qq) * 2]);
    float l = 0.f;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    const int c0 = (threadIdx.x & 7) * 4;               // this thread: channels c0 + 32 i .. + 3, i = 0..15
    for (int p = 0; p < S; ++p) {
        const long long row = ((long long)lb * S + p) * 128 + qq;
        const float w = __builtin_amdgcn_exp2f((part_ml[row * 2] - m) * sc);
        l += w * part_ml[row * 2 + 1];
        const float* po = part_o + row * 512 + c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 v = *(const float4*)(po + 32 * i);
            acc[4 * i] += w * v.x; acc[4 * i + 1] += w * v.y; acc[4 * i + 2] += w * v.z; acc[4 * i + 3] += w * v.w;
        }
    }
    const float inv = 1.f / l;
    h16_t* ob = O + ((long long)b * T + q) * ldo + c0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *(uint2*)(ob + 32 * i) = pack_h16x4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv);
}

bool flash_attn512_supported(int C) { return C == 512 && !gp_sw().no_flash512; }
// Workgroups to launch: the CU count, or -- when the query blocks would leave half of the chip idle (one image at 768^2 is 72 blocks on
// 256 CUs) -- a multiple of the block count, so that EVERY block is cut along the keys into G / blocks parts.
static int flash512_grid(int nblocks, int ncu, int T) {
    if (nblocks >= ncu) return ncu;
    const int nt = (T + 31) / 32;
    int parts = ncu / nblocks;
    if (parts > nt / 8) parts = nt / 8;          // at least eight 32-key tiles per part
    return parts >= 2 ? parts * nblocks : nblocks;
}
// floats of workspace the launch needs (0: the blocks divide evenly over the workgroups)
long long flash_attn512_workspace_floats(int B, int T, int ncu) {
    const int nblocks = B * ((T + 127) / 128), G = flash512_grid(nblocks, ncu, T);
    const int L = nblocks % G;
    return L ? (long long)(G / L) * L * 128 * (512 + 2) : 0;
}
void launch_flash_attn512(const h16_t* q, const h16_t* k, const h16_t* vt, h16_t* out, float* ws, int B, int T, int ldq,
                          int ldk, int Tpad, int ldo, float scale, int ncu, hipStream_t s) {
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] { (void)hipFuncSetAttribute((const void*)flash_attn512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F5_LDS); });
    const int nqb = (T + 127) / 128, nblocks = B * nqb, G = flash512_grid(nblocks, ncu, T);
    const int rounds = nblocks / G, L = nblocks - rounds * G, S = L ? G / L : 0;
    float* part_o = ws;
    float* part_ml = ws ? ws + (long long)S * L * 128 * 512 : nullptr;
    const int dbg = gp_sw().f5_dbg;  // timing ablations only: 1 no K DMA, 2 no V DMA
    hipLaunchKernelGGL(flash_attn512_kernel, dim3(G), dim3(256), F5_LDS, s, q, k, vt, out, part_o, part_ml, B, T, ldq, ldk, Tpad,
                       ldo, scale, dbg);
    if (L) hipLaunchKernelGGL(flash512_combine_kernel, dim3(L * 4), dim3(256), 0, s, part_o, part_ml, out, T, nqb, rounds * G, (L % B == 0) ? L / B : 0,
                              S, ldo, scale);
}

// ---- cross-attention with a tiny constant context -------------------------------------------------------------------
// One thread per (row, head): q (64 bf16) against L keys/values held in fp32 (folded at load time).
__global__ __launch_bounds__(256) void cross_attn_small_kernel(const h16_t* __restrict__ q, const float* __restrict__ kc,
                                                                const float* __restrict__ vc, h16_t* __restrict__ out, long long nrh,
                                                                int C, int heads, int L) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrh) return;
    const long long row = idx / heads;
    const int h = (int)(idx - row * heads);
    float qv[64];
    const h16_t* qp = q + row * C + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 raw = *(const uint4*)(qp + i * 8);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { qv[i * 8 + 2 * k] = h16_lo(w[k]); qv[i * 8 + 2 * k + 1] = h16_hi(w[k]); }
    }
    float o[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = -1e30f, l = 0.f;
    for (int j = 0; j < L; ++j) {
        const float* kp = kc + (long long)j * C + h * 64;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s += qv[d] * kp[d];
        s *= 0.125f;
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), pj = __expf(s - mn);
        m = mn;
        l = l * a + pj;
        const float* vp = vc + (long long)j * C + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = o[d] * a + pj * vp[d];
    }
    const float inv = 1.f / l;
    h16_t* op = out + row * C + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 r;
        r.x = pack_h16x2(o[i * 8] * inv, o[i * 8 + 1] * inv);
        r.y = pack_h16x2(o[i * 8 + 2] * inv, o[i * 8 + 3] * inv);
        r.z = pack_h16x2(o[i * 8 + 4] * inv, o[i * 8 + 5] * inv);
        r.w = pack_h16x2(o[i * 8 + 6] * inv, o[i * 8 + 7] * inv);
        *(uint4*)(op + i * 8) = r;
    }
}

void launch_cross_attn_small(const h16_t* q, const float* kc, const float* vc, h16_t* out, int rows, int C, int L, hipStream_t s) {
    const int heads = C / 64;
    const long long nrh = (long long)rows * heads;
    hipLaunchKernelGGL(cross_attn_small_kernel, dim3((unsigned)((nrh + 255) / 256)), dim3(256), 0, s, q, kc, vc, out, nrh, C, heads, L);
}

// ---- row softmax (fp32 logits -> bf16 probabilities), one workgroup per row ----------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, h16_t* __restrict__ out, int T, int ld, float scale) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const float* x = in + row * ld;
    h16_t* y = out + row * ld;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float sc = scale * 1.44269504088896340736f;
    float mx = -1e30f;
    for (int i = tid; i < T; i += 256) mx = fmaxf(mx, x[i] * sc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < T; i += 256) sum += exp2f(x[i] * sc - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid; i < ld; i += 256) y[i] = i < T ? f_to_h16(exp2f(x[i] * sc - mx) * inv) : (h16_t)0;
}

// The same with the row held in registers (ld <= 1024 * NV floats, ld % 4 == 0): ONE 16-byte-per-lane read pass instead of three
// 4-byte ones, raw v_exp_f32 computed once per element, 8-byte stores.  HBM-bound: 6 B per score.
GP_DEV float4 load4(const float* p, int i) { return ((const float4*)p)[i]; }
GP_DEV float4 load4(const _Float16* p, int i) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    const h4_t h = ((const h4_t*)p)[i];
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <int NV, typename TIN>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const TIN* __restrict__ in, h16_t* __restrict__ out, int T, int ld, float scale) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const TIN* x = in + row * ld;
    uint2* y = (uint2*)(out + row * ld);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float sc = scale * 1.44269504088896340736f;
    const int nvec = ld >> 2;
    float4 v[NV];
    float mx = -1e30f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = k * 256 + tid;
        v[k] = i < nvec ? load4(x, i) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int e0 = i * 4;
        v[k].x = e0 + 0 < T ? v[k].x : -1e30f;
        v[k].y = e0 + 1 < T ? v[k].y : -1e30f;
        v[k].z = e0 + 2 < T ? v[k].z : -1e30f;
        v[k].w = e0 + 3 < T ? v[k].w : -1e30f;
        mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));  // raw-logit units (scale > 0)
    const float nm = -mx * sc;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k].x = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].x, sc, nm));
        v[k].y = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].y, sc, nm));
        v[k].z = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].z, sc, nm));
        v[k].w = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].w, sc, nm));
        sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = k * 256 + tid;
        if (i < nvec) y[i] = pack_h16x4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);  // masked tail: exp2(-huge) = 0
    }
}

void launch_softmax_rows(const float* in, h16_t* out, int rows, int T, int ld, float scale, hipStream_t s) {
    if ((ld & 3) == 0 && scale > 0.f && ld <= 16384) {
        if (ld <= 4096) hipLaunchKernelGGL((softmax_rows_reg_kernel<4, float>), dim3(rows), dim3(256), 0, s, in, out, T, ld, scale);
        else if (ld <= 9216) hipLaunchKernelGGL((softmax_rows_reg_kernel<9, float>), dim3(rows), dim3(256), 0, s, in, out, T, ld, scale);
        else hipLaunchKernelGGL((softmax_rows_reg_kernel<16, float>), dim3(rows), dim3(256), 0, s, in, out, T, ld, scale);
        return;
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, in, out, T, ld, scale);
}

// fp16 logits (written by the score GEMM with out_fp32 == 2): same kernel, half the read traffic; ld % 4 == 0, ld <= 16384
bool softmax_rows_f16_supported(int ld) { return (ld & 3) == 0 && ld <= 16384; }
void launch_softmax_rows_f16(const void* in, h16_t* out, int rows, int T, int ld, float scale, hipStream_t s) {
    const _Float16* x = (const _Float16*)in;
    if (ld <= 4096) hipLaunchKernelGGL((softmax_rows_reg_kernel<4, _Float16>), dim3(rows), dim3(256), 0, s, x, out, T, ld, scale);
    else if (ld <= 9216) hipLaunchKernelGGL((softmax_rows_reg_kernel<9, _Float16>), dim3(rows), dim3(256), 0, s, x, out, T, ld, scale);
    else hipLaunchKernelGGL((softmax_rows_reg_kernel<16, _Float16>), dim3(rows), dim3(256), 0, s, x, out, T, ld, scale);
}

GP_SAT_TU(attention)  // fp16 build: address of this translation unit's saturation flag (common.h)
