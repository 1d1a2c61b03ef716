// Attention kernels (K4/K5/K6 of SURVEY.md §2.3).
//
// flash_attn64: UNet self-attention, head_dim 64, flash-style (scores never leave the CU) on v_mfma_f32_32x32x16_bf16.
//   Per 256-thread workgroup: 128 query rows of one (image, head); each wave owns 32 query rows.  K tiles [64 keys][64 d]
//   and V^T tiles [64 d][64 keys] are brought in by LDS-DMA (double-buffered, swizzled like the GEMM tiles).
//   The wave computes S^T = K Q^T (keys x queries), so every lane holds 32 scores of ONE query (q = lane & 31):
//   the softmax max/sum/rescale are lane-local (one cross-half exchange with lane ^ 32), and the probabilities are
//   already in B-operand order for O^T += V^T P^T -- no LDS round trip, no permutes.  The V^T A-operand is read in the
//   matching key order (keys 16j+4h+{0..3} and 16j+8+4h+{0..3} for half h), which is why V is produced transposed
//   ([C][Tpad], zero-padded beyond T) by the projection GEMM.
// cross_attn_small: cross-attention against the constant, tiny text context (L = 2 for GenPercept's empty prompt).
// softmax_rows: row softmax for the GEMM-based single-head VAE attention (head_dim 512).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

// 32-row MFMA fragments (rows = lane & 31, slots s / s+1 for the two half-waves) are conflict-free with slot ^ ((row >> 1) & 7);
// the conv kernels' slot ^ (row & 7) is not for this pattern (checked exhaustively), so the attention tiles keep their own swizzle.
GP_DEV int attn_off128(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// NKB: 32-key blocks per KV tile (2: 64 keys, 4: 128 keys -- one online-softmax update, one barrier and one DMA wait per 128 keys,
// longer independent MFMA runs)
// NST: depth of the K / V ring in LDS (16 KiB per stage at NKB = 2).  2 = one tile ahead, a full vmcnt(0) + barrier per tile; 3 = two tiles
// ahead with COUNTED waits and one raw s_barrier per tile -- the DMA of tile t+2 is issued before tile t computes and is only waited for at
// the top of tile t+2, so an L2 / MALL round trip no longer has to fit inside one tile's compute.
template <int NKB, int NST>
__global__ __launch_bounds__(256, 2) void flash_attn64_kernel(const h16_t* __restrict__ Q, const h16_t* __restrict__ K,
                                                            const h16_t* __restrict__ Vt, h16_t* __restrict__ O,
                                                            const h16_t* __restrict__ zero, int T, int heads, int ldq, int ldk, int Tpad,
                                                            int ldo) {
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
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NKB; ++i) {  // K rows: 8 * (wave + 4 i) + lane / 8
            const int g = wave + 4 * i;
            const int key = kt * KEYS + g * 8 + (lane >> 3);
            const h16_t* src = key < T ? Kb + (long long)key * ldk + chunk * 8 : zero + chunk * 8;
            glds16(src, sb + g * 1024);
        }
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // V^T rows = head channels d
                const int r = (wave + 4 * i) * 8 + (lane >> 3);
                const int k0 = kt * KEYS + hf * 64;
                const h16_t* vsrc = k0 < Tpad ? Vb + (long long)r * Tpad + k0 + chunk * 8 : zero + chunk * 8;  // (Tpad % 64 == 0)
                glds16(vsrc, sb + KBYTES + hf * 8192 + (wave + 4 * i) * 1024);
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
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) u32x2_t* lds_u2_ptr;

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
            const int kbase = kt * KEYS + 4 * hh;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= T) s_acc[kb][r] = -1e30f;
        }
        float mx = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);              // raw-score units
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
        const float nm = -m_new * sc;
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s_acc[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, nm));
                rs += s_acc[kb][r];
            }
        l_run = l_run * alpha + rs;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {  // (uniform) some query's maximum moved: rescale the accumulators
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        // ---- O^T += V^T P^T: k-steps (kb, j) of 16 keys; this lane's P for its own query is the B operand
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                union { h16x8_t v; unsigned u[4]; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_h16x2_ns(s_acc[kb][8 * j + 2 * e], s_acc[kb][8 * j + 2 * e + 1]);  // probabilities: in [0, 1]
                const int ko = (kb & 1) * 32 + 16 * j + 4 * hh;  // key offset inside the 64-key half (multiple of 4)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int row = d * 32 + l31;
                    const unsigned vr = sb + KBYTES + (kb >> 1) * 8192 + row * 128;
                    const int sw = (row >> 1) & 7;
                    union { h16x8_t v; u32x2_t h2[2]; } vf;
                    vf.h2[0] = *(lds_u2_ptr)(vr + ((((ko >> 3)) ^ sw) << 4) + (ko & 7) * 2);
                    vf.h2[1] = *(lds_u2_ptr)(vr + ((((ko >> 3) + 1) ^ sw) << 4) + (ko & 7) * 2);
                    o_acc[d] = mfma_32x32x16(vf.v, pf.v, o_acc[d]);
                }
            }
    };

    constexpr int LPS = NKB + 2 * NH;  // LDS-DMA instructions per wave and stage
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nt) stage(s0, s0);
    int nxt = NST - 1;
    for (int kt = 0; kt < nt; ++kt) {
        // my DMAs of tile kt have landed (NST - 2 younger stages may stay in flight) ...
        const int ahead = min(NST - 2, nt - 1 - kt);
        if (ahead >= 1) wait_vm<LPS>(); else wait_vm<0>();
        static_assert(NST == 2 || NST == 3, "ring depth");
        // ... and after the barrier everybody's have, and everybody is done reading the slot of tile kt - 1, which tile kt + NST - 1 refills
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nt) stage(nxt, kt + NST - 1);
        if (kt * KEYS + KEYS > T) tile(kt, IC<1>{}); else tile(kt, IC<0>{});
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
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

void launch_flash_attn64(const h16_t* q, const h16_t* k, const h16_t* vt, h16_t* out, const h16_t* zero, int B, int T, int heads,
                         int ldq, int ldk, int Tpad, int ldo, hipStream_t s) {
    dim3 grid(((T + 127) / 128) * heads * B);
    // 64-key tiles: 194 VGPRs as hipcc 7.2 allocates them, i.e. two waves per SIMD (forcing three, __launch_bounds__(256, 3), spills 38 registers).
    // Measured alternatives, all slower at T = 9216 (610 us, kernel only): 128-key tiles -8 % (r1); a three-stage K / V ring with counted waits
    // -2 % (stays as a switch, GENPERCEPT_FLASH_RING3: K / V latency is not what the kernel waits for); two 32-query blocks per wave so that each
    // K / V^T fragment read feeds two MFMAs: 1100 us with 256 VGPRs + 42 spilled, 1050 us at one wave per SIMD; one online-softmax update per
    // 32 keys instead of 64 (165 VGPRs): 644 us.  r2 ablations of the shipped shape: no softmax arithmetic 494 us, no P.V MFMAs 587 us, neither
    // 442 us, additionally no K.Q^T MFMAs 249 us (the LDS fragment reads, DMA and loop alone), no waits / barriers 601 us.
    static const bool ring2 = getenv("GENPERCEPT_FLASH_RING3") == nullptr;
    if (ring2) hipLaunchKernelGGL((flash_attn64_kernel<2, 2>), grid, dim3(256), 2 * 16384, s, q, k, vt, out, zero, T, heads, ldq, ldk, Tpad, ldo);
    else hipLaunchKernelGGL((flash_attn64_kernel<2, 3>), grid, dim3(256), 3 * 16384, s, q, k, vt, out, zero, T, heads, ldq, ldk, Tpad, ldo);
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
