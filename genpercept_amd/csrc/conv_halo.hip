// conv3x3 (stride 1, pad 1; optionally on a nearest-x2-upsampled input) with halo reuse — the kernel for the large VAE /
// UNet feature maps, where the generic implicit GEMM (igemm.hip) is bound by operand loads: it re-fetches every input
// pixel once per tap.  Here a workgroup owns a 16x16 OUTPUT-PIXEL tile x 128 output channels and, per 64-channel chunk,
// stages the 18x18 input halo (10x10 source pixels in the x2-upsample case) in LDS ONCE; the nine taps then read shifted
// rows of that halo.  Only the weight tiles ([128 cout][64 cin] per tap) stream per K-step, through a 3-deep LDS-DMA ring.
//
//   LDS: 2 x halo buffer (48 KiB, double-buffered across channel chunks) + 3 x 16 KiB weight ring = 144 KiB, 1 workgroup/CU,
//        8 waves (4 pixel-rows-of-4 x 2 channel halves), each wave 64 pixels x 64 channels (16 accumulator tiles 16x16).
//   Per K-step (tap, chunk) and wave: 2 weight DMAs (+ 6 halo DMAs once per chunk), 16 ds_read_b128, 32 MFMAs
//        (v_mfma_f32_16x16x32_bf16) -- 5.0 KB of operand traffic per MFLOP instead of 11-15 in the generic kernel.
//   Sync: counted s_waitcnt vmcnt + one raw s_barrier per K-step; waves 4-7 issue their DMA before their MFMAs, waves 0-3
//        after (role split), so each SIMD overlaps one wave's matrix work with its partner's memory work.
//   Epilogue: epilogue.h (LDS-staged coalesced stores, fused bias / residual / activation).
#include "common.h"
#include "epilogue.h"
#include "kernels.h"

constexpr bool PRIO = true;   // s_setprio around the MFMA cluster made hipcc wait lgkmcnt(0) before the first MFMA
template <int N>
GP_DEV void halo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int HALO_LDS = 147456;   // 2 x 48 KiB halo + 3 x 16 KiB weights (>= the 128 KiB epilogue staging, also in the x2 case)
constexpr int GN_MAXC = 2048;      // fused input transform: per-channel scale/shift of one image live in the last 16 KiB of LDS

template <bool UPS>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const IGemmParams p) {
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2, NW = 8, TM = 64, TN = 64, FM = 4, FN = 4, FP = 2;
    constexpr int HW_ = UPS ? 10 : 18;                 // halo edge (source pixels)
    constexpr int HROWS = HW_ * HW_;                   // 100 / 324 halo pixels = LDS rows of 128 B
    constexpr int A_IT = (HROWS + 63) / 64;            // 8-row DMA groups per wave: 2 / 6 (tail groups hit the zero page)
    constexpr int A_BUF = A_IT * 8 * 1024;             // 16 KiB / 48 KiB
    constexpr int NB = 3, B_STAGE = BN * 128;          // weight ring
    constexpr int B_IT = 2;                            // weight DMA groups per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + 2 * A_BUF;
    float* const s_gn = (float*)(smem + HALO_LDS);      // [GN_MAXC] scale, [GN_MAXC] shift of this image (fused GroupNorm apply)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool second_half = wave >= NW / 2;
    const int a15 = lane & 15;

    // ---- tile coordinates -------------------------------------------------------------------------------------------------
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
    const int cpt = Cin >> 6, ns = 9 * cpt;

    const int chunk_a = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const int chunk_w = (lane & 7) ^ (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2));

    // ---- halo row descriptors: LDS row r <-> source pixel (sy0 + r / HW_, sx0 + r % HW_) --------------------------------------
    const int sy0 = UPS ? ty * 8 - 1 : ty * 16 - 1, sx0 = UPS ? tx * 8 - 1 : tx * 16 - 1;
    const bf16_t* h_ptr[A_IT];
    unsigned h_ok = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = (wave + NW * i) * 8 + (lane >> 3);
        const int hy = r / HW_, hx = r - hy * HW_;
        const int iy = sy0 + hy, ix = sx0 + hx;
        const bool ok = r < HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        h_ptr[i] = p.in + (((long long)b * Hi + iy) * Wi + ix) * Cin + chunk_a * 8;
        if (ok) h_ok |= 1u << i;
    }
    const bf16_t* zsrc_a = p.zero + chunk_a * 8;
    const bf16_t* zsrc_w = p.zero + chunk_w * 8;
    const bf16_t* w_ptr[B_IT];
    bool w_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + (wave + NW * i) * 8 + (lane >> 3);
        w_ok[i] = n < p.n_rows;
        w_ptr[i] = p.wt + (long long)n * p.ldw + chunk_w * 8;
    }

    // ---- fused input transform x -> act(x * scale[b][c] + shift[b][c]) applied to the staged halo (GroupNorm apply + SiLU) ----
    // Padding pixels must stay exactly 0 (the reference pads the NORMALISED tensor), hence the per-item validity mask.
    constexpr int T_IT = (HROWS * 8 + 511) / 512;      // 16-byte items per thread per halo
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
        __syncthreads();  // before any LDS-DMA is in flight: a later __syncthreads would drain the DMA ring
    }
    auto transform = [&](int cc) {
        char* buf = a_lds + (cc & 1) * A_BUF;
        const float* sc = s_gn + (cc << 6);
        const float* sh = s_gn + GN_MAXC + (cc << 6);
#pragma unroll
        for (int k = 0; k < T_IT; ++k) {
            if (!((t_ok >> k) & 1u)) continue;
            const int item = tid + 512 * k, r = item >> 3;
            const int ls = ((item & 7) ^ ((r >> 1) & 7)) << 3;  // first channel (within the chunk) of this 16-byte slot
            uint4* ptr = (uint4*)(buf + item * 16);
            const uint4 raw = *ptr;
            const float4 s0 = *(const float4*)(sc + ls), s1 = *(const float4*)(sc + ls + 4);
            const float4 h0 = *(const float4*)(sh + ls), h1 = *(const float4*)(sh + ls + 4);
            float v[8] = {bflo(raw.x) * s0.x + h0.x, bfhi(raw.x) * s0.y + h0.y, bflo(raw.y) * s0.z + h0.z, bfhi(raw.y) * s0.w + h0.w,
                          bflo(raw.z) * s1.x + h1.x, bfhi(raw.z) * s1.y + h1.y, bflo(raw.w) * s1.z + h1.z, bfhi(raw.w) * s1.w + h1.w};
            if (p.in_silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *ptr = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS writes are done ...
        __builtin_amdgcn_s_barrier();                        // ... and so are everybody else's (raw: must not drain vmcnt)
    };

    auto stage_halo = [&](int cc) {
        char* dst = a_lds + (cc & 1) * A_BUF;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bf16_t* src = ((h_ok >> i) & 1u) ? h_ptr[i] + (cc << 6) : zsrc_a;
            glds16(src, dst + (wave + NW * i) * 1024);
        }
    };
    auto stage_w = [&](int slot, int tap, int cc) {
        char* dst = b_lds + slot * B_STAGE;
        const int woff = tap * Cin + (cc << 6);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bf16_t* src = w_ok[i] ? w_ptr[i] + woff : zsrc_w;
            glds16(src, dst + (wave + NW * i) * 1024);
        }
    };

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int xr_w = ((a15 >> 1) & 1) | (((a15 >> 2) & 1) << 1) | (((a15 >> 3) & 1) << 2);
    const int w_row_off = (wn * TN + 8 * (a15 >> 2) + (a15 & 3)) * 128;
    auto compute = [&](int slot, int cc, int ky, int kx) {
        const char* ab = a_lds + (cc & 1) * A_BUF;
        const char* wb = b_lds + slot * B_STAGE;
        int row[FM];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int py = 4 * wm + j;
            if (UPS) row[j] = (((py + ky - 1) >> 1) + 1) * HW_ + (((a15 + kx - 1) >> 1) + 1);
            else row[j] = (py + ky) * HW_ + a15 + kx;
        }
        bf16x8_t wf[2][FN], xf[2][FM];  // all 16 fragment reads first: the second k-half's LDS latency hides under the first half's MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sl = kk * 4 + (lane >> 4);
            const int so_w = (sl ^ xr_w) << 4;
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[kk][i] = *(const bf16x8_t*)(wb + w_row_off + (i >> 1) * 4096 + (i & 1) * 512 + so_w);
#pragma unroll
            for (int j = 0; j < FM; ++j) xf[kk][j] = *(const bf16x8_t*)(ab + row[j] * 128 + ((sl ^ ((row[j] >> 1) & 7)) << 4));
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: bias, halo of chunk 0, weight tiles of steps 0 and 1 ---------------------------------------
    // (bias first: an ordinary load issued while LDS-DMA is in flight makes hipcc drain vmcnt to 0 at its first use)
    float bcol[FP][8];
    load_bias_cols<FP>(p, 0, n0 + wn * TN, 8 * (lane >> 4), bcol);
    stage_halo(0);
    stage_w(0, 0, 0);
    stage_w(1, 1, 0);  // ns >= 9 always

    // ---- main loop over (chunk, tap) ------------------------------------------------------------------------------------------
    int tap = 0, cc = 0, ky = 0, kx = 0;            // current step
    int t2 = 2, c2 = 0;                             // (tap, chunk) of step s + 2
    int slot = 0, slot2 = 2;
    for (int s = 0; s < ns; ++s) {
        // loads issued after B(s) that may stay in flight: B(s+1), and halo(cc+1) when it was issued at this chunk's tap 0
        const bool more_w = s + 1 < ns;
        const bool halo_fly = (tap == 1 || tap == 2) && cc + 1 < cpt;
        if (halo_fly) { if (more_w) halo_wait_vm<A_IT + B_IT>(); else halo_wait_vm<A_IT>(); }
        else { if (more_w) halo_wait_vm<B_IT>(); else halo_wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        const bool issue_w = s + 2 < ns, issue_h = tap == 0 && cc + 1 < cpt;
        if (second_half) {
            if (issue_w) stage_w(slot2, t2, c2);
            if (issue_h) stage_halo(cc + 1);
        }
        if (fused && tap == 0) transform(cc);
        compute(slot, cc, ky, kx);
        if (!second_half) {
            if (issue_w) stage_w(slot2, t2, c2);
            if (issue_h) stage_halo(cc + 1);
        }
        slot = slot == NB - 1 ? 0 : slot + 1;
        slot2 = slot2 == NB - 1 ? 0 : slot2 + 1;
        if (++kx == 3) { kx = 0; ++ky; }
        if (++tap == 9) { tap = 0; ky = 0; ++cc; }
        if (++t2 == 9) { t2 = 0; ++c2; }
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------------
    conv_epilogue<BM, BN, WM, WN, 512>(p, acc, bcol, n0, 0, wave, lane, smem, [&](int pr) {
        const int oy = ty * 16 + (pr >> 4), ox = tx * 16 + (pr & 15);
        return (oy < Ho && ox < Wo) ? (b * Ho + oy) * Wo + ox : -1;
    });
}

bool conv_halo_applicable(const IGemmParams& p) {
    if (p.ks != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.batch > 1 || p.out_fp32 || p.act == GP_ACT_GEGLU) return false;
    if (p.bias_mode == GP_BIAS_ROW || (p.ldo & 7)) return false;
    if (p.in_scale && p.Cin > GN_MAXC) return false;
    if (p.ups) {
        if (p.Hu != 2 * p.Hi || p.Wu != 2 * p.Wi || p.Ho != p.Hu || p.Wo != p.Wu) return false;
    } else if (p.Ho != p.Hi || p.Wo != p.Wi) return false;
    return p.Ho >= 16 && p.Wo >= 16;
}

void launch_conv_halo(const IGemmParams& p, hipStream_t s) {
    const int ncols = p.N > p.n_store ? p.N : p.n_store;
    const int tiles = ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16) * p.B * ((ncols + 127) / 128);
    constexpr int LDS = HALO_LDS + 2 * GN_MAXC * (int)sizeof(float);  // 160 KiB: the whole LDS of a CU
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    if (p.ups) hipLaunchKernelGGL(conv3x3_halo_kernel<true>, dim3(tiles), dim3(512), LDS, s, p);
    else hipLaunchKernelGGL(conv3x3_halo_kernel<false>, dim3(tiles), dim3(512), LDS, s, p);
}
