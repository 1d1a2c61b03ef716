// conv_few: the VAE decoder's tail in ONE kernel
//     GroupNorm(conv_norm_out) apply + SiLU  ->  conv3x3(128 -> 3, pad 1)  ->  [mean of the 3 channels]  ->  clip(-1, 1), (x + 1) / 2  ->  fp32 NCHW
// (diffusers AutoencoderKL.decoder: conv_norm_out, conv_act, conv_out -- call site genpercept_pipeline.py:521-525; the clip / shift of
// single_infer, :469-472).  r2 ran it as three launches: gn_apply (reads and rewrites the 604 MB tensor: 222 us at batch 4, 768x768), the
// generic implicit GEMM with its 32-column tile (456 us: it re-fetches every input pixel once per tap, 5.4 GB through the vector caches
// for 16 GFLOP of useful work) and decode_epilogue (7 us, after rounding the three channels to bf16).  Here the input is read ONCE:
//   * a persistent 512-thread workgroup per CU walks 16x16-pixel tiles; per (tile, 64-channel chunk) step the 18x18 input halo
//     (41 KiB) arrives by LDS-DMA into a 3-deep ring, two steps ahead of its use (HBM-bound kernel: 767 MB at batch 4);
//   * the GroupNorm affine (+ SiLU) is applied to the halo in place (padding pixels stay exactly zero), per-(image, channel) scale /
//     shift rows arrive by LDS-DMA as well (an ordinary load next to an LDS-DMA in flight makes hipcc drain vmcnt to 0);
//   * the conv runs on v_mfma_f32_16x16x32 with the weights as A operand: all 9 x 2 weight tiles of the four real output rows stay
//     resident in LDS (9 KiB; MFMA rows 4..15 read one shared row of zeros), each wave owns 2 x 16 output pixels;
//   * the epilogue works on the fp32 accumulators (one rounding less than r2's bf16 detour) and writes 64-byte row segments of the
//     fp32 NCHW result.
// HBM-bound: algorithmic bytes = 1.27 x input (halo overlap) + output; MFMA work is 2.25 % of a full 128-column tile's.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int CF_HW = 18, CF_HROWS = CF_HW * CF_HW, CF_GROUPS = (CF_HROWS + 7) / 8;   // 324 halo pixels = 41 DMA groups of 8 rows
constexpr int CF_ABUF = CF_GROUPS * 1024;                                              // 41 KiB per halo buffer
constexpr int CF_CPT = 2;                                                              // 64-channel chunks (Cin = 128)
constexpr int CF_W_BYTES = 9 * CF_CPT * 4 * 128;                                       // [tap][chunk][4 rows][64 ch] = 9 KiB
constexpr int CF_ZERO_OFF = CF_W_BYTES;                                                // 128 bytes of zeros (MFMA rows 4..15)
constexpr int CF_H_OFF = CF_W_BYTES + 256;
constexpr int CF_T_OFF = CF_H_OFF + 3 * CF_ABUF;                                       // 3 tables: [2][128] fp32 scale | shift of one image
constexpr int CF_DUMP_OFF = CF_T_OFF + 3 * 1024;                                       // DMA groups beyond the 41st land here
constexpr int CF_LDS = CF_DUMP_OFF + 1024;
constexpr int CF_AIT = 6;                                                              // halo DMA instructions per wave and step
}  // namespace

struct ConvFewParams {
    const h16_t* in;        // NHWC [B][H][W][128]
    const h16_t* wt;        // packed [rows >= 4][9][128]
    const float* bias;      // [3]
    const float* scale;     // [B][128]  GroupNorm apply: x * scale + shift
    const float* shift;
    const h16_t* zero;      // >= 256 bytes of zeros
    float* out;             // fp32 NCHW [B][mean3 ? 1 : 3][H][W]
    int B, H, W, silu, mean3, raw;
};

template <int SILU>
__global__ __launch_bounds__(512) void conv_few_kernel(const ConvFewParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a15 = lane & 15, q = lane >> 4;
    const unsigned base = (unsigned)(unsigned long long)smem;
    constexpr int Cin = 64 * CF_CPT;

    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    const int tiles_img = tiles_x * tiles_y, tiles = tiles_img * p.B;
    const int my_tiles = blockIdx.x < tiles ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nsteps = my_tiles * CF_CPT;

    // ---- resident weights: rows 0..3 of every (tap, chunk) tile, 128 bytes each; slot s of row r at r * 128 + 16 * s ------------------------
    for (int i = tid; i < 9 * CF_CPT * 4 * 8; i += 512) {
        const int s = i & 7, r = (i >> 3) & 3, tc = i >> 5, cc = tc % CF_CPT, tap = tc / CF_CPT;
        *(uint4*)(smem + (tc * 4 + r) * 128 + s * 16) = *(const uint4*)(p.wt + ((long long)r * 9 + tap) * Cin + cc * 64 + s * 8);
    }
    if (tid < 16) *(uint4*)(smem + CF_ZERO_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
    float bias3[3] = {p.bias ? p.bias[0] : 0.f, p.bias ? p.bias[1] : 0.f, p.bias ? p.bias[2] : 0.f};
    __syncthreads();

    // ---- DMA of one step (tile, chunk): the wave's six halo groups (+ the image's scale / shift rows with chunk 0) -----------------------------
    auto tile_coords = [&](int lt, int& b, int& ty, int& tx) __attribute__((always_inline)) {
        const int t = blockIdx.x + lt * gridDim.x;
        b = t / tiles_img;
        const int r = t - b * tiles_img;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
    };
    auto stage = [&](int s) __attribute__((always_inline)) {
        const int lt = s / CF_CPT, cc = s - lt * CF_CPT;
        int b, ty, tx;
        tile_coords(lt, b, ty, tx);
        char* dst = smem + CF_H_OFF + (s % 3) * CF_ABUF;
        const h16_t* img = p.in + (long long)b * p.H * p.W * Cin + cc * 64;
#pragma unroll
        for (int i = 0; i < CF_AIT; ++i) {
            const int g = wave + 8 * i;
            const int r = g * 8 + (lane >> 3);
            const int hy = r / CF_HW, hx = r - hy * CF_HW;
            const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
            const bool ok = r < CF_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int chunk = (lane & 7) ^ (hx & 7);
            const h16_t* src = ok ? img + ((long long)iy * p.W + ix) * Cin + chunk * 8 : p.zero;
            glds16(src, g < CF_GROUPS ? dst + g * 1024 : smem + CF_DUMP_OFF);
        }
        if (cc == 0) {  // scale | shift of image b: 2 x 128 floats = 64 lanes x 16 bytes (every wave issues it: same bytes, same place, equal DMA counts)
            const float* src = (lane < 32 ? p.scale : p.shift) + (long long)b * Cin + (lane & 31) * 4;
            glds16(src, smem + CF_T_OFF + (lt % 3) * 1024);
        }
    };

    // ---- fragment bases -------------------------------------------------------------------------------------------------------------------------
    unsigned xb[3][2], wbase[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sl = kk * 4 + q;
        wbase[kk] = a15 < 4 ? base + a15 * 128 + sl * 16 : base + CF_ZERO_OFF;  // rows 4..15 of the MFMA's A operand are zero
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = a15 + kx;
            xb[kx][kk] = base + CF_H_OFF + ((2 * wave) * CF_HW + hx) * 128 + ((sl ^ (hx & 7)) << 4);
        }
    }
    const bool wreal = a15 < 4;

    if (nsteps > 0) stage(0);
    if (nsteps > 1) stage(1);
    f32x4_t acc[2];
    for (int s = 0; s < nsteps; ++s) {
        const int lt = s / CF_CPT, cc = s - lt * CF_CPT;
        // my DMA of step s has landed (step s+1's may stay in flight: 6 instructions, 7 when it carries a table)
        if (s + 1 < nsteps) { if (cc == CF_CPT - 1) wait_vm<CF_AIT + 1>(); else wait_vm<CF_AIT>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int b, ty, tx;
        tile_coords(lt, b, ty, tx);
        const unsigned hbuf = (unsigned)((s % 3) * CF_ABUF);
        // ---- GroupNorm apply (+ SiLU) of the halo, in place; pixels outside the image stay zero -------------------------------------------------
        {
            const unsigned tab = base + CF_T_OFF + (lt % 3) * 1024 + cc * 256;   // scale[cc * 64 ..], shift 512 bytes further
#pragma unroll
            for (int k = 0; k < (CF_HROWS * 8 + 511) / 512; ++k) {
                const int item = tid + 512 * k, r = item >> 3;
                const int hy = r / CF_HW, hx = r - hy * CF_HW;
                const bool ok = r < CF_HROWS && (unsigned)(ty * 16 - 1 + hy) < (unsigned)p.H && (unsigned)(tx * 16 - 1 + hx) < (unsigned)p.W;
                if (ok) {
                    const int grp = (item & 7) ^ (hx & 7);   // logical 8-channel group held by this physical slot
                    const unsigned a_item = base + CF_H_OFF + hbuf + item * 16;
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t raw = *(const __attribute__((address_space(3))) u32x4_t*)a_item;
                    const f32x4_t s0 = *(lds_f4_ptr)(tab + grp * 32), s1 = *(lds_f4_ptr)(tab + grp * 32 + 16);
                    const f32x4_t h0 = *(lds_f4_ptr)(tab + 512 + grp * 32), h1 = *(lds_f4_ptr)(tab + 512 + grp * 32 + 16);
                    float v[8] = {__builtin_fmaf(h16_lo(raw.x), s0.x, h0.x), __builtin_fmaf(h16_hi(raw.x), s0.y, h0.y),
                                  __builtin_fmaf(h16_lo(raw.y), s0.z, h0.z), __builtin_fmaf(h16_hi(raw.y), s0.w, h0.w),
                                  __builtin_fmaf(h16_lo(raw.z), s1.x, h1.x), __builtin_fmaf(h16_hi(raw.z), s1.y, h1.y),
                                  __builtin_fmaf(h16_lo(raw.w), s1.z, h1.z), __builtin_fmaf(h16_hi(raw.w), s1.w, h1.w)};
                    if (SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                    }
                    const u32x4_t ov = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
                    *(__attribute__((address_space(3))) u32x4_t*)a_item = ov;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nsteps) stage(s + 2);  // into the buffer step s-1 computed from (everybody is past it)
        // ---- 9 taps x 2 k-halves: one weight fragment, two pixel fragments, two MFMAs -----------------------------------------------------------
        if (cc == 0) { acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        const unsigned wofs = wreal ? (unsigned)(cc * 4 * 128) : 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const h16x8_t wf = *(lds_frag_ptr)(wbase[kk] + wofs + (wreal ? tap * CF_CPT * 4 * 128 : 0));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const h16x8_t xf = *(lds_frag_ptr)(xb[kx][kk] + hbuf + (j + ky) * CF_HW * 128);
                    acc[j] = mfma_16x16x32(wf, xf, acc[j]);
                }
            }
        }
        // ---- epilogue of the tile: lanes 0..15 hold channels 0..3 of pixel (row 2 wave + j, column a15) ----------------------------------------------
        if (cc == CF_CPT - 1 && q == 0) {
            const long long HWp = (long long)p.H * p.W;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = ty * 16 + 2 * wave + j, ox = tx * 16 + a15;
                if (oy < p.H && ox < p.W) {
                    float c[3] = {acc[j].x + bias3[0], acc[j].y + bias3[1], acc[j].z + bias3[2]};
                    const long long pix = (long long)oy * p.W + ox;
                    if (p.mean3) {
                        float v = (c[0] + c[1] + c[2]) / 3.0f;
                        if (!p.raw) v = (fminf(fmaxf(v, -1.f), 1.f) + 1.f) * 0.5f;
                        p.out[(long long)b * HWp + pix] = v;
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            float v = c[k];
                            if (!p.raw) v = (fminf(fmaxf(v, -1.f), 1.f) + 1.f) * 0.5f;
                            p.out[((long long)b * 3 + k) * HWp + pix] = v;
                        }
                    }
                }
            }
        }
    }
}

bool conv_few_applicable(int Cin, int Cout, int H, int W) {
    const bool off = gp_sw().no_conv_few;  // A/B switch
    return !off && Cin == 64 * CF_CPT && Cout == 3 && H >= 1 && W >= 1;
}

void launch_conv_few(const h16_t* in, const h16_t* wt, const float* bias, const float* scale, const float* shift, const h16_t* zero, float* out, int B,
                     int H, int W, int silu, int mean3, int raw, int ncu, hipStream_t s) {
    ConvFewParams p{in, wt, bias, scale, shift, zero, out, B, H, W, silu, mean3, raw};
    static unsigned long long attr_mask = 0;
    gp_once_per_device(&attr_mask, [&] {
        (void)hipFuncSetAttribute((const void*)conv_few_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS);
        (void)hipFuncSetAttribute((const void*)conv_few_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS);
    });
    const int tiles = ((W + 15) / 16) * ((H + 15) / 16) * B;
    int grid = ncu > 0 ? ncu : 256;
    if (grid > tiles) grid = tiles;
    if (silu) hipLaunchKernelGGL(conv_few_kernel<1>, dim3(grid), dim3(512), CF_LDS, s, p);
    else hipLaunchKernelGGL(conv_few_kernel<0>, dim3(grid), dim3(512), CF_LDS, s, p);
}

GP_SAT_TU(conv_few)  // fp16 build: address of this translation unit's saturation flag (common.h)
