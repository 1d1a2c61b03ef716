// Layout / elementwise kernels (K9-K12 of SURVEY.md §2.3).  All HBM-bound; vectorised to 16 B per lane where the
// layout allows, grid-stride loops capped at 2048 workgroups.
#include "common.h"
#include "kernels.h"

static inline unsigned grid_for(long long n, int per_block = 256) {
    long long g = (n + per_block - 1) / per_block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// RGB [B][3][H][W] (uint8 0..255, or float already normalised to [-1,1]) -> NHWC bf16 with Cpad channels (3 real + zeros).
// u8 path computes x/255*2-1 in fp32 (genpercept_pipeline.py:245).
__global__ __launch_bounds__(256) void rgb_prologue_kernel(const void* __restrict__ rgb, int is_u8, h16_t* __restrict__ out, int B,
                                                            long long HW, int Cpad) {
    const long long n = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / HW, p = i - b * HW;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long long src = (b * 3 + c) * HW + p;
            v[c] = is_u8 ? ((float)((const unsigned char*)rgb)[src] / 255.0f * 2.0f - 1.0f) : ((const float*)rgb)[src];
        }
        h16_t* o = out + i * Cpad;
        uint4 first;
        first.x = pack_h16x2(v[0], v[1]);
        first.y = pack_h16x2(v[2], 0.f);
        first.z = first.w = 0u;
        *(uint4*)o = first;
        const uint4 zz = make_uint4(0, 0, 0, 0);
        for (int c = 8; c < Cpad; c += 8) *(uint4*)(o + c) = zz;
    }
}
void launch_rgb_prologue(const void* rgb, int is_u8, h16_t* out, int B, int H, int W, int Cpad, hipStream_t s) {
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(rgb_prologue_kernel, dim3(grid_for(n)), dim3(256), 0, s, rgb, is_u8, out, B, (long long)H * W, Cpad);
}

// channel concat of two NHWC tensors (UNet skip connections: [hidden, skip])
__global__ __launch_bounds__(256) void concat_kernel(const h16_t* __restrict__ a, int Ca, const h16_t* __restrict__ b, int Cb,
                                                      h16_t* __restrict__ out, long long pixels) {
    const int va = Ca >> 3, vb = Cb >> 3, vt = va + vb;
    const long long n = pixels * vt;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i / vt;
        const int v = (int)(i - p * vt);
        const uint4 x = v < va ? *(const uint4*)(a + p * Ca + v * 8) : *(const uint4*)(b + p * Cb + (v - va) * 8);
        *(uint4*)(out + i * 8) = x;
    }
}
void launch_concat(const h16_t* a, int Ca, const h16_t* b, int Cb, h16_t* out, long long pixels, hipStream_t s) {
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(pixels * ((Ca + Cb) / 8))), dim3(256), 0, s, a, Ca, b, Cb, out, pixels);
}

// The same concat, also leaving the GroupNorm statistics of the tensor it writes: per (bm consecutive pixels, channel) {sum, sum of
// squares} in the layout of the conv epilogues (mode 0 of gn_finalize_tiles_kernel), so the resnet that consumes the concatenation
// skips its statistics read pass.  A thread owns one 8-channel slot for the bm pixels of its tile: no cross-thread reduction.
__global__ __launch_bounds__(256) void concat_stats_kernel(const h16_t* __restrict__ a, int Ca, const h16_t* __restrict__ b, int Cb,
                                                            h16_t* __restrict__ out, int bm, float* __restrict__ part) {
    const int va = Ca >> 3, vt = (Ca + Cb) >> 3, C = Ca + Cb;
    const int v = blockIdx.y * 256 + threadIdx.x;
    if (v >= vt) return;
    const long long p0 = (long long)blockIdx.x * bm;
    const bool from_a = v < va;
    const h16_t* src = from_a ? a + p0 * Ca + v * 8 : b + p0 * Cb + (v - va) * 8;
    const int ld = from_a ? Ca : Cb;
    h16_t* dst = out + p0 * C + v * 8;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
#pragma unroll 8
    for (int r = 0; r < bm; ++r) {  // (eight 16-byte loads in flight per thread: r3's four left the pass at half the HBM rate)
        const uint4 x = *(const uint4*)(src + (long long)r * ld);
        *(uint4*)(dst + (long long)r * C) = x;
        const float f[8] = {h16_lo(x.x), h16_hi(x.x), h16_lo(x.y), h16_hi(x.y), h16_lo(x.z), h16_hi(x.z), h16_lo(x.w), h16_hi(x.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
    float* po = part + ((long long)blockIdx.x * C + v * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { po[2 * e] = s[e]; po[2 * e + 1] = q[e]; }
}
// pixels per statistics tile: the largest divisor of an image's pixel count among {64, 48, 32, 16} that still gives the launch ~2 workgroups
// per CU (r3 always took the largest: the UNet's 24x24 maps then ran on 72 workgroups and the pass was latency-bound, 34 us for 24 MB)
int concat_stats_bm(long long hw, long long pixels, int channels) {
    const long long slices = (channels / 8 + 255) / 256;
    int best = 0;
    for (int bm : {64, 48, 32, 16})
        if (hw % bm == 0) {
            best = bm;  // (the smallest valid one if none reaches the target)
            if ((pixels / bm) * slices >= 512) return bm;
        }
    return best;
}
void launch_concat_stats(const h16_t* a, int Ca, const h16_t* b, int Cb, h16_t* out, long long pixels, int bm, float* part, hipStream_t s) {
    hipLaunchKernelGGL(concat_stats_kernel, dim3((unsigned)(pixels / bm), ((Ca + Cb) / 8 + 255) / 256), dim3(256), 0, s, a, Ca, b, Cb, out, bm, part);
}

// fp32 NCHW -> bf16 NHWC with zero-padded channels (stage-level entry points: latents / features handed in by the host)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_kernel(const float* __restrict__ in, h16_t* __restrict__ out, int B, int C,
                                                                long long HW, int Cpad) {
    const long long n = (long long)B * HW * Cpad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const long long bp = i / Cpad, b = bp / HW, p = bp - b * HW;
        out[i] = c < C ? f_to_h16(in[(b * C + c) * HW + p]) : (h16_t)0;
    }
}
void launch_nchw_f32_to_nhwc(const float* in, h16_t* out, int B, int C, int H, int W, int Cpad, hipStream_t s) {
    hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel, dim3(grid_for((long long)B * H * W * Cpad)), dim3(256), 0, s, in, out, B, C,
                       (long long)H * W, Cpad);
}

// ---- multi-step archs (marigold / rgb_blending; genpercept_pipeline.py:413-422,447-465) ---------------------------------------------
// The denoising state lives in fp32, pixel-major [B*h*w][L]; the UNet reads its 16-bit copy from channels [off, off+L) of the latent
// tensor the encoder wrote (marigold: [rgb_latent, pred_latent] in channels 0..2L-1, "this order is important" :452; rgb_blending:
// the state replaces the rgb latent in channels 0..L-1).
__global__ __launch_bounds__(256) void ddim_init_kernel(const float* __restrict__ noise, h16_t* __restrict__ lat, float* __restrict__ sample,
                                                         int B, long long HW, int L, int ld, int off) {
    const long long n = (long long)B * HW * L;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % L);
        const long long bp = i / L, b = bp / HW, p = bp - b * HW;
        const float s = noise ? noise[(b * L + c) * HW + p] : h16_to_f(lat[bp * ld + c]);
        sample[i] = s;
        if (noise) lat[bp * ld + off + c] = f_to_h16(s);
    }
}
void launch_ddim_init(const float* noise_nchw, h16_t* lat, float* sample, int B, int H, int W, int L, int ld, int off, hipStream_t s) {
    hipLaunchKernelGGL(ddim_init_kernel, dim3(grid_for((long long)B * H * W * L)), dim3(256), 0, s, noise_nchw, lat, sample, B,
                       (long long)H * W, L, ld, off);
}
// One scheduler step in its affine form (genpercept_amd/scheduler.py: step_coefficients):
//   x0 = clip(x0_sample * s + x0_model * m), eps = eps_sample * s + eps_model * m, s' = prev_x0 * x0 + prev_eps * eps.
__global__ __launch_bounds__(256) void ddim_step_kernel(const h16_t* __restrict__ model, int ldm, float* __restrict__ sample,
                                                         h16_t* __restrict__ uin, int ldu, int off, h16_t* __restrict__ x0_out, int ldx,
                                                         long long pixels, int L, DdimCoef k) {
    const long long n = pixels * L;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % L);
        const long long px = i / L;
        const float m = h16_to_f(model[px * ldm + c]), s = sample[i];
        float x0 = k.x0_sample * s + k.x0_model * m;
        if (k.clip > 0.f) x0 = fminf(fmaxf(x0, -k.clip), k.clip);
        const float eps = k.eps_sample * s + k.eps_model * m;
        const float prev = k.prev_x0 * x0 + k.prev_eps * eps;
        sample[i] = prev;
        uin[px * ldu + off + c] = f_to_h16(prev);
        if (x0_out) x0_out[px * ldx + c] = f_to_h16(x0);
    }
}
void launch_ddim_step(const h16_t* model, int ldm, float* sample, h16_t* uin, int ldu, int off, h16_t* x0_out, int ldx, long long pixels, int L,
                      const DdimCoef& k, hipStream_t s) {
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(pixels * L)), dim3(256), 0, s, model, ldm, sample, uin, ldu, off, x0_out, ldx, pixels, L, k);
}

// bf16 NHWC (row stride ld) -> fp32 NCHW
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const h16_t* __restrict__ in, float* __restrict__ out, int B, int C,
                                                                long long HW, int ld) {
    const long long n = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i % HW, bc = i / HW;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        out[i] = h16_to_f(in[(b * HW + p) * ld + c]);
    }
}
void launch_nhwc_to_nchw_f32(const h16_t* in, float* out, int B, int C, int H, int W, int ld, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, s, in, out, B, C, (long long)H * W, ld);
}

// decoder output NHWC (3 real channels, row stride ld) -> fp32 NCHW [B][1|3][H][W]:
// optional mean over the 3 channels, then (unless raw) clip(-1,1), (x+1)/2   (genpercept_pipeline.py:523-525,470-472)
__global__ __launch_bounds__(256) void decode_epilogue_kernel(const h16_t* __restrict__ in, float* __restrict__ out, int B, long long HW,
                                                               int ld, int mean3, int raw) {
    const long long n = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint2 rw = *(const uint2*)(in + i * ld);
        float c[3] = {h16_lo(rw.x), h16_hi(rw.x), h16_lo(rw.y)};
        const long long b = i / HW, p = i - b * HW;
        if (mean3) {
            float v = (c[0] + c[1] + c[2]) / 3.0f;
            if (!raw) v = (fminf(fmaxf(v, -1.f), 1.f) + 1.f) * 0.5f;
            out[b * HW + p] = v;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = c[k];
                if (!raw) v = (fminf(fmaxf(v, -1.f), 1.f) + 1.f) * 0.5f;
                out[(b * 3 + k) * HW + p] = v;
            }
        }
    }
}
void launch_decode_epilogue(const h16_t* in, float* out, int B, int H, int W, int ld, int mean3, int raw, hipStream_t s) {
    hipLaunchKernelGGL(decode_epilogue_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, s, in, out, B, (long long)H * W, ld, mean3, raw);
}

// out[p][0..ldo) = {in[p][0..C) * scale, 0...}
__global__ __launch_bounds__(256) void scale_pad_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, long long pixels, int C,
                                                         int ldi, int ldo, float scale) {
    const long long n = pixels * ldo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i / ldo;
        const int c = (int)(i - p * ldo);
        out[i] = c < C ? f_to_h16(h16_to_f(in[p * ldi + c]) * scale) : (h16_t)0;
    }
}
void launch_scale_pad(const h16_t* in, h16_t* out, long long pixels, int C, int ldi, int ldo, float scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_pad_kernel, dim3(grid_for(pixels * ldo)), dim3(256), 0, s, in, out, pixels, C, ldi, ldo, scale);
}

// tiny 1x1 conv (Cin, Cout <= 8) on the first Cin channels of a padded NHWC tensor: out = W (in * in_scale) + bias, zero padded
// to ldo channels.  Used for post_quant_conv (4->4) with in_scale = -1/0.18215 (pred_x0 = -v, then /scaling_factor).
__global__ __launch_bounds__(256) void pointwise_small_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, const float* __restrict__ w,
                                                               const float* __restrict__ bias, long long pixels, int Cin, int Cout, int ldi,
                                                               int ldo, float in_scale) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
        float x[8], y[8];
        for (int c = 0; c < Cin; ++c) x[c] = h16_to_f(in[p * ldi + c]) * in_scale;
        for (int o = 0; o < Cout; ++o) {
            float a = bias ? bias[o] : 0.f;
            for (int c = 0; c < Cin; ++c) a += w[o * Cin + c] * x[c];
            y[o] = a;
        }
        for (int o = 0; o < ldo; ++o) out[p * ldo + o] = o < Cout ? f_to_h16(y[o]) : (h16_t)0;
    }
}
void launch_pointwise_small(const h16_t* in, h16_t* out, const float* w, const float* bias, long long pixels, int Cin, int Cout, int ldi,
                            int ldo, float in_scale, hipStream_t s) {
    hipLaunchKernelGGL(pointwise_small_kernel, dim3(grid_for(pixels)), dim3(256), 0, s, in, out, w, bias, pixels, Cin, Cout, ldi, ldo, in_scale);
}

__global__ __launch_bounds__(256) void relu_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        uint4 x = *(const uint4*)(in + i * 8);
        unsigned* w = (unsigned*)&x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // clear negative halves (sign bit set)
            if (w[k] & 0x8000u) w[k] &= 0xffff0000u;
            if (w[k] & 0x80000000u) w[k] &= 0x0000ffffu;
        }
        *(uint4*)(out + i * 8) = x;
    }
}
void launch_relu(const h16_t* in, h16_t* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, in, out, n / 8);
}

__global__ __launch_bounds__(256) void add_kernel(const h16_t* __restrict__ a, const h16_t* __restrict__ b, h16_t* __restrict__ out,
                                                   long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint4 x = *(const uint4*)(a + i * 8), y = *(const uint4*)(b + i * 8);
        const unsigned xw[4] = {x.x, x.y, x.z, x.w}, yw[4] = {y.x, y.y, y.z, y.w};
        uint4 r;
        unsigned* rw = (unsigned*)&r;
#pragma unroll
        for (int k = 0; k < 4; ++k) rw[k] = pack_h16x2(h16_lo(xw[k]) + h16_lo(yw[k]), h16_hi(xw[k]) + h16_hi(yw[k]));
        *(uint4*)(out + i * 8) = r;
    }
}
void launch_add(const h16_t* a, const h16_t* b, h16_t* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, a, b, out, n / 8);
}

// bilinear resize of NHWC bf16, PyTorch semantics (align_corners True: src = dst*(in-1)/(out-1); False: (dst+.5)*in/out-.5, clamped at 0)
__global__ __launch_bounds__(256) void bilinear_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, int B, int Hi, int Wi, int Ho,
                                                        int Wo, int C, int align) {
    const int nvec = C >> 3;
    const long long n = (long long)B * Ho * Wo * nvec;
    const float sy = align ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f) : (float)Hi / (float)Ho;
    const float sx = align ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f) : (float)Wi / (float)Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int v = (int)(i % nvec);
        long long t = i / nvec;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const long long b = t / Ho;
        float fy = align ? oy * sy : fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
        float fx = align ? ox * sx : fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
        const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const h16_t* base = in + b * Hi * Wi * C + v * 8;
        const uint4 a = *(const uint4*)(base + ((long long)y0 * Wi + x0) * C), bq = *(const uint4*)(base + ((long long)y0 * Wi + x1) * C);
        const uint4 c = *(const uint4*)(base + ((long long)y1 * Wi + x0) * C), d = *(const uint4*)(base + ((long long)y1 * Wi + x1) * C);
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w}, cw[4] = {c.x, c.y, c.z, c.w}, dw[4] = {d.x, d.y, d.z, d.w};
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        uint4 r;
        unsigned* rw = (unsigned*)&r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = w00 * h16_lo(aw[k]) + w01 * h16_lo(bw[k]) + w10 * h16_lo(cw[k]) + w11 * h16_lo(dw[k]);
            const float hi = w00 * h16_hi(aw[k]) + w01 * h16_hi(bw[k]) + w10 * h16_hi(cw[k]) + w11 * h16_hi(dw[k]);
            rw[k] = pack_h16x2(lo, hi);
        }
        *(uint4*)(out + i * 8) = r;
    }
}
void launch_bilinear(const h16_t* in, h16_t* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, hipStream_t s) {
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((long long)B * Ho * Wo * (C / 8))), dim3(256), 0, s, in, out, B, Hi, Wi, Ho, Wo, C,
                       align_corners);
}

// DPT head tail: out[b][p] = sum_c w[c] * in[p][c] + bias  (input already ReLU'd by the producing conv), fp32 out
__global__ __launch_bounds__(256) void dpt_final_kernel(const h16_t* __restrict__ in, const float* __restrict__ w, float bias,
                                                         float* __restrict__ out, long long n, int Cin) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a = bias;
        for (int v = 0; v < Cin; v += 8) {
            const uint4 x = *(const uint4*)(in + i * Cin + v);
            const unsigned xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) a += h16_lo(xw[k]) * w[v + 2 * k] + h16_hi(xw[k]) * w[v + 2 * k + 1];
        }
        out[i] = a;
    }
}
void launch_dpt_final(const h16_t* in, const float* w, float bias, float* out, int B, int HW, int Cin, hipStream_t s) {
    const long long n = (long long)B * HW;
    hipLaunchKernelGGL(dpt_final_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, w, bias, out, n, Cin);
}

// per-image min-max normalisation (genpercept_pipeline.py:482, per image: SURVEY.md F10).  Two passes, deterministic.
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ ws) {
    __shared__ float smn[4], smx[4];
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * n;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = xb[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = ws + ((long long)b * gridDim.x + blockIdx.x) * 2;
        o[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        o[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}
__global__ __launch_bounds__(256) void minmax_apply_kernel(float* __restrict__ x, long long n, const float* __restrict__ ws, int nparts) {
    const int b = blockIdx.y;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int k = 0; k < nparts; ++k) { mn = fminf(mn, ws[((long long)b * nparts + k) * 2]); mx = fmaxf(mx, ws[((long long)b * nparts + k) * 2 + 1]); }
    const float den = mx - mn;  // constant maps give NaN exactly like the reference (Appendix B.13)
    float* xb = x + (long long)b * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) xb[i] = (xb[i] - mn) / den;
}
void launch_minmax_norm(float* x, int B, long long n, float* ws, hipStream_t s) {
    const int nparts = 64;
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts, B), dim3(256), 0, s, (const float*)x, n, ws);
    hipLaunchKernelGGL(minmax_apply_kernel, dim3(grid_for(n), B), dim3(256), 0, s, x, n, (const float*)ws, nparts);
}

// ---- VAE encoder conv_in fused with the RGB prologue ------------------------------------------------------------------------------------
// rgb [B][3][H][W] (uint8, or float already in [-1,1]) -> conv3x3(pad 1, 3 -> Cout) -> NHWC bf16, plus GroupNorm partial statistics of the
// output (16x16 tiles, the layout of the halo conv).  Through the generic path this layer cost a 64-channel zero-padded copy of the image
// (302 MB written and read back) and a K = 576 conv of which 27/576 was real work; here K = 27 (padded to 32) is ONE
// v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels, the image is read as uint8 and the layer is bound by its 2 B/element output.
//   workgroup: 4 waves, 16x16 output pixels x 128 channels; halo 18x18x3 normalised to bf16 in LDS ([pixel][4]); k = 3 * tap + c.
//   r4: PERSISTENT -- grid = (B x J, channel slices): a workgroup walks the tiles jw, jw + J, ... of one image with its weights, bias and
//   fragments loaded once, and leaves ONE statistics row (+ its pixel count: the "mode 2" layout of the persistent convs) instead of one per
//   16x16 tile: at 768x768 that is 128 rows per image for the finalize kernel instead of 2304 (its first launch cost 40 us, and the one-tile
//   workgroups spent as long on weights / bias / launch as on their 16 MFMA k-steps).
__global__ __launch_bounds__(256) void rgb_conv_in_kernel(const void* __restrict__ rgb, int is_u8, const h16_t* __restrict__ w27,
                                                           const float* __restrict__ bias, h16_t* __restrict__ out, float* __restrict__ stats,
                                                           int B, int H, int W, int Cout, int J) {
    __shared__ h16_t s_h[18 * 18 * 4];
    __shared__ float s_red[4 * 128 * 2];
    __shared__ __attribute__((aligned(16))) h16_t s_w[128 * 32];  // this slice's weights, [channel][k = 3 tap + c, zero for k >= 27]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = lane & 15, q = lane >> 4;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4, tiles_sp = tiles_x * tiles_y;
    const int b = blockIdx.x / J, jw = blockIdx.x - b * J;
    const int n0 = blockIdx.y * 128;
    const int npair = min(4, (Cout - n0) >> 5);  // 32-channel pairs of fragments handled here
    const long long HW = (long long)H * W;

    // weights: compact [Cout][32] matrix (pack_k27_kernel) -> LDS (two 16-byte pieces per thread), then 16 bytes per fragment and lane.
    // (Gathering the 27 taps per lane from the [rows][9][64] conv layout cost more than the whole rest of the kernel: ~32 cache lines
    //  per load instruction, 64 instructions per lane and workgroup.)
    for (int i = tid; i < 128 * 32 / 8; i += 256) {
        const int ch = i >> 2;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n0 + ch < Cout) v = *(const uint4*)(w27 + (long long)(n0 + ch) * 32 + (i & 3) * 8);
        *(uint4*)(s_w + i * 8) = v;
    }
    float bv[4][8];
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[ip][e] = (bias && ip < npair) ? bias[n0 + 32 * ip + 8 * q + e] : 0.f;
    __syncthreads();
    // weight fragments: MFMA row a of fragment i (pair ip = i / 2) is output channel 32 ip + 8 (a / 4) + 4 (i & 1) + (a & 3), so that a
    // lane ends up with 8 consecutive channels of its pixel; this lane's k = 8 q .. 8 q + 7
    h16x8_t wf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wf[i] = *(const h16x8_t*)(s_w + (32 * (i >> 1) + 8 * (a >> 2) + 4 * (i & 1) + (a & 3)) * 32 + 8 * q);

    float st_s[4][8], st_q[4][8];
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[ip][e] = st_q[ip][e] = 0.f;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    int run_px = 0;
    for (int sp = jw; sp < tiles_sp; sp += J) {
        const int ty = sp / tiles_x, tx = sp - ty * tiles_x;
        run_px += min(16, H - 16 * ty) * min(16, W - 16 * tx);
        __syncthreads();  // (the previous tile's fragment gathers are over)
        // halo: (pixel, channel) items, zero outside the image (the reference pads the NORMALISED image)
        for (int i = tid; i < 18 * 18 * 4; i += 256) {
            const int c = i & 3, pix = i >> 2;
            const int hy = pix / 18, hx = pix - hy * 18;
            const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
            float v = 0.f;
            if (c < 3 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const long long src = ((long long)b * 3 + c) * HW + (long long)iy * W + ix;
                v = is_u8 ? ((float)((const unsigned char*)rgb)[src] / 255.0f * 2.0f - 1.0f) : ((const float*)rgb)[src];
            }
            s_h[i] = f_to_h16(v);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = 4 * wave + j;
            h16x8_t xf;  // B operand: pixel a of row py, k = 8 q + e
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * q + e;
                const int tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
                xf[e] = k < 27 ? (short)s_h[((py + ky) * 18 + a + kx) * 4 + c] : (short)0;
            }
            const int oy = ty * 16 + py, ox = tx * 16 + a;
            const bool ok = oy < H && ox < W;
            h16_t* o = out + (((long long)b * H + oy) * W + ox) * Cout + n0 + 8 * q;
#pragma unroll
            for (int ip = 0; ip < 4; ++ip) {
                if (ip >= npair) break;
                const f32x4_t lo = mfma_16x16x32(wf[2 * ip], xf, zero4);
                const f32x4_t hi = mfma_16x16x32(wf[2 * ip + 1], xf, zero4);
                const float v[8] = {lo.x + bv[ip][0], lo.y + bv[ip][1], lo.z + bv[ip][2], lo.w + bv[ip][3],
                                    hi.x + bv[ip][4], hi.y + bv[ip][5], hi.z + bv[ip][6], hi.w + bv[ip][7]};
                uint4 pk;
                pk.x = pack_h16x2(v[0], v[1]); pk.y = pack_h16x2(v[2], v[3]); pk.z = pack_h16x2(v[4], v[5]); pk.w = pack_h16x2(v[6], v[7]);
                if (ok) {
                    *(uint4*)(o + 32 * ip) = pk;
                    const float r[8] = {h16_lo(pk.x), h16_hi(pk.x), h16_lo(pk.y), h16_hi(pk.y), h16_lo(pk.z), h16_hi(pk.z), h16_lo(pk.w), h16_hi(pk.w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { st_s[ip][e] += r[e]; st_q[ip][e] += r[e] * r[e]; }
                }
            }
        }
    }
    if (stats) {  // per (workgroup, channel) sums of the stored values: 16 pixel lanes -> lane a == 0, 4 waves -> LDS -> one thread per channel
#pragma unroll
        for (int ip = 0; ip < 4; ++ip)
#pragma unroll
            for (int e = 0; e < 8; ++e) { st_s[ip][e] = row16_sum(st_s[ip][e]); st_q[ip][e] = row16_sum(st_q[ip][e]); }
        if (a == 0) {
#pragma unroll
            for (int ip = 0; ip < 4; ++ip)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_red[(wave * 128 + 32 * ip + 8 * q + e) * 2] = st_s[ip][e];
                    s_red[(wave * 128 + 32 * ip + 8 * q + e) * 2 + 1] = st_q[ip][e];
                }
        }
        __syncthreads();
        const long long row = blockIdx.x;  // = b * J + jw: J rows per image
        if (tid < 128 && n0 + tid < Cout) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { ss += s_red[(w * 128 + tid) * 2]; qq += s_red[(w * 128 + tid) * 2 + 1]; }
            float* so = stats + (row * Cout + n0 + tid) * 2;
            so[0] = ss;
            so[1] = qq;
        }
        if (tid == 0 && blockIdx.y == 0) stats[(long long)B * J * Cout * 2 + row] = (float)run_px;
    }
}
// compact conv_in weights: [Cout][32] bf16 with k = 3 * tap + c (c < 3), zero for k >= 27, from the packed conv layout [rows][9][64]
__global__ __launch_bounds__(256) void pack_k27_kernel(const h16_t* __restrict__ wt, int ldw, int Cout, h16_t* __restrict__ w27) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * 32) return;
    const int ch = i >> 5, k = i & 31;
    w27[i] = k < 27 ? wt[(long long)ch * ldw + (k / 3) * 64 + (k % 3)] : (h16_t)0;
}
void launch_pack_k27(const h16_t* wt, int ldw, int Cout, h16_t* w27, hipStream_t s) {
    hipLaunchKernelGGL(pack_k27_kernel, dim3((Cout * 32 + 255) / 256), dim3(256), 0, s, wt, ldw, Cout, w27);
}
// workgroups (= statistics rows) per image of the persistent conv_in: ~2 per CU over the batch, at most one per tile
int rgb_conv_in_rows(int B, int H, int W) {
    static int ncu = 0;  // (queried once: hipGetDeviceProperties is not a launch-path call)
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    const int tiles_sp = ((W + 15) / 16) * ((H + 15) / 16);
    int J = (2 * ncu + B - 1) / B;
    if (J > tiles_sp) J = tiles_sp;
    return J < 1 ? 1 : J;
}
// Cout % 32 == 0; w27 = launch_pack_k27 output; stats (optional): [B * J][Cout][2] sums + [B * J] pixel counts, J = rgb_conv_in_rows() ("mode 2")
void launch_rgb_conv_in(const void* rgb, int is_u8, const h16_t* w27, const float* bias, h16_t* out, float* stats, int B, int H, int W, int Cout,
                        hipStream_t s) {
    const int J = rgb_conv_in_rows(B, H, W);
    hipLaunchKernelGGL(rgb_conv_in_kernel, dim3(B * J, (Cout + 127) / 128), dim3(256), 0, s, rgb, is_u8, w27, bias, out, stats, B, H, W, Cout, J);
}

GP_SAT_TU(elementwise)  // fp16 build: address of this translation unit's saturation flag (common.h)
