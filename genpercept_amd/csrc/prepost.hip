// Pre / post processing of GenPerceptPipeline.__call__ on the device: once the model runs at ~13 ms per image, the
// host-side torchvision resize and matplotlib colour map of the reference dominate run.py's wall clock.
//   resize      torchvision.transforms.functional.resize(img, size, BILINEAR | BICUBIC | NEAREST_EXACT, antialias=True) on NCHW tensors
//               (genpercept/util/image_util.py:104, genpercept_pipeline.py:301-307): ATen's separable anti-aliased triangle filter --
//               per output index i: scale = in / out, support = max(scale, 1), centre = scale (i + 0.5), taps xmin .. xmin + xsize - 1 with
//               weights max(0, 1 - |(x - centre + 0.5) / max(scale, 1)|) normalised to sum 1 -- W pass first, then H, fp32 throughout.
//               uint8 images are interpolated in fp32 and rounded half-to-even back to uint8 (torchvision v1's _cast_squeeze_in / _out:
//               what the reference does to the RGB input before normalising it).
//               BICUBIC (image_util.py:108-126 "bicubic"; r4): the same index ranges with interp_size = 4 (support = 2 max(scale, 1)) and the
//               Keys cubic with a = -0.5 as the filter (ATen HelperInterpCubic::aa_filter); fp32 results overshoot, uint8 ones are clamped.
//               Float images (the trainer-style `rgb_int`, genpercept_trainer.py:1151-1165) take the fp32 kernels: no rounding, no clamp.
//   colorize    matplotlib colour map as a 256-entry LUT: index = int(x * 256) clipped to 255 (Colormap.__call__ on floats), bytes =
//               (lut * 255).astype(uint8) (image_util.py:25-63 + genpercept_pipeline.py:318-325), written HWC.
//   quantise    (pred * 65535.0).astype(uint16) / (pred * 255.0).astype(uint8): fp32 product, truncation (run.py:449-455).
// All HBM-bound, one thread per output element.
#include "common.h"
#include "kernels.h"

namespace {

struct AaTaps { int xmin, xsize; float center, invscale, total; };

// filter F: 0 = triangle (bilinear), 1 = Keys cubic a = -0.5 (bicubic); written without contraction: the host computes these weights in scalar fp32
template <int F>
GP_DEV float aa_filter(float x) {
    x = fabsf(x);
    if (F == 0) return fmaxf(0.f, 1.f - x);
    if (x < 1.f) return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(1.5f, x), 2.5f), x), x), 1.f);                 // ((a + 2) x - (a + 3)) x x + 1
    if (x < 2.f) return __fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(-0.5f, x), 2.5f), x), 4.f), x), 2.f);  // ((a x - 5 a) x + 8 a) x - 4 a
    return 0.f;
}
// ATen HelperInterpBase::_compute_index_ranges_weights for the anti-aliased filters, align_corners = false
template <int F>
GP_DEV AaTaps aa_taps(int i, int in_size, float scale) {
    AaTaps t;
    constexpr float HALF = F == 0 ? 1.f : 2.f;  // interp_size / 2
    const float support = scale >= 1.f ? HALF * scale : HALF;
    t.center = scale * ((float)i + 0.5f);
    t.invscale = scale >= 1.f ? 1.f / scale : 1.f;
    int lo = (int)(t.center - support + 0.5f);
    if (lo < 0) lo = 0;
    int hi = (int)(t.center + support + 0.5f);
    if (hi > in_size) hi = in_size;
    t.xmin = lo;
    t.xsize = hi - lo;
    float tot = 0.f;
    for (int j = 0; j < t.xsize; ++j) tot += aa_filter<F>(((float)(j + lo) - t.center + 0.5f) * t.invscale);
    t.total = tot;
    return t;
}
template <int F>
GP_DEV float aa_weight(const AaTaps& t, int j) {
    const float w = aa_filter<F>(((float)(j + t.xmin) - t.center + 0.5f) * t.invscale);
    return t.total != 0.f ? w / t.total : w;
}

template <typename TIN>
GP_DEV float ld(const TIN* p, long long i);
template <>
GP_DEV float ld<unsigned char>(const unsigned char* p, long long i) { return (float)p[i]; }
template <>
GP_DEV float ld<float>(const float* p, long long i) { return p[i]; }

// pass 1: along W.  in [planes][H][Wi] -> tmp [planes][H][Wo] fp32
template <typename TIN, int F>
__global__ __launch_bounds__(256) void resize_aa_w_kernel(const TIN* __restrict__ in, float* __restrict__ tmp, long long planes_h, int Wi, int Wo,
                                                           float scale) {
    const long long n = planes_h * Wo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const long long row = idx / Wo;
        const int ox = (int)(idx - row * Wo);
        const AaTaps t = aa_taps<F>(ox, Wi, scale);
        const TIN* src = in + row * Wi + t.xmin;
        float acc = ld<TIN>(src, 0) * aa_weight<F>(t, 0);
        for (int j = 1; j < t.xsize; ++j) acc += ld<TIN>(src, j) * aa_weight<F>(t, j);
        tmp[idx] = acc;
    }
}
// pass 2: along H.  tmp [planes][Hi][Wo] -> out [planes][Ho][Wo]; OUT uint8: round half to even + clamp; CLIP01: clip to [0, 1]
template <typename TOUT, bool CLIP01, int F>
__global__ __launch_bounds__(256) void resize_aa_h_kernel(const float* __restrict__ tmp, TOUT* __restrict__ out, long long planes, int Hi, int Ho, int Wo,
                                                           float scale) {
    const long long n = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int ox = (int)(idx % Wo);
        const long long r = idx / Wo;
        const int oy = (int)(r % Ho);
        const long long pl = r / Ho;
        const AaTaps t = aa_taps<F>(oy, Hi, scale);
        const float* src = tmp + (pl * Hi + t.xmin) * Wo + ox;
        float acc = src[0] * aa_weight<F>(t, 0);
        for (int j = 1; j < t.xsize; ++j) acc += src[(long long)j * Wo] * aa_weight<F>(t, j);
        if constexpr (sizeof(TOUT) == 1) {
            out[idx] = (TOUT)fminf(fmaxf(rintf(acc), 0.f), 255.f);  // torch.round: half to even
        } else {
            out[idx] = (TOUT)(CLIP01 ? fminf(fmaxf(acc, 0.f), 1.f) : acc);
        }
    }
}
// NEAREST_EXACT: src = min(floor((dst + 0.5) * in / out), in - 1)
template <typename T, bool CLIP01>
__global__ __launch_bounds__(256) void resize_nearest_exact_kernel(const T* __restrict__ in, T* __restrict__ out, long long planes, int Hi, int Wi, int Ho,
                                                                    int Wo, float sy, float sx) {
    const long long n = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int ox = (int)(idx % Wo);
        const long long r = idx / Wo;
        const int oy = (int)(r % Ho);
        const long long pl = r / Ho;
        const int iy = min((int)floorf(((float)oy + 0.5f) * sy), Hi - 1), ix = min((int)floorf(((float)ox + 0.5f) * sx), Wi - 1);
        T v = in[(pl * Hi + iy) * Wi + ix];
        if constexpr (CLIP01 && sizeof(T) == 4) v = fminf(fmaxf(v, 0.f), 1.f);
        out[idx] = v;
    }
}
__global__ __launch_bounds__(256) void normalize_rgb_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = __fsub_rn(__fmul_rn(__fdiv_rn(in[i], 255.0f), 2.0f), 1.0f);
}
__global__ __launch_bounds__(256) void clip01_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = fminf(fmaxf(in[i], 0.f), 1.f);
}
// x [B][H][W] in [0, 1] -> rgb [B][H][W][3] uint8 through a 256 x 3 byte LUT
__global__ __launch_bounds__(256) void colorize_lut_kernel(const float* __restrict__ x, const unsigned char* __restrict__ lut, unsigned char* __restrict__ rgb,
                                                            long long n) {
    __shared__ unsigned char s_lut[768];
    for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = fminf(fmaxf(x[i], 0.f), 1.f) * 256.f;  // matplotlib: xa *= N; xa[xa == N] = N - 1; astype(int)
        int k = (int)v;
        if (k > 255) k = 255;
        rgb[i * 3] = s_lut[k * 3];
        rgb[i * 3 + 1] = s_lut[k * 3 + 1];
        rgb[i * 3 + 2] = s_lut[k * 3 + 2];
    }
}
template <typename TOUT>
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, TOUT* __restrict__ q, long long n, float mul) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) q[i] = (TOUT)(x[i] * mul);  // truncation, like astype
}

unsigned grid_of(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// in / out: NCHW with `planes` = B * C images of Hi x Wi; tmp: planes * Hi * Wo floats (the separable filters only).  mode 0: anti-aliased bilinear,
// 1: nearest-exact, 2: anti-aliased bicubic.  u8 = 1: uint8 in and out; 0: fp32 in and out (clip01 optionally clips the result to [0, 1]).
template <int F>
static void launch_resize_aa(const void* in, void* out, float* tmp, long long planes, int Hi, int Wi, int Ho, int Wo, int u8, int clip01, hipStream_t s) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;  // area_pixel_compute_scale, align_corners = false
    const long long n_out = planes * Ho * Wo, n_tmp = planes * Hi * Wo;
    if (u8) {
        hipLaunchKernelGGL((resize_aa_w_kernel<unsigned char, F>), dim3(grid_of(n_tmp)), dim3(256), 0, s, (const unsigned char*)in, tmp, planes * Hi, Wi, Wo, sx);
        hipLaunchKernelGGL((resize_aa_h_kernel<unsigned char, false, F>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (unsigned char*)out, planes,
                           Hi, Ho, Wo, sy);
    } else {
        hipLaunchKernelGGL((resize_aa_w_kernel<float, F>), dim3(grid_of(n_tmp)), dim3(256), 0, s, (const float*)in, tmp, planes * Hi, Wi, Wo, sx);
        if (clip01) hipLaunchKernelGGL((resize_aa_h_kernel<float, true, F>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (float*)out, planes, Hi,
                                       Ho, Wo, sy);
        else hipLaunchKernelGGL((resize_aa_h_kernel<float, false, F>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (float*)out, planes, Hi, Ho, Wo,
                                sy);
    }
}
void launch_resize(const void* in, void* out, float* tmp, long long planes, int Hi, int Wi, int Ho, int Wo, int mode, int u8, int clip01, hipStream_t s) {
    if (mode == 1) {
        const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
        const long long n_out = planes * Ho * Wo;
        if (u8) hipLaunchKernelGGL((resize_nearest_exact_kernel<unsigned char, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const unsigned char*)in,
                                   (unsigned char*)out, planes, Hi, Wi, Ho, Wo, sy, sx);
        else if (clip01) hipLaunchKernelGGL((resize_nearest_exact_kernel<float, true>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)in, (float*)out,
                                            planes, Hi, Wi, Ho, Wo, sy, sx);
        else hipLaunchKernelGGL((resize_nearest_exact_kernel<float, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)in, (float*)out, planes, Hi,
                                Wi, Ho, Wo, sy, sx);
        return;
    }
    if (mode == 2) launch_resize_aa<1>(in, out, tmp, planes, Hi, Wi, Ho, Wo, u8, clip01, s);
    else launch_resize_aa<0>(in, out, tmp, planes, Hi, Wi, Ho, Wo, u8, clip01, s);
}
// rgb / 255 * 2 - 1 on an fp32 image (genpercept_pipeline.py:245: the float-tensor input's normalisation, same operation order), in place allowed
void launch_normalize_rgb(const float* in, float* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(normalize_rgb_kernel, dim3(grid_of(n)), dim3(256), 0, s, in, out, n);
}
void launch_clip01(const float* in, float* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(clip01_kernel, dim3(grid_of(n)), dim3(256), 0, s, in, out, n);
}
void launch_colorize_lut(const float* x, const unsigned char* lut, unsigned char* rgb, long long n, hipStream_t s) {
    hipLaunchKernelGGL(colorize_lut_kernel, dim3(grid_of(n)), dim3(256), 0, s, x, lut, rgb, n);
}
void launch_quantize(const float* x, void* q, long long n, int bits, hipStream_t s) {
    if (bits == 16) hipLaunchKernelGGL((quantize_kernel<unsigned short>), dim3(grid_of(n)), dim3(256), 0, s, x, (unsigned short*)q, n, 65535.0f);
    else hipLaunchKernelGGL((quantize_kernel<unsigned char>), dim3(grid_of(n)), dim3(256), 0, s, x, (unsigned char*)q, n, 255.0f);
}
