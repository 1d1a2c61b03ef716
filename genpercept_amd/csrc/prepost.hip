// Pre / post processing of GenPerceptPipeline.__call__ on the device: once the model runs at ~13 ms per image, the
// host-side torchvision resize and matplotlib colour map of the reference dominate run.py's wall clock.
//   resize      torchvision.transforms.functional.resize(img, size, BILINEAR | NEAREST_EXACT, antialias=True) on NCHW tensors
//               (genpercept/util/image_util.py:104, genpercept_pipeline.py:301-307): ATen's separable anti-aliased triangle filter --
//               per output index i: scale = in / out, support = max(scale, 1), centre = scale (i + 0.5), taps xmin .. xmin + xsize - 1 with
//               weights max(0, 1 - |(x - centre + 0.5) / max(scale, 1)|) normalised to sum 1 -- W pass first, then H, fp32 throughout.
//               uint8 images are interpolated in fp32 and rounded half-to-even back to uint8 (torchvision v1's _cast_squeeze_in / _out:
//               what the reference does to the RGB input before normalising it).
//   colorize    matplotlib colour map as a 256-entry LUT: index = int(x * 256) clipped to 255 (Colormap.__call__ on floats), bytes =
//               (lut * 255).astype(uint8) (image_util.py:25-63 + genpercept_pipeline.py:318-325), written HWC.
//   quantise    (pred * 65535.0).astype(uint16) / (pred * 255.0).astype(uint8): fp32 product, truncation (run.py:449-455).
// All HBM-bound, one thread per output element.
#include "common.h"
#include "kernels.h"

namespace {

struct AaTaps { int xmin, xsize; float center, invscale, total; };

// ATen HelperInterpBase::_compute_index_ranges_weights for the anti-aliased triangle (bilinear) filter, align_corners = false
GP_DEV AaTaps aa_taps(int i, int in_size, float scale) {
    AaTaps t;
    const float support = scale >= 1.f ? scale : 1.f;  // (interp_size / 2) * scale with interp_size = 2
    t.center = scale * ((float)i + 0.5f);
    t.invscale = scale >= 1.f ? 1.f / scale : 1.f;
    int lo = (int)(t.center - support + 0.5f);
    if (lo < 0) lo = 0;
    int hi = (int)(t.center + support + 0.5f);
    if (hi > in_size) hi = in_size;
    t.xmin = lo;
    t.xsize = hi - lo;
    float tot = 0.f;
    for (int j = 0; j < t.xsize; ++j) {
        const float x = ((float)(j + lo) - t.center + 0.5f) * t.invscale;
        tot += fmaxf(0.f, 1.f - fabsf(x));
    }
    t.total = tot;
    return t;
}
GP_DEV float aa_weight(const AaTaps& t, int j) {
    const float x = ((float)(j + t.xmin) - t.center + 0.5f) * t.invscale;
    const float w = fmaxf(0.f, 1.f - fabsf(x));
    return t.total != 0.f ? w / t.total : w;
}

template <typename TIN>
GP_DEV float ld(const TIN* p, long long i);
template <>
GP_DEV float ld<unsigned char>(const unsigned char* p, long long i) { return (float)p[i]; }
template <>
GP_DEV float ld<float>(const float* p, long long i) { return p[i]; }

// pass 1: along W.  in [planes][H][Wi] -> tmp [planes][H][Wo] fp32
template <typename TIN>
__global__ __launch_bounds__(256) void resize_aa_w_kernel(const TIN* __restrict__ in, float* __restrict__ tmp, long long planes_h, int Wi, int Wo,
                                                           float scale) {
    const long long n = planes_h * Wo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const long long row = idx / Wo;
        const int ox = (int)(idx - row * Wo);
        const AaTaps t = aa_taps(ox, Wi, scale);
        const TIN* src = in + row * Wi + t.xmin;
        float acc = ld<TIN>(src, 0) * aa_weight(t, 0);
        for (int j = 1; j < t.xsize; ++j) acc += ld<TIN>(src, j) * aa_weight(t, j);
        tmp[idx] = acc;
    }
}
// pass 2: along H.  tmp [planes][Hi][Wo] -> out [planes][Ho][Wo]; OUT uint8: round half to even + clamp; CLIP01: clip to [0, 1]
template <typename TOUT, bool CLIP01>
__global__ __launch_bounds__(256) void resize_aa_h_kernel(const float* __restrict__ tmp, TOUT* __restrict__ out, long long planes, int Hi, int Ho, int Wo,
                                                           float scale) {
    const long long n = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int ox = (int)(idx % Wo);
        const long long r = idx / Wo;
        const int oy = (int)(r % Ho);
        const long long pl = r / Ho;
        const AaTaps t = aa_taps(oy, Hi, scale);
        const float* src = tmp + (pl * Hi + t.xmin) * Wo + ox;
        float acc = src[0] * aa_weight(t, 0);
        for (int j = 1; j < t.xsize; ++j) acc += src[(long long)j * Wo] * aa_weight(t, j);
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

// in / out: NCHW with `planes` = B * C images of Hi x Wi; tmp: planes * Hi * Wo floats (bilinear only).  mode 0: anti-aliased bilinear,
// 1: nearest-exact.  u8 = 1: uint8 in and out; 0: fp32 in and out (clip01 optionally clips the result to [0, 1]).
void launch_resize(const void* in, void* out, float* tmp, long long planes, int Hi, int Wi, int Ho, int Wo, int mode, int u8, int clip01, hipStream_t s) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;  // area_pixel_compute_scale, align_corners = false
    const long long n_out = planes * Ho * Wo;
    if (mode == 1) {
        if (u8) hipLaunchKernelGGL((resize_nearest_exact_kernel<unsigned char, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const unsigned char*)in,
                                   (unsigned char*)out, planes, Hi, Wi, Ho, Wo, sy, sx);
        else if (clip01) hipLaunchKernelGGL((resize_nearest_exact_kernel<float, true>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)in, (float*)out,
                                            planes, Hi, Wi, Ho, Wo, sy, sx);
        else hipLaunchKernelGGL((resize_nearest_exact_kernel<float, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)in, (float*)out, planes, Hi,
                                Wi, Ho, Wo, sy, sx);
        return;
    }
    const long long n_tmp = planes * Hi * Wo;
    if (u8) {
        hipLaunchKernelGGL((resize_aa_w_kernel<unsigned char>), dim3(grid_of(n_tmp)), dim3(256), 0, s, (const unsigned char*)in, tmp, planes * Hi, Wi, Wo, sx);
        hipLaunchKernelGGL((resize_aa_h_kernel<unsigned char, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (unsigned char*)out, planes, Hi,
                           Ho, Wo, sy);
    } else {
        hipLaunchKernelGGL((resize_aa_w_kernel<float>), dim3(grid_of(n_tmp)), dim3(256), 0, s, (const float*)in, tmp, planes * Hi, Wi, Wo, sx);
        if (clip01) hipLaunchKernelGGL((resize_aa_h_kernel<float, true>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (float*)out, planes, Hi, Ho,
                                       Wo, sy);
        else hipLaunchKernelGGL((resize_aa_h_kernel<float, false>), dim3(grid_of(n_out)), dim3(256), 0, s, (const float*)tmp, (float*)out, planes, Hi, Ho, Wo, sy);
    }
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
