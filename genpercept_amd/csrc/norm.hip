// GroupNorm(+SiLU) and LayerNorm for NHWC bf16 activations (K7/K8 of SURVEY.md §2.3).  HBM-bound: every kernel moves
// 16 bytes per lane per access, statistics in fp32, deterministic (no atomics).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

// pixel chunks per image = workgroups per image of the statistics / apply passes: at least 16 pixels each, at most 256 chunks.
// (HW / 256 left the UNet's 12x12 and 24x24 maps with ONE or two workgroups per image -- 4..8 on the whole GPU, 50 us for 1.5 MB.)
static inline int gn_nchunk(int HW) {
    int n = HW / 16;
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    return n;
}
int groupnorm_ws_floats(int B, int HW, int C, int G) { return B * gn_nchunk(HW) * G * 2; }

// ---- pass 1: per (image, pixel-chunk, group) partial sum / sum of squares -----------------------------------------
// Thread t owns VPT fixed 8-channel vectors (v = tv, tv + tpp, ...) and walks pixels p = pl, pl + PL, ... of the chunk.
template <int VPT>
__global__ __launch_bounds__(256) void gn_stats_kernel(const h16_t* __restrict__ x, float* __restrict__ ws, int HW, int C, int G,
                                                        int nchunk, int tpp, int PL) {
    extern __shared__ __attribute__((aligned(16))) float sred[];  // [PL][C][2]
    const int b = blockIdx.y, ck = blockIdx.x;
    const int nvec = C >> 3;
    const int tid = threadIdx.x;
    const int pl = tid / tpp, tv = tid - pl * tpp;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = ck * per, p1 = min(HW, p0 + per);
    float s[VPT][8], q[VPT][8];
#pragma unroll
    for (int u = 0; u < VPT; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[u][e] = q[u][e] = 0.f;
    if (pl < PL) {
        const h16_t* xb = x + (long long)b * HW * C;
        for (int p = p0 + pl; p < p1; p += PL) {
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const int v = tv + u * tpp;
                if (v < nvec) {
                    const uint4 raw = *(const uint4*)(xb + (long long)p * C + v * 8);
                    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = h16_lo(w[e]), hi = h16_hi(w[e]);
                        s[u][2 * e] += lo; q[u][2 * e] += lo * lo;
                        s[u][2 * e + 1] += hi; q[u][2 * e + 1] += hi * hi;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int v = tv + u * tpp;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sred[((pl * C) + v * 8 + e) * 2 + 0] = s[u][e];
                    sred[((pl * C) + v * 8 + e) * 2 + 1] = q[u][e];
                }
            }
        }
    }
    __syncthreads();
    if (tid < G) {
        const int cpg = C / G;
        float ss = 0.f, qq = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
            for (int l = 0; l < PL; ++l) { ss += sred[(l * C + c) * 2]; qq += sred[(l * C + c) * 2 + 1]; }
        float* o = ws + (((long long)b * nchunk + ck) * G + tid) * 2;
        o[0] = ss;
        o[1] = qq;
    }
}

// ---- pass 2: combine partials (Chan) and fold gamma/beta into per-(image, channel) scale/shift -----------------------------
//   y = x * scale[b][c] + shift[b][c],  scale = rstd_g * gamma_c,  shift = beta_c - mean_g * scale
// The pair is consumed either by gn_apply_kernel or, fused, by the conv kernel that reads the tensor (conv_halo.hip).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                           int HW, int C, int G, int nchunk, float eps) {
    // 8 lanes per group (G <= 32) walk the chunk partials in a fixed order and combine by xor-shuffles: deterministic, and
    // ~30x shorter than one thread per group looping over up to 256 chunks (that version cost 3.5 ms per forward pass).
    __shared__ float s_stat[2 * 32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    const int per = (HW + nchunk - 1) / nchunk;
    const int g = tid >> 3, sub = tid & 7;
    const bool act = g < G;
    const float* w = ws + ((long long)b * nchunk * G + (act ? g : 0)) * 2;
    float tot = 0.f;
    if (act)
        for (int k = sub; k < nchunk; k += 8) tot += w[(long long)k * G * 2];
    tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2); tot += __shfl_xor(tot, 4);
    const float n_all = (float)HW * (float)cpg;
    const float mean = tot / n_all;
    float m2 = 0.f;
    if (act)
        for (int k = sub; k < nchunk; k += 8) {
            const int cnt_px = min(HW, (k + 1) * per) - min(HW, k * per);
            if (cnt_px <= 0) continue;
            const float nk = (float)cnt_px * (float)cpg;
            const float sk = w[(long long)k * G * 2], qk = w[(long long)k * G * 2 + 1];
            const float mk = sk / nk;
            m2 += fmaxf(qk - sk * mk, 0.f) + nk * (mk - mean) * (mk - mean);
        }
    m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
    if (act && sub == 0) {
        s_stat[2 * g] = mean;
        s_stat[2 * g + 1] = rsqrtf(m2 / n_all + eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int gg = c / cpg;
        const float sc = s_stat[2 * gg + 1] * gamma[c];
        scale[(long long)b * C + c] = sc;
        shift[(long long)b * C + c] = beta[c] - s_stat[2 * gg] * sc;
    }
}

// ---- pass 3 (only when the consumer cannot fuse it): y = act(x * scale + shift) ----------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int HW, int C, int nchunk, int silu) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [C] scale, [C] shift
    float* s_scale = sm;
    float* s_shift = sm + C;
    const int b = blockIdx.y, ck = blockIdx.x, tid = threadIdx.x;
    const int per = (HW + nchunk - 1) / nchunk;
    for (int c = tid; c < C; c += 256) {
        s_scale[c] = scale[(long long)b * C + c];
        s_shift[c] = shift[(long long)b * C + c];
    }
    __syncthreads();
    const int nvec = C >> 3;
    const int p0 = ck * per, p1 = min(HW, p0 + per);
    const long long base = (long long)b * HW * C;
    const long long e0 = (long long)p0 * nvec, e1 = (long long)p1 * nvec;
    for (long long e = e0 + tid; e < e1; e += 256) {
        const int v = (int)(e % nvec);
        const uint4 raw = *(const uint4*)(x + base + e * 8);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = v * 8 + 2 * k;
            o[2 * k] = h16_lo(w[k]) * s_scale[c] + s_shift[c];
            o[2 * k + 1] = h16_hi(w[k]) * s_scale[c + 1] + s_shift[c + 1];
        }
        if (silu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = silu_f(o[k]);
        }
        uint4 r;
        r.x = pack_h16x2(o[0], o[1]); r.y = pack_h16x2(o[2], o[3]); r.z = pack_h16x2(o[4], o[5]); r.w = pack_h16x2(o[6], o[7]);
        *(uint4*)(y + base + e * 8) = r;
    }
}

// ---- small maps: statistics + apply in ONE launch ---------------------------------------------------------------------------------------
// One workgroup per (group, image) owns the group's HW x cpg block (cpg % 8 == 0): read it for the statistics (exact two-pass: sum, then
// squared deviations), read it again -- L2-hot -- to normalise and write.  The three-launch path (partials, finalize, apply) cost ~27 us on
// the UNet's 12x12 maps, 17 times per pass, for 1.5 MB of data.
__global__ __launch_bounds__(256) void gn_small_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int HW, int C, int G, float eps, int silu) {
    __shared__ float red[8];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, vpg = cpg >> 3, items = HW * vpg;
    const h16_t* xb = x + (long long)b * HW * C + g * cpg;
    h16_t* yb = y + (long long)b * HW * C + g * cpg;
    auto block_sum = [&](float v, int slot) -> float {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[slot * 4 + (tid >> 6)] = v;
        __syncthreads();
        return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
    };
    float s = 0.f;
    for (int i = tid; i < items; i += 256) {
        const int p = i / vpg, v = i - p * vpg;
        const uint4 raw = *(const uint4*)(xb + (long long)p * C + v * 8);
        s += (h16_lo(raw.x) + h16_hi(raw.x)) + (h16_lo(raw.y) + h16_hi(raw.y)) + (h16_lo(raw.z) + h16_hi(raw.z)) + (h16_lo(raw.w) + h16_hi(raw.w));
    }
    const float n_all = (float)HW * (float)cpg;
    const float mean = block_sum(s, 0) / n_all;
    float q = 0.f;
    for (int i = tid; i < items; i += 256) {
        const int p = i / vpg, v = i - p * vpg;
        const uint4 raw = *(const uint4*)(xb + (long long)p * C + v * 8);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float a = h16_lo(w[k]) - mean, c = h16_hi(w[k]) - mean; q += a * a + c * c; }
    }
    const float rstd = rsqrtf(block_sum(q, 1) / n_all + eps);
    for (int i = tid; i < items; i += 256) {
        const int p = i / vpg, v = i - p * vpg;
        const uint4 raw = *(const uint4*)(xb + (long long)p * C + v * 8);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        const float* gm = gamma + g * cpg + v * 8;
        const float* bt = beta + g * cpg + v * 8;
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = (h16_lo(w[k]) - mean) * rstd * gm[2 * k] + bt[2 * k];
            o[2 * k + 1] = (h16_hi(w[k]) - mean) * rstd * gm[2 * k + 1] + bt[2 * k + 1];
        }
        if (silu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = silu_f(o[k]);
        }
        uint4 r;
        r.x = pack_h16x2(o[0], o[1]); r.y = pack_h16x2(o[2], o[3]); r.z = pack_h16x2(o[4], o[5]); r.w = pack_h16x2(o[6], o[7]);
        *(uint4*)(yb + (long long)p * C + v * 8) = r;
    }
}
// r5: the same kernel with the workgroup's slice held in registers between its three passes (statistics mean, variance, apply): ONE read of the input
// instead of three dependent trips to L2 -- these launches are latency-bound (128 workgroups of 46 KiB each on the 24 x 24 x 1280 maps: 17-19 us for
// 6 MB).  Same per-thread summation order and the same block reduction as gn_small_kernel: bit-identical results.  items <= 256 * MAXIT.
template <int MAXIT>
__global__ __launch_bounds__(256) void gn_small_reg_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int HW, int C, int G, float eps, int silu) {
    __shared__ float red[8];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, vpg = cpg >> 3, items = HW * vpg;
    const h16_t* xb = x + (long long)b * HW * C + g * cpg;
    h16_t* yb = y + (long long)b * HW * C + g * cpg;
    auto block_sum = [&](float v, int slot) -> float {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[slot * 4 + (tid >> 6)] = v;
        __syncthreads();
        return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
    };
    uint4 raw[MAXIT];
    int off[MAXIT], vv[MAXIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int i = tid + 256 * it;
        raw[it] = make_uint4(0u, 0u, 0u, 0u);
        off[it] = 0;
        vv[it] = 0;
        if (i < items) {
            const int p = i / vpg, v = i - p * vpg;
            off[it] = p * C + v * 8;
            vv[it] = v * 8;
            raw[it] = *(const uint4*)(xb + off[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (tid + 256 * it < items)
            s += (h16_lo(raw[it].x) + h16_hi(raw[it].x)) + (h16_lo(raw[it].y) + h16_hi(raw[it].y)) + (h16_lo(raw[it].z) + h16_hi(raw[it].z)) +
                 (h16_lo(raw[it].w) + h16_hi(raw[it].w));
    const float n_all = (float)HW * (float)cpg;
    const float mean = block_sum(s, 0) / n_all;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (tid + 256 * it < items) {
            const unsigned w[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = h16_lo(w[k]) - mean, c = h16_hi(w[k]) - mean; q += a * a + c * c; }
        }
    const float rstd = rsqrtf(block_sum(q, 1) / n_all + eps);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (tid + 256 * it < items) {
            const unsigned w[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
            const float* gm = gamma + g * cpg + vv[it];
            const float* bt = beta + g * cpg + vv[it];
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = (h16_lo(w[k]) - mean) * rstd * gm[2 * k] + bt[2 * k];
                o[2 * k + 1] = (h16_hi(w[k]) - mean) * rstd * gm[2 * k + 1] + bt[2 * k + 1];
            }
            if (silu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = silu_f(o[k]);
            }
            uint4 r;
            r.x = pack_h16x2(o[0], o[1]); r.y = pack_h16x2(o[2], o[3]); r.z = pack_h16x2(o[4], o[5]); r.w = pack_h16x2(o[6], o[7]);
            *(uint4*)(yb + off[it]) = r;
        }
}
bool groupnorm_small_applicable(int B, int HW, int C, int G) {
    const int cpg = C / G;
    (void)B;  // must not depend on the batch size: an image's bits may not change with its batch mates (test_batch_equals_single)
    return (cpg & 7) == 0 && cpg * G == C && (long long)HW * (cpg >> 3) <= 8192;
}
void launch_groupnorm_small(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int silu,
                            hipStream_t s) {
    const int items = HW * ((C / G) >> 3);
    if (items <= 256 * 4 && !gp_sw().gn_small_old) hipLaunchKernelGGL(gn_small_reg_kernel<4>, dim3(G, B), dim3(256), 0, s, x, y, gamma, beta, HW, C, G, eps, silu);
    else if (items <= 256 * 12 && !gp_sw().gn_small_old) hipLaunchKernelGGL(gn_small_reg_kernel<12>, dim3(G, B), dim3(256), 0, s, x, y, gamma, beta, HW, C, G, eps, silu);
    else hipLaunchKernelGGL(gn_small_kernel, dim3(G, B), dim3(256), 0, s, x, y, gamma, beta, HW, C, G, eps, silu);
}

void launch_groupnorm_stats(const h16_t* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, float* ws,
                            float* scale, float* shift, hipStream_t s) {
    const int nchunk = gn_nchunk(HW);
    const int nvec = C / 8;
    const int vpt = (nvec + 255) / 256;           // 1 or 2 (C <= 4096)
    const int tpp = (nvec + vpt - 1) / vpt;       // threads per pixel
    const int PL = 256 / tpp;                     // pixel lanes
    const size_t lds1 = (size_t)PL * C * 2 * sizeof(float);
    dim3 grid(nchunk, B);
    if (vpt == 1) hipLaunchKernelGGL(gn_stats_kernel<1>, grid, dim3(256), lds1, s, x, ws, HW, C, G, nchunk, tpp, PL);
    else hipLaunchKernelGGL(gn_stats_kernel<2>, grid, dim3(256), lds1, s, x, ws, HW, C, G, nchunk, tpp, PL);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, (const float*)ws, gamma, beta, scale, shift, HW, C, G, nchunk, eps);
}

// r3: the same pass with a fixed 8-channel vector per thread.  gn_apply_kernel above spent its time on a 64-bit modulo and sixteen 4-byte LDS
// reads per 16 bytes moved (2.6 TB/s on the 300 MB tensors of the VAE, 5.6 % of a pass); here a thread keeps scale / shift of ITS eight
// channels in registers, a workgroup covers R = 256 / (C / 8) consecutive pixels per step (a contiguous 4 KiB of the NHWC tensor) and
// four steps' loads are in flight before the first one is used.  C / 8 > 256 (C = 2560): the vectors are walked in blocks of 256.
template <int SILU>
__global__ __launch_bounds__(256) void gn_apply2_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int HW, int C, int nchunk, int tpr, int R) {
    const int b = blockIdx.y, ck = blockIdx.x, tid = threadIdx.x;
    const int nvec = C >> 3;
    const int r = tid / tpr, v0 = tid - r * tpr;
    if (r >= R) return;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = ck * per, p1 = min(HW, p0 + per);
    const long long base = (long long)b * HW * C;
    for (int v = v0; v < nvec; v += 256) {
        float sc[8], sh[8];
        {
            const float4* sp = (const float4*)(scale + (long long)b * C + v * 8);
            const float4* hp = (const float4*)(shift + (long long)b * C + v * 8);
            const float4 s0 = sp[0], s1 = sp[1], h0 = hp[0], h1 = hp[1];
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        }
        const h16_t* xp = x + base + v * 8;
        h16_t* yp = y + base + v * 8;
        for (int p = p0 + r; p < p1; p += 4 * R) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = p + u * R;
                raw[u] = pp < p1 ? *(const uint4*)(xp + (long long)pp * C) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = p + u * R;
                if (pp >= p1) break;
                const unsigned w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                float o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[2 * k] = __builtin_fmaf(h16_lo(w[k]), sc[2 * k], sh[2 * k]);
                    o[2 * k + 1] = __builtin_fmaf(h16_hi(w[k]), sc[2 * k + 1], sh[2 * k + 1]);
                }
                if (SILU) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = silu_f(o[k]);
                }
                uint4 q;
                q.x = pack_h16x2(o[0], o[1]); q.y = pack_h16x2(o[2], o[3]); q.z = pack_h16x2(o[4], o[5]); q.w = pack_h16x2(o[6], o[7]);
                *(uint4*)(yp + (long long)pp * C) = q;
            }
        }
    }
}

void launch_groupnorm_apply(const h16_t* x, h16_t* y, const float* scale, const float* shift, int B, int HW, int C, int silu, hipStream_t s) {
    const bool old_kernel = gp_sw().gn_apply_old;  // A/B switch
    if (old_kernel) {
        const int nchunk = gn_nchunk(HW);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunk, B), dim3(256), (size_t)2 * C * sizeof(float), s, x, y, scale, shift, HW, C, nchunk, silu);
        return;
    }
    const int nvec = C >> 3;
    const int tpr = nvec < 256 ? nvec : 256;   // threads per pixel row
    const int R = 256 / tpr;                   // pixel rows per workgroup step
    // workgroups per image: ~8 per CU over the batch, at least 4 * R pixels each (one unrolled step)
    int nchunk = HW / (4 * R);
    const int cap = (2048 + B - 1) / B;
    if (nchunk > cap) nchunk = cap;
    if (nchunk < 1) nchunk = 1;
    if (silu) hipLaunchKernelGGL(gn_apply2_kernel<1>, dim3(nchunk, B), dim3(256), 0, s, x, y, scale, shift, HW, C, nchunk, tpr, R);
    else hipLaunchKernelGGL(gn_apply2_kernel<0>, dim3(nchunk, B), dim3(256), 0, s, x, y, scale, shift, HW, C, nchunk, tpr, R);
}

// ws: >= groupnorm_ws_floats() + 2*B*C floats (partials, then scale, then shift)
void launch_groupnorm(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int silu,
                      float* ws, hipStream_t s) {
    float* scale = ws + groupnorm_ws_floats(B, HW, C, G);
    float* shift = scale + (size_t)B * C;
    launch_groupnorm_stats(x, gamma, beta, B, HW, C, G, eps, ws, scale, shift, s);
    launch_groupnorm_apply(x, y, scale, shift, B, HW, C, silu, s);
}

// ---- scale/shift from the per-(pixel tile, channel) partials a conv epilogue wrote (IGemmParams::stats_out) --------------------
// One workgroup per (group, image): 256 threads walk the group's (tile, channel) partials in a fixed order; tile pixel counts
// come from the tiling (mode 0: bm consecutive rows; mode 1: 16x16 tiles clipped at the image edge); Chan-combined variance.
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(const float* __restrict__ part, int mode, int bm, int H, int W, int C, int G,
                                                                 float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float red[8];
    __shared__ float s_mean, s_rstd;
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    const int ntile = mode == 2 ? bm : mode ? tiles_x * tiles_y : (H * W) / bm;
    const float* pb = part + (long long)b * ntile * C * 2;
    const float* cnt = part + (long long)gridDim.y * ntile * C * 2 + (long long)b * ntile;  // mode 2: pixel count of every row
    const int items = ntile * cpg;
    auto tile_px = [&](int t) -> float {
        if (mode == 2) return cnt[t];
        if (!mode) return (float)bm;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        return (float)(min(16, H - ty * 16) * min(16, W - tx * 16));
    };
    float tot = 0.f;
    for (int i = tid; i < items; i += 256) {
        const int t = i / cpg, c = g * cpg + (i - t * cpg);
        tot += pb[((long long)t * C + c) * 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if ((tid & 63) == 0) red[tid >> 6] = tot;
    __syncthreads();
    const float n_all = (float)H * (float)W * (float)cpg;
    const float mean = (red[0] + red[1] + red[2] + red[3]) / n_all;
    float m2 = 0.f;
    for (int i = tid; i < items; i += 256) {
        const int t = i / cpg, c = g * cpg + (i - t * cpg);
        const float nk = tile_px(t), sk = pb[((long long)t * C + c) * 2], qk = pb[((long long)t * C + c) * 2 + 1];
        const float mk = sk / nk;
        m2 += fmaxf(qk - sk * mk, 0.f) + nk * (mk - mean) * (mk - mean);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = m2;
    __syncthreads();
    if (tid == 0) {
        s_mean = mean;
        s_rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / n_all + eps);
    }
    __syncthreads();
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const float sc = s_rstd * gamma[c];
        scale[(long long)b * C + c] = sc;
        shift[(long long)b * C + c] = beta[c] - s_mean * sc;
    }
}

void launch_groupnorm_from_partials(const float* partials, int mode, int bm, int B, int H, int W, int C, int G, float eps, const float* gamma,
                                    const float* beta, float* scale, float* shift, hipStream_t s) {
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(G, B), dim3(256), 0, s, partials, mode, bm, H, W, C, G, eps, gamma, beta, scale, shift);
}

// ---- LayerNorm: one wave per row, row kept in registers (C <= 4096), exact two-pass statistics -------------------
template <int VPT>
__global__ __launch_bounds__(256) void layernorm_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    float v[VPT][8];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = lane + u * 64;
        if (vi < nvec) {
            const uint4 raw = *(const uint4*)(x + (long long)row * C + vi * 8);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[u][2 * k] = h16_lo(w[k]); v[u][2 * k + 1] = h16_hi(w[k]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[u][k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[u][k] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = lane + u * 64;
        if (vi < nvec) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[u][k] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = lane + u * 64;
        if (vi < nvec) {
            const float4 g0 = *(const float4*)(gamma + vi * 8), g1 = *(const float4*)(gamma + vi * 8 + 4);
            const float4 b0 = *(const float4*)(beta + vi * 8), b1 = *(const float4*)(beta + vi * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[u][k] - mean) * rstd * gg[k] + bb[k];
            uint4 r;
            r.x = pack_h16x2(o[0], o[1]); r.y = pack_h16x2(o[2], o[3]); r.z = pack_h16x2(o[4], o[5]); r.w = pack_h16x2(o[6], o[7]);
            *(uint4*)(y + (long long)row * C + vi * 8) = r;
        }
    }
}

void launch_layernorm(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int rows, int C, float eps, hipStream_t s) {
    const int nvec = C / 8;
    const int vpt = (nvec + 63) / 64;
    dim3 grid((rows + 3) / 4);
    if (vpt <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else if (vpt <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else if (vpt <= 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
}

// ---- cross-attention against a TWO-token constant context, folded (GenPercept's empty prompt: BOS, EOS) -----------------
// With two keys the softmax is a sigmoid of the logit difference, and both projections around it collapse into per-head vectors:
//   d_h   = LN2(y) . (Wq_h^T (k0 - k1)_h) / 8            = yhat . U[h] + u0[h]        (LayerNorm affine folded into U, u0)
//   attn2 = v1 + sigmoid(d_h) (v0 - v1)_h  ->  to_out:    o = c0 + sum_h sigmoid(d_h) G[h],   c0 = Wo v1 + bo,  G[h] = Wo[:, h] (v0 - v1)_h
// so   y <- y + c0 + sum_h sigmoid(yhat . U[h] + u0[h]) G[h]   replaces LayerNorm, the to_q GEMM, the attention and the to_out GEMM
// (custom_unet.py / BasicTransformerBlock attn2, genpercept_pipeline.py:425-429) -- exactly, in fp32, with the row in registers.  The
// kernel also emits LN3 of the row it just stored (the input of the feed-forward GEMM), saving that pass too.  HBM-bound: 6 B / element.
// One wave handles R rows at a time (the U / G rows it streams from L2 are reused R times); lane owns 8-channel vectors lane + 64 u.
// HEADS is a template parameter so that both head loops unroll: the U / G loads of all heads are independent and issue together, and the
// HEADS x R logit reductions interleave -- as a run-time loop every head paid an L2 round trip plus a 6-step shuffle chain in sequence
// (78 us for 576 rows of 1280 channels, pure latency).
// LDSM (r3): 0 = U / G / c0 / g3 / b3 straight from global memory (every wave re-streams HEADS x C x 8 bytes per R rows from L2: 470 MB for
// 2304 rows of 1280 channels, 30 us for an 18 MB tensor); 1 = U and the three per-channel vectors are copied to LDS once per workgroup, G still
// streams (20 heads x 1280 channels: U alone is 100 KiB); 2 = G in LDS as well.  Workgroups are persistent (they walk row blocks with a grid
// stride) so that the copy is paid once per CU, not once per 4 R rows.
template <int VPT, int R, int HEADS, int LDSM>
__global__ __launch_bounds__(256) void cross_fold_kernel(const h16_t* __restrict__ y, h16_t* __restrict__ y_out, h16_t* __restrict__ n3_out,
                                                          const float* __restrict__ U, const float* __restrict__ u0, const float* __restrict__ G,
                                                          const float* __restrict__ c0, const float* __restrict__ g3, const float* __restrict__ b3,
                                                          int rows, int C, float eps) {
    extern __shared__ __attribute__((aligned(16))) float cf_sm[];  // LDSM: [HEADS][C] U, ([HEADS][C] G,) [C] c0, [C] g3, [C] b3
    const int lane = threadIdx.x & 63;
    const int nvec = C >> 3;
    const float invC = 1.f / (float)C;
    const unsigned sm_base = (unsigned)(unsigned long long)cf_sm;
    const unsigned sU = sm_base, sG = sU + (unsigned)HEADS * C * 4, sC = LDSM == 2 ? sG + (unsigned)HEADS * C * 4 : sG;
    if (LDSM) {
        const int nU = HEADS * C;
        for (int i = threadIdx.x * 4; i < nU; i += 256 * 4) {
            *(float4*)(cf_sm + i) = *(const float4*)(U + i);
            if (LDSM == 2) *(float4*)(cf_sm + nU + i) = *(const float4*)(G + i);
        }
        float* cs = cf_sm + (LDSM == 2 ? 2 : 1) * nU;
        for (int i = threadIdx.x * 4; i < C; i += 256 * 4) {
            *(float4*)(cs + i) = *(const float4*)(c0 + i);
            *(float4*)(cs + C + i) = g3 ? *(const float4*)(g3 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4*)(cs + 2 * C + i) = b3 ? *(const float4*)(b3 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
    }
    // 8 consecutive floats of a per-channel table: from LDS (byte address) or from global memory
    auto ld8 = [&](unsigned lds_addr, const float* gptr, float (&o)[8]) __attribute__((always_inline)) {
        f32x4_t a, b;
        if (LDSM) { a = *(lds_f4_ptr)lds_addr; b = *(lds_f4_ptr)(lds_addr + 16); }
        else { const float4 x = *(const float4*)gptr, y2 = *(const float4*)(gptr + 4); a = f32x4_t{x.x, x.y, x.z, x.w}; b = f32x4_t{y2.x, y2.y, y2.z, y2.w}; }
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    };
  for (int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R; row0 < rows; row0 += gridDim.x * 4 * R) {
    float v[R][VPT][8];
    // ---- load, LayerNorm statistics (exact two-pass, row in registers)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int vi = lane + u * 64;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (vi < nvec && row0 + r < rows) raw = *(const uint4*)(y + (long long)(row0 + r) * C + vi * 8);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[r][u][2 * k] = h16_lo(w[k]); v[r][u][2 * k + 1] = h16_hi(w[k]); }
        }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < VPT; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[r][u][k];
        s = wave_sum(s);
        mean[r] = s * invC;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if (lane + u * 64 < nvec) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[r][u][k] - mean[r]; q += d * d; }
            }
        q = wave_sum(q);
        rstd[r] = rsqrtf(q * invC + eps);
    }
    // ---- per-head logits d[r][h] = yhat . U[h]: partial sums over this lane's channels for every head, then one batch of reductions
    float d[R][HEADS];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < HEADS; ++h) d[r][h] = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = lane + u * 64;
        if (vi < nvec) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                float uu[8];
                ld8(sU + (unsigned)(h * C + vi * 8) * 4, U + (long long)h * C + vi * 8, uu);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[r][h] += (v[r][u][k] - mean[r]) * uu[k];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < HEADS; ++h) d[r][h] = wave_sum(d[r][h]);
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
        const float uh = u0[h];
#pragma unroll
        for (int r = 0; r < R; ++r) d[r][h] = __builtin_amdgcn_rcpf(1.f + __expf(-(d[r][h] * rstd[r] + uh)));  // softmax over 2 keys
    }
    // ---- y + c0 + sum_h p_h G[h]
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = lane + u * 64;
        if (vi < nvec) {
            float cc[8];
            ld8(sC + (unsigned)(vi * 8) * 4, c0 + vi * 8, cc);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[r][u][k] += cc[k];
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                float gg[8];
                if (LDSM == 2) ld8(sG + (unsigned)(h * C + vi * 8) * 4, nullptr, gg);
                else {
                    const float4 a = *(const float4*)(G + (long long)h * C + vi * 8), b = *(const float4*)(G + (long long)h * C + vi * 8 + 4);
                    gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w; gg[4] = b.x; gg[5] = b.y; gg[6] = b.z; gg[7] = b.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[r][u][k] += d[r][h] * gg[k];
            }
        }
    }
    // ---- store the new trunk row, then LN3 of the values AS STORED (what a separate LayerNorm pass would read)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int vi = lane + u * 64;
            uint4 pk;
            pk.x = pack_h16x2(v[r][u][0], v[r][u][1]); pk.y = pack_h16x2(v[r][u][2], v[r][u][3]);
            pk.z = pack_h16x2(v[r][u][4], v[r][u][5]); pk.w = pack_h16x2(v[r][u][6], v[r][u][7]);
            if (vi < nvec && row0 + r < rows) *(uint4*)(y_out + (long long)(row0 + r) * C + vi * 8) = pk;
            const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[r][u][2 * k] = h16_lo(w[k]); v[r][u][2 * k + 1] = h16_hi(w[k]); }
            if (vi < nvec) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[r][u][k];
            }
        }
        if (!n3_out) continue;
        s = wave_sum(s);
        const float m3 = s * invC;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if (lane + u * 64 < nvec) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float dd = v[r][u][k] - m3; q += dd * dd; }
            }
        q = wave_sum(q);
        const float r3 = rsqrtf(q * invC + eps);
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int vi = lane + u * 64;
            if (vi < nvec && row0 + r < rows) {
                float gg[8], bt[8];
                ld8(sC + (unsigned)(C + vi * 8) * 4, g3 + vi * 8, gg);
                ld8(sC + (unsigned)(2 * C + vi * 8) * 4, b3 + vi * 8, bt);
                float o8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o8[k] = (v[r][u][k] - m3) * r3 * gg[k] + bt[k];
                uint4 pk;
                pk.x = pack_h16x2(o8[0], o8[1]); pk.y = pack_h16x2(o8[2], o8[3]); pk.z = pack_h16x2(o8[4], o8[5]); pk.w = pack_h16x2(o8[6], o8[7]);
                *(uint4*)(n3_out + (long long)(row0 + r) * C + vi * 8) = pk;
            }
        }
    }
  }
}

// y_out may alias y (a wave reads its rows completely before it writes them); n3_out optional.  C % 8 == 0, C <= 1536.
bool cross_attn_fold_supported(int C, int heads) {
    return (C % 8) == 0 && C <= 1536 && C == 64 * heads && (heads == 1 || heads == 2 || heads == 4 || heads == 5 || heads == 10 || heads == 20);
}
template <int VPT, int R, int HEADS>
static void launch_cross_fold_one(const h16_t* y, h16_t* y_out, h16_t* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                                  const float* g3, const float* b3, int rows, int C, float eps, hipStream_t s) {
    const int mode_env = gp_sw().xfold_lds;  // A/B switch: 0 = the r2 kernel
    const int blocks = (rows + 4 * R - 1) / (4 * R);
    const size_t tab = (size_t)HEADS * C * 4, vec = (size_t)3 * C * 4;
    int mode = 2 * tab + vec <= 64 * 1024 ? 2 : (tab + vec <= 120 * 1024 ? 1 : 0);  // tables in LDS when >= 2 (mode 2) / 1 (mode 1) workgroups fit a CU
    if (mode_env >= 0 && mode_env < mode) mode = mode_env;
    if (mode == 0 || blocks < 64) {
        hipLaunchKernelGGL((cross_fold_kernel<VPT, R, HEADS, 0>), dim3(blocks), dim3(256), 0, s, y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps);
        return;
    }
    const size_t lds = (mode == 2 ? 2 : 1) * tab + vec;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = ncu * per_cu;
    if (grid > blocks) grid = blocks;
    static unsigned long long attr_mask = 0;
    if (mode == 2) {
        gp_once_per_device(&attr_mask, [&] { (void)hipFuncSetAttribute((const void*)cross_fold_kernel<VPT, R, HEADS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); });
        hipLaunchKernelGGL((cross_fold_kernel<VPT, R, HEADS, 2>), dim3(grid), dim3(256), lds, s, y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps);
    } else {
        static unsigned long long attr_mask1 = 0;
        gp_once_per_device(&attr_mask1, [&] { (void)hipFuncSetAttribute((const void*)cross_fold_kernel<VPT, R, HEADS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); });
        hipLaunchKernelGGL((cross_fold_kernel<VPT, R, HEADS, 1>), dim3(grid), dim3(256), lds, s, y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps);
    }
}
template <int VPT, int HEADS>
static void launch_cross_fold_r(const h16_t* y, h16_t* y_out, h16_t* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                                const float* g3, const float* b3, int rows, int C, float eps, hipStream_t s) {
    // rows per wave: 4 (the U / G rows a wave streams are reused four times) once that still leaves >= 2 waves per SIMD on the chip
    // r5: at C = 1280 (VPT 3, 20 heads) only U fits the LDS and every wave streams the 100 KiB of G from L2 per row GROUP: three rows per wave on the
    // 24 x 24 maps (2304 rows: 236 -> 78 MB of L2 reads per launch, 55 us before), two on the 12 x 12 maps (576 rows)
    if constexpr (VPT == 3) {
        if (rows >= 2048 && rows < 4096) { launch_cross_fold_one<VPT, 3, HEADS>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s); return; }
        if (rows >= 512 && rows < 4096) { launch_cross_fold_one<VPT, 2, HEADS>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s); return; }
    }
    if (rows >= 16384 && VPT * HEADS <= 20) launch_cross_fold_one<VPT, 4, HEADS>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s);
    else if (rows >= 4096) launch_cross_fold_one<VPT, 2, HEADS>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s);
    else launch_cross_fold_one<VPT, 1, HEADS>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s);
}
void launch_cross_attn_fold(const h16_t* y, h16_t* y_out, h16_t* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                            const float* g3, const float* b3, int rows, int C, int heads, float eps, hipStream_t s) {
    const int vpt = (C / 8 + 63) / 64;
#define GP_CF(V, H) launch_cross_fold_r<V, H>(y, y_out, n3_out, U, u0, G, c0, g3, b3, rows, C, eps, s)
    (void)vpt;  // C == 64 * heads: 8 * heads channel vectors per row
    switch (heads) {
        case 1: GP_CF(1, 1); break;
        case 2: GP_CF(1, 2); break;
        case 4: GP_CF(1, 4); break;
        case 5: GP_CF(1, 5); break;     // SD2.1: 320 / 640 / 1280 channels
        case 10: GP_CF(2, 10); break;
        default: GP_CF(3, 20); break;
    }
#undef GP_CF
}

GP_SAT_TU(norm)  // fp16 build: address of this translation unit's saturation flag (common.h)
