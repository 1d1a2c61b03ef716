// Engine: weight ingestion / constant folding / bf16 packing, activation pool, and the static forward schedules of
// the SD2.1 VAE encoder, UNet single step (t fixed, context fixed), VAE decoder and DPT head, all enqueued on one
// HIP stream.  Exposed through the C-ABI of include/genpercept_hip.h.
//
// Control flow mirrors (never copies) the reference: genpercept/genpercept_pipeline.py:399-526 (single_infer,
// encode_rgb, decode_pred), genpercept/models/custom_unet.py:109-119,146-170,273,305-415 (UNet forward, skip order,
// upsample_size, multi_level_feats), genpercept/models/dpt_head.py:213-335,443-582 (DPT neck/head); module internals
// follow the public SD2.1 architecture.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/genpercept_hip.h"
#include "kernels.h"

#define HIPCHK(x)                                                                                         \
    do {                                                                                                  \
        hipError_t _e = (x);                                                                              \
        if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
    } while (0)

namespace {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// fp32 -> the library's 16-bit element (common.h), round-to-nearest-even
inline h16_t f_to_h16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
#if GP_F16
    const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (h16_t)(sign | 0x7e00u);                    // NaN
    if (a >= 0x477ff000u) return (h16_t)(sign | 0x7bffu);                   // >= 65520 rounds past the largest finite value: saturate
    if (a < 0x33000001u) return (h16_t)sign;                                // <= 2^-25: rounds to zero
    if (a < 0x38800000u) {                                                  // subnormal result: value = m * 2^-24
        const int e = (int)(a >> 23);                                       // biased fp32 exponent, 102 .. 112
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;
        const int sh = 126 - e;                                             // 14 .. 24: bits dropped from the 24-bit significand
        const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        return (h16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = a + 0xfffu + ((a >> 13) & 1u);                       // round the 13 dropped bits to nearest even
    return (h16_t)(sign | ((r - 0x38000000u) >> 13));
#else
    if ((u & 0x7fffffffu) > 0x7f800000u) return (h16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (h16_t)(u >> 16);
#endif
}
inline float half_to_float(uint16_t h) {
    const uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400)) { mm <<= 1; ++sh; }
            u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ff) << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline float h16_to_float_host(h16_t h) {
#if GP_F16
    return half_to_float(h);
#else
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Pool {
    std::multimap<size_t, void*> free_;
    std::unordered_map<void*, size_t> size_;
    std::unordered_map<void*, int> live_;  // buffers handed out and not yet returned (an aborted call returns them all: release_live)
    size_t total = 0;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = free_.lower_bound(bytes);
        void* p = nullptr;
        if (it != free_.end() && it->first <= bytes * 2 + (1u << 20)) {
            p = it->second;
            free_.erase(it);
        } else {
            HIPCHK(hipMalloc(&p, bytes));
            size_[p] = bytes;
            total += bytes;
        }
        live_[p] = 1;
        return p;
    }
    void release(void* p) {
        if (!p) return;
        if (!live_.erase(p)) return;  // not outstanding (already returned by release_live)
        free_.insert({size_.at(p), p});
    }
    void release_live() {
        for (auto& kv : live_) free_.insert({size_.at(kv.first), kv.first});
        live_.clear();
    }
    void destroy() {
        for (auto& kv : size_) hipFree(kv.first);
        size_.clear();
        free_.clear();
        live_.clear();
    }
};

struct Act {
    h16_t* p = nullptr;
    // contract precision (contract.hip): a STORED tensor is fp32 (`f`, C channels per pixel); a matrix-product operand is a split tensor (`p`, with
    // C = 3 x the logical width: [hi | lo | hi] blocks).  Exactly one of p / f is set.
    float* f = nullptr;
    int B = 0, H = 0, W = 0, C = 0;  // C = allocated channels (row stride)
    // GroupNorm partial statistics written by the producing conv's epilogue ([pixel tile][C][2] fp32; owned with p)
    float* st = nullptr;
    int st_mode = 0, st_bm = 0;
    long long pixels() const { return (long long)B * H * W; }
};

struct PackedW {
    h16_t* w = nullptr;   // [n_rows][taps][cin_pad]
    h16_t* w_ph = nullptr;  // x2-upsample convs only: [n_rows][4 phases][2 x 2 taps][cin_pad], kernel rows / columns on the same source pixel summed (pack_phases)
    float* bias = nullptr; // [cout] or null
    int cout = 0, cin_pad = 0, ks = 1, n_rows = 0;
};
struct NormW {
    float* g = nullptr;
    float* b = nullptr;
    int C = 0;
};
struct ResW {
    NormW n1, n2;
    PackedW c1, c2, sc;
    bool has_sc = false;
    // time-embedding fold (UNet only)
    bool has_temb = false;
    std::vector<float> c1_bias_h, tw_h, tb_h;  // conv1.bias, time_emb_proj.{weight,bias}
    size_t temb_off = 0;                       // where c1.bias lives in the engine's time-embedding bias arena (floats)
};
struct TfW {
    NormW gn, ln1, ln2, ln3;
    PackedW proj_in, qkv, qk, v, o1, q2, o2, ff1, ff2, proj_out;  // qk / v: views of qkv's rows [0, 2C) / [2C, 3C)
    float* kc = nullptr;  // folded cross-attention keys / values [L][C] fp32
    float* vc = nullptr;
    std::vector<float> wk_h, wv_h;  // attn2.to_k / to_v [C][D]
    // two-token context: the whole attn2 branch folded into per-head vectors (norm.hip: cross_fold_kernel); null when L != 2
    float *fU = nullptr, *fu0 = nullptr, *fG = nullptr, *fc0 = nullptr;
    std::vector<float> wq2_h, wo2_h, bo2_h, ln2g_h, ln2b_h;  // attn2.to_q / to_out.0, norm2 (host copies for the fold)
    int C = 0, heads = 0;
};
struct VaeAttnW {
    NormW gn;
    PackedW qk, v, o;
    PackedW qkv;  // contract precision: to_q | to_k | to_v stacked, unscaled (the softmax kernel applies 1 / sqrt(C))
    int C = 0;
};

}  // namespace

struct SatFlags { unsigned* f[16]; int n; };
// one lane per translation unit: a set flag becomes one event of the engine's counter and is cleared for the next call
__global__ void collect_saturation_kernel(SatFlags fl, unsigned* counter) {
    const int i = threadIdx.x;
    if (i < fl.n) {
        const unsigned v = __hip_atomic_load(fl.f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) {
            __hip_atomic_store(fl.f[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(counter, 1u);
        }
    }
}

struct gp_engine {
    gp_config cfg;
    std::string err;
    std::unordered_map<std::string, HostTensor> host;
    bool finalized = false;
    hipStream_t st = nullptr;
    Pool pool;
    std::vector<void*> weights_dev;  // everything hipMalloc'ed for weights
    h16_t* zero = nullptr;
    float* gn_ws = nullptr;
    size_t gn_ws_floats = 0;
    hipStream_t last_stream = nullptr;  // the pool recycles buffers in stream order: a change of stream is fenced (check_ready)
    bool stream_seen = false;
    h16_t* conv_in_w27 = nullptr;  // VAE encoder conv_in as a [Cout][32] (K = 27) matrix for rgb_conv_in_kernel
    bool fuse_gn = true;   // GENPERCEPT_NO_GN_FUSION=1 keeps the separate apply pass (A/B measurements)
    int gn_fuse_max_slices = 1;        // GENPERCEPT_GN_FUSE_MAX_SLICES: on large maps fuse the apply only into convs with at most this many 128-channel output slices
    int gn_fuse_always_below_px = 0;   // GENPERCEPT_GN_FUSE_BELOW_PX: maps with fewer pixels per image fuse whatever the slice count (r1: 16384; r2 in-box
                                       // A/B with the faster apply / one-launch small-map GroupNorm: 0 is 1.3 ms per pass faster)
    bool fuse_stats = true;  // GENPERCEPT_NO_STATS_FUSION=1 keeps the separate statistics pass
    // gp_set_precision(GP_PREC_CONTRACT): fp32 storage + split-bf16 operands (three MFMAs per product over a tripled K), see contract.hip.
    // Weights are packed [n_rows][taps][3 cin_pad] in B order [hi | hi | lo]; every PackedW::cin_pad below is then the TRIPLED width.
    bool contract = false;

    std::unordered_map<std::string, PackedW> convs;
    std::unordered_map<std::string, NormW> norms;
    std::unordered_map<std::string, ResW> resnets;
    std::unordered_map<std::string, TfW> tfs;
    std::unordered_map<std::string, VaeAttnW> vattn;
    std::vector<float> te_w1, te_b1, te_w2, te_b2;  // time_embedding MLP (host)
    std::vector<float> ctx;                          // [L][D]
    int ctx_L = 0, ctx_D = 0;
    float timestep = 1.f;
    int ncu = 256;                        // compute units of the device (persistent launches size their grids by it)
    float* temb_arena = nullptr;          // conv1 biases of all UNet resnets at the current timestep
    size_t temb_total = 0;
    std::map<float, float*> temb_cache;   // timestep -> device copy of the arena's contents
    std::vector<float> pq_w, pq_b;  // post_quant_conv
    float* pq_w_dev = nullptr;
    float* pq_b_dev = nullptr;
    float* dpt_w_dev = nullptr;  // head.head.4 weight [32]
    float dpt_b = 0.f;
    // fp16 build: saturating conversions that clipped (common.h: gp_sat_flag per translation unit), collected at the end of every call
    double flops_halo_exec = 0.0;   // executed (not algorithmic) flops of the halo-conv launches, see gp_halo_executed_flops
    SatFlags sat_flags{};
    unsigned* sat_dev = nullptr;          // [1] events since the last reset
    void collect_saturation() {
        if (sat_dev && sat_flags.n > 0) hipLaunchKernelGGL(collect_saturation_kernel, dim3(1), dim3(64), 0, st, sat_flags, sat_dev);
    }

    // profiling
    int prof = 0;
    gp_timings tm{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_kind;
    size_t ev_used = 0;

    // ---------------------------------------------------------------------------------------------------------------
    const HostTensor& H(const std::string& n) const {
        auto it = host.find(n);
        if (it == host.end()) throw std::out_of_range("missing weight: " + n);
        return it->second;
    }
    bool has(const std::string& n) const { return host.find(n) != host.end(); }

    template <typename T>
    T* upload(const T* src, size_t n) {
        T* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, n * sizeof(T) + 256));
        HIPCHK(hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
        weights_dev.push_back(d);
        return d;
    }

    // GEGLU projection rows [value(0..C4) ; gate(0..C4)] -> packed order: every 32-row block holds 16 outputs, value j at
    // 8*(j%16/4) + j%4 and its gate 4 rows further, which is where the igemm epilogue finds them in one lane.
    static int geglu_row(int n, int cout) {
        const int half = cout / 2;
        const bool gate = n >= half;
        const int r = gate ? n - half : n;
        return (r / 16) * 32 + ((r % 16) / 4) * 8 + (gate ? 4 : 0) + (r % 4);
    }
    // Pack [cout][cin][ks][ks] fp32 -> [n_rows][taps][cin_pad] bf16 (+ optional GEGLU row interleave).
    // split: cin_pad is the logical padded width, the row holds 3 cin_pad elements per tap in B order [hi | hi | lo] (contract precision)
    static void pack_rows(const float* w, int cout, int cin, int ks, int cin_pad, bool geglu, std::vector<h16_t>& out, int row0, int n_rows_total,
                          bool split = false) {
        const int taps = ks * ks;
        (void)n_rows_total;
        const size_t kw = split ? (size_t)3 * cin_pad : (size_t)cin_pad;  // elements per tap
        for (int n = 0; n < cout; ++n) {
            int dst = n;
            if (geglu) {
                dst = geglu_row(n, cout);
            }
            h16_t* o = out.data() + (size_t)(row0 + dst) * taps * kw;
            const float* wi = w + (size_t)n * cin * taps;
            for (int c = 0; c < cin; ++c)
                for (int t = 0; t < taps; ++t) {
                    const float x = wi[(size_t)c * taps + t];
                    const h16_t hi = f_to_h16_host(x);
                    o[(size_t)t * kw + c] = hi;
                    if (split) {
                        o[(size_t)t * kw + cin_pad + c] = hi;
                        o[(size_t)t * kw + 2 * cin_pad + c] = f_to_h16_host(x - h16_to_float_host(hi));
                    }
                }
        }
    }
    // The x2-nearest-upsample 3x3 conv as four 2 x 2-tap phase convolutions on the source map (conv_halo.hip, PH): output pixel (2y + a, 2x + b) reads
    // source rows {y - 1 + a, y + a} with the kernel rows that fall onto the same source row summed -- a = 0: {w[0]}, {w[1] + w[2]}; a = 1: {w[0] + w[1]},
    // {w[2]} -- and the same along x.  Sums in fp32, ONE rounding to the element type.  Layout [n_rows][phase = 2 a + b][tap = 2 ty + tx][cin_pad].
    // split (contract precision): 3 cin_pad elements per tap in B order [hi | hi | lo] of the fp32 sum, like pack_rows
    static void pack_phase_rows(const float* w, int cout, int cin, int cin_pad, std::vector<h16_t>& out, bool split = false) {
        static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};  // [phase][tap]: kernel index range [lo, hi]
        const size_t kw = split ? (size_t)3 * cin_pad : (size_t)cin_pad;
        for (int n = 0; n < cout; ++n)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int ty = 0; ty < 2; ++ty)
                        for (int tx = 0; tx < 2; ++tx) {
                            h16_t* o = out.data() + (((size_t)n * 4 + (2 * a + b)) * 4 + (2 * ty + tx)) * kw;
                            for (int c = 0; c < cin; ++c) {
                                const float* wi = w + ((size_t)n * cin + c) * 9;
                                float acc = 0.f;
                                for (int ky = lo[a][ty]; ky <= hi[a][ty]; ++ky)
                                    for (int kx = lo[b][tx]; kx <= hi[b][tx]; ++kx) acc += wi[ky * 3 + kx];
                                const h16_t h = f_to_h16_host(acc);
                                o[c] = h;
                                if (split) {
                                    o[cin_pad + c] = h;
                                    o[2 * cin_pad + c] = f_to_h16_host(acc - h16_to_float_host(h));
                                }
                            }
                        }
    }
    void pack_phases(PackedW& pw, const std::string& name) {
        const HostTensor& w = H(name + ".weight");
        if (w.shape.size() != 4 || w.shape[2] != 3 || w.shape[3] != 3) throw std::invalid_argument(name + ": not a 3x3 conv");
        const int cout = (int)w.shape[0], cin = (int)w.shape[1];
        std::vector<h16_t> buf((size_t)pw.n_rows * 16 * pw.cin_pad, 0);  // (contract precision: pw.cin_pad is already the tripled width)
        pack_phase_rows(w.v.data(), cout, cin, contract ? pw.cin_pad / 3 : pw.cin_pad, buf, contract);
        pw.w_ph = upload(buf.data(), buf.size());
    }
    PackedW pack(const float* w, const float* bias, int cout, int cin, int ks, int cin_pad, bool geglu = false) {
        PackedW pw;
        pw.cout = cout; pw.cin_pad = contract ? 3 * cin_pad : cin_pad; pw.ks = ks; pw.n_rows = gp_packed_rows(cout);
        std::vector<h16_t> buf((size_t)pw.n_rows * ks * ks * pw.cin_pad, 0);
        pack_rows(w, cout, cin, ks, cin_pad, geglu, buf, 0, pw.n_rows, contract);
        pw.w = upload(buf.data(), buf.size());
        if (bias) {
            std::vector<float> b(bias, bias + cout);
            if (geglu) {
                for (int n = 0; n < cout; ++n) b[geglu_row(n, cout)] = bias[n];
            }
            pw.bias = upload(b.data(), b.size());
        }
        return pw;
    }
    PackedW pack_named(const std::string& name, int ks, int cin_pad_override = 0, bool geglu = false) {
        const HostTensor& w = H(name + ".weight");
        const int cout = (int)w.shape[0], cin = (int)w.shape[1];
        const float* b = has(name + ".bias") ? H(name + ".bias").v.data() : nullptr;
        const int cin_pad = cin_pad_override ? cin_pad_override : round_up(cin, 64);
        return pack(w.v.data(), b, cout, cin, ks, cin_pad, geglu);
    }
    // two linear layers stacked along the output dimension ([Wa; Wb]), optional biases
    // (scale_a multiplies the first layer's weight and bias: the softmax scale of an attention folded into its query projection)
    PackedW pack_stacked(const std::vector<std::string>& names, float scale_first = 1.f) {
        const int cin = (int)(H(names[0] + ".weight").numel() / H(names[0] + ".weight").shape[0]);
        int ctot = 0;
        for (auto& n : names) ctot += (int)H(n + ".weight").shape[0];
        std::vector<float> w((size_t)ctot * cin), bias;
        const bool with_bias = has(names[0] + ".bias");
        if (with_bias) bias.resize(ctot);
        int row = 0;
        for (size_t k = 0; k < names.size(); ++k) {
            const HostTensor& wk = H(names[k] + ".weight");
            const int ck = (int)wk.shape[0];
            if ((int)(wk.numel() / ck) != cin) throw std::invalid_argument("stacked projections must share their input width");
            memcpy(w.data() + (size_t)row * cin, wk.v.data(), (size_t)ck * cin * 4);
            if (with_bias) memcpy(bias.data() + row, H(names[k] + ".bias").v.data(), (size_t)ck * 4);
            if (k == 0 && scale_first != 1.f) {
                for (size_t i = 0; i < (size_t)ck * cin; ++i) w[i] *= scale_first;
                for (int i = 0; i < ck && with_bias; ++i) bias[i] *= scale_first;
            }
            row += ck;
        }
        return pack(w.data(), with_bias ? bias.data() : nullptr, ctot, cin, 1, round_up(cin, 64));
    }
    PackedW pack_stacked(const std::string& a, const std::string& b, float scale_a = 1.f) { return pack_stacked(std::vector<std::string>{a, b}, scale_a); }
    NormW norm_named(const std::string& name) {
        NormW n;
        const HostTensor& g = H(name + ".weight");
        n.C = (int)g.numel();
        n.g = upload(g.v.data(), g.v.size());
        n.b = upload(H(name + ".bias").v.data(), (size_t)n.C);
        return n;
    }

    void build_resnet(const std::string& p, bool temb) {
        ResW r;
        r.n1 = norm_named(p + ".norm1");
        r.n2 = norm_named(p + ".norm2");
        r.c1 = pack_named(p + ".conv1", 3);
        r.c2 = pack_named(p + ".conv2", 3);
        r.has_sc = has(p + ".conv_shortcut.weight");
        if (r.has_sc) r.sc = pack_named(p + ".conv_shortcut", 1);
        if (temb) {
            r.has_temb = true;
            r.c1_bias_h = H(p + ".conv1.bias").v;
            r.tw_h = H(p + ".time_emb_proj.weight").v;
            r.tb_h = H(p + ".time_emb_proj.bias").v;
        }
        resnets[p] = std::move(r);
    }
    void build_transformer(const std::string& p, int heads) {
        TfW t;
        const std::string b = p + ".transformer_blocks.0";
        t.gn = norm_named(p + ".norm");
        t.proj_in = pack_named(p + ".proj_in", 1);
        t.C = t.proj_in.cout;
        t.heads = heads;
        if (t.C != heads * 64) throw std::invalid_argument("self-attention head_dim must be 64 (" + p + ")");
        t.ln1 = norm_named(b + ".norm1");
        t.ln2 = norm_named(b + ".norm2");
        t.ln3 = norm_named(b + ".norm3");
        // to_q | to_k | to_v stacked along the output dimension: ONE projection launch writes q | k row-major and V transposed (pgemm.hip:
        // IGemmParams::vt_out); the two-launch form (q | k GEMM + transposed V GEMM) reads the same buffer through row views
        t.qkv = pack_stacked({b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"});
        t.qk = t.qkv; t.qk.cout = 2 * t.C;
        t.v = t.qkv; t.v.cout = t.C; t.v.w = t.qkv.w + (size_t)2 * t.C * t.qkv.cin_pad; t.v.n_rows = t.qkv.n_rows - 2 * t.C;
        if (t.qkv.bias) { t.v.bias = t.qkv.bias + 2 * t.C; }
        t.o1 = pack_named(b + ".attn1.to_out.0", 1);
        t.q2 = pack_named(b + ".attn2.to_q", 1);
        t.o2 = pack_named(b + ".attn2.to_out.0", 1);
        t.ff1 = pack_named(b + ".ff.net.0.proj", 1, 0, true);
        t.ff2 = pack_named(b + ".ff.net.2", 1);
        t.proj_out = pack_named(p + ".proj_out", 1);
        t.wk_h = H(b + ".attn2.to_k.weight").v;
        t.wv_h = H(b + ".attn2.to_v.weight").v;
        t.wq2_h = H(b + ".attn2.to_q.weight").v;
        t.wo2_h = H(b + ".attn2.to_out.0.weight").v;
        t.bo2_h = H(b + ".attn2.to_out.0.bias").v;
        t.ln2g_h = H(b + ".norm2.weight").v;
        t.ln2b_h = H(b + ".norm2.bias").v;
        tfs[p] = std::move(t);
    }
    void build_vae_attn(const std::string& p) {
        VaeAttnW a;
        const bool newn = has(p + ".to_q.weight");
        const std::string q = p + (newn ? ".to_q" : ".query"), k = p + (newn ? ".to_k" : ".key"), v = p + (newn ? ".to_v" : ".value"),
                          o = p + (newn ? ".to_out.0" : ".proj_attn");
        a.gn = norm_named(p + ".group_norm");
        a.C = a.gn.C;
        // softmax(q k^T / sqrt(C)): the scale goes into the query projection, so the logits the score GEMM writes are the SCALED ones
        // (what diffusers keeps in fp16 too); raw 512-term dot products of a real checkpoint can pass fp16's 65504
        if (contract) a.qkv = pack_stacked(std::vector<std::string>{q, k, v});
        else {
            a.qk = pack_stacked(q, k, 1.0f / std::sqrt((float)a.C));
            a.v = pack_named(v, 1);
        }
        a.o = pack_named(o, 1);
        vattn[p] = std::move(a);
    }

    // conv1 bias + time_emb_proj(SiLU(time_embedding(t))) of every UNet resnet at timestep t, concatenated in arena order; computed on
    // the host in double once per distinct t and kept on the device, so a change of timestep is one stream-ordered D2D copy
    // (the multi-step archs walk 10-50 timesteps per image; genpercept_pipeline.py:447-463).
    const float* timestep_biases(float t) {
        // an engine finalised without the UNet's time-embedding weights (a VAE-only engine) has nothing to fold and no arena to copy into
        if (te_w1.empty() || te_b1.empty() || !temb_arena) throw std::out_of_range("missing weight: unet.time_embedding.* (this engine holds no UNet)");
        auto it = temb_cache.find(t);
        if (it != temb_cache.end()) return it->second;
        const int c0 = cfg.unet_block_out[0], te = c0 * 4, half = c0 / 2;
        std::vector<double> temb(c0), e1(te), emb(te);
        for (int i = 0; i < half; ++i) {
            const double f = std::exp(-std::log(10000.0) * i / half);
            temb[i] = std::cos((double)t * f);
            temb[half + i] = std::sin((double)t * f);
        }
        for (int o = 0; o < te; ++o) {
            double a = te_b1[o];
            for (int i = 0; i < c0; ++i) a += (double)te_w1[(size_t)o * c0 + i] * temb[i];
            e1[o] = a / (1.0 + std::exp(-a));
        }
        for (int o = 0; o < te; ++o) {
            double a = te_b2[o];
            for (int i = 0; i < te; ++i) a += (double)te_w2[(size_t)o * te + i] * e1[i];
            emb[o] = a / (1.0 + std::exp(-a));  // SiLU(emb), the input of every time_emb_proj
        }
        std::vector<float> all(temb_total);
        std::vector<const ResW*> rs;
        for (auto& kv : resnets)
            if (kv.second.has_temb) rs.push_back(&kv.second);
        // 26 M multiply-adds in double: spread over a few host threads (a 50-step schedule folds 50 timesteps on its first image)
        const unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool_thr;
        for (unsigned w = 0; w < nthr; ++w)
            pool_thr.emplace_back([&, w] {
                for (size_t k = w; k < rs.size(); k += nthr) {
                    const ResW& r = *rs[k];
                    for (int o = 0; o < r.c1.cout; ++o) {
                        double a = (double)r.c1_bias_h[o] + r.tb_h[o];
                        const float* wrow = &r.tw_h[(size_t)o * te];
                        for (int i = 0; i < te; ++i) a += (double)wrow[i] * emb[i];
                        all[r.temb_off + o] = (float)a;
                    }
                }
            });
        for (auto& t : pool_thr) t.join();
        float* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, temb_total * sizeof(float)));
        weights_dev.push_back(d);
        HIPCHK(hipMemcpy(d, all.data(), temb_total * sizeof(float), hipMemcpyHostToDevice));
        temb_cache[t] = d;
        return d;
    }
    void fold_timestep() {  // synchronous form (gp_finalize / gp_set_timestep)
        if (te_w1.empty()) return;
        if (!temb_arena) {  // every temb resnet's conv1 bias re-pointed into one arena
            temb_total = 0;
            for (auto& kv : resnets)
                if (kv.second.has_temb) { kv.second.temb_off = temb_total; temb_total += (size_t)((kv.second.c1.cout + 63) / 64) * 64; }
            HIPCHK(hipMalloc((void**)&temb_arena, temb_total * sizeof(float)));
            weights_dev.push_back(temb_arena);
            for (auto& kv : resnets)
                if (kv.second.has_temb) kv.second.c1.bias = temb_arena + kv.second.temb_off;
        }
        HIPCHK(hipMemcpy(temb_arena, timestep_biases(timestep), temb_total * sizeof(float), hipMemcpyDeviceToDevice));
    }
    void set_timestep_on_stream(float t) {  // inside a pass: ordered after the launches that still read the previous biases
        const float* src = timestep_biases(t);
        timestep = t;
        HIPCHK(hipMemcpyAsync(temb_arena, src, temb_total * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    void fold_context() {
        if (ctx.empty()) return;
        for (auto& kv : tfs) {
            TfW& t = kv.second;
            const int C = t.C, D = ctx_D, L = ctx_L;
            if ((int)(t.wk_h.size() / C) != D) throw std::invalid_argument("context dim does not match attn2.to_k");
            std::vector<float> kc((size_t)L * C), vc((size_t)L * C);
            for (int l = 0; l < L; ++l)
                for (int c = 0; c < C; ++c) {
                    double ak = 0, av = 0;
                    for (int d = 0; d < D; ++d) {
                        ak += (double)t.wk_h[(size_t)c * D + d] * ctx[(size_t)l * D + d];
                        av += (double)t.wv_h[(size_t)c * D + d] * ctx[(size_t)l * D + d];
                    }
                    kc[(size_t)l * C + c] = (float)ak;
                    vc[(size_t)l * C + c] = (float)av;
                }
            t.kc = upload(kc.data(), kc.size());
            t.vc = upload(vc.data(), vc.size());
            t.fU = t.fu0 = t.fG = t.fc0 = nullptr;
            if (L == 2 && cross_attn_fold_supported(C, t.heads) && !gp_sw().no_cross_fold) {
                // softmax over two keys = sigmoid of the logit difference; see cross_fold_kernel (norm.hip) for the algebra
                const int hd = 64, nh = t.heads;
                std::vector<float> U((size_t)nh * C), u0(nh), G((size_t)nh * C), c0(C);
                for (int h = 0; h < nh; ++h) {
                    double b0 = 0;
                    for (int c = 0; c < C; ++c) {
                        double a = 0;
                        for (int j = h * hd; j < (h + 1) * hd; ++j) a += (double)t.wq2_h[(size_t)j * C + c] * ((double)kc[j] - (double)kc[(size_t)C + j]);
                        a *= 0.125;  // 1 / sqrt(head_dim)
                        U[(size_t)h * C + c] = (float)(a * t.ln2g_h[c]);
                        b0 += a * t.ln2b_h[c];
                    }
                    u0[h] = (float)b0;
                }
                for (int c = 0; c < C; ++c) {
                    double a = t.bo2_h[c];
                    for (int j = 0; j < C; ++j) a += (double)t.wo2_h[(size_t)c * C + j] * vc[(size_t)C + j];
                    c0[c] = (float)a;
                    for (int h = 0; h < nh; ++h) {
                        double g = 0;
                        for (int j = h * hd; j < (h + 1) * hd; ++j) g += (double)t.wo2_h[(size_t)c * C + j] * ((double)vc[j] - (double)vc[(size_t)C + j]);
                        G[(size_t)h * C + c] = (float)g;
                    }
                }
                t.fU = upload(U.data(), U.size());
                t.fu0 = upload(u0.data(), u0.size());
                t.fG = upload(G.data(), G.size());
                t.fc0 = upload(c0.data(), c0.size());
            }
        }
    }

    void finalize() {
        if (finalized) throw std::logic_error("gp_finalize called twice");
        HIPCHK(hipSetDevice(cfg.device));
        gp_switches_reload();  // the A/B switches of this process as of now (never read again on the launch path)
        fuse_gn = !gp_sw().no_gn_fusion;
        if (gp_sw().gn_fuse_max_slices >= 0) gn_fuse_max_slices = gp_sw().gn_fuse_max_slices;
        if (gp_sw().gn_fuse_below_px >= 0) gn_fuse_always_below_px = gp_sw().gn_fuse_below_px;
        fuse_stats = !gp_sw().no_stats_fusion;
        {
            std::vector<h16_t> z(2048, 0);
            zero = upload(z.data(), z.size());
        }
        const bool have_vae = has("vae.encoder.conv_in.weight") || has("vae.decoder.conv_in.weight");
        const bool have_unet = has("unet.conv_in.weight");
        // ---------------- VAE ----------------
        if (has("vae.encoder.conv_in.weight")) {
            convs["vae.encoder.conv_in"] = pack_named("vae.encoder.conv_in", 3);
            for (int i = 0; i < 4; ++i) {
                for (int j = 0; j < cfg.vae_layers_per_block; ++j) build_resnet("vae.encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), false);
                if (i != 3) convs["vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv"] = pack_named("vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", 3);
            }
            build_resnet("vae.encoder.mid_block.resnets.0", false);
            build_vae_attn("vae.encoder.mid_block.attentions.0");
            build_resnet("vae.encoder.mid_block.resnets.1", false);
            norms["vae.encoder.conv_norm_out"] = norm_named("vae.encoder.conv_norm_out");
            // fold quant_conv (1x1) into encoder.conv_out and keep only the mean half, times scaling_factor
            const HostTensor& wc = H("vae.encoder.conv_out.weight");
            const HostTensor& bc = H("vae.encoder.conv_out.bias");
            const HostTensor& wq = H("vae.quant_conv.weight");
            const HostTensor& bq = H("vae.quant_conv.bias");
            const int L = cfg.vae_latent_channels, M2 = 2 * L, cin = (int)wc.shape[1];
            std::vector<float> wf((size_t)L * cin * 9), bf(L);
            for (int o = 0; o < L; ++o) {
                double bb = bq.v[o];
                for (int m = 0; m < M2; ++m) bb += (double)wq.v[(size_t)o * M2 + m] * bc.v[m];
                bf[o] = (float)(bb * cfg.vae_scaling_factor);
                for (size_t e = 0; e < (size_t)cin * 9; ++e) {
                    double a = 0;
                    for (int m = 0; m < M2; ++m) a += (double)wq.v[(size_t)o * M2 + m] * wc.v[(size_t)m * cin * 9 + e];
                    wf[(size_t)o * cin * 9 + e] = (float)(a * cfg.vae_scaling_factor);
                }
            }
            convs["vae.encoder.conv_out_folded"] = pack(wf.data(), bf.data(), L, cin, 3, round_up(cin, 64));
        }
        if (has("vae.decoder.conv_in.weight")) {
            pq_w = H("vae.post_quant_conv.weight").v;
            pq_b = H("vae.post_quant_conv.bias").v;
            pq_w_dev = upload(pq_w.data(), pq_w.size());
            pq_b_dev = upload(pq_b.data(), pq_b.size());
            convs["vae.decoder.conv_in"] = pack_named("vae.decoder.conv_in", 3);
            build_resnet("vae.decoder.mid_block.resnets.0", false);
            build_vae_attn("vae.decoder.mid_block.attentions.0");
            build_resnet("vae.decoder.mid_block.resnets.1", false);
            for (int i = 0; i < 4; ++i) {
                for (int j = 0; j < cfg.vae_layers_per_block + 1; ++j) build_resnet("vae.decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), false);
                if (i != 3) {
                    const std::string un = "vae.decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
                    convs[un] = pack_named(un, 3);
                    pack_phases(convs[un], un);
                }
            }
            norms["vae.decoder.conv_norm_out"] = norm_named("vae.decoder.conv_norm_out");
            convs["vae.decoder.conv_out"] = pack_named("vae.decoder.conv_out", 3);
        }
        // ---------------- UNet ----------------
        if (have_unet) {
            te_w1 = H("unet.time_embedding.linear_1.weight").v;
            te_b1 = H("unet.time_embedding.linear_1.bias").v;
            te_w2 = H("unet.time_embedding.linear_2.weight").v;
            te_b2 = H("unet.time_embedding.linear_2.bias").v;
            convs["unet.conv_in"] = pack_named("unet.conv_in", 3);
            for (int i = 0; i < 4; ++i) {
                const std::string bp = "unet.down_blocks." + std::to_string(i);
                for (int j = 0; j < cfg.unet_layers_per_block; ++j) {
                    build_resnet(bp + ".resnets." + std::to_string(j), true);
                    if (cfg.unet_down_attn[i]) build_transformer(bp + ".attentions." + std::to_string(j), cfg.unet_num_heads[i]);
                }
                if (i != 3) convs[bp + ".downsamplers.0.conv"] = pack_named(bp + ".downsamplers.0.conv", 3);
            }
            build_resnet("unet.mid_block.resnets.0", true);
            build_transformer("unet.mid_block.attentions.0", cfg.unet_num_heads[3]);
            build_resnet("unet.mid_block.resnets.1", true);
            for (int i = 0; i < 4; ++i) {
                const std::string bp = "unet.up_blocks." + std::to_string(i);
                const bool attn = cfg.unet_down_attn[3 - i];
                for (int j = 0; j < cfg.unet_layers_per_block + 1; ++j) {
                    build_resnet(bp + ".resnets." + std::to_string(j), true);
                    if (attn) build_transformer(bp + ".attentions." + std::to_string(j), cfg.unet_num_heads[3 - i]);
                }
                if (i != 3) {
                    convs[bp + ".upsamplers.0.conv"] = pack_named(bp + ".upsamplers.0.conv", 3);
                    pack_phases(convs[bp + ".upsamplers.0.conv"], bp + ".upsamplers.0.conv");
                }
            }
            if (cfg.unet_has_out) {
                norms["unet.conv_norm_out"] = norm_named("unet.conv_norm_out");
                convs["unet.conv_out"] = pack_named("unet.conv_out", 3);
            }
            fold_timestep();
            fold_context();
        }
        // ---------------- DPT head ----------------
        if (cfg.dpt_enabled && has("dpt.neck.convs.0.weight")) {
            convs["dpt.feature_upsample_0.conv"] = pack_named("dpt.feature_upsample_0.conv", 3);
            pack_phases(convs["dpt.feature_upsample_0.conv"], "dpt.feature_upsample_0.conv");  // (x2 nearest + 3x3: four phase convolutions, conv_halo.hip PH)
            for (int i = 0; i < 4; ++i) {
                convs["dpt.neck.convs." + std::to_string(i)] = pack_named("dpt.neck.convs." + std::to_string(i), 3);
                const std::string lp = "dpt.neck.fusion_stage.layers." + std::to_string(i);
                convs[lp + ".projection"] = pack_named(lp + ".projection", 1);
                if (i != 0) {
                    convs[lp + ".residual_layer1.convolution1"] = pack_named(lp + ".residual_layer1.convolution1", 3);
                    convs[lp + ".residual_layer1.convolution2"] = pack_named(lp + ".residual_layer1.convolution2", 3);
                }
                convs[lp + ".residual_layer2.convolution1"] = pack_named(lp + ".residual_layer2.convolution1", 3);
                convs[lp + ".residual_layer2.convolution2"] = pack_named(lp + ".residual_layer2.convolution2", 3);
            }
            convs["dpt.head.projection"] = pack_named("dpt.head.projection", 3);
            convs["dpt.head.head.0"] = pack_named("dpt.head.head.0", 3);
            convs["dpt.head.head.2"] = pack_named("dpt.head.head.2", 3);
            const HostTensor& w4 = H("dpt.head.head.4.weight");
            dpt_w_dev = upload(w4.v.data(), w4.v.size());
            dpt_b = H("dpt.head.head.4.bias").v[0];
        }
        (void)have_vae;
        host.clear();
        finalized = true;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // run-time helpers (everything below only enqueues work on `st`)
    Act new_act(int B, int H, int W, int C) {
        Act a;
        a.B = B; a.H = H; a.W = W; a.C = C;
        a.p = (h16_t*)pool.alloc((size_t)B * H * W * C * sizeof(h16_t));
        return a;
    }
    void drop(Act& a) {
        pool.release(a.p);
        pool.release(a.f);
        if (a.st) pool.release(a.st);
        a.p = nullptr;
        a.f = nullptr;
        a.st = nullptr;
    }
    // ---- contract precision: stored tensors (fp32) and split operands -----------------------------------------------------------------
    Act new_act_f(int B, int H, int W, int C) {
        Act a;
        a.B = B; a.H = H; a.W = W; a.C = C;
        a.f = (float*)pool.alloc((size_t)B * H * W * C * sizeof(float));
        return a;
    }
    Act new_operand(int B, int H, int W, int C_logical) { return new_act(B, H, W, 3 * C_logical); }
    // A-order split of a stored tensor (optionally through ReLU): the operand form a conv / linear layer reads
    Act split_operand(const Act& x, int act = GP_ACT_NONE) {
        if (!x.f) throw std::logic_error("split_operand: not a stored fp32 tensor");
        Act y = new_operand(x.B, x.H, x.W, x.C);
        mark("split3 " + dims(x));
        launch_c_split3(x.f, x.C, y.p, x.pixels(), x.C, 0, act, 1.0f, st);
        return y;
    }
    // ask the kernel that is about to write `y` to leave per-tile channel statistics behind (next GroupNorm skips its read pass)
    void attach_stats(Act& y, IGemmParams& p) {
        if (!fuse_stats) return;
        int mode = 0, bm = 0;
        const int nt = igemm_tile_info(p, 0, &mode, &bm);
        if (nt <= 0) return;
        y.st = (float*)pool.alloc((size_t)nt * (p.N * 2 + 1) * sizeof(float));
        y.st_mode = mode;
        y.st_bm = bm;
        p.stats_out = y.st;
    }
    // profiling level 3: one event before every launch; a launch's cost is the time to the next mark (kernel + the gap behind it)
    struct Mark { hipEvent_t ev; std::string name; double flops; };
    std::vector<Mark> marks;
    size_t marks_used = 0;
    void mark(const std::string& name, double flops = 0.0, int n = 1) {
        tm.n_launches += n;
        if (prof < 3) return;
        if (marks_used == marks.size()) {
            Mark m;
            HIPCHK(hipEventCreate(&m.ev));
            marks.push_back(m);
        }
        marks[marks_used].name = name;
        marks[marks_used].flops = flops;
        HIPCHK(hipEventRecord(marks[marks_used].ev, st));
        ++marks_used;
    }
    static std::string dims(const Act& a) { return std::to_string(a.B) + "x" + std::to_string(a.H) + "x" + std::to_string(a.W) + "x" + std::to_string(a.C); }
    void prof_begin(int kind) {
        if (prof < 2) return;
        if (ev_used == ev_pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            ev_pool.push_back({a, b});
            ev_kind.push_back(kind);
        }
        ev_kind[ev_used] = kind;
        HIPCHK(hipEventRecord(ev_pool[ev_used].first, st));
    }
    void prof_end() {
        if (prof < 2) return;
        HIPCHK(hipEventRecord(ev_pool[ev_used].second, st));
        ++ev_used;
    }
    void run_igemm(const IGemmParams& p, int hint = 0) {
        const int taps = p.ks == 3 ? 9 : 1;
        // (contract precision: K is tripled -- hi.hi + lo.hi + hi.lo -- and the ALGORITHMIC count is a third of what the kernel executes)
        const double fl = 2.0 * (double)p.M * (double)p.N * (double)p.Cin * taps * (p.batch > 0 ? p.batch : 1) / (contract ? 3.0 : 1.0);
        const bool halo = conv_uses_halo(p, hint);
        tm.flops_igemm += fl;
        tm.n_igemm++;
        if (halo) { tm.flops_halo += fl; tm.n_halo++; flops_halo_exec += (conv_halo_uses_phases(p) ? fl * (4.0 / 9.0) : fl) * (contract ? 3.0 : 1.0); }  // (executed: the tripled K counts)
        if (prof >= 3) {
            const char* path = conv_uses_halo(p, hint) ? (p.in_scale ? (p.in_silu ? "halo+gn+silu" : "halo+gn") : "halo") : igemm_uses_pgemm(p, hint) ? "pgemm" : "igemm";
            mark(std::string(p.ks == 3 ? (p.ups ? "conv3x3up " : (p.stride == 2 ? "conv3x3s2 " : "conv3x3 ")) : (p.batch > 1 ? "bgemm " : "gemm ")) + path + " M=" + std::to_string(p.M) +
                     " N=" + std::to_string(p.N) + " K=" + std::to_string(p.Cin * taps) + (p.batch > 1 ? " batch=" + std::to_string(p.batch) : "") + (p.res ? " +res" : "") +
                     (p.act == GP_ACT_GEGLU ? " geglu" : "") + (p.stats_out ? " +stats" : "") + (p.vt_out ? " q|k|vT" : ""),
                 fl);
        } else {
            tm.n_launches++;
        }
        IGemmParams q = p;
        void* ws = nullptr;
        const int S = halo ? 1 : igemm_ksplit(q, hint);
        if (S > 1) {  // split-K partial sums: from this engine's pool (stream-ordered reuse), never process-global
            q.splitk_ws_floats = (long long)S * q.M * q.n_store;
            ws = pool.alloc((size_t)q.splitk_ws_floats * sizeof(float));
            q.splitk_ws = (float*)ws;
        }
        prof_begin(halo ? 2 : 0);
        launch_igemm(q, hint, st);
        prof_end();
        if (ws) pool.release(ws);
    }

    struct ConvOpt {
        int stride = 1, pad_t = 1, pad_l = 1;
        int Ho = 0, Wo = 0;     // 0: same as input (or upsampled size)
        int ups_h = 0, ups_w = 0;
        const h16_t* res = nullptr;
        const float* res_f = nullptr;  // contract precision: the residual is a stored fp32 tensor
        int act = GP_ACT_NONE;
        int n_store = 0;        // 0: cout
        bool want_stats = false;  // the output feeds a GroupNorm
    };
    IGemmParams conv_params(const Act& x, const PackedW& w, const ConvOpt& o, h16_t* out) {
        if (x.C != w.cin_pad) throw std::logic_error("conv: channel mismatch (" + std::to_string(x.C) + " vs " + std::to_string(w.cin_pad) + ")");
        const int Hin = o.ups_h ? o.ups_h : x.H, Win = o.ups_w ? o.ups_w : x.W;
        const int Ho = o.Ho ? o.Ho : Hin, Wo = o.Wo ? o.Wo : Win;
        const int nst = o.n_store ? o.n_store : w.cout;
        IGemmParams p{};
        p.in = x.p; p.wt = w.w; p.bias = w.bias; p.res = o.res; p.out = out; p.zero = zero;
        p.M = x.B * Ho * Wo; p.N = w.cout; p.Cin = w.cin_pad; p.n_rows = w.n_rows; p.ks = w.ks;
        p.B = x.B; p.Hi = x.H; p.Wi = x.W; p.Ho = Ho; p.Wo = Wo;
        p.stride = o.stride; p.pad_t = w.ks == 3 ? o.pad_t : 0; p.pad_l = w.ks == 3 ? o.pad_l : 0;
        p.ups = o.ups_h ? 1 : 0; p.Hu = o.ups_h; p.Wu = o.ups_w;
        p.wt_ph = o.ups_h ? w.w_ph : nullptr;  // (used where the size is exactly x2: conv_halo_uses_phases)
        p.lda = x.C; p.ldo = nst; p.ldres = nst; p.ldw = (w.ks == 3 ? 9 : 1) * w.cin_pad;
        p.n_store = nst; p.out_fp32 = 0; p.act = o.act; p.bias_mode = w.bias ? GP_BIAS_COL : GP_BIAS_NONE;
        p.batch = 1;
        return p;
    }
    // contract precision: split operand in (made here when x is a stored tensor), fp32 rows out, fp32 residual
    Act conv_c(const Act& x0, const PackedW& w, const ConvOpt& o) {
        Act xs = x0;
        const bool tmp = x0.p == nullptr;
        if (tmp) xs = split_operand(x0);
        IGemmParams p = conv_params(xs, w, o, nullptr);
        Act y = new_act_f(xs.B, p.Ho, p.Wo, p.n_store);
        p.out = y.f; p.out_fp32 = 1;
        p.res = (const h16_t*)o.res_f; p.res_f32 = o.res_f ? 1 : 0;
        if (o.want_stats) attach_stats(y, p);
        run_igemm(p);
        if (tmp) drop(xs);
        return y;
    }
    Act conv(const Act& x, const PackedW& w, const ConvOpt& o, const float* in_scale = nullptr, const float* in_shift = nullptr, bool in_silu = false) {
        if (contract) return conv_c(x, w, o);
        IGemmParams p = conv_params(x, w, o, nullptr);
        Act y = new_act(x.B, p.Ho, p.Wo, p.n_store);
        p.out = y.p;
        p.in_scale = in_scale; p.in_shift = in_shift; p.in_silu = in_silu ? 1 : 0;
        if (in_scale && !conv_uses_halo(p, 0)) throw std::logic_error("fused GroupNorm input needs the halo conv kernel");
        if (o.want_stats) attach_stats(y, p);
        run_igemm(p);
        return y;
    }
    // y[M][N] = x[M][K] W^T (+bias) (+res), N = w.cout (GEGLU halves it)
    // contract precision: y fp32 [M][N] = split(x) W^T (+bias) (+res fp32); out_inplace: write into that fp32 buffer (it may be `res`)
    Act linear_c(const Act& x0, const PackedW& w, const float* res = nullptr, int act = GP_ACT_NONE, float* out_inplace = nullptr, bool want_stats = false) {
        Act xs = x0;
        const bool tmp = x0.p == nullptr;
        if (tmp) xs = split_operand(x0);
        if (xs.C != w.cin_pad) throw std::logic_error("linear: channel mismatch");
        const int nout = act == GP_ACT_GEGLU ? w.cout / 2 : w.cout;
        Act y;
        if (out_inplace) { y.B = xs.B; y.H = xs.H; y.W = xs.W; y.C = nout; y.f = out_inplace; }
        else y = new_act_f(xs.B, xs.H, xs.W, nout);
        IGemmParams p{};
        p.in = xs.p; p.wt = w.w; p.bias = w.bias; p.res = (const h16_t*)res; p.res_f32 = res ? 1 : 0; p.out = y.f; p.out_fp32 = 1; p.zero = zero;
        p.M = (int)xs.pixels(); p.N = w.cout; p.Cin = w.cin_pad; p.n_rows = w.n_rows; p.ks = 1;
        p.B = xs.B; p.Hi = xs.H; p.Wi = xs.W; p.Ho = xs.H; p.Wo = xs.W; p.stride = 1;
        p.lda = xs.C; p.ldo = nout; p.ldres = nout; p.ldw = w.cin_pad; p.n_store = nout; p.act = act;
        p.bias_mode = w.bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        if (want_stats && !out_inplace) attach_stats(y, p);
        run_igemm(p);
        if (tmp) drop(xs);
        return y;
    }
    Act linear(const Act& x, const PackedW& w, const h16_t* res = nullptr, int act = GP_ACT_NONE, h16_t* out_inplace = nullptr,
               bool want_stats = false) {
        if (contract) {
            if (res || out_inplace) throw std::logic_error("linear: 16-bit residual in contract precision");
            return linear_c(x, w, nullptr, act, nullptr, want_stats);
        }
        if (x.C != w.cin_pad) throw std::logic_error("linear: channel mismatch");
        const int nout = act == GP_ACT_GEGLU ? w.cout / 2 : w.cout;
        Act y;
        if (out_inplace) { y = x; y.C = nout; y.p = out_inplace; }
        else y = new_act(x.B, x.H, x.W, nout);
        IGemmParams p{};
        p.in = x.p; p.wt = w.w; p.bias = w.bias; p.res = res; p.out = y.p; p.zero = zero;
        p.M = (int)x.pixels(); p.N = w.cout; p.Cin = w.cin_pad; p.n_rows = w.n_rows; p.ks = 1;
        p.B = x.B; p.Hi = x.H; p.Wi = x.W; p.Ho = x.H; p.Wo = x.W; p.stride = 1;
        p.lda = x.C; p.ldo = nout; p.ldres = nout; p.ldw = w.cin_pad; p.n_store = nout; p.act = act;
        p.bias_mode = w.bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        if (want_stats && !out_inplace) attach_stats(y, p);
        run_igemm(p);
        return y;
    }
    // V^T[b][c][t] = sum_k Wv[c][k] x[b][t][k] (+ bias[c]); zero-filled up to Tpad
    h16_t* v_transposed(const Act& x, const PackedW& wv, int T, int Tpad) {
        const int C = wv.cout;
        h16_t* vt = (h16_t*)pool.alloc((size_t)x.B * C * Tpad * sizeof(h16_t));
        IGemmParams p{};
        p.in = wv.w; p.wt = x.p; p.bias = wv.bias; p.out = vt; p.zero = zero;
        p.M = C; p.N = T; p.Cin = wv.cin_pad; p.n_rows = T; p.ks = 1; p.stride = 1;
        p.lda = wv.cin_pad; p.ldw = x.C; p.ldo = Tpad; p.n_store = Tpad;
        p.bias_mode = wv.bias ? GP_BIAS_ROW : GP_BIAS_NONE;
        p.batch = x.B; p.in_bs = 0; p.wt_bs = (long long)T * x.C; p.out_bs = (long long)C * Tpad; p.bias_bs = 0;
        // rows = channels (320 ... 1280), columns = tokens: 64x64 tiles when there are few tokens or the channel count is not a multiple of 128
        // (320 rows leave a 128-row tile half empty), 128x64 tiles otherwise -- measured per shape with GENPERCEPT_VT_TILE (igemm.hip: tile_hint),
        // 0.77 -> 0.67 ms per pass over the 18 projections
        const int vt_tile = gp_sw().vt_tile;
        run_igemm(p, vt_tile >= 0 ? vt_tile : (T <= 1024 || (C % 128)) ? 2 : 6);
        return vt;
    }
    // statistics -> per-(image, channel) scale / shift in the GN workspace; from the producer's tile partials when it left some
    void gn_scale_shift(const Act& x, const NormW& n, float eps, float*& scale, float*& shift) {
        if (x.C != n.C) throw std::logic_error("groupnorm: channel mismatch");
        float* ws = gn_workspace(x);
        scale = ws + groupnorm_ws_floats(x.B, x.H * x.W, x.C, cfg.norm_groups);
        shift = scale + (size_t)x.B * x.C;
        if (contract && !x.st) {  // statistics pass over the fp32 tensor: per-row partials in the layout the tile finaliser reads ("mode 2")
            const int R = c_gn_stat_rows(x.H * x.W, x.C, nullptr);
            float* part = (float*)pool.alloc((size_t)x.B * R * (2 * x.C + 1) * sizeof(float));
            mark("c_gn_stats+finalize " + dims(x), 0.0, 2);
            launch_c_gn_stats(x.f, part, x.B, x.H * x.W, x.C, st);
            launch_groupnorm_from_partials(part, 2, R, x.B, x.H, x.W, x.C, cfg.norm_groups, eps, n.g, n.b, scale, shift, st);
            pool.release(part);
            return;
        }
        if (x.st) {
            mark("gn_finalize_tiles " + dims(x));
            launch_groupnorm_from_partials(x.st, x.st_mode, x.st_bm, x.B, x.H, x.W, x.C, cfg.norm_groups, eps, n.g, n.b, scale, shift, st);
        } else {
            mark("gn_stats+finalize " + dims(x), 0.0, 2);
            launch_groupnorm_stats(x.p, n.g, n.b, x.B, x.H * x.W, x.C, cfg.norm_groups, eps, ws, scale, shift, st);
        }
    }
    // one-launch GroupNorm for small maps whose producer left no statistics behind (split-K convs of the 12x12 level, ...)
    bool gn_small(const Act& x) const { return !contract && !x.st && groupnorm_small_applicable(x.B, x.H * x.W, x.C, cfg.norm_groups) && !gp_sw().no_gn_small; }
    Act groupnorm_small(const Act& x, const NormW& n, float eps, bool silu) {
        if (x.C != n.C) throw std::logic_error("groupnorm: channel mismatch");
        Act y = new_act(x.B, x.H, x.W, x.C);
        mark("gn_small " + dims(x));
        launch_groupnorm_small(x.p, y.p, n.g, n.b, x.B, x.H * x.W, x.C, cfg.norm_groups, eps, silu ? 1 : 0, st);
        return y;
    }
    Act groupnorm(const Act& x, const NormW& n, float eps, bool silu) {
        if (gn_small(x)) return groupnorm_small(x, n, eps, silu);
        float *scale, *shift;
        gn_scale_shift(x, n, eps, scale, shift);
        if (contract) {  // the normalised tensor only ever feeds a matrix product: written as its split operand
            Act y = new_operand(x.B, x.H, x.W, x.C);
            mark("c_gn_apply_split " + dims(x));
            launch_c_gn_apply_split(x.f, y.p, scale, shift, x.B, x.H * x.W, x.C, silu ? 1 : 0, st);
            return y;
        }
        Act y = new_act(x.B, x.H, x.W, x.C);
        mark("gn_apply " + dims(x));
        launch_groupnorm_apply(x.p, y.p, scale, shift, x.B, x.H * x.W, x.C, silu ? 1 : 0, st);
        return y;
    }
    float* gn_workspace(const Act& x) {  // partial statistics + per-(image, channel) scale / shift
        const size_t need = (size_t)groupnorm_ws_floats(x.B, x.H * x.W, x.C, cfg.norm_groups) + 2 * (size_t)x.B * x.C;
        if (need > gn_ws_floats) {
            if (gn_ws) { pool.live_[gn_ws] = 1; pool.release(gn_ws); }
            gn_ws = (float*)pool.alloc(need * 4 * 2);
            pool.live_.erase(gn_ws);  // kept across calls: not part of a call's outstanding set
            gn_ws_floats = need * 2;
        }
        return gn_ws;
    }
    // conv(act(GroupNorm(x))): statistics pass, then the normalisation is applied either inside the conv kernel on the staged
    // input halo (conv_halo.hip) or, when that kernel does not take the layer, by the separate apply pass.
    Act conv_gn(const Act& x, const NormW& n, float eps, bool silu, const PackedW& w, const ConvOpt& o) {
        if (contract) {
            Act y = groupnorm(x, n, eps, silu);
            Act out = conv(y, w, o);
            drop(y);
            return out;
        }
        if (gn_small(x)) {
            IGemmParams p0 = conv_params(x, w, o, nullptr);
            if (!conv_uses_halo(p0, 0)) {  // (a halo conv would fuse the apply: keep the statistics path for it)
                Act y = groupnorm_small(x, n, eps, silu);
                Act out = conv(y, w, o);
                drop(y);
                return out;
            }
        }
        float *scale, *shift;
        gn_scale_shift(x, n, eps, scale, shift);
        IGemmParams p = conv_params(x, w, o, nullptr);
        p.in_scale = scale; p.in_shift = shift; p.in_silu = silu ? 1 : 0;
        // The fused transform is redone by every 128-channel slice of the output (each slice's workgroup stages its own halo) and it is
        // bound by the two quarter-rate transcendentals per element inside the conv.  One separate apply pass -- 2 x tensor bytes of HBM
        // traffic -- is cheaper than that redundant arithmetic as soon as there are two slices (measured r1: decoder -0.6 ms, encoder
        // -0.4 ms on the large VAE maps; r2: also on the 96x96 .. 24x24 maps, another -1.3 ms per pass), so only single-slice convs fuse.
        const int slices = (w.cout + 127) / 128;
        const bool fuse_here = slices <= gn_fuse_max_slices || x.H * x.W < gn_fuse_always_below_px;
        if (fuse_gn && fuse_here && conv_uses_halo(p, 0)) return conv(x, w, o, scale, shift, silu);
        Act y = new_act(x.B, x.H, x.W, x.C);
        mark("gn_apply " + dims(x));
        launch_groupnorm_apply(x.p, y.p, scale, shift, x.B, x.H * x.W, x.C, silu ? 1 : 0, st);
        Act out = conv(y, w, o);
        drop(y);
        return out;
    }
    Act layernorm(const Act& x, const NormW& n) {
        if (contract) {
            Act y = new_operand(x.B, x.H, x.W, x.C);
            mark("c_layernorm_split " + dims(x));
            launch_c_layernorm_split(x.f, y.p, n.g, n.b, (int)x.pixels(), x.C, 1e-5f, st);
            return y;
        }
        Act y = new_act(x.B, x.H, x.W, x.C);
        mark("layernorm " + dims(x));
        launch_layernorm(x.p, y.p, n.g, n.b, (int)x.pixels(), x.C, 1e-5f, st);
        return y;
    }

    Act resnet(const Act& x, const std::string& name, float eps) {
        const ResW& r = resnets.at(name);
        ConvOpt o1;
        o1.want_stats = true;
        Act h = conv_gn(x, r.n1, eps, true, r.c1, o1);
        Act sc = x;
        if (r.has_sc) sc = linear(x, r.sc);
        ConvOpt o;
        o.res = sc.p;
        o.res_f = sc.f;
        o.want_stats = true;
        Act y = conv_gn(h, r.n2, eps, true, r.c2, o);
        drop(h);
        if (r.has_sc) drop(sc);
        return y;
    }

    // GEMM -> scores in HBM -> row softmax -> GEMM (head dims other than 512, and the A/B switch GENPERCEPT_NO_FLASH512)
    void vae_attention_unfused(const Act& qk, const h16_t* vt, Act& o, int B, int T, int Tpad, int C) {
        // logits as fp16 (11 significant bits: finer than the bf16 probabilities they turn into) halve the score traffic, the largest HBM
        // item of this path.  They are the SCALED logits (1/sqrt(C) is folded into the query projection, build_vae_attn) and the fp16
        // conversion saturates at +-65504 (epilogue.h), so an outlier row degrades to a one-hot softmax instead of inf - inf = NaN.
        // GENPERCEPT_FP32_SCORES=1 keeps fp32 (A/B)
        const bool f32_scores = gp_sw().fp32_scores;
        const bool half_scores = !f32_scores && softmax_rows_f16_supported(Tpad);
        float* S = (float*)pool.alloc((size_t)B * T * Tpad * (half_scores ? 2 : 4));
        {
            IGemmParams p{};
            p.in = qk.p; p.wt = qk.p + C; p.out = S; p.zero = zero;
            p.M = T; p.N = T; p.Cin = C; p.n_rows = T; p.ks = 1; p.stride = 1;
            p.lda = 2 * C; p.ldw = 2 * C; p.ldo = Tpad; p.n_store = T; p.out_fp32 = half_scores ? 2 : 1;
            p.batch = B; p.in_bs = (long long)T * 2 * C; p.wt_bs = (long long)T * 2 * C; p.out_bs = (long long)T * Tpad;
            run_igemm(p);
        }
        h16_t* P = (h16_t*)pool.alloc((size_t)B * T * Tpad * sizeof(h16_t));
        mark("softmax_rows T=" + std::to_string(T));
        if (half_scores) launch_softmax_rows_f16(S, P, B * T, T, Tpad, 1.0f, st);
        else launch_softmax_rows(S, P, B * T, T, Tpad, 1.0f, st);
        pool.release(S);
        {
            IGemmParams p{};
            p.in = P; p.wt = vt; p.out = o.p; p.zero = zero;
            p.M = T; p.N = C; p.Cin = Tpad; p.n_rows = C; p.ks = 1; p.stride = 1;
            p.lda = Tpad; p.ldw = Tpad; p.ldo = C; p.n_store = C;
            p.batch = B; p.in_bs = (long long)T * Tpad; p.wt_bs = (long long)C * Tpad; p.out_bs = (long long)T * C;
            run_igemm(p);
        }
        pool.release(P);
    }

    // contract precision: softmax(scale q k^T) v per head with fp32 logits in HBM (head split -> batched logits GEMM -> row softmax -> batched
    // P.V GEMM -> head merge; every product over split operands).  qkv: stored [B*T][3C] with q | k | v at columns 0 | C | 2C; returns the
    // A-order operand [B*T][3C] the output projection reads.
    Act attention_c(const Act& qkv, int heads, int hd, float scale) {
        const int B = qkv.B, T = qkv.H * qkv.W, Tpad = round_up(T, 64), Z = B * heads;
        if (hd % 64 || qkv.C != 3 * heads * hd) throw std::logic_error("attention_c: layout");
        if (hd == 64 && !gp_sw().c_no_flash && (long long)T * qkv.C * 2 < 0x7fffffffll) {  // flash attention over split operands (attention.hip)
            const int C = heads * hd;
            h16_t* qk_hi = (h16_t*)pool.alloc((size_t)B * T * 2 * C * sizeof(h16_t));
            h16_t* qk_lo = (h16_t*)pool.alloc((size_t)B * T * 2 * C * sizeof(h16_t));
            h16_t* vt_hi = (h16_t*)pool.alloc((size_t)Z * hd * Tpad * sizeof(h16_t));
            h16_t* vt_lo = (h16_t*)pool.alloc((size_t)Z * hd * Tpad * sizeof(h16_t));
            mark("c_qkv_planes T=" + std::to_string(T) + " heads=" + std::to_string(heads), 0.0, 2);
            launch_c_qkv_planes(qkv.f, qkv.C, qk_hi, qk_lo, vt_hi, vt_lo, B, T, Tpad, heads, hd, st);
            Act a = new_operand(qkv.B, qkv.H, qkv.W, C);
            const double fl = 4.0 * Z * (double)T * T * hd;
            tm.flops_attn += fl;
            tm.n_attn++;
            mark("flash_attn64_split T=" + std::to_string(T) + " heads=" + std::to_string(heads), fl);
            prof_begin(1);
            launch_flash_attn64_split(qk_hi, qk_lo, vt_hi, vt_lo, a.p, B, T, heads, 2 * C, Tpad, st);
            prof_end();
            pool.release(qk_hi); pool.release(qk_lo); pool.release(vt_hi); pool.release(vt_lo);
            return a;
        }
        if (!c_softmax_split_supported(Tpad)) throw std::logic_error("attention_c: Tpad");
        h16_t* Qs = (h16_t*)pool.alloc((size_t)Z * T * 3 * hd * sizeof(h16_t));
        h16_t* Ks = (h16_t*)pool.alloc((size_t)Z * T * 3 * hd * sizeof(h16_t));
        h16_t* Vts = (h16_t*)pool.alloc((size_t)Z * hd * 3 * Tpad * sizeof(h16_t));
        mark("c_heads_split T=" + std::to_string(T) + " heads=" + std::to_string(heads), 0.0, 2);
        launch_c_heads_split(qkv.f, qkv.C, Qs, Ks, Vts, B, T, Tpad, heads, hd, st);
        float* S = (float*)pool.alloc((size_t)Z * T * Tpad * sizeof(float));
        {
            IGemmParams p{};
            p.in = Qs; p.wt = Ks; p.out = S; p.zero = zero;
            p.M = T; p.N = T; p.Cin = 3 * hd; p.n_rows = T; p.ks = 1; p.stride = 1;
            p.lda = 3 * hd; p.ldw = 3 * hd; p.ldo = Tpad; p.n_store = T; p.out_fp32 = 1;
            p.batch = Z; p.in_bs = (long long)T * 3 * hd; p.wt_bs = (long long)T * 3 * hd; p.out_bs = (long long)T * Tpad;
            run_igemm(p);
        }
        pool.release(Qs);
        pool.release(Ks);
        h16_t* P = (h16_t*)pool.alloc((size_t)Z * T * 3 * Tpad * sizeof(h16_t));
        mark("c_softmax_split T=" + std::to_string(T));
        launch_c_softmax_split(S, P, (long long)Z * T, T, Tpad, scale, st);
        pool.release(S);
        float* O = (float*)pool.alloc((size_t)Z * T * hd * sizeof(float));
        {
            IGemmParams p{};
            p.in = P; p.wt = Vts; p.out = O; p.zero = zero;
            p.M = T; p.N = hd; p.Cin = 3 * Tpad; p.n_rows = hd; p.ks = 1; p.stride = 1;
            p.lda = 3 * Tpad; p.ldw = 3 * Tpad; p.ldo = hd; p.n_store = hd; p.out_fp32 = 1;
            p.batch = Z; p.in_bs = (long long)T * 3 * Tpad; p.wt_bs = (long long)hd * 3 * Tpad; p.out_bs = (long long)T * hd;
            run_igemm(p);
        }
        pool.release(P);
        pool.release(Vts);
        Act a = new_operand(qkv.B, qkv.H, qkv.W, heads * hd);
        mark("c_heads_merge_split " + dims(a));
        launch_c_heads_merge_split(O, a.p, B, T, heads, hd, st);
        pool.release(O);
        tm.flops_attn += 4.0 * Z * (double)T * T * hd;
        tm.n_attn++;
        return a;
    }
    Act vae_attention_c(const Act& x, const VaeAttnW& a) {
        Act n = groupnorm(x, a.gn, cfg.vae_norm_eps, false);
        Act qkv = linear_c(n, a.qkv);
        drop(n);
        Act o = attention_c(qkv, 1, a.C, 1.0f / std::sqrt((float)a.C));
        drop(qkv);
        Act y = linear_c(o, a.o, x.f, GP_ACT_NONE, nullptr, true);
        drop(o);
        return y;
    }
    // BasicTransformerBlock inside Transformer2DModel (custom_unet.py call sites :305-327,341-352), contract precision
    Act transformer_c(const Act& x, const TfW& t) {
        const int C = t.C;
        Act n = groupnorm(x, t.gn, 1e-6f, false);
        Act y = linear_c(n, t.proj_in);
        drop(n);
        Act l1 = layernorm(y, t.ln1);
        Act qkv = linear_c(l1, t.qkv);
        drop(l1);
        Act a = attention_c(qkv, t.heads, 64, 0.125f);
        drop(qkv);
        linear_c(a, t.o1, y.f, GP_ACT_NONE, y.f);  // y += to_out(attn), in place
        drop(a);
        Act l3;
        if (t.fU && c_cross_fold_supported(C)) {
            l3 = new_operand(x.B, x.H, x.W, C);
            mark("c_cross_fold " + dims(y));
            launch_c_cross_fold(y.f, y.f, l3.p, t.fU, t.fu0, t.fG, t.fc0, t.ln3.g, t.ln3.b, (int)x.pixels(), C, t.heads, 1e-5f, st);
        } else {
            Act l2 = layernorm(y, t.ln2);
            Act q2 = linear_c(l2, t.q2);
            drop(l2);
            Act a2 = new_operand(x.B, x.H, x.W, C);
            mark("c_cross_attn_small " + dims(q2));
            launch_c_cross_attn_small(q2.f, t.kc, t.vc, a2.p, (int)x.pixels(), C, ctx_L, st);
            drop(q2);
            linear_c(a2, t.o2, y.f, GP_ACT_NONE, y.f);
            drop(a2);
            l3 = layernorm(y, t.ln3);
        }
        Act ff = linear_c(l3, t.ff1, nullptr, GP_ACT_GEGLU);
        drop(l3);
        linear_c(ff, t.ff2, y.f, GP_ACT_NONE, y.f);
        drop(ff);
        Act out = linear_c(y, t.proj_out, x.f, GP_ACT_NONE, nullptr, true);
        drop(y);
        return out;
    }

    Act vae_attention(const Act& x, const std::string& name) {
        const VaeAttnW& a = vattn.at(name);
        if (contract) return vae_attention_c(x, a);
        const int T = x.H * x.W, C = a.C, Tpad = round_up(T, 64), B = x.B;
        Act n = groupnorm(x, a.gn, cfg.vae_norm_eps, false);
        Act qk = linear(n, a.qk);  // [B*T][2C]
        h16_t* vt = v_transposed(n, a.v, T, Tpad);
        drop(n);
        Act o = new_act(x.B, x.H, x.W, C);
        if (flash_attn512_supported(C)) {
            // fused: scores and probabilities never leave the CU (attention.hip: flash_attn512_kernel).  The softmax scale 1/sqrt(C) is
            // folded into the query projection (build_vae_attn), logits exist in fp32 registers only.
            const long long wsf = flash_attn512_workspace_floats(B, T, ncu);
            float* ws = wsf ? (float*)pool.alloc((size_t)wsf * sizeof(float)) : nullptr;
            const double fl = 4.0 * B * (double)T * T * C;
            tm.flops_attn += fl;
            tm.n_attn++;
            mark("flash_attn512 T=" + std::to_string(T), fl);
            prof_begin(1);
            launch_flash_attn512(qk.p, qk.p + C, vt, o.p, ws, B, T, 2 * C, 2 * C, Tpad, C, 1.0f, ncu, st);
            prof_end();
            if (ws) pool.release(ws);
            drop(qk);
            pool.release(vt);
        } else {
            vae_attention_unfused(qk, vt, o, B, T, Tpad, C);
            drop(qk);
            pool.release(vt);
        }
        Act y = linear(o, a.o, x.p, GP_ACT_NONE, nullptr, true);
        drop(o);
        return y;
    }

    Act transformer(const Act& x, const std::string& name) {
        const TfW& t = tfs.at(name);
        if (!t.kc) throw std::logic_error("gp_set_context has not been called");
        if (contract) return transformer_c(x, t);
        const int T = x.H * x.W, C = t.C, Tpad = round_up(T, 64);
        Act n = groupnorm(x, t.gn, 1e-6f, false);
        Act y = linear(n, t.proj_in);
        drop(n);
        // self-attention
        Act l1 = layernorm(y, t.ln1);
        Act qk;
        h16_t* vt = nullptr;
        {   // q | k | V^T in ONE launch when the persistent GEMM takes it (T % 16 == 0 ...), else the q | k GEMM + the transposed V GEMM
            const bool no_fuse = gp_sw().no_qkv_fuse;  // A/B switch
            IGemmParams p{};
            p.in = l1.p; p.wt = t.qkv.w; p.zero = zero;
            p.M = (int)l1.pixels(); p.N = 3 * C; p.Cin = t.qkv.cin_pad; p.n_rows = t.qkv.n_rows; p.ks = 1;
            p.B = x.B; p.Hi = x.H; p.Wi = x.W; p.Ho = x.H; p.Wo = x.W; p.stride = 1;
            p.lda = l1.C; p.ldo = 2 * C; p.ldres = 2 * C; p.ldw = t.qkv.cin_pad; p.n_store = 2 * C; p.act = GP_ACT_NONE;
            p.bias_mode = GP_BIAS_NONE; p.batch = 1;
            p.vt_col0 = 2 * C; p.vt_T = T; p.vt_Tpad = Tpad;
            p.vt_out = (h16_t*)zero;  // (non-null for the applicability test)
            // measured (tools/kbench, weight-cold): 4 x 2304 tokens, C = 640: 43 us fused vs 25 + 25; 4 x 576, C = 1280: 46 vs 29 + 25; at
            // 4 x 9216 tokens, C = 320 the transposed stores of the V third (16 bytes per channel row and lane) eat the saving: 55 vs 29 + 25
            // (same-box pipeline A/B, tools/gpu_ab_r03.sh: fusing every level is 0.1-0.2 ms per pass ahead of fusing none, the 4 x 9216 level
            // included: one launch and one pass over the LayerNorm output less outweigh the slower V slices)
            const int fuse_max_rows = gp_sw().qkv_fuse_max_rows;
            if (!no_fuse && p.M <= fuse_max_rows && !t.qkv.bias && l1.C == t.qkv.cin_pad && igemm_uses_pgemm(p, 0)) {
                qk = new_act(x.B, x.H, x.W, 2 * C);
                vt = (h16_t*)pool.alloc((size_t)x.B * C * Tpad * sizeof(h16_t));
                if (Tpad != T) HIPCHK(hipMemsetAsync(vt, 0, (size_t)x.B * C * Tpad * sizeof(h16_t), st));  // keys beyond T must read as zero
                p.out = qk.p; p.vt_out = vt;
                run_igemm(p);
            } else {
                qk = linear(l1, t.qk);
                vt = v_transposed(l1, t.v, T, Tpad);
            }
        }
        drop(l1);
        Act a = new_act(x.B, x.H, x.W, C);
        tm.flops_attn += 4.0 * x.B * t.heads * (double)T * T * 64;
        tm.n_attn++;
        mark("flash_attn64 T=" + std::to_string(T) + " heads=" + std::to_string(t.heads), 4.0 * x.B * t.heads * (double)T * T * 64);
        prof_begin(1);
        launch_flash_attn64(qk.p, qk.p + C, vt, a.p, x.B, T, t.heads, 2 * C, 2 * C, Tpad, C, st);
        prof_end();
        drop(qk);
        pool.release(vt);
        linear(a, t.o1, y.p, GP_ACT_NONE, y.p);  // y += to_out(attn), in place
        drop(a);
        // cross-attention against the folded constant context
        Act l3;
        if (t.fU) {  // two context tokens: LayerNorm + to_q + attention + to_out + residual + the feed-forward's LayerNorm in ONE pass
            l3 = new_act(x.B, x.H, x.W, C);
            mark("cross_attn_fold " + dims(y));
            launch_cross_attn_fold(y.p, y.p, l3.p, t.fU, t.fu0, t.fG, t.fc0, t.ln3.g, t.ln3.b, (int)x.pixels(), C, t.heads, 1e-5f, st);
        } else {
            Act l2 = layernorm(y, t.ln2);
            Act q2 = linear(l2, t.q2);
            drop(l2);
            Act a2 = new_act(x.B, x.H, x.W, C);
            mark("cross_attn_small " + dims(q2));
            launch_cross_attn_small(q2.p, t.kc, t.vc, a2.p, (int)x.pixels(), C, ctx_L, st);
            drop(q2);
            linear(a2, t.o2, y.p, GP_ACT_NONE, y.p);
            drop(a2);
            l3 = layernorm(y, t.ln3);
        }
        // GEGLU feed-forward
        Act ff = linear(l3, t.ff1, nullptr, GP_ACT_GEGLU);
        drop(l3);
        linear(ff, t.ff2, y.p, GP_ACT_NONE, y.p);
        drop(ff);
        Act out = linear(y, t.proj_out, x.p, GP_ACT_NONE, nullptr, true);
        drop(y);
        return out;
    }

    // ---- stages ---------------------------------------------------------------------------------------------------
    // rgb (device NCHW) -> latent NHWC, 64 allocated channels (cfg.vae_latent_channels real, scaled by scaling_factor)
    Act vae_encode(const void* rgb, int is_u8, int B, int Hh, int Ww) {
        const PackedW& win = convs.at("vae.encoder.conv_in");
        Act h;
        if (contract) {  // x / 255 * 2 - 1 is not a bf16 number: the image itself enters as a split operand
            Act xs = new_operand(B, Hh, Ww, 64);
            mark("c_rgb_split");
            launch_c_rgb_split(rgb, is_u8, xs.p, B, Hh, Ww, st);
            ConvOpt oin;
            oin.want_stats = true;
            h = conv(xs, win, oin);
            drop(xs);
        } else if (win.cout % 32 == 0 && win.cin_pad == 64 && !gp_sw().no_rgb_conv) {
            // u8 image -> conv_in output in one kernel (K = 27), statistics for the first resnet's norm1 included
            h = new_act(B, Hh, Ww, win.cout);
            if (fuse_stats) {
                const int rows = rgb_conv_in_rows(B, Hh, Ww);  // persistent conv_in: one row per workgroup + its pixel count ("mode 2")
                h.st = (float*)pool.alloc((size_t)B * rows * (2 * win.cout + 1) * sizeof(float));
                h.st_mode = 2;
                h.st_bm = rows;
            }
            tm.flops_igemm += 2.0 * (double)h.pixels() * win.cout * 27.0;
            mark("rgb_conv_in " + dims(h), 2.0 * (double)h.pixels() * win.cout * 27.0);
            prof_begin(0);
            if (!conv_in_w27) {  // compact K = 27 weight matrix, built once from the packed conv weight
                HIPCHK(hipMalloc((void**)&conv_in_w27, (size_t)win.cout * 32 * sizeof(h16_t)));
                weights_dev.push_back(conv_in_w27);
                launch_pack_k27(win.w, 9 * win.cin_pad, win.cout, conv_in_w27, st);
            }
            launch_rgb_conv_in(rgb, is_u8, conv_in_w27, win.bias, h.p, h.st, B, Hh, Ww, win.cout, st);
            prof_end();
        } else {
            Act x = new_act(B, Hh, Ww, 64);
            mark("rgb_prologue");
            launch_rgb_prologue(rgb, is_u8, x.p, B, Hh, Ww, 64, st);
            ConvOpt oin;
            oin.want_stats = true;
            h = conv(x, win, oin);
            drop(x);
        }
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < cfg.vae_layers_per_block; ++j) {
                Act y = resnet(h, "vae.encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cfg.vae_norm_eps);
                drop(h);
                h = y;
            }
            if (i != 3) {
                ConvOpt o;  // pad (0,1,0,1) then stride-2 conv without padding (Appendix B.6)
                o.stride = 2; o.pad_t = 0; o.pad_l = 0; o.want_stats = true;
                o.Ho = (h.H + 1 - 3) / 2 + 1; o.Wo = (h.W + 1 - 3) / 2 + 1;
                Act y = conv(h, convs.at("vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv"), o);
                drop(h);
                h = y;
            }
        }
        Act y = resnet(h, "vae.encoder.mid_block.resnets.0", cfg.vae_norm_eps); drop(h); h = y;
        y = vae_attention(h, "vae.encoder.mid_block.attentions.0"); drop(h); h = y;
        y = resnet(h, "vae.encoder.mid_block.resnets.1", cfg.vae_norm_eps); drop(h); h = y;
        Act n = groupnorm(h, norms.at("vae.encoder.conv_norm_out"), cfg.vae_norm_eps, true);
        drop(h);
        ConvOpt o;
        o.n_store = 64;
        Act lat = conv(n, convs.at("vae.encoder.conv_out_folded"), o);
        drop(n);
        return lat;
    }

    // latent NHWC (64 allocated channels) -> sample NHWC (64 allocated, unet_out_channels real) and/or the 4 up-block features
    Act unet(const Act& latent, Act* feats /* [4] or null */, bool want_sample) {
        std::vector<Act> skips;
        const int nup = 3;
        const bool fwd_size = (latent.H % (1 << nup)) != 0 || (latent.W % (1 << nup)) != 0;
        ConvOpt oin;
        oin.want_stats = true;
        Act x = conv(latent, convs.at("unet.conv_in"), oin);
        skips.push_back(x);
        Act cur = x;  // `cur` aliases the top of the skip stack until replaced
        for (int i = 0; i < 4; ++i) {
            const std::string bp = "unet.down_blocks." + std::to_string(i);
            for (int j = 0; j < cfg.unet_layers_per_block; ++j) {
                Act y = resnet(cur, bp + ".resnets." + std::to_string(j), cfg.unet_norm_eps);
                if (cfg.unet_down_attn[i]) {
                    Act z = transformer(y, bp + ".attentions." + std::to_string(j));
                    drop(y);
                    y = z;
                }
                skips.push_back(y);
                cur = y;
            }
            if (i != 3) {
                ConvOpt o;
                o.stride = 2; o.want_stats = true;
                o.Ho = (cur.H + 2 - 3) / 2 + 1; o.Wo = (cur.W + 2 - 3) / 2 + 1;
                Act y = conv(cur, convs.at(bp + ".downsamplers.0.conv"), o);
                skips.push_back(y);
                cur = y;
            }
        }
        // mid (cur is also skips.back(): do not drop it)
        Act m = resnet(cur, "unet.mid_block.resnets.0", cfg.unet_norm_eps);
        Act m2 = transformer(m, "unet.mid_block.attentions.0");
        drop(m);
        m = resnet(m2, "unet.mid_block.resnets.1", cfg.unet_norm_eps);
        drop(m2);
        Act h = m;
        bool h_is_feat = false;  // h is one of the caller's feature maps (multi_level_feats): the caller drops it, not this function
        for (int i = 0; i < 4; ++i) {
            const std::string bp = "unet.up_blocks." + std::to_string(i);
            const bool attn = cfg.unet_down_attn[3 - i];
            const int nres = cfg.unet_layers_per_block + 1;
            for (int j = 0; j < nres; ++j) {
                Act skip = skips.back();
                skips.pop_back();
                Act cat = contract ? new_act_f(h.B, h.H, h.W, h.C + skip.C) : new_act(h.B, h.H, h.W, h.C + skip.C);
                const int cbm = (fuse_stats && !contract) ? concat_stats_bm((long long)h.H * h.W, h.pixels(), h.C + skip.C) : 0;
                mark("concat " + dims(cat));
                if (contract) {
                    launch_c_concat(h.f, h.C, skip.f, skip.C, cat.f, h.pixels(), st);
                } else if (cbm) {  // the copy also leaves the statistics the resnet's first GroupNorm needs
                    cat.st = (float*)pool.alloc((size_t)(h.pixels() / cbm) * cat.C * 2 * sizeof(float));
                    cat.st_mode = 0;
                    cat.st_bm = cbm;
                    launch_concat_stats(h.p, h.C, skip.p, skip.C, cat.p, h.pixels(), cbm, cat.st, st);
                } else {
                    launch_concat(h.p, h.C, skip.p, skip.C, cat.p, h.pixels(), st);
                }
                if (!h_is_feat) drop(h);
                else if (h.st) pool.release(h.st);
                h_is_feat = false;
                drop(skip);
                Act y = resnet(cat, bp + ".resnets." + std::to_string(j), cfg.unet_norm_eps);
                drop(cat);
                if (attn) {
                    Act z = transformer(y, bp + ".attentions." + std::to_string(j));
                    drop(y);
                    y = z;
                }
                h = y;
            }
            if (i != 3) {
                ConvOpt o;
                if (fwd_size) { o.ups_h = skips.back().H; o.ups_w = skips.back().W; }
                else { o.ups_h = h.H * 2; o.ups_w = h.W * 2; }
                Act y = conv(h, convs.at(bp + ".upsamplers.0.conv"), o);
                drop(h);
                h = y;
            }
            if (feats) {  // custom_unet.py:365,400: the output of every up block; retained, not copied (the buffer outlives its use here)
                feats[i] = h;
                feats[i].st = nullptr;  // (its statistics partials stay with h and are released below / by the next concat)
                h_is_feat = true;
            }
        }
        Act out;
        if (want_sample && cfg.unet_has_out) {
            Act n = groupnorm(h, norms.at("unet.conv_norm_out"), cfg.unet_norm_eps, true);
            ConvOpt o;
            o.n_store = 64;
            out = conv(n, convs.at("unet.conv_out"), o);
            drop(n);
        }
        if (!h_is_feat) drop(h);
        else if (h.st) { pool.release(h.st); }
        return out;
    }

    // z_in: NHWC with 64 allocated channels holding the UNet output v (in_scale = -1/scaling) or a pred latent (1/scaling)
    // out_dev != nullptr: the caller wants the fp32 NCHW result of decode_pred (+ clip / shift unless raw); when the fused tail kernel takes
    // the last three layers (conv_few.hip) it is written directly and the returned Act is empty, else the caller runs launch_decode_epilogue
    Act vae_decode(const Act& z_in, float in_scale, float* out_dev = nullptr, int mean3 = 0, int raw = 0) {
        const int L = cfg.vae_latent_channels;
        Act z = contract ? new_act_f(z_in.B, z_in.H, z_in.W, 64) : new_act(z_in.B, z_in.H, z_in.W, 64);
        mark("post_quant_conv");
        if (contract) launch_c_pointwise_small(z_in.f, z.f, pq_w_dev, pq_b_dev, z_in.pixels(), L, L, z_in.C, 64, in_scale, st);
        else launch_pointwise_small(z_in.p, z.p, pq_w_dev, pq_b_dev, z_in.pixels(), L, L, z_in.C, 64, in_scale, st);
        ConvOpt oin;
        oin.want_stats = true;
        Act h = conv(z, convs.at("vae.decoder.conv_in"), oin);
        drop(z);
        Act y = resnet(h, "vae.decoder.mid_block.resnets.0", cfg.vae_norm_eps); drop(h); h = y;
        y = vae_attention(h, "vae.decoder.mid_block.attentions.0"); drop(h); h = y;
        y = resnet(h, "vae.decoder.mid_block.resnets.1", cfg.vae_norm_eps); drop(h); h = y;
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < cfg.vae_layers_per_block + 1; ++j) {
                y = resnet(h, "vae.decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cfg.vae_norm_eps);
                drop(h);
                h = y;
            }
            if (i != 3) {
                ConvOpt o;
                o.ups_h = h.H * 2; o.ups_w = h.W * 2; o.want_stats = true;
                y = conv(h, convs.at("vae.decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv"), o);
                drop(h);
                h = y;
            }
        }
        const PackedW& wout = convs.at("vae.decoder.conv_out");
        if (!contract && out_dev && conv_few_applicable(h.C, wout.cout, h.H, h.W) && wout.cin_pad == h.C) {
            float *scale, *shift;
            gn_scale_shift(h, norms.at("vae.decoder.conv_norm_out"), cfg.vae_norm_eps, scale, shift);
            const double fl = 2.0 * (double)h.pixels() * wout.cout * 9.0 * h.C;
            tm.flops_igemm += fl;
            tm.n_igemm++;
            mark("conv_few gn+silu+conv3x3+decode " + dims(h), fl);
            prof_begin(0);
            launch_conv_few(h.p, wout.w, wout.bias, scale, shift, zero, out_dev, h.B, h.H, h.W, 1, mean3, raw, ncu, st);
            prof_end();
            drop(h);
            return Act{};
        }
        Act n = groupnorm(h, norms.at("vae.decoder.conv_norm_out"), cfg.vae_norm_eps, true);
        drop(h);
        ConvOpt o;
        o.n_store = 4;  // 3 real channels + one zero: 8-byte pixels for the epilogue
        Act out = conv(n, wout, o);
        drop(n);
        return out;
    }

    // decoder output NHWC (3 real channels) -> the caller's fp32 NCHW map: channel mean, clip / shift unless raw (genpercept_pipeline.py:523-525,469-472)
    void decode_epilogue(const Act& dec, float* out, int mean3, int raw) {
        mark("decode_epilogue");
        if (dec.f) launch_c_decode_epilogue(dec.f, out, dec.B, dec.H, dec.W, dec.C, mean3, raw, st);
        else launch_decode_epilogue(dec.p, out, dec.B, dec.H, dec.W, dec.C, mean3, raw, st);
    }

    Act bilinear(const Act& x, int Ho, int Wo, int align_corners) {
        Act y = contract ? new_act_f(x.B, Ho, Wo, x.C) : new_act(x.B, Ho, Wo, x.C);
        mark("bilinear " + dims(y));
        if (contract) launch_c_bilinear(x.f, y.f, x.B, x.H, x.W, Ho, Wo, x.C, align_corners, st);
        else launch_bilinear(x.p, y.p, x.B, x.H, x.W, Ho, Wo, x.C, align_corners, st);
        return y;
    }
    Act rcu(const Act& x, const std::string& p) {  // pre-activation residual unit (dpt_head.py:256-271)
        Act r;
        if (contract) {
            r = split_operand(x, GP_ACT_RELU);
        } else {
            r = new_act(x.B, x.H, x.W, x.C);
            mark("relu " + dims(x));
            launch_relu(x.p, r.p, x.pixels() * x.C, st);
        }
        ConvOpt o1;
        o1.act = GP_ACT_RELU;
        Act h = conv(r, convs.at(p + ".convolution1"), o1);
        drop(r);
        ConvOpt o2;
        o2.res = x.p;
        o2.res_f = x.f;
        Act y = conv(h, convs.at(p + ".convolution2"), o2);
        drop(h);
        return y;
    }
    // feats: reversed multi_level_feats [c0@h, c1@h, c2@h/2, c3@h/4] -> fp32 [B][8h*8w] written to out_dev
    void dpt_head(const Act* feats, float* out_dev) {
        ConvOpt ou;
        ou.ups_h = feats[0].H * 2; ou.ups_w = feats[0].W * 2;
        Act f0 = conv(feats[0], convs.at("dpt.feature_upsample_0.conv"), ou);
        Act nk[4];
        nk[0] = conv(f0, convs.at("dpt.neck.convs.0"), ConvOpt{});
        drop(f0);
        for (int i = 1; i < 4; ++i) nk[i] = conv(feats[i], convs.at("dpt.neck.convs." + std::to_string(i)), ConvOpt{});
        Act fused;
        for (int i = 0; i < 4; ++i) {
            const std::string lp = "dpt.neck.fusion_stage.layers." + std::to_string(i);
            Act& hsrc = nk[3 - i];
            Act x;
            if (i == 0) {
                x = hsrc;
            } else {
                Act r = hsrc;
                bool resized = false;
                if (r.H != fused.H || r.W != fused.W) {
                    Act rr = bilinear(r, fused.H, fused.W, 0);
                    drop(r);
                    r = rr;
                    resized = true;
                }
                (void)resized;
                Act rc = rcu(r, lp + ".residual_layer1");
                drop(r);
                x = contract ? new_act_f(fused.B, fused.H, fused.W, fused.C) : new_act(fused.B, fused.H, fused.W, fused.C);
                mark("add " + dims(fused));
                if (contract) launch_c_add(fused.f, rc.f, x.f, fused.pixels() * fused.C, st);
                else launch_add(fused.p, rc.p, x.p, fused.pixels() * fused.C, st);
                drop(rc);
                drop(fused);
            }
            Act x2 = rcu(x, lp + ".residual_layer2");
            drop(x);
            // dpt_head.py:303-309 interpolates (bilinear x2, align_corners) and THEN applies the 1x1 projection.  The two commute -- the projection mixes
            // channels per pixel, the interpolation mixes pixels per channel with weights that sum to one, so the bias passes through as well --
            // and the projection on the SOURCE map is a quarter of the GEMM and of its HBM traffic (r5)
            Act pr = linear(x2, convs.at(lp + ".projection"));
            drop(x2);
            fused = bilinear(pr, pr.H * 2, pr.W * 2, 1);
            drop(pr);
        }
        ConvOpt op;
        op.act = GP_ACT_RELU;
        Act x = conv(fused, convs.at("dpt.head.projection"), op);
        drop(fused);
        Act y = conv(x, convs.at("dpt.head.head.0"), ConvOpt{});
        drop(x);
        Act up = bilinear(y, y.H * 2, y.W * 2, 1);
        drop(y);
        ConvOpt o32;
        o32.act = GP_ACT_RELU;
        Act z = conv(up, convs.at("dpt.head.head.2"), o32);
        drop(up);
        mark("dpt_final " + dims(z));
        if (contract) launch_c_dpt_final(z.f, dpt_w_dev, dpt_b, out_dev, z.B, z.H * z.W, z.C, st);
        else launch_dpt_final(z.p, dpt_w_dev, dpt_b, out_dev, z.B, z.H * z.W, z.C, st);
        drop(z);
    }

    Act from_nchw_f32(const float* src, int B, int C, int Hh, int Ww, int Cpad) {
        Act a = contract ? new_act_f(B, Hh, Ww, Cpad) : new_act(B, Hh, Ww, Cpad);
        mark("nchw_f32_to_nhwc");
        if (contract) launch_c_nchw_to_nhwc(src, a.f, B, C, Hh, Ww, Cpad, st);
        else launch_nchw_f32_to_nhwc(src, a.p, B, C, Hh, Ww, Cpad, st);
        return a;
    }
    void to_nchw_f32(const Act& a, int C, float* dst) {
        mark("nhwc_to_nchw_f32");
        if (a.f) launch_c_nhwc_to_nchw(a.f, dst, a.B, C, a.H, a.W, a.C, st);
        else launch_nhwc_to_nchw_f32(a.p, dst, a.B, C, a.H, a.W, a.C, st);
    }

    void collect_profile() {
        if (prof < 2 || ev_used == 0) return;
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 0; i < ev_used; ++i) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
            if (ev_kind[i] == 1) tm.ms_attn += ms;
            else tm.ms_igemm += ms;                  // the whole conv / GEMM family ...
            if (ev_kind[i] == 2) tm.ms_halo += ms;   // ... and its dominant member, conv3x3_halo3_kernel, on its own
        }
        ev_used = 0;
    }
};

// =====================================================================================================================
// C-ABI
// =====================================================================================================================
template <typename F>
static gp_status guard(gp_engine* e, F&& f) {
    gp_status st = GP_OK;
    try {
        f();
        return GP_OK;
    } catch (const std::out_of_range& ex) {
        if (e) e->err = ex.what();
        st = GP_ERR_MISSING_WEIGHT;
    } catch (const std::invalid_argument& ex) {
        if (e) e->err = ex.what();
        st = GP_ERR_INVALID;
    } catch (const std::logic_error& ex) {
        if (e) e->err = ex.what();
        st = GP_ERR_STATE;
    } catch (const std::exception& ex) {
        if (e) e->err = ex.what();
        st = GP_ERR_HIP;
    }
    // a stage that threw abandoned its activations: wait for whatever it had enqueued, then hand every outstanding buffer back to the pool
    if (e && !e->pool.live_.empty()) {
        (void)hipSetDevice(e->cfg.device);
        (void)hipDeviceSynchronize();
        e->pool.release_live();
    }
    return st;
}

// Scratch of the per-kernel entry points (the parity-test interface below; engines use their own pool): one set per DEVICE, and the
// entry points that use it hold g_scratch_mu for the duration of their enqueue, so two host threads cannot resize it under each other.
struct DevScratch {
    h16_t* zero = nullptr;
    float* bufs[3] = {nullptr, nullptr, nullptr};  // 0: GroupNorm workspace, 1: statistics partials, 2: split-K partial sums
    size_t floats[3] = {0, 0, 0};
};
static std::mutex g_scratch_mu;
// what every per-kernel (test / tool) entry point opens with: the scratch lock, and the A/B switches as the environment has them NOW
struct KernelEntry {
    std::lock_guard<std::mutex> g;
    KernelEntry() : g(g_scratch_mu) { gp_switches_reload(); }
};
static std::map<int, DevScratch> g_scratch;
static DevScratch& dev_scratch() {  // call with g_scratch_mu held
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    return g_scratch[dev];
}
static h16_t* zero_page() {
    DevScratch& d = dev_scratch();
    if (!d.zero) {
        HIPCHK(hipMalloc((void**)&d.zero, 4096));
        HIPCHK(hipMemset(d.zero, 0, 4096));
    }
    return d.zero;
}
static float* scratch_floats(int which, size_t need) {
    DevScratch& d = dev_scratch();
    if (need > d.floats[which]) {
        if (d.bufs[which]) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(d.bufs[which])); d.bufs[which] = nullptr; d.floats[which] = 0; }
        HIPCHK(hipMalloc((void**)&d.bufs[which], need * sizeof(float)));
        d.floats[which] = need;
    }
    return d.bufs[which];
}
// split-K workspace for a per-kernel call (engines take theirs from the pool)
static void attach_splitk_scratch(IGemmParams& p, int tile_hint) {
    const int S = igemm_ksplit(p, tile_hint);
    if (S > 1) {
        p.splitk_ws_floats = (long long)S * p.M * p.n_store;
        p.splitk_ws = scratch_floats(2, (size_t)p.splitk_ws_floats);
    }
}

extern "C" {

const char* gp_version(void) { return GP_F16 ? "genpercept_hip 0.3 (gfx950, fp16 elements)" : "genpercept_hip 0.3 (gfx950, bf16 elements)"; }
int gp_abi_version(void) { return GP_ABI_VERSION; }
gp_dtype gp_element_dtype(void) { return GP_F16 ? GP_DT_F16 : GP_DT_BF16; }

void gp_default_config(gp_config* c) {
    memset(c, 0, sizeof(*c));
    c->device = 0;
    c->unet_in_channels = 4; c->unet_out_channels = 4;
    const int bo[4] = {320, 640, 1280, 1280}, nh[4] = {5, 10, 20, 20}, da[4] = {1, 1, 1, 0};
    const int vb[4] = {128, 256, 512, 512}, dn[4] = {320, 640, 1280, 1280};
    for (int i = 0; i < 4; ++i) { c->unet_block_out[i] = bo[i]; c->unet_num_heads[i] = nh[i]; c->unet_down_attn[i] = da[i]; c->vae_block_out[i] = vb[i]; c->dpt_neck[i] = dn[i]; }
    c->unet_layers_per_block = 2; c->unet_cross_dim = 1024; c->unet_has_out = 1; c->unet_norm_eps = 1e-5f;
    c->vae_layers_per_block = 2; c->vae_latent_channels = 4; c->vae_norm_eps = 1e-6f; c->vae_scaling_factor = 0.18215f;
    c->dpt_enabled = 0; c->dpt_fusion = 256; c->norm_groups = 32;
}

gp_status gp_create(const gp_config* cfg, gp_engine** out) {
    if (!cfg || !out) return GP_ERR_INVALID;
    gp_switches_reload();
    gp_engine* e = new gp_engine();
    e->cfg = *cfg;
    *out = e;
    return guard(e, [&] {
        int n = 0;
        HIPCHK(hipGetDeviceCount(&n));
        if (cfg->device < 0 || cfg->device >= n) throw std::invalid_argument("no such HIP device");
        HIPCHK(hipSetDevice(cfg->device));
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, cfg->device));
        if (pr.multiProcessorCount > 0) e->ncu = pr.multiProcessorCount;
        {   // saturation flags of every translation unit on THIS device (nullptr in the bf16 build: nothing to collect)
            void* (*const tus[])() = GP_SAT_TUS;
            for (auto fn : tus) {
                void* a = fn();
                if (a && e->sat_flags.n < 16) e->sat_flags.f[e->sat_flags.n++] = (unsigned*)a;
            }
            if (e->sat_flags.n > 0) {
                HIPCHK(hipMalloc((void**)&e->sat_dev, 2 * sizeof(unsigned)));
                HIPCHK(hipMemset(e->sat_dev, 0, 2 * sizeof(unsigned)));
            }
        }
        for (int i = 0; i < 4; ++i) {
            if (cfg->unet_block_out[i] % 64 || cfg->vae_block_out[i] % 64) throw std::invalid_argument("block_out_channels must be multiples of 64");
            if (cfg->unet_down_attn[i] && cfg->unet_block_out[i] != 64 * cfg->unet_num_heads[i]) throw std::invalid_argument("attention head_dim must be 64");
        }
        if (cfg->vae_latent_channels > 8 || cfg->vae_latent_channels < 1) throw std::invalid_argument("latent_channels must be in 1..8");
    });
}

void gp_destroy(gp_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    hipDeviceSynchronize();
    for (void* p : e->weights_dev) hipFree(p);
    if (e->sat_dev) hipFree(e->sat_dev);
    e->pool.destroy();
    for (auto& pr : e->ev_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& m : e->marks) hipEventDestroy(m.ev);
    for (int i = 0; i < 4; ++i) if (e->ev[i]) hipEventDestroy(e->ev[i]);
    delete e;
}

const char* gp_last_error(const gp_engine* e) { return e ? e->err.c_str() : "null engine"; }

gp_status gp_load_tensor(gp_engine* e, const char* name, const void* host_ptr, const int64_t* shape, int ndim, gp_dtype dtype) {
    if (!e || !name || !host_ptr || (ndim > 0 && !shape)) return GP_ERR_INVALID;
    return guard(e, [&] {
        if (e->finalized) throw std::logic_error("gp_load_tensor after gp_finalize");
        HostTensor t;
        t.shape.assign(shape, shape + ndim);
        const int64_t n = t.numel();
        t.v.resize((size_t)n);
        if (dtype == GP_DT_F32) memcpy(t.v.data(), host_ptr, (size_t)n * 4);
        else if (dtype == GP_DT_F16) { const uint16_t* s = (const uint16_t*)host_ptr; for (int64_t i = 0; i < n; ++i) t.v[i] = half_to_float(s[i]); }
        else if (dtype == GP_DT_BF16) { const uint16_t* s = (const uint16_t*)host_ptr; for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)s[i] << 16; memcpy(&t.v[i], &u, 4); } }
        else throw std::invalid_argument("unknown dtype");
        e->host[name] = std::move(t);
    });
}

gp_status gp_set_context(gp_engine* e, const float* embed, int L, int D) {
    if (!e || !embed || L < 1 || D < 1) return GP_ERR_INVALID;
    return guard(e, [&] {
        e->ctx.assign(embed, embed + (size_t)L * D);
        e->ctx_L = L;
        e->ctx_D = D;
        if (e->finalized) { HIPCHK(hipSetDevice(e->cfg.device)); HIPCHK(hipDeviceSynchronize()); e->fold_context(); }
    });
}

gp_status gp_set_timestep(gp_engine* e, float t) {
    if (!e) return GP_ERR_INVALID;
    return guard(e, [&] {
        e->timestep = t;
        if (e->finalized) { HIPCHK(hipSetDevice(e->cfg.device)); HIPCHK(hipDeviceSynchronize()); e->fold_timestep(); }
    });
}

gp_status gp_set_precision(gp_engine* e, gp_precision prec) {
    if (!e || (prec != GP_PREC_NATIVE && prec != GP_PREC_CONTRACT)) return GP_ERR_INVALID;
    return guard(e, [&] {
        if (e->finalized) throw std::logic_error("gp_set_precision after gp_finalize (the weights are packed per precision)");
        // the split is written for bf16 pieces (fp32 range: neither piece can saturate or flush); the fp16 library stays a 16-bit engine
        if (GP_F16 && prec == GP_PREC_CONTRACT) throw std::invalid_argument("contract precision lives in the bf16 library (libgenpercept_hip.so)");
        e->contract = prec == GP_PREC_CONTRACT;
    });
}
gp_precision gp_get_precision(const gp_engine* e) { return (e && e->contract) ? GP_PREC_CONTRACT : GP_PREC_NATIVE; }

gp_status gp_finalize(gp_engine* e) {
    if (!e) return GP_ERR_INVALID;
    return guard(e, [&] { e->finalize(); });
}

gp_status gp_set_profile(gp_engine* e, int level) {
    if (!e) return GP_ERR_INVALID;
    e->prof = level;
    return GP_OK;
}
gp_status gp_reset_timings(gp_engine* e) {
    if (!e) return GP_ERR_INVALID;
    e->tm = gp_timings{};
    e->flops_halo_exec = 0.0;
    e->ev_used = 0;
    return GP_OK;
}
/* MFMA flops the halo-conv launches EXECUTED since gp_reset_timings.  gp_timings.flops_halo counts algorithmic flops (2 M N 9 Cin, what a roofline
 * figure is quoted on); the two differ where a launch does less arithmetic than its algorithmic count: the x2-upsample convs run as four 2 x 2-tap phase
 * convolutions, 4/9 of the flops (conv_halo.hip, PH).  (A new entry point, not a new field: gp_timings is frozen.) */
gp_status gp_halo_executed_flops(gp_engine* e, double* flops) {
    if (!e || !flops) return GP_ERR_INVALID;
    *flops = e->flops_halo_exec;
    return GP_OK;
}
gp_status gp_get_timings(gp_engine* e, gp_timings* out) {
    if (!e || !out) return GP_ERR_INVALID;
    return guard(e, [&] {
        e->collect_profile();
        *out = e->tm;
    });
}

/* fp16 library: number of (call, translation unit) pairs in which a saturating fp32 -> fp16 conversion actually clipped since the last reset
 * (0 = every stored activation was inside the fp16 range: the output is not silently clipped).  Always 0 in the bf16 library.  Synchronises
 * the engine's stream. */
gp_status gp_saturation_events(gp_engine* e, long long* events, int reset) {
    if (!e || !events) return GP_ERR_INVALID;
    return guard(e, [&] {
        *events = 0;
        if (!e->sat_dev) return;
        HIPCHK(hipSetDevice(e->cfg.device));
        unsigned n = 0;
        // the flag words are per translation unit, process and device: a per-kernel entry point (gp_conv2d, gp_gemm, ... never collect) or another
        // engine on this device may have left one set.  Collect before reading so that a reset really starts from clean flags.
        e->collect_saturation();
        HIPCHK(hipStreamSynchronize(e->st));
        HIPCHK(hipMemcpy(&n, e->sat_dev, sizeof(n), hipMemcpyDeviceToHost));
        *events = (long long)n;
        if (reset) HIPCHK(hipMemset(e->sat_dev, 0, sizeof(unsigned)));
    });
}

/* Per-launch log of the last gp_infer at profiling level 3: one line "ms<TAB>flops<TAB>name" per launch (ms = time to the next launch's
 * start: kernel + gap).  Returns the number of bytes the full log needs (incl. NUL); writes at most cap bytes. */
int gp_get_launch_log(gp_engine* e, char* buf, int cap) {
    if (!e) return -1;
    std::string out;
    try {
        if (e->marks_used > 1) {
            HIPCHK(hipEventSynchronize(e->marks[e->marks_used - 1].ev));
            for (size_t i = 0; i + 1 < e->marks_used; ++i) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e->marks[i].ev, e->marks[i + 1].ev));
                char line[64];
                snprintf(line, sizeof line, "%.4f\t%.0f\t", ms, e->marks[i].flops);
                out += line;
                out += e->marks[i].name;
                out += "\n";
            }
        }
    } catch (const std::exception& ex) {
        e->err = ex.what();
        return -1;
    }
    if (buf && cap > 0) {
        const size_t n = std::min(out.size(), (size_t)cap - 1);
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size() + 1;
}

static void check_ready(gp_engine* e, void* stream) {
    if (!e->finalized) throw std::logic_error("engine not finalized");
    HIPCHK(hipSetDevice(e->cfg.device));
    // the activation pool hands a buffer to its next user as soon as the last consumer is ENQUEUED, which is only safe within one
    // stream: when the caller switches streams, everything issued on the previous one is waited for first
    if (e->stream_seen && e->last_stream != (hipStream_t)stream) HIPCHK(hipStreamSynchronize(e->last_stream));
    e->last_stream = (hipStream_t)stream;
    e->stream_seen = true;
    e->st = (hipStream_t)stream;
}

gp_status gp_infer(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, gp_mode mode, float* out_dev, void* stream) {
    if (!e || !rgb_dev || !out_dev) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        if (B < 1 || H < 8 || W < 8) throw std::invalid_argument("need B >= 1 and H, W >= 8");
        const bool stage_ev = e->prof >= 1;
        if (stage_ev) {
            for (int i = 0; i < 4; ++i) if (!e->ev[i]) HIPCHK(hipEventCreate(&e->ev[i]));
            HIPCHK(hipEventRecord(e->ev[0], e->st));
        }
        e->marks_used = 0;
        Act lat = e->vae_encode(rgb_dev, is_u8, B, H, W);
        if (stage_ev) HIPCHK(hipEventRecord(e->ev[1], e->st));
        if (!e->cfg.dpt_enabled) {
            Act v = e->unet(lat, nullptr, true);
            e->drop(lat);
            if (stage_ev) HIPCHK(hipEventRecord(e->ev[2], e->st));
            // scheduler step with beta == 1: pred_x0 = -v (F5); decode_pred divides by the scaling factor
            const int mean3 = !(mode == GP_MODE_NORMAL || mode == GP_MODE_SEG);
            Act dec = e->vae_decode(v, -1.0f / e->cfg.vae_scaling_factor, out_dev, mean3, 0);
            e->drop(v);
            if (dec.p || dec.f) {  // (the fused tail kernel wrote out_dev itself otherwise)
                e->decode_epilogue(dec, out_dev, mean3, 0);
                e->drop(dec);
            }
        } else {
            Act feats[4];
            e->unet(lat, feats, false);
            e->drop(lat);
            if (stage_ev) HIPCHK(hipEventRecord(e->ev[2], e->st));
            Act rev[4] = {feats[3], feats[2], feats[1], feats[0]};
            const long long out_px = (long long)gp_dpt_out_size(rev[0].H) * gp_dpt_out_size(rev[0].W);
            e->dpt_head(rev, out_dev);
            for (int i = 0; i < 4; ++i) e->drop(feats[i]);
            e->mark("minmax_norm", 0.0, 2);
            float* mm_ws = (float*)e->pool.alloc((size_t)B * 64 * 2 * sizeof(float));  // [B][64 partials][min, max]
            launch_minmax_norm(out_dev, B, out_px, mm_ws, e->st);
            e->pool.release(mm_ws);
        }
        e->mark("END", 0.0, 0);
        if (stage_ev) {
            HIPCHK(hipEventRecord(e->ev[3], e->st));
            HIPCHK(hipEventSynchronize(e->ev[3]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_encode, e->ev[0], e->ev[1]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_unet, e->ev[1], e->ev[2]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_head, e->ev[2], e->ev[3]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_total, e->ev[0], e->ev[3]));
        }
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_infer_steps(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, gp_mode mode, const gp_ddim_step* steps,
                         int n_steps, const float* noise_dev, float* out_dev, void* stream) {
    if (!e || !rgb_dev || !out_dev || !steps) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        if (B < 1 || H < 8 || W < 8 || n_steps < 1) throw std::invalid_argument("need B >= 1, H, W >= 8 and at least one step");
        if (e->cfg.dpt_enabled || !e->cfg.unet_has_out) throw std::invalid_argument("the multi-step archs use the VAE-decoder head");
        const int L = e->cfg.vae_latent_channels;
        if (e->cfg.unet_in_channels != (noise_dev ? 2 * L : L))
            throw std::invalid_argument(noise_dev ? "an initial noise sample needs a UNet with 2 x latent input channels (marigold, run.py:59-78)"
                                                  : "without an initial noise sample the UNet takes the latent channels only (rgb_blending)");
        const bool stage_ev = e->prof >= 1;
        if (stage_ev) {
            for (int i = 0; i < 4; ++i) if (!e->ev[i]) HIPCHK(hipEventCreate(&e->ev[i]));
            HIPCHK(hipEventRecord(e->ev[0], e->st));
        }
        e->marks_used = 0;
        const float t_before = e->timestep;
        Act lat = e->vae_encode(rgb_dev, is_u8, B, H, W);  // channels 0..L-1 = rgb latent; the UNet input tensor from here on
        if (stage_ev) HIPCHK(hipEventRecord(e->ev[1], e->st));
        const int off = noise_dev ? L : 0;
        float* sample = (float*)e->pool.alloc((size_t)lat.pixels() * L * sizeof(float));
        e->mark("ddim_init");
        if (e->contract) launch_c_ddim_init(noise_dev, lat.f, sample, B, lat.H, lat.W, L, lat.C, off, e->st);
        else launch_ddim_init(noise_dev, lat.p, sample, B, lat.H, lat.W, L, lat.C, off, e->st);
        Act x0 = e->contract ? e->new_act_f(B, lat.H, lat.W, 64) : e->new_act(B, lat.H, lat.W, 64);
        try {
            for (int i = 0; i < n_steps; ++i) {
                const gp_ddim_step& s = steps[i];
                if (s.timestep != e->timestep) e->set_timestep_on_stream(s.timestep);
                Act v = e->unet(lat, nullptr, true);
                const DdimCoef k{s.x0_sample, s.x0_model, s.eps_sample, s.eps_model, s.prev_x0, s.prev_eps, s.clip};
                e->mark("ddim_step");
                if (e->contract) launch_c_ddim_step(v.f, v.C, sample, lat.f, lat.C, off, i == n_steps - 1 ? x0.f : nullptr, x0.C, lat.pixels(), L, k, e->st);
                else launch_ddim_step(v.p, v.C, sample, lat.p, lat.C, off, i == n_steps - 1 ? x0.p : nullptr, x0.C, lat.pixels(), L, k, e->st);
                e->drop(v);
            }
        } catch (...) {  // a failed step must not leave the loop's timestep behind as the engine's (gp_set_timestep) one
            if (e->timestep != t_before) {
                try { e->set_timestep_on_stream(t_before); } catch (...) {}
            }
            throw;
        }
        e->pool.release(sample);
        e->drop(lat);
        if (e->timestep != t_before) e->set_timestep_on_stream(t_before);
        if (stage_ev) HIPCHK(hipEventRecord(e->ev[2], e->st));
        const int mean3 = !(mode == GP_MODE_NORMAL || mode == GP_MODE_SEG);
        Act dec = e->vae_decode(x0, 1.0f / e->cfg.vae_scaling_factor, out_dev, mean3, 0);
        e->drop(x0);
        if (dec.p || dec.f) {
            e->decode_epilogue(dec, out_dev, mean3, 0);
            e->drop(dec);
        }
        e->mark("END", 0.0, 0);
        if (stage_ev) {
            HIPCHK(hipEventRecord(e->ev[3], e->st));
            HIPCHK(hipEventSynchronize(e->ev[3]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_encode, e->ev[0], e->ev[1]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_unet, e->ev[1], e->ev[2]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_head, e->ev[2], e->ev[3]));
            HIPCHK(hipEventElapsedTime(&e->tm.ms_total, e->ev[0], e->ev[3]));
        }
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_vae_encode(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, float* latent_out, void* stream) {
    if (!e || !rgb_dev || !latent_out) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        Act lat = e->vae_encode(rgb_dev, is_u8, B, H, W);
        e->to_nchw_f32(lat, e->cfg.vae_latent_channels, latent_out);
        e->drop(lat);
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_unet(gp_engine* e, const float* latent_in, int B, int h, int w, float* sample_out, float* const* feats_out, void* stream) {
    if (!e || !latent_in) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        Act lat = e->from_nchw_f32(latent_in, B, e->cfg.unet_in_channels, h, w, 64);
        Act feats[4];
        Act v = e->unet(lat, feats_out ? feats : nullptr, sample_out != nullptr);
        e->drop(lat);
        if (sample_out) {
            if (!v.p && !v.f) throw std::logic_error("this UNet has no conv_out (DPT variant)");
            e->to_nchw_f32(v, e->cfg.unet_out_channels, sample_out);
        }
        if (v.p || v.f) e->drop(v);
        if (feats_out)
            for (int i = 0; i < 4; ++i) {
                if (feats_out[i]) e->to_nchw_f32(feats[i], feats[i].C, feats_out[i]);
                e->drop(feats[i]);
            }
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_vae_decode(gp_engine* e, const float* pred_latent, int B, int h, int w, int mean3, float* out, void* stream) {
    if (!e || !pred_latent || !out) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        Act z = e->from_nchw_f32(pred_latent, B, e->cfg.vae_latent_channels, h, w, 64);
        // decode_pred (genpercept_pipeline.py:507-526): channel mean for 1-channel modes, no clip / shift
        Act dec = e->vae_decode(z, 1.0f / e->cfg.vae_scaling_factor, out, mean3, 1);
        e->drop(z);
        if (dec.p || dec.f) {
            e->decode_epilogue(dec, out, mean3, 1);
            e->drop(dec);
        }
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_vae_mid_attention(gp_engine* e, int decoder, const float* x, int B, int h, int w, float* out, void* stream) {
    if (!e || !x || !out) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        const std::string name = decoder ? "vae.decoder.mid_block.attentions.0" : "vae.encoder.mid_block.attentions.0";
        const int C = e->vattn.at(name).C;
        Act a = e->from_nchw_f32(x, B, C, h, w, C);
        Act y = e->vae_attention(a, name);
        e->drop(a);
        e->to_nchw_f32(y, C, out);
        e->drop(y);
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

gp_status gp_dpt_head(gp_engine* e, const float* const* feats, int B, int h, int w, float* out, void* stream) {
    if (!e || !feats || !out) return GP_ERR_INVALID;
    return guard(e, [&] {
        check_ready(e, stream);
        if (!e->dpt_w_dev) throw std::logic_error("DPT head weights were not loaded");
        const int h2 = (h - 1) / 2 + 1, w2 = (w - 1) / 2 + 1;  // UNet stride-2 pad-1 downsamples
        const int hs[4] = {h, h, h2, (h2 - 1) / 2 + 1}, ws[4] = {w, w, w2, (w2 - 1) / 2 + 1};
        Act f[4];
        for (int i = 0; i < 4; ++i) f[i] = e->from_nchw_f32(feats[i], B, e->cfg.dpt_neck[i], hs[i], ws[i], e->cfg.dpt_neck[i]);
        e->dpt_head(f, out);
        for (int i = 0; i < 4; ++i) e->drop(f[i]);
        e->collect_saturation();
        HIPCHK(hipGetLastError());
    });
}

// ---- per-kernel entry points --------------------------------------------------------------------------------------
int gp_packed_rows(int cout) { return (cout + 255) / 256 * 256; }
int gp_latent_size(int x) { for (int i = 0; i < 3; ++i) x = (x - 2) / 2 + 1; return x; }
int gp_dpt_out_size(int latent) { for (int i = 0; i < 2; ++i) latent = (latent - 1) / 2 + 1; return 32 * latent; }

gp_status gp_pack_weight(const float* w, int cout, int cin, int ks, int cin_pad, int geglu, void* dev_out) {
    if (!w || !dev_out || cout < 1 || cin < 1 || (ks != 1 && ks != 3) || cin_pad < cin || (cin_pad % 64)) return GP_ERR_INVALID;
    try {
        const int n_rows = gp_packed_rows(cout);
        std::vector<h16_t> buf((size_t)n_rows * ks * ks * cin_pad, 0);
        gp_engine::pack_rows(w, cout, cin, ks, cin_pad, geglu != 0, buf, 0, n_rows);
        HIPCHK(hipMemcpy(dev_out, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_conv2d(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int Hi, int Wi, int Cin,
                    int Cout, int ks, int stride, int pad_t, int pad_l, int Ho, int Wo, int ups_h, int ups_w, int act, int n_store,
                    int out_fp32, int tile_hint, void* stream) {
    if (!in || !w_packed || !out || (Cin % 64) || (ks != 1 && ks != 3)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        IGemmParams p{};
        p.in = (const h16_t*)in; p.wt = (const h16_t*)w_packed; p.bias = bias; p.res = (const h16_t*)residual; p.out = out; p.zero = zero_page();
        p.M = B * Ho * Wo; p.N = Cout; p.Cin = Cin; p.n_rows = gp_packed_rows(Cout); p.ks = ks;
        p.B = B; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
        p.ups = ups_h > 0; p.Hu = ups_h; p.Wu = ups_w;
        const int nout = act == GP_ACT_GEGLU ? Cout / 2 : Cout;
        const int nst = n_store > 0 ? n_store : nout;
        p.lda = Cin; p.ldo = nst; p.ldres = nst; p.ldw = (ks == 3 ? 9 : 1) * Cin; p.n_store = nst; p.out_fp32 = out_fp32; p.act = act;
        p.bias_mode = bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        p.dbg = gp_sw().igemm_dbg;  // profiling ablations (tools/conv_bench.py)
        attach_splitk_scratch(p, tile_hint);
        launch_igemm(p, tile_hint, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_pack_weight_phases(const float* w, int cout, int cin, int cin_pad, void* dev_out) {
    if (!w || !dev_out || cout < 1 || cin < 1 || cin_pad < cin || (cin_pad % 64)) return GP_ERR_INVALID;
    try {
        std::vector<h16_t> buf((size_t)gp_packed_rows(cout) * 16 * cin_pad, 0);
        gp_engine::pack_phase_rows(w, cout, cin, cin_pad, buf);
        HIPCHK(hipMemcpy(dev_out, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_conv2d_up2(const void* in, const void* w_packed, const void* w_phases, const float* bias, const void* residual, void* out, int B, int Hi, int Wi,
                        int Cin, int Cout, void* stream) {
    if (!in || !w_packed || !w_phases || !out || (Cin % 64) || (Cout % 8)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        IGemmParams p{};
        p.in = (const h16_t*)in; p.wt = (const h16_t*)w_packed; p.wt_ph = (const h16_t*)w_phases; p.bias = bias; p.res = (const h16_t*)residual; p.out = out;
        p.zero = zero_page();
        p.M = B * 4 * Hi * Wi; p.N = Cout; p.Cin = Cin; p.n_rows = gp_packed_rows(Cout); p.ks = 3;
        p.B = B; p.Hi = Hi; p.Wi = Wi; p.Ho = 2 * Hi; p.Wo = 2 * Wi; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
        p.ups = 1; p.Hu = 2 * Hi; p.Wu = 2 * Wi;
        p.lda = Cin; p.ldo = Cout; p.ldres = Cout; p.ldw = 9 * Cin; p.n_store = Cout; p.bias_mode = bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        p.dbg = gp_sw().igemm_dbg;
        if (!conv_uses_halo(p, 5) || !conv_halo_uses_phases(p)) return GP_ERR_INVALID;  // (this entry point exists to test the phase kernel)
        launch_igemm(p, 5, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_conv2d_up2_stats(const void* in, const void* w_packed, const void* w_phases, const float* bias, const void* residual, void* out, int B, int Hi,
                              int Wi, int Cin, int Cout, const float* gamma, const float* beta, int groups, float eps, float* scale_out, float* shift_out,
                              void* stream) {
    if (!in || !w_packed || !w_phases || !out || !gamma || !beta || !scale_out || !shift_out || (Cin % 64) || (Cout % 8) || groups < 1 || (Cout % groups))
        return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        IGemmParams p{};
        p.in = (const h16_t*)in; p.wt = (const h16_t*)w_packed; p.wt_ph = (const h16_t*)w_phases; p.bias = bias; p.res = (const h16_t*)residual; p.out = out;
        p.zero = zero_page();
        p.M = B * 4 * Hi * Wi; p.N = Cout; p.Cin = Cin; p.n_rows = gp_packed_rows(Cout); p.ks = 3;
        p.B = B; p.Hi = Hi; p.Wi = Wi; p.Ho = 2 * Hi; p.Wo = 2 * Wi; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
        p.ups = 1; p.Hu = 2 * Hi; p.Wu = 2 * Wi;
        p.lda = Cin; p.ldo = Cout; p.ldres = Cout; p.ldw = 9 * Cin; p.n_store = Cout; p.bias_mode = bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        if (!conv_uses_halo(p, 5) || !conv_halo_uses_phases(p)) return GP_ERR_INVALID;
        int mode = 0, bm = 0;
        const int nt = igemm_tile_info(p, 5, &mode, &bm);
        if (nt <= 0) return GP_ERR_INVALID;
        float* part = scratch_floats(1, (size_t)nt * (Cout * 2 + 1));
        p.stats_out = part;
        launch_igemm(p, 5, (hipStream_t)stream);
        launch_groupnorm_from_partials(part, mode, bm, B, 2 * Hi, 2 * Wi, Cout, groups, eps, gamma, beta, scale_out, shift_out, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_conv2d_gn(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int H, int W, int Cin,
                       int Cout, int ups, int act, const float* gamma, const float* beta, int groups, float eps, int silu, void* stream) {
    if (!in || !w_packed || !out || !gamma || !beta || (Cin % 64) || (Cin % groups)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        const int Ho = ups ? 2 * H : H, Wo = ups ? 2 * W : W;
        IGemmParams p{};
        p.in = (const h16_t*)in; p.wt = (const h16_t*)w_packed; p.bias = bias; p.res = (const h16_t*)residual; p.out = out; p.zero = zero_page();
        p.M = B * Ho * Wo; p.N = Cout; p.Cin = Cin; p.n_rows = gp_packed_rows(Cout); p.ks = 3;
        p.B = B; p.Hi = H; p.Wi = W; p.Ho = Ho; p.Wo = Wo; p.stride = 1; p.pad_t = 1; p.pad_l = 1;
        p.ups = ups ? 1 : 0; p.Hu = ups ? Ho : 0; p.Wu = ups ? Wo : 0;
        p.lda = Cin; p.ldo = Cout; p.ldres = Cout; p.ldw = 9 * Cin; p.n_store = Cout; p.act = act;
        p.bias_mode = bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        const size_t need = (size_t)groupnorm_ws_floats(B, H * W, Cin, groups) + 2 * (size_t)B * Cin;
        float* g_gn_ws = scratch_floats(0, need);
        float* scale = g_gn_ws + groupnorm_ws_floats(B, H * W, Cin, groups);
        float* shift = scale + (size_t)B * Cin;
        launch_groupnorm_stats((const h16_t*)in, gamma, beta, B, H * W, Cin, groups, eps, g_gn_ws, scale, shift, (hipStream_t)stream);
        p.in_scale = scale; p.in_shift = shift; p.in_silu = silu;
        if (!conv_uses_halo(p, 5)) return GP_ERR_INVALID;
        launch_igemm(p, 5, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_rgb_conv_in(const void* rgb, int is_u8, const void* w_packed, const float* bias, void* out, int B, int H, int W, int Cout, void* stream) {
    if (!rgb || !w_packed || !out || B < 1 || H < 1 || W < 1 || (Cout % 32)) return GP_ERR_INVALID;
    h16_t* w27 = nullptr;
    if (hipMalloc((void**)&w27, (size_t)Cout * 32 * sizeof(h16_t)) != hipSuccess) return GP_ERR_HIP;
    launch_pack_k27((const h16_t*)w_packed, 9 * 64, Cout, w27, (hipStream_t)stream);
    launch_rgb_conv_in(rgb, is_u8, w27, bias, (h16_t*)out, nullptr, B, H, W, Cout, (hipStream_t)stream);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(w27);
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? GP_OK : GP_ERR_HIP;
}

gp_status gp_conv2d_stats(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int H, int W, int Cin,
                          int Cout, int ks, int ups, int tile_hint, const float* gamma, const float* beta, int groups, float eps,
                          float* scale_out, float* shift_out, void* stream) {
    if (!in || !w_packed || !out || !gamma || !beta || !scale_out || !shift_out || (Cin % 64) || (ks != 1 && ks != 3) || groups < 1 || (Cout % groups))
        return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        const int Ho = ups ? 2 * H : H, Wo = ups ? 2 * W : W;
        IGemmParams p{};
        p.in = (const h16_t*)in; p.wt = (const h16_t*)w_packed; p.bias = bias; p.res = (const h16_t*)residual; p.out = out; p.zero = zero_page();
        p.M = B * Ho * Wo; p.N = Cout; p.Cin = Cin; p.n_rows = gp_packed_rows(Cout); p.ks = ks;
        p.B = B; p.Hi = H; p.Wi = W; p.Ho = Ho; p.Wo = Wo; p.stride = 1; p.pad_t = ks == 3; p.pad_l = ks == 3;
        p.ups = ups ? 1 : 0; p.Hu = ups ? Ho : 0; p.Wu = ups ? Wo : 0;
        p.lda = Cin; p.ldo = Cout; p.ldres = Cout; p.ldw = (ks == 3 ? 9 : 1) * Cin; p.n_store = Cout;
        p.bias_mode = bias ? GP_BIAS_COL : GP_BIAS_NONE; p.batch = 1;
        p.dbg = gp_sw().igemm_dbg;
        int mode = 0, bm = 0;
        const int nt = igemm_tile_info(p, tile_hint, &mode, &bm);
        if (nt <= 0) return GP_ERR_INVALID;
        const size_t need = (size_t)nt * (Cout * 2 + 1);
        float* part = scratch_floats(1, need);
        p.stats_out = part;
        launch_igemm(p, tile_hint, (hipStream_t)stream);
        launch_groupnorm_from_partials(part, mode, bm, B, Ho, Wo, Cout, groups, eps, gamma, beta, scale_out, shift_out, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_gemm(const void* a, int lda, const void* bt, int ldb, const float* bias, int bias_mode, const void* residual, int ldres, void* out,
                  int ldo, int M, int N, int K, int n_rows_bt, int n_store, int act, int out_fp32, int batch, long long a_bs, long long bt_bs,
                  long long out_bs, int tile_hint, void* stream) {
    if (!a || !bt || !out || (K % 64)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        IGemmParams p{};
        p.in = (const h16_t*)a; p.wt = (const h16_t*)bt; p.bias = bias; p.res = (const h16_t*)residual; p.out = out; p.zero = zero_page();
        p.M = M; p.N = N; p.Cin = K; p.n_rows = n_rows_bt; p.ks = 1; p.stride = 1;
        p.lda = lda; p.ldw = ldb; p.ldo = ldo; p.ldres = ldres; p.n_store = n_store > 0 ? n_store : N; p.out_fp32 = out_fp32; p.act = act;
        p.bias_mode = bias ? bias_mode : GP_BIAS_NONE; p.batch = batch > 0 ? batch : 1; p.in_bs = a_bs; p.wt_bs = bt_bs; p.out_bs = out_bs;
        p.dbg = gp_sw().igemm_dbg;  // profiling ablations (tools/kbench)
        attach_splitk_scratch(p, tile_hint);
        launch_igemm(p, tile_hint, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_decoder_tail(const void* in, const void* w_packed, const float* bias, const float* gamma, const float* beta, int groups, float eps,
                          int B, int H, int W, int Cin, int mean3, int raw, float* out, void* stream) {
    if (!in || !w_packed || !gamma || !beta || !out || B < 1 || !conv_few_applicable(Cin, 3, H, W) || (Cin % groups)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        const size_t need = (size_t)groupnorm_ws_floats(B, H * W, Cin, groups) + 2 * (size_t)B * Cin;
        float* ws = scratch_floats(0, need);
        float* scale = ws + groupnorm_ws_floats(B, H * W, Cin, groups);
        float* shift = scale + (size_t)B * Cin;
        launch_groupnorm_stats((const h16_t*)in, gamma, beta, B, H * W, Cin, groups, eps, ws, scale, shift, (hipStream_t)stream);
        launch_conv_few((const h16_t*)in, (const h16_t*)w_packed, bias, scale, shift, zero_page(), out, B, H, W, 1, mean3, raw, 0, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_gemm_qkv(const void* a, int lda, const void* w_packed, int ldw, int n_rows_w, int K, void* qk_out, void* vt_out, int B, int T, int C,
                      int Tpad, void* stream) {
    if (!a || !w_packed || !qk_out || !vt_out || (K % 64) || B < 1 || T < 1 || C < 1 || Tpad < T) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        IGemmParams p{};
        p.in = (const h16_t*)a; p.wt = (const h16_t*)w_packed; p.out = qk_out; p.zero = zero_page();
        p.M = B * T; p.N = 3 * C; p.Cin = K; p.n_rows = n_rows_w; p.ks = 1; p.stride = 1;
        p.lda = lda; p.ldw = ldw; p.ldo = 2 * C; p.ldres = 2 * C; p.n_store = 2 * C; p.act = GP_ACT_NONE; p.bias_mode = GP_BIAS_NONE; p.batch = 1;
        p.vt_out = (h16_t*)vt_out; p.vt_col0 = 2 * C; p.vt_T = T; p.vt_Tpad = Tpad;
        p.dbg = gp_sw().igemm_dbg;
        if (!igemm_uses_pgemm(p, 0)) return GP_ERR_INVALID;  // only the persistent GEMM has the transposed epilogue
        if (Tpad != T) HIPCHK(hipMemsetAsync(vt_out, 0, (size_t)B * C * Tpad * sizeof(h16_t), (hipStream_t)stream));
        launch_igemm(p, 0, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int silu, void* stream) {
    if (!x || !y || !gamma || !beta || (C % 8) || (C % G)) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        const size_t need = (size_t)groupnorm_ws_floats(B, HW, C, G) + 2 * (size_t)B * C;
        float* g_gn_ws = scratch_floats(0, need);
        if (groupnorm_small_applicable(B, HW, C, G)) launch_groupnorm_small((const h16_t*)x, (h16_t*)y, gamma, beta, B, HW, C, G, eps, silu, (hipStream_t)stream);
        else launch_groupnorm((const h16_t*)x, (h16_t*)y, gamma, beta, B, HW, C, G, eps, silu, g_gn_ws, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps, void* stream) {
    if (!x || !y || (C % 8) || C > 4096) return GP_ERR_INVALID;
    launch_layernorm((const h16_t*)x, (h16_t*)y, gamma, beta, rows, C, eps, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_flash_attention(const void* q, const void* k, const void* vt, void* out, int B, int T, int heads, int ldq, int ldk, int Tpad, int ldo,
                             void* stream) {
    if (!q || !k || !vt || !out || (Tpad % 64) || Tpad < T) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        launch_flash_attn64((const h16_t*)q, (const h16_t*)k, (const h16_t*)vt, (h16_t*)out, B, T, heads, ldq, ldk, Tpad, ldo,
                            (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_flash_attention_split(const float* qkv, int ld, void* out_split, int B, int T, int heads, void* stream) {
    if (!qkv || !out_split || B < 1 || T < 1 || heads < 1 || ld < 3 * heads * 64 || GP_F16) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        const int C = heads * 64, Tpad = (T + 63) / 64 * 64;
        const size_t n_qk = (size_t)B * T * 2 * C, n_vt = (size_t)B * heads * 64 * Tpad;
        h16_t* buf = nullptr;
        HIPCHK(hipMalloc((void**)&buf, (2 * n_qk + 2 * n_vt) * sizeof(h16_t)));
        h16_t *qk_hi = buf, *qk_lo = buf + n_qk, *vt_hi = buf + 2 * n_qk, *vt_lo = vt_hi + n_vt;
        launch_c_qkv_planes(qkv, ld, qk_hi, qk_lo, vt_hi, vt_lo, B, T, Tpad, heads, 64, (hipStream_t)stream);
        launch_flash_attn64_split(qk_hi, qk_lo, vt_hi, vt_lo, (h16_t*)out_split, B, T, heads, 2 * C, Tpad, (hipStream_t)stream);
        const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(buf);
        if (e != hipSuccess) return GP_ERR_HIP;
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_flash_attention_hd512(const void* q, const void* k, const void* vt, void* out, int B, int T, int ldq, int ldk, int Tpad, int ldo,
                                   float scale, int ncu, void* stream) {
    if (!q || !k || !vt || !out || (Tpad % 64) || Tpad < T || B < 1 || T < 1 || ncu < 0) return GP_ERR_INVALID;
    try {
        KernelEntry lk;
        if (ncu == 0) {
            int dev = 0;
            hipDeviceProp_t pr;
            HIPCHK(hipGetDevice(&dev));
            HIPCHK(hipGetDeviceProperties(&pr, dev));
            ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        }
        const long long wsf = flash_attn512_workspace_floats(B, T, ncu);
        float* ws = wsf ? scratch_floats(2, (size_t)wsf) : nullptr;
        launch_flash_attn512((const h16_t*)q, (const h16_t*)k, (const h16_t*)vt, (h16_t*)out, ws, B, T, ldq, ldk, Tpad, ldo, scale,
                             ncu, (hipStream_t)stream);
        HIPCHK(hipGetLastError());
        return GP_OK;
    } catch (...) { return GP_ERR_HIP; }
}

gp_status gp_cross_attention(const void* q, const float* kc, const float* vc, void* out, int rows, int C, int L, void* stream) {
    if (!q || !kc || !vc || !out || (C % 64)) return GP_ERR_INVALID;
    launch_cross_attn_small((const h16_t*)q, kc, vc, (h16_t*)out, rows, C, L, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_cross_attention_fold(const void* y, void* y_out, void* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                                  const float* g3, const float* b3, int rows, int C, int heads, float eps, void* stream) {
    if (!y || !y_out || !U || !u0 || !G || !c0 || !cross_attn_fold_supported(C, heads) || (n3_out && (!g3 || !b3))) return GP_ERR_INVALID;
    gp_switches_reload();
    launch_cross_attn_fold((const h16_t*)y, (h16_t*)y_out, (h16_t*)n3_out, U, u0, G, c0, g3, b3, rows, C, heads, eps, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

void gp_resize_max_res_size(int H0, int W0, int max_edge, int* h, int* w) {
    // image_util.py:95-101: downscale_factor = min(max / W, max / H) in double, new size by int() truncation
    const double f = std::min((double)max_edge / (double)W0, (double)max_edge / (double)H0);
    if (h) *h = (int)((double)H0 * f);
    if (w) *w = (int)((double)W0 * f);
}

gp_status gp_preprocess(const void* rgb_u8, int B, int H0, int W0, void* out_u8, int h, int w, int resample, float* tmp, void* stream) {
    if (!rgb_u8 || !out_u8 || B < 1 || H0 < 1 || W0 < 1 || h < 1 || w < 1 || resample < 0 || resample > 2 || (resample != 1 && !tmp)) return GP_ERR_INVALID;
    launch_resize(rgb_u8, out_u8, tmp, (long long)B * 3, H0, W0, h, w, resample, 1, 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_preprocess_f32(const float* rgb, int B, int C, int H0, int W0, float* out, int h, int w, int resample, int normalize, float* tmp, void* stream) {
    if (!rgb || !out || B < 1 || C < 1 || H0 < 1 || W0 < 1 || h < 1 || w < 1 || resample < 0 || resample > 2) return GP_ERR_INVALID;
    const bool same = h == H0 && w == W0;
    if (!same && resample != 1 && !tmp) return GP_ERR_INVALID;
    if (same && !normalize && rgb != out) return GP_ERR_INVALID;  // nothing to do but a copy: the caller keeps its tensor
    hipStream_t s = (hipStream_t)stream;
    if (!same) launch_resize(rgb, out, tmp, (long long)B * C, H0, W0, h, w, resample, 0, 0, s);
    if (normalize) launch_normalize_rgb(same ? rgb : out, out, (long long)B * C * h * w, s);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_postprocess(const float* pred, int B, int C, int h, int w, float* pred_out, int Ho, int Wo, int resample, float* tmp,
                         const unsigned char* lut_dev, void* colored_out, void* q_out, int q_bits, void* stream) {
    if (!pred || !pred_out || B < 1 || C < 1 || h < 1 || w < 1 || Ho < 1 || Wo < 1 || resample < 0 || resample > 2) return GP_ERR_INVALID;
    if (colored_out && (!lut_dev || C != 1)) return GP_ERR_INVALID;
    if (q_out && q_bits != 16 && q_bits != 8) return GP_ERR_INVALID;
    const bool same = h == Ho && w == Wo;
    if (!same && resample != 1 && !tmp) return GP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * C * Ho * Wo;
    if (same) launch_clip01(pred, pred_out, n, s);
    else launch_resize(pred, pred_out, tmp, (long long)B * C, h, w, Ho, Wo, resample, 0, 1, s);
    if (colored_out) launch_colorize_lut(pred_out, lut_dev, (unsigned char*)colored_out, n, s);
    if (q_out) launch_quantize(pred_out, q_out, n, q_bits, s);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

double gp_mfma_peak_tflops(int device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) return -1.0;
    return mfma_peak_tflops(20, (hipStream_t)stream);
}
double gp_mfma_peak_tflops_shape(int device, int shape, void* stream) {
    if (hipSetDevice(device) != hipSuccess || (shape != 0 && shape != 1)) return -1.0;
    return mfma_peak_tflops(20, (hipStream_t)stream, shape);
}

double gp_mfma_lds_probe(int device, int reads_per_16_mfma, int waves_per_simd, int mode, void* stream) {
    if (hipSetDevice(device) != hipSuccess) return -1.0;
    return mfma_lds_probe_tflops(reads_per_16_mfma, waves_per_simd, mode, (hipStream_t)stream);
}

gp_status gp_softmax_rows(const float* in, void* out, int rows, int T, int ld, float scale, void* stream) {
    if (!in || !out || ld < T) return GP_ERR_INVALID;
    launch_softmax_rows(in, (h16_t*)out, rows, T, ld, scale, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_softmax_rows_f16(const void* in_f16, void* out, int rows, int T, int ld, float scale, void* stream) {
    if (!in_f16 || !out || ld < T || !softmax_rows_f16_supported(ld) || scale <= 0.f) return GP_ERR_INVALID;
    launch_softmax_rows_f16(in_f16, (h16_t*)out, rows, T, ld, scale, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

gp_status gp_bilinear(const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream) {
    if (!in || !out || (C % 8)) return GP_ERR_INVALID;
    launch_bilinear((const h16_t*)in, (h16_t*)out, B, Hi, Wi, Ho, Wo, C, align_corners, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? GP_OK : GP_ERR_HIP;
}

}  // extern "C"
