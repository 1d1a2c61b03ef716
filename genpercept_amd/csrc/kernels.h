// Host-side launchers of the gfx950 kernels (internal; the public C-ABI is include/genpercept_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#ifndef GP_F16
#define GP_F16 0  // 1: the library's 16-bit element is IEEE fp16 instead of bf16 (common.h)
#endif
typedef unsigned short h16_t;  // raw bits of one element (bf16 or fp16)

enum { GP_ACT_NONE = 0, GP_ACT_SILU = 1, GP_ACT_RELU = 2, GP_ACT_GEGLU = 3 };
enum { GP_BIAS_NONE = 0, GP_BIAS_COL = 1, GP_BIAS_ROW = 2 };

// Implicit-GEMM problem:  out[m][n] = act( sum_k A[m][k] * W[n][k] + bias ) (+ res[m][n])
//   rows m  = output pixels (b, oy, ox) of an NHWC tensor (or plain matrix rows when ks == 1)
//   cols n  = output channels; W is [n][tap][cin] (k-contiguous), i.e. "B transposed"
//   k       = (tap, cin), cin % 64 == 0
struct IGemmParams {
    const h16_t* in;     // NHWC input [B][Hi][Wi][Cin] (row stride lda elements when ks == 1)
    const h16_t* wt;     // [n_rows][taps][Cin]
    const float* bias;    // [Cout] (COL) or [M] (ROW) or nullptr
    const h16_t* res;    // residual [M][ldres] bf16 or nullptr
    void* out;            // bf16 or fp32 [M][ldo]
    const h16_t* zero;   // >= 256 bytes of zeros (source for padding / out-of-range rows)
    const float* in_scale;  // optional fused input transform x -> act(x * in_scale[b][c] + in_shift[b][c]) (GroupNorm apply);
    const float* in_shift;  //   only honoured by conv_halo.hip (callers check conv_uses_halo())
    int in_silu;
    float* stats_out;     // optional: per-(tile, output channel) {sum, sum of squares} of the STORED bf16 values, [tile_rows][N][2] fp32,
                          //   tile_rows = pixel tiles in launch order (igemm_tile_info); feeds the next GroupNorm without a read pass
    int M, N, Cin;        // N = valid output channels (before GEGLU halving)
    int n_rows;           // rows of `wt` that may be read (>= N; rows beyond read the zero page)
    int ks;               // 1 or 3
    int B, Hi, Wi, Ho, Wo;  // geometry (ks == 3, or ks == 1 with identical in/out geometry)
    int stride, pad_t, pad_l;
    int ups;              // 1: nearest-upsample the input to (Hu, Wu) before the conv
    int Hu, Wu;
    int lda, ldo, ldres;  // element strides of in (ks==1), out, res
    int ldw;              // element stride between rows of `wt` (taps*Cin for packed weights)
    int n_store;          // columns written (>= N_out; columns in [N_out, n_store) are written as 0)
    int out_fp32;         // 1: fp32 output, 2: fp16 output (direct epilogue path)
    int act, bias_mode;
    int dbg;              // ablation bits for profiling only (1: no DMA in the loop, 2: no MFMA work, 4: no waits/barriers)
    float* splitk_ws;     // caller-owned fp32 workspace for split-K partial sums (>= igemm_ksplit() * M * n_store floats) or nullptr: no split-K
    long long splitk_ws_floats;
    int batch;            // grid.y batches with the strides below (elements)
    int ksplit;           // > 1: grid.z = ksplit slices of the K loop, slice z writes fp32 partials at out + z * split_bs (set by launch_igemm)
    long long split_bs;
    long long in_bs, wt_bs, out_bs, res_bs, bias_bs;
    // Fused q | k | v projection of a self-attention (pgemm.hip only): output columns >= vt_col0 (a multiple of 128) are written TRANSPOSED,
    // vt_out[(b * (N - vt_col0) + (n - vt_col0)) * vt_Tpad + t] for row m = b * vt_T + t (vt_T % 16 == 0), instead of into `out` -- the layout
    // flash_attn64 reads its V^T operand in.  Columns [0, vt_col0) go to `out` as usual (n_store = ldo = vt_col0).  nullptr: plain GEMM.
    h16_t* vt_out;
    int vt_col0, vt_T, vt_Tpad;
    // x2-nearest-upsample 3x3 conv as four 2 x 2-tap phase convolutions (conv_halo.hip, PH): weights [n_rows][phase][2 x 2][Cin] with the kernel rows /
    // columns that fall onto the same source pixel summed (engine: pack_phases); nullptr: the nine-tap upsample kernel
    const h16_t* wt_ph;
    // contract precision (contract.hip): `res` points to an fp32 [M][ldres] tensor (only with out_fp32 == 1)
    int res_f32;
};

// A/B and profiling switches (GENPERCEPT_* environment variables).  Read from the environment in ONE place, gp_switches_reload(), which
// the C-ABI calls when an engine is created / finalised and at the per-kernel test entry points -- never on the launch path: launchers and
// the engine only read the cached struct (r3 had 27 getenv sites, several per conv launch, some cached per process and some not).
struct GpSwitches {
    int flash_ring3, no_flash512, f5_dbg, no_conv_few, no_conv_img, conv_img_s, no_cross_fold, no_gn_fusion, gn_fuse_max_slices,
        gn_fuse_below_px, no_stats_fusion, vt_tile, no_gn_small, fp32_scores, no_qkv_fuse, qkv_fuse_max_rows, no_rgb_conv, igemm_dbg, no_splitk,
        no_halo, no_pgemm, gn_apply_old, xfold_lds, no_fin_fuse, pgemm_ring3, gn_small_old, no_up_phases, c_no_flash;
};
const GpSwitches& gp_sw();
void gp_switches_reload();

// fp16 build: device address of each translation unit's saturation flag word (nullptr in the bf16 build); engine.hip collects them
void* gp_sat_flag_addr_igemm();
void* gp_sat_flag_addr_conv_halo();
void* gp_sat_flag_addr_pgemm();
void* gp_sat_flag_addr_conv_few();
void* gp_sat_flag_addr_norm();
void* gp_sat_flag_addr_attention();
void* gp_sat_flag_addr_elementwise();
#define GP_SAT_TUS {gp_sat_flag_addr_igemm, gp_sat_flag_addr_conv_halo, gp_sat_flag_addr_pgemm, gp_sat_flag_addr_conv_few, gp_sat_flag_addr_norm, gp_sat_flag_addr_attention, gp_sat_flag_addr_elementwise}

// tile_hint: 0 auto (halo conv / persistent GEMM / split-K / generic tile by heuristic), 1 = 128x128, 2 = 64x64, 3 = 256x32, 4 = 256x128,
//            5 = conv_halo.hip, 6 = 128x64, 7 = pgemm.hip
void launch_igemm(const IGemmParams& p, int tile_hint, hipStream_t s);
int igemm_ksplit(const IGemmParams& p, int tile_hint);   // K slices launch_igemm would like to use (1: none); needs p.splitk_ws to do so
// >64 KiB of dynamic LDS needs hipFuncSetAttribute once per (kernel, device).  `f` runs exactly once per (mask, current device); a second host
// thread that arrives meanwhile WAITS until it has finished (the r2 form set the "done" bit before the attribute call: another engine thread
// could launch in between and be refused for its LDS size).  The bit is published with release order after f(), read with acquire order.
std::mutex& gp_attr_mutex();
template <typename F>
inline void gp_once_per_device(unsigned long long* mask, F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { f(); return; }
    const unsigned long long bit = 1ull << dev;
    if (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & bit) return;
    std::lock_guard<std::mutex> lk(gp_attr_mutex());
    if (__atomic_load_n(mask, __ATOMIC_RELAXED) & bit) return;
    f();
    __atomic_store_n(mask, __atomic_load_n(mask, __ATOMIC_RELAXED) | bit, __ATOMIC_RELEASE);
}
// pgemm.hip: persistent GEMM for plain-row problems (ks == 1, bf16 out, column bias); tile_hint 7 forces it, 0 prefers it
bool pgemm_applicable(const IGemmParams& p);
int pgemm_bm(const IGemmParams& p);                 // rows per tile (256 or 128) launch_pgemm will use
void launch_pgemm(const IGemmParams& p, hipStream_t s);
bool igemm_uses_pgemm(const IGemmParams& p, int tile_hint);
// conv_halo.hip: 3x3 stride-1 convs on large maps (16x16-pixel tiles, input halo staged once per channel chunk); tile_hint 5
bool conv_halo_applicable(const IGemmParams& p);   // includes the Cin <= 2560 limit when in_scale is set
void launch_conv_halo(const IGemmParams& p, hipStream_t s);
bool conv_uses_halo(const IGemmParams& p, int tile_hint);
bool conv_halo_uses_phases(const IGemmParams& p);   // the x2-upsample conv will run as four phase convolutions (p.wt_ph set, exact x2, plain input)
int conv_halo_stat_rows(const IGemmParams& p);     // > 0: statistics rows per image (per-workgroup partials + pixel counts, mode 2)
// Statistics layout launch_igemm(p, tile_hint) will write: mode 0 = rows of BM consecutive pixels, mode 1 = 16x16 halo tiles per image,
// mode 2 = *bm rows per image, each with its own pixel count appended after the [rows][N][2] sums; returns the number of rows (callers
// allocate rows * (2 N + 1) floats), or 0 when that kernel path cannot produce stats_out (direct epilogue, GEGLU, fp32 output, ...).
int igemm_tile_info(const IGemmParams& p, int tile_hint, int* mode, int* bm);
// scale/shift from per-tile channel partials written by a conv epilogue (instead of launch_groupnorm_stats)
void launch_groupnorm_from_partials(const float* partials, int mode, int bm, int B, int H, int W, int C, int G, float eps, const float* gamma,
                                    const float* beta, float* scale, float* shift, hipStream_t s);

// GroupNorm over NHWC bf16 (fp32 statistics), optional fused SiLU.  Three passes: partial statistics, per-(image, channel)
// scale/shift, apply; the apply pass is skipped when the consuming conv fuses it (IGemmParams::in_scale).
// ws for launch_groupnorm: >= groupnorm_ws_floats() + 2*B*C floats.
void launch_groupnorm_stats(const h16_t* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, float* ws,
                            float* scale, float* shift, hipStream_t s);
void launch_groupnorm_apply(const h16_t* x, h16_t* y, const float* scale, const float* shift, int B, int HW, int C, int silu, hipStream_t s);
void launch_groupnorm(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps,
                      int silu, float* ws, hipStream_t s);
int groupnorm_ws_floats(int B, int HW, int C, int G);
// small maps (a group's HW x C/G block fits one workgroup's loop; (C / G) % 8 == 0): statistics + apply in one launch
bool groupnorm_small_applicable(int B, int HW, int C, int G);
void launch_groupnorm_small(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int silu,
                            hipStream_t s);

// LayerNorm over the last dim of [rows][C] bf16.
void launch_layernorm(const h16_t* x, h16_t* y, const float* gamma, const float* beta, int rows, int C, float eps, hipStream_t s);

// Flash self-attention, head_dim 64.  q/k: [B][T][ld] bf16 (head h at column h*64), vt: [B][heads*64][Tpad] bf16,
// out [B][T][ldo].
void launch_flash_attn64(const h16_t* q, const h16_t* k, const h16_t* vt, h16_t* out, int B, int T, int heads,
                         int ldq, int ldk, int Tpad, int ldo, hipStream_t s);
// VAE mid-block attention (one head, head_dim 512), fused; ws: flash_attn512_workspace_floats() floats (may be null when that is 0)
bool flash_attn512_supported(int C);
long long flash_attn512_workspace_floats(int B, int T, int ncu);
void launch_flash_attn512(const h16_t* q, const h16_t* k, const h16_t* vt, h16_t* out, float* ws, int B, int T, int ldq,
                          int ldk, int Tpad, int ldo, float scale, int ncu, hipStream_t s);

// Cross-attention against a small constant context: q [rows][C] bf16, kc/vc [L][C] fp32, head_dim 64.
void launch_cross_attn_small(const h16_t* q, const float* kc, const float* vc, h16_t* out, int rows, int C, int L, hipStream_t s);

// Cross-attention against a 2-token constant context folded into per-head vectors (norm.hip): y_out = y + c0 + sum_h sigmoid(LNhat(y) . U[h]
// + u0[h]) G[h]; optionally n3_out = LayerNorm(y_out; g3, b3).  U, G: [heads][C] fp32; u0 [heads]; c0, g3, b3 [C].  C <= 1536.
bool cross_attn_fold_supported(int C, int heads);  // C == 64 * heads, heads in {1, 2, 4, 5, 10, 20} (the instantiated head counts)
void launch_cross_attn_fold(const h16_t* y, h16_t* y_out, h16_t* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                            const float* g3, const float* b3, int rows, int C, int heads, float eps, hipStream_t s);

// Row softmax: in fp32 [rows][ld] (first T columns valid) -> bf16 [rows][ld], columns >= T written as 0.
void launch_softmax_rows(const float* in, h16_t* out, int rows, int T, int ld, float scale, hipStream_t s);
bool softmax_rows_f16_supported(int ld);
void launch_softmax_rows_f16(const void* in_f16, h16_t* out, int rows, int T, int ld, float scale, hipStream_t s);  // fp16 logits

// Elementwise / layout kernels
void launch_rgb_prologue(const void* rgb, int is_u8, h16_t* out, int B, int H, int W, int Cpad, hipStream_t s);  // NCHW -> NHWC, x/255*2-1
// fused RGB prologue + VAE-encoder conv_in (3 -> Cout, Cout % 32 == 0) + GroupNorm partial statistics (16x16 tiles) of the output
int rgb_conv_in_rows(int B, int H, int W);  // statistics rows per image launch_rgb_conv_in writes (mode 2: sums + pixel counts)
void launch_pack_k27(const h16_t* wt, int ldw, int Cout, h16_t* w27, hipStream_t s);  // [rows][9][64] conv layout -> compact [Cout][32]
void launch_rgb_conv_in(const void* rgb, int is_u8, const h16_t* w27, const float* bias, h16_t* out, float* stats, int B, int H, int W, int Cout,
                        hipStream_t s);
void launch_concat(const h16_t* a, int Ca, const h16_t* b, int Cb, h16_t* out, long long pixels, hipStream_t s);
int concat_stats_bm(long long hw, long long pixels, int channels);  // pixels per statistics tile launch_concat_stats can use for an image of hw pixels (0: none)
void launch_concat_stats(const h16_t* a, int Ca, const h16_t* b, int Cb, h16_t* out, long long pixels, int bm, float* part, hipStream_t s);
struct DdimCoef { float x0_sample, x0_model, eps_sample, eps_model, prev_x0, prev_eps, clip; };
void launch_ddim_init(const float* noise_nchw, h16_t* lat, float* sample, int B, int H, int W, int L, int ld, int off, hipStream_t s);
void launch_ddim_step(const h16_t* model, int ldm, float* sample, h16_t* uin, int ldu, int off, h16_t* x0_out, int ldx, long long pixels, int L,
                      const DdimCoef& k, hipStream_t s);
void launch_nchw_f32_to_nhwc(const float* in, h16_t* out, int B, int C, int H, int W, int Cpad, hipStream_t s);
void launch_nhwc_to_nchw_f32(const h16_t* in, float* out, int B, int C, int H, int W, int ld, hipStream_t s);
void launch_decode_epilogue(const h16_t* in, float* out, int B, int H, int W, int ld, int mean3, int raw, hipStream_t s);  // mean, clip, (x+1)/2
void launch_scale_pad(const h16_t* in, h16_t* out, long long pixels, int C, int ldi, int ldo, float scale, hipStream_t s);
void launch_pointwise_small(const h16_t* in, h16_t* out, const float* w, const float* bias, long long pixels, int Cin, int Cout, int ldi,
                            int ldo, float in_scale, hipStream_t s);  // 1x1 conv for Cin, Cout <= 8 (post_quant_conv)
void launch_relu(const h16_t* in, h16_t* out, long long n, hipStream_t s);
void launch_add(const h16_t* a, const h16_t* b, h16_t* out, long long n, hipStream_t s);
void launch_bilinear(const h16_t* in, h16_t* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, hipStream_t s);
void launch_dpt_final(const h16_t* in, const float* w, float bias, float* out, int B, int HW, int Cin, hipStream_t s);  // ReLU'd 32ch -> 1
void launch_minmax_norm(float* x, int B, long long n, float* ws, hipStream_t s);  // per-image (x-min)/(max-min)

// conv_img.hip: 3x3 stride-1 convs whose maps tile into 576-pixel units (24x24, 12x12 x 4): whole-image tiles, K split over workgroups,
// fp32 partial sums [S][M][N] into `part` (then splitk_reduce_kernel)
bool conv_img_applicable(const IGemmParams& p);
int conv_img_ksplit(const IGemmParams& p);
void launch_conv_img(const IGemmParams& p, float* part, int S, hipStream_t s);

// conv_few.hip: the VAE decoder's tail fused -- GroupNorm apply (+ SiLU) of the input, conv3x3 128 -> 3, [mean of the 3 channels], clip / shift
// unless raw, fp32 NCHW out (genpercept_pipeline.py:521-525,469-472).  scale / shift: the [B][128] affine form launch_groupnorm_* produce.
bool conv_few_applicable(int Cin, int Cout, int H, int W);
void launch_conv_few(const h16_t* in, const h16_t* wt, const float* bias, const float* scale, const float* shift, const h16_t* zero, float* out, int B,
                     int H, int W, int silu, int mean3, int raw, int ncu, hipStream_t s);

// prepost.hip: device-side pre / post processing of GenPerceptPipeline.__call__ (resize with torchvision semantics, colour map, quantisation)
void launch_resize(const void* in, void* out, float* tmp, long long planes, int Hi, int Wi, int Ho, int Wo, int mode, int u8, int clip01, hipStream_t s);
void launch_clip01(const float* in, float* out, long long n, hipStream_t s);
void launch_normalize_rgb(const float* in, float* out, long long n, hipStream_t s);
void launch_colorize_lut(const float* x, const unsigned char* lut, unsigned char* rgb, long long n, hipStream_t s);
void launch_quantize(const float* x, void* q, long long n, int bits, hipStream_t s);

// contract.hip: the kernels between the matrix products of the contract-precision mode (fp32 storage, split-bf16 MFMA operands:
// A order [hi | lo | hi], B order [hi | hi | lo] over a tripled K).  Split tensors have 3 * C elements per row.
void launch_c_split3(const float* x, int ldx, h16_t* out, long long rows, int C, int b_order, int act, float scale, hipStream_t s);
void launch_c_rgb_split(const void* rgb, int is_u8, h16_t* out, int B, int H, int W, hipStream_t s);  // -> [B*H*W][192]
int c_gn_stat_rows(int HW, int C, int* bm_out);  // statistics rows per image of launch_c_gn_stats ("mode 2" partials: B * R * (2 C + 1) floats)
void launch_c_gn_stats(const float* x, float* part, int B, int HW, int C, hipStream_t s);
void launch_c_gn_apply_split(const float* x, h16_t* out, const float* scale, const float* shift, int B, int HW, int C, int silu, hipStream_t s);
void launch_c_layernorm_split(const float* x, h16_t* out, const float* gamma, const float* beta, int rows, int C, float eps, hipStream_t s);
void launch_c_concat(const float* a, int Ca, const float* b, int Cb, float* out, long long pixels, hipStream_t s);
void launch_c_heads_split(const float* qkv, int ld, h16_t* Qs, h16_t* Ks, h16_t* Vts, int B, int T, int Tpad, int heads, int hd, hipStream_t s);
void launch_c_qkv_planes(const float* qkv, int ld, h16_t* qk_hi, h16_t* qk_lo, h16_t* vt_hi, h16_t* vt_lo, int B, int T, int Tpad, int heads, int hd,
                         hipStream_t s);
// attention.hip: flash attention (head_dim 64) over split operands; out = A-order split operand [B*T][3 * heads * 64]
void launch_flash_attn64_split(const h16_t* qk_hi, const h16_t* qk_lo, const h16_t* vt_hi, const h16_t* vt_lo, h16_t* out, int B, int T, int heads, int ld,
                               int Tpad, hipStream_t s);
bool c_softmax_split_supported(int ld);
void launch_c_softmax_split(const float* in, h16_t* out, long long rows, int T, int ld, float scale, hipStream_t s);
void launch_c_heads_merge_split(const float* O, h16_t* out, int B, int T, int heads, int hd, hipStream_t s);
bool c_cross_fold_supported(int C);
void launch_c_cross_fold(const float* y, float* y_out, h16_t* n3_out, const float* U, const float* u0, const float* G, const float* c0, const float* g3,
                         const float* b3, int rows, int C, int heads, float eps, hipStream_t s);
void launch_c_cross_attn_small(const float* q, const float* kc, const float* vc, h16_t* out, int rows, int C, int L, hipStream_t s);
void launch_c_pointwise_small(const float* in, float* out, const float* w, const float* bias, long long pixels, int Cin, int Cout, int ldi, int ldo,
                              float in_scale, hipStream_t s);
void launch_c_decode_epilogue(const float* in, float* out, int B, int H, int W, int ld, int mean3, int raw, hipStream_t s);
void launch_c_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int Cpad, hipStream_t s);
void launch_c_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, int ld, hipStream_t s);
void launch_c_add(const float* a, const float* b, float* out, long long n, hipStream_t s);
void launch_c_bilinear(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, hipStream_t s);
void launch_c_dpt_final(const float* in, const float* w, float bias, float* out, int B, int HW, int Cin, hipStream_t s);
void launch_c_ddim_init(const float* noise_nchw, float* lat, float* sample, int B, int H, int W, int L, int ld, int off, hipStream_t s);
void launch_c_ddim_step(const float* model, int ldm, float* sample, float* uin, int ldu, int off, float* x0_out, int ldx, long long pixels, int L,
                        const DdimCoef& k, hipStream_t s);

// microbench.hip: sustained TFLOP/s of back-to-back v_mfma_f32_32x32x16 on this chip (register operands, all CUs), or < 0 on error
double mfma_peak_tflops(int ms_target, hipStream_t s, int shape = 0);
double mfma_lds_probe_tflops(int reads_per_16_mfma, int waves_per_simd, int mode, hipStream_t s);  // shape 0: v_mfma_f32_32x32x16, 1: 16x16x32
