/* genpercept_hip.h — C-ABI of libgenpercept_hip.so: the MI355X (gfx950) engine behind GenPercept's one-step
 * inference path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * What each entry point replaces in the reference (/root/reference, aim-uofa/GenPercept):
 *   gp_infer          GenPerceptPipeline.single_infer               genpercept/genpercept_pipeline.py:375-486
 *   gp_vae_encode     GenPerceptPipeline.encode_rgb                 genpercept/genpercept_pipeline.py:488-505
 *                       (diffusers AutoencoderKL.encoder + quant_conv, call site :500-501)
 *   gp_unet           self.unet(...) / CustomUNet2DConditionModel.forward
 *                                                                   genpercept_pipeline.py:455-457,476-479; models/custom_unet.py:34-427
 *                       (+ scheduler.step == "-model_output", genpercept_pipeline.py:460-465, ddim.py:166-204)
 *   gp_vae_decode     GenPerceptPipeline.decode_pred                genpercept/genpercept_pipeline.py:507-526
 *   gp_dpt_head       DPTNeckHeadForUnetAfterUpsampleIdentity.forward  genpercept/models/dpt_head.py:443-560,585-593
 *   gp_load_tensor    from_pretrained / load_state_dict of the diffusers-layout checkpoints   run.py:296-357
 *   gp_set_context    encode_text + text_embed cache                genpercept_pipeline.py:360-372,425-429
 *   gp_set_timestep   scheduler.set_timesteps / fix_timesteps       genpercept_pipeline.py:403-408
 *   gp_infer_steps    single_infer for archs marigold / rgb_blending: the denoising loop with the DDIM update
 *                                                                   genpercept_pipeline.py:413-422,447-465; ddim.py:144-217; run.py:361-368
 * The per-kernel entry points (gp_conv2d ... gp_bilinear) exist for the parity tests; they are the same launchers the
 * engine uses.
 *
 * Ownership: the engine owns weights, folded constants and its activation pool.  The caller owns every in/out DEVICE
 * buffer (e.g. PyTorch-ROCm tensors passed by data_ptr()) and keeps them alive until the stream has been synchronised.
 * Errors: every call returns gp_status; gp_last_error() gives the message.  Nothing throws across the ABI.
 * Threading: one engine per GPU; an engine is not re-entrant; different engines (same or different GPUs) may run on different host
 *   threads concurrently: every workspace an engine call touches (activation pool, split-K partial sums, GroupNorm statistics, min-max
 *   partials) belongs to that engine, and per-device kernel attributes are set per device.  An engine recycles activation buffers in
 *   stream order; if consecutive calls pass different streams, the engine synchronises the previous stream first.  A call that fails
 *   returns every buffer it had taken to the engine's pool.  The per-kernel entry points (parity-test interface) share one scratch
 *   set per device and serialise on a process-wide mutex while they enqueue.
 */
#ifndef GENPERCEPT_HIP_H
#define GENPERCEPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gp_engine gp_engine;

typedef enum { GP_OK = 0, GP_ERR_INVALID = 1, GP_ERR_HIP = 2, GP_ERR_MISSING_WEIGHT = 3, GP_ERR_STATE = 4 } gp_status;
typedef enum { GP_DT_F32 = 0, GP_DT_F16 = 1, GP_DT_BF16 = 2 } gp_dtype;
/* depth / matting / dis / disparity average the 3 decoder channels (genpercept_pipeline.py:523-525); normal / seg keep 3 */
typedef enum { GP_MODE_DEPTH = 0, GP_MODE_NORMAL = 1, GP_MODE_SEG = 2, GP_MODE_MATTING = 3, GP_MODE_DIS = 4, GP_MODE_DISPARITY = 5 } gp_mode;

typedef struct gp_config {
    int device;                 /* HIP device ordinal */
    /* UNet2DConditionModel (SD2.1 unet/config.json) */
    int unet_in_channels, unet_out_channels;
    int unet_block_out[4];
    int unet_num_heads[4];      /* `attention_head_dim` of the config = number of heads; head_dim must be 64 */
    int unet_down_attn[4];      /* 1 = CrossAttnDownBlock2D, 0 = DownBlock2D */
    int unet_layers_per_block;
    int unet_cross_dim;
    int unet_has_out;           /* 0 for DPT-head UNets (conv_norm_out / conv_out deleted, run.py:316-318) */
    float unet_norm_eps;
    /* AutoencoderKL (SD2.1 vae/config.json) */
    int vae_block_out[4];
    int vae_layers_per_block;
    int vae_latent_channels;
    float vae_norm_eps;
    float vae_scaling_factor;   /* GenPerceptPipeline.latent_scale_factor = 0.18215 */
    /* DPT neck+head (hf_configs/dpt-sd2.1-unet-after-upsample-general/config.json); dpt_enabled = 0 -> VAE-decoder head */
    int dpt_enabled;
    int dpt_neck[4];
    int dpt_fusion;
    int norm_groups;            /* 32 */
} gp_config;

typedef struct gp_timings {
    float ms_encode, ms_unet, ms_head, ms_total;   /* last gp_infer, valid when profiling level >= 1 */
    double flops_igemm, flops_attn;                /* algorithmic flops issued since gp_reset_timings */
    float ms_igemm, ms_attn;                       /* summed kernel time of those launches (profiling level 2) */
    int n_igemm, n_attn, n_launches;
    double flops_halo;                             /* the dominant kernel alone: conv3x3_halo3_kernel (a subset of the igemm figures) */
    float ms_halo;
    int n_halo;
} gp_timings;   /* (frozen layout: new counters get their own entry point, e.g. gp_saturation_events, never a field appended here) */

void gp_default_config(gp_config* cfg);                                  /* SD2.1 values */
gp_status gp_create(const gp_config* cfg, gp_engine** out);
void gp_destroy(gp_engine* e);
const char* gp_last_error(const gp_engine* e);
const char* gp_version(void);
/* Bumped whenever a struct layout or an existing signature changes (additions do not bump it).  3: gp_timings lost its trailing `sat_events`
 * field in round 5 (gp_saturation_events replaced it) -- a client built against the older header must be rebuilt; it can check this at load time. */
#define GP_ABI_VERSION 3
int gp_abi_version(void);
/* Storage / MFMA-operand element of THIS library: GP_DT_BF16 (libgenpercept_hip.so) or GP_DT_F16 (libgenpercept_hip_f16.so, the
 * reference's half precision: run.py --half_precision / torch_dtype=torch.float16).  Per-kernel entry points take and return
 * tensors of this element type; stage-level entry points are fp32 at the boundary in both libraries. */
gp_dtype gp_element_dtype(void);

/* Arithmetic of the engine (call before gp_finalize; the weights are packed per precision):
 *   GP_PREC_NATIVE    16-bit storage and MFMA operands in this library's element type (the default; BASELINE.json's bf16 when this is
 *                     libgenpercept_hip.so, the reference's --half_precision when it is libgenpercept_hip_f16.so).
 *   GP_PREC_CONTRACT  what torch_dtype=float32 -- the reference's default, run.py:273-281 -- asks for: every STORED activation is fp32 and
 *                     every matrix product runs on the bf16 matrix cores with split operands (x = hi + lo, three MFMAs per product: hi.hi +
 *                     lo.hi + hi.lo, fp32 accumulation; csrc/contract.hip).  About 2^-16 relative per product: final maps within 1e-3 of the
 *                     fp32 path under both the mean-absolute and the relative-RMS reading, at roughly a quarter of the native throughput.
 *                     bf16 library only (GP_ERR_INVALID in the fp16 library). */
typedef enum { GP_PREC_NATIVE = 0, GP_PREC_CONTRACT = 1 } gp_precision;
gp_status gp_set_precision(gp_engine* e, gp_precision prec);
gp_precision gp_get_precision(const gp_engine* e);

/* One call per state-dict entry.  `name` = "<module>.<diffusers key>" with module in {vae, unet, dpt}.  The engine copies
 * (and converts to fp32); the caller keeps ownership of host_ptr. */
gp_status gp_load_tensor(gp_engine* e, const char* name, const void* host_ptr, const int64_t* shape, int ndim, gp_dtype dtype);
/* Text-encoder output for the prompt: embed is HOST fp32 [L][D] (L = 2 for the empty prompt). */
gp_status gp_set_context(gp_engine* e, const float* embed, int L, int D);
gp_status gp_set_timestep(gp_engine* e, float t);                        /* default 1 */
/* Fold constants (time embedding, cross-attention K/V, quant_conv), pack weights to bf16 device layouts, free host copies. */
gp_status gp_finalize(gp_engine* e);

/* rgb_dev: DEVICE [B][3][H][W], uint8 0..255 (is_u8 = 1) or fp32 already in [-1,1] (is_u8 = 0).
 * out_dev: DEVICE fp32 [B][C][8h][8w], C = 1 or 3 by mode, values in [0,1]; (h, w) = gp_latent_size(H), gp_latent_size(W)
 * (== H/8, W/8 when H, W are multiples of 8).  With the DPT head: [B][1][gp_dpt_out_size(h)][gp_dpt_out_size(w)].  stream: hipStream_t (NULL = default). */
int gp_latent_size(int pixels);   /* three VAE downsamples: x -> (x - 2) / 2 + 1 */
int gp_dpt_out_size(int latent);  /* DPT head output edge: 32 * (two UNet downsamples of the latent edge) == 8 * latent when latent % 4 == 0 */
gp_status gp_infer(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, gp_mode mode, float* out_dev, void* stream);

/* The multi-step archs (VAE-decoder head only).  One scheduler step in affine form (what diffusers' DDIMScheduler.step computes for
 * epsilon / sample / v_prediction with eta = 0; the host derives the numbers, genpercept_amd/scheduler.py):
 *   x0 = clip(x0_sample * s + x0_model * m, +-clip)   (clip <= 0: none),   eps = eps_sample * s + eps_model * m,
 *   s' = prev_x0 * x0 + prev_eps * eps,               m = UNet(input_t, timestep), s = current sample.
 * noise_dev != NULL (marigold): DEVICE fp32 [B][L][h][w] initial sample; the UNet reads [rgb_latent, sample] (unet_in_channels == 2L).
 * noise_dev == NULL (rgb_blending): the sample starts as the rgb latent and is itself the UNet input (unet_in_channels == L).
 * The result is the LAST step's x0 (pred_original_sample, :465), decoded / clipped / shifted exactly like gp_infer.  The engine's
 * timestep (gp_set_timestep) is restored on return; the first use of a new timestep value folds its time embedding on the host (tens
 * of ms, once per engine and value), later uses are one device copy. */
typedef struct gp_ddim_step {
    float timestep;
    float x0_sample, x0_model, eps_sample, eps_model, prev_x0, prev_eps, clip;
} gp_ddim_step;
gp_status gp_infer_steps(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, gp_mode mode, const gp_ddim_step* steps,
                         int n_steps, const float* noise_dev, float* out_dev, void* stream);

/* Stage-level entry points (DEVICE fp32 NCHW in/out). */
gp_status gp_vae_encode(gp_engine* e, const void* rgb_dev, int is_u8, int B, int H, int W, float* latent_out, void* stream);
gp_status gp_unet(gp_engine* e, const float* latent_in, int B, int h, int w, float* sample_out /* may be NULL */,
                  float* const* feats_out /* 4 pointers or NULL; multi_level_feats order */, void* stream);
gp_status gp_vae_decode(gp_engine* e, const float* pred_latent, int B, int h, int w, int mean3, float* out, void* stream);
/* The VAE mid-block attention alone (AutoencoderKL mid_block.attentions[0]: GroupNorm, 1 head x C, +residual; call sites
 * genpercept_pipeline.py:500-501 / :521-522); decoder = 0: the encoder's, 1: the decoder's.  x, out: DEVICE fp32 [B][C][h][w]. */
gp_status gp_vae_mid_attention(gp_engine* e, int decoder, const float* x, int B, int h, int w, float* out, void* stream);
gp_status gp_dpt_head(gp_engine* e, const float* const* feats /* reversed multi_level_feats: [c0@h, c1@h, c2@h/2, c3@h/4] */,
                      int B, int h, int w, float* out /* [B][gp_dpt_out_size(h)][gp_dpt_out_size(w)], not normalised */, void* stream);

/* Profiling: level 0 off, 1 per-stage hipEvents, 2 additionally per-launch events on the MFMA kernels. */
gp_status gp_set_profile(gp_engine* e, int level);
gp_status gp_get_timings(gp_engine* e, gp_timings* out);
/* Saturation is never silent in the fp16 library (libgenpercept_hip_f16.so; the reference's --half_precision, run.py:273-281, overflows to
 * inf / NaN on the same activations): conversions saturate at +-65504, and every call in which one actually clipped is counted here --
 * events = (call, kernel file) pairs since the last reset; 0 means no stored activation left the fp16 range.  The bf16 library (fp32
 * range) always reports 0.  Synchronises the engine's stream.  The Python pipeline logs a warning when a call raised events. */
gp_status gp_saturation_events(gp_engine* e, long long* events, int reset);
gp_status gp_reset_timings(gp_engine* e);
/* executed (not algorithmic) MFMA flops of the halo-conv launches since gp_reset_timings: equals gp_timings.flops_halo except for launches that do
 * less arithmetic than their algorithmic count (the x2-upsample convs as four 2x2-tap phase convolutions: 4/9) */
gp_status gp_halo_executed_flops(gp_engine* e, double* flops);
/* Profiling level 3: text log of the last gp_infer, one line "ms<TAB>algorithmic flops<TAB>description" per kernel launch
 * (ms = start-to-next-start on the stream: kernel time plus the gap behind it).  Returns the bytes needed (incl. NUL). */
int gp_get_launch_log(gp_engine* e, char* buf, int cap);

/* ---- pre / post processing of GenPerceptPipeline.__call__ on the device (DEVICE pointers, NCHW) ---------------------------------
 * resample (genpercept/util/image_util.py:108-126): 0 = bilinear with antialias (torchvision resize(..., antialias=True): ATen's separable
 * triangle filter), 1 = nearest-exact, 2 = bicubic with antialias (the same index ranges, Keys cubic a = -0.5; uint8 results clamped).
 * tmp: DEVICE fp32 scratch for the separable passes, >= B * C * H_in * W_out floats (unused for nearest-exact). */
/* new size of resize_max_res (genpercept/util/image_util.py:75-105): longest edge -> max_edge, int() truncation */
void gp_resize_max_res_size(int H0, int W0, int max_edge, int* h, int* w);
/* resize_max_res on the uint8 RGB [B][3][H0][W0] -> [B][3][h][w] uint8 (fp32 interpolation, round half to even; genpercept_pipeline.py:236-242) */
gp_status gp_preprocess(const void* rgb_u8, int B, int H0, int W0, void* out_u8, int h, int w, int resample, float* tmp, void* stream);
/* The same for FLOAT images (a float `input_image` tensor, e.g. the trainer's `rgb_int`, genpercept_trainer.py:1151-1165): fp32 [B][C][H0][W0]
 * -> [B][C][h][w] fp32, no rounding and no clamp (torchvision's float path; skipped when the sizes are equal), then with normalize != 0
 * x / 255 * 2 - 1 in that operation order (genpercept_pipeline.py:245) -- the [-1, 1] image gp_infer takes with is_u8 = 0. */
gp_status gp_preprocess_f32(const float* rgb, int B, int C, int H0, int W0, float* out, int h, int w, int resample, int normalize, float* tmp,
                            void* stream);
/* genpercept_pipeline.py:301-329 + run.py:449-455: pred fp32 [B][C][h][w] -> resize to (Ho, Wo) (skipped when equal) -> clip to [0, 1]
 * -> pred_out fp32 [B][C][Ho][Wo]; optionally colored_out uint8 [B][Ho][Wo][3] through the 256 x 3 byte colour LUT lut_dev (C == 1), and
 * q_out = (pred_out * 65535).astype(uint16) (q_bits 16) or (pred_out * 255).astype(uint8) (q_bits 8), [B][C][Ho][Wo]. */
gp_status gp_postprocess(const float* pred, int B, int C, int h, int w, float* pred_out, int Ho, int Wo, int resample, float* tmp,
                         const unsigned char* lut_dev, void* colored_out, void* q_out, int q_bits, void* stream);

/* Sustained TFLOP/s of back-to-back v_mfma_f32_32x32x16 (this library's element type) on every CU of `device`: the chip's own MFMA
 * peak under load, reported by bench.py beside the nominal 2.5 PFLOP/s.  < 0 on error. */
double gp_mfma_peak_tflops(int device, void* stream);
/* The same for one MFMA shape: 0 = v_mfma_f32_32x32x16 (the attention kernels), 1 = v_mfma_f32_16x16x32 (the conv / GEMM kernels; K = 32 per
 * instruction moves a quarter of the accumulator registers per flop).  The chip runs against its power budget, so the two sustain different
 * clocks on the same data. */
double gp_mfma_peak_tflops_shape(int device, int shape, void* stream);
/* Measurement probe (r4): TFLOP/s of the conv / GEMM inner loop in isolation -- per wave and iteration 16 independent
 * v_mfma_f32_16x16x32 plus `reads_per_16_mfma` (0, 2, 4, 8 or 16) conflict-free ds_read_b128 refilling the other fragment set -- at
 * `waves_per_simd` (1, 2 or 4) waves per SIMD on every CU.  mode 0: nothing else (how much of a fragment read's LDS -> register return overlaps
 * with matrix work on the same SIMD); mode bits (8 reads, 2 waves per SIMD only) add the real kernels' other ingredients per 32-MFMA step:
 * 1 = one s_barrier, 2 = the weight stream (16 KiB per workgroup by LDS-DMA into a 3-deep ring, counted vmcnt), 4 = weight fragments read from
 * that ring, 8 = the DMA as buffer_load ... lds, 16 = the halo stream (48 KiB per workgroup every ninth step from HBM).  < 0 on error or an
 * unsupported combination (csrc/microbench.hip lists them). */
double gp_mfma_lds_probe(int device, int reads_per_16_mfma, int waves_per_simd, int mode, void* stream);

/* ---- per-kernel entry points (DEVICE pointers, bf16 NHWC activations) -------------------------------------------- */
/* Pack an OIHW fp32 HOST weight into the device layout [n_rows][taps][cin_pad] bf16 (n_rows = gp_packed_rows(cout)). */
int gp_packed_rows(int cout);
gp_status gp_pack_weight(const float* w_oihw_host, int cout, int cin, int ks, int cin_pad, int geglu, void* dev_out);
/* x2-nearest-upsample 3x3 conv (Upsample2D, resnet.py of diffusers: F.interpolate(scale_factor=2, mode="nearest") then conv; call sites
 * custom_unet.py:372-400 up blocks, VAE decoder up blocks) as four 2 x 2-tap phase convolutions on the source map: gp_pack_weight_phases sums the kernel
 * rows / columns that fall onto the same source pixel ([rows][phase 2a+b][tap 2ty+tx][cin_pad], rows = gp_packed_rows(cout)); gp_conv2d_up2 runs
 * the phase kernel (w_packed = the ordinary 3x3 packing of the same weight, for the launcher's checks).  Test entry points. */
gp_status gp_pack_weight_phases(const float* w_oihw_host, int cout, int cin, int cin_pad, void* dev_out);
gp_status gp_conv2d_up2(const void* in, const void* w_packed, const void* w_phases, const float* bias, const void* residual, void* out, int B, int Hi,
                        int Wi, int Cin, int Cout, void* stream);
/* the same with the GroupNorm statistics of the tensor it writes (the phase kernel's per-(tile, phase) partial sums and pixel counts, what the VAE
 * decoder's upsampler convs leave for the next resnet's norm1), finalised to scale / shift like gp_conv2d_stats.  Test entry point. */
gp_status gp_conv2d_up2_stats(const void* in, const void* w_packed, const void* w_phases, const float* bias, const void* residual, void* out, int B, int Hi,
                              int Wi, int Cin, int Cout, const float* gamma, const float* beta, int groups, float eps, float* scale_out, float* shift_out,
                              void* stream);
gp_status gp_conv2d(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int Hi, int Wi, int Cin,
                    int Cout, int ks, int stride, int pad_t, int pad_l, int Ho, int Wo, int ups_h, int ups_w, int act, int n_store,
                    int out_fp32, int tile_hint, void* stream);
/* conv3x3(act(GroupNorm(in))) with the normalisation applied inside the conv kernel (statistics pass + fused apply);
 * stride 1, pad 1, optional nearest x2 upsample of the normalised input; H, W >= 16 (x2: >= 8). */
gp_status gp_conv2d_gn(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int H, int W, int Cin,
                       int Cout, int ups, int act, const float* gamma, const float* beta, int groups, float eps, int silu, void* stream);
/* VAE-encoder conv_in fused with the RGB prologue of GenPerceptPipeline.__call__ (genpercept_pipeline.py:245, x / 255 * 2 - 1) :
 * rgb [B][3][H][W] on the device (uint8 when is_u8, else float already in [-1,1]) -> conv3x3(pad 1, 3 -> Cout) + bias -> NHWC bf16.
 * w_packed: gp_pack_weight(.., cout, 3, 3, 64, 0, ..).  Cout % 32 == 0. */
gp_status gp_rgb_conv_in(const void* rgb, int is_u8, const void* w_packed, const float* bias, void* out, int B, int H, int W, int Cout, void* stream);
/* conv (3x3 pad 1 or 1x1, stride 1, optional nearest x2 upsample) whose epilogue also leaves the GroupNorm statistics of the tensor it
 * writes (per-tile channel sums), finalised to the affine form the next GroupNorm applies: scale[b][c] = gamma[c] * rstd[b][g(c)],
 * shift[b][c] = beta[c] - mean[b][g(c)] * scale[b][c].  This is how the engine skips the statistics read pass of
 * diffusers' ResnetBlock2D.norm2 / the following block's norm1 (resnet.py forward: norm -> nonlinearity -> conv).
 * Returns GP_ERR_INVALID when the selected kernel cannot produce statistics for this shape. */
gp_status gp_conv2d_stats(const void* in, const void* w_packed, const float* bias, const void* residual, void* out, int B, int H, int W, int Cin,
                          int Cout, int ks, int ups, int tile_hint, const float* gamma, const float* beta, int groups, float eps,
                          float* scale_out, float* shift_out, void* stream);
/* out[M][N] = A[M][K] * Bt[N][K]^T (+bias per column / per row), batched over `batch` with element strides. */
gp_status gp_gemm(const void* a, int lda, const void* bt, int ldb, const float* bias, int bias_mode, const void* residual, int ldres, void* out,
                  int ldo, int M, int N, int K, int n_rows_bt, int n_store, int act, int out_fp32 /* 0 bf16, 1 fp32, 2 fp16 */, int batch, long long a_bs, long long bt_bs,
                  long long out_bs, int tile_hint, void* stream);
/* The VAE decoder's tail in one kernel (diffusers AutoencoderKL.decoder conv_norm_out -> SiLU -> conv_out, call site
 * genpercept_pipeline.py:521-522; channel mean :523-525; clip / shift :469-472): in NHWC [B][H][W][128] (this library's element type),
 * w_packed = gp_pack_weight(conv_out.weight [3][128][3][3]), GroupNorm(groups, eps) statistics computed here, out fp32 NCHW
 * [B][mean3 ? 1 : 3][H][W]; raw = 1: no clip / shift (decode_pred alone).  Cin must be 128 (GP_ERR_INVALID otherwise). */
gp_status gp_decoder_tail(const void* in, const void* w_packed, const float* bias, const float* gamma, const float* beta, int groups, float eps,
                          int B, int H, int W, int Cin, int mean3, int raw, float* out, void* stream);
/* The self-attention input projections of a BasicTransformerBlock (attn1.to_q / to_k / to_v, bias-free in SD2.1; custom_unet.py's
 * Transformer2DModel blocks) as ONE GEMM over the stacked weight [3C][K] (gp_pack_weight of the concatenation): q | k go to qk_out
 * [B*T][2C] row-major, V goes to vt_out TRANSPOSED as [B][C][Tpad] (zero beyond T) -- the operand layout of gp_flash_attention.
 * Requirements: K % 64 == 0, (2C) % 128 == 0, T % 16 == 0, Tpad % 8 == 0, B*T >= 256; GP_ERR_INVALID otherwise (use gp_gemm twice). */
gp_status gp_gemm_qkv(const void* a, int lda, const void* w_packed, int ldw, int n_rows_w, int K, void* qk_out, void* vt_out, int B, int T, int C,
                      int Tpad, void* stream);
gp_status gp_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int silu, void* stream);
gp_status gp_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps, void* stream);
gp_status gp_flash_attention(const void* q, const void* k, const void* vt, void* out, int B, int T, int heads, int ldq, int ldk, int Tpad, int ldo,
                             void* stream);
/* Contract precision (gp_set_precision; csrc/contract.hip, attention.hip: flash_attn64_split_kernel): the UNet's self-attention core over SPLIT operands.
 * qkv: DEVICE fp32 [B*T][ld] with q | k | v at columns 0 | C | 2C (C = heads * 64; the output of the stacked attn1.to_q / to_k / to_v projection);
 * out_split: DEVICE 16-bit [B*T][3C] = the A-order split operand [hi | lo | hi] of softmax(q k^T / 8) v that the output projection reads (value = hi + lo).
 * bf16 library only.  Synchronises the stream (test entry point). */
gp_status gp_flash_attention_split(const float* qkv, int ld, void* out_split, int B, int T, int heads, void* stream);
/* The VAE mid-block attention's core (one head, head_dim 512; diffusers Attention inside AutoencoderKL, call sites
 * genpercept_pipeline.py:500-501,521-522), fused: softmax(scale * q k^T) v with scores and probabilities kept on the CU.  q, k: [B][T][ld]
 * (512 channels), vt: [B][512][Tpad] zero beyond T, out [B][T][ldo].  ncu: persistent workgroups to size the launch for (0 = the device's CU
 * count; the parity tests pass small values to exercise the key-split of the left-over query blocks). */
gp_status gp_flash_attention_hd512(const void* q, const void* k, const void* vt, void* out, int B, int T, int ldq, int ldk, int Tpad, int ldo,
                                   float scale, int ncu, void* stream);
gp_status gp_cross_attention(const void* q, const float* kc, const float* vc, void* out, int rows, int C, int L, void* stream);
/* BasicTransformerBlock.attn2 (+ norm2, + residual, + norm3 of the result) for a TWO-token context, folded into per-head vectors
 * (gp_set_context does the fold for the engine; csrc/norm.hip states the algebra): y_out = y + c0 + sum_h sigmoid(LNhat(y) . U[h] + u0[h]) G[h],
 * n3_out = LayerNorm(y_out; g3, b3) (optional).  U, G: [heads][C] fp32, u0 [heads], c0 / g3 / b3 [C]; y_out may alias y. */
gp_status gp_cross_attention_fold(const void* y, void* y_out, void* n3_out, const float* U, const float* u0, const float* G, const float* c0,
                                  const float* g3, const float* b3, int rows, int C, int heads, float eps, void* stream);
gp_status gp_softmax_rows(const float* in, void* out, int rows, int T, int ld, float scale, void* stream);
/* the same for fp16 logits (gp_gemm with out_fp32 = 2 writes them): ld % 4 == 0, ld <= 16384 */
gp_status gp_softmax_rows_f16(const void* in_f16, void* out, int rows, int T, int ld, float scale, void* stream);
gp_status gp_bilinear(const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENPERCEPT_HIP_H */
