"""Stage-level and end-to-end parity (GPU) of the HIP engine, through the C-ABI, against
  (1) the committed golden vectors of the fp32 oracle (tests/golden/e2e_tiny.npz; tiny widths, two image shapes),
  (2) the REFERENCE's own DPT head outputs (tests/golden/dpt_head_ref.npz, full-size head),
  (3) the fp32 oracle run live on the host CPU at the full SD2.1 architecture (small image).

Tolerances.  The engine stores activations in bf16 (8 mantissa bits, one rounding per layer output; accumulation and
statistics in fp32), the oracle is fp32 end to end.  Through ~60 (VAE) / ~250 (UNet) rounded layers the observed
deviation is ~1e-2 of the tensor's RMS; we therefore gate every stage on
      rel_rms = rms(out - ref) / rms(ref)  <= TOL_STAGE (3e-2)         and on the final [0,1] maps additionally
      mean |out - ref| <= TOL_MAP_MEAN (1e-2),  and for depth the reference's own protocol metric (eval.py:168-215):
      AbsRel(out, ref) after least-squares alignment <= TOL_ABSREL (3e-2; AbsRel divides by the reference map, which on
      these synthetic-weight outputs reaches 0, so it is the loosest of the three).
These are bf16-storage noise levels, not kernel defects: test_hip_is_at_least_as_close_as_torch_bf16 runs the SAME
modules in PyTorch's own bf16 on the host and requires the HIP engine to be no further from the fp32 oracle than that.
north_star's "1e-3 rel" is not reachable by ANY bf16-storage implementation (8 mantissa bits per rounding); the measured
numbers are logged to gpurun_out/parity_log.jsonl and reported in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# per precision of the engine (bf16 = libgenpercept_hip.so, fp16 = libgenpercept_hip_f16.so): r6 -- 1.25x the largest deviation measured on
# MI355X for each quantity (gpurun_out/parity_log.jsonl of tools/sessions/gpu_r06_s1.sh: stage rel-rms 1.83e-2 bf16 / 2.26e-3 fp16, map mean
# 6.74e-3 / 8.2e-4), not more (r5: 2x)
TOLS = {"bf16": dict(stage=2.3e-2, map_mean=8.4e-3, absrel=3e-2),
        "fp16": dict(stage=2.9e-3, map_mean=1e-3, absrel=4e-3),
        # contract precision (gp_set_precision(GP_PREC_CONTRACT), csrc/contract.hip): fp32 storage + split-bf16 matrix products
        "fp32c": dict(stage=3e-4, map_mean=5e-5, absrel=3e-4)}
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32c": torch.float32}
PRECISIONS = ["bf16", "fp16", "fp32c"]
TOL_STAGE, TOL_MAP_MEAN, TOL_ABSREL = TOLS["bf16"]["stage"], TOLS["bf16"]["map_mean"], TOLS["bf16"]["absrel"]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_rms(out, ref):
    out, ref = out.float().cpu(), torch.as_tensor(ref).float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all()
    return ((out - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-12)).item()


def stage_check(name, out, ref, log, tol=TOL_STAGE):
    r = rel_rms(out, ref)
    ref_t = torch.as_tensor(ref).float()
    log(name, rel_rms=r, max_abs=(out.float().cpu() - ref_t).abs().max().item(), ref_rms=ref_t.pow(2).mean().sqrt().item())
    assert r <= tol, f"{name}: rel rms {r:.3e} > {tol}"


def absrel_after_ls(pred, gt):
    """eval.py protocol on a pair of maps: least-squares scale/shift then mean |a-b|/b (src/util/alignment.py:29-76, metric.py:34-44)."""
    p, g = pred.reshape(-1, 1).astype(np.float64), gt.reshape(-1).astype(np.float64)
    a = np.concatenate([p, np.ones_like(p)], axis=1)
    x = np.linalg.lstsq(a, g, rcond=None)[0]
    al = (a @ x).clip(1e-3, None)
    gg = g.clip(1e-3, None)
    return float(np.mean(np.abs(al - gg) / gg))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLD, "e2e_tiny.npz"))


@pytest.fixture(scope="module")
def tiny_weights():
    from oracle import dpt as odpt
    from oracle import sd21 as osd
    uc, vc, dc = osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg.tiny()
    return dict(uc=uc, vc=vc, dc=dc, usd=osd.synth_state_dict(osd.unet_manifest(uc), 1), vsd=osd.synth_state_dict(osd.vae_manifest(vc), 2),
                dsd=osd.synth_state_dict(odpt.dpt_manifest(dc), 3))


def _engine(tw, dpt, ctx, precision="bf16"):
    from genpercept_amd.engine import Engine
    import dataclasses
    uc = tw["uc"] if not dpt else dataclasses.replace(tw["uc"], has_out=False)
    eng = Engine(0, uc, tw["vc"], tw["dc"] if dpt else None, precision=precision)
    eng.load_state_dict("vae", tw["vsd"])
    eng.load_state_dict("unet", {k: v for k, v in tw["usd"].items() if not (dpt and k.startswith(("conv_out", "conv_norm_out")))})
    if dpt:
        eng.load_state_dict("dpt", tw["dsd"])
    eng.set_context(torch.as_tensor(ctx))
    eng.finalize()
    return eng


@pytest.fixture(scope="module", params=PRECISIONS)
def eng_vae(request, tiny_weights, golden):
    e = _engine(tiny_weights, False, golden["sq_ctx"], request.param)
    yield e
    e.close()


@pytest.fixture(scope="module", params=PRECISIONS)
def eng_dpt(request, tiny_weights, golden):
    e = _engine(tiny_weights, True, golden["sq_ctx"], request.param)
    yield e
    e.close()


@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_stages_vs_golden(tag, eng_vae, golden, metric_log):
    d = torch.device("cuda", 0)
    pr = eng_vae.precision
    tol = TOLS[pr]["stage"]
    tag_p = f"{tag},{pr}"
    eng_vae.set_context(torch.as_tensor(golden[f"{tag}_ctx"]))
    rgb = torch.as_tensor(golden[f"{tag}_rgb_u8"]).to(d)
    lat = eng_vae.vae_encode(rgb)
    stage_check(f"vae_encode[{tag_p}]", lat, golden[f"{tag}_latent"], metric_log, tol)
    # feed the GOLDEN latent so each stage is judged on its own
    gl = torch.as_tensor(golden[f"{tag}_latent"]).to(d)
    v, feats = eng_vae.unet(gl, want_sample=True, want_feats=True)
    stage_check(f"unet[{tag_p}]", v, golden[f"{tag}_unet"], metric_log, tol)
    for i, f in enumerate(feats):  # (the golden features are stored as fp16: their own rounding, 5e-4, is inside the fp16 tolerance)
        stage_check(f"unet_feat{i}[{tag_p}]", f, golden[f"{tag}_feat{i}"].astype(np.float32), metric_log, max(tol, 6e-4))
    gv = torch.as_tensor(golden[f"{tag}_unet"]).to(d)
    dec = eng_vae.vae_decode(-gv, mean3=False)
    stage_check(f"vae_decode3[{tag_p}]", dec, golden[f"{tag}_dec3"], metric_log, tol)
    dec1 = eng_vae.vae_decode(-gv, mean3=True)
    stage_check(f"vae_decode1[{tag_p}]", dec1, golden[f"{tag}_dec3"].mean(axis=1, keepdims=True), metric_log, tol)


@pytest.mark.parametrize("tag", ["sq", "odd"])
@pytest.mark.parametrize("mode", ["depth", "normal"])
def test_infer_vs_golden(tag, mode, eng_vae, golden, metric_log):
    d = torch.device("cuda", 0)
    eng_vae.set_context(torch.as_tensor(golden[f"{tag}_ctx"]))
    out = eng_vae.infer(torch.as_tensor(golden[f"{tag}_rgb_u8"]).to(d), mode)
    ref = golden[f"{tag}_{mode}"]
    assert tuple(out.shape) == ref.shape
    o = out.cpu().numpy()
    assert o.min() >= 0.0 and o.max() <= 1.0
    mean_abs = float(np.abs(o - ref).mean())
    rec = dict(mean_abs=mean_abs, max_abs=float(np.abs(o - ref).max()), rel_rms=rel_rms(out, ref))
    if mode == "depth":
        rec["absrel_ls"] = absrel_after_ls(o, ref)
    metric_log(f"infer_{mode}[{tag},{eng_vae.precision}]", **rec)
    assert mean_abs <= TOLS[eng_vae.precision]["map_mean"], rec
    if mode == "depth":
        assert rec["absrel_ls"] <= TOLS[eng_vae.precision]["absrel"], rec


@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_infer_dpt_vs_golden(tag, eng_dpt, golden, metric_log):
    d = torch.device("cuda", 0)
    eng_dpt.set_context(torch.as_tensor(golden[f"{tag}_ctx"]))
    out = eng_dpt.infer(torch.as_tensor(golden[f"{tag}_rgb_u8"]).to(d), "disparity")
    ref = golden[f"{tag}_disp"]
    assert tuple(out.shape) == ref.shape
    o = out.cpu().numpy()
    mean_abs = float(np.abs(o - ref).mean())
    metric_log(f"infer_disp_dpt[{tag},{eng_dpt.precision}]", mean_abs=mean_abs, max_abs=float(np.abs(o - ref).max()), rel_rms=rel_rms(out, ref), mn=float(o.min()), mx=float(o.max()))
    assert abs(o.min()) < 1e-6 and abs(o.max() - 1.0) < 1e-6  # per-image min-max (genpercept_pipeline.py:482)
    assert mean_abs <= 2 * TOLS[eng_dpt.precision]["map_mean"]  # the min-max division rescales the head's rounding noise by 1 / (max - min)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_dpt_head_vs_reference_outputs(precision, metric_log):
    """Full-size DPT head against outputs of the reference's own dpt_head.py (generated in the build container)."""
    from genpercept_amd.engine import Engine
    from oracle import dpt as odpt
    from oracle import sd21 as osd
    g = np.load(os.path.join(GOLD, "dpt_head_ref.npz"))
    sd = osd.synth_state_dict(odpt.dpt_manifest(), int(g["seed"]))
    uc, vc = osd.UNetCfg.tiny(), osd.VAECfg.tiny()
    eng = Engine(0, uc, vc, odpt.DPTCfg(), precision=precision)
    eng.load_state_dict("dpt", sd)
    eng.finalize()
    d = torch.device("cuda", 0)
    try:
        for tag in "ab":
            h, w = (int(x) for x in g[f"{tag}_hw"])
            gen = torch.Generator().manual_seed(100 + h * w)
            feats = [torch.randn(1, 320, h, w, generator=gen), torch.randn(1, 640, h, w, generator=gen),
                     torch.randn(1, 1280, h // 2, w // 2, generator=gen), torch.randn(1, 1280, h // 4, w // 4, generator=gen)]
            out = eng.dpt_head([f.to(d) for f in feats])
            stage_check(f"dpt_head_ref[{tag},{precision}]", out, g[f"{tag}_out"], metric_log, TOLS[precision]["stage"])
    finally:
        eng.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_dpt_head_vs_reference_outputs_odd_shapes(precision, metric_log):
    """The HIP DPT head on odd feature shapes (latents 9x11, 13x10, 29x39) against the REFERENCE's own outputs: covers the
    bilinear resize of a neck feature to the fused map's size (dpt_head.py:297-300) end to end."""
    from genpercept_amd.engine import Engine
    from oracle import dpt as odpt
    from oracle import sd21 as osd
    from test_oracle import _odd_dpt_feats
    g = np.load(os.path.join(GOLD, "dpt_head_ref_odd.npz"))
    sd = osd.synth_state_dict(odpt.dpt_manifest(), int(g["seed"]))
    eng = Engine(0, osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg(), precision=precision)
    eng.load_state_dict("dpt", sd)
    eng.finalize()
    d = torch.device("cuda", 0)
    try:
        for tag in "cde":
            h, w = (int(x) for x in g[f"{tag}_hw"])
            out = eng.dpt_head([f.to(d) for f in _odd_dpt_feats(h, w)])
            stage_check(f"dpt_head_ref_odd[{tag},{precision}]", out, g[f"{tag}_out"], metric_log, TOLS[precision]["stage"])
    finally:
        eng.close()


def test_hip_is_at_least_as_close_as_torch_bf16(eng_vae, tiny_weights, golden, metric_log):
    """Yardstick for the tolerances: PyTorch running the same modules in bf16 (what `--dtype bf16` of the reference would
    do) deviates from the fp32 oracle by X; the HIP engine (bf16 storage, fp32 accumulate/statistics) must be <= 1.25 X."""
    from oracle import sd21 as osd
    if eng_vae.precision != "bf16":
        pytest.skip("yardstick for the bf16 library")
    d = torch.device("cuda", 0)
    tw = tiny_weights
    bf = lambda sd: {k: v.to(torch.bfloat16) for k, v in sd.items()}  # noqa: E731
    eng_vae.set_context(torch.as_tensor(golden["sq_ctx"]))
    gl, gv = torch.as_tensor(golden["sq_latent"]), torch.as_tensor(golden["sq_unet"])
    ctx = torch.as_tensor(golden["sq_ctx"])[None].expand(2, -1, -1)
    with torch.no_grad():
        t_unet, _ = osd.unet_forward(bf(tw["usd"]), tw["uc"], gl.to(torch.bfloat16), 1, ctx.to(torch.bfloat16))
        t_dec = osd.decode_pred(bf(tw["vsd"]), tw["vc"], (-gv).to(torch.bfloat16), "normal")
    h_unet = eng_vae.unet(gl.to(d))[0]
    h_dec = eng_vae.vae_decode(-gv.to(d), mean3=False)
    for name, hip, tor, ref in (("unet", h_unet, t_unet, golden["sq_unet"]), ("vae_decode", h_dec, t_dec, golden["sq_dec3"])):
        eh, et = rel_rms(hip, ref), rel_rms(tor, ref)
        metric_log(f"bf16_yardstick_{name}", hip_rel_rms=eh, torch_bf16_rel_rms=et, ratio=eh / et)
        assert eh <= 1.25 * et, (name, eh, et)


def test_batch_equals_single(eng_vae, golden, metric_log):
    """Sharding correctness: an image's result does not depend on what else is in the batch."""
    d = torch.device("cuda", 0)
    eng_vae.set_context(torch.as_tensor(golden["sq_ctx"]))
    rgb = torch.as_tensor(golden["sq_rgb_u8"]).to(d)
    both = eng_vae.infer(rgb, "depth")
    one0 = eng_vae.infer(rgb[:1], "depth")
    one1 = eng_vae.infer(rgb[1:], "depth")
    diff = max((both[:1] - one0).abs().max().item(), (both[1:] - one1).abs().max().item())
    metric_log(f"batch_vs_single[{eng_vae.precision}]", max_abs=diff, bitwise=float(diff == 0.0))
    assert diff <= 1e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("hw", [(64, 64), (232, 312)])
def test_full_sd21_architecture_small_image(hw, precision, metric_log):
    """Full-width SD2.1 UNet (865.9 M params) + VAE against the fp32 oracle run on the host CPU: 64x64 px (every map below the
    16x16-tile kernels' minimum: generic implicit GEMM, split-K, persistent GEMM) and 232x312 px (latent 29x39: the persistent halo
    kernels with ragged tile edges, several tiles per workgroup, fused GroupNorm inputs and epilogue statistics, at real widths)."""
    from genpercept_amd.engine import Engine
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    uc, vc = osd.UNetCfg(), osd.VAECfg()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 11)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 12)
    g = torch.Generator().manual_seed(77)
    rgb_u8 = torch.randint(0, 256, (1, 3, hw[0], hw[1]), generator=g, dtype=torch.uint8)
    rgb_u8[:, :, : hw[0] // 2] //= 2
    ctx = torch.randn(2, 1024, generator=g)
    with torch.no_grad():
        rgb = opipe.normalize_rgb(rgb_u8)
        lat = osd.encode_rgb(vsd, vc, rgb)
        v, _ = osd.unet_forward(usd, uc, lat, 1, ctx[None])
        ref = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "depth")
    eng = Engine(0, uc, vc, None, precision=precision)
    eng.load_state_dict("vae", vsd)
    eng.load_state_dict("unet", usd)
    eng.set_context(ctx)
    eng.finalize()
    d = torch.device("cuda", 0)
    tol = TOLS[precision]
    try:
        stage_check(f"full_vae_encode{hw}[{precision}]", eng.vae_encode(rgb_u8.to(d)), lat, metric_log, tol["stage"])
        stage_check(f"full_unet{hw}[{precision}]", eng.unet(lat.to(d))[0], v, metric_log, tol["stage"])
        out = eng.infer(rgb_u8.to(d), "depth").cpu().numpy()
        mean_abs = float(np.abs(out - ref.numpy()).mean())
        metric_log(f"full_infer_depth{hw}[{precision}]", mean_abs=mean_abs, max_abs=float(np.abs(out - ref.numpy()).max()), absrel_ls=absrel_after_ls(out, ref.numpy()))
        assert mean_abs <= tol["map_mean"]
    finally:
        eng.close()


def test_pipeline_surface(tiny_weights, golden, metric_log):
    """The GenPerceptPipeline mirror: same kwargs / asserts / outputs as genpercept_pipeline.py:146-337."""
    from PIL import Image
    from genpercept_amd import GenPerceptOutput, GenPerceptPipeline

    class Sched:  # what run.py:371 loads from hf_configs/scheduler_beta_1.0_1.0
        beta_start, beta_end, prediction_type, clip_sample = 1.0, 1.0, "v_prediction", False
        steps_offset, timestep_spacing = 1, "leading"  # => set_timesteps(1) == [1] (genpercept_pipeline.py:403)

    pipe = GenPerceptPipeline(unet=tiny_weights["usd"], vae=tiny_weights["vsd"], scheduler=Sched(), text_encoder=golden["sq_ctx"], tokenizer=None)
    pipe.to("cuda")
    assert pipe.default_denoising_steps == 1 and pipe.rgb_blending
    img = Image.fromarray(np.transpose(golden["sq_rgb_u8"][0], (1, 2, 0)))
    out = pipe(img, denoising_steps=1, ensemble_size=1, processing_res=0, match_input_res=True, batch_size=0, color_map="Spectral",
               show_progress_bar=False, resample_method="bilinear", mode="depth")
    assert isinstance(out, GenPerceptOutput) and out.pred_np.shape == (64, 64) and out.pred_np.dtype == np.float32
    assert out.pred_colored.size == (64, 64)
    mean_abs = float(np.abs(out.pred_np - golden["sq_depth"][0, 0]).mean())
    metric_log("pipeline_call_depth", mean_abs=mean_abs)
    assert mean_abs <= TOL_MAP_MEAN
    outn = pipe(img, processing_res=0, color_map=None, mode="normal")
    assert outn.pred_np.shape == (64, 64, 3)
    # processing_res resizes to max edge then back to the input size
    out2 = pipe(img.resize((80, 60)), processing_res=64, mode="depth")
    assert out2.pred_np.shape == (60, 80)
    with pytest.raises(AssertionError):
        pipe(img)  # mode is required (:199)
    with pytest.raises(AssertionError):
        pipe(img, mode="depth", ensemble_size=2)
    with pytest.raises(AssertionError):
        pipe(img, mode="normal", color_map="Spectral")  # color_map only for depth/disparity (:318)
    with pytest.raises(ValueError):
        pipe(img, mode="depth", resample_method="lanczos")
    with pytest.raises(TypeError):
        pipe(np.zeros((8, 8, 3)), mode="depth")
    outs = pipe.infer_batch(torch.as_tensor(golden["sq_rgb_u8"]), mode="depth", processing_res=0)
    assert len(outs) == 2 and np.abs(outs[0].pred_np - out.pred_np).max() <= 1e-6


def test_full_size_768_properties(metric_log):
    """BASELINE.json's full size (768x768, full SD2.1 widths) cannot be run through the CPU oracle in test time, so the
    HIP path is pinned there by size-independent properties: run-to-run determinism (bitwise), batch-permutation
    equivariance (bitwise: an image's result does not depend on its batch slot), and agreement between the two
    independent conv/GroupNorm code paths (fused halo kernel + epilogue statistics vs generic implicit GEMM + separate
    GroupNorm passes; they share no kernel for the 3x3 convolutions of the large maps)."""
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from genpercept_amd.engine import Engine
    d = torch.device("cuda", 0)
    ucfg, vcfg = gc.UNetConfig(), gc.VAEConfig()
    usd = gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0)
    vsd = gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1)
    ctx = torch.randn(2, 1024, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(3)
    rgb = torch.randint(0, 256, (4, 3, 768, 768), generator=g, dtype=torch.uint8)  # batch 4 = BASELINE.json configs[1] (grids depend on B)
    rgb[1, :, :, :384] //= 3
    rgb[2, :, 384:] //= 2
    rgb[3] = 255 - rgb[3] // 4

    def build(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            e = Engine(0, ucfg, vcfg, None)
            e.load_state_dict("vae", vsd)
            e.load_state_dict("unet", usd)
            e.set_context(ctx)
            e.finalize()
            return e
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    eng = build({})
    try:
        a = eng.infer(rgb.to(d), "depth")
        b = eng.infer(rgb.to(d), "depth")
        assert a.shape == (4, 1, 768, 768) and torch.isfinite(a).all() and 0.0 <= float(a.min()) and float(a.max()) <= 1.0
        assert torch.equal(a, b), "not deterministic"
        sw = eng.infer(rgb.flip(0).to(d), "depth")
        assert torch.equal(sw.flip(0), a), "result depends on the batch slot"
        # another batch size: the persistent kernels use other grids (#CU / B workgroups per image), so the GroupNorm partial sums are
        # added in another order -- the same image may differ in the last bf16 bit of some activations, not more
        two = eng.infer(rgb[1:3].to(d), "depth")
        dsz = (two - a[1:3]).abs()
        metric_log("full768_batch4_vs_batch2", mean_abs=dsz.mean().item(), max_abs=dsz.max().item(), bitwise=float(torch.equal(two, a[1:3])))
        assert dsz.mean().item() <= TOL_MAP_MEAN, "result depends on the batch size"  # (two bf16 runs rounding at different points: same bound as vs fp32)
        # the multi-step loop (gp_infer_steps, rgb_blending form on this 4-channel UNet) at the full size: one beta == 1 step is gp_infer
        # bit for bit, a 3-step DDIM walk is deterministic, in range and independent of the batch slot
        from genpercept_amd.scheduler import DDIMSchedulerCustomized
        base = dict(beta_schedule="scaled_linear", prediction_type="v_prediction", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
        assert torch.equal(eng.infer_steps(rgb.to(d), "depth", DDIMSchedulerCustomized(beta_start=1.0, beta_end=1.0, **base).plan(1)), a)
        plan3 = DDIMSchedulerCustomized(beta_start=0.00085, beta_end=0.012, **base).plan(3)
        m3 = eng.infer_steps(rgb.to(d), "depth", plan3)
        assert torch.isfinite(m3).all() and 0.0 <= float(m3.min()) and float(m3.max()) <= 1.0 and not torch.equal(m3, a)
        assert torch.equal(eng.infer_steps(rgb.to(d), "depth", plan3), m3), "loop not deterministic"
        assert torch.equal(eng.infer_steps(rgb.flip(0).to(d), "depth", plan3).flip(0), m3), "loop result depends on the batch slot"
        assert torch.equal(eng.infer(rgb.to(d), "depth"), a), "the loop left another timestep behind"
        n3 = eng.infer(rgb.to(d), "normal")
        assert n3.shape == (4, 3, 768, 768)
        # depth is the clipped channel mean of the same decode: equal to the mean of the normal channels wherever no
        # channel was clipped (both maps are bf16-rounded once, hence the small tolerance)
        inside = ((n3 > 1e-3) & (n3 < 1 - 1e-3)).all(dim=1, keepdim=True)
        assert inside.float().mean().item() > 0.05
        assert ((n3.mean(dim=1, keepdim=True) - a).abs() * inside).max().item() < 0.01
    finally:
        eng.close()
    # NO_HALO is read once per process by the launcher, so only the GroupNorm fusion switches are exercised in-process
    eng2 = build({"GENPERCEPT_NO_GN_FUSION": "1", "GENPERCEPT_NO_STATS_FUSION": "1"})
    try:
        c = eng2.infer(rgb.to(d), "depth")
    finally:
        eng2.close()
    diff = (c - a).abs()
    metric_log("full768_fused_vs_unfused", mean_abs=diff.mean().item(), max_abs=diff.max().item(), out_std=a.std().item())
    # two bf16 executions that round at different points decorrelate like either does from the fp32 oracle (measured 4e-3 mean on
    # a map of std 0.12; the kernel-level statistics test pins the fused path exactly): same bound as the oracle comparison
    assert diff.mean().item() <= TOL_MAP_MEAN, diff.mean().item()


@pytest.mark.parametrize("fused", [True, False], ids=["flash512", "gemm_softmax_gemm"])
@pytest.mark.parametrize("hw", [(12, 10), (16, 16)])
@pytest.mark.parametrize("rms", [30.0, 60.0])
def test_vae_attention_large_norm_logits(hw, rms, fused, metric_log, monkeypatch):
    """VAE mid-block attention (1 head x 512; genpercept_pipeline.py:500-501, 521-522) with q / k of RMS 30-60: raw q.k^T over 512
    dims reaches ~1e5, beyond fp16's 65504 -- the SD VAE's known fp16 overflow site (VERDICT r1 weak 3 / ADVICE r1).  The engine must
    stay finite and agree with an fp32 attention evaluated on the same bf16-rounded operands.  Reference: oracle/sd21._attention."""
    from genpercept_amd.engine import Engine
    from oracle import sd21 as osd
    if not fused:  # the path other widths take: scaled logits as saturating fp16 in HBM, row softmax, second GEMM
        monkeypatch.setenv("GENPERCEPT_NO_FLASH512", "1")
    uc, vc = osd.UNetCfg.tiny(), osd.VAECfg()
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 21)
    p = "decoder.mid_block.attentions.0"
    gn, nq, nk, nv, no = osd._vae_attn_names(vsd, p)
    c = vsd[gn + ".weight"].numel()
    for n in (nq, nk):  # GroupNorm output has unit variance and the synthetic projections preserve it: scale them to the wanted RMS
        vsd[n + ".weight"] = vsd[n + ".weight"] * rms
        vsd[n + ".bias"] = vsd[n + ".bias"] * rms
    g = torch.Generator().manual_seed(int(rms) + hw[0])
    x = torch.randn(2, c, hw[0], hw[1], generator=g)
    rb = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    with torch.no_grad():  # the engine's rounding points (bf16 operands / stored tensors), fp32 everywhere else
        xb = rb(x)
        n = rb(osd._gn(xb, vsd, gn, vc.norm_num_groups, vc.norm_eps)).reshape(2, c, -1).transpose(1, 2)
        s = c ** -0.5
        q = rb(F.linear(n, rb(vsd[nq + ".weight"].reshape(c, c) * s), vsd[nq + ".bias"] * s))
        k = rb(F.linear(n, rb(vsd[nk + ".weight"].reshape(c, c)), vsd[nk + ".bias"]))
        v = rb(F.linear(n, rb(vsd[nv + ".weight"].reshape(c, c)), vsd[nv + ".bias"]))
        assert (q / s).pow(2).mean().sqrt() > 0.8 * rms and float(((q / s) @ k.transpose(1, 2)).abs().max()) > 65504.0  # raw logits DO leave fp16
        pr = rb(torch.softmax(q @ k.transpose(1, 2), dim=-1))
        o = rb(pr @ v)
        ref = F.linear(o, rb(vsd[no + ".weight"].reshape(c, c)), vsd[no + ".bias"]).transpose(1, 2).reshape(x.shape) + xb
    eng = Engine(0, uc, vc, None)
    try:
        eng.load_state_dict("vae", vsd)
        eng.finalize()
        out = eng.vae_mid_attention(x.cuda(), decoder=True).cpu()
    finally:
        eng.close()
    assert torch.isfinite(out).all(), "non-finite attention output (fp16 logit overflow)"
    err = (out - ref).abs()
    bad = (err > 2e-2 * ref.abs().max()).float().mean().item()  # near-one-hot softmax: a near-tie may pick the other key in a few rows
    metric_log(f"vae_attn_large_logits{hw}rms{rms}{'' if fused else '[unfused]'}", rel_rms=rel_rms(out, ref), max_abs=err.max().item(), frac_bad=bad, ref_max=ref.abs().max().item())
    # (its own gate, not TOL_STAGE: on the unfused path the logits are SATURATED fp16, the softmax is one-hot and in ~0.3 % of the rows a near-tie picks
    #  the other key -- those rows carry the whole deviation: 2.84e-2 at rms 60, 1.7e-3 everywhere else; 1.25x)
    assert bad <= 5e-3 and rel_rms(out, ref) <= 3.6e-2


@pytest.mark.parametrize("hw", [(12, 10), (130, 128)])
def test_vae_attention_contract_precision(hw, metric_log):
    """The VAE mid-block attention alone (gp_vae_mid_attention) in the contract precision: the unfused split-operand path (head split -> batched logits GEMM with
    fp32 logits in HBM -> row softmax -> batched P.V GEMM -> merge; contract.hip) against the fp32 oracle, on a small map (register-resident softmax rows) and on
    one with more than 16384 tokens (130 x 128 = 16640: the three-pass long-row softmax kernel)."""
    from genpercept_amd.engine import Engine
    from oracle import sd21 as osd
    uc, vc = osd.UNetCfg.tiny(), osd.VAECfg()
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 21)
    p = "decoder.mid_block.attentions.0"
    for n in ("to_q", "to_k"):  # logits of a few units: a softmax with structure, not a uniform average
        vsd[f"{p}.{n}.weight"] = vsd[f"{p}.{n}.weight"] * 3.0
    g = torch.Generator().manual_seed(hw[0])
    x = torch.randn(1, 512, hw[0], hw[1], generator=g)
    with torch.no_grad():
        ref = osd.vae_mid_attention(x, vsd, p, vc.norm_num_groups, vc.norm_eps)
    eng = Engine(0, uc, vc, None, precision="fp32c")
    try:
        eng.load_state_dict("vae", vsd)
        eng.finalize()
        out = eng.vae_mid_attention(x.cuda(), decoder=True).cpu()
    finally:
        eng.close()
    # the attention branch alone (the residual x dominates the sum)
    r = ((out - ref).pow(2).mean().sqrt() / (ref - x).pow(2).mean().sqrt()).item()
    metric_log(f"vae_attn_contract{hw}", rel_rms_of_branch=r, max_abs=(out - ref).abs().max().item())
    assert torch.isfinite(out).all() and r <= 2e-4, r


def test_full_size_768_properties_dpt_head(metric_log):
    """BASELINE.json configs[3] (DPT head at 768x768, full widths, batch 4) pinned by the same size-independent properties: bitwise
    determinism, batch-slot and batch-size independence, per-image min-max normalisation (genpercept_pipeline.py:480-482)."""
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from genpercept_amd.engine import Engine
    d = torch.device("cuda", 0)
    ucfg, vcfg, dcfg = gc.UNetConfig(has_out=False), gc.VAEConfig(), gc.DPTConfig()
    eng = Engine(0, ucfg, vcfg, dcfg)
    try:
        eng.load_state_dict("vae", gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1))
        eng.load_state_dict("unet", gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0))
        eng.load_state_dict("dpt", gw.synth_state_dict(gw.dpt_manifest(dcfg), seed=3))
        eng.set_context(torch.randn(2, 1024, generator=torch.Generator().manual_seed(2)))
        eng.finalize()
        g = torch.Generator().manual_seed(5)
        rgb = torch.randint(0, 256, (4, 3, 768, 768), generator=g, dtype=torch.uint8)
        rgb[1, :, :, :384] //= 3
        rgb[2, :, 384:] //= 2
        a = eng.infer(rgb.to(d), "disparity")
        b = eng.infer(rgb.to(d), "disparity")
        assert a.shape == (4, 1, 768, 768) and torch.isfinite(a).all()
        assert torch.equal(a, b), "not deterministic"
        mn, mx = a.amin(dim=(1, 2, 3)), a.amax(dim=(1, 2, 3))
        assert float(mn.abs().max()) < 1e-6 and float((mx - 1).abs().max()) < 1e-6, "per-image min-max"
        assert torch.equal(eng.infer(rgb.flip(0).to(d), "disparity").flip(0), a), "result depends on the batch slot"
        dsz = (eng.infer(rgb[2:3].to(d), "disparity") - a[2:3]).abs()  # other grids, other summation order of the statistics
        assert dsz.mean().item() <= 2 * TOL_MAP_MEAN, "result depends on the batch size"
        metric_log("full768_dpt_properties", out_std=a.std().item(), batch1_vs_batch4_mean_abs=dsz.mean().item(), max_abs=dsz.max().item())
    finally:
        eng.close()


def _two_engine_job(dev_index, tw, ctx, rgb_u8, out, key, rounds=3):
    """Worker of the threading tests: build an engine, run `rounds` inferences, keep the last result."""
    try:
        torch.cuda.set_device(dev_index)
        from genpercept_amd.engine import Engine
        eng = Engine(dev_index, tw["uc"], tw["vc"], None)
        eng.load_state_dict("vae", tw["vsd"])
        eng.load_state_dict("unet", tw["usd"])
        eng.set_context(torch.as_tensor(ctx))
        eng.finalize()
        d = torch.device("cuda", dev_index)
        x = torch.as_tensor(rgb_u8).to(d)
        for _ in range(rounds):
            y = eng.infer(x, "depth")
        torch.cuda.synchronize(d)
        out[key] = y.cpu()
        eng.close()
    except Exception as ex:  # surfaced by the caller
        out[key] = ex


def test_two_engines_on_two_threads_one_gpu(tiny_weights, golden, metric_log):
    """The C-ABI's threading sentence (include/genpercept_hip.h): different engines may be driven from different host threads.  Two
    engines on the SAME GPU, each on its own thread and torch stream, through shapes that take the split-K path (12x12 .. 2x2 maps,
    long K) and the GroupNorm workspaces: results must equal the single-threaded result bit for bit."""
    import threading
    ref = {}
    _two_engine_job(0, tiny_weights, golden["sq_ctx"], golden["sq_rgb_u8"], ref, "ref", rounds=1)
    assert torch.is_tensor(ref["ref"]), ref["ref"]
    out = {}

    def run(key):
        with torch.cuda.stream(torch.cuda.Stream(device=0)):
            _two_engine_job(0, tiny_weights, golden["sq_ctx"], golden["sq_rgb_u8"], out, key, rounds=4)

    ths = [threading.Thread(target=run, args=(k,)) for k in ("a", "b")]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in ("a", "b"):
        assert torch.is_tensor(out[k]), out[k]
        assert torch.equal(out[k], ref["ref"]), f"engine {k} differs from the single-threaded run"
    metric_log("two_engines_two_threads_one_gpu", bitwise=1.0)


def test_two_engines_alternating_streams(tiny_weights, golden, metric_log):
    """Two engines on one GPU driven alternately, and one engine driven from two different streams in turn (the pool's stream fence)."""
    from genpercept_amd.engine import Engine
    d = torch.device("cuda", 0)
    engs = []
    for _ in range(2):
        e = Engine(0, tiny_weights["uc"], tiny_weights["vc"], None)
        e.load_state_dict("vae", tiny_weights["vsd"])
        e.load_state_dict("unet", tiny_weights["usd"])
        e.set_context(torch.as_tensor(golden["sq_ctx"]))
        e.finalize()
        engs.append(e)
    try:
        x = torch.as_tensor(golden["sq_rgb_u8"]).to(d)
        ref = engs[0].infer(x, "depth").clone()
        s1, s2 = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
        torch.cuda.synchronize()
        outs = []
        for i in range(6):
            with torch.cuda.stream(s1 if i % 2 else s2):
                outs.append(engs[i % 2].infer(x, "depth"))        # engines alternate, each always on its own stream
                outs.append(engs[0].infer(x, "depth"))            # ... and engine 0 additionally hops between the two streams
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, ref)
        metric_log("two_engines_alternating_streams", bitwise=1.0)
    finally:
        for e in engs:
            e.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_two_engines_two_gpus_two_threads(tiny_weights, golden, metric_log):
    """One engine per GPU, one host thread each (what an in-process 8-GPU driver does): per-device kernel attributes and workspaces."""
    import threading
    out = {}
    ths = [threading.Thread(target=_two_engine_job, args=(i, tiny_weights, golden["sq_ctx"], golden["sq_rgb_u8"], out, i)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert torch.is_tensor(out[0]) and torch.is_tensor(out[1]), out
    assert torch.equal(out[0], out[1])


def test_failed_call_returns_buffers(tiny_weights, golden):
    """A stage that throws mid-way (here: a UNet without context) must hand its activations back to the pool: the next calls work and the
    pool does not grow."""
    from genpercept_amd.engine import Engine
    d = torch.device("cuda", 0)
    e = Engine(0, tiny_weights["uc"], tiny_weights["vc"], None)
    try:
        e.load_state_dict("vae", tiny_weights["vsd"])
        e.load_state_dict("unet", tiny_weights["usd"])
        e.finalize()  # no set_context: the first transformer block raises after the encoder and several UNet layers have run
        x = torch.as_tensor(golden["sq_rgb_u8"]).to(d)
        for _ in range(3):
            with pytest.raises(RuntimeError):
                e.infer(x, "depth")
        e.set_context(torch.as_tensor(golden["sq_ctx"]))
        y = e.infer(x, "depth")
        assert torch.isfinite(y).all()
        assert float(np.abs(y.cpu().numpy() - golden["sq_depth"]).mean()) <= TOL_MAP_MEAN
    finally:
        e.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_infer_eval_loop_with_the_engine(precision, tiny_weights, tmp_path, metric_log):
    """infer.py -> eval.py end to end with the REAL pipeline (r1 only ran it with a stand-in): an NYU-layout tree of RGB
    images goes through genpercept_amd.infer_eval.run_inference (GenPerceptPipeline on the GPU, .npy per image named by get_pred_name),
    then evaluate_predictions with the reference's protocol (least-squares alignment, NYU range + Eigen crop, ten metrics) against ground
    truth DEFINED by the fp32 oracle's depth for the same image.  AbsRel of engine-vs-oracle is the "AbsRel unchanged" statement of
    north_star in the protocol's own units; the same for normals with the angular-error evaluator."""
    from PIL import Image
    from genpercept_amd import GenPerceptPipeline
    from genpercept_amd import eval_metrics as em
    from genpercept_amd import infer_eval as ie
    from oracle import pipeline as opipe
    tw = tiny_weights
    g = torch.Generator().manual_seed(31)
    ctx = torch.randn(2, tw["uc"].cross_attention_dim, generator=g)
    pipe = GenPerceptPipeline(unet=tw["usd"], vae=tw["vsd"], scheduler=dict(beta_start=1.0, beta_end=1.0, prediction_type="v_prediction", clip_sample=False,
                                                                                       steps_offset=1, timestep_spacing="leading"),
                              text_encoder=ctx, tokenizer=None, torch_dtype=TORCH_DTYPE[precision])
    pipe.to("cuda")
    base, outd = tmp_path / "data", tmp_path / "pred"
    samples, normals_ref = [], []
    for i in range(3):
        scene = base / "test" / f"room_{i:04d}"
        scene.mkdir(parents=True)
        rgb = torch.randint(0, 256, (3, 480, 640), generator=g, dtype=torch.uint8)
        rgb[:, :, : 200 + 100 * i] //= 2
        Image.fromarray(rgb.permute(1, 2, 0).numpy()).save(scene / f"rgb_{i:04d}.png")
        with torch.no_grad():
            x = opipe.normalize_rgb(rgb[None])
            d = opipe.single_infer(tw["vsd"], tw["vc"], tw["usd"], tw["uc"], x, ctx, "depth")[0, 0].numpy()
            normals_ref.append(opipe.single_infer(tw["vsd"], tw["vc"], tw["usd"], tw["uc"], x, ctx, "normal")[0].numpy())
        depth_m = 0.8 + 8.0 * d  # oracle depth in [0, 1] -> "metres" inside the NYU range
        Image.fromarray(np.round(depth_m * 1000).astype(np.uint16)).save(scene / f"depth_{i:04d}.png")
        samples.append([f"test/room_{i:04d}/rgb_{i:04d}.png", f"test/room_{i:04d}/depth_{i:04d}.png"])
    written = ie.run_inference(pipe, str(base), samples, str(outd), ie.FileNameMode.rgb_id, mode="depth", processing_res=0)
    assert len(written) == 3 and np.load(written[0]).shape == (480, 640)
    res = ie.evaluate_predictions(str(outd), str(base), samples, dataset="nyu", alignment="least_square", output_dir=str(tmp_path / "eval"))
    metric_log(f"infer_eval_loop_depth[{precision}]", absrel=res["abs_relative_difference"], delta1=res["delta1_acc"], rmse=res["rmse_linear"])
    # measured on MI355X: AbsRel 8.9e-3 (bf16) / 1.1e-3 (fp16) -- gates at 2x
    assert res["abs_relative_difference"] <= (1.8e-2 if precision == "bf16" else 2.3e-3) and res["delta1_acc"] >= 0.999
    errs = []
    for i, s in enumerate(samples):
        out = pipe(Image.open(base / s[0]), processing_res=0, mode="normal", color_map=None)
        errs.append(em.normal_angular_error(em.decode_normals(out.pred_np), normals_ref[i] * 2.0 - 1.0)["mean_deg"])
    metric_log(f"infer_eval_loop_normal[{precision}]", mean_angular_error_deg=float(np.mean(errs)))
    assert float(np.mean(errs)) <= (4.0 if precision == "bf16" else 1.7)  # measured 2.01 / 0.82 degrees (the evaluator's clamp alone leaves 0.81)


def test_dpt_head_more_than_64_images(eng_dpt, golden):
    """The per-image min-max workspace is sized from the batch (a fixed 64-image buffer was overrun by B > 64, reachable through
    infer_batch at small resolutions).  70 images: every map normalised on its own, and image 69 agrees with the same image run alone."""
    d = torch.device("cuda", 0)
    eng_dpt.set_context(torch.as_tensor(golden["sq_ctx"]))
    g = torch.Generator().manual_seed(70)
    rgb = torch.randint(0, 256, (70, 3, 32, 32), generator=g, dtype=torch.uint8).to(d)
    out = eng_dpt.infer(rgb, "disparity")
    assert out.shape[0] == 70 and torch.isfinite(out).all()
    mn, mx = out.amin(dim=(1, 2, 3)), out.amax(dim=(1, 2, 3))
    assert float(mn.abs().max()) < 1e-6 and float((mx - 1).abs().max()) < 1e-6
    # (70 images vs 1: the launchers pick other tiles / split-K factors for the other M, so the accumulation order -- not the math -- differs)
    alone = eng_dpt.infer(rgb[69:70], "disparity")
    assert float((alone - out[69:70]).abs().mean()) <= 2 * TOLS[eng_dpt.precision]["map_mean"]


def test_modes_seg_matting_dis_on_the_gpu(eng_vae, tiny_weights, golden, metric_log):
    """genpercept_pipeline.py:199,523-525: `seg` keeps the three decoder channels like `normal`, `matting` and `dis` average them like
    `depth` -- the same decode, so the engine's maps must be bit-equal to the normal / depth ones, through gp_infer and through
    GenPerceptPipeline.__call__ (these modes were mapped but never run on the GPU)."""
    d = torch.device("cuda", 0)
    rgb = torch.as_tensor(golden["sq_rgb_u8"]).to(d)
    dep, nrm = eng_vae.infer(rgb, "depth"), eng_vae.infer(rgb, "normal")
    assert torch.equal(eng_vae.infer(rgb, "seg"), nrm) and eng_vae.infer(rgb, "seg").shape[1] == 3
    for m in ("matting", "dis"):
        o = eng_vae.infer(rgb, m)
        assert o.shape[1] == 1 and torch.equal(o, dep), m
    from genpercept_amd import GenPerceptPipeline
    pipe = GenPerceptPipeline(unet=tiny_weights["usd"], vae=tiny_weights["vsd"], text_encoder=golden["sq_ctx"], tokenizer=None,
                              torch_dtype=TORCH_DTYPE[eng_vae.precision])
    try:
        img = torch.as_tensor(golden["sq_rgb_u8"])[:1]
        outs = {m: pipe(img, processing_res=0, mode=m, color_map=None) for m in ("depth", "normal", "seg", "matting", "dis")}
        assert outs["seg"].pred_np.shape == outs["normal"].pred_np.shape and outs["seg"].pred_np.ndim == 3 and outs["seg"].pred_np.shape[-1] == 3
        assert np.array_equal(outs["seg"].pred_np, outs["normal"].pred_np)
        for m in ("matting", "dis"):
            assert outs[m].pred_np.ndim == 2 and np.array_equal(outs[m].pred_np, outs["depth"].pred_np), m
        with pytest.raises(AssertionError):  # genpercept_pipeline.py:318: a colour map is only legal for depth / disparity
            pipe(img, processing_res=0, mode="seg", color_map="Spectral")
    finally:
        if pipe._engine is not None:
            pipe._engine.close()
