"""GPU parity of the multi-step archs (`marigold`: noise sample + [rgb_latent, sample] UNet input; `rgb_blending`: the sample starts as the
rgb latent) -- gp_infer_steps through the C-ABI against the fp32 oracle's denoising loop (tests/golden/e2e_multistep.npz, and live at the
full SD2.1 widths), plus the pipeline surface on top of it (steps, ensembling, generator, fix_timesteps).

Tolerances: the final [0,1] map's mean |out - ref|, per element type, at most 2x the largest deviation measured on MI355X
(gpurun_out/parity_log.jsonl): v-prediction schedules <= 8.8e-3 (bf16) / 1.2e-3 (fp16) over 1-10 steps; the worst case, 1.3e-2 / 2.1e-3, is
the epsilon-prediction scheduler with trailing spacing, whose first step (t = 999) divides the UNet's rounding by sqrt(alpha_bar) = 0.068.
The sample itself is kept in fp32 on the device; only the UNet's input and output pass through the 16-bit element type."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_MAP = {"bf16": 1.62e-2, "fp16": 2.5e-3, "fp32c": 5e-5}  # r6: 1.25x / 1.2x the worst case measured in any round (1.3e-2 / 2.1e-3); fp32c 2.3e-5 x 2  # fp32c: the contract precision (fp32 storage, split-bf16 products)
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32c": torch.float32}
SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
             prediction_type="v_prediction", timestep_spacing="leading")  # hf_configs/scheduler_beta_0.00085_0.012


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLD, "e2e_multistep.npz"))


@pytest.fixture(scope="module")
def weights():
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    vc, uc4 = osd.VAECfg.tiny(), osd.UNetCfg.tiny()
    uc8 = dataclasses.replace(uc4, in_channels=8)
    u4 = osd.synth_state_dict(osd.unet_manifest(uc4), seed=1)
    return dict(vc=vc, uc4=uc4, uc8=uc8, vsd=osd.synth_state_dict(osd.vae_manifest(vc), seed=2), u4=u4, u8=opipe.replace_unet_conv_in(u4))


def _engine(w, arch, ctx, precision):
    from genpercept_amd.engine import Engine
    eng = Engine(0, w["uc8"] if arch == "marigold" else w["uc4"], w["vc"], None, precision=precision)
    eng.load_state_dict("vae", w["vsd"])
    eng.load_state_dict("unet", w["u8"] if arch == "marigold" else w["u4"])
    eng.set_context(torch.as_tensor(ctx))
    eng.finalize()
    return eng


@pytest.fixture(scope="module", params=[("marigold", "bf16"), ("marigold", "fp16"), ("marigold", "fp32c"), ("blend", "bf16"), ("blend", "fp16"), ("blend", "fp32c")], ids=lambda p: "-".join(p))
def eng(request, weights, golden):
    arch, precision = request.param
    e = _engine(weights, arch, golden["ctx"], precision)
    e.arch = arch
    yield e
    e.close()


def _plan(steps, cfg=SCHED, fix=None):
    from genpercept_amd.scheduler import DDIMSchedulerCustomized
    return DDIMSchedulerCustomized(**cfg).plan(steps, fix)


@pytest.mark.parametrize("steps", [1, 4, 10])
@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_denoising_loop_vs_golden(tag, steps, eng, golden, metric_log):
    d = torch.device("cuda", 0)
    rgb = torch.as_tensor(golden[f"{tag}_rgb"]).to(d)
    noise = torch.as_tensor(golden[f"{tag}_noise"]).to(d) if eng.arch == "marigold" else None
    out = eng.infer_steps(rgb, "depth", _plan(steps), noise).cpu().numpy()
    ref = golden[f"{tag}_{eng.arch}_{steps}"]
    assert out.shape == ref.shape and np.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    mean_abs = float(np.abs(out - ref).mean())
    metric_log(f"multistep_{eng.arch}[{tag},{steps} steps,{eng.precision}]", mean_abs=mean_abs, max_abs=float(np.abs(out - ref).max()))
    assert mean_abs <= TOL_MAP[eng.precision]


def test_loop_variants_vs_golden(eng, golden, metric_log):
    """3-channel mode, fix_timesteps (every step at the same t, genpercept_pipeline.py:405-406), an epsilon-prediction scheduler with
    clip_sample, trailing spacing and set_alpha_to_one."""
    d = torch.device("cuda", 0)
    tol = TOL_MAP[eng.precision]
    for tag in ("sq", "odd"):
        rgb = torch.as_tensor(golden[f"{tag}_rgb"]).to(d)
        if eng.arch == "blend":
            out = eng.infer_steps(rgb, "normal", _plan(4)).cpu().numpy()
            assert out.shape[1] == 3
            m = float(np.abs(out - golden[f"{tag}_blend_normal_4"]).mean())
            metric_log(f"multistep_blend_normal[{tag},{eng.precision}]", mean_abs=m)
            assert m <= tol
            out = eng.infer_steps(rgb, "depth", _plan(3, fix=400)).cpu().numpy()
            m = float(np.abs(out - golden[f"{tag}_blend_fix_3"]).mean())
            metric_log(f"multistep_blend_fix_timesteps[{tag},{eng.precision}]", mean_abs=m)
            assert m <= tol
        else:
            cfg = dict(SCHED, prediction_type="epsilon", clip_sample=True, set_alpha_to_one=True, steps_offset=0, timestep_spacing="trailing")
            plan = _plan(4, cfg)
            assert plan[0]["clip"] == 1.0 and plan[0]["timestep"] == 999.0
            out = eng.infer_steps(rgb, "depth", plan, torch.as_tensor(golden[f"{tag}_noise"]).to(d)).cpu().numpy()
            m = float(np.abs(out - golden[f"{tag}_marigold_eps_4"]).mean())
            metric_log(f"multistep_marigold_eps_clip[{tag},{eng.precision}]", mean_abs=m)
            assert m <= tol


def test_loop_contract(eng, golden):
    """One beta == 1 step of the loop is gp_infer bit for bit; the engine's timestep survives a loop; the same call twice gives the same
    bits (timestep cache); argument errors come back as exceptions, not as garbage."""
    d = torch.device("cuda", 0)
    rgb = torch.as_tensor(golden["sq_rgb"]).to(d)
    noise = torch.as_tensor(golden["sq_noise"]).to(d)
    if eng.arch == "blend":
        one = eng.infer(rgb, "depth")
        b1 = dict(SCHED, beta_start=1.0, beta_end=1.0)
        assert torch.equal(eng.infer_steps(rgb, "depth", _plan(1, b1)), one)
        a = eng.infer_steps(rgb, "depth", _plan(4))
        assert torch.equal(eng.infer(rgb, "depth"), one)              # timestep 1 restored after walking 751 .. 1
        assert torch.equal(eng.infer_steps(rgb, "depth", _plan(4)), a)
        eng.set_timestep(400)
        at400 = eng.infer(rgb, "depth")
        eng.infer_steps(rgb, "depth", _plan(4))
        assert torch.equal(eng.infer(rgb, "depth"), at400)
        eng.set_timestep(1)
        assert torch.equal(eng.infer(rgb, "depth"), one)
        with pytest.raises(ValueError):
            eng.infer_steps(rgb, "depth", _plan(2), noise)              # a noise sample needs the 8-channel UNet
    else:
        a = eng.infer_steps(rgb, "depth", _plan(4), noise)
        assert torch.equal(eng.infer_steps(rgb, "depth", _plan(4), noise), a)
        assert not torch.equal(eng.infer_steps(rgb, "depth", _plan(4), noise.flip(0)), a)
        both = eng.infer_steps(rgb, "depth", _plan(2), noise)
        assert torch.equal(eng.infer_steps(rgb[1:], "depth", _plan(2), noise[1:]), both[1:])   # an image does not depend on its batch
        with pytest.raises(ValueError):
            eng.infer_steps(rgb, "depth", _plan(2))                     # the 8-channel UNet needs its noise sample
        with pytest.raises(ValueError):
            eng.infer_steps(rgb, "depth", _plan(2), noise[:, :, :4])
    with pytest.raises(ValueError):
        eng.infer_steps(rgb, "depth", [], noise if eng.arch == "marigold" else None)
    bad = [dict(p, std=0.1) for p in _plan(2)]
    with pytest.raises(ValueError):
        eng.infer_steps(rgb, "depth", bad, noise if eng.arch == "marigold" else None)


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32c"])
def test_pipeline_multistep_surface(precision, weights, golden, metric_log):
    """GenPerceptPipeline(genpercept_pipeline=False): run.py:361-368 construction, __call__ with denoising_steps / ensemble_size / generator
    (genpercept_pipeline.py:199-297), against the oracle's loop + the ensembling on the same noise."""
    from PIL import Image
    from genpercept_amd import GenPerceptPipeline
    from genpercept_amd.ensemble import ensemble_depth
    from oracle import pipeline as opipe
    w = weights
    ctx = torch.as_tensor(golden["ctx"])
    img_u8 = golden["sq_rgb"][0]
    img = Image.fromarray(np.transpose(img_u8, (1, 2, 0)))
    dt = TORCH_DTYPE[precision]
    tol = TOL_MAP[precision]
    # marigold from a 4-channel UNet checkpoint: conv_in is replaced like run.py:322-323 does
    pipe = GenPerceptPipeline(unet=w["u4"], vae=w["vsd"], scheduler=dict(SCHED), text_encoder=ctx, tokenizer=None, genpercept_pipeline=False,
                              rgb_blending=False, torch_dtype=dt).to("cuda")
    assert pipe.default_denoising_steps == 10 and not pipe.rgb_blending
    gen = torch.Generator().manual_seed(2024)
    out = pipe(img, denoising_steps=4, ensemble_size=3, batch_size=3, processing_res=0, generator=gen, mode="depth", show_progress_bar=False)
    # the reference draws the initial sample in the pipeline's dtype (genpercept_pipeline.py:416-420): a half-precision run consumes the generator differently
    noise = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(2024), dtype=dt).float()
    x = opipe.normalize_rgb(torch.as_tensor(img_u8)[None]).expand(3, -1, -1, -1)
    with torch.no_grad():
        members = opipe.multi_step_infer(w["vsd"], w["vc"], w["u8"], w["uc8"], x, ctx, "depth", opipe.DDIM(**SCHED), 4, noise)
    ref, _ = ensemble_depth(members, scale_invariant=True, shift_invariant=True, max_res=50)
    m = float(np.abs(out.pred_np - ref[0, 0].numpy()).mean())
    metric_log(f"pipeline_marigold_ensemble3[{precision}]", mean_abs=m)
    assert out.pred_np.shape == (64, 64) and out.pred_colored.size == (64, 64) and m <= 2 * tol
    assert pipe.unet_config.in_channels == 8
    # unseeded call, default steps: just the contract
    o2 = pipe(img, processing_res=0, mode="depth", show_progress_bar=False)
    assert o2.pred_np.shape == (64, 64) and 0 <= o2.pred_np.min() and o2.pred_np.max() <= 1
    # rgb_blending: deterministic, no noise; fix_timesteps
    pb = GenPerceptPipeline(unet=w["u4"], vae=w["vsd"], scheduler=dict(SCHED), text_encoder=ctx, tokenizer=None, genpercept_pipeline=False,
                            rgb_blending=True, torch_dtype=dt).to("cuda")
    o3 = pb(img, denoising_steps=4, processing_res=0, mode="depth", show_progress_bar=False)
    m = float(np.abs(o3.pred_np - golden["sq_blend_4"][0, 0]).mean())
    metric_log(f"pipeline_blend_4[{precision}]", mean_abs=m)
    assert m <= tol
    o4 = pb(img, denoising_steps=3, processing_res=0, mode="depth", fix_timesteps=400, show_progress_bar=False)
    assert float(np.abs(o4.pred_np - golden["sq_blend_fix_3"][0, 0]).mean()) <= tol
    with pytest.raises(AssertionError):
        pb(img, denoising_steps=0, mode="depth")
    # the one-step pipeline with a beta != 1 scheduler: one DDIM step through the same loop (x0 = sqrt(a) latent - sqrt(1 - a) v)
    pg = GenPerceptPipeline(unet=w["u4"], vae=w["vsd"], scheduler=dict(SCHED), text_encoder=ctx, tokenizer=None, torch_dtype=dt).to("cuda")
    o5 = pg(img, processing_res=0, mode="depth")
    assert float(np.abs(o5.pred_np - golden["sq_blend_1"][0, 0]).mean()) <= tol
    with pytest.raises(AssertionError):
        pg(img, denoising_steps=4, mode="depth")


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32c"])
def test_full_sd21_widths_marigold(precision, metric_log):
    """The loop at the real SD2.1 widths (8-channel conv_in, 865.9 M parameters), 64x64 px, 4 steps, against the oracle on the host."""
    from genpercept_amd.engine import Engine
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    uc, vc = osd.UNetCfg(in_channels=8), osd.VAECfg()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 11)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 12)
    g = torch.Generator().manual_seed(78)
    rgb_u8 = torch.randint(0, 256, (1, 3, 64, 64), generator=g, dtype=torch.uint8)
    ctx = torch.randn(2, 1024, generator=g)
    noise = torch.randn(1, 4, 8, 8, generator=g)
    with torch.no_grad():
        ref = opipe.multi_step_infer(vsd, vc, usd, uc, opipe.normalize_rgb(rgb_u8), ctx, "depth", opipe.DDIM(**SCHED), 4, noise).numpy()
    eng = Engine(0, uc, vc, None, precision=precision)
    try:
        eng.load_state_dict("vae", vsd)
        eng.load_state_dict("unet", usd)
        eng.set_context(ctx)
        eng.finalize()
        d = torch.device("cuda", 0)
        out = eng.infer_steps(rgb_u8.to(d), "depth", _plan(4), noise.to(d)).cpu().numpy()
        m = float(np.abs(out - ref).mean())
        metric_log(f"full_marigold_4steps[{precision}]", mean_abs=m, max_abs=float(np.abs(out - ref).max()))
        assert m <= TOL_MAP[precision]
    finally:
        eng.close()
