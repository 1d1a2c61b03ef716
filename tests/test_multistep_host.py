"""CPU tests of the multi-step archs' host logic (marigold / rgb_blending): the scheduler against the REFERENCE's
DDIMSchedulerCustomized (tests/golden/scheduler_ref.npz) and against the oracle's independent restatement of diffusers' step, the
ensembling against the REFERENCE's ensemble_depth (tests/golden/ensemble_ref.npz), the oracle's loop against its committed goldens."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from genpercept_amd.ensemble import ensemble_depth
from genpercept_amd.scheduler import DDIMSchedulerCustomized
from oracle import pipeline as opipe
from oracle import sd21 as osd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ref_cfgs():
    g = np.load(os.path.join(GOLD, "scheduler_ref.npz"))
    return g, [(str(n), json.loads(str(g[str(n) + "/cfg"]))) for n in g["names"]]


def test_scheduler_tables_equal_the_reference_class():
    """betas / alphas_cumprod / final_alpha_cumprod / _get_variance of every hf_configs scheduler (+ scaled_linear_power, zero-SNR rescale):
    product scheduler and oracle restatement both equal the reference's own class bit for bit (same torch ops on the same build)."""
    g, cfgs = _ref_cfgs()
    assert len(cfgs) == 9
    pairs = g["variance_pairs"]
    for name, cfg in cfgs:
        s = DDIMSchedulerCustomized(**cfg)
        assert np.array_equal(s.betas.numpy(), g[name + "/betas"]), name
        assert np.array_equal(s.alphas_cumprod.numpy(), g[name + "/alphas_cumprod"]), name
        assert np.float32(s.final_alpha_cumprod) == g[name + "/final_alpha_cumprod"], name
        var = np.array([float(s._get_variance(int(t), int(p))) for t, p in pairs])
        assert np.array_equal(np.isnan(var), np.isnan(g[name + "/variance"])), name
        ok = ~np.isnan(var)
        assert np.array_equal(var[ok], g[name + "/variance"][ok]), name
        if not cfg.get("rescale_betas_zero_snr"):
            o = opipe.DDIM(**cfg)
            assert np.array_equal(o.alphas_cumprod.numpy(), g[name + "/alphas_cumprod"]), name
            assert np.float32(o.final_alpha_cumprod) == g[name + "/final_alpha_cumprod"], name
            v2 = np.array([float(o.get_variance(int(t), int(p))) for t, p in pairs])
            assert np.array_equal(v2[ok], g[name + "/variance"][ok]), name


def test_scheduler_from_pretrained_and_timesteps(tmp_path):
    cfg = dict(_class_name="DDIMScheduler", _diffusers_version="0.29.2", beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085,
               clip_sample=False, num_train_timesteps=1000, prediction_type="v_prediction", set_alpha_to_one=False, skip_prk_steps=True,
               steps_offset=1, timestep_spacing="leading", trained_betas=None)
    os.makedirs(tmp_path / "ckpt" / "scheduler")
    (tmp_path / "ckpt" / "scheduler" / "scheduler_config.json").write_text(json.dumps(cfg))
    s = DDIMSchedulerCustomized.from_pretrained(str(tmp_path / "ckpt"), subfolder="scheduler")
    assert s.config.beta_end == 0.012 and s.config.steps_offset == 1 and not s.config.clip_sample
    with pytest.raises(FileNotFoundError):
        DDIMSchedulerCustomized.from_pretrained(str(tmp_path / "ckpt"))
    s.set_timesteps(1)
    assert s.timesteps.tolist() == [1]                                   # the one-step GenPercept case
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    s.set_timesteps(50)
    assert s.timesteps[0] == 981 and s.timesteps[-1] == 1 and len(s.timesteps) == 50
    for sp, first, last in (("trailing", 999, 99), ("linspace", 999, 0)):
        t = DDIMSchedulerCustomized(**{**{k: v for k, v in cfg.items() if not k.startswith("_") and k != "skip_prk_steps"}, "timestep_spacing": sp})
        t.set_timesteps(10)
        assert t.timesteps[0] == first and t.timesteps[-1] == last and len(t.timesteps) == 10
        o = opipe.DDIM(**{**cfg, "timestep_spacing": sp})
        assert o.set_timesteps(10).tolist() == t.timesteps.tolist()
    with pytest.raises(ValueError):
        s.set_timesteps(1001)
    with pytest.raises(TypeError):
        DDIMSchedulerCustomized(beta_start=1.0, not_a_field=3)
    with pytest.raises(NotImplementedError):
        DDIMSchedulerCustomized(beta_schedule="sigmoid")
    plan = s.plan(4, fix_timesteps=400)
    assert [p["timestep"] for p in plan] == [400.0] * 4                  # genpercept_pipeline.py:405-406


@pytest.mark.parametrize("kind,clip,alpha_one", [("v_prediction", False, False), ("epsilon", True, True), ("sample", False, True)])
def test_scheduler_step_equals_the_oracle_restatement(kind, clip, alpha_one):
    """The affine form handed to the engine == diffusers' step as the oracle restates it (different grouping of the same fp32 numbers)."""
    cfg = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=clip, set_alpha_to_one=alpha_one, steps_offset=1,
               prediction_type=kind)
    s, o = DDIMSchedulerCustomized(**cfg), opipe.DDIM(**cfg)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 9, 11, generator=g)
    s.set_timesteps(5)
    o.set_timesteps(5)
    xs, xo = x.clone(), x.clone()
    for t in s.timesteps:
        m = torch.randn(2, 4, 9, 11, generator=g)
        out = s.step(m, t, xs)
        prev_o, x0_o = o.step(m, int(t), xo)
        assert torch.allclose(out.pred_original_sample, x0_o, rtol=1e-5, atol=2e-5)
        assert torch.allclose(out.prev_sample, prev_o, rtol=1e-5, atol=2e-5)
        if clip:
            assert out.pred_original_sample.abs().max() <= 1.0
        xs, xo = out.prev_sample, prev_o
    # beta == 1: x0 = -v at any timestep (the closed form gp_infer uses)
    b1 = DDIMSchedulerCustomized(beta_start=1.0, beta_end=1.0, beta_schedule="scaled_linear", prediction_type="v_prediction", clip_sample=False,
                                 set_alpha_to_one=False, steps_offset=1)
    c = b1.plan(1)[0]
    assert (c["x0_sample"], c["x0_model"], c["timestep"]) == (0.0, -1.0, 1.0)
    # eta > 0 uses the reference's customised variance and a generator
    s.set_timesteps(5)
    t0 = int(s.timesteps[1])
    a = s.step(x, t0, x, eta=1.0, generator=torch.Generator().manual_seed(1)).prev_sample
    b = s.step(x, t0, x, eta=1.0, generator=torch.Generator().manual_seed(1)).prev_sample
    assert torch.equal(a, b) and not torch.equal(a, s.step(x, t0, x).prev_sample)
    std = float(s._get_variance(t0, t0 - 200) ** 0.5)
    assert abs(s.step_coefficients(t0, eta=1.0)["std"] - std) < 1e-7


def test_ensemble_depth_equals_the_reference():
    g = np.load(os.path.join(GOLD, "ensemble_ref.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # scipy's finite differences on the fp32 parameter vector the reference hands it
        for c in g["cases"]:
            c = str(c)
            kw = {"shift_invariant": True, **json.loads(str(g[c + "/kw"]))}
            pred, unc = ensemble_depth(torch.from_numpy(g[c + "/in"]), scale_invariant=True, max_res=50, **kw)
            assert pred.shape == (1, 1, 40, 48)
            assert np.abs(pred.numpy() - g[c + "/pred"]).max() <= 1e-6, c
            if c + "/unc" in g.files:
                assert np.abs(unc.numpy() - g[c + "/unc"]).max() <= 1e-6, c
        # above max_res the alignment runs on a nearest-exact reduction; the result still spans [0,1] and undoes the distortions
        yy, xx = np.mgrid[0:96, 0:128].astype(np.float32)
        base = torch.from_numpy(0.5 + 0.4 * np.sin(xx / 17.0) * np.cos(yy / 13.0))
        d = torch.stack([base * s + t for s, t in ((1.0, 0.0), (1.7, -0.3), (0.6, 0.2), (1.2, 0.1))])[:, None]
        pred, _ = ensemble_depth(d, max_res=50)
        ref = (base - base.min()) / (base.max() - base.min())
        assert pred.min() == 0 and pred.max() == 1 and (pred[0, 0] - ref).abs().max() < 2e-2
    with pytest.raises(ValueError):
        ensemble_depth(torch.zeros(3, 2, 4, 4))
    with pytest.raises(ValueError):
        ensemble_depth(torch.zeros(3, 1, 4, 4), reduction="mode")
    with pytest.raises(ValueError):
        ensemble_depth(torch.zeros(3, 1, 4, 4), scale_invariant=False, shift_invariant=True)


def test_replace_unet_conv_in_like_run_py():
    from genpercept_amd.weights import replace_unet_conv_in
    w, b = torch.randn(8, 4, 3, 3), torch.randn(8)
    sd = replace_unet_conv_in({"conv_in.weight": w, "conv_in.bias": b})
    assert sd["conv_in.weight"].shape == (8, 8, 3, 3) and torch.equal(sd["conv_in.bias"], b)
    assert torch.equal(sd["conv_in.weight"][:, :4], w * 0.5) and torch.equal(sd["conv_in.weight"][:, 4:], w * 0.5)
    assert torch.equal(opipe.replace_unet_conv_in({"conv_in.weight": w})["conv_in.weight"], sd["conv_in.weight"])


def test_multistep_golden_is_reproducible():
    """The committed multi-step fixture is what the oracle's loop computes today (one case per arch)."""
    g = np.load(os.path.join(GOLD, "e2e_multistep.npz"))
    sched_cfg = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                     prediction_type="v_prediction", timestep_spacing="leading")
    vc, uc4 = osd.VAECfg.tiny(), osd.UNetCfg.tiny()
    uc8 = osd.UNetCfg(in_channels=8, block_out_channels=uc4.block_out_channels, num_heads=uc4.num_heads, cross_attention_dim=uc4.cross_attention_dim)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), seed=2)
    u4 = osd.synth_state_dict(osd.unet_manifest(uc4), seed=1)
    u8 = opipe.replace_unet_conv_in(u4)
    x = opipe.normalize_rgb(torch.from_numpy(g["sq_rgb"]))
    ctx = torch.from_numpy(g["ctx"])
    with torch.no_grad():
        a = opipe.multi_step_infer(vsd, vc, u8, uc8, x, ctx, "depth", opipe.DDIM(**sched_cfg), 4, torch.from_numpy(g["sq_noise"]))
        b = opipe.multi_step_infer(vsd, vc, u4, uc4, x, ctx, "depth", opipe.DDIM(**sched_cfg), 4)
    assert np.abs(a.numpy() - g["sq_marigold_4"]).max() < 1e-4
    assert np.abs(b.numpy() - g["sq_blend_4"]).max() < 1e-4
    # one step of rgb_blending with the beta == 1 scheduler is the GenPercept one-step path
    one = dict(sched_cfg, beta_start=1.0, beta_end=1.0)
    with torch.no_grad():
        c = opipe.multi_step_infer(vsd, vc, u4, uc4, x, ctx, "depth", opipe.DDIM(**one), 1)
        d = opipe.single_infer(vsd, vc, u4, uc4, x, ctx, "depth")
    assert torch.allclose(c, d, atol=1e-6)


class _FakeLib:
    @staticmethod
    def gp_latent_size(x):  # three VAE downsamples, each x -> (x - 2) // 2 + 1 (include/genpercept_hip.h)
        for _ in range(3):
            x = (x - 2) // 2 + 1
        return x


class _FakeEngine:
    """Records what the pipeline hands to the engine (no GPU): the plumbing of steps / noise / ensembling / timesteps."""

    def __init__(self):
        self.lib, self.calls, self.ctx, self.timestep = _FakeLib(), [], None, 1

    def set_context(self, e):
        self.ctx = e

    def set_timestep(self, t):
        self.timestep = t

    def infer(self, rgb, mode):
        self.calls.append(("infer", tuple(rgb.shape), mode, self.timestep))
        return torch.full((rgb.shape[0], 1 if mode == "depth" else 3, rgb.shape[2], rgb.shape[3]), 0.5)

    def infer_steps(self, rgb, mode, plan, noise):
        self.calls.append(("steps", tuple(rgb.shape), mode, [p["timestep"] for p in plan], None if noise is None else noise.clone()))
        b, _, h, w = rgb.shape
        base = torch.linspace(0.1, 0.9, h * w).reshape(1, 1, h, w).repeat(b, 1, 1, 1)
        if noise is not None:  # members that differ by an affine map, like real ensemble members
            k = noise.reshape(b, -1)[:, :1].reshape(b, 1, 1, 1)
            base = base * (1 + 0.1 * k) + 0.05 * k
        return base if mode == "depth" else base.repeat(1, 3, 1, 1)


def _fake_pipe(monkeypatch, **kw):
    from types import SimpleNamespace
    from genpercept_amd import GenPerceptPipeline
    monkeypatch.setenv("GENPERCEPT_HOST_PREPOST", "1")          # host pre / post: no device kernels involved
    monkeypatch.setattr(GenPerceptPipeline, "_device", torch.device("cpu"), raising=False)
    pipe = GenPerceptPipeline(unet={}, vae={}, text_encoder=np.zeros((2, 8), np.float32), **kw)
    pipe._device = torch.device("cpu")
    pipe._engine, pipe._timestep, pipe._ctx_loaded = _FakeEngine(), 1, None
    pipe.vae_config = SimpleNamespace(latent_channels=4)
    return pipe


def test_pipeline_multistep_plumbing_without_gpu(monkeypatch):
    sched = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                 prediction_type="v_prediction")
    img = torch.randint(0, 256, (1, 3, 64, 80), dtype=torch.uint8)
    # marigold: noise from the generator, one draw per engine call, E members in batches of `batch_size`
    pipe = _fake_pipe(monkeypatch, scheduler=sched, genpercept_pipeline=False, rgb_blending=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = pipe(img, denoising_steps=4, ensemble_size=5, batch_size=2, processing_res=0, generator=torch.Generator().manual_seed(9), mode="depth",
                   color_map=None)
    calls = pipe._engine.calls
    assert [c[0] for c in calls] == ["steps"] * 3 and [c[1][0] for c in calls] == [2, 2, 1]       # 5 members = 2 + 2 + 1
    assert all(c[3] == [751.0, 501.0, 251.0, 1.0] for c in calls)
    g = torch.Generator().manual_seed(9)
    for c in calls:                                               # the draws follow the generator's sequence, fp32, latent shape
        assert torch.equal(c[4], torch.randn((c[1][0], 4, 8, 10), generator=g))
    assert out.pred_np.shape == (64, 80) and out.pred_np.min() == 0.0 and out.pred_np.max() == 1.0   # ensembled + rescaled to [0,1]
    # rgb_blending: no noise; fix_timesteps repeats one timestep; default steps
    pb = _fake_pipe(monkeypatch, scheduler=sched, genpercept_pipeline=False, rgb_blending=True)
    pb(img, processing_res=0, mode="depth", color_map=None, fix_timesteps=300)
    (c,) = pb._engine.calls
    assert c[0] == "steps" and c[3] == [300.0] * 10 and c[4] is None
    with pytest.raises(ValueError):                               # ensemble_depth takes 1-channel maps only, like the reference's
        pb(img, denoising_steps=2, ensemble_size=2, processing_res=0, mode="normal", color_map=None)
    # the one-step pipeline: beta == 1 -> gp_infer with the timestep set; another scheduler -> one step of the loop
    p1 = _fake_pipe(monkeypatch, scheduler=dict(sched, beta_start=1.0, beta_end=1.0))
    p1(img, processing_res=0, mode="depth", color_map=None, fix_timesteps=77)
    assert p1._engine.calls == [("infer", (1, 3, 64, 80), "depth", 77)]
    p2 = _fake_pipe(monkeypatch, scheduler=sched)
    p2(img, processing_res=0, mode="depth", color_map=None)
    assert p2._engine.calls[0][0] == "steps" and p2._engine.calls[0][3] == [1.0] and p2._engine.calls[0][4] is None
    with pytest.raises(AssertionError):
        p2(img, denoising_steps=2, mode="depth")


def test_from_pretrained_picks_up_the_checkpoints_scheduler_and_model_index(tmp_path):
    """run.py:361-368 leaves `scheduler` unset for archs marigold / rgb_blending: DiffusionPipeline.from_pretrained then loads
    <checkpoint>/scheduler, and the registered defaults (genpercept_pipeline.py:128-132) come from model_index.json."""
    import json
    from genpercept_amd import GenPerceptPipeline
    ck = tmp_path / "ckpt"
    (ck / "scheduler").mkdir(parents=True)
    cfg = dict(_class_name="DDIMScheduler", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
               set_alpha_to_one=False, steps_offset=1, prediction_type="v_prediction", timestep_spacing="trailing")
    (ck / "scheduler" / "scheduler_config.json").write_text(json.dumps(cfg))
    (ck / "model_index.json").write_text(json.dumps({"_class_name": "GenPerceptPipeline", "default_denoising_steps": 4,
                                                     "default_processing_resolution": 512, "rgb_blending": True}))
    pipe = GenPerceptPipeline.from_pretrained(str(ck), unet={}, vae={}, text_encoder=np.zeros((2, 8), np.float32), genpercept_pipeline=False)
    assert pipe.scheduler is not None and pipe.scheduler.config.timestep_spacing == "trailing" and pipe.scheduler.config.beta_end == 0.012
    assert pipe.default_denoising_steps == 4 and pipe.default_processing_resolution == 512 and pipe.rgb_blending is True
    # explicit keywords win over model_index.json; the one-step pipeline still forces steps = 1 (genpercept_pipeline.py:115-117)
    p2 = GenPerceptPipeline.from_pretrained(str(ck), unet={}, vae={}, text_encoder=np.zeros((2, 8), np.float32), default_processing_resolution=640)
    assert p2.default_processing_resolution == 640 and p2.default_denoising_steps == 1
    # without a scheduler folder the multi-step archs still refuse to build
    ck2 = tmp_path / "bare"
    ck2.mkdir()
    with pytest.raises(ValueError):
        GenPerceptPipeline.from_pretrained(str(ck2), unet={}, vae={}, text_encoder=np.zeros((2, 8), np.float32), genpercept_pipeline=False)


def test_pipeline_host_semantics_follow_the_reference(monkeypatch):
    """genpercept_pipeline.py:262-266 (batch size from the RESIZED image), :403 (the UNet timestep comes from scheduler.set_timesteps(1)),
    :416-420 (noise drawn in the pipeline's dtype), :474-482 (the DPT-head branch never consults scheduler.step)."""
    import genpercept_amd.pipeline as gp
    img = torch.randint(0, 256, (1, 3, 64, 80), dtype=torch.uint8)
    b11 = dict(beta_start=1.0, beta_end=1.0, beta_schedule="linear", prediction_type="v_prediction", clip_sample=False, steps_offset=1)
    # timestep of the one-step path: leading + offset 1 -> 1; trailing -> 999; offset 0 -> 0
    for extra, want in ((dict(), 1), (dict(timestep_spacing="trailing"), 999), (dict(steps_offset=0), 0)):
        p = _fake_pipe(monkeypatch, scheduler=dict(b11, **extra))
        p._timestep = -1
        p(img, processing_res=0, mode="depth", color_map=None)
        assert p._engine.calls == [("infer", (1, 3, 64, 80), "depth", want)], (extra, p._engine.calls)
    # half-precision marigold draws its noise in fp16 (then widened), fp32 pipelines in fp32
    sched = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                 prediction_type="v_prediction")
    for dt in (torch.float16, torch.float32, None):
        p = _fake_pipe(monkeypatch, scheduler=sched, genpercept_pipeline=False, rgb_blending=False, torch_dtype=dt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p(img, denoising_steps=2, processing_res=0, generator=torch.Generator().manual_seed(3), mode="depth", color_map=None)
        want = torch.randn((1, 4, 8, 10), generator=torch.Generator().manual_seed(3), dtype=dt or torch.float32).float()
        assert torch.equal(p._engine.calls[0][4], want) and p._engine.calls[0][4].dtype == torch.float32
    # find_batch_size sees the longest edge AFTER resize_max_res, not processing_res
    seen = []
    monkeypatch.setattr(gp, "find_batch_size", lambda ensemble_size, input_res, dtype=None: seen.append((ensemble_size, input_res, dtype)) or 1)
    p = _fake_pipe(monkeypatch, scheduler=b11)
    p(torch.randint(0, 256, (1, 3, 100, 50), dtype=torch.uint8), processing_res=32, mode="depth", color_map=None)
    p(torch.randint(0, 256, (1, 3, 40, 56), dtype=torch.uint8), processing_res=0, mode="depth", color_map=None)
    assert [s[:2] for s in seen] == [(1, 32), (1, 56)] and seen[0][2] == torch.float32
    # a customized head routes to the one-step engine call whatever the scheduler's prediction type
    p = _fake_pipe(monkeypatch, scheduler=dict(b11, prediction_type="epsilon", clip_sample=True), customized_head={"neck.fusion_stage.x": torch.zeros(1)},
                   head_type="identity")
    p(img, processing_res=0, mode="disparity", color_map=None)
    assert p._engine.calls[0][0] == "infer"
