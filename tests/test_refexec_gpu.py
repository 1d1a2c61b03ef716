"""HIP engine and pipeline against tests/golden/refexec_tiny.npz: outputs of the REFERENCE's own orchestration code (custom_unet.py:34-427,
genpercept_pipeline.py:146-337,375-526) EXECUTED over stub diffusers base classes whose blocks are the oracle's functions
(tests/golden/make_goldens.py: refexec).  What is compared here is therefore what the reference's tree itself computes around the diffusers
blocks -- skip order, upsample_size on a latent not divisible by 8, multi_level_feats order, latent scale, -v, channel mean, clip / shift,
per-image min-max, __call__'s squeeze / clip / colour -- not the build's restatement of it.  Tolerances: tests/test_e2e_gpu.py TOLS."""
import os

import numpy as np
import pytest
import torch

from test_e2e_gpu import TOLS, _engine, rel_rms, tiny_weights  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def refexec():
    return np.load(os.path.join(GOLD, "refexec_tiny.npz"))


@pytest.fixture(scope="module")
def inputs():
    return np.load(os.path.join(GOLD, "e2e_tiny.npz"))


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32c"])
@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_engine_vs_reference_executed(tag, precision, tiny_weights, refexec, inputs, metric_log):
    d = torch.device("cuda", 0)
    tol = TOLS[precision]
    eng = _engine(tiny_weights, False, inputs[f"{tag}_ctx"], precision)
    try:
        rgb = torch.as_tensor(inputs[f"{tag}_rgb_u8"]).to(d)
        lat = eng.vae_encode(rgb)
        r = rel_rms(lat, refexec[f"{tag}_latent"])
        metric_log(f"refexec_latent[{tag},{precision}]", rel_rms=r)
        assert r <= tol["stage"]
        v, feats = eng.unet(torch.as_tensor(refexec[f"{tag}_latent"]).to(d), want_sample=True, want_feats=True)
        r = rel_rms(v, refexec[f"{tag}_unet"])
        metric_log(f"refexec_unet[{tag},{precision}]", rel_rms=r)
        assert r <= tol["stage"]
        assert len(feats) == 4
        for i, f in enumerate(feats):  # the reference's multi_level_feats order and shapes (custom_unet.py:365-400)
            assert tuple(f.shape) == refexec[f"{tag}_feat{i}"].shape
            assert rel_rms(f, refexec[f"{tag}_feat{i}"].astype(np.float32)) <= tol["stage"]
        for mode in ("depth", "normal"):
            out = eng.infer(rgb, mode).cpu().numpy()
            ref = refexec[f"{tag}_{mode}"]
            assert out.shape == ref.shape
            e = float(np.abs(out - ref).mean())
            metric_log(f"refexec_{mode}[{tag},{precision}]", mean_abs=e, max_abs=float(np.abs(out - ref).max()))
            assert e <= tol["map_mean"]
        eng.set_timestep(400)  # fix_timesteps (genpercept_pipeline.py:405-408)
        e = float(np.abs(eng.infer(rgb, "depth").cpu().numpy() - refexec[f"{tag}_depth_fix400"]).mean())
        metric_log(f"refexec_depth_fix400[{tag},{precision}]", mean_abs=e)
        assert e <= tol["map_mean"]
    finally:
        eng.close()
    eng = _engine(tiny_weights, True, inputs[f"{tag}_ctx"], precision)
    try:
        out = eng.infer(torch.as_tensor(inputs[f"{tag}_rgb_u8"]).to(d), "disparity").cpu().numpy()
        ref = refexec[f"{tag}_disp"]  # the reference called once per image: per-image min-max (:480-482)
        assert out.shape == ref.shape
        e = float(np.abs(out - ref).mean())
        metric_log(f"refexec_disp[{tag},{precision}]", mean_abs=e)
        assert e <= 2 * tol["map_mean"]
    finally:
        eng.close()


@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_pipeline_call_vs_reference_call(tag, tiny_weights, refexec, inputs, metric_log):
    """GenPerceptPipeline.__call__ (device pre / post) against the reference's __call__ on the same uint8 tensor: pred_np and the colour image."""
    from genpercept_amd import GenPerceptPipeline
    sched = dict(beta_start=1.0, beta_end=1.0, beta_schedule="linear", prediction_type="v_prediction", clip_sample=False, steps_offset=1)
    rgb = torch.as_tensor(inputs[f"{tag}_rgb_u8"][:1])
    kw = dict(denoising_steps=1, ensemble_size=1, processing_res=0, match_input_res=False, batch_size=1, show_progress_bar=False)
    pipe = GenPerceptPipeline(unet=tiny_weights["usd"], vae=tiny_weights["vsd"], scheduler=sched, text_encoder=inputs[f"{tag}_ctx"], tokenizer=None,
                              torch_dtype=torch.float16)
    try:
        o = pipe(rgb, color_map="Spectral", mode="depth", **kw)
        ref = refexec[f"{tag}_call_depth_np"]
        assert o.pred_np.shape == ref.shape and o.pred_np.dtype == ref.dtype
        e = float(np.abs(o.pred_np - ref).mean())
        col = np.abs(np.asarray(o.pred_colored).astype(np.int32) - refexec[f"{tag}_call_depth_colored"].astype(np.int32))
        metric_log(f"refexec_call_depth[{tag}]", mean_abs=e, colored_mean_abs_lsb=float(col.mean()), colored_max_abs_lsb=int(col.max()))
        assert e <= TOLS["fp16"]["map_mean"]
        assert np.asarray(o.pred_colored).shape == refexec[f"{tag}_call_depth_colored"].shape and col.mean() <= 1.0
        o = pipe(rgb, color_map=None, mode="normal", **kw)
        ref = refexec[f"{tag}_call_normal_np"]
        assert o.pred_np.shape == ref.shape  # HWC (:331-332)
        assert float(np.abs(o.pred_np - ref).mean()) <= TOLS["fp16"]["map_mean"]
        img = np.abs(np.asarray(o.pred_colored).astype(np.int32) - refexec[f"{tag}_call_normal_colored"].astype(np.int32))
        assert img.max() <= 2 and img.mean() <= 0.6  # (x * 255).astype(uint8) of maps that differ by < 1e-3
    finally:
        if pipe._engine is not None:
            pipe._engine.close()
    head = {k: v for k, v in tiny_weights["dsd"].items()}
    pipe = GenPerceptPipeline(unet={k: v for k, v in tiny_weights["usd"].items() if not k.startswith(("conv_out", "conv_norm_out"))}, vae=tiny_weights["vsd"],
                              scheduler=sched, text_encoder=inputs[f"{tag}_ctx"], tokenizer=None, customized_head=head, head_type="identity",
                              torch_dtype=torch.float16)
    try:
        o = pipe(rgb, color_map="Spectral", mode="disparity", **kw)
        ref = refexec[f"{tag}_call_disp_np"]
        assert o.pred_np.shape == ref.shape
        e = float(np.abs(o.pred_np - ref).mean())
        metric_log(f"refexec_call_disp[{tag}]", mean_abs=e)
        assert e <= 2 * TOLS["fp16"]["map_mean"]
    finally:
        if pipe._engine is not None:
            pipe._engine.close()


@pytest.fixture(scope="module")
def v1_weights():
    """the tiny topology with cross_attention_dim = 1024 (the width of the v1 tree's empty_text_embed.npy), seeds of make_goldens.py: refexec_v1"""
    from oracle import dpt as odpt
    from oracle import sd21 as osd
    uc = osd.UNetCfg(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=1024)
    vc = osd.VAECfg.tiny()
    return dict(uc=uc, vc=vc, dc=odpt.DPTCfg.tiny(), usd=osd.synth_state_dict(osd.unet_manifest(uc), 21), vsd=osd.synth_state_dict(osd.vae_manifest(vc), 2),
                dsd=None)


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32c"])
@pytest.mark.parametrize("cname", ["ctx2", "ctx77"])
def test_engine_vs_genpercept_v1_single_infer(cname, precision, v1_weights, metric_log):
    """HIP engine against what GenPercept_v1's `single_infer` (pipeline_genpercept.py:263-309) computes when EXECUTED
    (tests/golden/refexec_v1_tiny.npz), with the v1 tree's shipped empty-prompt embedding as context: rows [0:2] take the folded 2-token
    cross-attention path, all 77 rows the general cross-attention kernel."""
    g = np.load(os.path.join(GOLD, "refexec_v1_tiny.npz"))
    ctx = g["embed_rows_0_2"] if cname == "ctx2" else g["embed_77"]
    d = torch.device("cuda", 0)
    tol = TOLS[precision]
    eng = _engine(v1_weights, False, ctx.astype(np.float32), precision)
    try:
        for tag in ("sq", "odd"):
            rgb = torch.as_tensor(g[f"{tag}_rgb_u8"]).to(d)
            for mode in ("depth", "normal"):
                out = eng.infer(rgb, mode).cpu().numpy()
                ref = (g[f"{tag}_{cname}_{mode}"] + 1.0) / 2.0   # v1 returns the clipped [-1, 1] map; the shift is in its __call__
                assert out.shape == ref.shape
                e = float(np.abs(out - ref).mean())
                metric_log(f"refexec_v1_{mode}[{tag},{cname},{precision}]", mean_abs=e, max_abs=float(np.abs(out - ref).max()))
                assert e <= tol["map_mean"]
    finally:
        eng.close()
