"""CPU tests of the oracle itself: what pins the restatement in lieu of reference tests."""
import os

import numpy as np
import pytest
import torch

from oracle import dpt as odpt
from oracle import pipeline as opipe
from oracle import sd21 as osd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_public_parameter_and_tensor_counts():
    """SD2.1 public figures: the layer list, widths and bias/no-bias choices are structurally right."""
    u = osd.unet_manifest()
    v = osd.vae_manifest()
    assert len(u) == 686 and osd.count_params(u) == 865_910_724
    assert len(v) == 248 and osd.count_params(v) == 83_653_863
    enc = {k: s for k, s in v.items() if k.startswith("encoder")}
    dec = {k: s for k, s in v.items() if k.startswith("decoder")}
    assert len(enc) == 106 and osd.count_params(enc) == 34_163_592
    assert osd.count_params(dec) == 49_490_179
    assert osd.count_params(odpt.dpt_manifest()) == 18_474_753
    # reference comments: run.py:61-62 "[320, 4, 3, 3]", custom_unet.py:366 "1, 1280, 24, 24"
    assert u["conv_in.weight"] == (320, 4, 3, 3)
    assert u["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert u["up_blocks.3.resnets.0.conv1.weight"] == (320, 960, 3, 3)


def test_dpt_oracle_matches_reference_outputs():
    """tests/golden/dpt_head_ref.npz was produced by the reference's own dpt_head.py (make_goldens.py)."""
    g = np.load(os.path.join(GOLD, "dpt_head_ref.npz"))
    sd = osd.synth_state_dict(odpt.dpt_manifest(), int(g["seed"]))
    for tag in "ab":
        h, w = (int(x) for x in g[f"{tag}_hw"])
        gen = torch.Generator().manual_seed(100 + h * w)
        feats = [torch.randn(1, 320, h, w, generator=gen), torch.randn(1, 640, h, w, generator=gen),
                 torch.randn(1, 1280, h // 2, w // 2, generator=gen), torch.randn(1, 1280, h // 4, w // 4, generator=gen)]
        with torch.no_grad():
            y = odpt.dpt_head_forward(sd, feats).numpy()
        np.testing.assert_allclose(y, g[f"{tag}_out"], rtol=1e-5, atol=1e-5)


def _odd_dpt_feats(h, w):
    h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    h4, w4 = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    gen = torch.Generator().manual_seed(200 + h * w)
    return [torch.randn(1, 320, h, w, generator=gen), torch.randn(1, 640, h, w, generator=gen),
            torch.randn(1, 1280, h2, w2, generator=gen), torch.randn(1, 1280, h4, w4, generator=gen)]


def test_dpt_oracle_matches_reference_outputs_odd_shapes():
    """dpt_head_ref_odd.npz: the reference's dpt_head.py on the feature shapes of latents 9x11, 13x10 and 29x39, where the fused map
    and the next neck feature differ in size -- pins the bilinear-resize branch (dpt_head.py:297-300 / oracle/dpt.py) to the reference."""
    g = np.load(os.path.join(GOLD, "dpt_head_ref_odd.npz"))
    sd = osd.synth_state_dict(odpt.dpt_manifest(), int(g["seed"]))
    for tag in "cde":
        h, w = (int(x) for x in g[f"{tag}_hw"])
        with torch.no_grad():
            y = odpt.dpt_head_forward(sd, _odd_dpt_feats(h, w)).numpy()
        assert y.shape == g[f"{tag}_out"].shape
        np.testing.assert_allclose(y, g[f"{tag}_out"], rtol=1e-5, atol=2e-5)


def test_scheduler_identity_and_timesteps():
    """beta == 1 => pred_original_sample == -model_output for ANY t; one leading-spaced step with offset 1 => t = 1."""
    assert opipe.ddim_timesteps(1).tolist() == [1]
    g = torch.Generator().manual_seed(0)
    v, x = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    for t in (0, 1, 500, 999):
        assert torch.equal(opipe.ddim_pred_original_sample(v, x, t), -v)
    # a non-degenerate schedule is NOT the identity (guards against a vacuous check)
    assert not torch.allclose(opipe.ddim_pred_original_sample(v, x, 1, 0.00085, 0.012), -v)


def test_timestep_embedding_cos_first():
    e = osd.timestep_embedding(torch.tensor([1.0]), 320)
    assert e.shape == (1, 320)
    assert abs(e[0, 0].item() - np.cos(1.0)) < 1e-6 and abs(e[0, 160].item() - np.sin(1.0)) < 1e-6


def test_e2e_golden_is_reproducible():
    """The committed e2e fixture is what the oracle computes today (same torch build => same bits up to fp reassociation)."""
    g = np.load(os.path.join(GOLD, "e2e_tiny.npz"))
    uc, vc, dc = osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg.tiny()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 1)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
    dsd = osd.synth_state_dict(odpt.dpt_manifest(dc), 3)
    rgb = opipe.normalize_rgb(torch.as_tensor(g["odd_rgb_u8"]))
    ctx = torch.as_tensor(g["odd_ctx"])
    with torch.no_grad():
        lat = osd.encode_rgb(vsd, vc, rgb)
        depth = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "depth")
        disp = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "disparity", dpt_sd=dsd)
    np.testing.assert_allclose(lat.numpy(), g["odd_latent"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(depth.numpy(), g["odd_depth"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(disp.numpy(), g["odd_disp"], rtol=1e-3, atol=2e-4)
    assert depth.shape == (1, 1, 72, 88) and disp.shape == (1, 1, 96, 96)


def test_unet_feature_shapes_and_upsample_size_path():
    """custom_unet.py:109-119,377-378: latent dims not divisible by 8 force nearest-to-skip-size upsampling."""
    uc = osd.UNetCfg.tiny()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 1)
    ctx = torch.zeros(1, 2, uc.cross_attention_dim)
    with torch.no_grad():
        v, feats = osd.unet_forward(usd, uc, torch.randn(1, 4, 15, 20), 1, ctx)
    assert v.shape == (1, 4, 15, 20)
    assert [tuple(f.shape[1:]) for f in feats] == [(256, 4, 5), (256, 8, 10), (128, 15, 20), (64, 15, 20)]
    with torch.no_grad():
        none, feats2 = osd.unet_forward(usd, uc, torch.randn(1, 4, 16, 16), 1, ctx, return_feature=True)
    assert none is None and len(feats2) == 4


def test_metric_restatement_matches_reference_functions():
    """genpercept_amd.eval_metrics vs outputs of src/util/metric.py + alignment.py (tests/golden/metrics_ref.npz)."""
    from genpercept_amd import eval_metrics as em
    g = np.load(os.path.join(GOLD, "metrics_ref.npz"))
    gt, pred, mask = g["gt"], g["pred"], g["mask"]
    aligned, s, t = em.align_depth_least_square(gt[0], pred[0], mask[0])
    np.testing.assert_allclose(aligned, g["aligned0"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose([s, t], g["scale_shift0"], rtol=1e-5)
    np.testing.assert_allclose(em.depth2disparity(gt[0])[0], g["disp0"], rtol=1e-6)
    al = np.clip(g["aligned0"], 1e-3, 10.0)[None].astype(np.float64)
    for name, fn in em.METRICS.items():
        np.testing.assert_allclose(fn(al, gt[:1].astype(np.float64), mask[:1]), float(g["m_" + name]), rtol=2e-5, err_msg=name)
    r = em.evaluate_depth(pred[0], gt[0], mask[0])
    np.testing.assert_allclose(r["abs_relative_difference"], float(g["m_abs_relative_difference"]), rtol=2e-5)


def test_oracle_matches_reference_executed_orchestration():
    """tests/golden/refexec_tiny.npz holds what the REFERENCE's own code computes when it is EXECUTED (make_goldens.py: refexec) --
    CustomUNet2DConditionModel.forward (custom_unet.py:34-427: skip stack and popping order, forward_upsample_size / upsample_size on the
    9x11 latent, multi_level_feats, return_feature) and GenPerceptPipeline.__call__ / single_infer / encode_rgb / decode_pred
    (genpercept_pipeline.py:146-337,375-526: latent scale, `pred_original_sample`, channel mean, clip / shift, feats[::-1], min-max, the
    [rgb_latent, pred_latent] order and noise of the multi-step archs) -- over stub diffusers base classes whose blocks are the oracle's
    resnet / transformer / VAE functions.  The oracle's own orchestration must reproduce it: these parts of the path are no longer only
    restated (the inside of the diffusers blocks still is: parity stays "partial")."""
    r = np.load(os.path.join(GOLD, "refexec_tiny.npz"))
    g = np.load(os.path.join(GOLD, "e2e_tiny.npz"))
    uc, vc, dc = osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg.tiny()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 1)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
    dsd = osd.synth_state_dict(odpt.dpt_manifest(dc), 3)
    tol = dict(rtol=0, atol=2e-5)  # fp32 reassociation between torch builds / thread counts; the generator asserted 1e-5 in its own process
    with torch.no_grad():
        for tag in ("sq", "odd"):
            rgb = opipe.normalize_rgb(torch.as_tensor(g[f"{tag}_rgb_u8"]))
            ctx = torch.as_tensor(g[f"{tag}_ctx"])
            b = rgb.shape[0]
            lat = torch.as_tensor(g[f"{tag}_latent"])
            v, feats = osd.unet_forward(usd, uc, lat, 1, ctx[None].expand(b, -1, -1))
            np.testing.assert_allclose(v.numpy(), r[f"{tag}_unet"], **tol)
            assert len(feats) == 4
            for i, f in enumerate(feats):  # the reference's multi_level_feats order (custom_unet.py:400), stored as fp16
                assert f.shape == r[f"{tag}_feat{i}"].shape
                np.testing.assert_allclose(f.numpy(), r[f"{tag}_feat{i}"].astype(np.float32), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(osd.encode_rgb(vsd, vc, rgb).numpy(), r[f"{tag}_latent"], **tol)
            for mode in ("depth", "normal"):
                np.testing.assert_allclose(opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, mode).numpy(), r[f"{tag}_{mode}"], **tol)
            np.testing.assert_allclose(opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "disparity", dpt_sd=dsd).numpy(), r[f"{tag}_disp"], **tol)
            np.testing.assert_allclose(opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "depth", timestep=400).numpy(), r[f"{tag}_depth_fix400"], **tol)
            # __call__'s tail on image 0 (:319-337): squeeze, clip(0, 1); HWC for 3-channel maps; uint8 image of a colour-less mode = (x * 255).astype(u8)
            np.testing.assert_allclose(r[f"{tag}_call_depth_np"], r[f"{tag}_depth"][0, 0].clip(0, 1), **tol)  # (batch of 1 vs batch of 2: fp32 reassociation)
            np.testing.assert_allclose(r[f"{tag}_call_normal_np"], np.transpose(r[f"{tag}_normal"][0], (1, 2, 0)).clip(0, 1), **tol)
            np.testing.assert_allclose(r[f"{tag}_call_disp_np"], r[f"{tag}_disp"][0, 0].clip(0, 1), **tol)
            assert np.array_equal(r[f"{tag}_call_normal_colored"], (r[f"{tag}_call_normal_np"] * 255.0).astype(np.uint8))
            from genpercept_amd import image_util as iu
            col = (iu.colorize_depth_maps(r[f"{tag}_call_depth_np"], 0, 1, cmap="Spectral").squeeze() * 255).astype(np.uint8)
            assert np.array_equal(iu.chw2hwc(col), r[f"{tag}_call_depth_colored"])
        ms = np.load(os.path.join(GOLD, "e2e_multistep.npz"))
        sched = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                     prediction_type="v_prediction", timestep_spacing="leading")
        uc8 = osd.UNetCfg(in_channels=8, block_out_channels=uc.block_out_channels, num_heads=uc.num_heads, cross_attention_dim=uc.cross_attention_dim)
        unet8 = opipe.replace_unet_conv_in(usd)
        ctx = torch.as_tensor(ms["ctx"])
        for tag in ("sq", "odd"):
            x = opipe.normalize_rgb(torch.as_tensor(ms[f"{tag}_rgb"]))
            noise = torch.as_tensor(ms[f"{tag}_noise"])
            got = opipe.multi_step_infer(vsd, vc, unet8, uc8, x, ctx, "depth", opipe.DDIM(**sched), 4, noise).numpy()
            np.testing.assert_allclose(got, r[f"{tag}_marigold_4"], **tol)
            got = opipe.multi_step_infer(vsd, vc, usd, uc, x, ctx, "depth", opipe.DDIM(**sched), 4).numpy()
            np.testing.assert_allclose(got, r[f"{tag}_blend_4"], **tol)


def test_oracle_matches_genpercept_v1_single_infer_executed():
    """tests/golden/refexec_v1_tiny.npz: the SECOND statement of the one-step math in the reference's tree -- GenPercept_v1's `single_infer`
    (pipeline_genpercept.py:263-309: `timesteps = [1]`, `pred_latent = - unet_pred`, encode_rgb / decode_pred :312-356, no scheduler object) --
    EXECUTED over the stub bases with the v1 tree's own `empty_text_embed.npy` as context: all 77 rows (what v1 feeds) and rows [0:2] (BOS, EOS:
    what v2's "do_not_pad" tokenisation feeds, genpercept_pipeline.py:360-372).  v1 returns the clipped [-1, 1] prediction (the shift to [0, 1] is
    in its __call__), the oracle the [0, 1] map of v2's single_infer (:469-472): they must agree through (x + 1) / 2."""
    r = np.load(os.path.join(GOLD, "refexec_v1_tiny.npz"))
    assert r["embed_77"].shape == (77, 1024) and np.array_equal(r["embed_rows_0_2"], r["embed_77"][:2])
    uc = osd.UNetCfg(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=1024)
    vc = osd.VAECfg.tiny()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 21)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
    with torch.no_grad():
        for tag in ("sq", "odd"):
            rgb = opipe.normalize_rgb(torch.as_tensor(r[f"{tag}_rgb_u8"]))
            for cname, ctx in (("ctx2", r["embed_rows_0_2"]), ("ctx77", r["embed_77"])):
                for mode in ("depth", "normal"):
                    out = opipe.single_infer(vsd, vc, usd, uc, rgb, torch.as_tensor(ctx).float(), mode).numpy()
                    v1 = r[f"{tag}_{cname}_{mode}"]
                    assert v1.min() >= -1.0 and v1.max() <= 1.0 and v1.shape == out.shape
                    np.testing.assert_allclose(out, (v1 + 1.0) / 2.0, rtol=0, atol=2e-5)
            # the 77-row context is NOT the 2-row context (75 padding keys take part in the softmax): the two fixtures differ, the engine has
            # to honour the context it is given (gp_set_context folds L = 2 and keeps the general kernel for other lengths)
            assert np.abs(r[f"{tag}_ctx2_depth"] - r[f"{tag}_ctx77_depth"]).max() > 1e-4


@pytest.mark.parametrize("shape", [(1, 8, 5, 7), (2, 16, 16, 16), (1, 4, 17, 33), (1, 3, 1, 1)])
def test_upsample_conv_phase_identity(shape):
    """The identity conv3x3_halo3_kernel<..., PH> is built on (genpercept_amd/csrc/conv_halo.hip; diffusers Upsample2D = nearest x2 then conv3x3, used by
    custom_unet.py:372-400's up blocks and the VAE decoder): output pixel (2y + a, 2x + b) of conv3x3(upsample2(s)) is a 2 x 2-tap convolution of the
    SOURCE map with the kernel rows / columns that fall onto the same source pixel summed -- a = 0: {w0}, {w1 + w2}; a = 1: {w0 + w1}, {w2} -- and the
    zero padding of the upsampled map is the zero padding of the source map.  Checked in float64 on maps with odd sizes down to one pixel; the kernel's
    own packing (engine.hip: pack_phase_rows) is checked against this construction on the GPU (tests/test_kernels_gpu.py)."""
    import torch.nn.functional as F
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(b, c, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(5, c, 3, 3, generator=g, dtype=torch.float64)
    bias = torch.randn(5, generator=g, dtype=torch.float64)
    exact = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)
    rng = {0: [(0, 0), (1, 2)], 1: [(0, 1), (2, 2)]}     # phase -> kernel index ranges of its two taps
    out = torch.empty_like(exact)
    for a in (0, 1):
        for bb in (0, 1):
            wp = torch.zeros(5, c, 2, 2, dtype=torch.float64)
            for ty, (y0, y1) in enumerate(rng[a]):
                for tx, (x0, x1) in enumerate(rng[bb]):
                    wp[:, :, ty, tx] = wt[:, :, y0:y1 + 1, x0:x1 + 1].sum(dim=(2, 3))
            out[:, :, a::2, bb::2] = F.conv2d(F.pad(x, (1 - bb, bb, 1 - a, a)), wp, bias)   # source rows y - 1 + a .. y + a, columns x - 1 + b .. x + b
    assert float((out - exact).abs().max()) < 1e-12
