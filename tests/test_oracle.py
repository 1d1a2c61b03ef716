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
