"""Device-side pre / post processing (gp_preprocess / gp_postprocess, csrc/prepost.hip) against the host path
(genpercept_amd/image_util.py = the reference's torchvision / matplotlib recipe, pinned to the reference's outputs in test_host.py).

Measured on MI355X + EPYC 9575F host: the device resize is BIT-EXACT with ATen's CPU kernel on every shape below, uint8 and fp32 (the kernel
repeats the separable anti-aliased filter in fp32, weight computation and accumulation order included).  The gates leave room only for a host
whose ATen build contracts multiply-adds differently: uint8 at most 1 LSB on at most 0.01 % of the pixels, fp32 5e-7 absolute on [0,1] maps;
colour LUT and quantisation: bit-exact given the same fp32 input.  Bicubic (r4: Keys cubic weights, evaluated without contraction on the device while
the host build may fuse them) gets 1 LSB on at most 0.1 % of the uint8 pixels and 2e-6 on [0,1] maps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(480, 640), (1200, 1600), (3024, 4032), (500, 333), (768, 768), (97, 1031)])
@pytest.mark.parametrize("mode", ["bilinear", "nearest-exact", "bicubic"])
def test_preprocess_resize_max_res_u8(hw, mode, metric_log):
    from genpercept_amd import engine as ge
    from genpercept_amd import image_util as iu
    g = torch.Generator().manual_seed(hw[0] + hw[1])
    img = torch.randint(0, 256, (2, 3, hw[0], hw[1]), generator=g, dtype=torch.uint8)
    img[0, :, : hw[0] // 2] = (torch.arange(hw[1]) % 256).to(torch.uint8)  # smooth part next to the noise
    ref = iu.resize_max_res(img, 768, mode)
    size = ge.resize_max_res_size(hw[0], hw[1], 768)
    assert tuple(ref.shape[-2:]) == size
    out = ge.preprocess(img.cuda(), size, mode).cpu()
    assert out.dtype == torch.uint8 and out.shape == ref.shape
    d = (out.int() - ref.int()).abs()
    frac = float((d > 0).float().mean())
    metric_log(f"preprocess_u8{hw}{mode}", max_lsb=int(d.max()), frac_diff=frac)
    assert int(d.max()) <= (0 if mode == "nearest-exact" else 1) and frac <= (1e-3 if mode == "bicubic" else 1e-4)


@pytest.mark.parametrize("hw", [(480, 640), (1200, 1600), (500, 333), (768, 768), (97, 1031), (300, 400)])
@pytest.mark.parametrize("mode", ["bilinear", "nearest-exact", "bicubic"])
def test_preprocess_float_image(hw, mode, metric_log):
    """A float `input_image` tensor (trainer-style rgb_int in [0, 255]): fp32 resize without rounding, then x / 255 * 2 - 1 (genpercept_pipeline.py:236-247)."""
    from genpercept_amd import engine as ge
    from genpercept_amd import image_util as iu
    g = torch.Generator().manual_seed(hw[0] * 7 + hw[1])
    img = torch.randint(0, 256, (2, 3, hw[0], hw[1]), generator=g, dtype=torch.uint8).float()
    img[0, :, : hw[0] // 2] = (torch.arange(hw[1]) % 256).float()
    img = (img * 0.5 + 63.75).contiguous()  # inside [63.75, 191.25]: the cubic filter's overshoot stays inside [0, 255]
    res = 768 if max(hw) != 400 else 640     # (300, 400) -> 480 x 640: an upscale
    size = ge.resize_max_res_size(hw[0], hw[1], res)
    ref_r = iu.resize_max_res(img, res, mode)
    ref_n = ref_r / 255.0 * 2.0 - 1.0
    out_r = ge.preprocess(img.cuda(), size, mode).cpu()
    out_n = ge.preprocess(img.cuda(), size, mode, normalize=True).cpu()
    assert out_r.dtype == out_n.dtype == torch.float32 and out_r.shape == out_n.shape == ref_r.shape
    er, en = float((out_r - ref_r).abs().max()), float((out_n - ref_n).abs().max())
    metric_log(f"preprocess_f32{hw}{mode}", max_abs_0_255=er, max_abs_normalised=en)
    assert er <= (0.0 if mode == "nearest-exact" else 2e-4) and en <= (0.0 if mode == "nearest-exact" else 2e-6)
    if size == hw:  # no resize: the normalisation alone, bit for bit
        assert torch.equal(out_n, img / 255.0 * 2.0 - 1.0)


@pytest.mark.parametrize("case", [(1, 576, 768, 3024, 4032), (3, 768, 768, 500, 500), (1, 384, 512, 384, 512), (1, 768, 576, 60, 45), (2, 96, 128, 97, 131)])
@pytest.mark.parametrize("mode", ["bilinear", "nearest-exact", "bicubic"])
def test_postprocess_resize_clip_colorize_quantize(case, mode, metric_log):
    from genpercept_amd import engine as ge
    from genpercept_amd import image_util as iu
    c, h, w, ho, wo = case
    g = torch.Generator().manual_seed(h + w + ho)
    yy, xx = torch.meshgrid(torch.linspace(-0.1, 1.1, h), torch.linspace(0, 1, w), indexing="ij")
    pred = (0.7 * (yy * xx)[None, None] + 0.3 * torch.rand(2, c, h, w, generator=g)).float()  # leaves [0, 1] so that the clip matters
    ref = iu.resize_to(pred, (ho, wo), mode).numpy().clip(0, 1)
    out, col, q = ge.postprocess(pred.cuda(), (ho, wo), mode, cmap="Spectral" if c == 1 else None, q_bits=16)
    out_np = out.cpu().numpy()
    err = float(np.abs(out_np - ref).max())
    metric_log(f"postprocess{case}{mode}", max_abs=err)
    assert out_np.shape == ref.shape and out_np.min() >= 0.0 and out_np.max() <= 1.0
    assert err <= {"bilinear": 5e-7, "bicubic": 2e-6, "nearest-exact": 0.0}[mode]
    # quantisation and colour map are exact functions of the fp32 map the device produced
    assert np.array_equal(q.cpu().numpy(), (out_np * 65535.0).astype(np.uint16))
    if c == 1:
        for i in range(out_np.shape[0]):
            want = iu.chw2hwc((iu.colorize_depth_maps(out_np[i, 0], 0, 1, cmap="Spectral").squeeze() * 255).astype(np.uint8))
            assert np.array_equal(col[i].cpu().numpy(), want)
    else:
        assert col is None
    q8 = ge.postprocess(pred.cuda(), (ho, wo), mode, q_bits=8)[2].cpu().numpy()
    assert np.array_equal(q8, (out_np * 255.0).astype(np.uint8))


def test_pipeline_device_prepost_matches_host_path(monkeypatch, metric_log):
    """GenPerceptPipeline.__call__ end to end: the device pre / post path against the host path (GENPERCEPT_HOST_PREPOST=1) on the same engine."""
    import os
    from PIL import Image
    from genpercept_amd import GenPerceptPipeline
    from oracle import sd21 as osd
    uc, vc = osd.UNetCfg.tiny(), osd.VAECfg.tiny()
    g = torch.Generator().manual_seed(9)
    pipe = GenPerceptPipeline(unet=osd.synth_state_dict(osd.unet_manifest(uc), 1), vae=osd.synth_state_dict(osd.vae_manifest(vc), 2),
                              scheduler=dict(beta_start=1.0, beta_end=1.0, prediction_type="v_prediction", clip_sample=False), text_encoder=torch.randn(2, 64, generator=g), tokenizer=None)
    pipe.to("cuda")
    arr = torch.randint(0, 256, (150, 200, 3), generator=g, dtype=torch.uint8).numpy()
    arr[:, :100] = np.linspace(0, 255, 100, dtype=np.uint8)[None, :, None]
    img = Image.fromarray(arr)
    img_f = torch.from_numpy(arr.copy()).permute(2, 0, 1)[None].float() * 0.5 + 63.75  # a float tensor image: resized in fp32, no rounding
    for kw in (dict(processing_res=0, mode="depth"), dict(processing_res=96, mode="depth"), dict(processing_res=96, mode="normal", color_map=None),
               dict(processing_res=64, mode="depth", match_input_res=False, resample_method="nearest"),
               dict(processing_res=96, mode="depth", resample_method="bicubic"), dict(processing_res=96, mode="depth", float_input=True),
               dict(processing_res=0, mode="normal", color_map=None, float_input=True),
               dict(processing_res=80, mode="depth", resample_method="bicubic", float_input=True)):
        kw = dict(kw)
        inp = img_f if kw.pop("float_input", False) else img
        dev = pipe(inp, **kw)
        monkeypatch.setenv("GENPERCEPT_HOST_PREPOST", "1")
        host = pipe(inp, **kw)
        monkeypatch.delenv("GENPERCEPT_HOST_PREPOST")
        assert dev.pred_np.shape == host.pred_np.shape and dev.pred_np.dtype == host.pred_np.dtype == np.float32
        assert dev.pred_colored.size == host.pred_colored.size and dev.pred_colored.mode == host.pred_colored.mode
        dm = float(np.abs(dev.pred_np - host.pred_np).mean())
        dc = float((np.abs(np.asarray(dev.pred_colored).astype(int) - np.asarray(host.pred_colored).astype(int)) > 2).mean())
        metric_log(f"pipeline_prepost{sorted(kw.items())}{'float' if inp is img_f else 'u8'}", mean_abs=dm, frac_colored_diff=dc)
        # without a resize of the INPUT the two paths feed the engine the same bytes; with one, a uint8 pixel may differ by 1 LSB on <0.1 %
        assert dm <= (1e-6 if kw["processing_res"] == 0 else 2e-3) and dc <= (1e-4 if kw["processing_res"] == 0 else 2e-2)
