"""Range stress of the fp16 library.  fp16 storage overflows at 65504; the reference's own `--half_precision`
(run.py:273-281) then produces inf / NaN.  This engine's conversions saturate instead -- which must never be SILENT: every call in which a
conversion actually clipped is counted (`gp_saturation_events`; the pipeline logs a warning).

Full SD2.1 widths, 64x64 input.  Weights are the benign variance-preserving synthetic ones with ONE tensor scaled so that a chosen class of
activations leaves the fp16 range in the fp32 oracle (asserted on the oracle's own tensors):
    trunk   unet.conv_in x K          the residual trunk / skip stack and every GroupNorm INPUT of the first down block
    vae     vae.encoder.conv_in x K   the encoder's trunk at full resolution (rgb_conv_in_kernel's epilogue)
    geglu   ff.net.0.proj x K         hidden * gelu(gate) of one feed-forward
For each: the engine must REPORT saturation (events > 0) unless its map is still inside the fp16 gate; with the unscaled weights it must
report none and stay inside the gate."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GATE = 1e-3  # mean |delta| on the [0,1] map, the fp16 library's contract (tests/test_fullsize_parity_gpu.py)


@pytest.fixture(scope="module")
def setup():
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    torch.set_num_threads(max(1, min(os.cpu_count() or 8, 32)))
    ucfg, vcfg = gc.UNetConfig(), gc.VAEConfig()
    usd = gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0)
    vsd = gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1)
    ctx = torch.randn(2, 1024, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(5)
    rgb = torch.randint(0, 256, (1, 3, 64, 64), generator=g, dtype=torch.uint8)
    return dict(ucfg=ucfg, vcfg=vcfg, usd=usd, vsd=vsd, ctx=ctx, rgb=rgb)


def _scaled(sd, prefix, k):
    out = dict(sd)
    for suf in (".weight", ".bias"):
        out[prefix + suf] = sd[prefix + suf] * k
    return out


def _oracle(usd, vsd, ctx, rgb, probe):
    """fp32 oracle map + the max |activation| at the probed site"""
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    seen = {"max": 0.0}
    real_conv, real_lin = osd._conv, osd._linear

    def conv(x, sd, p, stride=1, padding=1):
        y = real_conv(x, sd, p, stride, padding)
        if probe == ("conv", p) and sd is probe_sd[0]:
            seen["max"] = max(seen["max"], float(y.abs().max()))
        return y

    def lin(x, sd, p):
        y = real_lin(x, sd, p)
        if probe[0] == "geglu" and p == probe[1]:
            h, gate = y.chunk(2, dim=-1)
            seen["max"] = max(seen["max"], float((h * F.gelu(gate)).abs().max()))
        return y

    probe_sd = [usd if probe[1].startswith(("conv_in", "down_blocks", "mid", "up_blocks")) else vsd]
    osd._conv, osd._linear = conv, lin
    try:
        with torch.no_grad():
            ref = opipe.single_infer(vsd, osd.VAECfg(), usd, osd.UNetCfg(), opipe.normalize_rgb(rgb), ctx, "depth")
    finally:
        osd._conv, osd._linear = real_conv, real_lin
    return ref[0].numpy(), seen["max"]


def _engine(setup, usd, vsd):
    from genpercept_amd.engine import Engine
    eng = Engine(0, setup["ucfg"], setup["vcfg"], None, precision="fp16")
    eng.load_state_dict("vae", vsd)
    eng.load_state_dict("unet", usd)
    eng.set_context(setup["ctx"])
    eng.finalize()
    return eng


CASES = {
    "benign": (None, None, 1.0),
    "trunk": ("unet", "conv_in", 3.0e5),
    "vae": ("vae", "encoder.conv_in", 6.0e4),
    "geglu": ("unet", "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj", 1.5e2),
}


@pytest.mark.parametrize("case", list(CASES))
def test_fp16_saturation_is_never_silent(case, setup, metric_log):
    which, key, k = CASES[case]
    usd = _scaled(setup["usd"], key, k) if which == "unet" else setup["usd"]
    vsd = _scaled(setup["vsd"], key, k) if which == "vae" else setup["vsd"]
    probe = ("geglu", key) if case == "geglu" else ("conv", key or "conv_in")
    ref, site_max = _oracle(usd, vsd, setup["ctx"], setup["rgb"], probe)
    d = torch.device("cuda", 0)
    eng = _engine(setup, usd, vsd)
    try:
        eng.infer(setup["rgb"].to(d), "depth")
        eng.saturation_events(reset=True)  # (collects and discards flags left behind by per-kernel test entry points earlier in this process)
        out = eng.infer(setup["rgb"].to(d), "depth")[0].cpu().numpy()
        events = eng.saturation_events()
        assert eng.saturation_events(reset=True) == events and eng.saturation_events() == 0
    finally:
        eng.close()
    err = float(np.abs(out - ref).mean())
    metric_log(f"fp16_range[{case}]", oracle_site_max=site_max, events=events, mean_abs=err)
    assert np.isfinite(out).all() and 0.0 <= out.min() and out.max() <= 1.0  # saturating conversions: never inf / NaN in the map
    if case == "benign":
        assert site_max < 65504.0 and events == 0 and err <= GATE, (site_max, events, err)
    else:
        assert site_max > 65504.0, f"{case}: the probed activation stays inside the fp16 range in the oracle ({site_max:.3g}): raise the gain"
        assert events > 0 or err <= GATE, f"{case}: SILENT clipping -- oracle max {site_max:.3g}, no saturation reported, mean |delta| {err:.3g}"
        assert events > 0, f"{case}: an activation of {site_max:.3g} was stored without the engine noticing"


def test_bf16_library_reports_no_saturation(setup):
    """bf16 elements have the fp32 range: the same out-of-fp16-range trunk passes without clipping and the counter stays 0"""
    from genpercept_amd.engine import Engine
    usd = _scaled(setup["usd"], "conv_in", 3.0e5)
    eng = Engine(0, setup["ucfg"], setup["vcfg"], None, precision="bf16")
    try:
        eng.load_state_dict("vae", setup["vsd"])
        eng.load_state_dict("unet", usd)
        eng.set_context(setup["ctx"])
        eng.finalize()
        out = eng.infer(setup["rgb"].to(torch.device("cuda", 0)), "depth")
        assert torch.isfinite(out).all() and eng.saturation_events() == 0
    finally:
        eng.close()
