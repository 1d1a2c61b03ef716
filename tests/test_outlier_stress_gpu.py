"""Trained-weight-like stress: every parity number elsewhere is measured on benign random-init weights, whereas a trained SD2.1
checkpoint has a few OUTLIER channels per normalisation layer (activations tens to hundreds of times the layer's typical magnitude).  Here the
full-width synthetic weights get that pattern -- in every GroupNorm of the VAE and the UNet (resnet norm1 / norm2, the transformers' and the VAE
attention's group norms; not the two final conv_norm_out, which would only rescale the output map) two channels have their gain multiplied by
30 ... 100 -- and the whole one-step path (genpercept_pipeline.py:375-486) runs through all three engine precisions against the live fp32 oracle
on the same weights:

  fp32c (contract precision)  inside north_star's 1e-3 under both readings -- outliers are exactly what a 16-bit engine loses and the split-operand
                              products keep;
  bf16                        fp32 range, so no clipping: gated at 1.25x its measured deviation on this fixture;
  fp16                        either inside its gate, or the engine REPORTED the clip (gp_saturation_events > 0): never a silent deviation.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RES = 256
# measured on MI355X (gpurun_out/parity_log.jsonl: outlier_stress[...]; tools/sessions/gpu_r06_s1.sh) x 1.25, per head:
#   bf16  depth 9.5e-3 / 1.06e-1, normal 1.68e-2 / 1.13e-1   (benign weights at this size: 3.4e-3 / 4.0e-2 -- outliers cost bf16 a factor 2.7)
#   fp16  depth 1.03e-3 / 1.15e-2, normal 1.83e-3 / 1.22e-2  (benign: 4.4e-4 / 5.2e-3: the fp16 build LEAVES 1e-3 mean_abs on such weights, with
#         zero saturation events -- nothing clipped, the 11-bit operands are simply too coarse; this is the case the contract precision exists for)
#   fp32c depth 1.2e-5 / 1.4e-4, normal 2.2e-5 / 1.5e-4
GATE = {"bf16": {"depth": dict(mean_abs=1.19e-2, rel_rms=1.33e-1), "normal": dict(mean_abs=2.1e-2, rel_rms=1.42e-1)},
        "fp16": {"depth": dict(mean_abs=1.3e-3, rel_rms=1.45e-2), "normal": dict(mean_abs=2.3e-3, rel_rms=1.52e-2)}}


def outlierize(sd, seed, per_norm=2, lo=30.0, hi=100.0):
    """multiply the gain of `per_norm` random channels of every GroupNorm (not LayerNorm, not the final conv_norm_out) by a factor in [lo, hi]"""
    g = torch.Generator().manual_seed(seed)
    out, n = {}, 0
    for k, v in sd.items():
        v = v.clone()
        gn = v.dim() == 1 and k.endswith(".weight") and "transformer_blocks" not in k and "conv_norm_out" not in k and \
            (".norm1." in k or ".norm2." in k or k.endswith("group_norm.weight") or k.endswith(".norm.weight"))
        if gn:
            idx = torch.randperm(v.numel(), generator=g)[:per_norm]
            v[idx] *= lo + (hi - lo) * torch.rand(per_norm, generator=g)
            n += 1
        out[k] = v
    return out, n


def _rel_rms(out, ref):
    out, ref = out.astype(np.float64), ref.astype(np.float64)
    return float(np.sqrt(((out - ref) ** 2).mean()) / (np.sqrt(((ref - ref.mean()) ** 2).mean()) + 1e-30))


@pytest.fixture(scope="module")
def stress():
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    torch.set_num_threads(max(1, min(os.cpu_count() or 8, 32)))
    ucfg, vcfg = gc.UNetConfig(), gc.VAEConfig()
    usd, nu = outlierize(gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0), 11)
    vsd, nv = outlierize(gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1), 12)
    assert nu == 60 and nv == 50  # 44 resnet norms + 16 transformer norms; 2 x (22 + 1 attention norm) + ... of the VAE
    ctx = torch.randn(2, 1024, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(5)
    noise = torch.randint(0, 256, (2, 3, RES, RES), generator=g, dtype=torch.uint8).float()
    yy, xx = torch.meshgrid(torch.linspace(0, 1, RES), torch.linspace(0, 1, RES), indexing="ij")
    rgb8 = (0.5 * noise + 0.5 * torch.stack([yy, xx, (yy + xx) / 2])[None] * 255.0).round().clamp(0, 255).to(torch.uint8)
    with torch.no_grad():
        ref = {m: opipe.single_infer(vsd, osd.VAECfg(), usd, osd.UNetCfg(), opipe.normalize_rgb(rgb8[:1]), ctx, m)[0].numpy() for m in ("depth", "normal")}
    clipped = float(((ref["depth"] == 0) | (ref["depth"] == 1)).mean())
    assert ref["depth"].std() > 0.05 and clipped < 0.2, "the stress fixture must leave a non-degenerate map"
    return dict(ucfg=ucfg, vcfg=vcfg, usd=usd, vsd=vsd, ctx=ctx, rgb8=rgb8, ref=ref)


@pytest.mark.parametrize("precision", ["fp32c", "bf16", "fp16"])
def test_outlier_channels_vs_live_oracle(precision, stress, metric_log):
    from genpercept_amd.engine import Engine
    d = torch.device("cuda", 0)
    eng = Engine(0, stress["ucfg"], stress["vcfg"], None, precision=precision)
    try:
        eng.load_state_dict("vae", stress["vsd"])
        eng.load_state_dict("unet", stress["usd"])
        eng.set_context(stress["ctx"])
        eng.finalize()
        eng.saturation_events(reset=True)
        rec = {}
        for mode in ("depth", "normal"):
            out = eng.infer(stress["rgb8"].to(d), mode)
            assert torch.isfinite(out).all()
            o0, ref = out[0].cpu().numpy(), stress["ref"][mode]
            rec[mode] = dict(mean_abs=float(np.abs(o0 - ref).mean()), max_abs=float(np.abs(o0 - ref).max()), rel_rms=_rel_rms(o0, ref))
        events = eng.saturation_events(reset=True)
        metric_log(f"outlier_stress[{precision}]", saturation_events=events, **{f"{m}_{k}": v for m, r in rec.items() for k, v in r.items()})
        for mode, r in rec.items():
            if precision == "fp32c":
                assert r["mean_abs"] <= 1e-3 and r["rel_rms"] <= 1e-3, (mode, r)          # the contract itself
                assert r["mean_abs"] <= 3e-5 and r["rel_rms"] <= 1.9e-4, (mode, r)        # and 1.25x what it measures here
            elif precision == "bf16":
                assert events == 0
                assert r["mean_abs"] <= GATE["bf16"][mode]["mean_abs"] and r["rel_rms"] <= GATE["bf16"][mode]["rel_rms"], (mode, r)
            else:  # fp16: a deviation beyond the gate is only acceptable when the engine said it clipped
                inside = r["mean_abs"] <= GATE["fp16"][mode]["mean_abs"] and r["rel_rms"] <= GATE["fp16"][mode]["rel_rms"]
                assert inside or events > 0, (mode, r, events)
    finally:
        eng.close()
