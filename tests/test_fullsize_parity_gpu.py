"""Full-size parity against the LIVE fp32 oracle (oracle/), full SD2.1 widths, the reference's default processing resolution
(genpercept_pipeline.py:106 `default_processing_resolution = 768`; call shape run.py:420-432): BASELINE.json configs[1] (depth), configs[2]
(normal), configs[3] (DPT disparity head) at 768x768 with all three engine precisions -- bf16 library, fp16 library, and the contract precision "fp32c"
(fp32 storage + split-bf16 matrix products; gp_set_precision / torch_dtype=float32) --, configs[0]'s 384x384 image through the pipeline surface (fp32 = the
contract precision), and configs[4]'s rank-local shard (8 images per GPU).

Tolerance (north_star: "within 1e-3 rel of the reference"), measured under BOTH readings for every precision:
  mean_abs = mean |HIP - oracle| on the [0,1] map
  rel_rms  = rms(HIP - oracle) / rms(oracle - mean(oracle)) -- the deviation relative to the map's own signal.
  fp32c  inside 1e-3 under both (measured 4.6e-6 / 5.5e-5): test_contract_1e3 asserts it, the regression gates sit at ~2x the measured values.
  fp16   inside 1e-3 under mean_abs only (gated AT 1e-3); rel_rms 5e-3: no engine with single 16-bit MFMA operands reaches 1e-3 there (simulated floor with
         fp16 operands and everything else fp32: 3.4e-3 at 128 px) -- strict xfails in test_contract_1e3, regression gates at 1.25x measured.
  bf16   BASELINE.json's dtype, the benched default: outside the tolerance under both readings (3.35e-3 / 3.96e-2), bench.py says so in its own line;
         strict xfails in test_contract_1e3, regression gates at 1.25x measured (r6; shown to fail when one kernel's rounding doubles).

The oracle runs once per module (about 12 s of CPU at 768x768 on the GPU box's host cores: encoder, ONE UNet pass that returns the sample
and the multi-level features, the 3-channel decode, the DPT head)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# mean |delta| on [0,1] maps vs the fp32 oracle.  fp16: the contract.  bf16: 1.5x the simulated floor of bf16 MFMA operands
# (engine-bf16 row of profiles/r02_precision_ablation.json), i.e. a regression gate, not a parity claim.
# fp32c = the contract precision (gp_set_precision(GP_PREC_CONTRACT): fp32 storage, split-bf16 matrix products; what torch_dtype=float32 selects)
# r6: every bf16 / fp16 regression gate is 1.25x what this build measures on MI355X (gpurun_out/parity_log.jsonl of
# tools/sessions/gpu_r06_s1.sh: bf16 depth / normal / disparity mean_abs 3.35e-3 / 5.47e-3 / 3.05e-3, rel_rms 3.96e-2 / 3.48e-2 / 3.56e-2; fp16
# 4.39e-4 / 7.24e-4 / 4.91e-4 and 5.16e-3 / 4.57e-3 / 5.60e-3; AbsRel after alignment 1.18e-2 / 1.25e-3), so that ONE kernel whose rounding doubles
# fails them (profiles/r06_gate_sensitivity.json: the halo conv storing one mantissa bit less moves bf16 depth to 5.1e-3).  The fp16 mean_abs gates
# stay AT the contract (1e-3; 2e-3 for the min-max-rescaled DPT map).  fp32c: ~2x its measured 4.6e-6 / 7.7e-6 / 7.1e-6 and 5.5e-5 / 4.9e-5 / 7.9e-5.
MAP_TOL = {"fp16": {"depth": 1e-3, "normal": 1e-3, "disparity": 2e-3}, "bf16": {"depth": 4.2e-3, "normal": 6.9e-3, "disparity": 3.9e-3},
           "fp32c": {"depth": 1.5e-5, "normal": 1.5e-5, "disparity": 1.5e-5}}
ABSREL_TOL = {"fp16": 1.6e-3, "bf16": 1.5e-2, "fp32c": 1.2e-4}
# two runs of one build on other persistent grids (batch 8 vs batch 4: other summation order of the GroupNorm statistics) differ like two roundings
B8_VS_B4_TOL = {"fp16": 6.2e-4, "bf16": 4.9e-3, "fp32c": 1.5e-5}
# rel-RMS of the same maps: regression gates (~1.5x the simulated engine rows of profiles/r04_precision_ablation.json), NOT the contract
RELRMS_TOL = {"fp16": {"depth": 6.5e-3, "normal": 5.8e-3, "disparity": 7.0e-3}, "bf16": {"depth": 5.0e-2, "normal": 4.4e-2, "disparity": 4.5e-2},
              "fp32c": {"depth": 1e-4, "normal": 1e-4, "disparity": 1e-4}}
PRECISIONS = ["fp16", "bf16", "fp32c"]
# what the regression-gated tests above measured, keyed (precision, head): read by test_contract_1e3 below (same maps, no second engine run)
_MEASURED = {}
# north_star's tolerance itself -- "outputs within 1e-3 rel of the reference" -- per library, head and reading.  Where the build is KNOWN to
# be outside it the case is xfail(strict=True): the state of the contract is then in the GPU test record (xfailed = known miss, failed = a
# regression of a case that held, XPASS(strict) = a miss that was fixed and must be promoted), not in prose.  Why the
# misses cannot be bought back cheaply: profiles/r05_precision_attribution.json (the error is spread evenly over ~84 units; protecting the
# best 10 % of the FLOPs with split operands moves rel_rms from 5.0e-3 to 3.4e-3).
CONTRACT = 1e-3
CONTRACT_KNOWN_MISS = {("fp16", "depth", "rel_rms"), ("fp16", "normal", "rel_rms"), ("fp16", "disparity", "rel_rms"),
                       ("bf16", "depth", "mean_abs"), ("bf16", "normal", "mean_abs"), ("bf16", "disparity", "mean_abs"),
                       ("bf16", "depth", "rel_rms"), ("bf16", "normal", "rel_rms"), ("bf16", "disparity", "rel_rms")}


def _rel_rms(out, ref):
    out, ref = out.astype(np.float64), ref.astype(np.float64)
    return float(np.sqrt(((out - ref) ** 2).mean()) / (np.sqrt(((ref - ref.mean()) ** 2).mean()) + 1e-30))


def _bench_rgb(batch, res, seed=1234):
    """bench.py's synthetic input (uint8 noise blended 50/50 with a smooth field)."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randint(0, 256, (batch, 3, res, res), generator=g, dtype=torch.uint8).float()
    yy, xx = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    smooth = torch.stack([yy, xx, (yy + xx) / 2])[None] * 255.0
    return (0.5 * noise + 0.5 * smooth).round().clamp(0, 255).to(torch.uint8)


@pytest.fixture(scope="module")
def full():
    """Full-width synthetic weights + the oracle's maps of image 0 of the benched batch at 768x768 and of a 384x384 image."""
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from oracle import dpt as odpt
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    torch.set_num_threads(max(1, min(os.cpu_count() or 8, 32)))
    ucfg, vcfg, dcfg = gc.UNetConfig(), gc.VAEConfig(), gc.DPTConfig()
    usd = gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0)
    vsd = gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1)
    dsd = gw.synth_state_dict(gw.dpt_manifest(dcfg), seed=3)
    ctx = torch.randn(2, 1024, generator=torch.Generator().manual_seed(2))
    rgb8 = _bench_rgb(8, 768)  # images 0..3 are bench.py's batch
    ref = {}
    with torch.no_grad():
        x = opipe.normalize_rgb(rgb8[:1])
        lat = osd.encode_rgb(vsd, osd.VAECfg(), x)
        v, feats = osd.unet_forward(usd, osd.UNetCfg(), lat, 1, ctx.reshape(1, 2, -1))  # (sample, multi_level_feats) of ONE pass
        dec3 = osd.vae_decode(vsd, osd.VAECfg(), (-v) / osd.VAECfg().scaling_factor)  # genpercept_pipeline.py:465,507-526
        ref["normal"] = ((dec3.clip(-1, 1) + 1) / 2)[0].numpy()
        ref["depth"] = ((dec3.mean(dim=1, keepdim=True).clip(-1, 1) + 1) / 2)[0].numpy()
        # the DPT-head UNet has no conv_out: its features are the same tensors the pass above returned (custom_unet.py:365-400)
        pred = odpt.dpt_head_forward(dsd, feats[::-1])[:, None]
        ref["disparity"] = ((pred - pred.min()) / (pred.max() - pred.min()))[0].numpy()
        rgb384 = _bench_rgb(1, 384, seed=77)
        ref["depth384"] = opipe.single_infer(vsd, osd.VAECfg(), usd, osd.UNetCfg(), opipe.normalize_rgb(rgb384), ctx, "depth")[0].numpy()
    return dict(ucfg=ucfg, vcfg=vcfg, dcfg=dcfg, usd=usd, vsd=vsd, dsd=dsd, ctx=ctx, rgb8=rgb8, rgb384=rgb384, ref=ref)


def _engine(full, precision, dpt):
    from genpercept_amd.engine import Engine
    ucfg = dataclasses.replace(full["ucfg"], has_out=False) if dpt else full["ucfg"]
    eng = Engine(0, ucfg, full["vcfg"], full["dcfg"] if dpt else None, precision=precision)
    eng.load_state_dict("vae", full["vsd"])
    eng.load_state_dict("unet", {k: v for k, v in full["usd"].items() if not (dpt and k.startswith(("conv_out", "conv_norm_out")))})
    if dpt:
        eng.load_state_dict("dpt", full["dsd"])
    eng.set_context(full["ctx"])
    eng.finalize()
    return eng


def _absrel_ls(pred, gt):
    from genpercept_amd.eval_metrics import abs_relative_difference, align_depth_least_square
    gt = gt.astype(np.float64).clip(1e-3, None)
    al, _, _ = align_depth_least_square(gt, pred.astype(np.float64), np.ones_like(gt, dtype=bool))
    return float(abs_relative_difference(np.clip(al, 1e-3, None), gt))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_768_depth_and_normal_vs_live_oracle(precision, full, metric_log):
    """configs[1] / configs[2] at the benched size: image 0 alone (B = 1), inside the benched batch of 4 and inside configs[4]'s rank-local
    shard of 8 -- all against the fp32 oracle's map of that image."""
    d = torch.device("cuda", 0)
    eng = _engine(full, precision, dpt=False)
    try:
        for mode in ("depth", "normal"):
            ref = full["ref"][mode]
            out = eng.infer(full["rgb8"][:1].to(d), mode)[0].cpu().numpy()
            err = np.abs(out - ref)
            rec = dict(mean_abs=float(err.mean()), max_abs=float(err.max()), rel_rms=_rel_rms(out, ref), absrel_ls=_absrel_ls(out[0], ref[0]))
            rec["within_1e-3_mean_abs"], rec["within_1e-3_rel_rms"] = rec["mean_abs"] <= 1e-3, rec["rel_rms"] <= 1e-3
            metric_log(f"full768_{mode}_vs_oracle[{precision}]", **rec)
            _MEASURED[(precision, mode)] = (rec["mean_abs"], rec["rel_rms"])
            assert out.shape == ref.shape and np.isfinite(out).all()
            assert rec["mean_abs"] <= MAP_TOL[precision][mode], rec
            assert rec["rel_rms"] <= RELRMS_TOL[precision][mode], rec
            if mode == "depth":
                assert rec["absrel_ls"] <= ABSREL_TOL[precision], rec
        ref = full["ref"]["depth"]
        b4 = eng.infer(full["rgb8"][:4].to(d), "depth")
        b8 = eng.infer(full["rgb8"].to(d), "depth")  # configs[4]: 8 images per GPU
        assert b8.shape == (8, 1, 768, 768) and torch.isfinite(b8).all() and 0.0 <= float(b8.min()) and float(b8.max()) <= 1.0
        assert torch.equal(eng.infer(full["rgb8"].to(d), "depth"), b8), "batch 8 is not deterministic"
        assert torch.equal(eng.infer(full["rgb8"].flip(0).to(d), "depth").flip(0), b8), "batch 8: a result depends on its batch slot"
        for name, o in (("b4", b4[0]), ("b8", b8[0])):
            e = float(np.abs(o.cpu().numpy() - ref).mean())
            metric_log(f"full768_depth_{name}_image0_vs_oracle[{precision}]", mean_abs=e)
            assert e <= MAP_TOL[precision]["depth"], (name, e)
        # images 0..3 of the shard against the batch-4 call (other persistent grids: other summation order of the statistics)
        metric_log(f"full768_b8_vs_b4[{precision}]", mean_abs=float((b8[:4] - b4).abs().mean()), max_abs=float((b8[:4] - b4).abs().max()))
        assert float((b8[:4] - b4).abs().mean()) <= B8_VS_B4_TOL[precision]
    finally:
        eng.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_768_dpt_disparity_vs_live_oracle(precision, full, metric_log):
    """configs[3]: custom UNet features -> DPT neck / head -> per-image min-max (genpercept_pipeline.py:474-482) at 768x768."""
    d = torch.device("cuda", 0)
    eng = _engine(full, precision, dpt=True)
    try:
        ref = full["ref"]["disparity"]
        out = eng.infer(full["rgb8"][:1].to(d), "disparity")[0].cpu().numpy()
        err = np.abs(out - ref)
        rr = _rel_rms(out, ref)
        metric_log(f"full768_disparity_dpt_vs_oracle[{precision}]", mean_abs=float(err.mean()), max_abs=float(err.max()), rel_rms=rr)
        _MEASURED[(precision, "disparity")] = (float(err.mean()), rr)
        assert out.shape == ref.shape and np.isfinite(out).all() and abs(float(out.min())) < 1e-6 and abs(float(out.max()) - 1) < 1e-6
        assert float(err.mean()) <= MAP_TOL[precision]["disparity"], float(err.mean())
        assert rr <= RELRMS_TOL[precision]["disparity"], rr
    finally:
        eng.close()


def _contract_cases():
    for prec in PRECISIONS:
        for head in ("depth", "normal", "disparity"):
            for metric in ("mean_abs", "rel_rms"):
                marks = [pytest.mark.xfail(strict=True, reason=f"known: the {prec} library is outside 1e-3 under {metric} on the {head} map "
                                                              "(DESIGN.md section 4, profiles/r05_precision_attribution.json)")] \
                    if (prec, head, metric) in CONTRACT_KNOWN_MISS else []
                yield pytest.param(prec, head, metric, marks=marks, id=f"{prec}-{head}-{metric}")


@pytest.mark.parametrize("precision,head,metric", list(_contract_cases()))
def test_contract_1e3(precision, head, metric, full, metric_log):
    """north_star: "depth/normal outputs within 1e-3 rel of the reference" at 768x768, image 0 of the benched batch, vs the live fp32 oracle --
    asserted AT 1e-3 for every (library, head, reading); known misses are strict xfails (see CONTRACT_KNOWN_MISS)."""
    if (precision, head) not in _MEASURED:  # (run alone, e.g. with -k: measure here)
        d = torch.device("cuda", 0)
        eng = _engine(full, precision, dpt=head == "disparity")
        try:
            out = eng.infer(full["rgb8"][:1].to(d), head)[0].cpu().numpy()
        finally:
            eng.close()
        ref = full["ref"][head]
        _MEASURED[(precision, head)] = (float(np.abs(out - ref).mean()), _rel_rms(out, ref))
    value = _MEASURED[(precision, head)][0 if metric == "mean_abs" else 1]
    metric_log(f"contract_1e-3[{precision}-{head}-{metric}]", value=value, within=value <= CONTRACT)
    assert value <= CONTRACT, f"{precision} {head} {metric} = {value:.3e} > 1e-3"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_nyu_480x640_depth_vs_live_oracle(precision, full, metric_log):
    """The NYU evaluation resolution (north_star: "AbsRel unchanged on NYU eval split"; config/dataset/eval/data_nyu_test.yaml, infer.py:408-447 with
    processing_res = 0): 480 x 640 -> latent 60 x 80, neither a multiple of the 16- nor of the 12-row tiles, full SD2.1 widths, batch 2 (ragged
    tiles, several tiles per workgroup, the 12-row variant on the 60 x 80 x 512 maps).  Image 0 against the live fp32 oracle: mean |delta| at the
    library's map gate, and AbsRel after the reference's least-squares alignment (the evaluation protocol's metric, src/util/metric.py)."""
    from oracle import pipeline as opipe
    from oracle import sd21 as osd
    g = torch.Generator().manual_seed(480640)
    noise = torch.randint(0, 256, (2, 3, 480, 640), generator=g, dtype=torch.uint8).float()
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 480), torch.linspace(0, 1, 640), indexing="ij")
    rgb8 = (0.5 * noise + 0.5 * torch.stack([yy, xx, (yy + xx) / 2])[None] * 255.0).round().clamp(0, 255).to(torch.uint8)
    if "depth_nyu" not in full["ref"]:
        with torch.no_grad():
            full["ref"]["depth_nyu"] = opipe.single_infer(full["vsd"], osd.VAECfg(), full["usd"], osd.UNetCfg(), opipe.normalize_rgb(rgb8[:1]), full["ctx"],
                                                          "depth")[0].numpy()
    ref = full["ref"]["depth_nyu"]
    d = torch.device("cuda", 0)
    eng = _engine(full, precision, dpt=False)
    try:
        out = eng.infer(rgb8.to(d), "depth")
        assert out.shape == (2, 1, 480, 640) and torch.isfinite(out).all()
        assert torch.equal(eng.infer(rgb8.to(d), "depth"), out), "not deterministic"
        o0 = out[0].cpu().numpy()
        rec = dict(mean_abs=float(np.abs(o0 - ref).mean()), max_abs=float(np.abs(o0 - ref).max()), rel_rms=_rel_rms(o0, ref), absrel_ls=_absrel_ls(o0[0], ref[0]))
        metric_log(f"nyu480x640_depth_vs_oracle[{precision}]", **rec)
        assert rec["mean_abs"] <= MAP_TOL[precision]["depth"], rec
        assert rec["absrel_ls"] <= ABSREL_TOL[precision], rec
    finally:
        eng.close()


def test_384_image_through_the_pipeline_vs_live_oracle(full, metric_log):
    """configs[0]'s shape (one 384x384 RGB image, depth, the reference's fp32 run) through GenPerceptPipeline.__call__ on the HIP path:
    torch_dtype=float32 (the reference's default, run.py:273-281) selects the contract precision."""
    from PIL import Image
    from genpercept_amd import GenPerceptPipeline
    pipe = GenPerceptPipeline(unet=full["usd"], vae=full["vsd"], scheduler=dict(beta_start=1.0, beta_end=1.0, beta_schedule="linear",
                              prediction_type="v_prediction", clip_sample=False, steps_offset=1), text_encoder=full["ctx"].numpy(), tokenizer=None,
                              torch_dtype=torch.float32)
    img = Image.fromarray(full["rgb384"][0].permute(1, 2, 0).numpy())
    try:
        out = pipe(img, denoising_steps=1, ensemble_size=1, processing_res=384, match_input_res=True, batch_size=1, color_map="Spectral",
                   show_progress_bar=False, mode="depth")
        ref = full["ref"]["depth384"][0]
        err = np.abs(out.pred_np - ref)
        metric_log("full384_pipeline_depth_vs_oracle[fp32c]", mean_abs=float(err.mean()), max_abs=float(err.max()), rel_rms=_rel_rms(out.pred_np, ref),
                   absrel_ls=_absrel_ls(out.pred_np, ref))
        assert out.pred_np.shape == (384, 384) and out.pred_colored.size == (384, 384)
        assert pipe._engine.precision == "fp32c" and pipe.dtype == torch.float32
        assert float(err.mean()) <= 1e-4 and _rel_rms(out.pred_np, ref) <= 1e-3, (float(err.mean()), _rel_rms(out.pred_np, ref))
    finally:
        if pipe._engine is not None:
            pipe._engine.close()
