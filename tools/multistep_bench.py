#!/usr/bin/env python3
"""Throughput of the multi-step archs on one MI355X (not the headline metric: bench.py measures the one-step path BASELINE.json names).

    python tools/multistep_bench.py [--archs marigold|rgb_blending] [--denoise-steps 10] [--batch 4] [--res 768] [--precision bf16]

One JSON line: images/s, ms per batch, ms per denoising step (encode + n x UNet + decode on synthetic weights / images)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--archs", default="marigold", choices=["marigold", "rgb_blending"])
    ap.add_argument("--denoise-steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=768)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32c"])
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from genpercept_amd.engine import Engine
    from genpercept_amd.scheduler import DDIMSchedulerCustomized
    marigold = a.archs == "marigold"
    ucfg, vcfg = gc.UNetConfig(in_channels=8 if marigold else 4), gc.VAEConfig()
    eng = Engine(0, ucfg, vcfg, None, precision=a.precision)
    eng.load_state_dict("vae", gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1))
    eng.load_state_dict("unet", gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0))
    eng.set_context(torch.randn(2, ucfg.cross_attention_dim, generator=torch.Generator().manual_seed(2)))
    eng.finalize()
    sched = DDIMSchedulerCustomized(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                                    steps_offset=1, prediction_type="v_prediction")  # hf_configs/scheduler_beta_0.00085_0.012
    plan = sched.plan(a.denoise_steps)
    d = torch.device("cuda", 0)
    g = torch.Generator(device=d).manual_seed(0)
    rgb = torch.randint(0, 256, (a.batch, 3, a.res, a.res), dtype=torch.uint8, device=d, generator=g)
    noise = torch.randn(a.batch, 4, a.res // 8, a.res // 8, device=d, generator=g) if marigold else None
    t0 = time.perf_counter()
    out = eng.infer_steps(rgb, "depth", plan, noise)  # first pass folds the time embedding of every timestep of the schedule (host)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    eng.infer_steps(rgb, "depth", plan, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = eng.infer_steps(rgb, "depth", plan, noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    assert torch.isfinite(out).all()
    eng.set_profile(1)
    eng.infer_steps(rgb, "depth", plan, noise)
    tm = eng.timings()
    print(json.dumps({"archs": a.archs, "denoise_steps": a.denoise_steps, "batch": a.batch, "res": a.res, "dtype": a.precision,
                      "images_per_s": round(a.batch / dt, 2), "ms_per_batch": round(dt * 1e3, 2),
                      "ms_encode": round(tm["ms_encode"], 2), "ms_loop": round(tm["ms_unet"], 2), "ms_per_step": round(tm["ms_unet"] / a.denoise_steps, 3),
                      "ms_decode": round(tm["ms_head"], 2), "first_call_s": round(first, 2)}))


if __name__ == "__main__":
    main()
