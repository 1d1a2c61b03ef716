#!/usr/bin/env python3
"""run.py-equivalent wall time per image, host pre/post (torch-CPU resize + matplotlib colour map, the reference's recipe) vs the device
path (gp_preprocess / gp_postprocess): full SD2.1 widths, one 3024x4032 photo-sized input, processing_res 768, match_input_res, Spectral.
usage: python tools/prepost_bench.py [--iters 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--h", type=int, default=3024)
    ap.add_argument("--w", type=int, default=4032)
    args = ap.parse_args()
    from genpercept_amd import GenPerceptPipeline
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    ucfg, vcfg = gc.UNetConfig(), gc.VAEConfig()
    pipe = GenPerceptPipeline(unet=gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0), vae=gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1),
                              scheduler=dict(beta_start=1.0, beta_end=1.0, prediction_type="v_prediction"),
                              text_encoder=torch.randn(2, 1024, generator=torch.Generator().manual_seed(2)), tokenizer=None)
    pipe.to("cuda")
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (args.h, args.w, 3), dtype=np.uint8))
    res = {}
    for name, env in (("device", None), ("host", "1")):
        if env:
            os.environ["GENPERCEPT_HOST_PREPOST"] = env
        else:
            os.environ.pop("GENPERCEPT_HOST_PREPOST", None)
        pipe(img, processing_res=768, mode="depth")  # warm-up (engine build on the first call)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = pipe(img, processing_res=768, mode="depth", color_map="Spectral")
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / args.iters * 1e3
        assert out.pred_np.shape == (args.h, args.w)
    os.environ.pop("GENPERCEPT_HOST_PREPOST", None)
    print(json.dumps({"input": f"{args.h}x{args.w} uint8 PIL image", "processing_res": 768, "ms_per_image_host_prepost": round(res["host"], 1),
                      "ms_per_image_device_prepost": round(res["device"], 1), "speedup": round(res["host"] / res["device"], 2)}))


if __name__ == "__main__":
    main()
