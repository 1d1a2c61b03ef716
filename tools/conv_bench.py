#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv/linear kernel on the shapes of the 768x768 path (through the C-ABI).
Usage: python tools/conv_bench.py [--iters 10] [--shapes vae128,vae256,...] [--tiles 0,1,4]
Prints one line per (shape, tile config): time, TFLOP/s."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpercept_amd import engine as e  # noqa: E402

SHAPES = {
    # name: (B, H, W, Cin, Cout, ks)
    "vae128": (4, 768, 768, 128, 128, 3),
    "vae256": (4, 384, 384, 256, 256, 3),
    "vae512": (4, 192, 192, 512, 512, 3),
    "vae512_96": (4, 96, 96, 512, 512, 3),
    "unet320": (4, 96, 96, 320, 320, 3),
    "unet640": (4, 48, 48, 640, 640, 3),
    "unet1280_24": (4, 24, 24, 1280, 1280, 3),
    "unet1280_12": (4, 12, 12, 1280, 1280, 3),
    "unet2560_12": (4, 12, 12, 2560, 1280, 3),
    "lin320": (4, 96, 96, 320, 320, 1),
    "ff320": (4, 96, 96, 320, 2560, 1),
    "ff1280": (4, 96, 96, 1280, 320, 1),
    "lin1280_24": (4, 24, 24, 1280, 1280, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--dbg", default="0", help="comma list of GENPERCEPT_IGEMM_DBG ablation values")
    ap.add_argument("--gn", type=int, default=0, help="1: conv3x3(SiLU(GroupNorm(x))) with the apply fused into the halo conv (gp_conv2d_gn)")
    args = ap.parse_args()
    d = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    for name in args.shapes.split(","):
        b, h, w, cin, cout, ks = SHAPES[name]
        x = torch.randn(b, h, w, cin, generator=g).to(torch.bfloat16).to(d)
        wt = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
        wp = e.pack_weight(wt, device=d)
        bias = torch.randn(cout, generator=g).to(d)
        gamma, beta = torch.ones(cin, device=d), torch.zeros(cin, device=d)
        flops = 2.0 * b * h * w * cout * cin * ks * ks
        variants = [(int(t), dv) for t in args.tiles.split(",") for dv in args.dbg.split(",")]
        times = {v: [] for v in variants}
        for rnd in range(args.rounds + 1):  # round 0 = warm-up; variants interleaved inside every round (DVFS drifts between runs)
            for tile, dbg in variants:
                os.environ["GENPERCEPT_IGEMM_DBG"] = dbg
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(args.iters):
                    if args.gn:
                        y = e.conv2d_gn(x, wp, bias, cout, gamma, beta, 32, 1e-6, True)
                    else:
                        y = e.conv2d(x, wp, bias, cout, ks, tile=tile)
                en.record()
                torch.cuda.synchronize()
                if rnd:
                    times[(tile, dbg)].append(st.elapsed_time(en) / args.iters)
        for (tile, dbg), ts in times.items():
            ts.sort()
            med, mn = ts[len(ts) // 2], ts[0]
            print(f"{name:14s} tile={tile} dbg={dbg:>3s} M={b*h*w:8d} N={cout:5d} K={cin*ks*ks:6d}  median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  "
                  f"{flops/med/1e9:7.1f} TFLOP/s", flush=True)
        del x, wp, y


if __name__ == "__main__":
    main()
