#!/usr/bin/env python3
"""Per-launch cost of one gp_infer pass (profiling level 3: an event before every launch, cost = start-to-next-start, i.e. kernel +
the gap behind it).  Prints the launches in issue order and a summary grouped by kernel kind and by stage; the full list goes to
gpurun_out/launch_log_<tag>.txt.   usage: python tools/launch_log.py [--batch 4] [--res 768] [--head vae|dpt] [--tag r02]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=768)
    ap.add_argument("--head", default="vae")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--precision", default="bf16", help="bf16 | fp16 | fp32c (contract precision)")
    args = ap.parse_args()
    from bench import synthetic_rgb
    from genpercept_amd import config as gc
    from genpercept_amd import weights as gw
    from genpercept_amd.engine import Engine

    dpt = args.head == "dpt"
    ucfg, vcfg = gc.UNetConfig(has_out=not dpt), gc.VAEConfig()
    dcfg = gc.DPTConfig() if dpt else None
    eng = Engine(0, ucfg, vcfg, dcfg, precision=args.precision)
    eng.load_state_dict("vae", gw.synth_state_dict(gw.vae_manifest(vcfg), seed=1))
    eng.load_state_dict("unet", gw.synth_state_dict(gw.unet_manifest(ucfg), seed=0))
    if dpt:
        eng.load_state_dict("dpt", gw.synth_state_dict(gw.dpt_manifest(dcfg), seed=3))
    eng.set_context(torch.randn(2, ucfg.cross_attention_dim, generator=torch.Generator().manual_seed(2)))
    eng.finalize()
    rgb = synthetic_rgb(args.batch, args.res, 1234, "cuda")
    mode = "disparity" if dpt else "depth"
    for _ in range(2):
        eng.infer(rgb, mode)
    torch.cuda.synchronize()
    eng.set_profile(3)
    logs = []
    for _ in range(args.passes):
        eng.infer(rgb, mode)
        torch.cuda.synchronize()
        logs.append(eng.launch_log())
    eng.set_profile(0)
    n = len(logs[0])
    rows = []
    for i in range(n):
        ms = sorted(l[i][0] for l in logs)[len(logs) // 2]
        rows.append((ms, logs[0][i][1], logs[0][i][2]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"launch_log_{args.tag}_{args.head}_b{args.batch}_{args.res}" + ("" if args.precision == "bf16" else "_" + args.precision) + ".txt")
    # stage boundaries by name: encoder ends at the first 'unet' conv_in = the conv with K = 576 after the encoder's conv_out
    tot = sum(r[0] for r in rows)
    with open(path, "w") as f:
        f.write(f"# {n} launches, {tot:.3f} ms (median of {args.passes} passes; cost = start-to-next-start)\n")
        f.write("# idx\tms\tTFLOP/s\tdescription\n")
        for i, (ms, fl, name) in enumerate(rows):
            f.write(f"{i}\t{ms:.4f}\t{(fl / ms / 1e9) if ms > 0 else 0:.1f}\t{name}\n")
    kinds = collections.OrderedDict()
    for ms, fl, name in rows:
        k = " ".join(name.split()[:2]) if name.split()[0] in ("gemm", "bgemm", "conv3x3", "conv3x3up", "conv3x3s2") else name.split()[0]
        if args.precision == "fp32c" and name.split()[0] == "bgemm":
            k += " " + [t for t in name.split() if t.startswith("K=")][0]
        a = kinds.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += fl
    print(f"{n} launches, {tot:.3f} ms -> {path}")
    for k, (c, ms, fl) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} n={c:4d}  {ms:8.3f} ms  {fl / 1e12:7.3f} TFLOP  {(fl / ms / 1e9) if ms else 0:8.1f} TFLOP/s")
    eng.close()


if __name__ == "__main__":
    main()
