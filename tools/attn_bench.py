#!/usr/bin/env python3
"""Kernel-only timing of flash_attn64 at the UNet's four levels and of flash_attn512 at the VAE's mid block (batch 4): HIP events around
back-to-back launches.
usage: python tools/attn_bench.py [--hd512-only]  (GENPERCEPT_FLASH_RING3=1 selects the three-stage K / V ring of flash_attn64: run twice for an A/B)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genpercept_amd import engine as e  # noqa: E402


def hd512(d):
    for b, t in ((4, 9216), (1, 9216), (4, 2304)):
        c = 512
        g = torch.Generator().manual_seed(t)
        qk = (torch.randn(b, t, 2 * c, generator=g)).to(d).to(e.act_dtype())
        vt = torch.randn(b, c, t, generator=g).to(d).to(e.act_dtype())
        scale = 2.0 / c ** 0.5
        for _ in range(2):
            e.flash_attention_hd512(qk[:, :, :c], qk[:, :, c:], vt, scale)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                e.flash_attention_hd512(qk[:, :, :c], qk[:, :, c:], vt, scale)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        fl = 4.0 * b * t * t * c
        print(f"hd512 B={b} T={t:5d}  {best * 1e3:8.1f} us  {fl / best / 1e9:7.1f} TFLOP/s", flush=True)


def main():
    d = torch.device("cuda", 0)
    hd512(d)
    if "--hd512-only" in sys.argv:
        return
    for t, heads in ((9216, 5), (2304, 10), (576, 20), (144, 20)):
        b, c = 4, heads * 64
        tpad = (t + 63) // 64 * 64
        g = torch.Generator().manual_seed(t)
        qk = torch.randn(b, t, 2 * c, generator=g).to(d).to(e.act_dtype())
        vt = torch.zeros(b, c, tpad, dtype=e.act_dtype(), device=d)
        vt[:, :, :t] = torch.randn(b, c, t, generator=g).to(d).to(e.act_dtype())
        for _ in range(3):
            e.flash_attention(qk[:, :, :c], qk[:, :, c:], vt, heads)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                e.flash_attention(qk[:, :, :c], qk[:, :, c:], vt, heads)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        fl = 4.0 * b * heads * t * t * 64
        print(f"T={t:5d} heads={heads:2d}  {best * 1e3:8.1f} us  {fl / best / 1e9:7.1f} TFLOP/s  ring={'3' if os.environ.get('GENPERCEPT_FLASH_RING3') else '2'}", flush=True)


if __name__ == "__main__":
    main()
