#!/usr/bin/env python3
"""Per-layer attribution of the 16-bit engines' deviation from the fp32 oracle (VERDICT r4 "next round" item 1).

tools/precision_ablation.py answers WHICH ROUNDINGS carry the error (weights / operands / stores / trunk), globally.  This tool answers
WHERE: the same value-rounding hooks (W, X, O, T = what the fp16 or bf16 library rounds), but switched on for ONE unit of the network at a
time -- a unit is a resnet block, a transformer block, the VAE mid attention or a free-standing conv (conv_in / conv_out / down- and
upsamplers / quant convs): 87 units at full SD2.1 widths -- and then cumulatively:

  solo[u]      everything fp32 except unit u            -> the error unit u injects into the final map, alone
  loo[u]       everything 16-bit except unit u (fp32)   -> what protecting u alone buys (a split-operand = 3-MFMA unit is modelled as fp32)
  curve[k]     units sorted by solo error^2 per FLOP; the first k protected (fp32), the rest 16-bit -> final-map error against the FLOP share
               that would run at 3x MFMA cost
  budget       the best protected set under a FLOP budget (10 % of the path's FLOPs = +20 % MFMA work): does ANY cheap subset reach
               mean_abs <= 1e-3 and rel_rms <= 2e-3?

CPU only (fp32 oracle, oracle/sd21.py, full widths, seeded weights, one px x px image); writes profiles/r05_precision_attribution.json.
usage: python tools/precision_attribution.py [--px 128] [--kinds fp16,bf16] [--out ...]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import precision_ablation as pa  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402
from oracle import sd21 as osd  # noqa: E402


class ScopedHooks(pa.Hooks):
    """pa.Hooks whose roundings apply only while the CURRENT unit (set by the wrappers below) is in `on` (a set of unit names)."""

    def __init__(self, kind, on):
        super().__init__()
        r = pa.rounder(kind)
        self.on, self.cur = on, None
        gate = lambda t: r(t) if self.cur in self.on else t  # noqa: E731
        self.w = self.x = self.o = self.t = self.p = gate

    def weight(self, sd, key):  # the base class caches per key; a key belongs to exactly one unit, so the cache stays valid
        k = (id(sd), key)
        if k not in self._wcache:
            self._wcache[k] = self.w(sd[key])
        return self._wcache[k]


class Tracker:
    """names the unit a patched oracle function runs in, and (counting pass) sums 2*MAC per unit"""

    def __init__(self, tags):
        self.tags, self.hooks, self.flops, self.order, self.cur, self.frozen = tags, None, {}, [], None, False

    def unit(self, sd, p):
        tag = self.tags[id(sd)]
        if tag == "vae":
            tag = "vae_dec" if (p.startswith("decoder.") or p.startswith("post_quant")) else "vae_enc"
        for tail in (".conv1", ".conv2", ".conv_shortcut"):
            if p.endswith(tail):
                p = p[: -len(tail)]
        return f"{tag}:{p}"

    def enter(self, sd, p):
        if self.cur is not None:      # already inside a block-level unit
            return False
        self.cur = self.unit(sd, p)
        if self.cur not in self.flops:
            self.flops[self.cur] = 0.0
            self.order.append(self.cur)
        if self.hooks is not None:
            self.hooks.cur = self.cur
        return True

    def leave(self):
        self.cur = None
        if self.hooks is not None:
            self.hooks.cur = None

    def add(self, fl):
        if self.cur is not None and not self.frozen:   # (counted in the fp32 pass only)
            self.flops[self.cur] += fl


def wrap_scoped(trk):
    """wrap the (possibly rounding) block functions of oracle.sd21 so that each call runs under its unit's name"""
    names = ("_conv", "_linear", "resnet_block", "transformer_2d", "vae_mid_attention")
    inner = {k: getattr(osd, k) for k in names}

    def make(fn, attn_flops=None):
        def f(x, sd, p, *a, **k):
            own = trk.enter(sd, p)
            try:
                if own and attn_flops is not None:
                    trk.add(attn_flops(x))
                return fn(x, sd, p, *a, **k)
            finally:
                if own:
                    trk.leave()
        return f

    osd._conv, osd._linear = make(inner["_conv"]), make(inner["_linear"])
    osd.resnet_block = make(inner["resnet_block"])
    # self-attention: QK^T and PV = 4 T^2 C per image; the 2-token cross-attention matmuls are negligible
    osd.transformer_2d = make(inner["transformer_2d"], lambda x: 4.0 * x.shape[0] * (x.shape[2] * x.shape[3]) ** 2 * x.shape[1])
    osd.vae_mid_attention = make(inner["vae_mid_attention"], lambda x: 4.0 * x.shape[0] * (x.shape[2] * x.shape[3]) ** 2 * x.shape[1])
    return inner


def count_flops(trk, forward):
    """one fp32 pass with F.conv2d / F.linear wrapped: 2*MAC per unit"""
    c2d, lin = F.conv2d, F.linear

    def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
        y = c2d(x, w, b, stride, padding, *a, **k)
        trk.add(2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3])
        return y

    def linear(x, w, b=None):
        y = lin(x, w, b)
        trk.add(2.0 * y.numel() * w.shape[1])
        return y

    F.conv2d, F.linear = conv2d, linear
    try:
        return forward()
    finally:
        F.conv2d, F.linear = c2d, lin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--px", type=int, default=128)
    ap.add_argument("--kinds", default="fp16,bf16")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_precision_attribution.json"))
    ap.add_argument("--budget", type=float, default=0.10, help="FLOP share that may be protected (3x MFMA cost there)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    uc, vc = osd.UNetCfg(), osd.VAECfg()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 11)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 12)
    g = torch.Generator().manual_seed(77)
    px = args.px
    rgb_u8 = torch.randint(0, 256, (1, 3, px, px), generator=g, dtype=torch.uint8)
    rgb_u8[:, :, : px // 2] //= 2
    ctx = torch.randn(2, 1024, generator=g)
    rgb = opipe.normalize_rgb(rgb_u8)
    trk = Tracker({id(usd): "unet", id(vsd): "vae"})

    def forward():
        with torch.no_grad():
            lat = osd.encode_rgb(vsd, vc, rgb)
            v, _ = osd.unet_forward(usd, uc, lat, 1, ctx[None])
            dec = osd.decode_pred(vsd, vc, -v, "normal")
            return ((dec.mean(dim=1, keepdim=True).clamp(-1, 1)) + 1) / 2, (dec.clamp(-1, 1) + 1) / 2

    def metrics(out, ref):
        d = {}
        for name, o, r in (("depth", out[0], ref[0]), ("normal", out[1], ref[1])):
            d[name + "_mean_abs"] = float((o - r).abs().mean())
            d[name + "_rel_rms"] = pa.map_rel_rms(o, r)
        return d

    # fp32 reference + FLOPs per unit
    inner = wrap_scoped(trk)
    t0 = time.time()
    ref = count_flops(trk, forward)
    trk.frozen = True
    for k, v in inner.items():
        setattr(osd, k, v)
    units = [u for u in trk.order if trk.flops[u] > 0 and "time_embedding" not in u and "time_emb_proj" not in u]
    total = sum(trk.flops[u] for u in units)
    print(f"fp32 reference {time.time() - t0:.1f} s at {px}x{px}; {len(units)} units, {total / 1e12:.3f} TFLOP", flush=True)

    def run_with(kind, on):
        h = ScopedHooks(kind, set(on))
        saved = pa.install(h)
        trk.hooks = h
        inner2 = wrap_scoped(trk)
        try:
            return metrics(forward(), ref)
        finally:
            for k, v in inner2.items():
                setattr(osd, k, v)
            pa.restore(saved)
            trk.hooks = None

    result = {"px": px, "weights": "seeded synthetic, full SD2.1 widths (oracle.synth_state_dict seeds 11 / 12)", "total_tflop": total / 1e12,
              "units": [{"unit": u, "flop_share": trk.flops[u] / total} for u in units], "kinds": {}}
    for kind in args.kinds.split(","):
        t0 = time.time()
        full = run_with(kind, units)
        print(f"[{kind}] all units rounded: depth {full['depth_mean_abs']:.2e} / {full['depth_rel_rms']:.2e}  normal {full['normal_mean_abs']:.2e} / "
              f"{full['normal_rel_rms']:.2e}  ({time.time() - t0:.1f} s)", flush=True)
        solo, loo = {}, {}
        for i, u in enumerate(units):
            solo[u] = run_with(kind, [u])
            loo[u] = run_with(kind, [w for w in units if w != u])
            print(f"[{kind}] {i + 1:3d}/{len(units)} {u:55s} share {trk.flops[u] / total:6.3%}  solo rel_rms {solo[u]['depth_rel_rms']:.2e}  "
                  f"without it {loo[u]['depth_rel_rms']:.2e}", flush=True)
        # cumulative curve: protect units in the order of error variance per FLOP (depth rel_rms^2 + normal rel_rms^2)
        var = {u: solo[u]["depth_rel_rms"] ** 2 + solo[u]["normal_rel_rms"] ** 2 for u in units}
        order = sorted(units, key=lambda u: -var[u] / trk.flops[u])
        curve, prot, share = [], [], 0.0
        marks = sorted(set(list(range(0, len(order) + 1, 4)) + [len(order)]))
        for k in range(len(order) + 1):
            if k:
                prot.append(order[k - 1])
                share += trk.flops[order[k - 1]] / total
            if k in marks:
                m = run_with(kind, [u for u in units if u not in prot])
                curve.append({"protected_units": k, "protected_flop_share": share, **m})
                print(f"[{kind}] protect {k:3d} units = {share:6.2%} of the FLOPs (x3 there = +{2 * share:.0%} MFMA work): depth {m['depth_mean_abs']:.2e} / "
                      f"{m['depth_rel_rms']:.2e}  normal {m['normal_mean_abs']:.2e} / {m['normal_rel_rms']:.2e}", flush=True)
        # the best set under the budget: greedy by variance per FLOP, skipping units that do not fit
        best, used = [], 0.0
        for u in order:
            s = trk.flops[u] / total
            if used + s <= args.budget:
                best.append(u)
                used += s
        mb = run_with(kind, [u for u in units if u not in best])
        print(f"[{kind}] best set under {args.budget:.0%} of the FLOPs: {len(best)} units, {used:.2%}: depth {mb['depth_mean_abs']:.2e} / {mb['depth_rel_rms']:.2e}  "
              f"normal {mb['normal_mean_abs']:.2e} / {mb['normal_rel_rms']:.2e}", flush=True)
        sumvar = sum(solo[u]["depth_rel_rms"] ** 2 for u in units) ** 0.5
        result["kinds"][kind] = {"all": full, "solo": solo, "leave_one_out": loo, "order_by_variance_per_flop": order, "curve": curve,
                                 "budget": {"flop_share": args.budget, "units": best, "used_share": used, **mb},
                                 "rss_of_solo_depth_rel_rms": sumvar}
        print(f"[{kind}] root-sum-square of the solo depth rel_rms: {sumvar:.2e} (all units together: {full['depth_rel_rms']:.2e})", flush=True)
        json.dump(result, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
