#!/usr/bin/env python3
"""Error-budget ablation for north_star's "within 1e-3 rel" (VERDICT r1 item 1): which roundings carry the deviation from the fp32
oracle?  Runs on the CPU (no GPU needed): the fp32 oracle (oracle/sd21.py, full SD2.1 widths, seeded weights) is re-executed with
value-rounding hooks at the points where a reduced-precision engine rounds:

    W  weights (MFMA operand)                       X  activations entering a conv / linear / attention matmul (MFMA operand)
    O  stored outputs of non-residual layers         T  the residual trunk: resnet / attention / feed-forward sums and the skip stack
    (softmax probabilities are an X of the P.V matmul; GroupNorm / LayerNorm / softmax / SiLU arithmetic is fp32 everywhere, as in the
     engine; the final [0,1] map is fp32)

Each hook rounds to bf16 (8 significant bits), fp16 (11) or leaves fp32.  Rows of the table:
    engine-bf16        W,X,O,T = bf16     what libgenpercept_hip.so (bf16 build) does
    trunk-fp32         W,X,O = bf16, T = fp32        VERDICT's proposal: fp32 residual stream, bf16 only at MFMA operands and plain stores
    operands-only      W,X = bf16, O,T = fp32        the floor of ANY engine whose matrix cores take bf16 operands
    weights-only       W = bf16
    engine-fp16        W,X,O,T = fp16     what libgenpercept_hip_f16.so (fp16 build: v_mfma_f32_16x16x32_f16, same rate) does
    fp16 + bf16 probs  as engine-fp16 but softmax probabilities bf16
Reported: rel-RMS of latent / UNet output / decoded image against fp32 and mean|d| of the final depth map (in [0,1]).

usage: python tools/precision_ablation.py [--px 128] [--out profiles/r02_precision_ablation.json]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pipeline as opipe  # noqa: E402
from oracle import sd21 as osd  # noqa: E402


def rounder(kind):
    if kind == "bf16":
        return lambda t: t.to(torch.bfloat16).float()
    if kind == "fp16":
        return lambda t: t.clamp(-65504.0, 65504.0).to(torch.float16).float()
    return lambda t: t


class Hooks:
    def __init__(self, W="fp32", X="fp32", O="fp32", T="fp32", P=None, winograd=False):
        self.w, self.x, self.o, self.t = rounder(W), rounder(X), rounder(O), rounder(T)
        self.p = rounder(P if P is not None else X)
        self.winograd = winograd  # r4: the stride-1 3x3 convs (Cin >= 64) as Winograd F(2x2, 3x3) with the TRANSFORMED operands rounded
        self._wcache = {}

    def weight(self, sd, key):
        k = (id(sd), key)
        if k not in self._wcache:
            self._wcache[k] = self.w(sd[key])
        return self._wcache[k]


def install(h: Hooks):
    """Monkeypatch the oracle's building blocks with rounding versions (same math, same order)."""

    BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
    G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
    AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])

    def winograd_conv(x, w, bias):
        """F(2x2, 3x3): Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input tile (stride 2), the elementwise products summed over the input channels in
        fp32 (the MFMA accumulator); what a 16-bit engine has to round are the transformed operands U = G g G^T (once, from the fp32 weight) and
        V = B^T d B (from the 16-bit activations, AFTER the additions)."""
        b, c, hh, ww = x.shape
        he, we = (hh + 1) // 2 * 2, (ww + 1) // 2 * 2
        xp = F.pad(h.x(x), (1, 1 + we - ww, 1, 1 + he - hh))
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [b, c, he/2, we/2, 4, 4]
        V = h.x(torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT))
        key = (id(w), "wino")
        if key not in h._wcache:
            h._wcache[key] = h.w(torch.einsum("ij,ocjk,lk->ocil", G, w, G))     # [o, c, 4, 4]
        U = h._wcache[key]
        M = torch.einsum("ocil,bcyxil->boyxil", U, V)
        Y = torch.einsum("ij,boyxjk,lk->boyxil", AT, M, AT)                      # [b, o, he/2, we/2, 2, 2]
        out = Y.permute(0, 1, 2, 4, 3, 5).reshape(b, w.shape[0], he, we)[:, :, :hh, :ww]
        return out + bias[None, :, None, None] if bias is not None else out

    def winograd1d_conv(x, w, bias):
        """F(2, 3) along x only, direct along y (1.5x fewer multiplications, 4 accumulator sets instead of 16): per output row pair
        Y[., 2q + {0,1}] = A^T sum_ky [ (G g[ky]) . (B^T d[y + ky, 2q .. 2q+3]) ]"""
        b, c, hh, ww = x.shape
        we = (ww + 1) // 2 * 2
        xp = F.pad(h.x(x), (1, 1 + we - ww, 1, 1))
        d = xp.unfold(3, 4, 2)                                                    # [b, c, hh+2, we/2, 4]
        V = h.x(torch.einsum("ij,bcyxj->bcyxi", BT, d))                          # [b, c, hh+2, we/2, 4]
        key = (id(w), "wino1d")
        if key not in h._wcache:
            h._wcache[key] = h.w(torch.einsum("ij,ockj->ocki", G, w))            # [o, c, ky, 4]
        U = h._wcache[key]
        M = sum(torch.einsum("oci,bcyxi->boyxi", U[:, :, ky], V[:, :, ky:ky + hh]) for ky in range(3))
        Y = torch.einsum("ij,boyxj->boyxi", AT, M)                               # [b, o, hh, we/2, 2]
        out = Y.reshape(b, w.shape[0], hh, we)[:, :, :, :ww]
        return out + bias[None, :, None, None] if bias is not None else out

    def conv(x, sd, p, stride=1, padding=1):
        w = sd[p + ".weight"]
        if h.winograd == "1d" and stride == 1 and padding == 1 and w.shape[2] == 3 and w.shape[1] >= 64 and w.shape[0] >= 64:
            return winograd1d_conv(x, w, sd.get(p + ".bias"))
        if h.winograd and stride == 1 and padding == 1 and w.shape[2] == 3 and w.shape[1] >= 64 and w.shape[0] >= 64:
            return winograd_conv(x, w, sd.get(p + ".bias"))
        return F.conv2d(h.x(x), h.weight(sd, p + ".weight"), sd.get(p + ".bias"), stride=stride, padding=padding)

    def linear(x, sd, p):
        return F.linear(h.x(x), h.weight(sd, p + ".weight"), sd.get(p + ".bias"))

    def attention(q, k, v, heads):
        b, tq, c = q.shape
        hd = c // heads
        q, k, v = h.x(h.o(q)), h.x(h.o(k)), h.x(h.o(v))
        qh = q.view(b, tq, heads, hd).transpose(1, 2)
        kh = k.view(b, -1, heads, hd).transpose(1, 2)
        vh = v.view(b, -1, heads, hd).transpose(1, 2)
        w = torch.softmax((qh @ kh.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        return h.o((h.p(w) @ vh).transpose(1, 2).reshape(b, tq, c))

    def resnet_block(x, sd, p, groups, eps, temb):
        a = F.silu(osd._gn(x, sd, p + ".norm1", groups, eps))
        a = conv(a, sd, p + ".conv1")
        if temb is not None:
            a = a + F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])[:, :, None, None]  # folded in fp64
        a = h.o(a)
        a = F.silu(osd._gn(a, sd, p + ".norm2", groups, eps))
        a = conv(a, sd, p + ".conv2")
        if (p + ".conv_shortcut.weight") in sd:
            x = conv(x, sd, p + ".conv_shortcut", padding=0)
        return h.t(x + a)

    def transformer_2d(x, sd, p, heads, ctx, groups):
        b, c, hh, ww = x.shape
        res = x
        y = osd._gn(x, sd, p + ".norm", groups, 1e-6).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        y = h.t(linear(y, sd, p + ".proj_in"))
        bp = p + ".transformer_blocks.0"
        ln = lambda t, n: F.layer_norm(t, (c,), sd[bp + f".{n}.weight"], sd[bp + f".{n}.bias"], 1e-5)  # noqa: E731
        n = ln(y, "norm1")
        a = attention(linear(n, sd, bp + ".attn1.to_q"), linear(n, sd, bp + ".attn1.to_k"), linear(n, sd, bp + ".attn1.to_v"), heads)
        y = h.t(y + linear(a, sd, bp + ".attn1.to_out.0"))
        n = ln(y, "norm2")
        kc = F.linear(ctx, sd[bp + ".attn2.to_k.weight"])  # folded constants: fp32 in the engine
        vc = F.linear(ctx, sd[bp + ".attn2.to_v.weight"])
        q2 = h.o(linear(n, sd, bp + ".attn2.to_q"))
        a = h.o(osd._attention(q2, kc, vc, heads))
        y = h.t(y + linear(a, sd, bp + ".attn2.to_out.0"))
        n = ln(y, "norm3")
        hidden, gate = linear(n, sd, bp + ".ff.net.0.proj").chunk(2, dim=-1)
        y = h.t(y + linear(h.o(hidden * F.gelu(gate)), sd, bp + ".ff.net.2"))
        y = linear(y, sd, p + ".proj_out")
        return h.t(y.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + res)

    def vae_mid_attention(x, sd, p, groups, eps):
        b, c, hh, ww = x.shape
        gn, nq, nk, nv, no = osd._vae_attn_names(sd, p)
        y = osd._gn(x, sd, gn, groups, eps).reshape(b, c, hh * ww).transpose(1, 2)

        def lin(t, name):
            return F.linear(h.x(t), h.w(sd[name + ".weight"].reshape(c, c)), sd[name + ".bias"])

        a = attention(lin(y, nq), lin(y, nk), lin(y, nv), 1)
        return h.t(lin(a, no).transpose(1, 2).reshape(b, c, hh, ww) + x)

    saved = {k: getattr(osd, k) for k in ("_conv", "_linear", "resnet_block", "transformer_2d", "vae_mid_attention")}
    osd._conv, osd._linear = (lambda x, sd, p, stride=1, padding=1: h.o(conv(x, sd, p, stride, padding))), (lambda x, sd, p: h.o(linear(x, sd, p)))
    osd.resnet_block, osd.transformer_2d, osd.vae_mid_attention = resnet_block, transformer_2d, vae_mid_attention
    return saved


def restore(saved):
    for k, v in saved.items():
        setattr(osd, k, v)


def rel_rms(a, b):
    return float(((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)))


def map_rel_rms(out, ref):
    """rms(out - ref) / rms(ref - mean(ref)) of a final [0,1] map (the deviation relative to the map's own signal, not to its offset)"""
    return float((out - ref).pow(2).mean().sqrt() / ((ref - ref.mean()).pow(2).mean().sqrt() + 1e-12))


def run(cfgs, px, seed_u=11, seed_v=12):
    uc, vc = osd.UNetCfg(), osd.VAECfg()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), seed_u)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), seed_v)
    g = torch.Generator().manual_seed(77)
    rgb_u8 = torch.randint(0, 256, (1, 3, px, px), generator=g, dtype=torch.uint8)
    rgb_u8[:, :, : px // 2] //= 2
    ctx = torch.randn(2, 1024, generator=g)
    rgb = opipe.normalize_rgb(rgb_u8)

    def forward():
        with torch.no_grad():
            lat = osd.encode_rgb(vsd, vc, rgb)
            v, _ = osd.unet_forward(usd, uc, lat, 1, ctx[None])
            dec = osd.decode_pred(vsd, vc, -v, "normal")
            depth = ((dec.mean(dim=1, keepdim=True).clamp(-1, 1)) + 1) / 2
            normal = (dec.clamp(-1, 1) + 1) / 2
        return lat, v, dec, depth, normal

    def staged(ref):
        """each stage fed with the fp32 reference input, so stages are judged on their own (like tests/test_e2e_gpu.py)"""
        with torch.no_grad():
            lat = osd.encode_rgb(vsd, vc, rgb)
            v, _ = osd.unet_forward(usd, uc, ref[0], 1, ctx[None])
            dec = osd.decode_pred(vsd, vc, -ref[1], "normal")
        return lat, v, dec

    t0 = time.time()
    ref = forward()
    print(f"fp32 reference: {time.time() - t0:.1f} s at {px}x{px}", flush=True)
    rows = []
    for name, kw in cfgs:
        saved = install(Hooks(**kw))
        try:
            e2e = forward()
            st = staged(ref)
        finally:
            restore(saved)
        row = {"config": name, "hooks": kw,
               "stage_rel_rms": {"vae_encode": rel_rms(st[0], ref[0]), "unet": rel_rms(st[1], ref[1]), "vae_decode": rel_rms(st[2], ref[2])},
               "e2e_rel_rms": {"latent": rel_rms(e2e[0], ref[0]), "unet": rel_rms(e2e[1], ref[1]), "decoded": rel_rms(e2e[2], ref[2])},
               "depth_mean_abs": float((e2e[3] - ref[3]).abs().mean()), "depth_max_abs": float((e2e[3] - ref[3]).abs().max()),
               "normal_mean_abs": float((e2e[4] - ref[4]).abs().mean()),
               # rel-RMS of the final maps = rms(out - ref) / rms(ref - mean(ref)): the "relative" reading of north_star's 1e-3
               "depth_rel_rms": map_rel_rms(e2e[3], ref[3]), "normal_rel_rms": map_rel_rms(e2e[4], ref[4])}
        rows.append(row)
        s = row["stage_rel_rms"]
        print(f"{name:22s} stages enc {s['vae_encode']:.2e} unet {s['unet']:.2e} dec {s['vae_decode']:.2e} | depth mean|d| {row['depth_mean_abs']:.2e} "
              f"max {row['depth_max_abs']:.2e} rel-rms {row['depth_rel_rms']:.2e} | normal mean|d| {row['normal_mean_abs']:.2e} rel-rms {row['normal_rel_rms']:.2e}",
              flush=True)
    return rows


CONFIGS = [
    ("engine-bf16", dict(W="bf16", X="bf16", O="bf16", T="bf16")),
    ("trunk-fp32", dict(W="bf16", X="bf16", O="bf16", T="fp32")),
    ("operands-only-bf16", dict(W="bf16", X="bf16")),
    ("weights-only-bf16", dict(W="bf16")),
    ("engine-fp16", dict(W="fp16", X="fp16", O="fp16", T="fp16")),
    ("fp16+bf16-probs", dict(W="fp16", X="fp16", O="fp16", T="fp16", P="bf16")),
    ("bf16+fp16-trunk", dict(W="bf16", X="bf16", O="bf16", T="fp16")),
    # r4: what would it take to reach 1e-3 under the RELATIVE reading (rel-RMS of the final maps)?
    ("fp16+fp32-trunk", dict(W="fp16", X="fp16", O="fp16", T="fp32")),       # fp16 MFMA operands and plain stores, fp32 residual stream / skip stack
    ("operands-only-fp16", dict(W="fp16", X="fp16")),                         # floor of ANY engine with fp16 MFMA operands (everything stored fp32)
    ("weights-only-fp16", dict(W="fp16")),
    ("fp16-acts+fp32-weights", dict(X="fp16", O="fp16", T="fp16")),           # (not buildable on MFMA: shows what the weight rounding alone carries)
    # r4: would Winograd F(2x2, 3x3) for the stride-1 3x3 convs (2.25x fewer MFMA flops) keep the parity?  transformed operands rounded to 16 bits
    ("winograd-fp32", dict(winograd=True)),                                    # sanity: the transform alone (fp32 reassociation only)
    ("engine-fp16+winograd", dict(W="fp16", X="fp16", O="fp16", T="fp16", winograd=True)),
    ("engine-bf16+winograd", dict(W="bf16", X="bf16", O="bf16", T="bf16", winograd=True)),
    ("winograd1d-fp32", dict(winograd="1d")),                                  # F(2,3) along x only (1.5x fewer flops, 4 accumulator sets)
    ("engine-fp16+winograd1d", dict(W="fp16", X="fp16", O="fp16", T="fp16", winograd="1d")),
    ("engine-bf16+winograd1d", dict(W="bf16", X="bf16", O="bf16", T="bf16", winograd="1d")),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--px", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_precision_ablation.json"))
    ap.add_argument("--only", default="", help="comma-separated config names (default: all)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    rows = run([c for c in CONFIGS if not args.only or c[0] in args.only.split(",")], args.px)
    if args.only and os.path.exists(args.out):  # (re-)measure a few rows: the others stay as they are in the file
        old = json.load(open(args.out))
        if old.get("px") == args.px:
            names = {r["config"] for r in rows}
            rows = [r for r in old["rows"] if r["config"] not in names] + rows
    json.dump({"px": args.px, "weights": "seeded synthetic, full SD2.1 widths (oracle.synth_state_dict seeds 11 / 12)", "rows": rows},
              open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
