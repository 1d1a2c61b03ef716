"""Weight manifests (diffusers key -> shape) of the three modules and a seeded synthetic initialiser.

There are no checkpoints in the build/bench environment (no network), so benchmarks and GPU tests run on random-init
weights of the exact SD2.1 / GenPercept architectures; real checkpoints use the same keys (run.py:296-357) and load through
GenPerceptPipeline unchanged.  tests/test_host.py checks these manifests against the oracle's and against the public
parameter counts (865 910 724 / 83 653 863 / 18 474 753).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import torch

from .config import DPTConfig, UNetConfig, VAEConfig

Manifest = "OrderedDict[str, Tuple[int, ...]]"


class _M(OrderedDict):
    def conv(self, p, cout, cin, k=3, bias=True):
        self[p + ".weight"] = (cout, cin, k, k)
        if bias:
            self[p + ".bias"] = (cout,)

    def lin(self, p, cout, cin, bias=True):
        self[p + ".weight"] = (cout, cin)
        if bias:
            self[p + ".bias"] = (cout,)

    def norm(self, p, c):
        self[p + ".weight"] = (c,)
        self[p + ".bias"] = (c,)

    def resnet(self, p, cin, cout, temb: Optional[int]):
        self.norm(p + ".norm1", cin)
        self.conv(p + ".conv1", cout, cin)
        if temb:
            self.lin(p + ".time_emb_proj", cout, temb)
        self.norm(p + ".norm2", cout)
        self.conv(p + ".conv2", cout, cout)
        if cin != cout:
            self.conv(p + ".conv_shortcut", cout, cin, 1)

    def transformer(self, p, c, ctx):
        self.norm(p + ".norm", c)
        self.lin(p + ".proj_in", c, c)
        b = p + ".transformer_blocks.0"
        self.norm(b + ".norm1", c)
        for n in ("to_q", "to_k", "to_v"):
            self.lin(f"{b}.attn1.{n}", c, c, bias=False)
        self.lin(b + ".attn1.to_out.0", c, c)
        self.norm(b + ".norm2", c)
        self.lin(b + ".attn2.to_q", c, c, bias=False)
        self.lin(b + ".attn2.to_k", c, ctx, bias=False)
        self.lin(b + ".attn2.to_v", c, ctx, bias=False)
        self.lin(b + ".attn2.to_out.0", c, c)
        self.norm(b + ".norm3", c)
        self.lin(b + ".ff.net.0.proj", 8 * c, c)
        self.lin(b + ".ff.net.2", c, 4 * c)
        self.lin(p + ".proj_out", c, c)


def unet_manifest(cfg: UNetConfig = UNetConfig()) -> Manifest:
    m = _M()
    bo, te, ctx, L = cfg.block_out_channels, cfg.block_out_channels[0] * 4, cfg.cross_attention_dim, cfg.layers_per_block
    m.conv("conv_in", bo[0], cfg.in_channels)
    m.lin("time_embedding.linear_1", te, bo[0])
    m.lin("time_embedding.linear_2", te, te)
    ch = bo[0]
    for i, co in enumerate(bo):
        for j in range(L):
            m.resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, te)
            if cfg.down_has_attn[i]:
                m.transformer(f"down_blocks.{i}.attentions.{j}", co, ctx)
        ch = co
        if i != len(bo) - 1:
            m.conv(f"down_blocks.{i}.downsamplers.0.conv", co, co)
    m.resnet("mid_block.resnets.0", bo[-1], bo[-1], te)
    m.transformer("mid_block.attentions.0", bo[-1], ctx)
    m.resnet("mid_block.resnets.1", bo[-1], bo[-1], te)
    rev, rattn = bo[::-1], cfg.down_has_attn[::-1]
    out = rev[0]
    for i in range(len(rev)):
        prev, out, inp = out, rev[i], rev[min(i + 1, len(rev) - 1)]
        for j in range(L + 1):
            skip = inp if j == L else out
            m.resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out) + skip, out, te)
            if rattn[i]:
                m.transformer(f"up_blocks.{i}.attentions.{j}", out, ctx)
        if i != len(rev) - 1:
            m.conv(f"up_blocks.{i}.upsamplers.0.conv", out, out)
    if cfg.has_out:
        m.norm("conv_norm_out", bo[0])
        m.conv("conv_out", cfg.out_channels, bo[0])
    return m


def vae_manifest(cfg: VAEConfig = VAEConfig()) -> Manifest:
    m = _M()
    bo, L, z = cfg.block_out_channels, cfg.layers_per_block, cfg.latent_channels

    def mid(p, c):
        m.resnet(p + ".resnets.0", c, c, None)
        a = p + ".attentions.0"
        m.norm(a + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            m.lin(f"{a}.{n}", c, c)
        m.resnet(p + ".resnets.1", c, c, None)

    m.conv("encoder.conv_in", bo[0], cfg.in_channels)
    ch = bo[0]
    for i, co in enumerate(bo):
        for j in range(L):
            m.resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, None)
        ch = co
        if i != len(bo) - 1:
            m.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
    mid("encoder.mid_block", bo[-1])
    m.norm("encoder.conv_norm_out", bo[-1])
    m.conv("encoder.conv_out", 2 * z, bo[-1])
    m.conv("quant_conv", 2 * z, 2 * z, 1)
    m.conv("post_quant_conv", z, z, 1)
    m.conv("decoder.conv_in", bo[-1], z)
    mid("decoder.mid_block", bo[-1])
    ch = bo[-1]
    for i, co in enumerate(bo[::-1]):
        for j in range(L + 1):
            m.resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, None)
        ch = co
        if i != len(bo) - 1:
            m.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
    m.norm("decoder.conv_norm_out", bo[0])
    m.conv("decoder.conv_out", cfg.out_channels, bo[0])
    return m


def dpt_manifest(cfg: DPTConfig = DPTConfig()) -> Manifest:
    m = _M()
    f, neck = cfg.fusion_hidden_size, cfg.neck_hidden_sizes
    m.conv("feature_upsample_0.conv", neck[0], neck[0])
    for i, c in enumerate(neck):
        m.conv(f"neck.convs.{i}", f, c, bias=False)
    for i in range(len(neck)):
        p = f"neck.fusion_stage.layers.{i}"
        m.conv(p + ".projection", f, f, 1)
        for r in ((2,) if i == 0 else (1, 2)):
            m.conv(f"{p}.residual_layer{r}.convolution1", f, f, bias=False)
            m.conv(f"{p}.residual_layer{r}.convolution2", f, f, bias=False)
    m.conv("head.projection", f, f)
    m.conv("head.head.0", f // 2, f)
    m.conv("head.head.2", 32, f // 2)
    m.conv("head.head.4", 1, 32, 1)
    return m


def count_params(manifest: Dict[str, Sequence[int]]) -> int:
    return sum(int(math.prod(s)) for s in manifest.values())


def synth_state_dict(manifest: Dict[str, Sequence[int]], seed: int = 0, gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 weights, tensor by tensor in manifest order from ONE CPU generator: variance-preserving uniform for
    conv/linear weights, N(0, 0.05) biases, norm affine ~ (1 + 0.1 N, 0.1 N).  Deterministic for a fixed torch build."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in manifest.items():
        shape = tuple(int(s) for s in shape)
        is_norm = len(shape) == 1 and (".norm" in name or "group_norm" in name or "conv_norm_out" in name)
        if is_norm:
            t = 0.1 * torch.randn(shape, generator=g)
            if name.endswith("weight"):
                t = 1.0 + t
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            a = gain * math.sqrt(3.0 / max(int(math.prod(shape[1:])), 1))
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * a
        sd[name] = t.float().contiguous()
    return sd


def merge_lora_state_dict(sd: Dict[str, torch.Tensor], scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Fold PEFT LoRA adapters into their base weights so the engine sees a plain diffusers state dict.

    The reference adds rank-r adapters (alpha = r) to `to_q / to_k / to_v / to_out.0` before `load_state_dict` (run.py:345-357),
    which makes the checkpoint's keys `<module>.base_layer.weight`, `<module>.lora_A.<adapter>.weight` ([r, in]) and
    `<module>.lora_B.<adapter>.weight` ([out, r]); the module computes base(x) + (alpha / r) * B(A(x)), i.e. the effective weight is
    W + (alpha / r) * B @ A.  `scale` defaults to alpha / r = 1 (the reference's configuration).  A dict without adapter keys is
    returned unchanged (same object).
    """
    if not any(".lora_A." in k or ".base_layer." in k for k in sd):
        return sd
    s = 1.0 if scale is None else float(scale)
    out: Dict[str, torch.Tensor] = OrderedDict()
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        if ".base_layer." in k:
            mod, leaf = k.split(".base_layer.")
            if leaf == "weight":
                a = [t for kk, t in sd.items() if kk.startswith(mod + ".lora_A.") and kk.endswith(".weight")]
                b = [t for kk, t in sd.items() if kk.startswith(mod + ".lora_B.") and kk.endswith(".weight")]
                if len(a) != len(b) or len(a) > 1:
                    raise ValueError(f"{mod}: expected one lora_A / lora_B pair, found {len(a)} / {len(b)}")
                w = v.float()
                if a:
                    delta = b[0].float().reshape(b[0].shape[0], -1) @ a[0].float().reshape(a[0].shape[0], -1)
                    w = w + s * delta.reshape(w.shape)
                out[mod + ".weight"] = w.to(v.dtype)
            else:
                out[mod + "." + leaf] = v
        else:
            out[k] = v
    return out


def replace_unet_conv_in(unet_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """run.py:59-78 (`_replace_unet_conv_in`): the marigold arch feeds [rgb_latent, pred_latent]; a UNet saved with the plain latent
    input gets its conv_in weight repeated along the input axis and halved ("half the activation magnitude"), bias unchanged."""
    sd = dict(unet_sd)
    w = sd["conv_in.weight"]
    sd["conv_in.weight"] = (w.repeat(1, 2, 1, 1) * 0.5).contiguous()
    return sd
