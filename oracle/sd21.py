"""CPU oracle (TEST INFRASTRUCTURE, not product code): plain-PyTorch fp32 restatement of the
SD2.1 modules on GenPercept's one-step inference path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (genpercept_amd/) never imports it.

PARITY UNPINNED by the reference's own tests: the reference ships no tests/fixtures for this path
 and all arithmetic lives in the un-vendored third-party `diffusers`
(requirements.txt:2 `diffusers>=0.25.0`; configs stamped 0.29.2).  This file restates the published
SD2.1 architecture that the reference instantiates at
  run.py:316-320 (UNet2DConditionModel / CustomUNet2DConditionModel.from_pretrained(sd21,'unet')),
  run.py:309 (AutoencoderKL), genpercept_pipeline.py:455-463,500-501,521-522 (call sites),
and follows the control flow of genpercept/models/custom_unet.py:109-119,146-170,273,305-327,
341-352,365-415 for the UNet forward (skip popping order, upsample_size, multi_level_feats,
return_feature).  What pins it instead: exact public parameter/tensor counts (tests/test_oracle.py),
the DPT head cross-check against the reference's own dpt_head.py (tests/golden/make_goldens.py),
and the closed-form scheduler identity.

State dicts use the diffusers key layout so real checkpoints load as-is.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class UNetCfg:
    """SD2.1 `unet/config.json` values."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # `attention_head_dim` in the SD2.1 config is the NUMBER OF HEADS per level (Appendix B.4)
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)
    # which down blocks carry attention (CrossAttnDownBlock2D x3, DownBlock2D)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    has_out: bool = True  # False for the DPT-head UNets (conv_norm_out/conv_out deleted, run.py:316-318)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @staticmethod
    def tiny() -> "UNetCfg":
        """Reduced-width config with the same topology (head_dim stays 64) for fast parity tests."""
        return UNetCfg(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=64)


@dataclass(frozen=True)
class VAECfg:
    """SD2.1 `vae/config.json` values."""
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @staticmethod
    def tiny() -> "VAECfg":
        return VAECfg(block_out_channels=(64, 128, 128, 128))


# --------------------------------------------------------------------------------------
# manifests: ordered {diffusers key: shape}
# --------------------------------------------------------------------------------------
def _resnet_keys(m: "OrderedDict[str, Tuple[int, ...]]", p: str, cin: int, cout: int, temb: Optional[int]):
    m[p + ".norm1.weight"] = (cin,)
    m[p + ".norm1.bias"] = (cin,)
    m[p + ".conv1.weight"] = (cout, cin, 3, 3)
    m[p + ".conv1.bias"] = (cout,)
    if temb is not None:
        m[p + ".time_emb_proj.weight"] = (cout, temb)
        m[p + ".time_emb_proj.bias"] = (cout,)
    m[p + ".norm2.weight"] = (cout,)
    m[p + ".norm2.bias"] = (cout,)
    m[p + ".conv2.weight"] = (cout, cout, 3, 3)
    m[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        m[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        m[p + ".conv_shortcut.bias"] = (cout,)


def _transformer_keys(m, p: str, c: int, ctx: int):
    m[p + ".norm.weight"] = (c,)
    m[p + ".norm.bias"] = (c,)
    m[p + ".proj_in.weight"] = (c, c)
    m[p + ".proj_in.bias"] = (c,)
    b = p + ".transformer_blocks.0"
    m[b + ".norm1.weight"] = (c,)
    m[b + ".norm1.bias"] = (c,)
    m[b + ".attn1.to_q.weight"] = (c, c)
    m[b + ".attn1.to_k.weight"] = (c, c)
    m[b + ".attn1.to_v.weight"] = (c, c)
    m[b + ".attn1.to_out.0.weight"] = (c, c)
    m[b + ".attn1.to_out.0.bias"] = (c,)
    m[b + ".norm2.weight"] = (c,)
    m[b + ".norm2.bias"] = (c,)
    m[b + ".attn2.to_q.weight"] = (c, c)
    m[b + ".attn2.to_k.weight"] = (c, ctx)
    m[b + ".attn2.to_v.weight"] = (c, ctx)
    m[b + ".attn2.to_out.0.weight"] = (c, c)
    m[b + ".attn2.to_out.0.bias"] = (c,)
    m[b + ".norm3.weight"] = (c,)
    m[b + ".norm3.bias"] = (c,)
    m[b + ".ff.net.0.proj.weight"] = (8 * c, c)
    m[b + ".ff.net.0.proj.bias"] = (8 * c,)
    m[b + ".ff.net.2.weight"] = (c, 4 * c)
    m[b + ".ff.net.2.bias"] = (c,)
    m[p + ".proj_out.weight"] = (c, c)
    m[p + ".proj_out.bias"] = (c,)


def unet_up_plan(cfg: UNetCfg) -> List[dict]:
    """Channel plan of the up path (diffusers get_up_block wiring; Appendix A.1)."""
    rev = list(reversed(cfg.block_out_channels))
    rev_heads = list(reversed(cfg.num_heads))
    rev_attn = list(reversed(cfg.down_has_attn))
    n = len(rev)
    plan = []
    out_ch = rev[0]
    for i in range(n):
        prev = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, n - 1)]
        resnets = []
        for j in range(cfg.layers_per_block + 1):
            skip = in_ch if j == cfg.layers_per_block else out_ch
            rin = prev if j == 0 else out_ch
            resnets.append((rin, skip, out_ch))
        plan.append(dict(resnets=resnets, out=out_ch, attn=rev_attn[i], heads=rev_heads[i], upsample=(i != n - 1)))
    return plan


def unet_manifest(cfg: UNetCfg = UNetCfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    c0 = cfg.block_out_channels[0]
    te = cfg.time_embed_dim
    m["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    m["conv_in.bias"] = (c0,)
    m["time_embedding.linear_1.weight"] = (te, c0)
    m["time_embedding.linear_1.bias"] = (te,)
    m["time_embedding.linear_2.weight"] = (te, te)
    m["time_embedding.linear_2.bias"] = (te,)
    ch = c0
    nb = len(cfg.block_out_channels)
    for i, co in enumerate(cfg.block_out_channels):
        for j in range(cfg.layers_per_block):
            _resnet_keys(m, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, te)
            if cfg.down_has_attn[i]:
                _transformer_keys(m, f"down_blocks.{i}.attentions.{j}", co, cfg.cross_attention_dim)
        ch = co
        if i != nb - 1:
            m[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            m[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
    cm = cfg.block_out_channels[-1]
    _resnet_keys(m, "mid_block.resnets.0", cm, cm, te)
    _transformer_keys(m, "mid_block.attentions.0", cm, cfg.cross_attention_dim)
    _resnet_keys(m, "mid_block.resnets.1", cm, cm, te)
    for i, blk in enumerate(unet_up_plan(cfg)):
        for j, (rin, skip, rout) in enumerate(blk["resnets"]):
            _resnet_keys(m, f"up_blocks.{i}.resnets.{j}", rin + skip, rout, te)
            if blk["attn"]:
                _transformer_keys(m, f"up_blocks.{i}.attentions.{j}", rout, cfg.cross_attention_dim)
        if blk["upsample"]:
            m[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (blk["out"], blk["out"], 3, 3)
            m[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (blk["out"],)
    if cfg.has_out:
        m["conv_norm_out.weight"] = (c0,)
        m["conv_norm_out.bias"] = (c0,)
        m["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
        m["conv_out.bias"] = (cfg.out_channels,)
    return m


def _vae_attn_keys(m, p: str, c: int):
    m[p + ".group_norm.weight"] = (c,)
    m[p + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        m[f"{p}.{n}.weight"] = (c, c)
        m[f"{p}.{n}.bias"] = (c,)


def vae_manifest(cfg: VAECfg = VAECfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    bo = cfg.block_out_channels
    nb = len(bo)
    # encoder
    m["encoder.conv_in.weight"] = (bo[0], cfg.in_channels, 3, 3)
    m["encoder.conv_in.bias"] = (bo[0],)
    ch = bo[0]
    for i, co in enumerate(bo):
        for j in range(cfg.layers_per_block):
            _resnet_keys(m, f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, None)
        ch = co
        if i != nb - 1:
            m[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            m[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
    cm = bo[-1]
    _resnet_keys(m, "encoder.mid_block.resnets.0", cm, cm, None)
    _vae_attn_keys(m, "encoder.mid_block.attentions.0", cm)
    _resnet_keys(m, "encoder.mid_block.resnets.1", cm, cm, None)
    m["encoder.conv_norm_out.weight"] = (cm,)
    m["encoder.conv_norm_out.bias"] = (cm,)
    m["encoder.conv_out.weight"] = (2 * cfg.latent_channels, cm, 3, 3)
    m["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    m["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1)
    m["quant_conv.bias"] = (2 * cfg.latent_channels,)
    m["post_quant_conv.weight"] = (cfg.latent_channels, cfg.latent_channels, 1, 1)
    m["post_quant_conv.bias"] = (cfg.latent_channels,)
    # decoder
    m["decoder.conv_in.weight"] = (cm, cfg.latent_channels, 3, 3)
    m["decoder.conv_in.bias"] = (cm,)
    _resnet_keys(m, "decoder.mid_block.resnets.0", cm, cm, None)
    _vae_attn_keys(m, "decoder.mid_block.attentions.0", cm)
    _resnet_keys(m, "decoder.mid_block.resnets.1", cm, cm, None)
    rev = list(reversed(bo))
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet_keys(m, f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, None)
        ch = co
        if i != nb - 1:
            m[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            m[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    m["decoder.conv_norm_out.weight"] = (rev[-1],)
    m["decoder.conv_norm_out.bias"] = (rev[-1],)
    m["decoder.conv_out.weight"] = (cfg.out_channels, rev[-1], 3, 3)
    m["decoder.conv_out.bias"] = (cfg.out_channels,)
    return m


def count_params(manifest: Dict[str, Sequence[int]]) -> int:
    return sum(int(math.prod(s)) for s in manifest.values())


# --------------------------------------------------------------------------------------
# seeded synthetic weights (there are no checkpoints in this environment)
# --------------------------------------------------------------------------------------
def synth_state_dict(manifest: Dict[str, Sequence[int]], seed: int = 0, gain: float = 1.0) -> "OrderedDict[str, Tensor]":
    """Deterministic fp32 weights: variance-preserving uniform for conv/linear weights, small biases,
    norm affine near (1, 0).  Same torch build => same bits on every machine."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    for name, shape in manifest.items():
        shape = tuple(int(s) for s in shape)
        is_norm = (".norm" in name or "group_norm" in name or "conv_norm_out" in name or name.startswith("norm")) and len(shape) == 1
        if is_norm:
            if name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = int(math.prod(shape[1:])) if len(shape) > 1 else shape[0]
            a = gain * math.sqrt(3.0 / max(fan_in, 1))
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * a
        sd[name] = t.float().contiguous()
    return sd


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _gn(x: Tensor, sd, p: str, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x: Tensor, sd, p: str, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _linear(x: Tensor, sd, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def resnet_block(x: Tensor, sd, p: str, groups: int, eps: float, temb: Optional[Tensor]) -> Tensor:
    """diffusers ResnetBlock2D (Appendix A.1): GN-SiLU-conv, +time proj, GN-SiLU-conv, (1x1 shortcut), add."""
    h = F.silu(_gn(x, sd, p + ".norm1", groups, eps))
    h = _conv(h, sd, p + ".conv1")
    if temb is not None:
        h = h + _linear(F.silu(temb), sd, p + ".time_emb_proj")[:, :, None, None]
    h = F.silu(_gn(h, sd, p + ".norm2", groups, eps))
    h = _conv(h, sd, p + ".conv2")
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".conv_shortcut", padding=0)
    return x + h


def _attention(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """softmax(q k^T / sqrt(hd)) v in fp32.  q [B,Tq,C], k/v [B,Tk,C]."""
    b, tq, c = q.shape
    hd = c // heads
    qh = q.view(b, tq, heads, hd).transpose(1, 2)
    kh = k.view(b, -1, heads, hd).transpose(1, 2)
    vh = v.view(b, -1, heads, hd).transpose(1, 2)
    w = torch.softmax((qh @ kh.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    return (w @ vh).transpose(1, 2).reshape(b, tq, c)


def transformer_2d(x: Tensor, sd, p: str, heads: int, ctx: Tensor, groups: int) -> Tensor:
    """diffusers Transformer2DModel (use_linear_projection, 1 BasicTransformerBlock, GEGLU FF)."""
    b, c, h, w = x.shape
    res = x
    y = _gn(x, sd, p + ".norm", groups, 1e-6)  # Transformer2DModel.norm eps 1e-6 (Appendix B.5)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = _linear(y, sd, p + ".proj_in")
    bp = p + ".transformer_blocks.0"
    n = F.layer_norm(y, (c,), sd[bp + ".norm1.weight"], sd[bp + ".norm1.bias"], 1e-5)
    a = _attention(_linear(n, sd, bp + ".attn1.to_q"), _linear(n, sd, bp + ".attn1.to_k"), _linear(n, sd, bp + ".attn1.to_v"), heads)
    y = y + _linear(a, sd, bp + ".attn1.to_out.0")
    n = F.layer_norm(y, (c,), sd[bp + ".norm2.weight"], sd[bp + ".norm2.bias"], 1e-5)
    a = _attention(_linear(n, sd, bp + ".attn2.to_q"), _linear(ctx, sd, bp + ".attn2.to_k"), _linear(ctx, sd, bp + ".attn2.to_v"), heads)
    y = y + _linear(a, sd, bp + ".attn2.to_out.0")
    n = F.layer_norm(y, (c,), sd[bp + ".norm3.weight"], sd[bp + ".norm3.bias"], 1e-5)
    hidden, gate = _linear(n, sd, bp + ".ff.net.0.proj").chunk(2, dim=-1)  # value first, gate second (B.8)
    y = y + _linear(hidden * F.gelu(gate), sd, bp + ".ff.net.2")
    y = _linear(y, sd, p + ".proj_out")
    return y.reshape(b, h, w, c).permute(0, 3, 1, 2) + res


def unet_downsample(x: Tensor, sd, p: str) -> Tensor:
    """diffusers Downsample2D as the SD2.1 UNet configures it (downsample_padding 1): Conv3x3 stride 2, SYMMETRIC padding 1 (Appendix B.6)."""
    return _conv(x, sd, p, stride=2, padding=1)


def vae_downsample(h: Tensor, sd, p: str) -> Tensor:
    """diffusers Downsample2D inside DownEncoderBlock2D (padding 0): zero-pad RIGHT / BOTTOM only, then Conv3x3 stride 2 without padding (B.6)."""
    return _conv(F.pad(h, (0, 1, 0, 1)), sd, p, stride=2, padding=0)


def upsample_conv(x: Tensor, sd, p: str, size=None) -> Tensor:
    """diffusers Upsample2D: nearest x2 (or nearest to `size`, the custom UNet's upsample_size rule, custom_unet.py:115-119,377-378), then Conv3x3 pad 1."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest") if size is None else F.interpolate(x, size=tuple(size), mode="nearest")
    return _conv(x, sd, p)


def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos, sin], fp32 (B.9)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    e = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)


def time_embed(sd, cfg: UNetCfg, t: Tensor) -> Tensor:
    te = timestep_embedding(t, cfg.block_out_channels[0])
    te = te.to(sd["time_embedding.linear_1.weight"].dtype)  # custom_unet.py:165-168: fp32 sinusoid cast to the model dtype
    return _linear(F.silu(_linear(te, sd, "time_embedding.linear_1")), sd, "time_embedding.linear_2")


# --------------------------------------------------------------------------------------
# UNet single step (custom_unet.py forward control flow)
# --------------------------------------------------------------------------------------
def unet_forward(sd, cfg: UNetCfg, sample: Tensor, timestep, ctx: Tensor, return_feature: bool = False):
    """Returns (sample or None, multi_level_feats).  custom_unet.py:109-119 upsample_size rule,
    :273 conv_in, :305-327 down, :341-352 mid, :365-400 up (+feats), :402-415 out."""
    b = sample.shape[0]
    g = cfg.norm_num_groups
    n_up = len(cfg.block_out_channels) - 1
    forward_upsample_size = any(d % (2 ** n_up) != 0 for d in sample.shape[-2:])
    t = torch.as_tensor(timestep).reshape(-1).expand(b) if torch.as_tensor(timestep).numel() == 1 else torch.as_tensor(timestep)
    emb = time_embed(sd, cfg, t)
    x = _conv(sample, sd, "conv_in")
    skips = [x]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            x = resnet_block(x, sd, f"down_blocks.{i}.resnets.{j}", g, cfg.norm_eps, emb)
            if cfg.down_has_attn[i]:
                x = transformer_2d(x, sd, f"down_blocks.{i}.attentions.{j}", cfg.num_heads[i], ctx, g)
            skips.append(x)
        if i != nb - 1:
            x = unet_downsample(x, sd, f"down_blocks.{i}.downsamplers.0.conv")
            skips.append(x)
    x = resnet_block(x, sd, "mid_block.resnets.0", g, cfg.norm_eps, emb)
    x = transformer_2d(x, sd, "mid_block.attentions.0", cfg.num_heads[-1], ctx, g)
    x = resnet_block(x, sd, "mid_block.resnets.1", g, cfg.norm_eps, emb)
    feats = []
    for i, blk in enumerate(unet_up_plan(cfg)):
        nres = len(blk["resnets"])
        res = skips[-nres:]
        skips = skips[:-nres]
        upsample_size = None
        if blk["upsample"] and forward_upsample_size:
            upsample_size = skips[-1].shape[2:]
        for j in range(nres):
            x = torch.cat([x, res.pop()], dim=1)  # [hidden, skip], skips consumed from the end (B.7)
            x = resnet_block(x, sd, f"up_blocks.{i}.resnets.{j}", g, cfg.norm_eps, emb)
            if blk["attn"]:
                x = transformer_2d(x, sd, f"up_blocks.{i}.attentions.{j}", blk["heads"], ctx, g)
        if blk["upsample"]:
            x = upsample_conv(x, sd, f"up_blocks.{i}.upsamplers.0.conv", upsample_size)
        feats.append(x)
    if return_feature or not cfg.has_out:
        return None, feats
    x = F.silu(_gn(x, sd, "conv_norm_out", g, cfg.norm_eps))
    return _conv(x, sd, "conv_out"), feats


# --------------------------------------------------------------------------------------
# VAE
# --------------------------------------------------------------------------------------
def _vae_attn_names(sd, p: str):
    if (p + ".to_q.weight") in sd:
        return p + ".group_norm", p + ".to_q", p + ".to_k", p + ".to_v", p + ".to_out.0"
    return p + ".group_norm", p + ".query", p + ".key", p + ".value", p + ".proj_attn"  # older files (A.4)


def vae_mid_attention(x: Tensor, sd, p: str, groups: int, eps: float) -> Tensor:
    b, c, h, w = x.shape
    gn, nq, nk, nv, no = _vae_attn_names(sd, p)
    y = _gn(x, sd, gn, groups, eps).reshape(b, c, h * w).transpose(1, 2)

    def lin(t, name):
        wgt = sd[name + ".weight"].reshape(c, c)
        return F.linear(t, wgt, sd[name + ".bias"])

    a = _attention(lin(y, nq), lin(y, nk), lin(y, nv), 1)
    a = lin(a, no)
    return a.transpose(1, 2).reshape(b, c, h, w) + x


def vae_encode_moments(sd, cfg: VAECfg, x: Tensor) -> Tensor:
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    nb = len(cfg.block_out_channels)
    h = _conv(x, sd, "encoder.conv_in")
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet_block(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", g, eps, None)
        if i != nb - 1:
            h = vae_downsample(h, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv")  # right/bottom only (B.6)
    h = resnet_block(h, sd, "encoder.mid_block.resnets.0", g, eps, None)
    h = vae_mid_attention(h, sd, "encoder.mid_block.attentions.0", g, eps)
    h = resnet_block(h, sd, "encoder.mid_block.resnets.1", g, eps, None)
    h = F.silu(_gn(h, sd, "encoder.conv_norm_out", g, eps))
    h = _conv(h, sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def encode_rgb(sd, cfg: VAECfg, rgb: Tensor) -> Tensor:
    """genpercept_pipeline.py:488-505: mean half of the moments x 0.18215."""
    mean, _ = torch.chunk(vae_encode_moments(sd, cfg, rgb), 2, dim=1)
    return mean * cfg.scaling_factor


def vae_decode(sd, cfg: VAECfg, z: Tensor) -> Tensor:
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    nb = len(cfg.block_out_channels)
    h = _conv(z, sd, "post_quant_conv", padding=0)
    h = _conv(h, sd, "decoder.conv_in")
    h = resnet_block(h, sd, "decoder.mid_block.resnets.0", g, eps, None)
    h = vae_mid_attention(h, sd, "decoder.mid_block.attentions.0", g, eps)
    h = resnet_block(h, sd, "decoder.mid_block.resnets.1", g, eps, None)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = resnet_block(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", g, eps, None)
        if i != nb - 1:
            h = upsample_conv(h, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(_gn(h, sd, "decoder.conv_norm_out", g, eps))
    return _conv(h, sd, "decoder.conv_out")


def decode_pred(sd, cfg: VAECfg, pred_latent: Tensor, mode: str) -> Tensor:
    """genpercept_pipeline.py:507-526."""
    stacked = vae_decode(sd, cfg, pred_latent / cfg.scaling_factor)
    if mode in ("depth", "matting", "dis", "disparity"):
        stacked = stacked.mean(dim=1, keepdim=True)
    return stacked
