"""Architecture descriptions of the modules on the one-step path, inferred from diffusers-layout state dicts
(the reference reads them from `unet/config.json` / `vae/config.json` via from_pretrained, run.py:309-320)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)  # SD2.1 `attention_head_dim` == number of heads; head_dim 64
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    has_out: bool = True


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


@dataclass(frozen=True)
class DPTConfig:
    neck_hidden_sizes: Tuple[int, ...] = (320, 640, 1280, 1280)
    fusion_hidden_size: int = 256


def _shape(sd: Dict, key: str):
    if key not in sd:
        raise KeyError(f"state dict has no `{key}` (expected the diffusers key layout)")
    return tuple(sd[key].shape)


def infer_unet_config(sd: Dict) -> UNetConfig:
    c0, cin = _shape(sd, "conv_in.weight")[:2]
    block_out, has_attn = [], []
    i = 0
    while f"down_blocks.{i}.resnets.0.conv1.weight" in sd:
        block_out.append(_shape(sd, f"down_blocks.{i}.resnets.0.conv1.weight")[0])
        has_attn.append(f"down_blocks.{i}.attentions.0.proj_in.weight" in sd)
        i += 1
    if len(block_out) != 4:
        raise ValueError(f"expected 4 UNet down blocks, found {len(block_out)}")
    layers = 0
    while f"down_blocks.0.resnets.{layers}.conv1.weight" in sd:
        layers += 1
    ctx_key = next((k for k in sd if k.endswith("attn2.to_k.weight")), None)
    cross = _shape(sd, ctx_key)[1] if ctx_key else 1024
    has_out = "conv_out.weight" in sd
    out_ch = _shape(sd, "conv_out.weight")[0] if has_out else 4
    for c in block_out:
        if c % 64:
            raise ValueError("UNet widths must be multiples of 64 (attention head_dim 64)")
    return UNetConfig(in_channels=cin, out_channels=out_ch, block_out_channels=tuple(block_out), layers_per_block=layers,
                      num_heads=tuple(c // 64 for c in block_out), down_has_attn=tuple(has_attn), cross_attention_dim=cross, has_out=has_out)


def infer_vae_config(sd: Dict) -> VAEConfig:
    block_out = []
    i = 0
    prefix = "encoder" if "encoder.conv_in.weight" in sd else None
    if prefix:
        while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in sd:
            block_out.append(_shape(sd, f"encoder.down_blocks.{i}.resnets.0.conv1.weight")[0])
            i += 1
        layers = 0
        while f"encoder.down_blocks.0.resnets.{layers}.conv1.weight" in sd:
            layers += 1
        latent = _shape(sd, "encoder.conv_out.weight")[0] // 2
    else:
        while f"decoder.up_blocks.{i}.resnets.0.conv1.weight" in sd:
            block_out.append(_shape(sd, f"decoder.up_blocks.{i}.resnets.0.conv1.weight")[0])
            i += 1
        block_out = block_out[::-1]
        layers = 0
        while f"decoder.up_blocks.0.resnets.{layers}.conv1.weight" in sd:
            layers += 1
        layers -= 1
        latent = _shape(sd, "decoder.conv_in.weight")[1]
    if len(block_out) != 4:
        raise ValueError(f"expected 4 VAE blocks, found {len(block_out)}")
    return VAEConfig(block_out_channels=tuple(block_out), layers_per_block=layers, latent_channels=latent)


def infer_dpt_config(sd: Dict) -> DPTConfig:
    neck = tuple(_shape(sd, f"neck.convs.{i}.weight")[1] for i in range(4))
    return DPTConfig(neck_hidden_sizes=neck, fusion_hidden_size=_shape(sd, "neck.convs.0.weight")[0])
