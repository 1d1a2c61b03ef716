"""CPU oracle (TEST INFRASTRUCTURE): fp32 restatement of the DPT neck+head GenPercept puts on the
UNet up-block features (config 4, disparity).

Follows /root/reference/genpercept/models/dpt_head.py:
  Upsample2D.forward :160-210 (nearest x2 + conv3x3)           -> feature_upsample_0 (:426,534)
  DPTNeck.forward :367-388 (4x conv3x3 -> 256, no bias)        -> neck.convs
  DPTFeatureFusionStage/Layer :274-335, DPTPreActResidualLayer :213-271
  DPTDepthEstimationHeadIdentity :564-582 (+ projection :66-67,82-84)
and hf_configs/dpt-sd2.1-unet-after-upsample-general/config.json (neck sizes, fusion 256, no BN, no bias
in the residual units, add_projection, head_in_index -1).

This restatement IS pinned: tests/golden/dpt_head_ref.npz holds outputs of the reference's own class run in the
build container (tests/golden/make_goldens.py), and tests/test_oracle.py checks this file against them.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class DPTCfg:
    neck_hidden_sizes: Tuple[int, ...] = (320, 640, 1280, 1280)
    fusion_hidden_size: int = 256

    @staticmethod
    def tiny() -> "DPTCfg":
        return DPTCfg(neck_hidden_sizes=(64, 128, 256, 256), fusion_hidden_size=128)


def dpt_manifest(cfg: DPTCfg = DPTCfg()) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    f = cfg.fusion_hidden_size
    c0 = cfg.neck_hidden_sizes[0]
    m["feature_upsample_0.conv.weight"] = (c0, c0, 3, 3)
    m["feature_upsample_0.conv.bias"] = (c0,)
    for i, c in enumerate(cfg.neck_hidden_sizes):
        m[f"neck.convs.{i}.weight"] = (f, c, 3, 3)
    for i in range(len(cfg.neck_hidden_sizes)):
        p = f"neck.fusion_stage.layers.{i}"
        m[p + ".projection.weight"] = (f, f, 1, 1)
        m[p + ".projection.bias"] = (f,)
        if i != 0:
            m[p + ".residual_layer1.convolution1.weight"] = (f, f, 3, 3)
            m[p + ".residual_layer1.convolution2.weight"] = (f, f, 3, 3)
        m[p + ".residual_layer2.convolution1.weight"] = (f, f, 3, 3)
        m[p + ".residual_layer2.convolution2.weight"] = (f, f, 3, 3)
    m["head.projection.weight"] = (f, f, 3, 3)
    m["head.projection.bias"] = (f,)
    m["head.head.0.weight"] = (f // 2, f, 3, 3)
    m["head.head.0.bias"] = (f // 2,)
    m["head.head.2.weight"] = (32, f // 2, 3, 3)
    m["head.head.2.bias"] = (32,)
    m["head.head.4.weight"] = (1, 32, 1, 1)
    m["head.head.4.bias"] = (1,)
    return m


def _rcu(x: Tensor, sd, p: str) -> Tensor:
    """Pre-activation residual conv unit (dpt_head.py:256-271), bias-free, no BN."""
    h = F.conv2d(F.relu(x), sd[p + ".convolution1.weight"], None, padding=1)
    h = F.conv2d(F.relu(h), sd[p + ".convolution2.weight"], None, padding=1)
    return h + x


def dpt_head_forward(sd: Dict[str, Tensor], feats: Sequence[Tensor]) -> Tensor:
    """feats = multi_level_feats[::-1] = [c0@h, c1@h, c2@h/2, c3@h/4] -> [B, 8h, 8w] (Identity head)."""
    assert len(feats) == 4
    f0 = F.interpolate(feats[0], scale_factor=2.0, mode="nearest")
    f0 = F.conv2d(f0, sd["feature_upsample_0.conv.weight"], sd["feature_upsample_0.conv.bias"], padding=1)
    hs = [f0, feats[1], feats[2], feats[3]]
    hs = [F.conv2d(h, sd[f"neck.convs.{i}.weight"], None, padding=1) for i, h in enumerate(hs)]
    hs = hs[::-1]  # fusion runs from the coarsest map (dpt_head.py:322-335)
    fused = None
    for i, h in enumerate(hs):
        p = f"neck.fusion_stage.layers.{i}"
        if i == 0:
            x = h
        else:
            r = h
            if fused.shape != r.shape:
                r = F.interpolate(r, size=fused.shape[2:], mode="bilinear", align_corners=False)
            x = fused + _rcu(r, sd, p + ".residual_layer1")
        x = _rcu(x, sd, p + ".residual_layer2")
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        fused = F.conv2d(x, sd[p + ".projection.weight"], sd[p + ".projection.bias"])
    x = F.relu(F.conv2d(fused, sd["head.projection.weight"], sd["head.projection.bias"], padding=1))
    x = F.conv2d(x, sd["head.head.0.weight"], sd["head.head.0.bias"], padding=1)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.relu(F.conv2d(x, sd["head.head.2.weight"], sd["head.head.2.bias"], padding=1))
    x = F.conv2d(x, sd["head.head.4.weight"], sd["head.head.4.bias"])
    return x.squeeze(1)
