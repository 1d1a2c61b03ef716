"""Depth evaluation protocol next to the hot path: least-squares alignment and the ten metrics
`eval.py` reports, restated in numpy (float64) so "AbsRel unchanged" can be stated for predictions of this engine.

Semantics follow /root/reference/src/util/alignment.py:29-94 (align_depth_least_square, depth2disparity) and
/root/reference/src/util/metric.py:34-158; pinned by tests/golden/metrics_ref.npz (outputs of those reference functions).
Per-image masked means, then a mean over images — exactly the reference's reduction order.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def _nearest_downscale(x: np.ndarray, scale: float) -> np.ndarray:
    """What alignment.py:45-55 does to a [H, W] array: torch.nn.Upsample(scale_factor, mode="nearest") applied to the [1, H, W] tensor,
    which torch reads as (batch, channels, length) -- so only the LAST axis is resampled: width floor(W * scale), source column
    floor(dst * (1 / scale)).  Kept as is: the evaluation protocol is defined by what the reference computes."""
    w = x.shape[-1]
    ow = int(np.floor(w * scale))
    ix = np.minimum(np.floor(np.arange(ow) * np.float32(1.0 / scale)).astype(np.int64), w - 1)
    return x[..., ix]


def align_depth_least_square(gt: np.ndarray, pred: np.ndarray, valid_mask: np.ndarray, max_resolution: Optional[int] = None) -> Tuple[np.ndarray, float, float]:
    """min_{s,t} || s * pred + t - gt ||^2 over valid pixels; returns (s * pred + t, s, t).  With max_resolution the fit runs on a
    nearest-downscaled copy (alignment.py:43-55) and the result is applied to the full-resolution prediction."""
    g = np.asarray(gt).squeeze()
    p = np.asarray(pred).squeeze()
    m = np.asarray(valid_mask).squeeze().astype(bool)
    if max_resolution is not None:
        scale = float(np.min(max_resolution / np.array(np.asarray(pred).shape[-2:])))
        if scale < 1:
            g, p, m = _nearest_downscale(g, scale), _nearest_downscale(p, scale), _nearest_downscale(m, scale)
    a = np.stack([p[m], np.ones_like(p[m])], axis=1)
    x = np.linalg.lstsq(a, g[m], rcond=None)[0]
    scale, shift = float(x[0]), float(x[1])
    return np.asarray(pred) * scale + shift, scale, shift


def depth2disparity(depth: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    d = np.zeros_like(depth)
    pos = depth > 0
    d[pos] = 1.0 / depth[pos]
    return d, pos


def disparity2depth(disparity: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    return depth2disparity(disparity)  # alignment.py:93-94


def _masked_mean(x: np.ndarray, mask: Optional[np.ndarray]) -> np.ndarray:
    if mask is None:
        return x.mean(axis=(-1, -2))
    return np.where(mask, x, 0.0).sum(axis=(-1, -2)) / mask.sum(axis=(-1, -2))


def abs_relative_difference(out, tgt, mask=None) -> float:
    return float(_masked_mean(np.abs(out - tgt) / tgt, mask).mean())


def squared_relative_difference(out, tgt, mask=None) -> float:
    return float(_masked_mean(np.abs(out - tgt) ** 2 / tgt, mask).mean())


def rmse_linear(out, tgt, mask=None) -> float:
    return float(np.sqrt(_masked_mean((out - tgt) ** 2, mask)).mean())


def rmse_log(out, tgt, mask=None) -> float:
    return float(np.sqrt(_masked_mean((np.log(out) - np.log(tgt)) ** 2, mask)).mean())


def log10(out, tgt, mask=None) -> float:
    d = np.abs(np.log10(out) - np.log10(tgt))
    return float(d[mask].mean() if mask is not None else d.mean())


def threshold_percentage(out, tgt, thr: float, mask=None) -> float:
    r = np.maximum(out / tgt, tgt / out)
    return float(_masked_mean((r < thr).astype(np.float64), mask).mean())


def delta1_acc(out, tgt, mask=None) -> float:
    return threshold_percentage(out, tgt, 1.25, mask)


def delta2_acc(out, tgt, mask=None) -> float:
    return threshold_percentage(out, tgt, 1.25 ** 2, mask)


def delta3_acc(out, tgt, mask=None) -> float:
    return threshold_percentage(out, tgt, 1.25 ** 3, mask)


def i_rmse(out, tgt, mask=None) -> float:
    return float(np.sqrt(_masked_mean((1.0 / out - 1.0 / tgt) ** 2, mask)).mean())


def silog_rmse(out, tgt, mask=None) -> float:
    diff = np.log(out) - np.log(tgt)
    if mask is not None:
        diff = np.where(mask, diff, 0.0)
        n = mask.sum(axis=(-1, -2))
    else:
        n = diff.shape[-1] * diff.shape[-2]
    first = (diff ** 2).sum(axis=(-1, -2)) / n
    second = diff.sum(axis=(-1, -2)) ** 2 / (n ** 2)
    return float(np.sqrt(np.mean(first - second)) * 100)


METRICS = {f.__name__: f for f in (abs_relative_difference, squared_relative_difference, rmse_linear, rmse_log, log10, delta1_acc, delta2_acc,
                                    delta3_acc, i_rmse, silog_rmse)}


def evaluate_depth(pred: np.ndarray, gt: np.ndarray, valid_mask: np.ndarray, min_depth: float = 1e-3, max_depth: float = 10.0,
                   alignment: str = "least_square", alignment_max_res: Optional[int] = None) -> Dict[str, float]:
    """eval.py:168-215 for one image: LS alignment in depth space ("least_square") or in disparity space ("least_square_disparity"),
    clip to the dataset range and to d > 0, ten metrics."""
    pred = np.asarray(pred)
    gt_a = np.asarray(gt)
    vm = np.asarray(valid_mask).astype(bool)
    if alignment == "least_square":
        aligned, _, _ = align_depth_least_square(gt_a, pred, vm, alignment_max_res)
    elif alignment == "least_square_disparity":
        gt_disp, gt_pos = depth2disparity(gt_a)
        m = vm & gt_pos & (pred > 0)
        disp, _, _ = align_depth_least_square(gt_disp, pred, m, alignment_max_res)
        aligned, _ = disparity2depth(np.clip(disp, 1e-3, None))
    elif alignment in (None, "", "none"):
        aligned = pred
    else:
        raise NotImplementedError(alignment)
    aligned = np.clip(np.clip(aligned, min_depth, max_depth), 1e-6, None)
    a, g, m = (np.asarray(x, dtype=np.float64)[None] for x in (aligned.squeeze(), gt_a.squeeze(), vm.squeeze()))
    m = m.astype(bool)
    return {k: f(a, g, m) for k, f in METRICS.items()}


# ---- surface normals ---------------------------------------------------------------------------------------------------------------------
# The reference has no normal EVALUATION script; the only definition of "angular error" in its tree is angular_loss
# (genpercept/losses/geometry_losses.py:550-590): cosine similarity along the channel axis, clamped to [-1 + eps, 1 - eps] (eps 1e-4),
# acos, mean over the valid pixels.  The evaluator below is that quantity (radians -> degrees) plus the summary statistics normal
# benchmarks report; pinned to the reference function by tests/golden/datasets_ref.npz.
def normal_angular_error(pred: np.ndarray, gt: np.ndarray, valid_mask: Optional[np.ndarray] = None, eps: float = 1e-4) -> Dict[str, float]:
    """pred, gt: [..., 3, H, W] normal maps (any scale: only directions matter); valid_mask: [..., 1 | absent, H, W] or None.
    pred may be the pipeline's [0, 1] encoding: pass decode_normals(pred_np) first."""
    p, g = np.asarray(pred, dtype=np.float64), np.asarray(gt, dtype=np.float64)
    num = (p * g).sum(axis=-3)
    den = np.maximum(np.linalg.norm(p, axis=-3), 1e-8) * np.maximum(np.linalg.norm(g, axis=-3), 1e-8)  # torch.cosine_similarity's eps
    ang = np.arccos(np.clip(num / den, -1.0 + eps, 1.0 - eps))
    if valid_mask is not None:
        m = np.asarray(valid_mask).astype(bool)
        if m.ndim == ang.ndim + 1:
            m = m[..., 0, :, :]
        ang = ang[m]
    else:
        ang = ang.reshape(-1)
    deg = np.degrees(ang)
    return {"mean_rad": float(ang.mean()), "mean_deg": float(deg.mean()), "median_deg": float(np.median(deg)), "rmse_deg": float(np.sqrt((deg ** 2).mean())),
            "within_11.25": float((deg < 11.25).mean()), "within_22.5": float((deg < 22.5).mean()), "within_30": float((deg < 30.0).mean())}


def decode_normals(pred_np: np.ndarray) -> np.ndarray:
    """GenPerceptOutput.pred_np of mode='normal' is [H, W, 3] in [0, 1] = (n + 1) / 2 (genpercept_pipeline.py:469-472): back to [3, H, W] in [-1, 1]."""
    return np.moveaxis(np.asarray(pred_np, dtype=np.float64) * 2.0 - 1.0, -1, -3)
