"""Host-side pre/post processing of GenPerceptPipeline.__call__ (reference: genpercept/util/image_util.py:25-126 and
genpercept_pipeline.py:222-247,301-337).  torchvision is not a dependency: the same ATen resize kernels are reached
through torch.nn.functional.interpolate."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

_RESAMPLE = {"bilinear": "bilinear", "bicubic": "bicubic", "nearest": "nearest-exact", "nearest-exact": "nearest-exact"}


def get_resample_method(method_str: str) -> str:
    """image_util.py:108-126: unknown names raise ValueError; 'nearest' means NEAREST_EXACT."""
    m = _RESAMPLE.get(method_str)
    if m is None:
        raise ValueError(f"Unknown resampling method: {method_str}")
    return m


def _interp(img: torch.Tensor, size, mode: str) -> torch.Tensor:
    if mode == "nearest-exact":
        return F.interpolate(img, size=size, mode=mode)
    return F.interpolate(img, size=size, mode=mode, align_corners=False, antialias=True)


def resize_to(img: torch.Tensor, size, mode: str) -> torch.Tensor:
    """torchvision.transforms.functional.resize(img, size, mode, antialias=True) for [B,C,H,W] tensors.  Integer images
    are interpolated in fp32 and rounded back to their dtype (torchvision's _cast_squeeze_out), which is what the
    reference does to the uint8 RGB before normalisation (Appendix B.11)."""
    size = (int(size[0]), int(size[1]))
    if tuple(img.shape[-2:]) == size:
        return img
    if img.dtype == torch.uint8:
        out = _interp(img.float(), size, mode)
        return out.round().clamp(0, 255).to(torch.uint8)
    return _interp(img, size, mode)


def resize_max_res_size(h: int, w: int, max_edge_resolution: int):
    """The size rule of resize_max_res alone (image_util.py:95-100): keep aspect ratio, longest edge -> max_edge_resolution, int() truncation."""
    f = min(max_edge_resolution / w, max_edge_resolution / h)
    return int(h * f), int(w * f)


def resize_max_res(img: torch.Tensor, max_edge_resolution: int, resample_method: str = "bilinear") -> torch.Tensor:
    """image_util.py:75-105: keep aspect ratio, longest edge -> max_edge_resolution, int() truncation of the new size."""
    assert img.dim() == 4, f"Invalid input shape {img.shape}"
    h, w = img.shape[-2:]
    return resize_to(img, resize_max_res_size(int(h), int(w), max_edge_resolution), resample_method)


def colorize_depth_maps(depth_map: np.ndarray, min_depth: float, max_depth: float, cmap: str = "Spectral") -> np.ndarray:
    """image_util.py:25-63 (numpy branch): [ (B,) H, W ] -> [B, 3, H, W] floats in [0,1]."""
    import matplotlib

    depth = np.asarray(depth_map).copy().squeeze()
    if depth.ndim < 3:
        depth = depth[np.newaxis]
    cm = matplotlib.colormaps[cmap]
    depth = ((depth - min_depth) / (max_depth - min_depth)).clip(0, 1)
    img = cm(depth, bytes=False)[:, :, :, 0:3]
    return np.rollaxis(img, 3, 1)


def chw2hwc(chw: np.ndarray) -> np.ndarray:
    assert chw.ndim == 3
    return np.moveaxis(chw, 0, -1)
