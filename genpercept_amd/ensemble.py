"""Test-time ensembling of the multi-step archs (reference: genpercept/util/ensemble.py:43-205, called from
genpercept_pipeline.py:290-297 with scale_invariant=True, shift_invariant=True, max_res=50).

Each of the E predictions of one image is affine-invariant, so they are first aligned (per-member scale s and shift t minimising the
summed pairwise RMS distance plus a regulariser pulling the ensembled map to [0,1]; BFGS from the min/max initialisation, at most
`max_iter` steps on maps reduced to `max_res` pixels), then reduced per pixel (median = torch.median, i.e. the LOWER middle value
for even E) and rescaled to [0,1].  The optimiser works on E x (<= 50 x 50) numbers on the host; aligning and reducing the
full-resolution maps are tensor ops wherever `depth` lives.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .image_util import resize_max_res


def _lower_median(a: np.ndarray) -> np.ndarray:
    """torch.median(dim=0) semantics on [E, ...]: element (E-1)//2 of the sorted values (no averaging)."""
    return np.sort(a, axis=0)[(a.shape[0] - 1) // 2]


def _reduce(aligned: torch.Tensor, reduction: str, want_uncertainty: bool):
    if reduction == "mean":
        pred = aligned.mean(dim=0, keepdim=True)
        unc = aligned.std(dim=0, keepdim=True) if want_uncertainty else None
    else:
        pred = aligned.median(dim=0, keepdim=True).values
        unc = (aligned - pred).abs().median(dim=0, keepdim=True).values if want_uncertainty else None
    return pred, unc


def ensemble_depth(depth: torch.Tensor, scale_invariant: bool = True, shift_invariant: bool = True, output_uncertainty: bool = False,
                   reduction: str = "median", regularizer_strength: float = 0.02, max_iter: int = 2, tol: float = 1e-3,
                   max_res: Optional[int] = 1024) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """depth [E,1,H,W] -> ([1,1,H,W] in [0,1], uncertainty [1,1,H,W] or None)."""
    if depth.dim() != 4 or depth.shape[1] != 1:
        raise ValueError(f"Expecting 4D tensor of shape [B,1,H,W]; got {depth.shape}.")
    if reduction not in ("mean", "median"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    if not scale_invariant and shift_invariant:
        raise ValueError("Pure shift-invariant ensembling is not supported.")
    n = depth.shape[0]
    affine = scale_invariant and shift_invariant

    def apply(maps, param):
        p = np.asarray(param, dtype=np.float64)
        s = p[:n]
        if torch.is_tensor(maps):
            st = torch.from_numpy(s).to(maps).view(n, 1, 1, 1)
            return maps * st + torch.from_numpy(p[n:]).to(maps).view(n, 1, 1, 1) if affine else maps * st
        s32 = s.astype(np.float32).reshape(n, 1)  # the reference casts the optimiser's float64 vector to the maps' fp32
        return maps * s32 + p[n:].astype(np.float32).reshape(n, 1) if affine else maps * s32

    if scale_invariant:
        small = depth.to(torch.float32)
        if max_res is not None and max(small.shape[2:]) > max_res:
            small = resize_max_res(small, max_res, "nearest-exact")
        flat = small.reshape(n, -1).cpu().numpy()
        lo, hi = flat.min(axis=1), flat.max(axis=1)
        if affine:
            s0 = 1.0 / np.maximum(hi - lo, np.float32(1e-6))
            x0 = np.concatenate([s0, -s0 * lo])
        else:
            x0 = 1.0 / np.maximum(hi, np.float32(1e-6))
        iu, ju = np.triu_indices(n, k=1)

        def cost(param) -> float:
            al = apply(flat, param)
            c = float(np.sqrt(np.mean((al[iu] - al[ju]) ** 2, axis=1, dtype=np.float32)).astype(np.float64).sum()) if len(iu) else 0.0
            if regularizer_strength > 0:
                pred = al.mean(axis=0, dtype=np.float32) if reduction == "mean" else _lower_median(al)
                c += (abs(0.0 - float(pred.min())) + abs(1.0 - float(pred.max()))) * regularizer_strength
            return c

        import scipy.optimize
        res = scipy.optimize.minimize(cost, x0.astype(np.float32), method="BFGS", tol=tol, options={"maxiter": max_iter, "disp": False})
        depth = apply(depth, res.x)

    pred, unc = _reduce(depth, reduction, output_uncertainty)
    d_max = pred.max()
    if affine:
        d_min = pred.min()
    elif scale_invariant:
        d_min = 0
    else:
        raise ValueError("Unrecognized alignment.")
    rng = (d_max - d_min).clamp(min=1e-6)
    pred = (pred - d_min) / rng
    if output_uncertainty:
        unc = unc / rng
    return pred, unc
