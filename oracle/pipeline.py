"""CPU oracle (TEST INFRASTRUCTURE): fp32 restatement of GenPerceptPipeline.single_infer and the
scheduler identity it relies on.

Follows /root/reference/genpercept/genpercept_pipeline.py:
  :245-247 rgb normalisation, :399-486 single_infer (both head branches), :488-505 encode_rgb,
  :507-526 decode_pred, :469-472 clip/shift, :480-482 DPT min-max;
/root/reference/src/customized_modules/ddim.py:166-204 + hf_configs/scheduler_beta_1.0_1.0/scheduler_config.json
for the scheduler constants (beta == 1 => alphas_cumprod == 0 => v-prediction x0 = -model_output; leading spacing
with steps_offset 1 and one step => timesteps == [1]); second statement of the same identity:
GenPercept_v1/genpercept/pipeline_genpercept.py:284,301.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import dpt as odpt
from . import sd21 as osd

Tensor = torch.Tensor


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000, steps_offset: int = 1) -> np.ndarray:
    """diffusers DDIMScheduler.set_timesteps, timestep_spacing == 'leading'."""
    step_ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
    return ts + steps_offset


def ddim_pred_original_sample(model_output: Tensor, sample: Tensor, t: int, beta_start: float = 1.0, beta_end: float = 1.0,
                              num_train_timesteps: int = 1000) -> Tensor:
    """diffusers DDIMScheduler.step for prediction_type == 'v_prediction', beta_schedule 'scaled_linear':
    pred_x0 = sqrt(acp_t) * sample - sqrt(1 - acp_t) * v.  With beta == 1 every acp_t == 0 => -v."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    a = acp[t]
    return (a ** 0.5) * sample - ((1 - a) ** 0.5) * model_output


def normalize_rgb(rgb_u8: Tensor) -> Tensor:
    """genpercept_pipeline.py:245: [0,255] -> [-1,1]."""
    return rgb_u8.float() / 255.0 * 2.0 - 1.0


def single_infer(vae_sd, vae_cfg: osd.VAECfg, unet_sd, unet_cfg: osd.UNetCfg, rgb_norm: Tensor, ctx: Tensor, mode: str,
                 timestep: int = 1, dpt_sd: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """rgb_norm [B,3,H,W] in [-1,1]; ctx [L,D] or [1,L,D]; returns [B,C,H,W] in [0,1] (VAE head) or
    min-max normalised [B,1,H,W] (DPT head; min/max PER IMAGE, which is what the reference computes because it
    only ever sees B == 1)."""
    b = rgb_norm.shape[0]
    ctx = ctx.reshape(1, -1, ctx.shape[-1]).to(rgb_norm.dtype).expand(b, -1, -1)
    latent = osd.encode_rgb(vae_sd, vae_cfg, rgb_norm)
    if dpt_sd is None:
        v, _ = osd.unet_forward(unet_sd, unet_cfg, latent, timestep, ctx)
        pred_latent = -v  # == ddim_pred_original_sample(v, latent, t) for beta == 1 (tested)
        pred = osd.decode_pred(vae_sd, vae_cfg, pred_latent, mode)
        return (torch.clip(pred, -1.0, 1.0) + 1.0) / 2.0
    _, feats = osd.unet_forward(unet_sd, unet_cfg, latent, timestep, ctx, return_feature=True)
    pred = odpt.dpt_head_forward(dpt_sd, feats[::-1])[:, None]
    mn = pred.amin(dim=(1, 2, 3), keepdim=True)
    mx = pred.amax(dim=(1, 2, 3), keepdim=True)
    return (pred - mn) / (mx - mn)
