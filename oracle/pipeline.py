"""CPU oracle (TEST INFRASTRUCTURE): fp32 restatement of GenPerceptPipeline.single_infer and the
scheduler identity it relies on.

Follows /root/reference/genpercept/genpercept_pipeline.py:
  :245-247 rgb normalisation, :399-486 single_infer (both head branches), :488-505 encode_rgb,
  :507-526 decode_pred, :469-472 clip/shift, :480-482 DPT min-max;
/root/reference/src/customized_modules/ddim.py:166-204 + hf_configs/scheduler_beta_1.0_1.0/scheduler_config.json
for the scheduler constants (beta == 1 => alphas_cumprod == 0 => v-prediction x0 = -model_output; leading spacing
with steps_offset 1 and one step => timesteps == [1]); second statement of the same identity:
GenPercept_v1/genpercept/pipeline_genpercept.py:284,301.

Multi-step archs (marigold / rgb_blending; genpercept_pipeline.py:413-422,447-465, run.py:59-78,361-368): `DDIM` below restates the
reference's DDIMSchedulerCustomized.__init__ / _get_variance (ddim.py:144-217; pinned to the reference class itself by
tests/golden/scheduler_ref.npz) and diffusers' DDIMScheduler.set_timesteps / step (un-vendored, configs stamped 0.29.2; the published
DDIM update, arXiv 2010.02502 eq. 12, in diffusers' variable names -- PARITY UNPINNED for that method, nothing in the reference's tree
holds a vector for it); `multi_step_infer` is the denoising loop.
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


class DDIM:
    """fp32 tensors throughout, like the reference's scheduler."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", clip_sample_range=1.0, timestep_spacing="leading",
                 power_beta_curve=1.0, **_unused):
        n = num_train_timesteps
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
        elif beta_schedule == "scaled_linear_power":
            self.betas = torch.linspace(beta_start ** (1 / power_beta_curve), beta_end ** (1 / power_beta_curve), n, dtype=torch.float32) ** power_beta_curve
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.n, self.steps_offset, self.spacing = n, steps_offset, timestep_spacing
        self.prediction_type, self.clip_sample, self.clip_range = prediction_type, clip_sample, clip_sample_range
        self.num_inference_steps = None

    def set_timesteps(self, k: int):
        self.num_inference_steps = k
        if self.spacing == "leading":
            self.timesteps = ddim_timesteps(k, self.n, self.steps_offset)
        elif self.spacing == "trailing":
            self.timesteps = np.round(np.arange(self.n, 0, -self.n / k)).astype(np.int64) - 1
        elif self.spacing == "linspace":
            self.timesteps = np.linspace(0, self.n - 1, k).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(self.spacing)
        return self.timesteps

    def get_variance(self, t: int, prev: int) -> Tensor:
        """ddim.py:204-217."""
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - torch.prod(self.alphas[prev + 1:t + 1]))

    def step(self, model_output: Tensor, t: int, sample: Tensor):
        """eta = 0.  Returns (prev_sample, pred_original_sample)."""
        prev = t - self.n // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(self.prediction_type)
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        direction = (1 - a_prev) ** 0.5 * eps
        return a_prev ** 0.5 * x0 + direction, x0


def replace_unet_conv_in(unet_sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """run.py:59-78: a 4-channel conv_in becomes the 8-channel one of the marigold arch (weights repeated along the input axis, halved)."""
    sd = dict(unet_sd)
    sd["conv_in.weight"] = sd["conv_in.weight"].repeat(1, 2, 1, 1) * 0.5
    return sd


def multi_step_infer(vae_sd, vae_cfg: osd.VAECfg, unet_sd, unet_cfg: osd.UNetCfg, rgb_norm: Tensor, ctx: Tensor, mode: str, sched: DDIM,
                     num_inference_steps: int, noise: Optional[Tensor] = None, fix_timesteps: Optional[int] = None) -> Tensor:
    """single_infer without a customised head (genpercept_pipeline.py:399-472).  noise given -> marigold (UNet input [rgb_latent,
    pred_latent], 8 channels); None -> rgb_blending (pred_latent starts as the rgb latent and is the UNet input)."""
    b = rgb_norm.shape[0]
    ctx = ctx.reshape(1, -1, ctx.shape[-1]).to(rgb_norm.dtype).expand(b, -1, -1)
    ts = sched.set_timesteps(num_inference_steps)
    if fix_timesteps:
        ts = np.full_like(ts, int(fix_timesteps))
    rgb_latent = osd.encode_rgb(vae_sd, vae_cfg, rgb_norm)
    pred_latent = noise if noise is not None else rgb_latent
    x0 = None
    for t in ts:
        unet_input = torch.cat([rgb_latent, pred_latent], dim=1) if noise is not None else pred_latent
        v, _ = osd.unet_forward(unet_sd, unet_cfg, unet_input, int(t), ctx)
        pred_latent, x0 = sched.step(v, int(t), pred_latent)
    pred = osd.decode_pred(vae_sd, vae_cfg, x0, mode)
    return (torch.clip(pred, -1.0, 1.0) + 1.0) / 2.0
