"""Host side of the multi-step archs (`marigold`, `rgb_blending`): the DDIM scheduler as scalars.

Mirror of `DDIMSchedulerCustomized` (/root/reference/src/customized_modules/ddim.py:144-217: beta schedules incl. the reference's
own `scaled_linear_power`, `final_alpha_cumprod`, `_get_variance`) on top of the published update of diffusers' `DDIMScheduler`
(un-vendored dependency, configs stamped 0.29.2: `set_timesteps` spacings, `step` for epsilon / sample / v_prediction with
eta, clip_sample).  The scheduler never touches device memory: one denoising step is the affine map

    pred_x0     = clip(x0_sample * sample + x0_model * model_output)
    pred_eps    = eps_sample * sample + eps_model * model_output
    prev_sample = prev_x0 * pred_x0 + prev_eps * pred_eps

whose seven numbers `step_coefficients()` hands to the engine (`gp_infer_steps`, include/genpercept_hip.h); `step()` applies the same
numbers to torch tensors for callers that drive the UNet themselves (genpercept_pipeline.py:455-463).
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch

_DEFAULTS = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, power_beta_curve=1.0)


def _rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """ddim.py:25-58 (arXiv 2305.08891, algorithm 1)."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


def _cosine_betas(n: int, max_beta: float = 0.999) -> torch.Tensor:
    """diffusers `betas_for_alpha_bar` (squaredcos_cap_v2)."""
    bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


class StepOutput(SimpleNamespace):
    """`.prev_sample`, `.pred_original_sample` (diffusers DDIMSchedulerOutput)."""


class DDIMSchedulerCustomized:
    order = 1

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS) - {"_class_name", "_diffusers_version", "skip_prk_steps"}
        if unknown:
            raise TypeError(f"unexpected scheduler arguments: {sorted(unknown)}")
        cfg = dict(_DEFAULTS)
        cfg.update({k: v for k, v in kwargs.items() if k in _DEFAULTS})
        self.config = SimpleNamespace(**cfg)
        n, b0, b1, sched = cfg["num_train_timesteps"], cfg["beta_start"], cfg["beta_end"], cfg["beta_schedule"]
        if cfg["trained_betas"] is not None:
            self.betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
        elif sched == "linear":
            self.betas = torch.linspace(b0, b1, n, dtype=torch.float32)
        elif sched == "scaled_linear":
            self.betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, n, dtype=torch.float32) ** 2
        elif sched == "scaled_linear_power":  # ddim.py:173-175
            p = cfg["power_beta_curve"]
            self.betas = torch.linspace(b0 ** (1 / p), b1 ** (1 / p), n, dtype=torch.float32) ** p
            self.power_beta_curve = p
        elif sched == "squaredcos_cap_v2":
            self.betas = _cosine_betas(n)
        else:
            raise NotImplementedError(f"{sched} does is not implemented for {self.__class__}")
        if cfg["rescale_betas_zero_snr"]:
            self.betas = _rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy().astype(np.int64))
        self.beta_schedule = sched

    # ---- construction from a diffusers scheduler directory (run.py:363,371) -----------------------------------------
    @classmethod
    def from_config(cls, config) -> "DDIMSchedulerCustomized":
        if not isinstance(config, dict):
            config = {k: getattr(config, k) for k in _DEFAULTS if hasattr(config, k)}
        return cls(**{k: v for k, v in config.items() if k in _DEFAULTS})

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **_ignored) -> "DDIMSchedulerCustomized":
        d = os.path.join(path, subfolder) if subfolder else path
        f = os.path.join(d, "scheduler_config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"no scheduler_config.json under {d}")
        with open(f) as fh:
            return cls.from_config(json.load(fh))

    # ---- diffusers DDIMScheduler surface -----------------------------------------------------------------------------
    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`: {n}")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            ts = ts + self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(f"{sp} is not supported. Please make sure to choose one of 'leading' or 'trailing'.")
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def _prev_timestep(self, timestep: int) -> int:
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        return int(timestep) - self.config.num_train_timesteps // self.num_inference_steps

    def _get_variance(self, timestep: int, prev_timestep: int) -> torch.Tensor:
        """ddim.py:204-217 (the customised form: product of the alphas between the two timesteps)."""
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        between = torch.prod(self.alphas[(prev_timestep + 1):(timestep + 1)])
        return ((1 - a_prev) / (1 - a_t)) * (1 - between)

    def step_coefficients(self, timestep: int, eta: float = 0.0, use_clipped_model_output: bool = False) -> dict:
        """The affine form of one `step` (module docstring); fp32 arithmetic like the tensors diffusers multiplies with."""
        if self.config.thresholding:
            raise NotImplementedError("dynamic thresholding is a per-sample quantile, not an affine step")
        t = int(timestep)
        prev = self._prev_timestep(t)
        a_t = self.alphas_cumprod[t].to(torch.float32)
        a_prev = (self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod).to(torch.float32)
        sa, sb = float(a_t ** 0.5), float((1 - a_t) ** 0.5)
        kind = self.config.prediction_type
        if kind == "epsilon":
            c = dict(x0_sample=1.0 / sa, x0_model=-sb / sa, eps_sample=0.0, eps_model=1.0)
        elif kind == "sample":
            c = dict(x0_sample=0.0, x0_model=1.0, eps_sample=1.0 / sb, eps_model=-sa / sb)
        elif kind == "v_prediction":
            c = dict(x0_sample=sa, x0_model=-sb, eps_sample=sb, eps_model=sa)
        else:
            raise ValueError(f"prediction_type given as {kind} must be one of `epsilon`, `sample`, or `v_prediction`")
        std = float(eta * self._get_variance(t, prev) ** 0.5) if eta else 0.0
        c.update(timestep=float(t), prev_x0=float(a_prev ** 0.5), prev_eps=float((1 - a_prev - std ** 2) ** 0.5), std=std,
                 clip=float(self.config.clip_sample_range) if self.config.clip_sample else 0.0,
                 # use_clipped_model_output: eps re-derived from the CLIPPED x0 -> eps = (sample - sa * x0) / sb
                 eps_from_x0=bool(use_clipped_model_output), sqrt_alpha=sa, sqrt_beta=sb)
        return c

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise: Optional[torch.Tensor] = None, return_dict: bool = True):
        c = self.step_coefficients(int(timestep), eta, use_clipped_model_output)
        x0 = c["x0_sample"] * sample + c["x0_model"] * model_output
        eps = c["eps_sample"] * sample + c["eps_model"] * model_output
        if c["clip"] > 0:
            x0 = x0.clamp(-c["clip"], c["clip"])
        if use_clipped_model_output:
            eps = (sample - c["sqrt_alpha"] * x0) / c["sqrt_beta"]
        prev = c["prev_x0"] * x0 + c["prev_eps"] * eps
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + c["std"] * variance_noise
        if not return_dict:
            return (prev,)
        return StepOutput(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        return (acp[timesteps] ** 0.5).reshape(shape) * original_samples + ((1 - acp[timesteps]) ** 0.5).reshape(shape) * noise

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        acp = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        shape = (-1,) + (1,) * (sample.dim() - 1)
        return (acp[timesteps] ** 0.5).reshape(shape) * noise - ((1 - acp[timesteps]) ** 0.5).reshape(shape) * sample

    def __len__(self) -> int:
        return self.config.num_train_timesteps

    # ---- what the engine consumes --------------------------------------------------------------------------------------
    def plan(self, num_inference_steps: int, fix_timesteps: Optional[int] = None) -> List[dict]:
        """Coefficients of every step of single_infer's loop (genpercept_pipeline.py:403-409,447-463)."""
        self.set_timesteps(num_inference_steps)
        ts: Sequence[int] = [int(fix_timesteps)] * len(self.timesteps) if fix_timesteps else [int(t) for t in self.timesteps]
        return [self.step_coefficients(t) for t in ts]
