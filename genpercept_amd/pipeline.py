"""GenPerceptPipeline: host-side mirror of the reference's `genpercept.GenPerceptPipeline`
(/root/reference/genpercept/genpercept_pipeline.py:64-526): the one-step `archs=genpercept` path and the multi-step
`marigold` / `rgb_blending` archs (run.py:361-368).

Same constructor keywords, same `__call__` keywords, same assertions and exceptions, same `GenPerceptOutput`
(`pred_np`, `pred_colored`), so `run.py:420-432` / `infer.py:417-430` style drivers work unchanged.  Differences, all
behind the same surface:
  * `unet` / `vae` / `customized_head` may be anything exposing `state_dict()` (real diffusers modules included), a plain
    dict of tensors in the diffusers key layout, or a directory holding a diffusers safetensors/bin checkpoint — `diffusers`
    itself is never imported;
  * `scheduler` is reduced to scalars on the host (`genpercept_amd.scheduler.DDIMSchedulerCustomized`, built from any object / dict /
    directory carrying a DDIM scheduler config).  beta_start == beta_end == 1 with v_prediction makes DDIM's pred_original_sample equal
    to -model_output (genpercept_pipeline.py:465; src/customized_modules/ddim.py:166-204): that case is `gp_infer`; every other
    scheduler, and the multi-step archs, run the denoising loop inside the engine (`gp_infer_steps`);
  * `text_encoder` may be a precomputed [L, D] embedding of the prompt (e.g. the v1 `empty_text_embed.npy`);
  * all tensor math runs in libgenpercept_hip.so on one MI355X; there is no PyTorch/CPU fallback.
New: `infer_batch` for a list of images / a [B,3,H,W] tensor (the reference is one image per call).
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from PIL import Image

from . import config as gcfg
from .image_util import chw2hwc, colorize_depth_maps, get_resample_method, resize_max_res, resize_to
from .batchsize import find_batch_size
from .weights import merge_lora_state_dict


@dataclass
class GenPerceptOutput:
    """genpercept_pipeline.py:50-62.  pred_np: [H,W] (1-channel modes) or [H,W,3], values in [0,1]; pred_colored: PIL image."""
    pred_np: np.ndarray
    pred_colored: Union[None, Image.Image]


def _load_file(f: str) -> Dict[str, torch.Tensor]:
    if f.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(f)
    return torch.load(f, map_location="cpu", weights_only=True)


def _load_checkpoint_dir(path: str) -> Dict[str, torch.Tensor]:
    """diffusers on-disk layouts consumed by run.py:296-333."""
    cands = ["diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin"]
    if os.path.isfile(path):
        files = [path]
    else:
        files = [os.path.join(path, c) for c in cands if os.path.exists(os.path.join(path, c))]
        if not files and os.path.isdir(os.path.join(path, "unet")):
            return _load_checkpoint_dir(os.path.join(path, "unet"))
    if not files:
        raise FileNotFoundError(f"no diffusers checkpoint under {path}")
    return _load_file(files[0])


def compose_finetuned_vae(base_vae, ckpt_dir: str) -> Dict[str, torch.Tensor]:
    """run.py:308-312: a fine-tuned decoder is saved RELATIVE to its sub-module -- `<ckpt>/vae_decoder/model.safetensors` holds
    `conv_in.weight` ... (no `decoder.` prefix), `<ckpt>/vae_post_quant_conv/model.safetensors` holds `weight`, `bias` -- and is loaded
    over the base VAE's decoder / post_quant_conv.  Returns the full VAE state dict (encoder and quant_conv from `base_vae`)."""
    sd = dict(_state_dict_of(base_vae))
    dec = _load_checkpoint_dir(os.path.join(ckpt_dir, "vae_decoder"))
    pq = _load_checkpoint_dir(os.path.join(ckpt_dir, "vae_post_quant_conv"))
    missing = [k for k in sd if k.startswith("decoder.") and k[len("decoder."):] not in dec]
    extra = [k for k in dec if "decoder." + k not in sd]
    if missing or extra:
        raise KeyError(f"vae_decoder checkpoint does not match the base VAE's decoder (missing {missing[:3]}, unexpected {extra[:3]})")
    for k, v in dec.items():
        sd["decoder." + k] = v
    for k, v in pq.items():
        if "post_quant_conv." + k not in sd:
            raise KeyError(f"unexpected key {k} in vae_post_quant_conv")
        sd["post_quant_conv." + k] = v
    return sd


# class names of the reference's two DPT heads (genpercept/models/dpt_head.py:391, 585): identical parameter sets, but only the
# ...Identity one is accepted by single_infer (genpercept_pipeline.py:474, 483-484); the other ends in a ReLU (dpt_head.py:69-76)
_HEAD_IDENTITY = "DPTNeckHeadForUnetAfterUpsampleIdentity"
_HEAD_RELU = "DPTNeckHeadForUnetAfterUpsample"


def _head_kind(obj, declared: Optional[str]) -> Optional[str]:
    """'identity' / 'relu' / None (unknown) for a customized_head argument: explicit `head_type`, else the module's class name, else the
    checkpoint directory name (run.py:296-307 loads `dpt_head_identity/` into the Identity class and `dpt_head/` into the other)."""
    if declared is not None:
        if declared not in ("identity", "relu"):
            raise ValueError(f"head_type must be 'identity' or 'relu', got {declared!r}")
        return declared
    name = type(obj).__name__
    if name == _HEAD_IDENTITY:
        return "identity"
    if name == _HEAD_RELU:
        return "relu"
    if isinstance(obj, (str, os.PathLike)):
        parts = [p for p in os.path.normpath(str(obj)).split(os.sep) if p]
        if "dpt_head_identity" in parts:
            return "identity"
        if "dpt_head" in parts:
            return "relu"
    return None


def _state_dict_of(obj) -> Optional[Dict[str, torch.Tensor]]:
    """Weights of a module-like object, with PEFT LoRA adapters (run.py:345-357) folded into their base layers."""
    if obj is None:
        return None
    if isinstance(obj, dict):
        return merge_lora_state_dict(obj)
    if isinstance(obj, (str, os.PathLike)):
        return merge_lora_state_dict(_load_checkpoint_dir(str(obj)))
    if hasattr(obj, "state_dict"):
        return merge_lora_state_dict(obj.state_dict())
    raise TypeError(f"cannot take weights from {type(obj)}: expected state_dict(), a dict of tensors or a checkpoint directory")


def _sched_attr(s, name, default=None):
    if hasattr(s, name):
        return getattr(s, name)
    cfg = getattr(s, "config", None)
    if cfg is not None:
        if isinstance(cfg, dict):
            return cfg.get(name, default)
        return getattr(cfg, name, default)
    if isinstance(s, dict):
        return s.get(name, default)
    return default


def _as_scheduler(s):
    """Anything carrying a DDIM scheduler config -> the host-side scheduler (None stays None)."""
    from .scheduler import DDIMSchedulerCustomized, _DEFAULTS
    if s is None or isinstance(s, DDIMSchedulerCustomized):
        return s
    if "LCM" in type(s).__name__ or "LCM" in str(_sched_attr(s, "_class_name", "")):
        raise NotImplementedError("LCMScheduler checkpoints are not supported (the reference's run.py only builds DDIMSchedulerCustomized)")
    if isinstance(s, (str, os.PathLike)):
        d = str(s)
        return DDIMSchedulerCustomized.from_pretrained(d, subfolder="scheduler" if os.path.isdir(os.path.join(d, "scheduler")) else None)
    cfg = {k: _sched_attr(s, k) for k in _DEFAULTS if _sched_attr(s, k, None) is not None}
    if "beta_start" not in cfg or "beta_end" not in cfg:
        raise TypeError(f"cannot read a DDIM scheduler config from {type(s)}")
    return DDIMSchedulerCustomized(**cfg)


class GenPerceptPipeline:
    latent_scale_factor = 0.18215  # genpercept_pipeline.py:96

    def __init__(self, unet, vae, scheduler=None, text_encoder=None, tokenizer=None, default_denoising_steps: Optional[int] = 10,
                 default_processing_resolution: Optional[int] = 768, rgb_blending=False, customized_head=None, genpercept_pipeline=True,
                 device: Union[str, int, torch.device, None] = None, torch_dtype: Optional[torch.dtype] = None, head_type: Optional[str] = None):
        self.genpercept_pipeline = genpercept_pipeline
        if genpercept_pipeline:  # genpercept_pipeline.py:115-117
            default_denoising_steps = 1
            rgb_blending = True
        self.scheduler = _as_scheduler(scheduler)
        if self.scheduler is None and not genpercept_pipeline:
            raise ValueError("the multi-step archs (genpercept_pipeline=False) need a scheduler")
        # beta == 1 everywhere with v_prediction: alphas_cumprod == 0, so one step's pred_original_sample is -model_output for any t
        c = self.scheduler.config if self.scheduler is not None else None
        self._x0_is_neg_v = c is None or (c.beta_start == 1 and c.beta_end == 1 and c.prediction_type == "v_prediction"
                                          and c.trained_betas is None and not c.rescale_betas_zero_snr and not c.clip_sample and not c.thresholding)
        if customized_head is not None:  # genpercept_pipeline.py:141-143
            assert rgb_blending and self.scheduler is not None and c.beta_start == 1 and c.beta_end == 1
            assert genpercept_pipeline
        self._unet_src, self._vae_src, self._head_src = unet, vae, customized_head
        self._head_kind = _head_kind(customized_head, head_type) if customized_head is not None else None
        if self._head_kind == "relu":  # genpercept_pipeline.py:474,483-484: only the ...Identity head is a valid customized_head
            raise ValueError("unsupported customized_head: DPTNeckHeadForUnetAfterUpsample (ReLU-terminated); "
                             "single_infer accepts DPTNeckHeadForUnetAfterUpsampleIdentity only (genpercept_pipeline.py:483-484)")
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.default_denoising_steps = default_denoising_steps
        self.default_processing_resolution = default_processing_resolution
        self.rgb_blending = rgb_blending
        self.customized_head = customized_head
        self.text_embed: Optional[torch.Tensor] = None
        self._embed_prompt = None
        if text_encoder is not None and not hasattr(text_encoder, "forward") and tokenizer is None:
            self.text_embed = torch.as_tensor(np.asarray(text_encoder, dtype=np.float32) if not torch.is_tensor(text_encoder) else text_encoder).float()
            self.text_embed = self.text_embed.reshape(1, -1, self.text_embed.shape[-1])
            self.text_encoder = None
        self._engine = None
        # precision of the engine: bf16 (default; BASELINE.json's dtype), fp16 (the reference's --half_precision) or -- for torch_dtype=float32,
        # run.py's default -- the contract precision "fp32c": fp32 storage + split-bf16 matrix products, final maps within 1e-3 of the fp32
        # path under both readings
        from .engine import precision_of
        self._precision = precision_of(torch_dtype)
        # dtype single_infer draws marigold's initial noise in (genpercept_pipeline.py:416-420: torch.randn(..., dtype=self.dtype)): the dtype the
        # caller asked for (run.py:273-281: fp32 unless --half_precision), the engine's element type when none was given
        self._noise_dtype = torch_dtype if torch_dtype is not None else torch.float32  # (diffusers loads fp32 modules when no torch_dtype is given)
        self._timestep = None
        self._device = torch.device("cuda", 0) if device is None else torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.mode = None
        self.last_saturation_events = 0  # fp16 engine: saturation events of the last __call__ (see _warn_if_saturated)

    # ---- diffusers-like conveniences ---------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, checkpoint: str, variant=None, torch_dtype=None, **kwargs):
        """run.py:370-376 call shape: sub-folders unet/, vae/ (+ text_encoder/, tokenizer/) unless passed as kwargs."""
        kw = dict(kwargs)
        kw.setdefault("torch_dtype", torch_dtype)
        # run.py:296-312 (`--load_decoder_ckpt`): a fine-tuned head lives next to the UNet checkpoint as dpt_head_identity/ (DPT head),
        # dpt_head/ (the ReLU-terminated head single_infer rejects) or vae_decoder/ + vae_post_quant_conv/ (fine-tuned VAE decoder)
        dec_dir = kw.pop("load_decoder_ckpt", None)
        if dec_dir:
            have = set(os.listdir(dec_dir))
            if "dpt_head_identity" in have:
                kw.setdefault("customized_head", os.path.join(dec_dir, "dpt_head_identity"))
            elif "dpt_head" in have:
                kw.setdefault("customized_head", os.path.join(dec_dir, "dpt_head"))
            elif "vae_decoder" in have and "vae_post_quant_conv" in have and kw.get("vae") is None:
                kw["vae"] = compose_finetuned_vae(os.path.join(checkpoint, "vae"), dec_dir)
        for name in ("unet", "vae"):
            if kw.get(name) is None:
                kw[name] = os.path.join(checkpoint, name)
        # DiffusionPipeline.from_pretrained fills every module the caller did not pass from the checkpoint's own sub-folders: run.py leaves
        # `scheduler` unset for archs marigold / rgb_blending (run.py:361-368) and relies on <checkpoint>/scheduler/scheduler_config.json
        if kw.get("scheduler") is None and os.path.isfile(os.path.join(checkpoint, "scheduler", "scheduler_config.json")):
            from .scheduler import DDIMSchedulerCustomized
            kw["scheduler"] = DDIMSchedulerCustomized.from_pretrained(checkpoint, subfolder="scheduler")
            logging.info("GenPerceptPipeline.from_pretrained: no scheduler= argument, loaded %s", os.path.join(checkpoint, "scheduler", "scheduler_config.json"))
        # ... and the registered config values (register_to_config, genpercept_pipeline.py:128-132) from model_index.json
        mi = os.path.join(checkpoint, "model_index.json")
        if os.path.isfile(mi):
            import json
            with open(mi) as fh:
                idx = json.load(fh)
            for key in ("default_denoising_steps", "default_processing_resolution", "rgb_blending"):
                if key in idx and idx[key] is not None:
                    kw.setdefault(key, idx[key])
        if kw.get("text_encoder") is None and os.path.isdir(os.path.join(checkpoint, "text_encoder")):
            from transformers import CLIPTextModel, CLIPTokenizer
            kw["text_encoder"] = CLIPTextModel.from_pretrained(os.path.join(checkpoint, "text_encoder"))
            kw["tokenizer"] = CLIPTokenizer.from_pretrained(os.path.join(checkpoint, "tokenizer"))
        return cls(**kw)

    def to(self, device=None, dtype=None):
        if device is not None and not isinstance(device, torch.dtype):
            dev = torch.device(device)
            if dev.type != "cuda":
                raise RuntimeError("genpercept_amd runs on MI355X GPUs only")
            if self._engine is not None and dev != self._device:
                raise RuntimeError("the engine is already bound to " + str(self._device))
            self._device = dev if dev.index is not None else torch.device("cuda", 0)
        dt = dtype if dtype is not None else (device if isinstance(device, torch.dtype) else None)
        if dt is not None:
            from .engine import precision_of
            if self._engine is not None and precision_of(dt) != self._precision:
                raise RuntimeError("the engine is already built with " + self._precision + " elements")
            self._precision = precision_of(dt)
            self._noise_dtype = dt
        return self

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        from .engine import ELT_DTYPE
        if self._precision == "fp32c":  # contract precision: activations are stored in fp32 (the matrix products run on split bf16 operands)
            return torch.float32
        return ELT_DTYPE[self._precision]  # storage / MFMA-operand dtype of the engine; accumulation is fp32

    def enable_xformers_memory_efficient_attention(self):  # run.py:382-385 — attention is always flash-style here
        return None

    def set_progress_bar_config(self, **kwargs):
        return None

    # ---- engine ------------------------------------------------------------------------------------------------------
    def _ensure_engine(self):
        if self._engine is not None:
            return self._engine
        from .engine import Engine
        unet_sd, vae_sd, head_sd = _state_dict_of(self._unet_src), _state_dict_of(self._vae_src), _state_dict_of(self._head_src)
        if unet_sd is None or vae_sd is None:
            raise ValueError("unet and vae weights are required")
        if not (self.rgb_blending or self.genpercept_pipeline):
            lat = int(vae_sd["post_quant_conv.weight"].shape[0]) if "post_quant_conv.weight" in vae_sd else 4
            if int(unet_sd["conv_in.weight"].shape[1]) == lat:  # run.py:322-323: marigold on a 4-channel UNet checkpoint
                from .weights import replace_unet_conv_in
                unet_sd = replace_unet_conv_in(unet_sd)
                logging.info("Unet conv_in layer is replaced")
        ucfg, vcfg = gcfg.infer_unet_config(unet_sd), gcfg.infer_vae_config(vae_sd)
        dcfg = gcfg.infer_dpt_config(head_sd) if head_sd is not None else None
        if head_sd is not None:
            if not any(k.startswith("neck.fusion_stage") for k in head_sd):
                raise ValueError("unsupported customized_head (genpercept_pipeline.py:483-484)")
            if self._head_kind is None:  # a bare state dict: the two reference classes have identical keys, so it cannot be told apart
                logging.warning("customized_head given as a plain state dict: assuming DPTNeckHeadForUnetAfterUpsampleIdentity "
                                "(pass head_type='identity' to silence, 'relu' raises like the reference)")
            ucfg = gcfg.UNetConfig(**{**ucfg.__dict__, "has_out": False})
        eng = Engine(self._device.index or 0, ucfg, vcfg, dcfg, precision=self._precision)
        eng.load_state_dict("vae", vae_sd)
        eng.load_state_dict("unet", {k: v for k, v in unet_sd.items() if not (dcfg and (k.startswith("conv_out") or k.startswith("conv_norm_out")))})
        if head_sd is not None:
            eng.load_state_dict("dpt", head_sd)
        eng.finalize()
        self._engine, self.unet_config, self.vae_config, self.dpt_config = eng, ucfg, vcfg, dcfg
        self._timestep = 1
        self._ctx_loaded = None
        return eng

    def encode_text(self, prompt):
        """genpercept_pipeline.py:360-372 (padding='do_not_pad' => BOS,EOS for the empty prompt)."""
        if self.text_encoder is None or self.tokenizer is None:
            if self.text_embed is None:
                raise ValueError("no text_encoder/tokenizer and no precomputed text embedding were given")
            if prompt not in ("", None):
                logging.warning("a precomputed text embedding is in use; prompt %r is ignored", prompt)
            return
        ti = self.tokenizer(prompt, padding="do_not_pad", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt")
        with torch.no_grad():
            self.text_embed = self.text_encoder(ti.input_ids.to(self.text_encoder.device))[0].float().cpu()
        self._embed_prompt = prompt

    def _prepare(self, fix_timesteps, prompt):
        eng = self._ensure_engine()
        if self.text_embed is None:
            self.encode_text(prompt)
        if self._ctx_loaded is not self.text_embed:
            eng.set_context(self.text_embed)
            self._ctx_loaded = self.text_embed
        if fix_timesteps:
            t = int(fix_timesteps)
        elif self.scheduler is not None:  # genpercept_pipeline.py:403: scheduler.set_timesteps(1) -- [1] for the shipped config (leading
            self.scheduler.set_timesteps(1)  # spacing, steps_offset 1), 999 for trailing spacing, 0 for steps_offset 0 / linspace
            t = int(self.scheduler.timesteps[0])
        else:
            t = 1
        if t != self._timestep:
            eng.set_timestep(t)
            self._timestep = t
            # announced once per change: which timestep the one UNet pass runs at and where it came from, and which path follows
            src = "fix_timesteps" if fix_timesteps else ("scheduler.set_timesteps(1)" if self.scheduler is not None else "no scheduler: the shipped default")
            path = ("fused one-step path (gp_infer: x0 = -v)" if self.genpercept_pipeline and (self._x0_is_neg_v or self.customized_head is not None)
                    else "scheduler plan path (gp_infer_steps: DDIM update per step)")
            logging.info("GenPerceptPipeline: UNet timestep %d (%s); %s", t, src, path)
        return eng

    # ---- reference methods -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def single_infer(self, rgb_in: torch.Tensor, num_inference_steps: int = 1, generator=None, show_pbar: bool = False, fix_timesteps=None,
                     prompt="") -> torch.Tensor:
        """genpercept_pipeline.py:375-486.  rgb_in: [B,3,h,w] in [-1,1] (or uint8 0..255) -> [B,C,h',w'] in [0,1]."""
        if self.genpercept_pipeline:
            assert num_inference_steps == 1, "GenPercept only forward once."
        rgb_in = rgb_in.to(self._device)
        mode = self.mode or "depth"
        # the DPT-head branch never consults scheduler.step (:474-482: one UNet feature pass, then the head), whatever its prediction type
        if self.genpercept_pipeline and (self._x0_is_neg_v or self.customized_head is not None):
            return self._prepare(fix_timesteps, prompt).infer(rgb_in, mode)
        # the denoising loop (:447-465): the scheduler becomes one affine update per step, the loop itself runs in the engine
        eng = self._prepare(None, prompt)
        plan = self.scheduler.plan(int(num_inference_steps), fix_timesteps)
        noise = None
        if not (self.rgb_blending or self.genpercept_pipeline):  # marigold: the sample starts as noise (:413-420)
            lh, lw = eng.lib.gp_latent_size(rgb_in.shape[-2]), eng.lib.gp_latent_size(rgb_in.shape[-1])
            shape = (rgb_in.shape[0], self.vae_config.latent_channels, lh, lw)
            gdev = generator.device if generator is not None else self._device
            # drawn in the pipeline's dtype like the reference (:416-420): a half-precision run consumes the generator differently from fp32
            noise = torch.randn(shape, device=gdev, dtype=self._noise_dtype, generator=generator).to(device=self._device, dtype=torch.float32)
        return eng.infer_steps(rgb_in, mode, plan, noise)

    @torch.no_grad()
    def encode_rgb(self, rgb_in: torch.Tensor) -> torch.Tensor:
        """genpercept_pipeline.py:488-505."""
        return self._ensure_engine().vae_encode(rgb_in.to(self._device))

    @torch.no_grad()
    def decode_pred(self, pred_latent: torch.Tensor) -> torch.Tensor:
        """genpercept_pipeline.py:507-526 (channel mean for depth/matting/dis/disparity)."""
        from .engine import ONE_CHANNEL_MODES
        return self._ensure_engine().vae_decode(pred_latent.to(self._device), (self.mode or "depth") in ONE_CHANNEL_MODES)

    @torch.no_grad()
    def __call__(self, input_image: Union[Image.Image, torch.Tensor], denoising_steps: Optional[int] = None, ensemble_size: int = 1,
                 processing_res: Optional[int] = None, match_input_res: bool = True, resample_method: str = "bilinear", batch_size: int = 0,
                 generator=None, color_map: str = "Spectral", show_progress_bar: bool = True, ensemble_kwargs: Dict = None, mode=None,
                 fix_timesteps=None, prompt="") -> GenPerceptOutput:
        assert mode is not None, "mode of GenPerceptPipeline can be chosen from ['depth', 'normal', 'seg', 'matting', 'dis']."
        self.mode = mode
        if denoising_steps is None:
            denoising_steps = self.default_denoising_steps
        if processing_res is None:
            processing_res = self.default_processing_resolution
        assert processing_res >= 0
        assert ensemble_size >= 1
        if self.genpercept_pipeline:  # genpercept_pipeline.py:211-216
            assert ensemble_size == 1
            assert denoising_steps == 1
        else:
            self._check_inference_step(denoising_steps)
        resample = get_resample_method(resample_method)
        if isinstance(input_image, Image.Image):
            arr = np.asarray(input_image.convert("RGB"))
            rgb = torch.from_numpy(arr.copy()).permute(2, 0, 1).unsqueeze(0)  # [1, rgb, H, W] uint8
        elif isinstance(input_image, torch.Tensor):
            rgb = input_image
        else:
            raise TypeError(f"Unknown input type: {type(input_image) = }")
        input_size = rgb.shape
        assert 4 == rgb.dim() and 3 == input_size[-3], f"Wrong input shape {input_size}, expected [1, rgb, H, W]"
        # genpercept_pipeline.py:258-266: batch size of the ensemble loader from the longest edge of the RESIZED image, max(rgb_norm.shape[1:])
        # (always 1 on the one-step path: ensemble_size == 1)
        if batch_size <= 0:
            h, w = int(input_size[-2]), int(input_size[-1])
            if processing_res > 0:
                from .image_util import resize_max_res_size
                h, w = resize_max_res_size(h, w, int(processing_res))
            batch_size = find_batch_size(ensemble_size=ensemble_size, input_res=max(3, h, w), dtype=self._noise_dtype)
        assert batch_size >= 1
        opts = dict(steps=int(denoising_steps), ensemble_size=int(ensemble_size), batch_size=int(batch_size), generator=generator,
                    ensemble_kwargs=ensemble_kwargs)
        outs = self._run(rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts)
        return outs[0] if rgb.shape[0] == 1 else outs

    def _check_inference_step(self, n_step: int) -> None:
        """genpercept_pipeline.py:338-358 (DDIM branch; LCM schedulers are refused at construction)."""
        assert n_step >= 1
        if n_step < 10:
            logging.warning(f"Too few denoising steps: {n_step}. Recommended to use the LCM checkpoint for few-step inference.")

    def _warn_if_saturated(self):
        """fp16 engine only: saturation is never silent.  The reference's own half precision (run.py --half_precision) turns an activation
        beyond 65504 into inf / NaN; this engine clips it and COUNTS the call (`gp_saturation_events`): the map just produced is then not the
        fp32 path's map, and the caller is told so."""
        eng = self._engine
        if eng is None or getattr(eng, "precision", None) != "fp16" or not hasattr(eng, "saturation_events"):
            return
        n = eng.saturation_events(reset=True)
        self.last_saturation_events = n
        if n:
            logging.warning("GenPerceptPipeline: %d kernel group(s) of the last call clipped activations at the fp16 range (+-65504): the result "
                            "deviates from the fp32 path.  Use torch_dtype=torch.float32 (contract precision, fp32 range) or torch.bfloat16 for "
                            "this checkpoint / input.", n)

    def _predict(self, x: torch.Tensor, fix_timesteps, prompt, opts: Optional[dict]) -> torch.Tensor:
        """The batched prediction + test-time ensembling of __call__ (genpercept_pipeline.py:250-297), per image of `x`."""
        o = opts or {}
        steps, e = o.get("steps", self.default_denoising_steps if not self.genpercept_pipeline else 1), o.get("ensemble_size", 1)
        gen = o.get("generator")
        if e == 1:
            return self.single_infer(x, steps, gen, False, fix_timesteps, prompt)
        from .ensemble import ensemble_depth
        bs = max(1, o.get("batch_size", 1))
        outs = []
        for i in range(x.shape[0]):
            dup = x[i:i + 1].expand(e, -1, -1, -1)
            preds = torch.cat([self.single_infer(dup[j:j + bs], steps, gen, False, fix_timesteps, prompt) for j in range(0, e, bs)], dim=0)
            pred, _ = ensemble_depth(preds, scale_invariant=True, shift_invariant=True, max_res=50, **(o.get("ensemble_kwargs") or {}))
            outs.append(pred)
        return torch.cat(outs, dim=0)

    def infer_batch(self, images: Union[Sequence[Image.Image], torch.Tensor], mode: str, processing_res: Optional[int] = None,
                    match_input_res: bool = True, resample_method: str = "bilinear", color_map: Optional[str] = "Spectral", fix_timesteps=None,
                    prompt="", denoising_steps: Optional[int] = None, ensemble_size: int = 1, generator=None) -> List[GenPerceptOutput]:
        """Batch entry (new): images of one common size, one engine call; results are per image (DPT min-max per image)."""
        self.mode = mode
        if processing_res is None:
            processing_res = self.default_processing_resolution
        if not torch.is_tensor(images):
            images = torch.stack([torch.from_numpy(np.asarray(im.convert("RGB")).copy()).permute(2, 0, 1) for im in images])
        assert images.dim() == 4 and images.shape[1] == 3
        steps = self.default_denoising_steps if denoising_steps is None else int(denoising_steps)
        if self.genpercept_pipeline:
            assert steps == 1 and ensemble_size == 1
        opts = dict(steps=steps, ensemble_size=int(ensemble_size), batch_size=max(1, int(ensemble_size)), generator=generator, ensemble_kwargs=None)
        return self._run(images, processing_res, match_input_res, get_resample_method(resample_method), color_map, fix_timesteps, prompt, opts)

    def _run(self, rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts=None) -> List[GenPerceptOutput]:
        from .engine import RESAMPLE_CODE
        # device pre / post for uint8 and fp32 images; fp16 / bf16 / fp64 tensors keep the host recipe, which -- like the reference
        # (genpercept_pipeline.py:240-247, image_util.py:75-105) -- resizes in the tensor's OWN dtype before the normalisation (the
        # device path would promote them to fp32 first and deviate slightly)
        if rgb.dtype in (torch.uint8, torch.float32) and resample in RESAMPLE_CODE and not os.environ.get("GENPERCEPT_HOST_PREPOST"):
            return self._run_device(rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts)
        return self._run_host(rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts)

    def _run_device(self, rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts=None) -> List[GenPerceptOutput]:
        """Pre / post processing on the GPU (gp_preprocess / gp_postprocess): the image goes up once, resize_max_res, the model, the
        resize back to the input size, clip, colour map and the 8-bit image all run on the device; pred_np and the coloured bytes come down.
        uint8 images (PIL input) stay uint8 up to the engine's prologue; float tensors (genpercept_trainer.py:1151-1165) are resized in fp32
        without rounding and normalised to [-1, 1] by gp_preprocess_f32, with the reference's range assertion (genpercept_pipeline.py:247)."""
        from . import engine as ge
        input_size = rgb.shape
        if color_map is not None:
            assert self.mode in ["depth", "disparity"]
        x = rgb.to(self._device, non_blocking=True)
        size_in = tuple(int(v) for v in input_size[-2:])
        size_p = ge.resize_max_res_size(size_in[0], size_in[1], int(processing_res)) if processing_res > 0 else size_in
        if x.dtype == torch.uint8:
            x = ge.preprocess(x, size_p, resample)
        else:
            x = ge.preprocess(x, size_p, resample, normalize=True)
            assert float(x.amin()) >= -1.0 and float(x.amax()) <= 1.0
        pred = self._predict(x, fix_timesteps, prompt, opts)
        self._warn_if_saturated()
        size = tuple(int(v) for v in input_size[-2:]) if match_input_res else tuple(pred.shape[-2:])
        one_ch = pred.shape[1] == 1
        pred_out, col, q8 = ge.postprocess(pred, size, resample, cmap=color_map if one_ch else None, q_bits=0 if color_map is not None else 8)
        pred_np = pred_out.cpu().numpy()
        col_np = col.cpu().numpy() if col is not None else None
        q8_np = q8.cpu().numpy() if q8 is not None else None
        outs = []
        for i in range(pred_np.shape[0]):
            p = pred_np[i].squeeze()
            if color_map is not None:
                col_img = Image.fromarray(col_np[i])
            else:
                c8 = q8_np[i].squeeze()
                if c8.ndim == 3 and c8.shape[0] == 3:
                    c8 = np.transpose(c8, (1, 2, 0))
                col_img = Image.fromarray(np.ascontiguousarray(c8))
            if p.ndim == 3 and p.shape[0] == 3:
                p = np.transpose(p, (1, 2, 0))
            outs.append(GenPerceptOutput(pred_np=p, pred_colored=col_img))
        return outs

    def _run_host(self, rgb, processing_res, match_input_res, resample, color_map, fix_timesteps, prompt, opts=None) -> List[GenPerceptOutput]:
        """The same steps with the resizes / colour map on the host (torch CPU + matplotlib): GENPERCEPT_HOST_PREPOST=1, and the A/B reference of
        the device path in tests/test_prepost_gpu.py."""
        input_size = rgb.shape
        if processing_res > 0:
            rgb = resize_max_res(rgb, max_edge_resolution=processing_res, resample_method=resample)
        if rgb.dtype == torch.uint8:
            rgb_in = rgb  # x/255*2-1 happens in the engine's prologue kernel (same fp32 formula)
        else:
            rgb_in = (rgb / 255.0 * 2.0 - 1.0).float()  # in the tensor's own dtype first, like genpercept_pipeline.py:245
            assert rgb_in.min() >= -1.0 and rgb_in.max() <= 1.0
        pred = self._predict(rgb_in, fix_timesteps, prompt, opts)
        self._warn_if_saturated()
        if match_input_res:
            pred = resize_to(pred, input_size[-2:], resample)
        pred = pred.cpu().numpy().clip(0, 1)
        outs = []
        for i in range(pred.shape[0]):
            p = pred[i].squeeze()
            if color_map is not None:
                assert self.mode in ["depth", "disparity"]
                col = colorize_depth_maps(p, 0, 1, cmap=color_map).squeeze()
                col_img = Image.fromarray(chw2hwc((col * 255).astype(np.uint8)))
            else:
                c8 = (p * 255.0).astype(np.uint8)
                if c8.ndim == 3 and c8.shape[0] == 3:
                    c8 = np.transpose(c8, (1, 2, 0))
                col_img = Image.fromarray(c8)
            if p.ndim == 3 and p.shape[0] == 3:
                p = np.transpose(p, (1, 2, 0))
            outs.append(GenPerceptOutput(pred_np=p, pred_colored=col_img))
        return outs
