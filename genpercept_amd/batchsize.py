"""Operating batch size for the ensemble loop — mirror of the reference's `find_batch_size`
(/root/reference/genpercept/util/batchsize.py:51-81; called from genpercept_pipeline.py:264-270).

The reference looks the batch up in a table calibrated on NVIDIA cards (A100 / RTX / GTX, fp32 and fp16 weights) and then
clamps it against the ensemble size.  On the one-step path ensemble_size == 1, so the answer is always 1; the function is
mirrored because callers of the pipeline use it.  Here the "fits" figure comes from the engine's own memory model (bf16
weights + the activation arena, measured on MI355X) instead of a card table; the clamp rules are the reference's:

  * no GPU                                  -> 1                    (batchsize.py:64-65)
  * bs > ensemble_size                      -> ensemble_size        (batchsize.py:75-76)
  * ceil(ens / 2) < bs < ensemble_size      -> ceil(ens / 2)        (batchsize.py:77-78)
  * nothing fits                            -> 1                    (batchsize.py:81)
"""
import math
from typing import Optional

import torch

# engine memory model (MI355X, bf16): resident weights + per-image activation arena at 768x768, growing with the pixel count
WEIGHTS_GB = 2.1
ARENA_GB_PER_IMAGE_768 = 2.6


def fit_batch_size(input_res: int, total_vram_gb: float) -> int:
    """Largest batch whose arena fits next to the weights (0 when not even one image fits)."""
    per_image = ARENA_GB_PER_IMAGE_768 * (max(int(input_res), 64) / 768.0) ** 2
    return max(int((total_vram_gb * 0.9 - WEIGHTS_GB) // per_image), 0)


def clamp_batch_size(bs: int, ensemble_size: int) -> int:
    """The reference's clamp of a fitting batch size against the ensemble size (batchsize.py:74-79)."""
    if bs > ensemble_size:
        return ensemble_size
    if math.ceil(ensemble_size / 2) < bs < ensemble_size:
        return math.ceil(ensemble_size / 2)
    return bs


def find_batch_size(ensemble_size: int, input_res: int, dtype: Optional[torch.dtype] = None, total_vram_gb: Optional[float] = None) -> int:
    """Same signature and meaning as the reference; `dtype` is accepted for compatibility (the engine always stores bf16)."""
    if total_vram_gb is None:
        if not torch.cuda.is_available():
            return 1
        total_vram_gb = torch.cuda.mem_get_info()[1] / 1024.0 ** 3
    bs = fit_batch_size(input_res, total_vram_gb)
    if bs < 1:
        return 1
    return clamp_batch_size(bs, ensemble_size)
