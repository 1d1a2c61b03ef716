"""Dataset inference loop and evaluation driver next to the hot path.

Mirrors, for RGB + depth datasets listed in the reference's `data_split/*/filename_list_*.txt` format:
  * infer.py:408-447        -- per image: PIL RGB -> pipe(..., batch_size=0, color_map=None, ...) -> `.npy` under the scene directory,
                               named by get_pred_name (src/dataset/base_dataset.py:531-545, PerceptionFileNameMode :43-49);
  * eval.py:143-244         -- per image: load prediction, least-squares alignment (depth or disparity space), clip to the dataset's
                               depth range, ten metrics (eval_metrics.py), mean over images, `eval_metrics-<alignment>.txt` + per-sample csv;
  * src/dataset/nyu_dataset.py:39-58, base_dataset.py:399-413 -- NYUv2 decoding (png / 1000), validity (min < d < max) and Eigen crop.
Pinned to the reference by tests/golden/infer_eval_ref.npz (naming modes; disparity-space alignment and max_resolution fits come from
the reference's alignment.py).  Nothing here touches the GPU except through the pipeline object handed in.
"""
from __future__ import annotations

import os
from enum import Enum
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image

from . import eval_metrics as em


class FileNameMode(Enum):
    """Prediction file naming modes (PerceptionFileNameMode, base_dataset.py:43-49)."""
    id = 1        # id.png
    rgb_id = 2    # rgb_id.png
    i_d_rgb = 3   # i_d_1_rgb.png
    rgb_i_d = 4


def get_pred_name(rgb_basename: str, name_mode: FileNameMode, suffix: str = ".png") -> str:
    if name_mode == FileNameMode.rgb_id:
        pred = "pred_" + rgb_basename.split("_")[1]
    elif name_mode == FileNameMode.i_d_rgb:
        pred = rgb_basename.replace("_rgb.", "_pred.")
    elif name_mode == FileNameMode.id:
        pred = "pred_" + rgb_basename
    elif name_mode == FileNameMode.rgb_i_d:
        pred = "pred_" + "_".join(rgb_basename.split("_")[1:])
    else:
        raise NotImplementedError(name_mode)
    return os.path.splitext(pred)[0] + suffix


def read_filename_list(path: str) -> List[List[str]]:
    """One sample per line: `rgb_rel_path depth_rel_path [filled_rel_path]` (base_dataset.py:97-103)."""
    with open(path, "r") as f:
        return [ln.split() for ln in f.read().splitlines() if ln.strip()]


# dataset conventions the evaluation needs (nyu_dataset.py:28-58): depth range, naming mode, decode scale, Eigen evaluation crop
DATASETS: Dict[str, dict] = {
    "nyu": dict(min_depth=1e-3, max_depth=10.0, name_mode=FileNameMode.rgb_id, depth_scale=1.0 / 1000.0, eval_crop=(45, 471, 41, 601)),
}


def read_depth_png(path: str, depth_scale: float) -> np.ndarray:
    return np.asarray(Image.open(path)).astype(np.float32) * np.float32(depth_scale)


def valid_mask_of(depth: np.ndarray, min_depth: float, max_depth: float, eval_crop: Optional[Tuple[int, int, int, int]] = None) -> np.ndarray:
    m = (depth > min_depth) & (depth < max_depth)          # base_dataset.py:410-413
    if eval_crop is not None:                               # nyu_dataset.py:51-56
        c = np.zeros_like(m)
        y0, y1, x0, x1 = eval_crop
        c[y0:y1, x0:x1] = True
        m = m & c
    return m


def run_inference(pipe, base_dir: str, samples: Sequence[Sequence[str]], output_dir: str, name_mode: FileNameMode, mode: str = "depth",
                  denoise_steps: int = 1, ensemble_size: int = 1, processing_res: int = 0, match_input_res: bool = True,
                  resample_method: str = "bilinear", fix_timesteps=None, prompt: str = "") -> List[str]:
    """infer.py:408-447.  Returns the paths written."""
    written = []
    for s in samples:
        rgb_rel = s[0]
        img = Image.open(os.path.join(base_dir, rgb_rel)).convert("RGB")
        out = pipe(img, denoising_steps=denoise_steps, ensemble_size=ensemble_size, processing_res=processing_res, match_input_res=match_input_res,
                   batch_size=0, color_map=None, show_progress_bar=False, resample_method=resample_method, mode=mode,
                   fix_timesteps=fix_timesteps, prompt=prompt)
        scene_dir = os.path.join(output_dir, os.path.dirname(rgb_rel))
        os.makedirs(scene_dir, exist_ok=True)
        save_to = os.path.join(scene_dir, get_pred_name(os.path.basename(rgb_rel), name_mode, suffix=".npy"))
        np.save(save_to, out.pred_np)
        written.append(save_to)
    return written


def evaluate_predictions(prediction_dir: str, base_dir: str, samples: Sequence[Sequence[str]], dataset: str = "nyu",
                         alignment: str = "least_square", alignment_max_res: Optional[int] = None, output_dir: Optional[str] = None,
                         pred_suffix: str = ".npy", read_gt: Optional[Callable[[str], np.ndarray]] = None) -> Dict[str, float]:
    """eval.py:143-244: mean of the per-image metrics; optionally writes `eval_metrics-<alignment>.txt` and `per_sample_metrics-...csv`."""
    cfg = DATASETS[dataset]
    names = list(em.METRICS.keys())
    sums = {k: 0.0 for k in names}
    per_sample = []
    n = 0
    for s in samples:
        rgb_rel, depth_rel = s[0], s[1]
        gt = read_gt(os.path.join(base_dir, depth_rel)) if read_gt else read_depth_png(os.path.join(base_dir, depth_rel), cfg["depth_scale"])
        vm = valid_mask_of(gt, cfg["min_depth"], cfg["max_depth"], cfg.get("eval_crop"))
        pred_name = os.path.join(os.path.dirname(rgb_rel), get_pred_name(os.path.basename(rgb_rel), cfg["name_mode"], suffix=pred_suffix))
        pred_path = os.path.join(prediction_dir, pred_name)
        if not os.path.exists(pred_path):
            continue
        pred = np.load(pred_path)
        m = em.evaluate_depth(pred, gt, vm, cfg["min_depth"], cfg["max_depth"], alignment=alignment, alignment_max_res=alignment_max_res)
        for k in names:
            sums[k] += m[k]
        per_sample.append((pred_name, [m[k] for k in names]))
        n += 1
    result = {k: (sums[k] / n if n else float("nan")) for k in names}
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        tag = f"-{alignment}" if alignment else ""
        with open(os.path.join(output_dir, f"per_sample_metrics{tag}.csv"), "w") as f:
            f.write("filename," + ",".join(names) + "\n")
            for nm, vals in per_sample:
                f.write(nm + "," + ",".join(str(v) for v in vals) + "\n")
        with open(os.path.join(output_dir, f"eval_metrics{tag}.txt"), "w") as f:
            f.write(f"Evaluation metrics:\n    of predictions: {prediction_dir}\n    on dataset: {dataset}\n")
            f.write(f"min_depth = {cfg['min_depth']}\nmax_depth = {cfg['max_depth']}\n")
            f.write("  ".join(names) + "\n" + "  ".join(f"{result[k]:.6g}" for k in names) + "\n")
    return result
