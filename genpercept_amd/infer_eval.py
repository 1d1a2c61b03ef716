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


# Dataset conventions the evaluation needs (config/dataset/eval/*.yaml + src/dataset/{nyu,kitti,eth3d,scannet,diode}_dataset.py): depth
# range, naming mode, how the ground truth is stored, evaluation crop / mask.  Pinned to the reference classes by tests/golden/datasets_ref.npz.
#   gt: ("png", divisor) | ("eth3d_bin", (H, W)) | ("npy", None)
DATASETS: Dict[str, dict] = {
    "nyu": dict(min_depth=1e-3, max_depth=10.0, name_mode=FileNameMode.rgb_id, gt=("png", 1000.0), eval_crop=(45, 471, 41, 601)),
    "kitti": dict(min_depth=1e-5, max_depth=80.0, name_mode=FileNameMode.id, gt=("png", 256.0), kitti_bm_crop=True, valid_mask_crop="eigen"),
    "eth3d": dict(min_depth=1e-5, max_depth=float("inf"), name_mode=FileNameMode.id, gt=("eth3d_bin", (4032, 6048))),
    "scannet": dict(min_depth=1e-3, max_depth=10.0, name_mode=FileNameMode.id, gt=("png", 1000.0)),
    "diode": dict(min_depth=0.6, max_depth=350.0, name_mode=FileNameMode.id, gt=("npy", None), mask_from_file=True),
}
DATASETS["nyu_v2"] = DATASETS["nyu"]  # `name:` of config/dataset/eval/data_nyu_test.yaml


def read_depth_png(path: str, depth_scale: float) -> np.ndarray:
    return np.asarray(Image.open(path)).astype(np.float32) * np.float32(depth_scale)


def kitti_benchmark_crop(img: np.ndarray) -> np.ndarray:
    """kitti_dataset.py:83-110: bottom-aligned, horizontally centred 352 x 1216 window of a [.., H, W] array (RGB and depth alike)."""
    h, w = img.shape[-2:]
    top, left = int(h - 352), int((w - 1216) / 2)
    return img[..., top:top + 352, left:left + 1216]


def kitti_eval_mask(h: int, w: int, kind: Optional[str]) -> np.ndarray:
    """kitti_dataset.py:112-133: Garg (ECCV16) / Eigen (NIPS14) evaluation window as a boolean [h, w] mask (None: everything)."""
    m = np.zeros((h, w), dtype=bool)
    if kind is None:
        m[:] = True
    elif kind == "garg":
        m[int(0.40810811 * h):int(0.99189189 * h), int(0.03594771 * w):int(0.96405229 * w)] = True
    elif kind == "eigen":
        m[int(0.3324324 * h):int(0.91351351 * h), int(0.0359477 * w):int(0.96405229 * w)] = True
    else:
        raise ValueError(f"Unknown crop type: {kind}")
    return m


def read_gt_depth(path: str, dataset: str) -> np.ndarray:
    """Ground-truth depth [H, W] float32 in metres as the reference's dataset class decodes it (after the KITTI benchmark crop)."""
    cfg = DATASETS[dataset]
    kind, arg = cfg["gt"]
    if kind == "png":          # kitti / 256 (kitti_dataset.py:60-68), scannet and nyu / 1000 (scannet_dataset.py:31-38, nyu_dataset.py:39-46)
        d = np.asarray(Image.open(path)).astype(np.float32) / np.float32(arg)
    elif kind == "eth3d_bin":  # eth3d_dataset.py:38-57: raw float32, inf = no measurement -> 0
        d = np.fromfile(path, dtype=np.float32).copy()
        d[d == np.inf] = 0.0
        d = d.reshape(arg)
    elif kind == "npy":        # diode_dataset.py:37-50
        d = np.load(path).squeeze().astype(np.float32)
    else:
        raise ValueError(kind)
    if cfg.get("kitti_bm_crop"):
        d = kitti_benchmark_crop(d)
    return d


def dataset_valid_mask(depth: np.ndarray, dataset: str, mask_path: Optional[str] = None) -> np.ndarray:
    """base_dataset.py:410-413 range test, then the dataset's evaluation crop (NYU Eigen crop, KITTI Garg / Eigen window) or, for DIODE,
    the mask file that ships with the sample (diode_dataset.py:72-78)."""
    cfg = DATASETS[dataset]
    if cfg.get("mask_from_file"):
        if mask_path is None:
            raise ValueError("DIODE samples carry their validity mask as a third path")
        return np.load(mask_path).squeeze().astype(bool)
    m = valid_mask_of(depth, cfg["min_depth"], cfg["max_depth"], cfg.get("eval_crop"))
    if "valid_mask_crop" in cfg:
        m = m & kitti_eval_mask(depth.shape[-2], depth.shape[-1], cfg["valid_mask_crop"])
    return m


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
                  resample_method: str = "bilinear", fix_timesteps=None, prompt: str = "", rgb_crop: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                  prefetch: int = 2) -> List[str]:
    """infer.py:408-447.  Returns the paths written.  rgb_crop: e.g. kitti_benchmark_crop (applied to the [3, H, W] image).
    prefetch > 0: the next `prefetch` images are decoded (and cropped) by a helper thread while the engine works on the current one, and the .npy
    files are written by another -- the role the DataLoader workers play in the reference's loop; order and results are those of prefetch = 0."""
    written = []
    samples = [s for s in samples if len(s) < 2 or s[1] != "None"]  # kitti_dataset.py:47: entries without ground truth are skipped

    def load(rgb_rel):
        img = Image.open(os.path.join(base_dir, rgb_rel)).convert("RGB")
        if rgb_crop is not None:  # KITTI: the benchmark crop is applied to the RGB as well (kitti_dataset.py:70-74)
            img = Image.fromarray(np.ascontiguousarray(np.moveaxis(rgb_crop(np.moveaxis(np.asarray(img), -1, 0)), 0, -1)))
        return img

    def save(rgb_rel, pred_np):
        scene_dir = os.path.join(output_dir, os.path.dirname(rgb_rel))
        os.makedirs(scene_dir, exist_ok=True)
        save_to = os.path.join(scene_dir, get_pred_name(os.path.basename(rgb_rel), name_mode, suffix=".npy"))
        np.save(save_to, pred_np)
        return save_to

    def infer(img):
        return pipe(img, denoising_steps=denoise_steps, ensemble_size=ensemble_size, processing_res=processing_res, match_input_res=match_input_res,
                    batch_size=0, color_map=None, show_progress_bar=False, resample_method=resample_method, mode=mode,
                    fix_timesteps=fix_timesteps, prompt=prompt)

    if prefetch <= 0 or len(samples) < 2:
        for s in samples:
            written.append(save(s[0], infer(load(s[0])).pred_np))
        return written
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as loader, ThreadPoolExecutor(max_workers=1) as writer:
        pending, saves = deque(), []
        it = iter(samples)
        for s in it:
            pending.append((s[0], loader.submit(load, s[0])))
            if len(pending) > prefetch:
                break
        while pending:
            rgb_rel, fut = pending.popleft()
            img = fut.result()                       # (a decode error surfaces here, at the image it belongs to)
            nxt = next(it, None)
            if nxt is not None:
                pending.append((nxt[0], loader.submit(load, nxt[0])))
            saves.append(writer.submit(save, rgb_rel, infer(img).pred_np))
        written = [f.result() for f in saves]
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
        if depth_rel == "None":
            continue
        gt = read_gt(os.path.join(base_dir, depth_rel)) if read_gt else read_gt_depth(os.path.join(base_dir, depth_rel), dataset)
        vm = dataset_valid_mask(gt, dataset, os.path.join(base_dir, s[2]) if cfg.get("mask_from_file") and len(s) > 2 else None)
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
