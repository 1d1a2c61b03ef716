#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Runs ONLY in the build container (needs /root/reference).

Fixtures are data (seeds, inputs, expected outputs) — never reference source.

  dpt_head_ref.npz   outputs of the REFERENCE's own DPTNeckHeadForUnetAfterUpsampleIdentity
                     (/root/reference/genpercept/models/dpt_head.py) on seeded weights/inputs.  `diffusers` is not
                     installed, so the two symbols dpt_head.py imports from it (LoRACompatibleConv, USE_PEFT_BACKEND;
                     dpt_head.py:20-21,130) are provided by a stub module.
  dpt_head_ref_odd.npz  the same class on odd feature shapes (9x11, 13x10, 29x39 latents): the bilinear-resize branch dpt_head.py:297-300.
  metrics_ref.npz    outputs of the REFERENCE's src/util/metric.py + src/util/alignment.py on seeded arrays.
  batchsize_ref.npz  the REFERENCE's find_batch_size (genpercept/util/batchsize.py) on a grid of cards / resolutions / ensembles.
  infer_eval_ref.npz the REFERENCE's get_pred_name (all naming modes) and alignment variants (max_resolution, disparity-space protocol).
  datasets_ref.npz   the REFERENCE's dataset conventions beyond NYU (KITTI crop / masks / decode, ETH3D, ScanNet, DIODE) and angular_loss.
  image_util_ref.npz the REFERENCE's colorize_depth_maps / chw2hwc outputs, resize_max_res size rule, resample-method names.
  scheduler_ref.npz  the REFERENCE's DDIMSchedulerCustomized (src/customized_modules/ddim.py:144-217): betas / alphas_cumprod /
                     final_alpha_cumprod of every hf_configs/scheduler_* config (+ the scaled_linear_power schedule) and _get_variance;
                     its diffusers base class is a stub (the class's own __init__ and _get_variance never call into it).
  ensemble_ref.npz   the REFERENCE's ensemble_depth (genpercept/util/ensemble.py) on seeded affine-distorted maps (inputs below
                     max_res, so torchvision -- stubbed -- is never reached).
  refexec_tiny.npz   outputs of the REFERENCE's CustomUNet2DConditionModel.forward and GenPerceptPipeline.__call__ / single_infer /
                     encode_rgb / decode_pred EXECUTED over stub diffusers base classes whose blocks are the oracle's functions (tiny configs,
                     the inputs of e2e_tiny.npz / e2e_multistep.npz; needs those two files: run after "e2e" and "multistep").
  refexec_v1_tiny.npz  outputs of GenPercept_v1's single_infer (pipeline_genpercept.py:263-309: timesteps=[1], pred_latent = -unet_pred) EXECUTED
                     over the same stub bases, with the v1 tree's own empty_text_embed.npy as context (all 77 rows, and rows [0:2]).
  e2e_multistep.npz  oracle goldens of the multi-step archs (marigold: noise + 8-channel conv_in; rgb_blending) on the tiny configs.
  e2e_tiny.npz       end-to-end goldens of the fp32 oracle (oracle/) on the tiny configs: inputs + expected outputs
                     for every stage (latent, unet out, feats, decode, final) — what the HIP path is checked against
                     on the GPU box where neither /root/reference nor large weights exist.

Weights are regenerated from seeds by oracle.sd21.synth_state_dict (deterministic for a fixed torch build).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import dpt as odpt  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402
from oracle import sd21 as osd  # noqa: E402


def _stub_diffusers():
    d = types.ModuleType("diffusers")
    dm = types.ModuleType("diffusers.models")
    dl = types.ModuleType("diffusers.models.lora")
    du = types.ModuleType("diffusers.utils")
    dl.LoRACompatibleConv = torch.nn.Conv2d
    du.USE_PEFT_BACKEND = True
    d.models, dm.lora, d.utils = dm, dl, du
    sys.modules.update({"diffusers": d, "diffusers.models": dm, "diffusers.models.lora": dl, "diffusers.utils": du})


def make_dpt_golden():
    _stub_diffusers()
    sys.path.insert(0, REF)
    from transformers import DPTConfig
    import importlib.util
    # load the file directly: genpercept/__init__.py imports the whole pipeline (needs real diffusers/torchvision)
    spec = importlib.util.spec_from_file_location("ref_dpt_head", os.path.join(REF, "genpercept/models/dpt_head.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DPTNeckHeadForUnetAfterUpsampleIdentity = mod.DPTNeckHeadForUnetAfterUpsampleIdentity

    import json
    with open(os.path.join(REF, "hf_configs/dpt-sd2.1-unet-after-upsample-general/config.json")) as f:
        cfg = DPTConfig(**json.load(f))
    head = DPTNeckHeadForUnetAfterUpsampleIdentity(cfg).eval()
    manifest = odpt.dpt_manifest()
    ref_keys = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    assert ref_keys == {k: tuple(v) for k, v in manifest.items()}, "DPT manifest differs from the reference class"
    sd = osd.synth_state_dict(manifest, seed=7)
    head.load_state_dict(sd, strict=True)
    out = {}
    for tag, (h, w) in {"a": (12, 12), "b": (16, 12)}.items():
        g = torch.Generator().manual_seed(100 + h * w)
        feats = [torch.randn(1, 320, h, w, generator=g), torch.randn(1, 640, h, w, generator=g),
                 torch.randn(1, 1280, h // 2, w // 2, generator=g), torch.randn(1, 1280, h // 4, w // 4, generator=g)]
        with torch.no_grad():
            y = head(hidden_states=[f.clone() for f in feats], return_depth_only=True)
        out[f"{tag}_hw"] = np.array([h, w])
        out[f"{tag}_out"] = y.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "dpt_head_ref.npz"), seed=np.array(7), **out)
    print("dpt_head_ref.npz", {k: v.shape for k, v in out.items()})


def make_dpt_golden_odd():
    """DPT head on the feature shapes a NON-multiple-of-4 latent produces (UNet stride-2 pad-1 downsamples: 9x11 -> 5x6 -> 3x3, 13x10 ->
    7x5 -> 4x3): the fused map and the next neck feature then differ in size and the reference takes its bilinear-resize branch
    (dpt_head.py:297-300), which the even-shape fixture above never reaches."""
    _stub_diffusers()
    sys.path.insert(0, REF)
    from transformers import DPTConfig
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ref_dpt_head", os.path.join(REF, "genpercept/models/dpt_head.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(REF, "hf_configs/dpt-sd2.1-unet-after-upsample-general/config.json")) as f:
        cfg = DPTConfig(**json.load(f))
    head = mod.DPTNeckHeadForUnetAfterUpsampleIdentity(cfg).eval()
    sd = osd.synth_state_dict(odpt.dpt_manifest(), seed=7)
    head.load_state_dict(sd, strict=True)
    out = {}
    for tag, (h, w) in {"c": (9, 11), "d": (13, 10), "e": (29, 39)}.items():
        h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        h4, w4 = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
        g = torch.Generator().manual_seed(200 + h * w)
        feats = [torch.randn(1, 320, h, w, generator=g), torch.randn(1, 640, h, w, generator=g),
                 torch.randn(1, 1280, h2, w2, generator=g), torch.randn(1, 1280, h4, w4, generator=g)]
        with torch.no_grad():
            y = head(hidden_states=[f.clone() for f in feats], return_depth_only=True)
        out[f"{tag}_hw"] = np.array([h, w])
        out[f"{tag}_out"] = y.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "dpt_head_ref_odd.npz"), seed=np.array(7), **out)
    print("dpt_head_ref_odd.npz", {k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "dpt_head_ref_odd.npz")) // 1024, "KiB")


def make_metrics_golden():
    sys.path.insert(0, REF)
    from src.util import metric as rmetric
    from src.util.alignment import align_depth_least_square, depth2disparity

    g = torch.Generator().manual_seed(11)
    gt = torch.rand(2, 48, 64, generator=g) * 9.0 + 0.5
    pred = (gt * 0.37 + 0.8 + 0.3 * torch.randn(2, 48, 64, generator=g)).clamp_min(0.05)
    mask = torch.rand(2, 48, 64, generator=g) > 0.2
    out = {"gt": gt.numpy(), "pred": pred.numpy(), "mask": mask.numpy()}
    aligned, scale, shift = align_depth_least_square(gt_arr=gt[0].numpy(), pred_arr=pred[0].numpy(), valid_mask_arr=mask[0].numpy(),
                                                     return_scale_shift=True, max_resolution=None)
    out["aligned0"] = aligned
    out["scale_shift0"] = np.array([scale, shift], dtype=np.float64).reshape(-1)
    disp, dmask = depth2disparity(gt[0].numpy(), return_mask=True)
    out["disp0"] = disp
    names = ["abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10", "delta1_acc",
             "delta2_acc", "delta3_acc", "i_rmse", "silog_rmse"]
    al = torch.from_numpy(np.clip(aligned, 1e-3, 10.0))[None]
    for n in names:
        fn = getattr(rmetric, n)
        out["m_" + n] = np.array(float(fn(al, gt[:1], mask[:1])))
    np.savez_compressed(os.path.join(HERE, "metrics_ref.npz"), **out)
    print("metrics_ref.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("m_")})


def make_batchsize_golden():
    """Clamp behaviour of the REFERENCE's find_batch_size (genpercept/util/batchsize.py:51-81): for a card / resolution / dtype
    the table answers with `bs_table` (queried with a huge ensemble so nothing clamps); the golden is what the reference then
    returns for real ensemble sizes.  torch.cuda is monkey-patched (no GPU here; the function only reads the total memory)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_batchsize", os.path.join(REF, "genpercept/util/batchsize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = []
    real_avail, real_info = torch.cuda.is_available, torch.cuda.mem_get_info
    try:
        torch.cuda.is_available = lambda: True
        for vram in (8, 11, 24, 40, 80, 288):
            torch.cuda.mem_get_info = lambda *a, v=vram: (0, v * 1024 ** 3)
            for res in (384, 512, 768, 1024, 2048):
                for dt_i, dt in enumerate((torch.float32, torch.float16)):
                    bs_table = mod.find_batch_size(10 ** 9, res, dt)
                    for ens in (1, 2, 3, 5, 7, 10, 16, 33, 50):
                        rows.append((vram, res, dt_i, ens, bs_table, mod.find_batch_size(ens, res, dt)))
        torch.cuda.is_available = lambda: False
        no_gpu = mod.find_batch_size(10, 768, torch.float32)
    finally:
        torch.cuda.is_available, torch.cuda.mem_get_info = real_avail, real_info
    np.savez_compressed(os.path.join(HERE, "batchsize_ref.npz"), rows=np.array(rows, dtype=np.int64), no_gpu=np.array(no_gpu),
                        columns=np.array(["vram_gb", "res", "dtype(0=f32,1=f16)", "ensemble", "bs_table", "bs_returned"]))
    print("batchsize golden:", len(rows), "rows; no-gpu ->", no_gpu)


def make_infer_eval_golden():
    """Naming modes of the REFERENCE's get_pred_name (src/dataset/base_dataset.py:531-545; cv2 / torchvision are stubbed, the function
    needs neither) and its alignment variants (src/util/alignment.py: max_resolution fit, disparity-space protocol of eval.py:181-200)."""
    import importlib.util
    for name in ("cv2", "torchvision", "torchvision.transforms", "tarfile_stub"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms"].InterpolationMode = types.SimpleNamespace(NEAREST=0, BILINEAR=1, BICUBIC=2, NEAREST_EXACT=3)
    sys.modules["torchvision.transforms"].Resize = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_base_dataset", os.path.join(REF, "src/dataset/base_dataset.py"))
    bd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bd)
    basenames = ["rgb_0001.png", "rgb_0250.jpg", "0000000005_rgb.png", "12_34_rgb.png", "frame.png", "rgb_1_2.png", "a_b_c_d.jpeg"]
    names = {}
    for mode in bd.PerceptionFileNameMode:
        for bn in basenames:
            for suf in (".npy", ".png"):
                try:
                    names[f"{mode.name}|{bn}|{suf}"] = bd.get_pred_name(bn, mode, suffix=suf)
                except Exception as e:  # e.g. IndexError for names without '_'
                    names[f"{mode.name}|{bn}|{suf}"] = "!" + type(e).__name__
    spec = importlib.util.spec_from_file_location("ref_alignment2", os.path.join(REF, "src/util/alignment.py"))
    al = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(al)
    rng = np.random.RandomState(5)
    gt = (rng.rand(97, 130) * 9 + 0.5).astype(np.float32)
    pred = (0.1 * gt + 0.05 + 0.02 * rng.randn(97, 130)).astype(np.float32)
    mask = rng.rand(97, 130) > 0.2
    out = {"gt": gt, "pred": pred, "mask": mask}
    for mr in (32, 64, 200):
        a, s_, t_ = al.align_depth_least_square(gt, pred, mask, return_scale_shift=True, max_resolution=mr)
        out[f"align_maxres{mr}"] = np.asarray(a, dtype=np.float64)
        out[f"align_maxres{mr}_st"] = np.array([float(s_), float(t_)])
    # disparity-space protocol (eval.py:181-200)
    gdisp, gpos = al.depth2disparity(gt, return_mask=True)
    pdisp = (1.0 / np.clip(gt, 1e-3, None) * 0.7 + 0.02 + 0.01 * rng.randn(97, 130)).astype(np.float32)
    m = mask & gpos & (pdisp > 0)
    d, s_, t_ = al.align_depth_least_square(gdisp, pdisp, m, return_scale_shift=True, max_resolution=None)
    out["pred_disp"] = pdisp
    out["disp_protocol_depth"] = np.asarray(al.disparity2depth(np.clip(d, 1e-3, None)), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "infer_eval_ref.npz"), name_keys=np.array(list(names.keys())), name_vals=np.array(list(names.values())), **out)
    print("infer/eval golden:", len(names), "names")


def make_datasets_golden():
    """Evaluation-side conventions of the REFERENCE's dataset classes beyond NYU (src/dataset/{kitti,eth3d,scannet,diode,nyu}_dataset.py;
    cv2 / torchvision stubbed, the code under test uses neither): depth ranges and naming modes, KITTI benchmark crop and Garg / Eigen
    evaluation masks, depth decoding (KITTI / 256, ScanNet and NYU / 1000, ETH3D raw float32 with inf -> 0, DIODE .npy + mask), and the
    angular error of genpercept/losses/geometry_losses.py:angular_loss, from which the normal evaluator is defined."""
    import importlib.util
    import tempfile
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms"].InterpolationMode = types.SimpleNamespace(NEAREST=0, BILINEAR=1, BICUBIC=2, NEAREST_EXACT=3)
    sys.modules["torchvision.transforms"].Resize = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    from src.dataset import base_dataset as bd
    from src.dataset.diode_dataset import DIODEDataset
    from src.dataset.eth3d_dataset import ETH3DDataset
    from src.dataset.kitti_dataset import KITTIDataset
    from src.dataset.nyu_dataset import NYUDataset
    from src.dataset.scannet_dataset import ScanNetDataset
    real_init = bd.BaseDataset.__init__
    bd.BaseDataset.__init__ = lambda self, **kw: self.__dict__.update(kw)  # keep only what the subclass passes up (ranges, naming mode)
    out = {}
    try:
        files = [["a.png", "a_d.png"], ["b.png", "None"], ["c.png", "c_d.png"]]
        insts = {"kitti": KITTIDataset(kitti_bm_crop=True, valid_mask_crop="eigen", filenames=[list(f) for f in files]),
                 "eth3d": ETH3DDataset(), "scannet": ScanNetDataset(), "diode": DIODEDataset(), "nyu": NYUDataset(eigen_valid_mask=True)}
        for k, d in insts.items():
            out[f"{k}_range"] = np.array([float(d.min_depth), float(d.max_depth)], dtype=np.float64)
            out[f"{k}_name_mode"] = np.array(d.name_mode.name)
        out["kitti_filtered"] = np.array([f[0] for f in insts["kitti"].filenames])
        # KITTI benchmark crop + evaluation masks on raw KITTI sizes
        sizes = [(375, 1242), (370, 1226), (376, 1241), (374, 1238), (352, 1216)]
        boxes, masks = [], {}
        for h, w in sizes:
            idx = torch.arange(h * w).reshape(h, w)
            c = KITTIDataset.kitti_benchmark_crop(idx)
            boxes.append([h, w, int(c[0, 0]) // w, int(c[0, 0]) % w, c.shape[0], c.shape[1]])
            depth = torch.full((1, c.shape[0], c.shape[1]), 5.0)
            depth[0, ::7, ::5] = 0.0      # invalid: below min_depth
            depth[0, 1::11, 2::9] = 90.0  # invalid: beyond max_depth
            for crop in ("eigen", "garg", None):
                k = insts["kitti"]
                k.valid_mask_crop = crop
                masks[f"kitti_mask_{crop}_{h}x{w}"] = np.packbits(k._get_valid_mask(depth).numpy().astype(np.uint8))
        out["kitti_crop_boxes"] = np.array(boxes)  # h, w, top, left, crop_h, crop_w
        out.update(masks)
        # decoders
        raw = (np.arange(6 * 8, dtype=np.float32).reshape(6, 8) * 37.0 + 5.0)
        for k in ("kitti", "scannet", "nyu"):
            d = insts[k]
            d.is_exr_data = False
            d._read_image = lambda rel, raw=raw: raw.copy()
            out[f"{k}_decoded"] = np.asarray(d._read_depth_file("x.png"), dtype=np.float64)
        out["raw_png_values"] = raw
        with tempfile.TemporaryDirectory() as td:
            e = insts["eth3d"]
            e.is_tar, e.tar_obj, e.dataset_dir, e.HEIGHT, e.WIDTH = False, None, td, 6, 8
            buf = raw.copy()
            buf[2, 3] = np.inf
            buf.tofile(os.path.join(td, "d.bin"))
            out["eth3d_decoded"] = e._read_depth_file("d.bin")
            di = insts["diode"]
            di.is_tar, di.tar_obj, di.dataset_dir = False, None, td
            np.save(os.path.join(td, "d.npy"), raw[:, :, None])
            out["diode_decoded"] = di._read_depth_file("d.npy")
    finally:
        bd.BaseDataset.__init__ = real_init
    # angular error (radians) of the reference's angular_loss on seeded normals
    spec = importlib.util.spec_from_file_location("ref_geometry_losses", os.path.join(REF, "genpercept/losses/geometry_losses.py"))
    gl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gl)
    g = torch.Generator().manual_seed(21)
    gt = torch.nn.functional.normalize(torch.randn(2, 3, 24, 32, generator=g), dim=1)
    pred = torch.nn.functional.normalize(gt + 0.3 * torch.randn(2, 3, 24, 32, generator=g), dim=1)
    pred[0, :, 0, :4] = gt[0, :, 0, :4]          # exact agreement: exercises the clamp at 1 - eps
    pred[1, :, 1, :4] = -gt[1, :, 1, :4]         # opposite normals: clamp at -1 + eps
    mask = (torch.rand(2, 1, 24, 32, generator=g) > 0.25)
    out.update(normal_gt=gt.numpy(), normal_pred=pred.numpy(), normal_mask=mask.numpy(),
               angular_loss_mean_rad=np.array(float(gl.angular_loss(pred, gt, mask.float()))))
    np.savez_compressed(os.path.join(HERE, "datasets_ref.npz"), **out)
    print("datasets_ref.npz", len(out), "entries,", os.path.getsize(os.path.join(HERE, "datasets_ref.npz")) // 1024, "KiB")


def make_image_util_golden():
    """The REFERENCE's genpercept/util/image_util.py: colorize_depth_maps (matplotlib Spectral), chw2hwc, the output-size rule of
    resize_max_res (int() truncation; torchvision's `resize` is replaced by a stub that records the requested size -- the resampling
    itself is torchvision's, not the reference's) and get_tv_resample_method's name handling."""
    import importlib.util
    tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")

    class IM:  # stand-in enum members
        BILINEAR, BICUBIC, NEAREST, NEAREST_EXACT = "bilinear", "bicubic", "nearest", "nearest-exact"
    requested = []

    def fake_resize(img, size, interpolation=None, antialias=None):
        requested.append((tuple(img.shape[-2:]), tuple(size), interpolation, antialias))
        return img
    tvt.InterpolationMode = IM
    tvf.resize = fake_resize
    tv.transforms, tvt.functional = tvt, tvf
    saved = {k: sys.modules.get(k) for k in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional")}
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    try:
        spec = importlib.util.spec_from_file_location("ref_image_util", os.path.join(REF, "genpercept/util/image_util.py"))
        iu = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(iu)
        rng = np.random.RandomState(3)
        depth = rng.rand(2, 13, 17).astype(np.float32) * 1.4 - 0.2          # values outside [0,1] exercise the clip
        out = {"depth": depth, "colored_spectral": iu.colorize_depth_maps(depth, 0, 1, cmap="Spectral"),
               "colored_single": iu.colorize_depth_maps(depth[0], 0.1, 0.9, cmap="Spectral"),
               "chw": rng.rand(3, 5, 7).astype(np.float32)}
        out["hwc"] = iu.chw2hwc(out["chw"])
        sizes = []
        for (h, w) in [(480, 640), (768, 1024), (1000, 333), (375, 1242), (64, 64), (100, 768), (1080, 1920), (513, 511)]:
            for mr in (768, 512, 384):
                requested.clear()
                iu.resize_max_res(torch.zeros(1, 3, h, w), mr)
                (_, size, interp, aa) = requested[0]
                assert aa is True and interp == "bilinear"
                sizes.append((h, w, mr, size[0], size[1]))
        out["resize_sizes"] = np.array(sizes, dtype=np.int64)
        names = {}
        for nm in ("bilinear", "bicubic", "nearest", "nearest-exact", "lanczos", ""):
            try:
                names[nm] = str(iu.get_tv_resample_method(nm))
            except ValueError:
                names[nm] = "!ValueError"
        out["resample_keys"] = np.array(list(names.keys()))
        out["resample_vals"] = np.array(list(names.values()))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    np.savez_compressed(os.path.join(HERE, "image_util_ref.npz"), **out)
    print("image_util golden:", len(sizes), "size cases;", dict(names))


def _load_ref_module(name, rel):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make_scheduler_golden():
    import glob
    import json
    d, dc = types.ModuleType("diffusers"), types.ModuleType("diffusers.configuration_utils")

    class _Base:  # DDIMScheduler / DDPMScheduler stand-ins: the customised classes override __init__ and never call super()
        pass
    d.DDIMScheduler, d.DDPMScheduler = _Base, type("DDPMScheduler", (), {})
    dc.ConfigMixin, dc.register_to_config = object, (lambda f: f)
    d.configuration_utils = dc
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.configuration_utils")}
    sys.modules.update({"diffusers": d, "diffusers.configuration_utils": dc})
    try:
        ddim = _load_ref_module("ref_ddim", "src/customized_modules/ddim.py")
        out, names = {}, []
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "trained_betas", "clip_sample", "set_alpha_to_one",
                "steps_offset", "prediction_type", "thresholding", "dynamic_thresholding_ratio", "clip_sample_range", "sample_max_value",
                "timestep_spacing", "rescale_betas_zero_snr")
        cfgs = {}
        for f in sorted(glob.glob(os.path.join(REF, "hf_configs/scheduler_*/scheduler_config.json"))):
            cfgs[os.path.basename(os.path.dirname(f))] = {k: v for k, v in json.load(open(f)).items() if k in keys}
        cfgs["power_2.0"] = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear_power", power_beta_curve=2.0,
                                 set_alpha_to_one=False, prediction_type="v_prediction", clip_sample=False, steps_offset=1)
        cfgs["power_3.0_zero_snr"] = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear_power", power_beta_curve=3.0,
                                          rescale_betas_zero_snr=True, set_alpha_to_one=True)
        pairs = [(999, 899), (501, 451), (51, 1), (1, -49), (20, 19), (981, 961)]
        for name, cfg in cfgs.items():
            s = ddim.DDIMSchedulerCustomized(**cfg)
            names.append(name)
            out[name + "/cfg"] = np.array(json.dumps(cfg))
            out[name + "/betas"] = s.betas.numpy()
            out[name + "/alphas_cumprod"] = s.alphas_cumprod.numpy()
            out[name + "/final_alpha_cumprod"] = np.float32(s.final_alpha_cumprod)
            out[name + "/variance"] = np.array([float(s._get_variance(t, p)) for t, p in pairs], dtype=np.float64)
        out["names"] = np.array(names)
        out["variance_pairs"] = np.array(pairs, dtype=np.int64)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None) if v is None else sys.modules.__setitem__(k, v)
    np.savez_compressed(os.path.join(HERE, "scheduler_ref.npz"), **out)
    print("scheduler_ref.npz", names, os.path.getsize(os.path.join(HERE, "scheduler_ref.npz")) // 1024, "KiB")


def make_ensemble_golden():
    tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")

    class IM:
        BILINEAR, BICUBIC, NEAREST, NEAREST_EXACT = "bilinear", "bicubic", "nearest", "nearest-exact"

    def no_resize(*a, **k):
        raise AssertionError("the ensemble fixtures stay below max_res: torchvision must not be reached")
    tvt.InterpolationMode, tvf.resize = IM, no_resize
    tv.transforms, tvt.functional = tvt, tvf
    pk, pu = types.ModuleType("ref_gp"), types.ModuleType("ref_gp.util")
    pk.__path__, pu.__path__ = [os.path.join(REF, "genpercept")], [os.path.join(REF, "genpercept/util")]
    names = ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "ref_gp", "ref_gp.util")
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf, "ref_gp": pk, "ref_gp.util": pu})
    try:
        import importlib
        ens = importlib.import_module("ref_gp.util.ensemble")
        rng = np.random.RandomState(11)
        out, cases = {}, []
        yy, xx = np.mgrid[0:40, 0:48].astype(np.float32)
        base = 0.5 + 0.35 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.1 * (yy / 40.0)
        for tag, e, kw in (("affine_med5", 5, {}), ("affine_med4", 4, {}), ("affine_mean3", 3, dict(reduction="mean")),
                           ("scale_med6", 6, dict(shift_invariant=False)), ("affine_unc7", 7, dict(output_uncertainty=True)),
                           ("affine_iter8", 5, dict(max_iter=8, tol=1e-5, regularizer_strength=0.05)), ("single", 1, {})):
            s = rng.uniform(0.6, 1.6, size=(e, 1, 1, 1)).astype(np.float32)
            t = rng.uniform(-0.2, 0.3, size=(e, 1, 1, 1)).astype(np.float32) * (0 if kw.get("shift_invariant") is False else 1)
            d = (base[None, None] * s + t + rng.normal(0, 0.02, size=(e, 1, 40, 48))).astype(np.float32)
            if kw.get("shift_invariant") is False:
                d = np.abs(d)
            pred, unc = ens.ensemble_depth(torch.from_numpy(d), scale_invariant=True, max_res=50, **{"shift_invariant": True, **kw})
            out[tag + "/in"], out[tag + "/pred"] = d, pred.numpy()
            if unc is not None:
                out[tag + "/unc"] = unc.numpy()
            import json
            out[tag + "/kw"] = np.array(json.dumps(kw))
            cases.append(tag)
        out["cases"] = np.array(cases)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None) if v is None else sys.modules.__setitem__(k, v)
    np.savez_compressed(os.path.join(HERE, "ensemble_ref.npz"), **out)
    print("ensemble_ref.npz", cases, os.path.getsize(os.path.join(HERE, "ensemble_ref.npz")) // 1024, "KiB")


MULTISTEP_SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                       steps_offset=1, prediction_type="v_prediction", timestep_spacing="leading")  # hf_configs/scheduler_beta_0.00085_0.012


def make_e2e_multistep():
    vc = osd.VAECfg.tiny()
    uc4 = osd.UNetCfg.tiny()
    uc8 = osd.UNetCfg(in_channels=8, block_out_channels=uc4.block_out_channels, num_heads=uc4.num_heads, cross_attention_dim=uc4.cross_attention_dim)
    vae_sd = osd.synth_state_dict(osd.vae_manifest(vc), seed=2)
    unet4 = osd.synth_state_dict(osd.unet_manifest(uc4), seed=1)
    unet8 = opipe.replace_unet_conv_in(unet4)  # run.py:59-78 on the same seeded 4-channel UNet
    g = torch.Generator().manual_seed(77)
    ctx = torch.randn(2, uc4.cross_attention_dim, generator=g)
    out = {"ctx": ctx.numpy()}
    with torch.no_grad():
        for tag, (h, w) in (("sq", (64, 64)), ("odd", (72, 88))):
            rgb = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8)
            noise = torch.randn(2, 4, h // 8, w // 8, generator=g)
            out[f"{tag}_rgb"], out[f"{tag}_noise"] = rgb.numpy(), noise.numpy()
            x = opipe.normalize_rgb(rgb)
            for steps in (1, 4, 10):
                out[f"{tag}_marigold_{steps}"] = opipe.multi_step_infer(vae_sd, vc, unet8, uc8, x, ctx, "depth", opipe.DDIM(**MULTISTEP_SCHED), steps, noise).numpy()
                out[f"{tag}_blend_{steps}"] = opipe.multi_step_infer(vae_sd, vc, unet4, uc4, x, ctx, "depth", opipe.DDIM(**MULTISTEP_SCHED), steps).numpy()
            out[f"{tag}_blend_normal_4"] = opipe.multi_step_infer(vae_sd, vc, unet4, uc4, x, ctx, "normal", opipe.DDIM(**MULTISTEP_SCHED), 4).numpy()
            out[f"{tag}_blend_fix_3"] = opipe.multi_step_infer(vae_sd, vc, unet4, uc4, x, ctx, "depth", opipe.DDIM(**MULTISTEP_SCHED), 3, fix_timesteps=400).numpy()
            eps_cfg = dict(MULTISTEP_SCHED, prediction_type="epsilon", clip_sample=True, set_alpha_to_one=True, steps_offset=0, timestep_spacing="trailing")
            out[f"{tag}_marigold_eps_4"] = opipe.multi_step_infer(vae_sd, vc, unet8, uc8, x, ctx, "depth", opipe.DDIM(**eps_cfg), 4, noise).numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_multistep.npz"), **out)
    print("e2e_multistep.npz", os.path.getsize(os.path.join(HERE, "e2e_multistep.npz")) // 1024, "KiB")


def make_e2e_tiny():
    uc, vc, dc = osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg.tiny()
    usd = osd.synth_state_dict(osd.unet_manifest(uc), 1)
    vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
    dsd = osd.synth_state_dict(odpt.dpt_manifest(dc), 3)
    g = torch.Generator().manual_seed(1234)
    out = {}
    for tag, (b, h, w) in {"sq": (2, 64, 64), "odd": (1, 72, 88)}.items():
        noise = torch.randint(0, 256, (b, 3, h, w), generator=g, dtype=torch.uint8).float()
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        smooth = torch.stack([yy, xx, (yy + xx) / 2])[None] * 255.0
        rgb_u8 = (0.5 * noise + 0.5 * smooth).round().clamp(0, 255).to(torch.uint8)
        ctx = torch.randn(2, uc.cross_attention_dim, generator=g)
        rgb = opipe.normalize_rgb(rgb_u8)
        with torch.no_grad():
            lat = osd.encode_rgb(vsd, vc, rgb)
            v, feats = osd.unet_forward(usd, uc, lat, 1, ctx[None].expand(b, -1, -1))
            dec3 = osd.decode_pred(vsd, vc, -v, "normal")
            depth = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "depth")
            normal = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "normal")
            disp = opipe.single_infer(vsd, vc, usd, uc, rgb, ctx, "disparity", dpt_sd=dsd)
        out[f"{tag}_rgb_u8"] = rgb_u8.numpy()
        out[f"{tag}_ctx"] = ctx.numpy()
        out[f"{tag}_latent"] = lat.numpy()
        out[f"{tag}_unet"] = v.numpy()
        for i, f in enumerate(feats):
            out[f"{tag}_feat{i}"] = f.numpy().astype(np.float16)
        out[f"{tag}_dec3"] = dec3.numpy()
        out[f"{tag}_depth"] = depth.numpy()
        out[f"{tag}_normal"] = normal.numpy()
        out[f"{tag}_disp"] = disp.numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_tiny.npz"), seeds=np.array([1, 2, 3]), **out)
    print("e2e_tiny.npz", os.path.getsize(os.path.join(HERE, "e2e_tiny.npz")) // 1024, "KiB")


# ---------------------------------------------------------------------------------------------------------------------------
# refexec_tiny.npz: the REFERENCE's own orchestration code, executed.  diffusers is absent, so the inside of the
# diffusers blocks stays the oracle's restatement -- but everything the reference's tree itself holds for the hot path now RUNS instead of
# being restated:  CustomUNet2DConditionModel.forward (custom_unet.py:34-427: time embedding call order, skip stack, popping order
# `down_block_res_samples[-len(resnets):]`, forward_upsample_size / upsample_size, multi_level_feats, return_feature) on a stub
# `UNet2DConditionModel` whose blocks call oracle.sd21's resnet_block / transformer_2d;  GenPerceptPipeline.__call__ / single_infer /
# encode_rgb / decode_pred (genpercept_pipeline.py:146-337,375-526: normalisation, latent scale, [rgb_latent, pred_latent] concat order,
# `pred_original_sample`, channel mean, clip / shift, `multi_level_feats[::-1]`, `[:, None]`, min-max, squeeze / clip / colourise) on a
# stub `DiffusionPipeline`, a stub VAE (oracle encoder / decoder halves) and the REAL DPTNeckHeadForUnetAfterUpsampleIdentity.
# The scheduler's `step` / `set_timesteps` are diffusers' (un-vendored): the stub delegates them to oracle.pipeline.DDIM.
# ---------------------------------------------------------------------------------------------------------------------------
def _stub_refexec_modules():
    """stand-ins for the import lines of custom_unet.py:11-17, genpercept_pipeline.py:22-36, dpt_head.py:20-21, image_util.py:21-22"""
    import dataclasses

    class BaseOutput(dict):  # diffusers' BaseOutput: an OrderedDict whose items are also attributes (dataclass subclasses AND plain keyword construction)
        def __init__(self, *a, **k):
            super().__init__()
            for key, v in dict(*a, **k).items():
                self[key] = v

        def __setitem__(self, key, v):
            super().__setitem__(key, v)
            object.__setattr__(self, key, v)

    class UNet2DConditionModel(torch.nn.Module):
        pass

    class DiffusionPipeline:
        def __init__(self):
            self._cfg = {}

        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        def register_to_config(self, **kw):
            self._cfg.update(kw)

        device = property(lambda self: torch.device("cpu"))
        dtype = property(lambda self: torch.float32)

    class DDIMScheduler:
        pass

    class LCMScheduler:
        pass

    class AutoencoderKL:
        pass

    mods = {n: types.ModuleType(n) for n in (
        "diffusers", "diffusers.utils", "diffusers.models", "diffusers.models.lora", "diffusers.models.unets",
        "diffusers.models.unets.unet_2d_condition", "torchvision", "torchvision.transforms", "torchvision.transforms.functional")}
    d, du = mods["diffusers"], mods["diffusers.utils"]
    d.AutoencoderKL, d.DDIMScheduler, d.DiffusionPipeline, d.LCMScheduler, d.UNet2DConditionModel = (
        AutoencoderKL, DDIMScheduler, DiffusionPipeline, LCMScheduler, UNet2DConditionModel)
    du.BaseOutput, du.USE_PEFT_BACKEND = BaseOutput, True
    du.deprecate = lambda *a, **k: None
    du.logging = types.SimpleNamespace(get_logger=lambda *a, **k: None)
    du.scale_lora_layers = du.unscale_lora_layers = lambda *a, **k: None
    mods["diffusers.models.lora"].LoRACompatibleConv = torch.nn.Conv2d
    mods["diffusers.models.unets.unet_2d_condition"].UNet2DConditionOutput = dataclasses.make_dataclass("UNet2DConditionOutput", [("sample", object, None)])

    class InterpolationMode:
        BILINEAR, BICUBIC, NEAREST_EXACT = "bilinear", "bicubic", "nearest-exact"

    def _no_tv(*a, **k):
        raise RuntimeError("torchvision is not installed: the fixtures use processing_res=0 / match_input_res=False")

    mods["torchvision.transforms"].InterpolationMode = InterpolationMode
    mods["torchvision.transforms.functional"].pil_to_tensor = _no_tv
    mods["torchvision.transforms.functional"].resize = _no_tv
    mods["torchvision"].transforms = mods["torchvision.transforms"]
    mods["torchvision.transforms"].functional = mods["torchvision.transforms.functional"]
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    return saved, d


def _refexec_import():
    """import the reference's package (NOT this repository's shim) under the stubs; returns (pipeline module, custom_unet module, dpt module)"""
    import importlib
    # transformers probes `torchvision` through importlib.util.find_spec at import time: load everything the reference takes from it first
    from transformers import CLIPTextModel, CLIPTokenizer, DPTConfig, DPTPreTrainedModel  # noqa: F401
    import transformers.models.dpt.modeling_dpt  # noqa: F401
    saved, d = _stub_refexec_modules()
    path0 = list(sys.path)
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    for name in [m for m in sys.modules if m == "genpercept" or m.startswith("genpercept.")]:
        del sys.modules[name]
    try:
        gpp = importlib.import_module("genpercept.genpercept_pipeline")
        cu = importlib.import_module("genpercept.models.custom_unet")
        dh = importlib.import_module("genpercept.models.dpt_head")
    finally:
        sys.path[:] = path0
    assert gpp.__file__.startswith(REF) and cu.__file__.startswith(REF)
    return gpp, cu, dh, d, saved


def _refexec_cleanup(saved):
    for name in [m for m in sys.modules if m == "genpercept" or m.startswith("genpercept.")]:
        del sys.modules[name]
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _refexec_unet(cu, sd, cfg):
    """an instance of the REFERENCE's CustomUNet2DConditionModel whose sub-modules are the oracle's functions over the state dict `sd`"""
    g = cfg.norm_num_groups

    class Down:
        def __init__(self, i):
            self.i, self.has_cross_attention = i, cfg.down_has_attn[i]

        def __call__(self, hidden_states, temb, encoder_hidden_states=None, **kw):
            x, outs, i = hidden_states, (), self.i
            for j in range(cfg.layers_per_block):
                x = osd.resnet_block(x, sd, f"down_blocks.{i}.resnets.{j}", g, cfg.norm_eps, temb)
                if self.has_cross_attention:
                    x = osd.transformer_2d(x, sd, f"down_blocks.{i}.attentions.{j}", cfg.num_heads[i], encoder_hidden_states, g)
                outs += (x,)
            if i != len(cfg.block_out_channels) - 1:
                x = osd._conv(x, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)
                outs += (x,)
            return x, outs

    class Mid:
        has_cross_attention = True

        def __call__(self, x, temb, encoder_hidden_states=None, **kw):
            x = osd.resnet_block(x, sd, "mid_block.resnets.0", g, cfg.norm_eps, temb)
            x = osd.transformer_2d(x, sd, "mid_block.attentions.0", cfg.num_heads[-1], encoder_hidden_states, g)
            return osd.resnet_block(x, sd, "mid_block.resnets.1", g, cfg.norm_eps, temb)

    class Up:
        def __init__(self, i, blk):
            self.i, self.blk, self.has_cross_attention = i, blk, blk["attn"]
            self.resnets = [None] * len(blk["resnets"])  # custom_unet.py:372 takes len() of it

        def __call__(self, hidden_states, temb, res_hidden_states_tuple, encoder_hidden_states=None, upsample_size=None, **kw):
            x, i = hidden_states, self.i
            for j in range(len(self.resnets)):
                res = res_hidden_states_tuple[-1]                     # diffusers' up blocks pop the tuple from the END
                res_hidden_states_tuple = res_hidden_states_tuple[:-1]
                x = torch.cat([x, res], dim=1)
                x = osd.resnet_block(x, sd, f"up_blocks.{i}.resnets.{j}", g, cfg.norm_eps, temb)
                if self.has_cross_attention:
                    x = osd.transformer_2d(x, sd, f"up_blocks.{i}.attentions.{j}", self.blk["heads"], encoder_hidden_states, g)
            if self.blk["upsample"]:
                x = (torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest") if upsample_size is None
                     else torch.nn.functional.interpolate(x, size=tuple(upsample_size), mode="nearest"))
                x = osd._conv(x, sd, f"up_blocks.{i}.upsamplers.0.conv")
            return x

    m = cu.CustomUNet2DConditionModel()
    m.config = types.SimpleNamespace(center_input_sample=False, class_embed_type=None, addition_embed_type=None, encoder_hid_dim_type=None,
                                     class_embeddings_concat=False)
    m.num_upsamplers = len(cfg.block_out_channels) - 1
    m.time_proj = lambda t: osd.timestep_embedding(t, cfg.block_out_channels[0])
    m.time_embedding = lambda te, cond=None: osd._linear(torch.nn.functional.silu(osd._linear(te, sd, "time_embedding.linear_1")), sd, "time_embedding.linear_2")
    m.class_embedding = m.time_embed_act = m.encoder_hid_proj = None
    m.conv_in = lambda x: osd._conv(x, sd, "conv_in")
    m.down_blocks = [Down(i) for i in range(len(cfg.block_out_channels))]
    m.mid_block = Mid()
    m.up_blocks = [Up(i, blk) for i, blk in enumerate(osd.unet_up_plan(cfg))]
    has_out = "conv_out.weight" in sd
    m.conv_norm_out = (lambda x: osd._gn(x, sd, "conv_norm_out", g, cfg.norm_eps)) if has_out else None
    m.conv_act = torch.nn.functional.silu
    m.conv_out = (lambda x: osd._conv(x, sd, "conv_out")) if has_out else None
    return m


def make_refexec_golden():
    import json
    gpp, cu, dh, d, saved = _refexec_import()
    try:
        uc, vc, dc = osd.UNetCfg.tiny(), osd.VAECfg.tiny(), odpt.DPTCfg.tiny()
        usd = osd.synth_state_dict(osd.unet_manifest(uc), 1)      # the seeds / inputs of e2e_tiny.npz
        vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
        dsd = osd.synth_state_dict(odpt.dpt_manifest(dc), 3)
        usd_noout = {k: v for k, v in usd.items() if not k.startswith(("conv_out", "conv_norm_out"))}
        e2e = np.load(os.path.join(HERE, "e2e_tiny.npz"))

        class VAE:  # genpercept_pipeline.py:500-501,521-522 call vae.encoder / quant_conv / post_quant_conv / decoder separately
            @staticmethod
            def encoder(x):
                sd_ = dict(vsd)
                sd_["quant_conv.weight"] = torch.eye(2 * vc.latent_channels)[:, :, None, None]
                sd_["quant_conv.bias"] = torch.zeros(2 * vc.latent_channels)
                return osd.vae_encode_moments(sd_, vc, x)          # identity quant_conv: the encoder's own output

            quant_conv = staticmethod(lambda h: osd._conv(h, vsd, "quant_conv", padding=0))
            post_quant_conv = staticmethod(lambda z: osd._conv(z, vsd, "post_quant_conv", padding=0))

            @staticmethod
            def decoder(z):
                sd_ = dict(vsd)
                sd_["post_quant_conv.weight"] = torch.eye(vc.latent_channels)[:, :, None, None]
                sd_["post_quant_conv.bias"] = torch.zeros(vc.latent_channels)
                return osd.vae_decode(sd_, vc, z)

        class Sched(d.DDIMScheduler):  # diffusers' step / set_timesteps are un-vendored: oracle.pipeline.DDIM stands in
            def __init__(self, **cfg):
                self.o = opipe.DDIM(**cfg)
                self.beta_start, self.beta_end = cfg["beta_start"], cfg["beta_end"]

            def set_timesteps(self, n, device=None):
                self.o.set_timesteps(n)
                self.timesteps = torch.as_tensor(np.asarray(self.o.timesteps).copy())

            def step(self, model_output, t, sample, generator=None):
                prev, x0 = self.o.step(model_output, int(t), sample)
                return types.SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

        with open(os.path.join(REF, "hf_configs/dpt-sd2.1-unet-after-upsample-general/config.json")) as f:
            from transformers import DPTConfig
            hc = json.load(f)
        hc.update(neck_hidden_sizes=list(dc.neck_hidden_sizes), fusion_hidden_size=dc.fusion_hidden_size)
        import dataclasses
        if not dataclasses.is_dataclass(dh.DepthEstimatorOutput):  # (the installed transformers wants ModelOutput subclasses decorated; dpt_head.py:24 is not)
            dataclasses.dataclass(dh.DepthEstimatorOutput)
        head = dh.DPTNeckHeadForUnetAfterUpsampleIdentity(DPTConfig(**hc)).eval()
        assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == {k: tuple(v) for k, v in odpt.dpt_manifest(dc).items()}
        head.load_state_dict(dsd, strict=True)

        beta1 = dict(beta_start=1.0, beta_end=1.0, beta_schedule="linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                     prediction_type="v_prediction", timestep_spacing="leading")  # hf_configs/scheduler_beta_1.0_1.0
        out = {}
        with torch.no_grad():
            for tag in ("sq", "odd"):
                rgb_u8 = torch.as_tensor(e2e[f"{tag}_rgb_u8"])
                ctx = torch.as_tensor(e2e[f"{tag}_ctx"])
                b = rgb_u8.shape[0]
                # --- the reference's UNet forward alone (golden latent in) --------------------------------------------------------
                lat = torch.as_tensor(e2e[f"{tag}_latent"])
                unet = _refexec_unet(cu, usd, uc)
                r = unet(lat, 1, encoder_hidden_states=ctx[None].expand(b, -1, -1))
                out[f"{tag}_unet"] = r.sample.numpy()
                for i, f_ in enumerate(r.multi_level_feats):
                    out[f"{tag}_feat{i}"] = f_.numpy().astype(np.float16)
                rf = _refexec_unet(cu, usd_noout, uc)(lat, torch.tensor([1]), encoder_hidden_states=ctx[None].expand(b, -1, -1), return_feature=True)
                assert rf.sample is None and all(torch.equal(a, c) for a, c in zip(rf.multi_level_feats, r.multi_level_feats))
                ov, of = osd.unet_forward(usd, uc, lat, 1, ctx[None].expand(b, -1, -1))
                assert torch.equal(ov, r.sample) and all(torch.equal(a, c) for a, c in zip(of, r.multi_level_feats)), "oracle.unet_forward != reference forward"
                # --- the reference's pipeline: single_infer and __call__ ----------------------------------------------------------
                def pipe(head_=None, unet_sd=usd, sched=beta1, **kw):
                    p = gpp.GenPerceptPipeline(unet=_refexec_unet(cu, unet_sd, uc), vae=VAE, scheduler=Sched(**sched), text_encoder=None, tokenizer=None,
                                               customized_head=head_, **kw)
                    p.text_embed = ctx[None]  # genpercept_pipeline.py:425-429: cached embedding, encode_text is skipped
                    return p
                rgb_norm = rgb_u8.float() / 255.0 * 2.0 - 1.0
                for mode in ("depth", "normal"):
                    p = pipe()
                    p.mode = mode
                    out[f"{tag}_{mode}"] = p.single_infer(rgb_norm, 1, None, False).numpy()
                    # (not bitwise: the pipeline calls vae.encoder / quant_conv / post_quant_conv / decoder as four modules, the oracle's VAE
                    #  functions fold them into two calls -- a few fp32 ulps of the [0,1] map)
                    np.testing.assert_allclose(out[f"{tag}_{mode}"], e2e[f"{tag}_{mode}"], rtol=0, atol=1e-5, err_msg=f"oracle single_infer != reference ({tag}, {mode})")
                p = pipe(head_=head, unet_sd=usd_noout)
                p.mode = "disparity"
                out[f"{tag}_disp"] = torch.cat([p.single_infer(rgb_norm[i:i + 1], 1, None, False) for i in range(b)]).numpy()  # one image per call (F10)
                np.testing.assert_allclose(out[f"{tag}_disp"], e2e[f"{tag}_disp"], rtol=0, atol=1e-5)  # (reference head = nn modules, oracle = functional)
                out[f"{tag}_latent"] = p.encode_rgb(rgb_norm).numpy()
                np.testing.assert_allclose(out[f"{tag}_latent"], e2e[f"{tag}_latent"], rtol=0, atol=1e-5)
                # __call__ (processing_res=0, match_input_res=False: the two torchvision calls are not reached), image 0
                for mode, cmap in (("depth", "Spectral"), ("normal", None)):
                    o = pipe()(rgb_u8[:1], denoising_steps=1, ensemble_size=1, processing_res=0, match_input_res=False, batch_size=1, color_map=cmap,
                               show_progress_bar=False, mode=mode)
                    out[f"{tag}_call_{mode}_np"] = o.pred_np
                    out[f"{tag}_call_{mode}_colored"] = np.asarray(o.pred_colored)
                o = pipe(head_=head, unet_sd=usd_noout)(rgb_u8[:1], denoising_steps=1, ensemble_size=1, processing_res=0, match_input_res=False, batch_size=1,
                                                        color_map="Spectral", show_progress_bar=False, mode="disparity")
                out[f"{tag}_call_disp_np"], out[f"{tag}_call_disp_colored"] = o.pred_np, np.asarray(o.pred_colored)
                # fix_timesteps (genpercept_pipeline.py:405-408)
                p = pipe()
                p.mode = "depth"
                out[f"{tag}_depth_fix400"] = p.single_infer(rgb_norm, 1, None, False, fix_timesteps=400).numpy()
            # --- multi-step archs through the reference's loop (:413-422,447-465): marigold (noise + 8-channel conv_in), rgb_blending ------
            ms = np.load(os.path.join(HERE, "e2e_multistep.npz"))
            ctx = torch.as_tensor(ms["ctx"])
            uc8 = osd.UNetCfg(in_channels=8, block_out_channels=uc.block_out_channels, num_heads=uc.num_heads, cross_attention_dim=uc.cross_attention_dim)
            unet8 = opipe.replace_unet_conv_in(usd)
            for tag in ("sq", "odd"):
                rgb_norm = torch.as_tensor(ms[f"{tag}_rgb"]).float() / 255.0 * 2.0 - 1.0
                noise = torch.as_tensor(ms[f"{tag}_noise"])

                class FixedNoise:  # stands in for torch.randn(..., generator=g) of :416-420: the fixture's noise tensor
                    pass
                def run(unet_sd, ucfg, blending, steps):
                    p = gpp.GenPerceptPipeline(unet=_refexec_unet(cu, unet_sd, ucfg), vae=VAE, scheduler=Sched(**MULTISTEP_SCHED), text_encoder=None,
                                               tokenizer=None, genpercept_pipeline=False, rgb_blending=blending)
                    p.text_embed, p.mode = ctx[None], "depth"
                    if blending:
                        return p.single_infer(rgb_norm, steps, None, False).numpy()
                    real_randn = torch.randn
                    try:
                        gpp.torch.randn = lambda *a, **k: noise.clone()
                        return p.single_infer(rgb_norm, steps, None, False).numpy()
                    finally:
                        gpp.torch.randn = real_randn
                for steps in (1, 4):
                    out[f"{tag}_marigold_{steps}"] = run(unet8, uc8, False, steps)
                    out[f"{tag}_blend_{steps}"] = run(usd, uc, True, steps)
                    np.testing.assert_allclose(out[f"{tag}_marigold_{steps}"], ms[f"{tag}_marigold_{steps}"], rtol=0, atol=1e-5)
                    np.testing.assert_allclose(out[f"{tag}_blend_{steps}"], ms[f"{tag}_blend_{steps}"], rtol=0, atol=1e-5)
    finally:
        _refexec_cleanup(saved)
    np.savez_compressed(os.path.join(HERE, "refexec_tiny.npz"), **out)
    print("refexec_tiny.npz", os.path.getsize(os.path.join(HERE, "refexec_tiny.npz")) // 1024, "KiB;", len(out), "arrays; oracle == reference-executed on every stage")


# ---------------------------------------------------------------------------------------------------------------------------
# refexec_v1_tiny.npz: the SECOND statement of the one-step math the reference's tree holds -- GenPercept_v1/genpercept/pipeline_genpercept.py
# `single_infer` (:263-309: `timesteps = torch.tensor([1])`, `pred_latent = - unet_pred`, no scheduler object at all), `encode_rgb` (:312-336) and
# `decode_pred` (:338-356) -- EXECUTED over the same stub bases as refexec_tiny.npz.  Its text context is the file the v1 tree
# ships, GenPercept_v1/empty_text_embed.npy ([77, 1024] fp16: CLIP's embedding of the empty prompt padded to 77 tokens): used whole (what v1 does)
# and as rows [0:2] (BOS, EOS: what v2's `padding="do_not_pad"` tokenisation produces, genpercept_pipeline.py:360-372 -- SURVEY F6's "[B, 2, 1024]").
# The UNet is the tiny topology with cross_attention_dim = 1024 so that the real embedding rows are its context.
# ---------------------------------------------------------------------------------------------------------------------------
V1 = os.path.join(REF, "GenPercept_v1")


def make_refexec_v1_golden():
    import importlib
    gpp2, cu, dh, d, saved = _refexec_import()        # v2's custom_unet.py supplies the executed UNet forward (v1 calls a plain diffusers UNet)
    path0 = list(sys.path)
    try:
        for name in [m for m in sys.modules if m == "genpercept" or m.startswith("genpercept.")]:
            if name != "genpercept.models.custom_unet":
                del sys.modules[name]
        sys.path[:] = [V1] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
        v1 = importlib.import_module("genpercept.pipeline_genpercept")
        assert v1.__file__.startswith(V1)
        sys.path[:] = path0
        embed = np.load(os.path.join(V1, "empty_text_embed.npy"))
        assert embed.shape == (77, 1024) and embed.dtype == np.float16
        uc = osd.UNetCfg(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=1024)
        vc = osd.VAECfg.tiny()
        usd = osd.synth_state_dict(osd.unet_manifest(uc), 21)
        vsd = osd.synth_state_dict(osd.vae_manifest(vc), 2)
        e2e = np.load(os.path.join(HERE, "e2e_tiny.npz"))

        class VAE:  # pipeline_genpercept.py:327-328,352-353 call vae.encoder / quant_conv / post_quant_conv / decoder separately
            @staticmethod
            def encoder(x):
                sd_ = dict(vsd)
                sd_["quant_conv.weight"] = torch.eye(2 * vc.latent_channels)[:, :, None, None]
                sd_["quant_conv.bias"] = torch.zeros(2 * vc.latent_channels)
                return osd.vae_encode_moments(sd_, vc, x)

            quant_conv = staticmethod(lambda h: osd._conv(h, vsd, "quant_conv", padding=0))
            post_quant_conv = staticmethod(lambda z: osd._conv(z, vsd, "post_quant_conv", padding=0))

            @staticmethod
            def decoder(z):
                sd_ = dict(vsd)
                sd_["post_quant_conv.weight"] = torch.eye(vc.latent_channels)[:, :, None, None]
                sd_["post_quant_conv.bias"] = torch.zeros(vc.latent_channels)
                return osd.vae_decode(sd_, vc, z)

        out = {"embed_rows_0_2": embed[:2].copy(), "embed_77": embed.copy()}
        with torch.no_grad():
            for tag in ("sq", "odd"):
                rgb_u8 = torch.as_tensor(e2e[f"{tag}_rgb_u8"])
                rgb_norm = rgb_u8.float() / 255.0 * 2.0 - 1.0
                out[f"{tag}_rgb_u8"] = rgb_u8.numpy()
                for cname, ctx in (("ctx2", torch.as_tensor(embed[:2]).float()), ("ctx77", torch.as_tensor(embed).float())):
                    for mode in ("depth", "normal"):
                        p = v1.GenPerceptPipeline(unet=_refexec_unet(cu, usd, uc), vae=VAE, customized_head=None, empty_text_embed=ctx[None])
                        pred = p.single_infer(rgb_norm, mode=mode)             # clipped to [-1, 1]; v1 shifts / min-maxes in __call__
                        out[f"{tag}_{cname}_{mode}"] = pred.numpy()
                        ref = opipe.single_infer(vsd, vc, usd, uc, rgb_norm, ctx, mode)   # the oracle's [0, 1] map
                        np.testing.assert_allclose((pred.numpy() + 1.0) / 2.0, ref.numpy(), rtol=0, atol=1e-5,
                                                   err_msg=f"oracle single_infer != GenPercept_v1 single_infer ({tag}, {cname}, {mode})")
                # v1's latent (encode_rgb) is v2's
                p = v1.GenPerceptPipeline(unet=None, vae=VAE, customized_head=None, empty_text_embed=None)
                np.testing.assert_allclose(p.encode_rgb(rgb_norm).numpy(), osd.encode_rgb(vsd, vc, rgb_norm).numpy(), rtol=0, atol=1e-5)
    finally:
        sys.path[:] = path0
        _refexec_cleanup(saved)
    np.savez_compressed(os.path.join(HERE, "refexec_v1_tiny.npz"), **out)
    print("refexec_v1_tiny.npz", os.path.getsize(os.path.join(HERE, "refexec_v1_tiny.npz")) // 1024, "KiB;", len(out),
          "arrays; oracle == GenPercept_v1 single_infer executed, with the shipped empty_text_embed.npy (77 rows and rows [0:2])")


if __name__ == "__main__":
    which = sys.argv[1:] or ["dpt", "metrics", "e2e", "batchsize", "infer_eval", "image_util", "datasets", "scheduler", "ensemble", "multistep", "refexec"]
    if "scheduler" in which:
        make_scheduler_golden()
    if "ensemble" in which:
        make_ensemble_golden()
    if "multistep" in which:
        make_e2e_multistep()
    if "dpt" in which:
        make_dpt_golden()
    if "dpt" in which or "dpt_odd" in which:
        make_dpt_golden_odd()
    if "metrics" in which:
        make_metrics_golden()
    if "e2e" in which:
        make_e2e_tiny()
    if "batchsize" in which:
        make_batchsize_golden()
    if "infer_eval" in which:
        make_infer_eval_golden()
    if "image_util" in which:
        make_image_util_golden()
    if "datasets" in which:
        make_datasets_golden()
    if "refexec" in which:
        make_refexec_golden()
    if "refexec" in which or "refexec_v1" in which:
        make_refexec_v1_golden()
