#!/usr/bin/env python3
"""One-off validation tool: compare the fp32 oracle (oracle/) with the real `diffusers` modules.

The build container and the GPU boxes have neither `diffusers` nor SD2.1 weights, which is why the VAE / UNet restatement is
"parity unpinned" by reference fixtures.  On any machine that has `diffusers` (>= 0.25) and, optionally, a
local SD2.1 checkpoint, this script closes that gap:

    python tools/crosscheck_diffusers.py                       # random-initialised diffusers modules (architecture check)
    python tools/crosscheck_diffusers.py --sd21 /path/to/sd21  # the real stabilityai/stable-diffusion-2-1 weights

It instantiates UNet2DConditionModel / AutoencoderKL, copies their state dicts into the oracle (zero missing / unexpected keys is the
first check), runs both on the same seeded inputs on the CPU in fp32 and reports max / rms differences per stage.  It is a
validation tool, not a code path: nothing in genpercept_amd/, tests/ or bench.py imports it.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sd21", default=None, help="directory with unet/ and vae/ in diffusers layout")
    ap.add_argument("--size", type=int, default=128, help="test image edge (multiple of 8)")
    ap.add_argument("--tol", type=float, default=2e-4, help="max |oracle - diffusers| / max |diffusers| per stage")
    args = ap.parse_args()
    try:
        from diffusers import AutoencoderKL, UNet2DConditionModel
    except ImportError:
        print("diffusers is not installed here: nothing to cross-check (this is expected in the build container)")
        return 2
    from oracle import sd21 as osd

    torch.manual_seed(0)
    ucfg, vcfg = osd.UNetCfg(), osd.VAECfg()
    if args.sd21:
        unet = UNet2DConditionModel.from_pretrained(args.sd21, subfolder="unet").eval()
        vae = AutoencoderKL.from_pretrained(args.sd21, subfolder="vae").eval()
    else:  # architecture-only check with random weights of the SD2.1 configuration (Appendix A of SURVEY.md)
        unet = UNet2DConditionModel(sample_size=96, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                                    down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                                    up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, layers_per_block=2, cross_attention_dim=1024,
                                    attention_head_dim=(5, 10, 20, 20), use_linear_projection=True).eval()
        vae = AutoencoderKL(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                            down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4, norm_num_groups=32,
                            sample_size=768).eval()
    usd = {k: v.float() for k, v in unet.state_dict().items()}
    vsd = {k: v.float() for k, v in vae.state_dict().items()}
    ok = True
    for name, sd, man in (("unet", usd, osd.unet_manifest(ucfg)), ("vae", vsd, osd.vae_manifest(vcfg))):
        missing, extra = sorted(set(man) - set(sd)), sorted(set(sd) - set(man))
        bad_shape = [k for k in man if k in sd and tuple(sd[k].shape) != tuple(man[k])]
        print(f"{name}: {len(sd)} tensors; missing {len(missing)}, unexpected {len(extra)}, shape mismatches {len(bad_shape)}")
        ok &= not (missing or extra or bad_shape)
    g = torch.Generator().manual_seed(1)
    rgb = torch.rand(1, 3, args.size, args.size, generator=g) * 2 - 1
    ctx = torch.randn(1, 2, 1024, generator=g)

    def report(stage, a, b):
        nonlocal ok
        d = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        print(f"{stage:14s} rel max diff {d:.3e}")
        ok &= d <= args.tol

    with torch.no_grad():
        lat_ref = vae.quant_conv(vae.encoder(rgb))[:, :4] * 0.18215          # genpercept_pipeline.py:488-505 (mean of the posterior)
        report("vae.encode", osd.encode_rgb(vsd, vcfg, rgb), lat_ref)
        v_ref = unet(lat_ref, 1, encoder_hidden_states=ctx).sample             # genpercept_pipeline.py:455-457, t = 1
        report("unet(t=1)", osd.unet_forward(usd, ucfg, lat_ref, 1, ctx)[0], v_ref)
        dec_ref = vae.decoder(vae.post_quant_conv(-v_ref / 0.18215))           # scheduler beta = 1: pred_x0 = -v; :507-526
        report("vae.decode", osd.vae_decode(vsd, vcfg, -v_ref / 0.18215), dec_ref)
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
