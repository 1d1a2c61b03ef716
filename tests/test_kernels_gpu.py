"""Per-kernel parity tests (GPU): every HIP kernel, called through the C-ABI, against a plain PyTorch fp32 reference of
the same op evaluated on the same bf16-rounded operands.

Tolerances (written here, used below):
  * bf16 output of an fp32-accumulated kernel: one rounding to bf16 => relative error <= 2^-8 per element; we require
    max|err| <= 1.2e-2 * max|ref| and mean|err| <= 2e-3 * mean|ref|   (TOL_BF16); the fp16-element library (11 significant
    bits) is held to 1/8 of both
  * fp32 output: accumulation-order differences only => max|err| <= 2e-4 * max|ref|                      (TOL_F32)
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_BF16 = (1.2e-2, 2e-3)
TOL_F32 = 2e-4


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda", 0)


def _eng():
    from genpercept_amd import engine
    return engine


@pytest.fixture(autouse=True, params=["bf16", "fp16"])
def precision(request):
    """Every kernel test runs against both libraries: bf16 elements (libgenpercept_hip.so) and fp16 elements (libgenpercept_hip_f16.so)."""
    from genpercept_amd import engine
    engine.set_default_precision(request.param)
    yield request.param
    engine.set_default_precision("bf16")


def rbf(x):  # round to the library's 16-bit element and back: the kernels see exactly these values
    return x.to(_eng().act_dtype()).float()


def _tol16():
    """bf16: one rounding to 8 significant bits; fp16: 11 bits => 8x tighter."""
    return TOL_BF16 if _eng().act_dtype() == torch.bfloat16 else (TOL_BF16[0] / 8, TOL_BF16[1] / 8)


def check(name, out, ref, log, fp32=False, mean_factor=1.0):
    out = out.float().cpu()
    ref = ref.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    err = (out - ref).abs()
    mx, mean = err.max().item(), err.mean().item()
    rmx, rmean = ref.abs().max().item() + 1e-12, ref.abs().mean().item() + 1e-12
    log(name + ("" if _eng().act_dtype() == torch.bfloat16 else "[fp16]"), max_err=mx, mean_err=mean, ref_max=rmx, ref_mean=rmean, rel_max=mx / rmx, rel_mean=mean / rmean)
    if fp32:
        assert mx <= TOL_F32 * rmx, f"{name}: max err {mx:.3e} vs ref max {rmx:.3e}"
    else:
        t = _tol16()
        assert mx <= t[0] * rmx, f"{name}: max err {mx:.3e} vs ref max {rmx:.3e}"
        assert mean <= mean_factor * t[1] * rmean + 1e-6, f"{name}: mean err {mean:.3e} vs ref mean {rmean:.3e}"


def nhwc_to_nchw(y, c=None):
    y = y.float().permute(0, 3, 1, 2)
    return y if c is None else y[:, :c]


CONV_CASES = [
    # B, H, W, Cin, Cout, tile
    (2, 16, 16, 64, 64, 1), (2, 16, 16, 64, 64, 2), (2, 16, 16, 64, 64, 3),
    (1, 24, 20, 128, 320, 0), (1, 13, 15, 192, 100, 1), (3, 9, 7, 64, 3, 0), (1, 12, 12, 1280, 1280, 0), (1, 40, 36, 256, 128, 1),
    (2, 16, 16, 64, 64, 4), (1, 40, 36, 256, 128, 4), (2, 33, 31, 128, 320, 4), (1, 96, 96, 128, 128, 0), (1, 17, 19, 64, 200, 2),
    # split-K path (tiny maps, long K): 4 / 8 / 2 slices, ragged rows and columns
    (4, 12, 12, 1280, 1280, 0), (1, 12, 12, 2560, 1280, 0), (2, 12, 12, 640, 320, 0), (1, 9, 11, 1280, 200, 2),
    # whole-image tiles (conv_img.hip): 24x24 maps (one image per 576-pixel unit) and 12x12 maps (four per unit), K slices of uneven length,
    # one and several units, odd chunk counts; (4, 12, 12, 1280, 1280, 0) above takes this path too
    (4, 24, 24, 128, 128, 0), (1, 24, 24, 640, 64, 0), (2, 24, 24, 1280, 320, 0), (8, 12, 12, 192, 192, 0),
    # ((4, 12, 12, 64, 64) and (4, 6, 24, 320, 128) did not qualify -- one chunk per slice forces S = 1, 4 x 8 x 26 halo rows exceed
    #  CI_HROWS_MAX -- and fell back to the generic path; kept as fall-back cases, with two shapes that DO reach conv_img: the minimum of two
    #  chunks per K slice, and a non-square map with two images per unit)
    (4, 12, 12, 64, 64, 0), (4, 6, 24, 320, 128, 0), (4, 12, 12, 128, 64, 0), (2, 12, 24, 320, 128, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_s1(case, metric_log):
    e = _eng()
    b, h, w, cin, cout, tile = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    res = rbf(torch.randn(b, cout, h, w, generator=g))
    ref = F.conv2d(x, wt, bias, padding=1) + res
    d = _dev()
    wp = e.pack_weight(wt, device=d)
    y = e.conv2d(e.to_nhwc_bf16(x.to(d)), wp, bias.to(d), cout, 3, residual=e.to_nhwc_bf16(res.to(d)), tile=tile)
    check(f"conv3x3_s1{case}", nhwc_to_nchw(y), ref, metric_log)


@pytest.mark.parametrize("case", [(2, 64, 64, 128, True), (1, 37, 50, 64, True), (1, 16, 16, 160, False), (3, 33, 17, 128, False)])
def test_rgb_conv_in_fused_prologue(case, metric_log):
    """RGB prologue (uint8 -> x/255*2-1, genpercept_pipeline.py:245) fused with the VAE encoder's conv_in (K = 27 as one MFMA k-step)."""
    e = _eng()
    b, h, w, cout, u8 = case
    g = torch.Generator().manual_seed(h * w + cout)
    if u8:
        rgb = torch.randint(0, 256, (b, 3, h, w), generator=g, dtype=torch.uint8)
        x = rbf(rgb.float() / 255.0 * 2.0 - 1.0)
    else:
        rgb = torch.rand(b, 3, h, w, generator=g) * 2 - 1
        x = rbf(rgb)
    wt = rbf(torch.randn(cout, 3, 3, 3, generator=g) / math.sqrt(27))
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    d = _dev()
    y = e.rgb_conv_in(rgb.to(d), e.pack_weight(wt, device=d), bias.to(d), cout)
    check(f"rgb_conv_in{case}", nhwc_to_nchw(y), ref, metric_log)


HALO_CASES = [
    # B, H, W, Cin, Cout, ups, act, residual
    (1, 16, 16, 64, 128, False, "none", False), (2, 32, 48, 128, 128, False, "none", True), (1, 40, 36, 256, 320, False, "none", True),
    (1, 17, 23, 64, 64, False, "relu", False), (3, 16, 20, 192, 200, False, "silu", True), (1, 96, 96, 128, 128, False, "none", False),
    (1, 8, 8, 128, 128, True, "none", False), (2, 12, 10, 64, 256, True, "none", False), (1, 24, 24, 320, 320, True, "relu", False),
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_kernel(case, metric_log):
    """conv_halo.hip (tile hint 5): 16x16-pixel tiles with the input halo staged once per channel chunk; also x2-upsample form."""
    e = _eng()
    b, h, w, cin, cout, ups, act, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]))
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, wt, bias, padding=1)
    res = rbf(torch.randn(ref.shape, generator=g)) if with_res else None
    if with_res:
        ref = ref + res
    ref = {"none": lambda t: t, "relu": F.relu, "silu": F.silu}[act](ref)
    d = _dev()
    y = e.conv2d(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, ups_hw=(2 * h, 2 * w) if ups else None,
                 residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, act=act, tile=5)
    check(f"conv_halo{case}", nhwc_to_nchw(y), ref, metric_log)


@pytest.mark.parametrize("case", [
    # (B, H, W, Cin, Cout, ks, ups, residual, tile hint): halo kernels (5: persistent, several tiles per workgroup when B*tiles > #CU) and
    # the generic implicit GEMM (0 -> heuristic tile, 1 = 128x128, 4 = 256x128)
    (2, 32, 32, 128, 128, 3, False, True, 5), (1, 40, 24, 64, 192, 3, False, False, 5), (4, 144, 160, 64, 128, 3, False, True, 5),
    (1, 24, 24, 128, 128, 3, True, True, 5), (3, 112, 112, 64, 256, 3, False, False, 5),
    (2, 32, 32, 128, 128, 3, False, True, 4), (2, 32, 32, 128, 256, 1, False, True, 1), (1, 16, 16, 320, 320, 3, False, False, 1),
    # persistent GEMM (hint 7): 128-row and 256-row tiles
    (4, 48, 48, 320, 640, 1, False, True, 7), (2, 96, 96, 320, 320, 1, False, False, 7), (4, 224, 224, 64, 128, 1, False, True, 7),
])
def test_conv_epilogue_groupnorm_statistics(case, metric_log):
    """GroupNorm statistics accumulated in the conv epilogue (per-tile channel sums of the bf16 values as stored) and finalised to
    scale/shift: must equal the statistics of the tensor the conv wrote (diffusers resnet.py: norm2(conv1(...)) etc.)."""
    e = _eng()
    b, h, w, cin, cout, ks, ups, with_res, tile = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(cin + cout + h)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks))
    bias = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, wt, bias, padding=ks // 2)
    res = rbf(torch.randn(ref.shape, generator=g)) if with_res else None
    if with_res:
        ref = ref + res
    d = _dev()
    y, scale, shift = e.conv2d_stats(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, ks, gamma.to(d), beta.to(d), groups, eps,
                                     ups=ups, residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, tile=tile)
    check(f"conv_stats_out{case}", nhwc_to_nchw(y), ref, metric_log)
    # statistics of exactly what was stored
    ys = nhwc_to_nchw(y).float().cpu()
    yg = ys.reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    rstd = (var + eps).rsqrt()
    cpg = cout // groups
    sc_ref = gamma[None, :] * rstd.repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


# conv3x3_halo3_kernel<false, 0, 0, 4, PH = true> (r5): the x2-nearest-upsample 3x3 conv (diffusers Upsample2D) as four 2 x 2-tap phase convolutions on the
# source map -- 4/9 of the MFMA work.  The reference is built from the SAME operands the kernel sees (source pixels and the phase-summed weights, each
# rounded once to the element type), so the per-kernel tolerance applies; the un-decomposed conv of the upsampled map (fp32 weights) is checked beside it.
UP2_CASES = [
    # B, Hi, Wi, Cin, Cout, residual
    (1, 16, 16, 64, 128, False), (2, 24, 40, 128, 128, True), (1, 17, 33, 64, 64, False), (2, 31, 47, 192, 320, True), (1, 48, 48, 512, 512, False),
    (4, 96, 96, 128, 256, False), (1, 20, 16, 1280, 200, False),
]


def _phase_weights(wt):
    """[O][C][3][3] -> {(a, b): [O][C][2][2]}: kernel rows / columns that fall onto the same source pixel summed (engine.hip: pack_phase_rows)"""
    rng = {0: [(0, 0), (1, 2)], 1: [(0, 1), (2, 2)]}
    out = {}
    for a in (0, 1):
        for b in (0, 1):
            w = torch.zeros(wt.shape[0], wt.shape[1], 2, 2)
            for ty, (y0, y1) in enumerate(rng[a]):
                for tx, (x0, x1) in enumerate(rng[b]):
                    w[:, :, ty, tx] = wt[:, :, y0:y1 + 1, x0:x1 + 1].sum(dim=(2, 3))
            out[(a, b)] = w
    return out


@pytest.mark.parametrize("case", UP2_CASES)
def test_conv_upsample_x2_phase_kernel(case, metric_log):
    e = _eng()
    b, h, w, cin, cout, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 22)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g)
    exact = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)   # the op itself, fp32 weights
    ref = torch.empty_like(exact)                                                                # the same from the kernel's operands
    for (a, bb), wp in _phase_weights(wt).items():
        xp = F.pad(x, (1 - bb, bb, 1 - a, a))   # phase (a, b) reads source rows y - 1 + a .. y + a, columns x - 1 + b .. x + b
        ref[:, :, a::2, bb::2] = F.conv2d(xp, rbf(wp), bias)
    res = rbf(torch.randn(exact.shape, generator=g)) if with_res else None
    if with_res:
        ref, exact = ref + res, exact + res
    d = _dev()
    y = e.conv2d_up2(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), e.pack_weight_phases(wt, device=d), bias.to(d), cout,
                     residual=e.to_nhwc_bf16(res.to(d)) if with_res else None)
    check(f"conv_up2_phases{case}", nhwc_to_nchw(y), ref, metric_log)
    # against the un-decomposed conv: the only extra difference is where the weights are rounded (after the sum instead of before)
    rel = ((nhwc_to_nchw(y).float().cpu() - exact).abs().mean() / exact.abs().mean()).item()
    metric_log(f"conv_up2_vs_exact{case}", rel_mean=rel)
    assert rel <= 3 * _tol16()[1], rel
    # and against the nine-tap upsample kernel on the same input
    y9 = e.conv2d(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, ups_hw=(2 * h, 2 * w),
                  residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, tile=5)
    rel9 = ((nhwc_to_nchw(y).float() - nhwc_to_nchw(y9).float()).abs().mean() / exact.abs().mean()).item()
    metric_log(f"conv_up2_vs_ninetap{case}", rel_mean=rel9)
    assert rel9 <= 4 * _tol16()[1], rel9
    assert torch.equal(e.conv2d_up2(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), e.pack_weight_phases(wt, device=d), bias.to(d), cout,
                                    residual=e.to_nhwc_bf16(res.to(d)) if with_res else None), y)


# conv3x3_halo3_kernel<false, 0, 0, 3>: 12-row x 16-column tiles (r5; chosen by halo_plan where 16 x 16 tiles quantise badly over the persistent grid,
# e.g. 4 x 96 x 96 x 512 -> 512).  Forced through IGemmParams::dbg bits 20-21 (GENPERCEPT_IGEMM_DBG = 1 << 20; 2 << 20 forbids it): heights that
# are / are not multiples of 12 and 16, ragged right edges, one and several tiles per workgroup, several channel slices incl. a ragged one,
# chunk counts 1 .. 8, with and without residual.  Every output pixel is accumulated in the same (chunk, tap, k) order as in the 16-row kernel, so
# the two must agree BIT FOR BIT.
TR3_CASES = [
    # B, H, W, Cin, Cout, residual
    (1, 16, 16, 64, 128, False), (2, 24, 48, 64, 128, True), (1, 48, 96, 128, 256, True), (1, 17, 33, 64, 64, True), (3, 40, 72, 192, 320, False),
    (1, 19, 40, 320, 200, True), (4, 96, 96, 512, 512, True), (2, 31, 95, 256, 128, True), (4, 144, 160, 64, 128, False),
]


@pytest.mark.parametrize("case", TR3_CASES)
def test_conv3x3_halo3_12row_tiles(case, metric_log, monkeypatch):
    e = _eng()
    b, h, w, cin, cout, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 12)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    res = rbf(torch.randn(ref.shape, generator=g)) if with_res else None
    if with_res:
        ref = ref + res
    d = _dev()
    args = (e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3)
    kw = dict(residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, tile=5)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    y12 = e.conv2d(*args, **kw)
    check(f"conv_halo3_tr3{case}", nhwc_to_nchw(y12), ref, metric_log)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(2 << 20))
    y16 = e.conv2d(*args, **kw)
    assert torch.equal(y12, y16), "12-row and 16-row tiles must give identical outputs (same per-pixel summation order)"
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    assert torch.equal(e.conv2d(*args, **kw), y12)


@pytest.mark.parametrize("case", [(2, 24, 64, 128, 128, True), (1, 40, 72, 64, 192, False), (4, 96, 96, 64, 256, True), (1, 17, 33, 64, 320, True)])
def test_conv_halo3_12row_tiles_groupnorm_statistics(case, metric_log, monkeypatch):
    """the statistics the 12-row-tile kernel leaves for the next GroupNorm (per-workgroup partial rows + pixel counts) are those of the tensor it stored"""
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    e = _eng()
    b, h, w, cin, cout, with_res = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(cin + cout + h + 12)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    res = rbf(torch.randn(ref.shape, generator=g)) if with_res else None
    if with_res:
        ref = ref + res
    d = _dev()
    y, scale, shift = e.conv2d_stats(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, gamma.to(d), beta.to(d), groups, eps,
                                     ups=False, residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, tile=5)
    check(f"conv_halo3_tr3_stats_out{case}", nhwc_to_nchw(y), ref, metric_log)
    yg = nhwc_to_nchw(y).float().cpu().reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    cpg = cout // groups
    sc_ref = gamma[None, :] * (var + eps).rsqrt().repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_halo3_tr3_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


@pytest.mark.parametrize("case", [(1, 16, 16, 64, 128, False), (2, 24, 40, 128, 128, True), (2, 31, 47, 192, 320, False), (4, 96, 96, 128, 256, False)])
def test_conv_upsample_x2_phase_kernel_groupnorm_statistics(case, metric_log):
    """the phase kernel's statistics path is live in the product (the VAE decoder's upsampler convs feed the next resnet's norm1): per (tile,
    phase) unit sums and pixel counts, partial tiles at the edges -- the scale / shift finalised from them are those of the tensor it stored."""
    e = _eng()
    b, h, w, cin, cout, with_res = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 77)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    res = rbf(torch.randn(b, cout, 2 * h, 2 * w, generator=g)) if with_res else None
    d = _dev()
    y, scale, shift = e.conv2d_up2_stats(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), e.pack_weight_phases(wt, device=d), bias.to(d), cout,
                                         gamma.to(d), beta.to(d), groups, eps, residual=e.to_nhwc_bf16(res.to(d)) if with_res else None)
    y0 = e.conv2d_up2(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), e.pack_weight_phases(wt, device=d), bias.to(d), cout,
                      residual=e.to_nhwc_bf16(res.to(d)) if with_res else None)
    assert torch.equal(y, y0), "the statistics epilogue changed the stored tensor"
    yg = nhwc_to_nchw(y).float().cpu().reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    cpg = cout // groups
    sc_ref = gamma[None, :] * (var + eps).rsqrt().repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_up2_phase_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


@pytest.mark.parametrize("case", [(2, 256, 5), (1, 1200, 10), (3, 100, 2), (1, 4800, 5), (2, 64, 20), (1, 37, 1)])
def test_flash_attention_split_operands(case, metric_log):
    """flash_attn64_split_kernel (+ c_qkv_planes): softmax(q k^T / 8) v per head over hi / lo bf16 pieces of fp32 q, k, v (three MFMAs per product, fp32 softmax, P split
    in registers), against fp32 torch attention on the SAME fp32 inputs; token counts that are / are not multiples of the 64-key tile and of the 128-query block."""
    from genpercept_amd import engine as e
    if e.act_dtype() != torch.bfloat16:
        pytest.skip("the contract precision lives in the bf16 library")
    b, t, heads = case
    c = heads * 64
    g = torch.Generator().manual_seed(t + heads)
    qkv = torch.randn(b * t, 3 * c, generator=g) * torch.tensor([1.5] * c + [1.5] * c + [1.0] * c)  # logits of a few units: a softmax with structure
    q, k, v = (qkv[:, i * c:(i + 1) * c].reshape(b, t, heads, 64).transpose(1, 2).double() for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(b * t, c).float()
    out = e.flash_attention_split(qkv.to(_dev()), b, t, heads).cpu()
    err = (out - ref).abs()
    rel = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    metric_log(f"flash_attn64_split{case}", rel_rms=rel, max_abs=err.max().item())
    assert torch.isfinite(out).all() and rel <= 3e-5, rel


@pytest.mark.parametrize("case", [(2, 32, 32, 128, 128, 0), (1, 40, 24, 192, 320, 0), (4, 96, 96, 512, 512, 1), (1, 50, 70, 64, 200, 0), (2, 48, 48, 640, 640, 1)])
def test_conv_halo3_fp32_rows_epilogue(case, metric_log, monkeypatch):
    """conv3x3_halo3_kernel<..., HALO_F32O> (r6, the contract precision's conv epilogue: fp32 rows out instead of 16-bit): same operands as the 16-bit kernel, so the
    result is the fp32 accumulator itself -- against fp32 torch on identical 16-bit operands it must agree to fp32 summation-order noise, for 16-row and (case[5] = 1:
    forced) 12-row tiles, ragged tiles and channel counts that are not multiples of 128."""
    e = _eng()
    b, h, w, cin, cout, tr3 = case
    if tr3:
        monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    g = torch.Generator().manual_seed(cin + cout + h + 5)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1).float()
    d = _dev()
    y = e.conv2d(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, out_fp32=True, tile=5)
    assert y.dtype == torch.float32
    out = y.permute(0, 3, 1, 2).cpu()
    err = (out - ref).abs()
    rel = (err.max() / ref.abs().max()).item()
    metric_log(f"conv_halo3_f32o{case}", rel_max=rel, mean_err=err.mean().item())
    assert torch.isfinite(out).all() and rel <= 2e-6 * math.sqrt(cin * 9 / 64), rel  # fp32 accumulation over K = 9 Cin
    # the 16-bit kernel's output is this tensor rounded once
    y16 = e.conv2d(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, tile=5)
    assert torch.equal(y16.float().cpu(), y.cpu().to(e.act_dtype()).float()), "16-bit and fp32-row epilogues disagree beyond the final rounding"


@pytest.mark.parametrize("case", [(2, 32, 32, 128, 128, False, True), (1, 40, 24, 320, 192, False, True), (2, 16, 16, 64, 128, False, False),
                                  (1, 12, 16, 256, 128, True, True), (1, 17, 21, 960, 64, False, True), (1, 24, 24, 2560, 128, False, True),
                                  (4, 48, 48, 1920, 128, False, True)])
def test_conv3x3_fused_groupnorm_input(case, metric_log):
    """GroupNorm apply (+SiLU) fused into the halo conv's input staging; zero padding applies to the NORMALISED tensor."""
    e = _eng()
    b, h, w, cin, cout, ups, silu = case
    g = torch.Generator().manual_seed(cin + h)
    x = rbf(torch.randn(b, cin, h, w, generator=g) * 1.5 + 0.5)
    gamma, beta = 1 + 0.2 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    n = F.group_norm(x, 32, gamma, beta, 1e-6)
    if silu:
        n = F.silu(n)
    n = rbf(n)  # the kernel rounds the normalised value to bf16 before the MFMA, like the unfused path stores it
    if ups:
        n = F.interpolate(n, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(n, wt, bias, padding=1)
    d = _dev()
    y = e.conv2d_gn(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, gamma.to(d), beta.to(d), 32, 1e-6, silu, ups=ups)
    check(f"conv_gn_fused{case}", nhwc_to_nchw(y), ref, metric_log)


def test_conv_small_cin_padded(metric_log):
    """conv_in-style layers: 3 (or 4) real input channels zero-padded to 64; small Cout zero-padded on store."""
    e = _eng()
    g = torch.Generator().manual_seed(5)
    x = rbf(torch.randn(2, 3, 24, 24, generator=g))
    wt = rbf(torch.randn(128, 3, 3, 3, generator=g) / math.sqrt(27))
    bias = torch.randn(128, generator=g)
    d = _dev()
    y = e.conv2d(e.to_nhwc_bf16(x.to(d), 64), e.pack_weight(wt, 64, device=d), bias.to(d), 128, 3)
    check("conv_in_pad64", nhwc_to_nchw(y), F.conv2d(x, wt, bias, padding=1), metric_log)
    # Cout = 4 stored into 64 zero-padded channels (latent layout)
    x2 = rbf(torch.randn(1, 128, 10, 12, generator=g))
    w2 = rbf(torch.randn(4, 128, 3, 3, generator=g) / math.sqrt(128 * 9))
    b2 = torch.randn(4, generator=g)
    y2 = e.conv2d(e.to_nhwc_bf16(x2.to(d)), e.pack_weight(w2, device=d), b2.to(d), 4, 3, n_store=64)
    assert y2.shape[-1] == 64 and float(y2[..., 4:].float().abs().max()) == 0.0
    check("conv_cout4_store64", nhwc_to_nchw(y2, 4), F.conv2d(x2, w2, b2, padding=1), metric_log)


@pytest.mark.parametrize("hw", [(16, 16), (15, 13), (30, 40)])
def test_conv_stride2_both_paddings(hw, metric_log):
    e = _eng()
    h, w = hw
    g = torch.Generator().manual_seed(h * 100 + w)
    x = rbf(torch.randn(2, 64, h, w, generator=g))
    wt = rbf(torch.randn(128, 64, 3, 3, generator=g) / math.sqrt(64 * 9))
    bias = torch.randn(128, generator=g)
    d = _dev()
    xd, wp = e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d)
    # UNet Downsample2D: symmetric pad 1
    ref = F.conv2d(x, wt, bias, stride=2, padding=1)
    y = e.conv2d(xd, wp, bias.to(d), 128, 3, stride=2, pad_t=1, pad_l=1, out_hw=ref.shape[2:])
    check(f"conv_s2_sym{hw}", nhwc_to_nchw(y), ref, metric_log)
    # VAE encoder Downsample2D: pad right/bottom only, then stride 2 without padding (Appendix B.6)
    ref2 = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, bias, stride=2, padding=0)
    y2 = e.conv2d(xd, wp, bias.to(d), 128, 3, stride=2, pad_t=0, pad_l=0, out_hw=ref2.shape[2:])
    check(f"conv_s2_asym{hw}", nhwc_to_nchw(y2), ref2, metric_log)


@pytest.mark.parametrize("sizes", [((8, 8), (16, 16)), ((7, 10), (15, 20)), ((4, 5), (8, 10)), ((8, 10), (15, 20))])
def test_conv_fused_nearest_upsample(sizes, metric_log):
    """Upsample2D: nearest x2 or nearest-to-size (custom_unet.py:377-378) fused into the conv's gather."""
    e = _eng()
    (h, w), (uh, uw) = sizes
    g = torch.Generator().manual_seed(h * 31 + uw)
    x = rbf(torch.randn(2, 128, h, w, generator=g))
    wt = rbf(torch.randn(64, 128, 3, 3, generator=g) / math.sqrt(128 * 9))
    bias = torch.randn(64, generator=g)
    ref = F.conv2d(F.interpolate(x, size=(uh, uw), mode="nearest"), wt, bias, padding=1)
    d = _dev()
    y = e.conv2d(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), 64, 3, ups_hw=(uh, uw))
    check(f"conv_ups{sizes}", nhwc_to_nchw(y), ref, metric_log)


GEMM_CASES = [(64, 64, 64, 0), (576, 1280, 320, 0), (100, 72, 128, 2), (1000, 320, 1280, 1), (36, 2560, 320, 0), (300, 24, 192, 3), (4800, 320, 320, 1),
              (4800, 320, 320, 4), (1000, 130, 64, 4), (36864, 320, 320, 0), (70, 64, 2304, 4),
              # tile hint 7: persistent GEMM (pgemm.hip); 128-row tiles, several tiles per workgroup, ragged last tile, K up to 40 steps,
              # and (last case) the 256-row form with a ragged tail
              (36864, 320, 320, 7), (4801, 320, 320, 7), (2304, 1280, 1280, 7), (9216, 640, 2560, 7), (300, 64, 64, 7), (200003, 128, 128, 7)]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_bias_residual(case, metric_log):
    e = _eng()
    m, n, k, tile = case
    g = torch.Generator().manual_seed(m + n + k)
    a = rbf(torch.randn(m, k, generator=g))
    bt = rbf(torch.randn(n, k, generator=g) / math.sqrt(k))
    bias = torch.randn(n, generator=g)
    res = rbf(torch.randn(m, n, generator=g))
    d = _dev()
    ad, bd = a.to(d).to(_eng().act_dtype()), bt.to(d).to(_eng().act_dtype())
    nst = (n + 3) // 4 * 4
    resd = torch.zeros(m, nst, dtype=_eng().act_dtype(), device=d)
    resd[:, :n] = res.to(d).to(_eng().act_dtype())
    y = e.gemm(ad, bd, bias=bias.to(d), residual=resd, n_store=nst, tile=tile)
    check(f"gemm{case}", y[:, :n], a @ bt.t() + bias + res, metric_log)
    y32 = e.gemm(ad, bd, out_fp32=True, n_store=nst, tile=tile)
    check(f"gemm_f32{case}", y32[:, :n], a @ bt.t(), metric_log, fp32=True)
    if tile != 7:
        y16 = e.gemm(ad, bd, out_fp32=2, n_store=nst, tile=tile)  # fp16 output (attention logits)
        assert y16.dtype == torch.float16
        check(f"gemm_f16{case}", y16[:, :n], a @ bt.t(), metric_log)


def test_gemm_row_bias_batched_zero_fill(metric_log):
    """The V^T projection form: out[b][c][t] = sum_k W[c][k] x[b][t][k] + bias[c], zero-filled to Tpad columns."""
    e = _eng()
    g = torch.Generator().manual_seed(3)
    bsz, t, c = 3, 100, 128
    tpad = 128
    wv = rbf(torch.randn(c, c, generator=g) / math.sqrt(c))
    x = rbf(torch.randn(bsz, t, c, generator=g))
    bias = torch.randn(c, generator=g)
    d = _dev()
    lib = e.load_library()
    out = torch.full((bsz, c, tpad), 7.0, dtype=_eng().act_dtype(), device=d)
    wd, xd, bd = wv.to(d).to(_eng().act_dtype()), x.to(d).to(_eng().act_dtype()), bias.to(d)
    st = lib.gp_gemm(wd.data_ptr(), c, xd.data_ptr(), c, bd.data_ptr(), 2, None, 0, out.data_ptr(), tpad, c, t, c, t, tpad, 0, 0, bsz, 0, t * c, c * tpad, 0,
                     torch.cuda.current_stream().cuda_stream)
    assert st == 0
    ref = torch.einsum("ck,btk->bct", wv, x) + bias[None, :, None]
    check("gemm_vt_rowbias", out[:, :, :t], ref, metric_log)
    assert float(out[:, :, t:].float().abs().max()) == 0.0, "padding columns must be written as zeros"


@pytest.mark.parametrize("case", [(1, 16, 16), (2, 40, 56), (1, 33, 17), (3, 64, 64)])
@pytest.mark.parametrize("mean3", [True, False])
def test_decoder_tail_fused(case, mean3, metric_log):
    """gp_decoder_tail: GroupNorm(32, eps 1e-6) + SiLU + conv3x3(128 -> 3) + [channel mean] + clip / shift in one kernel, fp32 NCHW out --
    against torch on the same 16-bit-rounded input and weights (the normalised activations are rounded to the element type before the conv,
    like every other conv input of the engine); ragged tiles, several tiles / images per persistent workgroup."""
    e = _eng()
    b, h, w = case
    g = torch.Generator().manual_seed(h * 100 + w + b)
    x = rbf(torch.randn(b, 128, h, w, generator=g) * 1.5 + 0.3)
    wt = rbf(torch.randn(3, 128, 3, 3, generator=g) / math.sqrt(128 * 9) * 3.0)
    bias = torch.randn(3, generator=g) * 0.2
    gamma, beta = torch.randn(128, generator=g) * 0.3 + 1.0, torch.randn(128, generator=g) * 0.2
    d = _dev()
    wp = e.pack_weight(wt, device=d)
    xd = e.to_nhwc_h16(x.to(d))
    out = e.decoder_tail(xd, wp, bias.to(d), gamma.to(d), beta.to(d), 32, 1e-6, mean3)
    n = rbf(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-6)))
    y = F.conv2d(n, wt, bias, padding=1)
    if mean3:
        y = y.mean(dim=1, keepdim=True)
    ref = (y.clamp(-1, 1) + 1) / 2
    err = (out.cpu() - ref).abs()
    tol = 3e-3 if e.act_dtype() == torch.bfloat16 else 5e-4   # the only 16-bit rounding on the path is the normalised conv input
    metric_log(f"decoder_tail{case}[mean3={mean3}]" + ("" if e.act_dtype() == torch.bfloat16 else "[fp16]"), max_err=err.max().item(), mean_err=err.mean().item())
    assert err.max().item() <= 4 * tol and err.mean().item() <= tol, (err.max().item(), err.mean().item())
    raw = e.decoder_tail(xd, wp, bias.to(d), gamma.to(d), beta.to(d), 32, 1e-6, mean3, raw=True)
    assert (raw.cpu() - y).abs().mean().item() <= 2 * tol


@pytest.mark.parametrize("case", [(4, 576, 320), (2, 2304, 640), (4, 144, 1280), (1, 400, 64), (3, 272, 128)])
def test_gemm_qkv_fused_projection(case, metric_log):
    """gp_gemm_qkv: attn1.to_q | to_k | to_v of a BasicTransformerBlock as one GEMM -- q | k row-major, V written transposed [B][C][Tpad] with
    zero padding beyond T (the operand layouts of gp_flash_attention); 128- and 256-row tiles, T with and without padding columns, a ragged
    last V slice (3C not a multiple of 128)."""
    e = _eng()
    b, t, c = case
    g = torch.Generator().manual_seed(b * 1000 + t + c)
    x = rbf(torch.randn(b * t, c, generator=g))
    w = rbf(torch.randn(3 * c, c, generator=g) / math.sqrt(c))
    d = _dev()
    wp = e.pack_weight(w, device=d)
    qk, vt = e.gemm_qkv(x.to(d).to(e.act_dtype()), wp.reshape(wp.shape[0], -1), b, t, c)
    ref = x @ w.t()
    check(f"gemm_qkv_qk{case}", qk, ref[:, :2 * c], metric_log)
    vref = ref[:, 2 * c:].reshape(b, t, c).permute(0, 2, 1)
    check(f"gemm_qkv_vt{case}", vt[:, :, :t], vref, metric_log)
    if vt.shape[2] > t:
        assert float(vt[:, :, t:].float().abs().max()) == 0.0, "keys beyond T must read as zero"
    # the two-launch form (q | k GEMM + row-bias-free transposed V GEMM) gives the same numbers: same MFMA order per output
    qk2 = e.gemm(x.to(d).to(e.act_dtype()), wp.reshape(wp.shape[0], -1)[:2 * c])
    assert torch.equal(qk2, qk)


@pytest.mark.parametrize("mc", [(200, 128), (4801, 128), (36864, 320), (2304, 64)])
def test_gemm_geglu(mc, metric_log):
    """GEGLU feed-forward GEMM: M = 200 stays on igemm_kernel's direct path, the larger ones run the persistent GEMM's GEGLU epilogue."""
    e = _eng()
    g = torch.Generator().manual_seed(4)
    m, c = mc
    a = rbf(torch.randn(m, c, generator=g))
    w = rbf(torch.randn(8 * c, c, generator=g) / math.sqrt(c))
    bias = torch.randn(8 * c, generator=g)
    proj = a @ w.t() + bias
    hidden, gate = proj.chunk(2, dim=-1)
    ref = hidden * F.gelu(gate)
    d = _dev()
    wp = e.pack_weight(w, geglu=True, device=d)
    # GEGLU-permuted bias (same row permutation as gp_pack_weight(geglu=1)): per 32-row block, value j at 8*(j%16//4) + j%4, gate +4
    half = 4 * c
    idx = torch.arange(8 * c)
    r = torch.where(idx >= half, idx - half, idx)
    dst = (r // 16) * 32 + ((r % 16) // 4) * 8 + (idx >= half).long() * 4 + (r % 4)
    pb = torch.empty_like(bias)
    pb[dst] = bias
    y = e.conv2d(a.to(d).to(_eng().act_dtype()).reshape(1, 1, m, c), wp, pb.to(d), 8 * c, 1, act="geglu")
    check(f"gemm_geglu{mc}", y.reshape(m, 4 * c), ref, metric_log)


@pytest.mark.parametrize("case", [(2, 64, 64, True), (1, 128, 4096, True), (2, 320, 300, False), (1, 960, 144, True), (3, 2560, 36, True),
                                  (1, 512, 1, True), (1, 128, 70000, True),
                                  # the one-launch small-map kernel (a group's HW x C/32 block per workgroup, (C / 32) % 8 == 0)
                                  (4, 1280, 144, True), (4, 2560, 144, True), (2, 1280, 576, False), (4, 256, 37, True)])
def test_groupnorm(case, metric_log):
    e = _eng()
    b, c, hw, silu = case
    g = torch.Generator().manual_seed(c + hw)
    x = rbf(torch.randn(b, c, hw, 1, generator=g) * 2.0 + 0.7)
    gamma = 1 + 0.2 * torch.randn(c, generator=g)
    beta = 0.3 * torch.randn(c, generator=g)
    for eps in (1e-5, 1e-6):
        ref = F.group_norm(x, 32, gamma, beta, eps)
        if silu:
            ref = F.silu(ref)
        d = _dev()
        y = e.groupnorm(e.to_nhwc_bf16(x.to(d)), gamma.to(d), beta.to(d), 32, eps, silu)
        check(f"groupnorm{case}eps{eps}", nhwc_to_nchw(y), ref, metric_log)


@pytest.mark.parametrize("case", [(4, 1280, 144, True), (4, 2560, 144, True), (2, 1280, 576, False), (4, 256, 37, True), (1, 1280, 600, True)])
def test_groupnorm_small_register_kernel_equals_three_pass_kernel(case, metric_log, monkeypatch):
    """gn_small_reg_kernel (r5: the workgroup's slice stays in registers between the mean, variance and apply passes) keeps gn_small_kernel's
    summation order: bit-identical outputs (GENPERCEPT_GN_SMALL_OLD=1 selects the three-pass kernel)."""
    e = _eng()
    b, c, hw, silu = case
    g = torch.Generator().manual_seed(c + hw + 5)
    x = e.to_nhwc_bf16((rbf(torch.randn(b, c, hw, 1, generator=g) * 2.0 + 0.7)).to(_dev()))
    gamma, beta = (1 + 0.2 * torch.randn(c, generator=g)).to(_dev()), (0.3 * torch.randn(c, generator=g)).to(_dev())
    y_new = e.groupnorm(x, gamma, beta, 32, 1e-5, silu)
    monkeypatch.setenv("GENPERCEPT_GN_SMALL_OLD", "1")
    y_old = e.groupnorm(x, gamma, beta, 32, 1e-5, silu)
    assert torch.equal(y_new, y_old)


@pytest.mark.parametrize("case", [(100, 64), (577, 320), (64, 640), (1000, 1280), (3, 2560)])
def test_layernorm(case, metric_log):
    e = _eng()
    rows, c = case
    g = torch.Generator().manual_seed(rows + c)
    x = rbf(torch.randn(rows, c, generator=g) * 3 - 1)
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.3 * torch.randn(c, generator=g)
    d = _dev()
    y = e.layernorm(x.to(d).to(_eng().act_dtype()), gamma.to(d), beta.to(d))
    check(f"layernorm{case}", y, F.layer_norm(x, (c,), gamma, beta, 1e-5), metric_log)


def _attn_ref(q, k, v, heads):
    b, t, c = q.shape
    hd = c // heads
    qh, kh, vh = (z.view(b, -1, heads, hd).transpose(1, 2) for z in (q, k, v))
    w = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, dim=-1)
    return (w @ vh).transpose(1, 2).reshape(b, t, c)


@pytest.mark.parametrize("case", [(1, 1, 1), (2, 16, 2), (1, 64, 1), (2, 100, 5), (1, 144, 20), (1, 576, 4), (2, 1200, 2), (1, 2304, 10)])
def test_flash_attention_hd64(case, metric_log):
    e = _eng()
    b, t, heads = case
    c = heads * 64
    g = torch.Generator().manual_seed(t * 7 + heads)
    qk = rbf(torch.randn(b, t, 2 * c, generator=g) * 1.5)  # fused [q | k] rows like the engine's projection output
    v = rbf(torch.randn(b, t, c, generator=g))
    q, k = qk[..., :c], qk[..., c:]
    ref = _attn_ref(q, k, v, heads)
    d = _dev()
    tpad = (t + 63) // 64 * 64
    vt = torch.zeros(b, c, tpad, dtype=_eng().act_dtype(), device=d)
    vt[:, :, :t] = v.transpose(1, 2).to(d).to(_eng().act_dtype())
    qkd = qk.to(d).to(_eng().act_dtype())
    y = e.flash_attention(qkd[..., :c], qkd[..., c:], vt, heads)
    check(f"flash64{case}", y, ref, metric_log)


def test_flash_attention_spiky_scores(metric_log):
    """Online-softmax rescale path: one key dominates late in the sequence (cdna guide rule 26)."""
    e = _eng()
    g = torch.Generator().manual_seed(9)
    b, t, heads, c = 1, 320, 1, 64
    q = rbf(torch.randn(b, t, c, generator=g))
    k = rbf(torch.randn(b, t, c, generator=g))
    k[0, 250] = rbf(q[0, 7] * 6.0)  # huge score for query 7 at a late tile
    v = rbf(torch.randn(b, t, c, generator=g))
    ref = _attn_ref(q, k, v, heads)
    d = _dev()
    vt = torch.zeros(b, c, 320, dtype=_eng().act_dtype(), device=d)
    vt[:, :, :t] = v.transpose(1, 2).to(d).to(_eng().act_dtype())
    y = e.flash_attention(q.to(d).to(_eng().act_dtype()).contiguous(), k.to(d).to(_eng().act_dtype()).contiguous(), vt, heads)
    check("flash64_spiky", y, ref, metric_log)


@pytest.mark.parametrize("case", [(1, 64, 0), (1, 1131, 0), (2, 300, 4), (1, 1131, 4), (3, 200, 5), (2, 1000, 3), (1, 4100, 0), (1, 1000, 40)])
def test_flash_attention_hd512(case, metric_log):
    """The fused VAE mid-block attention (one head, head_dim 512).  ncu != 0 sizes the launch for that many workgroups: (2, 300, 4) = 6 query
    blocks on 4 -> one whole round + 2 left-over blocks cut into 2 key parts each; (1, 1131, 4) = 9 blocks -> two rounds + 1 block in 4 parts;
    (3, 200, 5) = 6 blocks -> 1 left-over block in 5 parts of 7 tiles (uneven); (2, 1000, 3) = 16 blocks -> 5 rounds + 1 block in 3 parts;
    fewer blocks than half the workgroups -> EVERY block is cut along the keys: (1, 4100, 0) = 33 blocks on 256 CUs in 7 parts each,
    (1, 1000, 40) = 8 blocks in 4 parts (the cap of eight tiles per part; 40 / 8 = 5 would leave parts of 6 tiles)."""
    e = _eng()
    b, t, ncu = case
    c = 512
    g = torch.Generator().manual_seed(t * 3 + b)
    qk = rbf(torch.randn(b, t, 2 * c, generator=g))
    v = rbf(torch.randn(b, t, c, generator=g))
    q, k = qk[..., :c], qk[..., c:]
    scale = 2.5 / c ** 0.5                      # logits of std 2.5: a peaked but not one-hot softmax
    ref = torch.softmax((q @ k.transpose(1, 2)) * scale, dim=-1) @ v
    d = _dev()
    tpad = (t + 63) // 64 * 64
    vt = torch.zeros(b, c, tpad, dtype=e.act_dtype(), device=d)
    vt[:, :, :t] = v.transpose(1, 2).to(d).to(e.act_dtype())
    qkd = qk.to(d).to(e.act_dtype())
    y = e.flash_attention_hd512(qkd[..., :c], qkd[..., c:], vt, scale, ncu)
    # the outputs are averages over hundreds of keys, all of similar magnitude: the rounding of the stored result alone is 2.1e-3 (bf16) /
    # 2.6e-4 (fp16) of the mean |ref|, i.e. exactly the generic mean gate; measured 1.9-2.1e-3 / 2.3-2.6e-4 in every case, split or not
    check(f"flash512{case}", y, ref, metric_log, mean_factor=1.25)
    if case == (2, 300, 4):
        # which query blocks are cut along the keys does not depend on the image's slot (each image gives its last L / B blocks):
        # a permuted batch gives the permuted result, bit for bit
        y2 = e.flash_attention_hd512(qkd.flip(0)[..., :c], qkd.flip(0)[..., c:], vt.flip(0).contiguous(), scale, ncu)
        assert torch.equal(y2.flip(0), y)


def test_flash_attention_hd512_outlier_logits(metric_log):
    """Logits far outside fp16's range (|q.k| * scale up to ~3e5) and a maximum that moves late in the sequence: the lazy rescale of the
    reference maximum (threshold 2^8) and the fp32 logits keep the result finite and equal to the fp32 softmax (r1 verdict item 2)."""
    e = _eng()
    b, t, c = 1, 700, 512
    g = torch.Generator().manual_seed(3)
    q = rbf(torch.randn(b, t, c, generator=g))
    k = rbf(torch.randn(b, t, c, generator=g))
    v = rbf(torch.randn(b, t, c, generator=g))
    k[0, 600] = rbf(q[0, 5] * 8.0)            # one huge score for query 5 in tile 18
    q[0, 100] = rbf(q[0, 100] * 40.0)         # a whole row of huge logits
    k[0, 650] = rbf(k[0, 650] * 30.0)         # a whole column of huge logits
    scale = 1.0
    ref = torch.softmax((q.double() @ k.double().transpose(1, 2)) * scale, dim=-1).float() @ v
    d = _dev()
    vt = torch.zeros(b, c, 704, dtype=e.act_dtype(), device=d)
    vt[:, :, :t] = v.transpose(1, 2).to(d).to(e.act_dtype())
    for ncu in (0, 2):
        y = e.flash_attention_hd512(q.to(d).to(e.act_dtype()).contiguous(), k.to(d).to(e.act_dtype()).contiguous(), vt, scale, ncu)
        assert torch.isfinite(y.float()).all()
        check(f"flash512_outliers[ncu={ncu}]", y, ref, metric_log)


@pytest.mark.parametrize("L", [2, 77])
def test_cross_attention_small(L, metric_log):
    e = _eng()
    g = torch.Generator().manual_seed(L)
    rows, c = 500, 320
    q = rbf(torch.randn(rows, c, generator=g))
    kc, vc = torch.randn(L, c, generator=g), torch.randn(L, c, generator=g)
    ref = _attn_ref(q[None], kc[None], vc[None], c // 64)[0]
    d = _dev()
    y = e.cross_attention(q.to(d).to(_eng().act_dtype()), kc.to(d), vc.to(d))
    check(f"cross_attn_L{L}", y, ref, metric_log)


@pytest.mark.parametrize("shape", [(37, 1000, 1024), (5, 9216, 9216), (3, 9000, 9216), (2, 12001, 12004), (2, 20000, 20000), (3, 50, 56),
                                   (4, 1001, 1002)])
def test_softmax_rows(shape, metric_log):
    """Row softmax of the GEMM-based VAE attention: register-resident kernel (row <= 16384, ld % 4 == 0) and the generic fallback."""
    e = _eng()
    g = torch.Generator().manual_seed(1)
    rows, t, ld = shape
    x = torch.randn(rows, ld, generator=g) * 20
    d = _dev()
    y = e.softmax_rows(x.to(d), t, 0.05)
    check(f"softmax_rows{shape}", y[:, :t], torch.softmax(x[:, :t] * 0.05, dim=-1), metric_log)
    if ld <= 16384 and ld % 4 == 0:  # fp16 logits (what the VAE attention's score GEMM writes)
        xh = x.to(torch.float16)
        yh = e.softmax_rows(xh.to(d), t, 0.05)
        check(f"softmax_rows_f16{shape}", yh[:, :t], torch.softmax(xh.float()[:, :t] * 0.05, dim=-1), metric_log)
    if ld > t:
        assert float(y[:, t:].float().abs().max()) == 0.0


@pytest.mark.parametrize("case", [((6, 6), (12, 12), True), ((5, 7), (10, 14), True), ((12, 12), (24, 24), False), ((9, 12), (18, 23), False)])
def test_bilinear(case, metric_log):
    e = _eng()
    (h, w), (ho, wo), align = case
    g = torch.Generator().manual_seed(h + wo)
    x = rbf(torch.randn(2, 64, h, w, generator=g))
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=align)
    d = _dev()
    y = e.bilinear(e.to_nhwc_bf16(x.to(d)), (ho, wo), align)
    check(f"bilinear{case}", nhwc_to_nchw(y), ref, metric_log)


HEAVY_CASES = [
    # (Cin, Cout, H = W of the INPUT, x2 upsample, fused GroupNorm+SiLU input) -- the heaviest 3x3 layers of the 768x768 pass
    # (SURVEY.md Appendix C), at BASELINE.json configs[1]'s batch 4: the persistent kernels size their grid as #CU / B, so B = 4 walks
    # different tile lists than the small cases above
    (512, 512, 192, False, False), (256, 256, 384, False, False), (128, 128, 768, False, False),
    (512, 512, 192, True, False), (256, 256, 384, True, False), (512, 256, 384, False, False), (256, 128, 768, False, False),
    (128, 128, 768, False, True), (256, 128, 768, False, True),
]


@pytest.mark.parametrize("case", HEAVY_CASES)
def test_heaviest_layers_batch4_vs_fp32(case, metric_log):
    """Per-layer parity at the benched batch: HIP conv (auto tile selection = what the engine launches) on [4, Cin, H, W] against an
    fp32 torch conv of the same bf16-rounded operands, evaluated on the host CPU for the first and the last image of the batch."""
    e = _eng()
    cin, cout, hw, ups, fused = case
    b = 4
    g = torch.Generator().manual_seed(cin + cout + hw + 7 * ups + 13 * fused)
    x = rbf(torch.randn(b, cin, hw, hw, generator=g) * (1.0 + torch.arange(b).float().view(b, 1, 1, 1) * 0.25))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    d = _dev()
    xd = e.to_nhwc_bf16(x.to(d))
    wp = e.pack_weight(wt, device=d)
    if fused:
        gamma = 1.0 + 0.1 * torch.randn(cin, generator=g)
        beta = 0.1 * torch.randn(cin, generator=g)
        y = e.conv2d_gn(xd, wp, bias.to(d), cout, gamma.to(d), beta.to(d), 32, 1e-6, silu=True)
    else:
        y = e.conv2d(xd, wp, bias.to(d), cout, 3, ups_hw=(2 * hw, 2 * hw) if ups else None, tile=0)
    torch.cuda.synchronize()
    yc = nhwc_to_nchw(y[[0, b - 1]]).cpu()
    del y, xd
    with torch.no_grad():
        xi = x[[0, b - 1]]
        if fused:
            xi = rbf(F.silu(F.group_norm(xi, 32, gamma, beta, 1e-6)))  # the kernel rounds the normalised operand to bf16 for the MFMA
        if ups:
            xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(xi, wt, bias, padding=1)
    check(f"heavy_b4{case}", yc, ref, metric_log)


@pytest.mark.parametrize("case", [(150, 320, 5), (37, 640, 10), (64, 1280, 20), (5, 256, 4), (20000, 320, 5), (5000, 128, 2),
                                  # C = 1280: two rows per wave from 512 rows, three from 2048 (r5; ragged row counts)
                                  (577, 1280, 20), (2304, 1280, 20), (2050, 1280, 20), (9217, 640, 10)])
def test_cross_attention_two_token_fold(case, metric_log):
    """attn2 of the BasicTransformerBlock against a 2-token context (GenPercept's empty prompt, genpercept_pipeline.py:425-429) folded into
    per-head vectors: the kernel must equal LayerNorm -> to_q -> softmax(q k^T / 8) v -> to_out -> + residual, then norm3, computed
    the long way in fp32 (oracle/sd21.transformer_2d's attn2 lines).  The fold is redone here in float64, independently of the engine's."""
    e = _eng()
    rows, c, heads = case
    g = torch.Generator().manual_seed(rows + c)
    y = rbf(torch.randn(rows, c, generator=g) * 1.5 + 0.3)
    wq, wo = torch.randn(c, c, generator=g) / math.sqrt(c), torch.randn(c, c, generator=g) / math.sqrt(c)
    bo = 0.1 * torch.randn(c, generator=g)
    kc, vc = torch.randn(2, c, generator=g), torch.randn(2, c, generator=g)
    g2, b2 = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    g3, b3 = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    n2 = F.layer_norm(y, (c,), g2, b2, 1e-5)
    q = (n2 @ wq.t()).view(rows, heads, 64)
    w = torch.softmax(torch.einsum("rhd,lhd->rhl", q, kc.view(2, heads, 64)) / 8.0, dim=-1)
    a = torch.einsum("rhl,lhd->rhd", w, vc.view(2, heads, 64)).reshape(rows, c)
    ref_y = y + a @ wo.t() + bo
    ref_n3 = F.layer_norm(rbf(ref_y), (c,), g3, b3, 1e-5)  # norm3 reads the trunk as stored
    dk, dv = (kc[0] - kc[1]).double(), (vc[0] - vc[1]).double()
    A = torch.stack([(wq.double()[h * 64:(h + 1) * 64] * dk[h * 64:(h + 1) * 64, None]).sum(0) for h in range(heads)]) / 8.0  # [heads][C]
    U, u0 = (A * g2.double()).float(), (A @ b2.double()).float()
    c0 = (wo.double() @ vc[1].double() + bo.double()).float()
    G = torch.stack([wo.double()[:, h * 64:(h + 1) * 64] @ dv[h * 64:(h + 1) * 64] for h in range(heads)]).float()
    d = _dev()
    yo, n3 = e.cross_attention_fold(y.to(d).to(e.act_dtype()), U.to(d).contiguous(), u0.to(d), G.to(d).contiguous(), c0.to(d), g3.to(d), b3.to(d))
    check(f"cross_fold_y{case}", yo, ref_y, metric_log)
    check(f"cross_fold_n3{case}", n3, ref_n3, metric_log)


def test_mfma_lds_probe_entry(metric_log):
    """gp_mfma_lds_probe (measurement probe of DESIGN.md section 5): supported combinations report a plausible rate, others are refused"""
    from genpercept_amd import engine as ge
    bare = ge.mfma_lds_probe(0, 0, 2, 0)
    conv = ge.mfma_lds_probe(0, 8, 2, 0)
    full = ge.mfma_lds_probe(0, 8, 2, 19)
    metric_log("mfma_lds_probe", bare=bare, with_reads=conv, with_barrier_dma=full)
    assert 500.0 < full <= conv * 1.05 and conv <= bare * 1.05 and bare < 2600.0
    assert ge.mfma_lds_probe(0, 3, 2, 0) < 0 and ge.mfma_lds_probe(0, 8, 1, 19) < 0 and ge.mfma_lds_probe(0, 8, 3, 0) < 0
