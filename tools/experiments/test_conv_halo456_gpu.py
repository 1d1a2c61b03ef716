"""GPU parity tests of the r4 experimental conv kernels (conv_halo4 / 5 / 6.hip in this directory).  Not collected by the suite: they need a
library built with those sources (tools/experiments/README.md).  Kept as they ran green in round 4."""
# helpers come from tests/test_kernels_gpu.py
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_kernels_gpu import *  # noqa: F401,F403
from test_kernels_gpu import _eng, _dev, _tol16  # noqa: F401


# conv_halo4.hip: 32 x 16-pixel tiles, 32-channel chunks, v_mfma_f32_32x32x16.  Forced through IGemmParams::dbg bits 20-21 (GENPERCEPT_IGEMM_DBG =
# 1 << 20; the launcher would otherwise pick it only where its tile count fills the persistent grid): whole tiles, ragged right / bottom
# edges, one and several tiles per workgroup, several channel slices incl. a ragged one, chunk counts 2 .. 10, with and without residual.
HALO4_CASES = [
    # B, H, W, Cin, Cout, residual
    (1, 16, 32, 64, 128, False), (2, 32, 64, 64, 128, True), (1, 48, 96, 128, 256, True), (1, 17, 33, 64, 64, True), (3, 40, 72, 192, 320, False),
    (1, 16, 40, 320, 200, True), (4, 144, 160, 64, 128, True), (1, 96, 96, 128, 512, False), (2, 31, 95, 256, 128, True),
]


@pytest.mark.parametrize("case", HALO4_CASES)
def test_conv3x3_halo4_kernel(case, metric_log, monkeypatch):
    e = _eng()
    b, h, w, cin, cout, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 4)
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
    y4 = e.conv2d(*args, **kw)
    check(f"conv_halo4{case}", nhwc_to_nchw(y4), ref, metric_log)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(2 << 20))
    y3 = e.conv2d(*args, **kw)  # the 16 x 16-tile kernel on the same operands: same products, other summation order
    dlt = (nhwc_to_nchw(y4) - nhwc_to_nchw(y3)).abs().max().item()
    metric_log(f"conv_halo4_vs_halo3{case}", max_abs=dlt)
    assert dlt <= 4 * _tol16()[0] * ref.abs().max().item()
    # run-to-run determinism (fixed tile walk, fixed summation order)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    assert torch.equal(e.conv2d(*args, **kw), y4)


# conv_halo5.hip: 16 x 16-pixel tiles, 32-channel chunks, TWO workgroups per CU (128 registers per wave, 78.5 KiB LDS each).  Forced through
# IGemmParams::dbg bits 28-29 (GENPERCEPT_IGEMM_DBG = 1 << 28; GENPERCEPT_HALO5=1 enables it in the launcher): whole tiles, ragged edges, one and
# several tiles per workgroup, several channel slices incl. a ragged one, chunk counts 2 .. 16, with and without residual.
HALO5_CASES = [
    # B, H, W, Cin, Cout, residual
    (1, 16, 16, 64, 128, False), (2, 32, 64, 64, 128, True), (1, 48, 96, 128, 256, True), (1, 17, 33, 64, 64, True), (3, 40, 72, 192, 320, False),
    (1, 16, 40, 320, 200, True), (4, 144, 160, 64, 128, True), (1, 96, 96, 128, 512, False), (2, 31, 95, 256, 128, True), (4, 192, 192, 512, 512, False),
]


@pytest.mark.parametrize("case", HALO5_CASES)
def test_conv3x3_halo5_kernel(case, metric_log, monkeypatch):
    e = _eng()
    b, h, w, cin, cout, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 5)
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
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 28))
    y5 = e.conv2d(*args, **kw)
    check(f"conv_halo5{case}", nhwc_to_nchw(y5), ref, metric_log)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(2 << 28))
    y3 = e.conv2d(*args, **kw)  # conv3x3_halo3_kernel on the same operands: same products, other summation order (32- instead of 64-channel steps)
    dlt = (nhwc_to_nchw(y5) - nhwc_to_nchw(y3)).abs().max().item()
    metric_log(f"conv_halo5_vs_halo3{case}", max_abs=dlt)
    assert dlt <= 4 * _tol16()[0] * ref.abs().max().item()
    # run-to-run determinism (fixed tile walk, fixed summation order; two workgroups per CU change the timing, not the order)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 28))
    for _ in range(3):
        assert torch.equal(e.conv2d(*args, **kw), y5)


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 128, True), (1, 40, 72, 64, 192, False), (4, 144, 160, 64, 128, True), (3, 112, 112, 64, 256, False),
                                  (1, 17, 33, 64, 320, True)])
def test_conv_halo5_groupnorm_statistics(case, metric_log, monkeypatch):
    """the statistics the two-workgroups-per-CU kernel leaves for the next GroupNorm (per-workgroup partial rows + pixel counts, finalised to scale /
    shift) must be those of the tensor it stored"""
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 28))
    e = _eng()
    b, h, w, cin, cout, with_res = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(cin + cout + h + 5)
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
    check(f"conv_halo5_stats_out{case}", nhwc_to_nchw(y), ref, metric_log)
    yg = nhwc_to_nchw(y).float().cpu().reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    cpg = cout // groups
    sc_ref = gamma[None, :] * (var + eps).rsqrt().repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_halo5_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 128, True), (1, 40, 72, 64, 192, False), (4, 144, 160, 64, 128, True), (3, 112, 112, 64, 256, False),
                                  (1, 17, 33, 64, 320, True)])
def test_conv_halo4_groupnorm_statistics(case, metric_log, monkeypatch):
    """the statistics the 512-pixel-tile kernel leaves for the next GroupNorm (per-workgroup partial rows + pixel counts, finalised to scale / shift)
    must be those of the tensor it stored"""
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(1 << 20))
    e = _eng()
    b, h, w, cin, cout, with_res = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(cin + cout + h + 4)
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
    check(f"conv_halo4_stats_out{case}", nhwc_to_nchw(y), ref, metric_log)
    yg = nhwc_to_nchw(y).float().cpu().reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    cpg = cout // groups
    sc_ref = gamma[None, :] * (var + eps).rsqrt().repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_halo4_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


# conv_halo6.hip: Winograd F(2, 3) along x (2/3 of the MFMA work); forced through IGemmParams::dbg bits 28-29 = 3; GENPERCEPT_WINO=1 enables it
@pytest.mark.parametrize("case", HALO5_CASES)
def test_conv3x3_halo6_winograd_kernel(case, metric_log, monkeypatch):
    e = _eng()
    b, h, w, cin, cout, with_res = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:5]) + 6)
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
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(3 << 28))
    y6 = e.conv2d(*args, **kw)
    # the transformed operands (V = B^T d, U = G g) are rounded AFTER their additions: per layer 1.3-1.5x the direct conv's mean error (measured);
    # end to end nothing (profiles/r04_precision_ablation.json, rows *winograd1d*)
    check(f"conv_halo6{case}", nhwc_to_nchw(y6), ref, metric_log, mean_factor=1.8)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(2 << 28))
    y3 = e.conv2d(*args, **kw)
    dlt = (nhwc_to_nchw(y6) - nhwc_to_nchw(y3)).abs().max().item()
    metric_log(f"conv_halo6_vs_halo3{case}", max_abs=dlt)
    assert dlt <= 6 * _tol16()[0] * ref.abs().max().item()  # (the transformed operands are rounded after their additions)
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(3 << 28))
    for _ in range(2):
        assert torch.equal(e.conv2d(*args, **kw), y6)


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 128, True), (1, 40, 72, 64, 192, False), (4, 144, 160, 64, 128, True), (1, 17, 33, 64, 320, True)])
def test_conv_halo6_groupnorm_statistics(case, metric_log, monkeypatch):
    monkeypatch.setenv("GENPERCEPT_IGEMM_DBG", str(3 << 28))
    e = _eng()
    b, h, w, cin, cout, with_res = case
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(cin + cout + h + 6)
    x = rbf(torch.randn(b, cin, h, w, generator=g))
    wt = rbf(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    bias = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    res = rbf(torch.randn(b, cout, h, w, generator=g)) if with_res else None
    d = _dev()
    y, scale, shift = e.conv2d_stats(e.to_nhwc_bf16(x.to(d)), e.pack_weight(wt, device=d), bias.to(d), cout, 3, gamma.to(d), beta.to(d), groups, eps,
                                     ups=False, residual=e.to_nhwc_bf16(res.to(d)) if with_res else None, tile=5)
    yg = nhwc_to_nchw(y).float().cpu().reshape(b, groups, -1)
    mean, var = yg.mean(dim=2), yg.var(dim=2, unbiased=False)
    cpg = cout // groups
    sc_ref = gamma[None, :] * (var + eps).rsqrt().repeat_interleave(cpg, dim=1)
    sh_ref = beta[None, :] - mean.repeat_interleave(cpg, dim=1) * sc_ref
    e_sc = ((scale.cpu() - sc_ref).abs() / sc_ref.abs().clamp_min(1e-3)).max().item()
    e_sh = (shift.cpu() - sh_ref).abs().max().item()
    metric_log(f"conv_halo6_stats{case}", scale_rel=e_sc, shift_abs=e_sh)
    assert e_sc < 2e-4 and e_sh < 2e-4, (e_sc, e_sh)


