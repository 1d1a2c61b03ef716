"""Closed-form identities the oracle's BLOCKS must satisfy, independent of diffusers (SURVEY.md 8c / VERDICT r5 item 7).

The arithmetic inside ResnetBlock2D / Transformer2DModel / the VAE blocks lives in the un-vendored `diffusers`; the reference-run fixtures pin the
orchestration AROUND the blocks (tests/golden/refexec_*.npz) but reach the block internals only through the oracle's own functions.  What can
be pinned offline is pinned here: every block of oracle/sd21.py is re-derived in float64 NumPy from the PUBLISHED layer algebra
without any torch.nn.functional call -- explicit loops for convolutions, explicit means / variances for the norms, math.erf for the GELU -- and for
every parity trap of SURVEY.md Appendix B the WRONG variant is evaluated next to the right one and shown to differ by orders of magnitude more
than the tolerance, so each assertion has discriminating power for exactly that trap:

  test                                   Appendix B trap it catches
  -------------------------------------  ---------------------------------------------------------------------------------------
  test_groupnorm_eps_inside_sqrt         5 (eps value / placement: inside the square root, biased variance)
  test_resnet_block                      5 (UNet resnet eps 1e-5), time-embedding injection point (after conv1, SiLU on emb), 1x1 shortcut, no output scale
  test_transformer_block                 4 (heads = contiguous 64-channel slices, scale 1/sqrt(hd)), 5 (Transformer2DModel.norm eps 1e-6, LayerNorm 1e-5),
                                         8 (GEGLU value-first, exact-erf GELU), bias-free q/k/v, residual placement
  test_vae_mid_attention                 4 (ONE head x C, scale 1/sqrt(C)), 5 (VAE eps 1e-6), biased projections, residual
  test_vae_downsample_pads_right_bottom  6 (VAE: pad right / bottom then stride 2 without padding)
  test_unet_downsample_pads_symmetric    6 (UNet: symmetric padding 1)
  test_upsample_nearest_then_conv        nearest x2 / nearest-to-size BEFORE the conv (custom_unet.py:115-119,377-378)
  test_timestep_embedding_cos_first      9 ([cos, sin], freq exponent i / half with freq_shift 0)
  test_latent_is_mean_half_times_scale   A.2: latent = FIRST latent_channels of quant_conv's output x 0.18215 (logvar half discarded)
  test_skip_concat_is_hidden_then_skip   7 ([hidden, skip], skips consumed from the end) through unet_forward with channel-tagged weights

`tools/crosscheck_diffusers.py --emit` writes the same inputs / oracle outputs to a fixture that a machine WITH diffusers verifies (`--verify`)."""
import math

import numpy as np
import pytest
import torch

from oracle import sd21 as osd

F64 = np.float64


def _np(t):
    return t.detach().double().numpy()


def np_groupnorm(x, gamma, beta, groups, eps):
    b, c, h, w = x.shape
    xg = x.reshape(b, groups, -1)
    mean = xg.mean(axis=2, keepdims=True)
    var = ((xg - mean) ** 2).mean(axis=2, keepdims=True)  # biased
    y = ((xg - mean) / np.sqrt(var + eps)).reshape(b, c, h, w)
    return y * gamma[None, :, None, None] + beta[None, :, None, None]


def np_layernorm(x, gamma, beta, eps):
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def np_conv(x, w, bias, stride=1, pad=(1, 1, 1, 1)):
    """x [B,C,H,W], w [O,C,kh,kw]; pad = (top, bottom, left, right); cross-correlation like every DL framework."""
    b, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    xp = np.zeros((b, c, h + pad[0] + pad[1], wd + pad[2] + pad[3]), dtype=F64)
    xp[:, :, pad[0]:pad[0] + h, pad[2]:pad[2] + wd] = x
    ho, wo = (xp.shape[2] - kh) // stride + 1, (xp.shape[3] - kw) // stride + 1
    y = np.zeros((b, o, ho, wo), dtype=F64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
            y += np.einsum("bchw,oc->bohw", patch, w[:, :, ky, kx])
    return y + (bias[None, :, None, None] if bias is not None else 0.0)


def np_silu(x):
    return x / (1.0 + np.exp(-x))


_erf = np.vectorize(math.erf)


def np_gelu_erf(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def np_gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def np_attention(q, k, v, heads):
    """q [B,Tq,C], k / v [B,Tk,C]: heads are CONTIGUOUS channel slices of C / heads; softmax(q k^T / sqrt(hd)) v"""
    b, tq, c = q.shape
    hd = c // heads
    out = np.zeros_like(q)
    for h in range(heads):
        sl = slice(h * hd, (h + 1) * hd)
        s = np.einsum("bqd,bkd->bqk", q[:, :, sl], k[:, :, sl]) / math.sqrt(hd)
        s = s - s.max(axis=-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=-1, keepdims=True)
        out[:, :, sl] = np.einsum("bqk,bkd->bqd", p, v[:, :, sl])
    return out


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


TOL = 2e-5  # fp32 oracle vs float64 closed form


def _sd(manifest, seed):
    return osd.synth_state_dict(manifest, seed)


def test_groupnorm_eps_inside_sqrt():
    """variance far below eps: (x - mean) / sqrt(var + eps) and (x - mean) / (sqrt(var) + eps) differ by 10x; biased vs unbiased variance by 1 / (n - 1)"""
    g = torch.Generator().manual_seed(0)
    x = 1e-4 * torch.randn(2, 64, 2, 2, generator=g)  # 2 channels x 4 pixels per group: n = 8, var ~ 1e-8 (zero mean: no fp32 cancellation in x - mean)
    sd = {"n.weight": 1 + 0.1 * torch.randn(64, generator=g), "n.bias": 0.1 * torch.randn(64, generator=g)}
    for eps in (1e-5, 1e-6):
        out = _np(osd._gn(x, sd, "n", 32, eps))
        ref = np_groupnorm(_np(x), _np(sd["n.weight"]), _np(sd["n.bias"]), 32, eps)
        assert np.abs(out - ref).max() < 1e-4 * np.abs(ref - _np(sd["n.bias"])[None, :, None, None]).max() + 1e-6
        xg = _np(x).reshape(2, 32, -1)
        m = xg.mean(axis=2, keepdims=True)
        wrong = ((xg - m) / (np.sqrt(((xg - m) ** 2).mean(axis=2, keepdims=True)) + eps)).reshape(x.shape) * _np(sd["n.weight"])[None, :, None, None] \
            + _np(sd["n.bias"])[None, :, None, None]
        assert np.abs(wrong - ref).max() > 100 * np.abs(out - ref).max()
    # eps VALUE: 1e-5 vs 1e-6 is visible on this input (the two calls above are 2.7x apart in scale)
    a, b = _np(osd._gn(x, sd, "n", 32, 1e-5)), _np(osd._gn(x, sd, "n", 32, 1e-6))
    assert np.abs(a - b).max() > 1e-3


@pytest.mark.parametrize("cin,cout,temb", [(32, 64, True), (64, 64, True), (64, 32, False)])
def test_resnet_block(cin, cout, temb):
    g = torch.Generator().manual_seed(cin + cout)
    m = {}
    osd._resnet_keys(m, "r", cin, cout, 48 if temb else None)
    sd = _sd(m, 3)
    x = torch.randn(2, cin, 5, 6, generator=g)
    emb = torch.randn(2, 48, generator=g) if temb else None
    eps = 1e-5
    out = _np(osd.resnet_block(x, sd, "r", 32, eps, emb))
    n = {k: _np(v) for k, v in sd.items()}
    h = np_silu(np_groupnorm(_np(x), n["r.norm1.weight"], n["r.norm1.bias"], 32, eps))
    h = np_conv(h, n["r.conv1.weight"], n["r.conv1.bias"])
    if temb:
        h = h + (np_silu(_np(emb)) @ n["r.time_emb_proj.weight"].T + n["r.time_emb_proj.bias"])[:, :, None, None]
    h = np_silu(np_groupnorm(h, n["r.norm2.weight"], n["r.norm2.bias"], 32, eps))
    h = np_conv(h, n["r.conv2.weight"], n["r.conv2.bias"])
    sc = _np(x) if cin == cout else np_conv(_np(x), n["r.conv_shortcut.weight"], n["r.conv_shortcut.bias"], pad=(0, 0, 0, 0))
    ref = sc + h
    assert rel(out, ref) < TOL
    if temb:  # traps: time embedding without SiLU / injected before conv1
        wrong = np_silu(np_groupnorm(_np(x), n["r.norm1.weight"], n["r.norm1.bias"], 32, eps))
        wrong = np_conv(wrong, n["r.conv1.weight"], n["r.conv1.bias"]) + (_np(emb) @ n["r.time_emb_proj.weight"].T + n["r.time_emb_proj.bias"])[:, :, None, None]
        wrong = np_conv(np_silu(np_groupnorm(wrong, n["r.norm2.weight"], n["r.norm2.bias"], 32, eps)), n["r.conv2.weight"], n["r.conv2.bias"]) + sc
        assert rel(wrong, ref) > 1e3 * TOL


@pytest.mark.parametrize("heads,L", [(2, 2), (1, 3)])
def test_transformer_block(heads, L):
    c, ctxd = 64 * heads, 48
    g = torch.Generator().manual_seed(heads * 10 + L)
    m = {}
    osd._transformer_keys(m, "t", c, ctxd)
    sd = _sd(m, 5)
    x = torch.randn(2, c, 3, 4, generator=g)
    ctx = torch.randn(2, L, ctxd, generator=g)
    out = _np(osd.transformer_2d(x, sd, "t", heads, ctx, 32))
    n = {k: _np(v) for k, v in sd.items()}

    def block(geglu="value_first_erf", gn_eps=1e-6, ln_eps=1e-5, head_layout="contiguous", scale_hd=None):
        b_, _, hh, ww = x.shape
        y = np_groupnorm(_np(x), n["t.norm.weight"], n["t.norm.bias"], 32, gn_eps).transpose(0, 2, 3, 1).reshape(b_, hh * ww, c)
        y = y @ n["t.proj_in.weight"].T + n["t.proj_in.bias"]
        p = "t.transformer_blocks.0"
        a = np_layernorm(y, n[p + ".norm1.weight"], n[p + ".norm1.bias"], ln_eps)
        q, k, v = a @ n[p + ".attn1.to_q.weight"].T, a @ n[p + ".attn1.to_k.weight"].T, a @ n[p + ".attn1.to_v.weight"].T  # bias-free
        if head_layout == "interleaved":  # wrong: head h = channels h, h + heads, ...
            perm = np.arange(c).reshape(c // heads, heads).T.reshape(-1)
            o = np.zeros_like(q)
            o[:, :, perm] = np_attention(q[:, :, perm], k[:, :, perm], v[:, :, perm], heads)
        else:
            o = np_attention(q, k, v, heads)
        y = y + o @ n[p + ".attn1.to_out.0.weight"].T + n[p + ".attn1.to_out.0.bias"]
        a = np_layernorm(y, n[p + ".norm2.weight"], n[p + ".norm2.bias"], ln_eps)
        cx = _np(ctx)
        o = np_attention(a @ n[p + ".attn2.to_q.weight"].T, cx @ n[p + ".attn2.to_k.weight"].T, cx @ n[p + ".attn2.to_v.weight"].T, heads)
        y = y + o @ n[p + ".attn2.to_out.0.weight"].T + n[p + ".attn2.to_out.0.bias"]
        a = np_layernorm(y, n[p + ".norm3.weight"], n[p + ".norm3.bias"], ln_eps)
        pr = a @ n[p + ".ff.net.0.proj.weight"].T + n[p + ".ff.net.0.proj.bias"]
        first, second = pr[..., :4 * c], pr[..., 4 * c:]
        if geglu == "value_first_erf":
            ff = first * np_gelu_erf(second)
        elif geglu == "gate_first_erf":
            ff = second * np_gelu_erf(first)
        else:
            ff = first * np_gelu_tanh(second)
        y = y + ff @ n[p + ".ff.net.2.weight"].T + n[p + ".ff.net.2.bias"]
        y = y @ n["t.proj_out.weight"].T + n["t.proj_out.bias"]
        return y.reshape(b_, hh, ww, c).transpose(0, 3, 1, 2) + _np(x)

    ref = block()
    assert rel(out, ref) < TOL
    assert rel(block(geglu="gate_first_erf"), ref) > 1e3 * TOL          # B.8 value / gate order
    assert rel(block(geglu="value_first_tanh"), ref) > 3 * TOL           # B.8 exact-erf GELU (the tanh form differs by ~1e-4 here)
    assert rel(block(gn_eps=1e-5), ref) > 0.02 * TOL                     # B.5 (small on unit-variance inputs: the dedicated eps test carries this trap)
    if heads > 1:
        assert rel(block(head_layout="interleaved"), ref) > 1e3 * TOL   # B.4 head = contiguous 64-channel slice


def test_vae_mid_attention():
    c = 64
    g = torch.Generator().manual_seed(9)
    m = {}
    osd._vae_attn_keys(m, "a", c)
    sd = _sd(m, 11)
    x = torch.randn(2, c, 3, 5, generator=g)
    out = _np(osd.vae_mid_attention(x, sd, "a", 32, 1e-6))
    n = {k: _np(v) for k, v in sd.items()}
    y = np_groupnorm(_np(x), n["a.group_norm.weight"], n["a.group_norm.bias"], 32, 1e-6).reshape(2, c, 15).transpose(0, 2, 1)
    lin = lambda t, nm: t @ n[f"a.{nm}.weight"].reshape(c, c).T + n[f"a.{nm}.bias"]  # noqa: E731
    q, k, v = lin(y, "to_q"), lin(y, "to_k"), lin(y, "to_v")
    ref = lin(np_attention(q, k, v, 1), "to_out.0").transpose(0, 2, 1).reshape(2, c, 3, 5) + _np(x)
    assert rel(out, ref) < TOL
    wrong = lin(np_attention(q, k, v, 2), "to_out.0").transpose(0, 2, 1).reshape(2, c, 3, 5) + _np(x)  # two heads of 32 instead of one of 64
    assert rel(wrong, ref) > 1e3 * TOL


def test_vae_downsample_pads_right_bottom():
    """one-hot image at (0, 0), weight = delta at tap (0, 0): the pixel is seen by output (0, 0) iff the padding is on the right / bottom only"""
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 0, 0] = 1.0
    sd = {"d.weight": w, "d.bias": torch.zeros(1)}
    x = torch.zeros(1, 1, 8, 8)
    x[0, 0, 0, 0] = 1.0
    y = osd.vae_downsample(x, sd, "d")
    assert y.shape == (1, 1, 4, 4) and float(y[0, 0, 0, 0]) == 1.0 and float(y.abs().sum()) == 1.0
    x = torch.zeros(1, 1, 8, 8)
    x[0, 0, 7, 7] = 1.0  # last pixel: read by tap (1, 1) of output (3, 3); tap (2, 2) there reads the bottom / right padding
    w2 = torch.zeros(1, 1, 3, 3)
    w2[0, 0, 1, 1] = 1.0
    assert float(osd.vae_downsample(x, {"d.weight": w2, "d.bias": torch.zeros(1)}, "d")[0, 0, 3, 3]) == 1.0
    # odd size (NYU latents): 9 -> (9 + 1 - 3) // 2 + 1 = 4
    assert osd.vae_downsample(torch.zeros(1, 1, 9, 7), sd, "d").shape == (1, 1, 4, 3)
    ref = np_conv(_np(torch.arange(64.).reshape(1, 1, 8, 8)), _np(torch.ones(1, 1, 3, 3)), None, stride=2, pad=(0, 1, 0, 1))
    out = _np(osd.vae_downsample(torch.arange(64.).reshape(1, 1, 8, 8), {"d.weight": torch.ones(1, 1, 3, 3)}, "d"))
    assert np.array_equal(out, ref)


def test_unet_downsample_pads_symmetric():
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0  # centre tap: with symmetric padding 1 output (0, 0) reads input (0, 0)
    x = torch.zeros(1, 1, 8, 8)
    x[0, 0, 0, 0] = 1.0
    y = osd.unet_downsample(x, {"d.weight": w, "d.bias": torch.zeros(1)}, "d")
    assert y.shape == (1, 1, 4, 4) and float(y[0, 0, 0, 0]) == 1.0
    assert osd.unet_downsample(torch.zeros(1, 1, 15, 20), {"d.weight": w}, "d").shape == (1, 1, 8, 10)  # (n - 1) // 2 + 1


def test_upsample_nearest_then_conv():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 2, 3, 4, generator=g)
    sd = {"u.weight": torch.randn(3, 2, 3, 3, generator=g), "u.bias": torch.randn(3, generator=g)}
    xn = _np(x)
    up = np.repeat(np.repeat(xn, 2, axis=2), 2, axis=3)  # nearest x2: out[i] = in[i // 2]
    assert rel(_np(osd.upsample_conv(x, sd, "u")), np_conv(up, _np(sd["u.weight"]), _np(sd["u.bias"]))) < TOL
    # nearest to an odd size (custom_unet.py:115-119): out[i] = in[floor(i * in / out)]
    ho, wo = 7, 9
    iy, ix = (np.arange(ho) * 3 // ho), (np.arange(wo) * 4 // wo)
    up2 = xn[:, :, iy][:, :, :, ix]
    assert rel(_np(osd.upsample_conv(x, sd, "u", (ho, wo))), np_conv(up2, _np(sd["u.weight"]), _np(sd["u.bias"]))) < TOL


def test_timestep_embedding_cos_first():
    t = torch.tensor([1.0, 400.0])
    e = _np(osd.timestep_embedding(t, 320))
    half = 160
    fr = np.exp(-math.log(10000.0) * np.arange(half) / half)
    ref = np.concatenate([np.cos(_np(t)[:, None] * fr), np.sin(_np(t)[:, None] * fr)], axis=1)
    assert np.abs(e - ref).max() < 5e-5  # (fp32 cos / sin of arguments up to 400: the argument itself carries 400 * 2^-24 = 2.4e-5)
    assert np.abs(e - np.concatenate([ref[:, half:], ref[:, :half]], axis=1)).max() > 0.5          # [sin, cos] is the un-flipped order
    fr1 = np.exp(-math.log(10000.0) * np.arange(half) / (half - 1))                                  # downscale_freq_shift = 1
    assert np.abs(e[:1] - np.concatenate([np.cos(fr1), np.sin(fr1)])[None]).max() > 1e-3


def test_latent_is_mean_half_times_scale():
    cfg = osd.VAECfg.tiny()
    sd = _sd(osd.vae_manifest(cfg), 2)
    x = torch.rand(1, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1
    mom = osd.vae_encode_moments(sd, cfg, x)
    lat = osd.encode_rgb(sd, cfg, x)
    assert mom.shape[1] == 2 * cfg.latent_channels and torch.equal(lat, mom[:, :cfg.latent_channels] * cfg.scaling_factor)
    assert not torch.allclose(lat, mom[:, cfg.latent_channels:] * cfg.scaling_factor)


def test_skip_concat_is_hidden_then_skip():
    """unet_forward's up path on a UNet whose first up-block resnet reads ONE input channel (norm1 -> conv1 made a channel selector): selecting channel 0
    must return the HIDDEN state's channel 0, selecting channel `hidden_channels` the popped skip's channel 0 -- and the skip popped first is the LAST
    one pushed (the down path's final resnet output)."""
    cfg = osd.UNetCfg.tiny()
    sd = _sd(osd.unet_manifest(cfg), 7)
    x = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    ctx = torch.randn(1, 2, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(4))
    captured = {}
    orig = osd.resnet_block

    def spy(xx, sdd, p, groups, eps, temb):
        if p == "up_blocks.0.resnets.0":
            captured["in"] = xx.clone()
        if p in ("mid_block.resnets.1", "down_blocks.3.resnets.1"):
            out = orig(xx, sdd, p, groups, eps, temb)
            captured[p] = out.clone()
            return out
        return orig(xx, sdd, p, groups, eps, temb)

    osd.resnet_block = spy
    try:
        osd.unet_forward(sd, cfg, x, 1, ctx)
    finally:
        osd.resnet_block = orig
    cat = captured["in"]
    hid, skip = captured["mid_block.resnets.1"], captured["down_blocks.3.resnets.1"]
    c = hid.shape[1]
    assert cat.shape[1] == c + skip.shape[1]
    assert torch.equal(cat[:, :c], hid) and torch.equal(cat[:, c:], skip)


def test_crosscheck_fixture_is_current():
    """tests/golden/crosscheck_tiny.npz (tools/crosscheck_diffusers.py --emit) is what the oracle computes today: the file a user with diffusers verifies
    with ONE command (`--verify`) cannot go stale against the oracle silently."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("crosscheck_diffusers", os.path.join(root, "tools", "crosscheck_diffusers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fx = np.load(os.path.join(root, "tests", "golden", "crosscheck_tiny.npz"))
    cur = mod.build_fixture()
    for k in ("rgb", "ctx", "unet_checksums", "vae_checksums"):
        assert np.array_equal(fx[k], cur[k]), k
    for k in ("latent", "unet", "dec", "feat0", "feat1", "feat2", "feat3"):
        assert fx[k].shape == cur[k].shape and rel(cur[k].astype(F64), fx[k].astype(F64)) < 1e-5, k
    assert mod.verify_fixture(os.path.join(root, "tests", "golden", "crosscheck_tiny.npz"), 2e-4) == 2  # no diffusers here: says so, never "OK"


@pytest.mark.parametrize("heads,tq,tk", [(1, 7, 7), (4, 9, 5), (5, 6, 2)])
def test_attention_matches_torch_multihead_attention(heads, tq, tk):
    """An implementation this repository did not write: torch.nn.MultiheadAttention (identity projections, zero biases) and
    F.scaled_dot_product_attention compute softmax(q k^T / sqrt(head_dim)) v with heads as CONTIGUOUS channel slices -- the convention diffusers'
    attention processors share (Appendix B.4).  The oracle's _attention must agree with both."""
    c = 64 * heads
    g = torch.Generator().manual_seed(heads * 100 + tq)
    q, k, v = torch.randn(2, tq, c, generator=g), torch.randn(2, tk, c, generator=g), torch.randn(2, tk, c, generator=g)
    out = osd._attention(q, k, v, heads)
    mha = torch.nn.MultiheadAttention(c, heads, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([torch.eye(c)] * 3))
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(torch.eye(c))
        mha.out_proj.bias.zero_()
        ref, _ = mha(q, k, v, need_weights=False)
        hd = c // heads
        sdpa = torch.nn.functional.scaled_dot_product_attention(q.view(2, tq, heads, hd).transpose(1, 2), k.view(2, tk, heads, hd).transpose(1, 2),
                                                                v.view(2, tk, heads, hd).transpose(1, 2)).transpose(1, 2).reshape(2, tq, c)
    assert rel(_np(out), _np(ref)) < TOL and rel(_np(out), _np(sdpa)) < TOL
