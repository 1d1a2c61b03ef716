"""The F(2, 3) identities conv3x3_halo6_kernel is built on (genpercept_amd/csrc/conv_halo6.hip), checked on the CPU in fp64 and tied to the formulas in
the source: input transform V = B^T d, weight transform U = G g (wino_weights_kernel), output transform y = A^T M, applied along x with direct
taps along y -- and the plane order the kernel streams its weights in ([n][4 ky + p][Cin])."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _src():
    return open(os.path.join(ROOT, "tools", "experiments", "conv_halo6.hip")).read()


def test_source_uses_these_transforms():
    s = _src()
    for line in ["v0[w] = pack_h16x2(a0 - a2, b0 - b2);", "v1[w] = pack_h16x2(a1 + a2, b1 + b2);", "v2[w] = pack_h16x2(a2 - a1, b2 - b1);",
                 "v3[w] = pack_h16x2(a1 - a3, b1 - b3);",
                 "dst[0] = f_to_h16(g0);", "dst[Cin] = f_to_h16(0.5f * ((g0 + g1) + g2));", "dst[2 * Cin] = f_to_h16(0.5f * ((g0 - g1) + g2));",
                 "dst[3 * Cin] = f_to_h16(g2);"]:
        assert line in s, line
    assert re.search(r"y0a = \(\(acc\[0\]\[2 \* c\]\[jj\] \+ acc\[1\]\[2 \* c\]\[jj\]\) \+ acc\[2\]\[2 \* c\]\[jj\]\) \+ bv0;", s)
    assert re.search(r"y1a = \(\(acc\[1\]\[2 \* c\]\[jj\] - acc\[2\]\[2 \* c\]\[jj\]\) - acc\[3\]\[2 \* c\]\[jj\]\) \+ bv0;", s)


def _winograd_x(x, w):
    """x [C][H+2][W+2] (zero-padded input), w [O][C][3][3] -> y [O][H][W] through the kernel's formulas (W even)"""
    C, hp, wp = x.shape
    H, W = hp - 2, wp - 2
    O = w.shape[0]
    # U[o][c][ky][p]
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    U = np.stack([g0, 0.5 * ((g0 + g1) + g2), 0.5 * ((g0 - g1) + g2), g2], axis=-1)
    y = np.zeros((O, H, W))
    for q in range(W // 2):
        d0, d1, d2, d3 = (x[:, :, 2 * q + e] for e in range(4))          # [C][H+2]
        V = np.stack([d0 - d2, d1 + d2, d2 - d1, d1 - d3], axis=-1)      # [C][H+2][4]
        M = np.zeros((O, H, 4))
        for ky in range(3):
            M += np.einsum("ocp,chp->ohp", U[:, :, ky, :], V[:, ky:ky + H, :])
        y[:, :, 2 * q] = M[..., 0] + M[..., 1] + M[..., 2]
        y[:, :, 2 * q + 1] = M[..., 1] - M[..., 2] - M[..., 3]
    return y


def test_f23_along_x_equals_the_direct_conv():
    rng = np.random.default_rng(0)
    C, O, H, W = 5, 4, 6, 8
    x = np.zeros((C, H + 2, W + 2))
    x[:, 1:-1, 1:-1] = rng.standard_normal((C, H, W))
    w = rng.standard_normal((O, C, 3, 3))
    ref = np.zeros((O, H, W))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("oc,chw->ohw", w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W])
    assert np.abs(_winograd_x(x, w) - ref).max() < 1e-12


def test_plane_order_of_the_weight_stream():
    """plane index 4 ky + p; step t of a chunk reads planes 2 t and 2 t + 1 = (ky = t >> 1, positions 2 (t & 1), 2 (t & 1) + 1)"""
    s = _src()
    assert "h16_t* dst = u + row * (12LL * Cin) + (long long)(4 * ky) * Cin + c;" in s
    assert "PI = PL % 12, KY = PI >> 2, POS = PI & 3, SLOT = PL % 6;" in s
    for t in range(6):
        for pp in range(2):
            pl = 2 * t + pp
            assert (pl >> 2, pl & 3) == (t >> 1, 2 * (t & 1) + pp)
