"""LDS bank-model checks of the r4 experimental conv kernels (moved out of tests/ with their sources; run: pytest tools/experiments)"""
"""LDS layout claims of the MFMA kernels, checked on the CPU against the ds_read_b128 banking model of MI355X_MICROARCH.md
(64 banks x 4 B = 256-byte bank row; a wave's b128 read is served in four groups of 16 lanes, only lanes of one group can conflict,
identical addresses broadcast).  The swizzle constants are duplicated here on purpose and asserted to still be the ones in the sources,
so a kernel edit that changes a swizzle has to come back through this model.

  * pixel / A tiles ([rows][64 bf16] = 128-byte rows, 16-byte slots): physical slot = slot ^ (row & 7)            common.h swz_slot
  * 18-wide conv halo: key = hx & 7; 10-wide halo of the x2-upsample conv: nibble table 0x4016642254              conv_halo.hip halo_key
  * weight tiles read as rows {8 (a >> 2) + 4 h + (a & 3)}: key = b1 | b3 << 1 | b4 << 2 of the row               igemm / conv_halo / pgemm
  * attention K / V^T tiles read as 32 consecutive rows by the 32x32x16 MFMA: key = (row >> 1) & 7                attention.hip attn_off128
  * head_dim-512 attention: K tile rows of 1 KiB (64 slots), key = row & 15 on the low four slot bits; V^T tile rows of 64 bytes
    (4 slots), key = (row >> 2) & 3                                                                              attention.hip k512_off / v512_off
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "tools", "experiments")

# ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS table)
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def conflicts(addr16_of_lane):
    """extra LDS cycles of one wave-wide ds_read_b128: lanes give their address in 16-byte units; a 256-byte bank row has 16 of them."""
    extra = 0
    for g in GROUPS:
        by_bank = {}
        for lane in g:
            a = addr16_of_lane(lane)
            by_bank.setdefault(a % 16, set()).add(a)
        extra += sum(len(v) - 1 for v in by_bank.values())
    return extra


def src(name):
    with open(os.path.join(CSRC, name)) as f:
        return f.read()


def write_conflicts(addr16_of_lane):
    extra = 0
    for g0 in range(0, 64, 8):
        by_bank = {}
        for lane in range(g0, g0 + 8):
            a = addr16_of_lane(lane)
            by_bank.setdefault(a % 8, set()).add(a)
        extra += sum(len(v) - 1 for v in by_bank.values())
    return extra


def test_conv_halo4_layouts_are_conflict_free():
    """conv_halo4.hip: 64-byte LDS rows (four 16-byte slots) read by the 32x32x16 MFMA (lane = row l & 31, k-half l >> 5)."""
    s = src("conv_halo4.hip")
    assert "GP_DEV int h4_key(int hx) { return (hx >> 2) & 3; }" in s
    assert "GP_DEV int h4_stg_key(int px) { return ((px >> 1) & 3) | ((px & 1) << 2); }" in s
    key = lambda hx: (hx >> 2) & 3  # noqa: E731
    # halo: LDS row = hy * 34 + hx, hx = (l & 31) + kx, logical slot 2 kk + (l >> 5)
    for hy in range(18):
        for kx in range(3):
            for kk in (0, 1):
                assert conflicts(lambda l: (hy * 34 + (l & 31) + kx) * 4 + ((2 * kk + (l >> 5)) ^ key((l & 31) + kx))) == 0, (hy, kx, kk)
    # (an unswizzled 64-byte-row image would be 4-way: the reason the key exists)
    assert conflicts(lambda l: ((l & 31)) * 4 + (l >> 5)) > 0
    # weight tile: 32-row blocks at multiples of 32 rows, key of the row
    for base in range(0, 128, 32):
        for kk in (0, 1):
            assert conflicts(lambda l: (base + (l & 31)) * 4 + ((2 * kk + (l >> 5)) ^ key(base + (l & 31)))) == 0
    # epilogue staging block [32 px][8 units]: accumulator writes (lane = pixel l & 31, unit 2 g + (l >> 5)) and the row read-back
    # (lane = pixel (l >> 2) (+ 16), units 2 (l & 3) and 2 (l & 3) + 1)
    f = lambda px: ((px >> 1) & 3) | ((px & 1) << 2)  # noqa: E731
    for g in range(4):
        assert write_conflicts(lambda l: (l & 31) * 8 + ((2 * g + (l >> 5)) ^ f(l & 31))) == 0
    for add in (0, 16):
        for h in (0, 1):
            assert conflicts(lambda l: ((l >> 2) + add) * 8 + ((2 * (l & 3) + h) ^ f((l >> 2) + add))) == 0


def test_conv_halo5_layouts_are_conflict_free():
    """conv_halo5.hip: 64-byte halo / weight rows read by v_mfma_f32_16x16x32 fragments (lane = row a + 16 q, q = 16-byte slot), 128-byte
    epilogue staging rows written per (pixel, 8-channel group) and read back per (pixel, 8-channel slot)."""
    text = src("conv_halo5.hip")
    assert "return 3 * ((hx >> 2) & 1);" in text and "return 3 * ((row >> 3) & 1);" in text
    assert "return ((px >> 1) & 1) | ((px & 1) << 1) | (px & 4);" in text
    key = lambda hx: 3 * ((hx >> 2) & 1)  # noqa: E731
    for hy in range(18):
        for kx in range(3):
            assert conflicts(lambda l: (hy * 18 + (l & 15) + kx) * 4 + ((l >> 4) ^ key((l & 15) + kx))) == 0, (hy, kx)
    wkey = lambda r: 3 * ((r >> 3) & 1)  # noqa: E731
    for wn in range(2):
        for i in range(4):
            row = lambda a: wn * 64 + 32 * (i >> 1) + 4 * (i & 1) + 8 * (a >> 2) + (a & 3)  # noqa: E731
            assert conflicts(lambda l: row(l & 15) * 4 + ((l >> 4) ^ wkey(row(l & 15)))) == 0, (wn, i)
    f = lambda px: ((px >> 1) & 1) | ((px & 1) << 1) | (px & 4)  # noqa: E731
    for ib in (0, 1):
        assert write_conflicts(lambda l: (l & 15) * 8 + ((2 * (l >> 4) + ib) ^ f(l & 15))) == 0
    for t in (0, 1):
        assert conflicts(lambda l: (l >> 2) * 8 + ((2 * (l & 3) + t) ^ f(l >> 2))) == 0
    # the plain keys conflict: (hx >> 2) & 3 on the halo rows, px & 7 on the staging read-back
    assert sum(conflicts(lambda l: ((l & 15) + kx) * 4 + ((l >> 4) ^ ((((l & 15) + kx) >> 2) & 3))) for kx in range(3)) > 0
    assert sum(conflicts(lambda l: (l >> 2) * 8 + ((2 * (l & 3) + t) ^ ((l >> 2) & 7))) for t in (0, 1)) > 0


def test_conv_halo6_layouts_are_conflict_free():
    """conv_halo6.hip (Winograd F(2,3) along x): V rows [position][halo row][column pair] of 64 bytes read as 16 consecutive rows starting at a
    multiple of 8; weight rows like conv_halo5's; staging block [2 rows x 16 px][8 units] written per (patch, 8-channel group) for two pixels."""
    text = src("conv_halo6.hip")
    assert "return 3 * ((row >> 2) & 1);" in text and "return 3 * ((row >> 3) & 1);" in text and "return (px >> 1) & 7;" in text
    for base in range(0, 4 * 18 * 8, 8):
        assert conflicts(lambda l: (base + (l & 15)) * 4 + ((l >> 4) ^ (3 * (((base + (l & 15)) >> 2) & 1)))) == 0, base
    for pp in range(2):
        for wn in range(2):
            for i in range(4):
                row = lambda a: pp * 128 + wn * 64 + 32 * (i >> 1) + 4 * (i & 1) + 8 * (a >> 2) + (a & 3)  # noqa: E731
                assert conflicts(lambda l: row(l & 15) * 4 + ((l >> 4) ^ (3 * ((row(l & 15) >> 3) & 1)))) == 0
    f = lambda px: (px >> 1) & 7  # noqa: E731
    wpx = lambda l, e: ((l & 15) >> 3) * 16 + 2 * (l & 7) + e  # noqa: E731
    for e in (0, 1):
        for u in (0, 1):
            assert write_conflicts(lambda l: wpx(l, e) * 8 + ((2 * (l >> 4) + u) ^ f(wpx(l, e)))) == 0
    for rb in (0, 1):
        for t in (0, 1):
            assert conflicts(lambda l: (rb * 16 + (l >> 2)) * 8 + ((2 * (l & 3) + t) ^ f(rb * 16 + (l >> 2)))) == 0
