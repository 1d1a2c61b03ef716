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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genpercept_amd", "csrc")

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


def test_sources_still_use_these_swizzles():
    assert "return slot ^ (row & 7);" in src("common.h")
    assert "0x4016642254ull" in src("conv_halo.hip") and "(hx & 7)" in src("conv_halo.hip")
    assert "((slot ^ ((row >> 1) & 7)) << 4)" in src("attention.hip")
    assert "return row * 1024 + ((slot ^ (row & 15)) << 4);" in src("attention.hip")
    assert "return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);" in src("attention.hip")
    for f in ("igemm.hip", "conv_halo.hip", "pgemm.hip"):
        assert re.search(r"\(\(a15 >> 1\) & 1\) \| \(\(\(a15 >> 2\) & 1\) << 1\) \| \(\(\(a15 >> 3\) & 1\) << 2\)", src(f)), f


def test_pixel_tile_any_16_row_window_is_conflict_free():
    # MFMA 16x16x32 B/A fragment: lane (a = lane & 15, q = lane >> 4) reads row r0 + a, logical slot 4 kk + q
    for r0 in range(0, 64):
        for kk in (0, 1):
            assert conflicts(lambda l: (r0 + (l & 15)) * 8 + ((4 * kk + (l >> 4)) ^ ((r0 + (l & 15)) & 7))) == 0, (r0, kk)
    # the older (row >> 1) & 7 key is only conflict-free for windows starting at multiples of 4 (what cost the first halo kernel 28 %)
    bad = sum(conflicts(lambda l: (r0 + (l & 15)) * 8 + ((l >> 4) ^ (((r0 + (l & 15)) >> 1) & 7))) for r0 in range(1, 16, 2))
    assert bad > 0


def test_conv_halo_keys_are_conflict_free_for_every_tap():
    # 18-wide halo: LDS row = hy * 18 + hx, hx = a + kx, key = hx & 7
    for hy in range(18):
        for kx in range(3):
            for kk in (0, 1):
                assert conflicts(lambda l: (hy * 18 + (l & 15) + kx) * 8 + ((4 * kk + (l >> 4)) ^ (((l & 15) + kx) & 7))) == 0
    # 10-wide halo of the x2-upsample conv: hx = ((a + kx - 1) >> 1) + 1 (lane pairs share a column), key = nibble table
    table = [(0x4016642254 >> (4 * hx)) & 7 for hx in range(10)]
    assert table == [4, 5, 2, 2, 4, 6, 6, 1, 0, 4]
    for hy in range(10):
        for kx in range(3):
            for kk in (0, 1):
                hx = lambda a: ((a + kx - 1) >> 1) + 1  # noqa: E731
                assert conflicts(lambda l: (hy * 10 + hx(l & 15)) * 8 + ((4 * kk + (l >> 4)) ^ table[hx(l & 15)])) == 0
    # (hx & 7 would conflict there: that is why the table exists)
    assert sum(conflicts(lambda l: (((l & 15) + kx - 1) >> 1) * 8 + 8 + ((l >> 4) ^ (((((l & 15) + kx - 1) >> 1) + 1) & 7))) for kx in range(3)) > 0


def test_weight_tile_swizzle_is_conflict_free():
    # weight fragment i of a pair: MFMA row a -> tile row 32 ip + 8 (a >> 2) + 4 (i & 1) + (a & 3); key = b1 | b3 << 1 | b4 << 2 of the row
    key = lambda r: ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | (((r >> 4) & 1) << 2)  # noqa: E731
    for ip in range(4):
        for h in (0, 1):
            for kk in (0, 1):
                row = lambda a: 32 * ip + 8 * (a >> 2) + 4 * h + (a & 3)  # noqa: E731
                assert conflicts(lambda l: row(l & 15) * 8 + ((4 * kk + (l >> 4)) ^ key(row(l & 15)))) == 0, (ip, h, kk)
    # what the kernels compute per lane (xr_w from a15) is that key for every fragment of the pair
    for a in range(16):
        xr_w = ((a >> 1) & 1) | (((a >> 2) & 1) << 1) | (((a >> 3) & 1) << 2)
        for ip in range(4):
            for h in (0, 1):
                assert key(32 * ip + 8 * (a >> 2) + 4 * h + (a & 3)) == xr_w
    # DMA side: a wave writes 8 rows (lane >> 3) of group g = wave + NW * i; chunk_w uses ((lane >> 4) & 1) | (wave & 1) << 1 | ((wave >> 1) & 1) << 2
    for nw in (4, 8):
        for wave in range(nw):
            for i in range(4):
                for lane in range(64):
                    r = (wave + nw * i) * 8 + (lane >> 3)
                    assert key(r) == (((lane >> 4) & 1) | ((wave & 1) << 1) | (((wave >> 1) & 1) << 2))


def test_attention_tile_swizzle_is_conflict_free_for_32_row_fragments():
    # 32x32x16 MFMA A/B fragment: lane (r = lane & 31, half hh = lane >> 5) reads row r0 + r, slot 2 ks + hh; key = (row >> 1) & 7
    for r0 in (0, 32):
        for ks in range(4):
            assert conflicts(lambda l: (r0 + (l & 31)) * 8 + ((2 * ks + (l >> 5)) ^ (((r0 + (l & 31)) >> 1) & 7))) == 0
    # row & 7 (the conv tiles' key) is NOT conflict-free for this pattern
    assert sum(conflicts(lambda l: (l & 31) * 8 + ((2 * ks + (l >> 5)) ^ ((l & 31) & 7))) for ks in range(4)) > 0


def test_hd512_attention_tiles_are_conflict_free():
    # K tile [32 keys][512 d]: 64 slots of 16 bytes per row; lane (row = lane & 31, hh) reads slot 2 ks + hh, ks = 0..31
    for ks in range(32):
        assert conflicts(lambda l: (l & 31) * 64 + ((2 * ks + (l >> 5)) ^ ((l & 31) & 15))) == 0, ks
    # without the swizzle every row of a lane group lands on the same 16-byte bank column
    assert conflicts(lambda l: (l & 31) * 64 + (l >> 5)) > 0
    # V^T tile [512 d][32 keys]: 4 slots per row; lane (row = 32 blk + lane & 31, hh) reads slot 2 j + hh
    for blk in range(16):
        for j in range(2):
            assert conflicts(lambda l: (32 * blk + (l & 31)) * 4 + ((2 * j + (l >> 5)) ^ (((32 * blk + (l & 31)) >> 2) & 3))) == 0, (blk, j)
    assert conflicts(lambda l: (l & 31) * 4 + (l >> 5)) > 0
    # the key order of the staged K rows: LDS row i holds key pi(i) = i with bits 2 and 3 swapped, so accumulator register r of half hh
    # (LDS row (r & 3) + 4 hh + 8 (r >> 2)) belongs to key 8 hh + (r & 7) + 16 (r >> 3): 8 consecutive keys per 16-key k-step and half
    pi = lambda i: (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)  # noqa: E731
    assert sorted(pi(i) for i in range(32)) == list(range(32))
    for hh in (0, 1):
        for r in range(16):
            assert pi((r & 3) + 4 * hh + 8 * (r >> 2)) == 8 * hh + (r & 7) + 16 * (r >> 3)


# ds_write_b128: served in contiguous groups of 8 lanes; bank = (a / 4) % 32, i.e. a 128-byte bank row of eight 16-byte units
def write_conflicts(addr16_of_lane):
    extra = 0
    for g0 in range(0, 64, 8):
        by_bank = {}
        for lane in range(g0, g0 + 8):
            a = addr16_of_lane(lane)
            by_bank.setdefault(a % 8, set()).add(a)
        extra += sum(len(v) - 1 for v in by_bank.values())
    return extra
