"""The divisions by constants that the ORB kernels do with one 24-bit multiply and a shift (a compiler-generated division by a
constant is a quarter-rate v_mul_hi plus fix-ups): every constant quoted from the source, every identity checked over the
whole range the kernel can produce.  (CPU only: integer arithmetic.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORB = open(os.path.join(ROOT, "gslam_amd", "csrc", "orb.hip")).read()
QT = open(os.path.join(ROOT, "gslam_amd", "csrc", "orb_quadtree.hip")).read()


def _exact(mult, shift, divisor, n_end):
    assert n_end * mult < 1 << 32 and mult < 1 << 24 and n_end <= 1 << 24  # fits the 24-bit multiplier and 32 bits
    return all((n * mult) >> shift == n // divisor for n in range(n_end))


def test_tile_load_row_of_item_is_item_div_6():
    assert "__umul24((uint32_t)i, 10923u) >> 16" in ORB and "kTileW / 16 == 6" in ORB
    tile_h = int(re.search(r"constexpr int kTileH = (\d+);", ORB).group(1))
    assert _exact(10923, 16, 6, tile_h * 6)


def test_pass1_row_of_item_is_item_div_17():
    assert "(item * 3856) >> 16 == item / 17" in ORB
    score_h = int(re.search(r"constexpr int kScoreH = (\d+);", ORB).group(1))
    assert _exact(3856, 16, 17, max(score_h * 17 + 256, 3855))  # (items past the window are computed, then masked)


def test_pass2_row_of_position_is_position_div_68():
    assert "15421u) >> 20" in ORB and "kScoreW == 68" in ORB
    assert _exact(15421, 20, 68, 68 * 72)


def test_quadtree_tile_row_is_index_div_11():
    assert "__umul24((uint32_t)idx, 5958u) >> 16" in QT and "constexpr int kWRowDw = 11;" in QT
    assert _exact(5958, 16, 11, (32 + 6) * 11)


def test_quadtree_pixel_row_is_pixel_div_cell_width():
    assert "(65536u + (uint32_t)cw - 1u) / (uint32_t)cw" in QT
    for cw in range(1, 33):
        inv = (65536 + cw - 1) // cw
        assert inv <= 1 << 16 and all((p * inv) >> 16 == p // cw for p in range(32 * 32))
